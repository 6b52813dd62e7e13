"""VGPRs / scratch / LDS of every k_derivatives instantiation (hipcc cross-compiles; no GPU).  python tools/kernel_resources.py"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "slam-sam_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "ndt_derivs.hip")
pat = sys.argv[2] if len(sys.argv) > 2 else "k_derivatives"
with tempfile.TemporaryDirectory() as d:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(d, "d.o")] + sys.argv[3:], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-3000:]
name, rows = None, {}
for ln in p.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", ln)
    if m: name = m.group(1)
    m = re.search(r"(VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", ln)
    if m and name and pat in name: rows.setdefault(name, {})[m.group(1).split(" ")[0]] = int(m.group(2))
for nm, u in rows.items():
    t = re.search(r"k_derivativesILb(\d)ELi(\d)ELi(\d)ELb(\d)", nm)
    tag = "batch%s mode%s nb%s mbox%s" % t.groups() if t else nm[:60]
    print("%-28s VGPR %3d SGPR %3d scratch %3d occ %d LDS %d" % (tag, u.get("VGPRs", -1), u.get("SGPRs", -1), u.get("ScratchSize", -1), u.get("Occupancy", -1), u.get("LDS", -1)))
