"""Build-time guard: the derivative kernel must stay inside the register budget that lets four
waves share a SIMD (<= 128 VGPRs, the budget `__launch_bounds__(1024)` imposes) WITHOUT spilling
to scratch.  A round-2 restructuring that inlined two final-sum variants into one kernel compiled
to 82 spilled VGPRs / 168 bytes of scratch per lane and ran 23 us instead of 16.6 -- no test
noticed, the A/B did.  hipcc cross-compiles gfx950 here; no GPU needed."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_derivative_kernels_do_not_spill(tmp_path):
    src = os.path.join(ROOT, "slam-sam_amd", "csrc", "ndt_derivs.hip")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", str(tmp_path / "d.o")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    usage, name = {}, None
    for ln in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            name = m.group(1)
        m = re.search(r"(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and name and "k_derivatives" in name:
            usage.setdefault(name, {})[m.group(1).split(" ")[0]] = int(m.group(2))
    assert len(usage) >= 32
    for name, u in usage.items():
        tpl = re.search(r"k_derivativesILb(\d)ELi(\d)ELi(\d)E", name)
        assert tpl, name
        assert u["VGPRs"] <= 128 and u["Occupancy"] >= 4, (name, u)
        # no instantiation spills (round 3: three records in flight instead of four; the Gauss-Newton x DIRECT7
        # kernel -- svn_ndt's default engine -- carried 12 bytes of scratch per lane until then)
        # (round 5: with the 64 sums of a finishing wave scoped to its branch -- no merge with the other waves' undefined
        # values -- the multi-grid kernels lost their 92 / 120 bytes too: all 84 instantiations carry none)
        assert u["ScratchSize"] == 0, (name, u)
