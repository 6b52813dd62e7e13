"""Two library builds against each other on ONE box (tuning aid, not collected by pytest): tools/step_ab.py is run in child
processes that alternate between the libraries (A B A B ...), so box-to-box differences (+-5 % on this pool) and drift
cancel.  python tools/lib_ab.py <libA.so> <libB.so> [rounds]   (names relative to slam-sam_amd/)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(rounds):
    for lib in libs:
        env = dict(os.environ, NDT_HIP_LIB=os.path.join(ROOT, "slam-sam_amd", lib), NDT_STEP_AB_DEFERRED="1")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_ab.py"), "%s #%d" % (lib, r)], env=env, capture_output=True, text=True, timeout=300)
        out = [ln for ln in p.stdout.splitlines() if "step" in ln]
        print(out[-1] if out else "FAILED %s: %s" % (lib, p.stderr[-400:]), flush=True)
