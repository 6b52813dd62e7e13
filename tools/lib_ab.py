"""Two library builds against each other on ONE box (tuning aid, not collected by pytest): tools/step_ab.py is run in child
processes that alternate between the libraries (A B A B ...), so box-to-box differences (+-5 % on this pool) and drift
cancel.  python tools/lib_ab.py <libA.so>[:VAR=val[,VAR=val]] <libB.so>[:...] [rounds]   (names relative to slam-sam_amd/;
the variables are the historical tuning names, which step_ab.py turns into ndt_set_tuning)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args.pop()) if args and args[-1].isdigit() else 3
libs = args
for r in range(rounds):
    for lib in libs:
        name, _, extra = lib.partition(":")
        env = dict(os.environ, NDT_HIP_LIB=os.path.join(ROOT, "slam-sam_amd", name), NDT_STEP_AB_DEFERRED="1")
        env.update(kv.split("=", 1) for kv in extra.split(",") if kv)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_ab.py"), "%s #%d" % (lib, r)], env=env, capture_output=True, text=True, timeout=300)
        out = [ln for ln in p.stdout.splitlines() if "step" in ln]
        print(out[-1] if out else "FAILED %s: %s" % (lib, p.stderr[-400:]), flush=True)
