"""A/B of the target build on C3 (tuning aid, not collected by pytest): every knob set runs in its own
process (the knobs are read once), interleaved twice.  Prints build wall / device time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
hip = pkg.ranks.Hip(0)
cfg = getattr(S, "config_" + sys.argv[2])()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=float(sys.argv[3]), step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tp = [hip.upload(cfg["target"][:, a]) for a in range(3)]; nt = len(cfg["target"])
W, D = [], []
for i in range(60):
    t0 = time.perf_counter(); ndt.setInputTargetDevice(tp[0], tp[1], tp[2], nt); t1 = time.perf_counter()
    if i >= 10: W.append(t1 - t0); D.append(ndt.getGridInfo()["ms_build"])
gi = ndt.getGridInfo()
print("%%-60s build wall %%.1f us  ms_build %%.1f us  (leaves %%d, counters %%s)" %% (sys.argv[1], 1e6 * float(np.median(W)), 1e3 * float(np.median(D)), gi["n_leaves"], ndt.buildCounters()), flush=True)
''' % ROOT
CFGS = [("c3", "0.5"), ("c2", "1.0")]
SETS = [
    ("bucketed (2 launches), device time", {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "1"}),
] + ([("libndt_hip_base.so: bucketed, device time", {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "1", "NDT_HIP_LIB": os.path.join(ROOT, "slam-sam_amd", "libndt_hip_base.so")})]
     if os.path.exists(os.path.join(ROOT, "slam-sam_amd", "libndt_hip_base.so")) else []) + (
    [("%s: bucketed, device time" % nm, {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "1", "NDT_HIP_LIB": os.path.join(ROOT, "slam-sam_amd", nm)})
     for nm in ("libndt_hip_ab.so", "libndt_hip_prev.so") if os.path.exists(os.path.join(ROOT, "slam-sam_amd", nm))]) + [
    ("bucketed, 8192-point tiles, device time", {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "1", "NDT_BUCKET_TILE": "8192"}),
    ("bucketed, 2048-point tiles (where <= 256 tiles), device time", {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "1", "NDT_BUCKET_TILE": "2048"}),
    ("sort-based (8 launches), device time", {"NDT_BUILD_EVENTS": "1", "NDT_BUCKET_BUILD": "0"}),
    ("bucketed, wall", {"NDT_BUILD_EVENTS": "0", "NDT_BUCKET_BUILD": "1"}),
    ("sort-based, wall", {"NDT_BUILD_EVENTS": "0", "NDT_BUCKET_BUILD": "0"}),
]
for rep in range(2):
  for cfg, res in CFGS:
    for name, env in SETS:
        e = dict(os.environ); e.update(env)
        name = cfg + " " + name
        r = subprocess.run([sys.executable, "-c", CHILD, name, cfg, res], env=e, capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if "build wall" in l]
        print(out[0] if out else "FAILED %s rc=%d %s" % (name, r.returncode, r.stderr[-400:]), flush=True)
