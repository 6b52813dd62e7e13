"""A/B of the build's launch shapes on C3 (tuning aid, not collected by pytest): every knob set runs
in its own process (the knobs are read once), interleaved twice.  Prints build wall / device time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
torch.cuda.init(); dev = torch.device("cuda:0")
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tgt = [torch.from_numpy(np.ascontiguousarray(cfg["target"][:, a])).to(dev) for a in range(3)]
torch.cuda.synchronize()
tp = [t.data_ptr() for t in tgt]; nt = len(cfg["target"])
W, D = [], []
for i in range(60):
    t0 = time.perf_counter(); ndt.setInputTargetDevice(tp[0], tp[1], tp[2], nt); t1 = time.perf_counter()
    if i >= 10: W.append(t1 - t0); D.append(ndt.getGridInfo()["ms_build"])
gi = ndt.getGridInfo()
print("%%-44s build wall %%.1f us  device %%.1f us  (leaves %%d)" %% (sys.argv[1], 1e6 * float(np.median(W)), 1e3 * float(np.median(D)), gi["n_leaves"]), flush=True)
''' % ROOT
SETS = [
    ("build timed by events (device time)", {"NDT_BUILD_EVENTS": "1"}),
    ("no events (ms_build = wall)", {"NDT_BUILD_EVENTS": "0"}),
]
for rep in range(2):
    for name, env in SETS:
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD, name], env=e, capture_output=True, text=True, timeout=300)
        out = [l for l in r.stdout.splitlines() if "build wall" in l]
        print(out[0] if out else "FAILED %s rc=%d %s" % (name, r.returncode, r.stderr[-400:]), flush=True)
