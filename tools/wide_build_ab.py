"""Build of the 4M-point C3-wide map: big-tile fused passes against the classic passes (tuning aid)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
torch.cuda.init(); dev = torch.device("cuda:0")
cfg = S.config_c3_wide()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tgt = [torch.from_numpy(np.ascontiguousarray(cfg["target"][:, a])).to(dev) for a in range(3)]
torch.cuda.synchronize()
tp = [t.data_ptr() for t in tgt]; nt = len(cfg["target"])
W = []
for i in range(30):
    t0 = time.perf_counter(); ndt.setInputTargetDevice(tp[0], tp[1], tp[2], nt); t1 = time.perf_counter()
    if i >= 5: W.append(t1 - t0)
import hashlib
L = ndt.getLeaves(); h = hashlib.sha256()
for k in ("cell", "count", "mean", "icov"): h.update(np.ascontiguousarray(L[k]).tobytes())
print("%%-28s %%d points: build wall %%.1f us, leaves %%d, hash %%s, fallbacks %%d" %% (sys.argv[1], nt, 1e6 * float(np.median(W)), len(L["cell"]), h.hexdigest()[:12], ndt.buildCounters()[0]), flush=True)
''' % ROOT
for rep in range(2):
    for name, env in (("fused (16384-pair tiles)", {}), ("classic passes", {"NDT_FUSED_SORT": "0"})):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([sys.executable, "-c", CHILD, name], env=e, capture_output=True, text=True, timeout=600)
        out = [l for l in r.stdout.splitlines() if "build wall" in l]
        print(out[0] if out else "FAILED %s rc=%d %s" % (name, r.returncode, r.stderr[-600:]), flush=True)
