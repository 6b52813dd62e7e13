"""Wall time per derivative evaluation inside align() for a small and the C3 source (tuning aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
tag = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
big = cfg["target"] @ np.linalg.inv(cfg["gt"])[:3, :3].T.astype(np.float32) + np.linalg.inv(cfg["gt"])[:3, 3].astype(np.float32)
for n in (1000, 20000, 200000, 500000, 1000000):
    src = cfg["source"][:: max(1, 200000 // n)][:n] if n <= 200000 else big[:n]
    ndt.setInputSource(np.ascontiguousarray(src, dtype=np.float32))
    for _ in range(5): ndt.align(cfg["guess"])
    ts, ev = [], 0
    for _ in range(30):
        t = time.perf_counter(); ndt.align(cfg["guess"]); ts.append(time.perf_counter() - t)
        ev = ndt.getResult()["n_evaluations"]
    print("%s n=%6d  %.2f us/eval wall (median align %.3f ms, %d evals)" %
          (tag, n, 1e6 * np.median(ts) / ev, 1e3 * np.median(ts), ev), flush=True)
