"""k_derivatives on the C2 map (128 x 1024 scan into the scan before it, 1.0 m voxels) vs source size; NDT_DERIV_BLOCK
from the environment selects the block shape (tuning aid).  python tools/c2_shape_sweep.py [tag]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
tag = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = S.config_c2()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
src, T = cfg["source"], cfg["gt"]
ndt.enableKernelTiming(True)
out = []
for n in (32768, 65536, 100000, 120000, 131072):
    ndt.setInputSource(src[:n]); ndt.setParams(max_iterations=0)
    for _ in range(3): ndt.align(T)
    t0 = ndt.getTiming()
    for _ in range(40): ndt.align(T)
    t1 = ndt.getTiming()
    k = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
    out.append("%d: %.2f us" % (n, k))
ndt.enableKernelTiming(False)
ndt.setInputSource(src); ndt.setParams(max_iterations=35)
import time
for _ in range(5): ndt.align(cfg["guess"])
ts = []
for _ in range(20):
    t = time.perf_counter(); ndt.align(cfg["guess"]); ts.append(time.perf_counter() - t)
r = ndt.getResult()
print("%-18s %s | align %.3f ms, %d ev -> %.2f us/eval wall" % (tag, "  ".join(out), 1e3 * np.median(ts), r["n_evaluations"], 1e6 * np.median(ts) / r["n_evaluations"]), flush=True)
