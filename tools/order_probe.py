"""Does the order of the source points matter to k_derivatives? (tuning aid, not collected by pytest)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
src = cfg["source"]; G = cfg["guess"]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
rng = np.random.default_rng(0)
w = (src @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
def cellkey(p, leaf):
    q = np.floor(p / leaf).astype(np.int64); q -= q.min(0)
    return (q[:, 2] * 4096 + q[:, 1]) * 4096 + q[:, 0]
def morton(p, leaf):
    q = np.floor(p / leaf).astype(np.int64); q -= q.min(0)
    k = np.zeros(len(p), np.int64)
    for b in range(12):
        for a in range(3):
            k |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return k
orders = {
    "as given (scan order)": np.arange(len(src)),
    "shuffled": rng.permutation(len(src)),
    "sorted by map cell at the guess": np.argsort(cellkey(w, 0.5), kind="stable"),
    "sorted by sensor-frame cell (0.5 m)": np.argsort(cellkey(src, 0.5), kind="stable"),
    "sorted by sensor-frame morton (0.5 m)": np.argsort(morton(src, 0.5), kind="stable"),
    "sorted by sensor-frame morton (0.25 m)": np.argsort(morton(src, 0.25), kind="stable"),
}
for name, o in orders.items():
    ndt.setInputSource(np.ascontiguousarray(src[o]))
    for _ in range(3): ndt.align(G)
    ts = []
    for _ in range(10):
        t = time.perf_counter(); ndt.align(G); ts.append((time.perf_counter() - t) * 1e3)
    r = ndt.getResult()
    ndt.enableKernelTiming(True)
    t0 = ndt.getTiming()
    for _ in range(5): ndt.align(G)
    t1 = ndt.getTiming()
    ndt.enableKernelTiming(False)
    k_us = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
    print("%-40s align %.3f ms  it %d ev %d  k_derivatives %.2f us  score %.9f" % (name, np.median(ts), r["iterations"], r["n_evaluations"], k_us, r["score"]), flush=True)
