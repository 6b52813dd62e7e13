"""C3 vs C3-wide: the device-resident step and the derivative kernel, for PMC / rocprof passes
(tuning aid, not collected by pytest).  `python tools/wide_bench.py [c3|wide] [shuffle]`;
`shuffle` permutes the source (no scan coherence at all)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
which = sys.argv[1] if len(sys.argv) > 1 else "wide"
cfg = S.config_c3_wide() if which == "wide" else S.config_c3()
src = cfg["source"]
if "shuffle" in sys.argv:
    src = src[np.random.default_rng(0).permutation(len(src))]
if "sorted" in sys.argv:   # by map voxel at the guess: the ordering north_star's LDS staging presumes
    p = S.transform(cfg["guess"], src)
    key = np.floor(p.astype(np.float64) / 0.5).astype(np.int64)
    key -= key.min(0)
    src = src[np.lexsort((key[:, 0], key[:, 1], key[:, 2]))]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
B = []
for _ in range(6):
    ndt.setInputTarget(cfg["target"]); B.append(ndt.getGridInfo()["ms_build"])
ndt.setInputSource(np.ascontiguousarray(src))
for _ in range(3): ndt.align(cfg["guess"])
ts = []
for _ in range(10):
    t = time.perf_counter(); ndt.align(cfg["guess"]); ts.append(time.perf_counter() - t)
r = ndt.getResult()
ndt.enableKernelTiming(True); t0 = ndt.getTiming()
for _ in range(10): ndt.align(cfg["guess"])
t1 = ndt.getTiming()
k = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
gi = ndt.getGridInfo()
nbar = r["n_pairs"] / len(src)
algo = len(src) * (12 + 28 + 48 * nbar)
print("%s %s: leaves %d, build %.3f ms | align %.3f ms it %d ev %d -> %.2f us/eval wall | k_derivatives %.2f us, nbar %.2f, algorithmic %.2f TB/s (%.1f %% of 8) | err %.4f m"
      % (which, " ".join(sys.argv[2:]), gi["n_leaves"], np.median(B[1:]), 1e3 * np.median(ts), r["iterations"], r["n_evaluations"],
         1e6 * np.median(ts) / r["n_evaluations"], k, nbar, algo / (k * 1e-6) / 1e12, 100 * algo / (k * 1e-6) / 8e12,
         S.pose_error(r["T"], cfg["gt"])[0]), flush=True)
