"""Derivative-kernel time vs source size on the C3 map and on the C3-wide map (tuning aid, not
collected by pytest).  `python tools/size_sweep.py [c3|wide|both]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, cfg in (("C3", S.config_c3() if which in ("c3", "both") else None),
                  ("C3-wide", S.config_c3_wide() if which in ("wide", "both") else None)):
    if cfg is None:
        continue
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    ndt.setInputTarget(cfg["target"])
    gi = ndt.getGridInfo()
    print("%s: %d leaves (%.1f MB of records), %d cells (%.1f MB index grid), build %.3f ms"
          % (name, gi["n_leaves"], gi["n_leaves"] * 80 / 1e6, gi["n_cells"], gi["n_cells"] * 4 / 1e6, gi["ms_build"]), flush=True)
    src = cfg["source"]
    T = cfg["gt"]
    rng = np.random.default_rng(0)
    big = np.concatenate([src + rng.normal(0, 0.01, src.shape).astype(np.float32) for _ in range(20)])
    ndt.enableKernelTiming(True)
    for n in (1000, 12500, 25000, 50000, 100000, 200000, 400000, 800000, 1600000, 4000000):
        ndt.setInputSource(big[:n])
        ndt.setParams(max_iterations=0)
        for _ in range(3): ndt.align(T)
        t0 = ndt.getTiming()
        for _ in range(30): ndt.align(T)
        t1 = ndt.getTiming()
        k = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"])
        r = ndt.getResult()
        nbar = r["n_pairs"] / n
        algo = n * (12 + 28 + 48 * nbar)
        print("%-8s n=%8d  k_derivatives %8.2f us  %6.2f ns/pt  nbar %.2f  algorithmic %.2f TB/s (%.1f%% of 8)"
              % (name, n, k, 1e3 * k / n, nbar, algo / (k * 1e-6) / 1e12, 100 * algo / (k * 1e-6) / 8e12), flush=True)
    ndt.close()
