"""Kernel-level timing on the C3 workload (bring-up / tuning aid, not collected by pytest)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
tag = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
builds = []
for _ in range(6):
    t = time.perf_counter(); ndt.setInputTarget(cfg["target"]); builds.append(ndt.getGridInfo()["ms_build"])
ndt.setInputSource(cfg["source"])
for _ in range(3): ndt.align(cfg["guess"])
ts = []
for _ in range(10):
    t = time.perf_counter(); ndt.align(cfg["guess"]); ts.append((time.perf_counter() - t) * 1e3)
r = ndt.getResult()
ndt.enableKernelTiming(True)
t0 = ndt.getTiming()
for _ in range(10): ndt.align(cfg["guess"])
t1 = ndt.getTiming()
n = t1["n_timed_evals"] - t0["n_timed_evals"]
k_us = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / n
p = r["pose"]
t0 = ndt.getTiming()
for _ in range(20): ndt.evalDerivatives(np.tile(p, (20, 1)))
t1 = ndt.getTiming()
b_us = 1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"]) / 20
print("%s build %.3f ms | align %.3f ms (min %.3f) it %d ev %d -> %.1f us/eval wall | k_derivatives %.2f us | batch20 %.2f us/pose | score %.9f"
      % (tag, np.median(builds), np.median(ts), min(ts), r["iterations"], r["n_evaluations"],
         1e3 * np.median(ts) / r["n_evaluations"], k_us, b_us, r["score"]), flush=True)
