"""Timing-only ablations of k_derivatives (tuning aid, not collected by pytest): the C3 evaluation at the converged
pose as ordinary single-pose launches, HIP-event kernel time.  Run once per library variant (NDT_HIP_LIB);
the ablated kernels compute wrong values on purpose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
tag = sys.argv[1] if len(sys.argv) > 1 else ""
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
if os.environ.get("NDT_ABL_PACKED") == "1": ndt.setRecordFormat(pkg.RECORDS_PACKED48)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
import importlib
p = np.array(S.matrix_to_pose(cfg["gt"]) if hasattr(S, "matrix_to_pose") else importlib.import_module("oracle.oracle").matrix_to_pose(cfg["gt"]), np.float64)
for _ in range(20): ndt.evalDerivatives(p)
ndt.enableKernelTiming(True)
out = []
for rep in range(3):
    t0 = ndt.getTiming()
    for _ in range(200): e = ndt.evalDerivatives(p)
    t1 = ndt.getTiming()
    out.append(1e3 * (t1["ms_eval_kernel_total"] - t0["ms_eval_kernel_total"]) / (t1["n_timed_evals"] - t0["n_timed_evals"]))
print("%-34s k_derivatives %.2f %.2f %.2f us  (score %.6f, pairs %d)" % (tag, out[0], out[1], out[2], e[0]["score"], e[0]["n_pairs"]), flush=True)
