"""k_bucket_pass / k_bucket_leaves against the size of the target (tuning aid, not collected by pytest): the C3 map thinned
to 1/STRIDE of its points, 40 steady-state builds.  Run under rocprofv3 --kernel-trace --stats for the per-kernel times
(what would half-size buckets buy?).  Usage: python tools/build_size_probe.py STRIDE"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
stride = int(sys.argv[1]) if len(sys.argv) > 1 else 1
hip = pkg.ranks.Hip(0)
cfg = S.config_c3()
tgt = np.ascontiguousarray(cfg["target"][::stride])
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tp = [hip.upload(tgt[:, a]) for a in range(3)]
hip.synchronize()
for _ in range(45):
    ndt.setInputTargetDevice(tp[0], tp[1], tp[2], len(tgt))
gi = ndt.getGridInfo()
print("stride %d: %d points, %d leaves, build %.1f us (host timer)" % (stride, len(tgt), gi["n_leaves"], 1e3 * gi["ms_build"]), flush=True)
