"""Distribution of the host hand-off's call times over many scans (C3, PointXYZI host clouds): a worker that is not on a
CPU when its piece is due shows as a slow call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
def xyzi(a):
    o = np.zeros((len(a), 8), np.float32); o[:, :3] = a; o[:, 3] = 1.0; return o
t, s = xyzi(cfg["target"]), xyzi(cfg["source"])
g = pkg.ColMajor4f(cfg["guess"])
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(5):
    ndt.setInputTarget(t); ndt.setInputSource(s); ndt.align(g, return_transform=False)
A = np.zeros((K, 3))
for i in range(K):
    t0 = time.perf_counter(); ndt.setInputTarget(t)
    t1 = time.perf_counter(); ndt.setInputSource(s)
    t2 = time.perf_counter(); ndt.align(g, return_transform=False)
    A[i] = (t1 - t0, t2 - t1, time.perf_counter() - t2)
A *= 1e3
for k, name in enumerate(("setInputTarget", "setInputSource", "align")):
    v = A[:, k]
    print("%-15s min %.3f  p10 %.3f  median %.3f  p90 %.3f  p99 %.3f  max %.3f  mean %.3f ms" % (name, v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), v.mean()))
print("scan: median %.3f mean %.3f ms; threads %d" % (np.median(A.sum(1)), A.sum(1).mean(), ndt.getHandoffTiming()["target"]["threads"]))
