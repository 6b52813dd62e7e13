"""Host-cloud (PCIe-inclusive) path timing: setInputTarget / setInputSource from host memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tgt12 = np.ascontiguousarray(cfg["target"]); src12 = np.ascontiguousarray(cfg["source"])
tgt32 = np.zeros((len(tgt12), 8), np.float32); tgt32[:, :3] = tgt12
src32 = np.zeros((len(src12), 8), np.float32); src32[:, :3] = src12
def med(f, n=8):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(1e3 * (time.perf_counter() - t))
    return np.median(ts)
for name, t, s in (("packed xyz (12 B)", tgt12, src12), ("PointXYZI AoS (32 B)", tgt32, src32)):
    a = med(lambda: ndt.setInputTarget(t)); b = med(lambda: ndt.setInputSource(s)); c = med(lambda: ndt.align(cfg["guess"]))
    print("%-22s setInputTarget %.3f ms (device build %.3f)  setInputSource %.3f ms  align %.3f ms  -> ms/scan %.3f"
          % (name, a, ndt.getGridInfo()["ms_build"], b, c, a + b + c))
x, y, z = (np.ascontiguousarray(tgt12[:, k]) for k in range(3))
print("SoA target: %.3f ms" % med(lambda: ndt.setInputTargetSoA(x, y, z)))
