"""What the FIRST evaluation of an align costs (an ordinary launch) against the later, pre-launched ones (C3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3(); hip = pkg.ranks.Hip(0)
tp = [hip.upload(cfg["target"][:, a]) for a in range(3)]; sp = [hip.upload(cfg["source"][:, a]) for a in range(3)]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTargetDevice(tp[0], tp[1], tp[2], len(cfg["target"])); ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], len(cfg["source"]))
g = pkg.ColMajor4f(cfg["guess"])
def med(f, n=200):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e6 * np.median(ts)
full = med(lambda: ndt.align(g, return_transform=False)); ev = ndt.getNumEvaluations()
ndt.setMaximumIterations(0)
one = med(lambda: ndt.align(g, return_transform=False)); ev1 = ndt.getNumEvaluations()
print("align: %.1f us for %d evaluations; an align of max_iterations = 0: %.1f us for %d evaluations" % (full, ev, one, ev1))
print("-> later evaluations %.2f us each, the first %.1f us" % ((full - one) / max(ev - ev1, 1), one - (ev1 - 1) * (full - one) / max(ev - ev1, 1)))
