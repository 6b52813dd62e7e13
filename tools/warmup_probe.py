"""Does the timed region of bench.py (W = 5 warm-up steps, then K = 20) start on a device that is still ramping up?
(tuning aid, not collected by pytest).  After an idle pause: per-step times of 60 consecutive C3 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
hip = pkg.ranks.Hip(0)
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
tp = [hip.upload(cfg["target"][:, a]) for a in range(3)]; sp = [hip.upload(cfg["source"][:, a]) for a in range(3)]
hip.synchronize()
g = pkg.ColMajor4f(cfg["guess"]); nt, ns = len(cfg["target"]), len(cfg["source"])
def step():
    t0 = time.perf_counter()
    ndt.setInputTargetDeviceDeferred(tp[0], tp[1], tp[2], nt); ndt.setInputSourceDeviceView(sp[0], sp[1], sp[2], ns)
    ndt.align(g, return_transform=False)
    return 1e3 * (time.perf_counter() - t0)
for _ in range(20): step()
for pause in (0.0, 0.5, 5.0, 20.0):
    time.sleep(pause)
    t = [step() for _ in range(80)]
    print("after %4.1f s idle: steps 0-4 %.3f | 5-24 %.3f | 25-44 %.3f | 45-79 %.3f ms (means); first five: %s"
          % (pause, np.mean(t[:5]), np.mean(t[5:25]), np.mean(t[25:45]), np.mean(t[45:]), " ".join("%.3f" % v for v in t[:5])), flush=True)
# a 10 Hz driver: one step every 100 ms
for rep in range(2):
    for _ in range(5): step()
    t = []
    for _ in range(40):
        time.sleep(0.1)
        t.append(step())
    print("one step every 100 ms: median %.3f  p10 %.3f  p90 %.3f ms" % (np.median(t), np.percentile(t, 10), np.percentile(t, 90)), flush=True)
