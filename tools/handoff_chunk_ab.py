"""A/B of the asynchronous host hand-off with the target's partition launch under the transfer (ndt_tuning::handoff_chunk_pass = 1,
round 5) and behind it (0): C3 through setInputTarget / setInputSource / align on host PointXYZI clouds, one process, the two
variants in alternating blocks of K scans.    python tools/handoff_chunk_ab.py [K] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
def xyzi(a):
    out = np.zeros((len(a), 8), np.float32); out[:, :3] = a; out[:, 3] = 1.0
    return out
t, s = xyzi(cfg["target"]), xyzi(cfg["source"])
guess = pkg.ColMajor4f(cfg["guess"])
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setHandoffMode(pkg.HANDOFF_ASYNC)
def step():
    t0 = time.perf_counter(); ndt.setInputTarget(t)
    t1 = time.perf_counter(); ndt.setInputSource(s)
    t2 = time.perf_counter(); ndt.align(guess, return_transform=False)
    t3 = time.perf_counter()
    return t3 - t0, t1 - t0, t2 - t1, t3 - t2, ndt.getHandoffTiming()["ms_build_wait"] * 1e-3
for _ in range(30): step()
res = {0: [], 1: []}
for r in range(R):
    for v in (1, 0):
        pkg.set_tuning(handoff_chunk_pass=v)
        for _ in range(3): step()
        res[v] += [step() for _ in range(K)]
pkg.set_tuning(handoff_chunk_pass=1)
for v, name in ((0, "partition behind the transfer"), (1, "partition under the transfer ")):
    a = np.array(res[v]) * 1e3
    print("%s: ms/scan median %.3f mean %.3f | setInputTarget %.3f  setInputSource %.3f  align %.3f (of which waiting for transfer + build %.3f) | %d scans"
          % (name, np.median(a[:, 0]), a[:, 0].mean(), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), np.median(a[:, 4]), len(a)))
print("hand-off counters (builds partitioned under the transfer, launches):", ndt.handoffCounters())
