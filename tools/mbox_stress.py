"""Stress of the pose hand-over to pre-launched kernels under host CPU contention (debugging aid, not
collected by pytest): a background thread keeps the BLAS worker threads busy while the main thread
aligns; every failure is reported with the engine's counters.  Usage: mbox_stress.py [aligns [source points]]
(131072 source points = the launch shape of C2 / C5: 256 point blocks, no dedicated summing block)"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
if len(sys.argv) > 2:
    cfg["source"] = cfg["source"][:int(sys.argv[2])]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
g = pkg.ColMajor4f(cfg["guess"])
ndt.align(g); ref = ndt.getResult()
stop = False
def hog():
    a = np.random.default_rng(0).normal(size=(600, 600))
    while not stop:
        a = (a @ a) / 600.0
        b = np.concatenate([a, a + 1.0])
hogs = [threading.Thread(target=hog, daemon=True) for _ in range(3)]
for t in hogs: t.start()
bad = fails = 0
t0 = time.perf_counter()
for i in range(n):
    try:
        ndt.align(g, return_transform=False)
    except pkg.NdtError as e:
        fails += 1
        print("FAIL at align %d: %s | counters (used, quit, timeouts) %s" % (i, e, ndt.prelaunchCounters()), flush=True)
        if fails > 5: break
        continue
    if ndt.getFinalNumIteration() != ref["iterations"]:
        bad += 1; print("MISMATCH iterations at", i, flush=True)
    elif i % 53 == 0:
        r = ndt.getResult()
        if r["score"] != ref["score"] or not np.array_equal(r["T"], ref["T"]):
            bad += 1; print("MISMATCH result at", i, r["score"], ref["score"], flush=True)
stop = True
print("%d aligns under host contention in %.1f s: %d failures, %d mismatches, counters (used, quit, timeouts) %s"
      % (n, time.perf_counter() - t0, fails, bad, ndt.prelaunchCounters()), flush=True)
