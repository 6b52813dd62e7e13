import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
g = pkg.ColMajor4f(cfg["guess"])
ndt.setInputSource(np.ascontiguousarray(cfg["source"]))
ndt.align(g); ref = ndt.getResult()
fails = 0
t0 = time.perf_counter()
for i in range(6000):
    try:
        ndt.align(g, return_transform=False)
    except pkg.NdtError as e:
        fails += 1
        print("align", i, "FAILED after %.2f s" % (time.perf_counter() - t0), e, ndt.prelaunchCounters(), flush=True)
        if fails > 3: break
        continue
    x = ndt.getNumEvaluations()
    if i % 97 == 0:
        r = ndt.getResult()
print("done", fails, ndt.prelaunchCounters())
