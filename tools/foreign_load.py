"""Rebuild + align while ANOTHER process keeps the same GPU busy with long kernels (debugging aid, not
collected by pytest): the engine's in-kernel waits (fused sort passes, pre-launched evaluations) must
fall back, never hang or change a number."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HOG = r'''
import torch, time, sys
d = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=d); b = torch.randn(8192, 8192, device=d)
t0 = time.time(); n = 0
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(20): a = (a @ b) * 1e-2
    torch.cuda.synchronize(); n += 20
print("hog: %d matmuls" % n, flush=True)
'''
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"]); ndt.setInputSource(cfg["source"])
g = pkg.ColMajor4f(cfg["guess"])
ndt.align(g); ref = ndt.getResult(); refL = ndt.getLeaves()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
hog = subprocess.Popen([sys.executable, "-c", HOG, str(secs)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
time.sleep(4.0)   # let the other process get going
bad = fails = n = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs - 6.0:
    try:
        ndt.setInputTarget(cfg["target"])
        ndt.align(g, return_transform=False)
    except pkg.NdtError as e:
        fails += 1; print("FAIL", n, e, flush=True)
        if fails > 5: break
        continue
    r = ndt.getResult()
    if r["score"] != ref["score"] or not np.array_equal(r["T"], ref["T"]):
        bad += 1; print("MISMATCH", n, r["score"], ref["score"], flush=True)
    if n % 20 == 0:
        L = ndt.getLeaves()
        if not np.array_equal(L["cell"], refL["cell"]) or not np.array_equal(L["mean"], refL["mean"]):
            bad += 1; print("LEAF MISMATCH", n, flush=True)
    n += 1
el = time.perf_counter() - t0
out = hog.communicate(timeout=120)[0]
print(out.strip().splitlines()[-1] if out.strip() else "hog: no output")
print("%d rebuild+align steps beside a foreign GPU load in %.1f s (%.2f ms each): %d failures, %d mismatches; fused-sort fallbacks %d, pre-launch (used, quit, timeouts) %s"
      % (n, el, 1e3 * el / max(n, 1), fails, bad, ndt.buildCounters()[0], ndt.prelaunchCounters()), flush=True)
