"""Host-cloud (PCIe-inclusive) hand-off: setInputTarget + setInputSource + align per scan, the way the drivers call them
(ref: run/pipeline.cpp:554-561), asynchronous against blocking hand-off, with the engine's own breakdown.
    python tools/handoff_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def xyzi(a):
    out = np.zeros((len(a), 8), np.float32); out[:, :3] = a; out[:, 3] = 1.0
    return out
layouts = {"PointXYZI AoS (32 B)": (xyzi(cfg["target"]), xyzi(cfg["source"])),
           "packed xyz (12 B)": (np.ascontiguousarray(cfg["target"]), np.ascontiguousarray(cfg["source"]))}
guess = pkg.ColMajor4f(cfg["guess"])
for mode, mname in ((pkg.HANDOFF_SYNC, "sync "), (pkg.HANDOFF_ASYNC, "async")):
    ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
    ndt.setHandoffMode(mode)
    for name, (t, s) in layouts.items():
        def step():
            t0 = time.perf_counter(); ndt.setInputTarget(t)
            t1 = time.perf_counter(); ndt.setInputSource(s)
            t2 = time.perf_counter(); ndt.align(guess, return_transform=False)
            return t1 - t0, t2 - t1, time.perf_counter() - t2
        for _ in range(5): step()
        acc = np.zeros(3); rp = np.zeros(3)
        t0 = time.perf_counter()
        for _ in range(K):
            acc += step()
            h = ndt.getHandoffTiming()
            rp += (h["target"]["ms_repack"], h["source"]["ms_repack"], h["ms_build_wait"])
        el = time.perf_counter() - t0
        acc *= 1e3 / K; rp /= K
        print("%s %-22s ms/scan %.3f | setInputTarget %.3f (engine %.3f)  setInputSource %.3f (engine %.3f)  align %.3f (build wait %.3f) | it %d threads %d"
              % (mname, name, 1e3 * el / K, acc[0], rp[0], acc[1], rp[1], acc[2], rp[2], ndt.getFinalNumIteration(), h["target"]["threads"]))
    # DMA rate: one instrumented scan
    t, s = layouts["PointXYZI AoS (32 B)"]
    ndt.enableKernelTiming(True)
    ndt.setInputTarget(t); ndt.setInputSource(s); ndt.align(guess, return_transform=False); ndt.wait()
    h = ndt.getHandoffTiming(); ndt.enableKernelTiming(False)
    print("%s DMA target %.3f ms = %.1f GB/s, source %.3f ms = %.1f GB/s; cpu budget %d, workers %d; device build %.3f ms"
          % (mname, h["target"]["ms_dma"], h["target"]["dma_gb_per_s"], h["source"]["ms_dma"], h["source"]["dma_gb_per_s"],
             h["cpu_budget"], h["repack_workers"], ndt.getGridInfo()["ms_build"]))
    ndt.close()
