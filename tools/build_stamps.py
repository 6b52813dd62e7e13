"""Reads the in-kernel phase stamps of the build kernels from a -DNDT_STAMPS diagnostic build
(libndt_hip_stamps.so via NDT_HIP_LIB; not collected by pytest)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
which = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = getattr(S, "config_" + which)()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution={"c3": 0.5, "c2": 1.0}.get(which, 1.0), step_size=0.1, trans_epsilon=1e-4, max_iterations=0)
for _ in range(5): ndt.setInputTarget(cfg["target"])
L = pkg.lib()
L.ndt_debug_read_build_stamps.argtypes = [C.c_void_p]
raw = np.zeros(6 * 512 * 8, np.uint64)
assert L.ndt_debug_read_build_stamps(raw.ctypes.data) == 6 * 512
t = raw.reshape(6, 512, 8).astype(np.int64)
n = len(cfg["target"])
print("build counters (fused fallbacks, bucket fallbacks, bucketed builds):", ndt.buildCounters())
names = {0: ["entry", "keys loaded", "ranked", "table read", "staged", "stores done"], 3: ["entry", "counted", "offset known", "emitted"],
         1: ["table complete", "dense ids", "ids fetched", "ranked", "digit bases"],
         4: ["entry", "loaded+bounds", "ranked", "offsets", "staged", "stores done"],
         5: ["entry", "geometry", "points in", "sorted", "runs+slots", "sums", "statistics", "column"]}
tile = next(t for t in (2048, 4096, 8192) if (n + t - 1) // t <= 256)   # bucket_rounds_for()
for slot, nb, label in ((0, (n + 8191) // 8192, "sort pass 0 (from points)"), (1, (n + 8191) // 8192, "sort pass 1"),
                        (2, (n + 8191) // 8192, "sort pass 2"), (3, (n + 2047) // 2048, "run search (fused)"),
                        (4, (n + tile - 1) // tile, "k_bucket_pass (%d-point tiles)" % tile), (5, 256, "k_bucket_leaves")):
    nm = names.get(slot, names[0])
    b = t[slot, :min(nb, 512), :len(nm)]
    if b[:, 0].max() == 0:
        print("%s: not run" % label)
        continue
    if slot == 1 and t[5, :, 0].max() > 0:   # a bucketed build: the rows of sort pass 1 hold k_bucket_leaves' finer stamps
        b = t[1, :256, :len(nm)]
        rel = (b - t[5, :256, 0].min()) * 0.01
        print("k_bucket_leaves, inside 'points in' -> 'sorted' (us since the first block's entry):")
        for k, name in enumerate(nm):
            print("  %-14s min %6.2f  median %6.2f  max %6.2f" % (name, rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
        continue
    rel = (b - b[:, 0].min()) * 0.01
    print("%s: %d blocks (us since the first block's entry)" % (label, nb))
    for k, name in sorted(enumerate(nm), key=lambda kn: float(np.median(rel[:, kn[0]]))):
        print("  %-13s min %6.2f  median %6.2f  max %6.2f" % (name, rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
# per-wave stamps of k_bucket_leaves' blocks 0..31 (rows of slot 0): sums / statistics phases
w = t[0].reshape(32, 16, 8)
if w[:, :, 0].max() > 0:
    ent = t[5, :32, 0][:, None, None]   # the block's entry
    relw = (w - ent) * 0.01
    wn = ["sums start", "head+tree", "crowded done", "sums written", "barrier", "finalised"]
    print("k_bucket_leaves per wave, median over blocks 0..31 (us since the block's entry):")
    print("  wave          " + "".join("%15s" % x for x in wn))
    for wv in (0, 1, 2, 4, 8, 10, 12, 15):
        print("  wave %2d       " % wv + "".join("%15.2f" % np.median(relw[:, wv, k]) for k in range(len(wn))))
    print("  max over waves" + "".join("%15.2f" % np.median(relw[:, :, k].max(axis=1)) for k in range(len(wn))))
