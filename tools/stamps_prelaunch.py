"""In-kernel stamps of the LAST pre-launched evaluation of a C3 align (-DNDT_STAMPS build via NDT_HIP_LIB; not
collected by pytest): how long a block waited for its pose, how far apart the blocks saw it, and the chain
from there to the result."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
cfg = S.config_c3()
if os.environ.get("NDT_STAMPS_NSRC"):   # a rank's share of the scan
    cfg["source"] = cfg["source"][:int(os.environ["NDT_STAMPS_NSRC"])]
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
ndt.setInputSource(cfg["source"])
L = pkg.lib()
L.ndt_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
n = len(cfg["source"])
bt = int(os.environ.get("NDT_DERIV_BLOCK", "0")) or ((((n + 255) // 256 + 63) // 64) * 64 if 131072 < n <= 262144 else
                                                    (min(512, max(256, (((n + 195) // 196 + 63) // 64) * 64)) if n <= 196 * 512 else 512))
ded = 0 if os.environ.get("NDT_DERIV_DEDICATED") == "0" else 1   # block 0 owns no points: it only adds the rows
if not int(os.environ.get("NDT_DERIV_BLOCK", "0")) and 131072 < n <= 262144:
    bt = (((n + 254) // 255 + 63) // 64) * 64                      # one compute unit is left to the summing block
nb = (n + bt - 1) // bt + ded
for rep in range(4):
    ndt.align(cfg["guess"])
    raw = np.zeros(nb * 11, np.uint64)
    assert L.ndt_debug_read_stamps(raw.ctypes.data, nb) == nb
    t = raw[:nb * 8].reshape(nb, 8).astype(np.int64)
    ms = raw[nb * 9:].reshape(nb, 2).astype(np.int64)
    seen0 = ms[:, 1].min()
    rel = (t - seen0) * 0.01
    comp = slice(ded, nb)   # the blocks that own points
    print("align %d: blocks %d, counters %s overlapped %d" % (rep, nb, ndt.prelaunchCounters(), ndt.prelaunchOverlapped()))
    print("  waited for the pose (entry -> seen): min %.2f median %.2f max %.2f us" % tuple(np.percentile((ms[:, 1] - ms[:, 0]) * 0.01, [0, 50, 100])))
    print("  pose seen, spread over blocks: median %.2f max %.2f us after the first" % tuple(np.percentile((ms[:, 1] - seen0) * 0.01, [50, 100])))
    names = ["(entry)", "xyz loaded", "pairs done", "expanded", "row stored"]
    for k in range(1, 5):
        print("  %-12s  min %6.2f  median %6.2f  max %6.2f   (us after the first block saw the pose)" % (names[k], rel[comp, k].min(), np.median(rel[comp, k]), rel[comp, k].max()))
    rs = rel[comp, 4]
    print("  row stored percentiles: p50 %.2f  p75 %.2f  p90 %.2f  p95 %.2f  p99 %.2f  max %.2f" % tuple(np.percentile(rs, [50, 75, 90, 95, 99, 100])))
    hw = raw[nb * 8:nb * 9].view(np.uint32).reshape(nb, 2)
    order = np.argsort(-rs)[:12]
    # HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (se 2 bits on some parts)
    print("  slowest blocks (block: row stored, pairs done | XCC, SE, SH, CU): " +
          "  ".join("%d: %.1f %.1f | %d %d %d %d" % (b + comp.start, rs[b], rel[comp, 2][b], hw[b + comp.start, 1] & 0xf, (hw[b + comp.start, 0] >> 13) & 7,
                                                     (hw[b + comp.start, 0] >> 12) & 1, (hw[b + comp.start, 0] >> 8) & 15) for b in order))
    print("  block 0 (%s): polling since %.2f, rows summed %.2f, result stored %.2f" % ("summing only" if ded else "computes too", rel[0, 5], rel[0, 6], rel[0, 7]))
