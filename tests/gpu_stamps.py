"""Reads the in-kernel phase stamps of a -DNDT_STAMPS diagnostic build (not collected by pytest)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=0)
ndt.setInputTarget(cfg["target"])
L = pkg.lib()
L.ndt_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
for n in (1000, 200000):
    ndt.setInputSource(cfg["source"][:n])
    nb = (n + 511) // 512
    for _ in range(5): ndt.align(cfg["gt"])
    buf = np.zeros((nb, 8), np.uint64)
    got = L.ndt_debug_read_stamps(buf.ctypes.data, nb)
    assert got == nb, got
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01  # us
    last = np.argmax(t[:, 7])
    names = ["entry", "xyz loaded", "pairs done", "expanded", "row stored", "ticket back", "final sum", "flag out"]
    print("n=%d blocks=%d  (us since first block entry)" % (n, nb))
    for k in range(6):
        print("  %-12s  min %6.2f  median %6.2f  max %6.2f" % (names[k], rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
    print("  last block %d: ticket %.2f  final sum %.2f  flag %.2f" % (last, rel[last, 5], rel[last, 6], rel[last, 7]))
