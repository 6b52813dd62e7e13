"""Reads the in-kernel phase stamps of a -DNDT_STAMPS diagnostic build (not collected by pytest)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=0)
ndt.setInputTarget(cfg["target"])
L = pkg.lib()
L.ndt_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
for n in (1000, 200000):
    ndt.setInputSource(cfg["source"][:n])
    bt = int(os.environ.get("NDT_DERIV_BLOCK", "0")) or ((((n + 255) // 256 + 63) // 64) * 64 if 131072 < n <= 262144 else 512)
    nb = (n + bt - 1) // bt
    for _ in range(5): ndt.align(cfg["gt"])
    raw = np.zeros(nb * 9, np.uint64)
    got = L.ndt_debug_read_stamps(raw.ctypes.data, nb)
    assert got == nb, got
    buf = raw[:nb * 8].reshape(nb, 8)
    hw = raw[nb * 8:].view(np.uint32).reshape(nb, 2)
    t = buf.astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01  # us
    last = np.argmax(t[:, 7])
    names = ["entry", "xyz loaded", "pairs done", "expanded", "row stored", "ticket back", "final sum", "flag out"]
    print("n=%d blocks=%d  (us since first block entry)" % (n, nb))
    for k in range(6):
        print("  %-12s  min %6.2f  median %6.2f  max %6.2f" % (names[k], rel[:, k].min(), np.median(rel[:, k]), rel[:, k].max()))
    print("  last block %d: ticket %.2f  final sum %.2f  flag %.2f" % (last, rel[last, 5], rel[last, 6], rel[last, 7]))
    if n > 1000:
        xcc = hw[:, 1] & 0xF
        cu = (hw[:, 0] >> 8) & 0xF; sh = (hw[:, 0] >> 12) & 0x1; se = (hw[:, 0] >> 13) & 0x7
        key = xcc.astype(np.int64) * 1000 + se * 100 + sh * 20 + cu
        uniq, cnt = np.unique(key, return_counts=True)
        print("  distinct CUs used %d; blocks per CU histogram %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
        per = {k: c for k, c in zip(uniq, cnt)}
        two = np.array([per[k] for k in key]) >= 2
        if two.any() and (~two).any():
            print("  pairs-done (us): CUs with 1 block: median %.2f max %.2f | CUs with >=2 blocks: median %.2f max %.2f"
                  % (np.median(rel[~two, 2]), rel[~two, 2].max(), np.median(rel[two, 2]), rel[two, 2].max()))
        print("  per-XCC block counts", dict(zip(*np.unique(xcc, return_counts=True))))
