"""Per-WAVE in-kernel stamps of k_derivatives on C3 (-DNDT_STAMPS build via NDT_HIP_LIB; not collected by pytest).

For the ordinary launch (ndt_eval_derivatives at the converged pose) and for the last pre-launched evaluation of an
align: when every wave of every block finished its pairs, its expansion, its reduce-scatter, passed the block barrier,
and when the block's row went out; which SIMD each wave ran on; what the summing block's poll trips saw.
Answers VERDICT r04 item 1(a): is the time between wave 0's `expanded` and the block's `row stored` the other waves'
pair phase (VALU issue on the SIMD that holds four waves) or the reduction / hand-off?  (profiles/r05_stamps_per_wave.txt is
the run BEFORE the finishing-wave change; since then stamps 4 and 5 of a wave that is not one of its block's last four are
just the moments it reaches the block barrier.)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); pkg.apply_env_tuning(); S = pkg.synth
WS_WAVES, WS_N, WS_TRIPS = 16, 10, 64
NAMES = ["entry", "xyz loaded", "pairs done", "pair sums published", "expanded (finishing waves: all their items)", "wave reduced (finishing waves)",
         "block barrier", "row issued / leaving", "row acknowledged"]
cfg = S.config_c3()
ndt = pkg.NormalDistributionsTransform(device_id=0, resolution=0.5, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
ndt.setInputTarget(cfg["target"])
ndt.setInputSource(cfg["source"])
L = pkg.lib()
L.ndt_debug_read_wave_stamps.argtypes = [C.c_void_p, C.c_int]
L.ndt_debug_read_wave_stamps.restype = C.c_int
n = len(cfg["source"])
bt = int(os.environ.get("NDT_DERIV_BLOCK", "0")) or (((n + 254) // 255 + 63) // 64) * 64
split = int(os.environ.get("NDT_DERIV_SUMMER_SPLIT", "1"))
ded = 0 if os.environ.get("NDT_DERIV_DEDICATED") == "0" else (1 if split == 0 else (4 if split == 1 else split))   # summing blocks in front of the point blocks
nb = (n + bt - 1) // bt + ded
nwv = bt // 64


def read():
    nw = nb * WS_WAVES
    raw = np.zeros(nw * WS_N + nw // 2 + 2 * WS_TRIPS + 1, np.uint64)
    got = L.ndt_debug_read_wave_stamps(raw.ctypes.data, nb)
    assert got == nb, got
    t = raw[:nw * WS_N].reshape(nb, WS_WAVES, WS_N).astype(np.int64)
    hw = raw[nw * WS_N:nw * WS_N + nw // 2].view(np.uint32).reshape(nb, WS_WAVES)
    trips = raw[nw * WS_N + nw // 2:].astype(np.int64)
    return t, hw, trips


def pct(a):
    return "min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))


def report(title, t, hw, trips, t0):
    print("== %s: %d blocks x %d waves of %d threads (+%d summing block), us after %s" % (title, nb - ded, nwv, bt, ded, t0[1]))
    comp = t[ded:, :nwv, :]          # [block][wave][stamp]
    valid = comp[:, :, 0] > 0
    rel = (comp - t0[0]) * 0.01
    simd = (hw[ded:, :nwv] >> 4) & 3
    for k in range(8):
        if k == 3 and not (comp[:, :, 3] > 0).any():
            continue
        a = rel[:, :, k][valid & (comp[:, :, k] > 0)]
        if a.size:
            print("  all waves   %-22s %s" % (NAMES[k], pct(a)))
    a = rel[:, 0, 8][comp[:, 0, 8] > 0]
    if a.size:
        print("  wave 0      %-22s %s" % (NAMES[8], pct(a)))
    ent = np.where(valid, rel[:, :, 0], np.nan)
    print("  a block's waves enter within %.2f us of each other (median; max %.2f)" % (np.nanmedian(np.nanmax(ent, 1) - np.nanmin(ent, 1)), np.nanmax(np.nanmax(ent, 1) - np.nanmin(ent, 1))))
    # per block: the first and the last wave through each phase
    print("  per block, first / last wave through a phase (median over blocks):")
    for k in (2, 4, 5):
        x = np.where(valid, rel[:, :, k], np.nan)
        print("    %-22s first %6.2f  last %6.2f   (last - first: median %5.2f max %5.2f)"
              % (NAMES[k], np.nanmedian(np.nanmin(x, 1)), np.nanmedian(np.nanmax(x, 1)), np.nanmedian(np.nanmax(x, 1) - np.nanmin(x, 1)),
                 np.nanmax(np.nanmax(x, 1) - np.nanmin(x, 1))))
    last_red = np.nanmax(np.where(valid, rel[:, :, 5], np.nan), 1)
    bar = np.nanmax(np.where(valid, rel[:, :, 6], np.nan), 1)
    issued = rel[:, 0, 7]
    acked = rel[:, 0, 8]
    print("    slowest wave reduced -> barrier released (last wave through) %5.2f ; -> row issued (wave 0) %5.2f ; -> acknowledged %5.2f   (medians)"
          % (np.nanmedian(bar - last_red), np.nanmedian(issued - last_red), np.nanmedian(acked - last_red)))
    # by SIMD occupancy: waves per SIMD of the block's compute unit, and when the waves of the fullest SIMD finish
    cnt = np.stack([(simd == s_).sum(1) for s_ in range(4)], 1)       # [block][simd]
    print("  waves per SIMD of a block (sorted): %s" % {tuple(int(v) for v in r): int(c) for r, c in zip(*np.unique(np.sort(cnt, 1)[:, ::-1], axis=0, return_counts=True))})
    full = cnt.max(1)
    for k in (2, 4, 5):
        on_full, on_rest = [], []
        for b in range(comp.shape[0]):
            fs = int(np.argmax(cnt[b]))
            m = valid[b]
            on_full += list(rel[b, (simd[b] == fs) & m, k])
            on_rest += list(rel[b, (simd[b] != fs) & m, k])
        print("    %-22s waves on the fullest SIMD: median %6.2f max %6.2f | on the others: median %6.2f max %6.2f"
              % (NAMES[k], np.median(on_full), np.max(on_full), np.median(on_rest), np.max(on_rest)))
    # rank of a wave on its SIMD (by the hardware's wave slot order = age): how the pair phase stretches with every older wave
    for k in (2, 5):
        rows = []
        for rank in range(4):
            v = []
            for b in range(comp.shape[0]):
                for s_ in range(4):
                    w = np.where((simd[b] == s_) & valid[b])[0]
                    if len(w) == int(full[b]) and len(w) > rank:
                        order = w[np.argsort(rel[b, w, k])]
                        v.append(rel[b, order[rank], k])
            rows.append(np.median(v) if v else float("nan"))
        print("    %-22s on the fullest SIMD, 1st..4th wave to get there: %s" % (NAMES[k], "  ".join("%6.2f" % r for r in rows)))
    # the finishing waves (derivs_item_owners): the last wave of a SIMD with one wave more than the others, the first elsewhere
    pmin = nwv // 4
    fw = [q + 4 * (((nwv - q + 3) // 4) - 1) if (nwv >= 4 and (nwv - q + 3) // 4 > pmin) else q for q in range(min(4, nwv))]
    for k in (2, 4, 5):
        print("    finishing waves %-46s %s" % (NAMES[k], "  ".join("w%d %.2f" % (w, np.median(rel[:, w, k])) for w in fw)))
    for sb in range(ded):
        live = t[sb, :nwv, 6] > 0       # (a block that adds a quarter of the words uses a quarter of its waves)
        sm = (t[sb, :nwv, :] - t0[0]) * 0.01
        print("  summing block %d: polling from %.2f | a wave has all its slots: first %.2f last %.2f | barrier passed %.2f | result issued %.2f | acknowledged (wave 0) %.2f"
              % (sb, np.min(sm[live, 5]), np.min(sm[live, 6]), np.max(sm[live, 6]), np.max(sm[:, 7]), sm[0, 8], sm[0, 9]))
    if ded:
        sm = (t[0, :nwv, :] - t0[0]) * 0.01
        live = t[0, :nwv, 6] > 0
        ntr = int(trips[2 * WS_TRIPS])
        tr = [((trips[2 * i] - t0[0]) * 0.01, int(trips[2 * i + 1])) for i in range(min(ntr, WS_TRIPS))]
        print("  summing block, wave 0: %d poll trips; (us, lanes still missing a slot): %s" % (ntr, " ".join("(%.2f,%d)" % x for x in tr[-12:])))
        print("  last row acknowledged %.2f -> summing wave complete (block 0) %.2f -> result issued %.2f" % (np.nanmax(acked), np.max(sm[live, 6]), sm[0, 8]))


# --- ordinary launches at the converged pose ---------------------------------------------------------------------
# (with kernel timing on, every evaluation of an align is an ordinary launch: the kernel `roofline` prices)
ndt.align(cfg["guess"])
ndt.enableKernelTiming(True)
for rep in range(3):
    ndt.align(cfg["guess"])
    t, hw, trips = read()
    report("ordinary launch, repeat %d" % rep, t, hw, trips, (t[:, :nwv, 0][t[:, :nwv, 0] > 0].min(), "the first wave's entry"))

ndt.enableKernelTiming(False)
# --- the last pre-launched evaluation of an align ----------------------------------------------------------------
L.ndt_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    ndt.align(cfg["guess"])
    t, hw, trips = read()
    raw = np.zeros(nb * 11, np.uint64)
    assert L.ndt_debug_read_stamps(raw.ctypes.data, nb) == nb
    ms = raw[nb * 9:].reshape(nb, 2).astype(np.int64)
    report("pre-launched evaluation (last of an align), repeat %d, counters %s" % (rep, ndt.prelaunchCounters()), t, hw, trips,
           (ms[:, 1].min(), "the first block saw the pose"))
    if os.environ.get("NDT_STAMPS_DUMP"):   # raw arrays for offline analysis: [block][wave][stamp], hw ids, per-block {entry, pose seen}
        np.savez(os.environ["NDT_STAMPS_DUMP"] + "_%d.npz" % rep, t=t, hw=hw, ms=ms, trips=trips, xcc=raw[nb * 8:nb * 9].view(np.uint32).reshape(nb, 2))
