"""Stress of the asynchronous host hand-off under host CPU contention (not collected by pytest): background threads keep
the BLAS pool busy while the main thread registers scans of varying size through setInputTarget / setInputSource / align,
overwriting its clouds the moment each call returns; every result must equal the blocking hand-off's.
Usage: python tools/handoff_stress.py [scans]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package(); S = pkg.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
a, b = S.config_c2(), S.config_c3()
kw = dict(resolution=1.0, step_size=0.1, trans_epsilon=1e-4, max_iterations=35)
cases = []
rng = np.random.default_rng(4)
for k in range(12):
    cfg = a if k % 3 else b
    nt = int(rng.integers(20000, len(cfg["target"])))
    ns = int(rng.integers(5000, len(cfg["source"])))
    cases.append((cfg["target"][:nt], cfg["source"][:ns], cfg["guess"], float(cfg["resolution"])))
ref = pkg.NormalDistributionsTransform(device_id=0, **kw); ref.setHandoffMode(pkg.HANDOFF_SYNC)
want = []
for t, s, g, res in cases:
    ref.setResolution(res); ref.setInputTarget(t); ref.setInputSource(s); T = ref.align(g); r = ref.getResult()
    want.append((T.copy(), r["iterations"], r["score"]))
ref.close()
ndt = pkg.NormalDistributionsTransform(device_id=0, **kw)
stop = False
def hog():
    m = np.random.default_rng(0).normal(size=(500, 500))
    while not stop:
        m = (m @ m) / 500.0
hogs = [threading.Thread(target=hog, daemon=True) for _ in range(3)]
for h in hogs: h.start()
bad = fails = 0
def rss_mb():
    return int(open('/proc/self/statm').read().split()[1]) * os.sysconf('SC_PAGE_SIZE') / 2**20
rss0 = None
t0 = time.perf_counter()
def xyzi(c):
    o = np.zeros((len(c), 8), np.float32); o[:, :3] = c; return o
for i in range(n):
    k = int(rng.integers(0, len(cases)))
    t, s, g, res = cases[k]
    try:
        ndt.setResolution(res)
        tt = xyzi(t) if i % 2 else t.copy(); ndt.setInputTarget(tt); tt[:] = np.nan
        ss = xyzi(s) if i % 3 else s.copy(); ndt.setInputSource(ss); ss[:] = np.nan
        T = ndt.align(g); r = ndt.getResult()
    except pkg.NdtError as e:
        fails += 1; print("FAIL at scan %d: %s" % (i, e), flush=True)
        if fails > 5: break
        continue
    if i == 200: rss0 = rss_mb()   # (allocations settle during the first scans)
    if not (np.array_equal(T, want[k][0]) and r["iterations"] == want[k][1] and r["score"] == want[k][2]):
        bad += 1; print("MISMATCH at scan %d case %d" % (i, k), flush=True)
stop = True
print("%d scans through the asynchronous hand-off under host contention in %.1f s: %d failures, %d mismatches; prelaunch counters %s; "
      "first evaluations behind a running build (kept, discarded) %s; resident set %.0f MB after 200 scans, %.0f MB at the end"
      % (n, time.perf_counter() - t0, fails, bad, ndt.prelaunchCounters(), ndt.speculationCounters(), rss0 or 0.0, rss_mb()), flush=True)
