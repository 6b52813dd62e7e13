"""H2D copy cost on the box (tuning aid): one 12 MB copy against many small ones, pinned staging."""
import ctypes as C, time
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
vp = C.c_void_p
def chk(r): assert r == 0, r
h = vp(); d = vp(); s = vp()
N = 12_000_000
chk(hip.hipHostMalloc(C.byref(h), C.c_size_t(N), 0)); chk(hip.hipMalloc(C.byref(d), C.c_size_t(N))); chk(hip.hipStreamCreate(C.byref(s)))
C.memset(h, 1, N)
def run(nchunks, reps=30):
    sz = N // nchunks
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for c in range(nchunks):
            chk(hip.hipMemcpyAsync(vp(d.value + c * sz), vp(h.value + c * sz), C.c_size_t(sz), 1, s))
        chk(hip.hipStreamSynchronize(s))
        best = min(best, time.perf_counter() - t0)
    return best
for n in (1, 3, 8, 24, 72):
    t = run(n)
    print("%2d copies of %7.0f KB: %.3f ms  (%.1f GB/s)" % (n, N / n / 1024, 1e3 * t, N / t / 1e9), flush=True)
