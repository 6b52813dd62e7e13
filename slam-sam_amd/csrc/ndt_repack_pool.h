// ndt_repack_pool.h -- the HOST half of the cloud hand-off (internal; plain C++ with SSE2 so that tests/cpp can
// drive it under the sanitizers without a GPU): persistent worker threads, the AoS / SoA -> chunk-major repack into
// pinned staging, and the loop that hands every finished chunk to the device while later ones are repacked.
#pragma once

#include <emmintrin.h>
#include <sched.h>
#include <xmmintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace ndt {

// CPUs this process may actually use: the affinity mask cut down to the cgroup CPU quota (a 1-GPU box shows all 256
// host CPUs to a job that is allowed 16 of them).
inline int host_cpu_budget() {
  static const int budget = [] {
    int cpus = (int)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) cpus = CPU_COUNT(&set);
    double quota = 0.0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
      char q[64];
      long long per = 0;
      if (std::fscanf(f, "%63s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = std::atof(q) / (double)per;
      std::fclose(f);
    } else {
      long long q = -1, per = 0;  // cgroup v1
      if (FILE* a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(a, "%lld", &q) != 1) q = -1; std::fclose(a); }
      if (FILE* b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(b, "%lld", &per) != 1) per = 0; std::fclose(b); }
      if (q > 0 && per > 0) quota = (double)q / (double)per;
    }
    if (quota > 0.0) cpus = std::min(cpus, std::max(1, (int)(quota + 0.5)));
    return std::max(1, cpus);
  }();
  return budget;
}

// Repack workers of one hand-off: half of the CPU budget (the drivers run 6-7 threads of their own), at most 12;
// NDT_UPLOAD_THREADS overrides.  The calling thread repacks too, so 0 workers is a valid answer on a 1-2 CPU budget.
inline unsigned repack_workers() {
  static const unsigned n = [] {
    if (const char* e = std::getenv("NDT_UPLOAD_THREADS")) {
      const int v = std::atoi(e);
      if (v >= 0) return (unsigned)std::min(v, 64);
    }
    const int b = host_cpu_budget();
    return (unsigned)std::min(12, std::max(0, b / 2));
  }();
  return n;
}

// A few persistent host threads.  A worker that has finished a job keeps polling for the next one for a short while
// (setInputSource follows setInputTarget within microseconds; waking a sleeping thread costs 30-60 us, a third of a
// 200 k-point scan's whole repack) and then sleeps on a condition variable.
struct RepackPool {
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv, done_cv;
  std::function<void()> job;         // written under `m` before `gen` is raised; workers call it in place
  std::atomic<unsigned long> gen{0};
  std::atomic<int> pending{0};       // workers that have not finished the current generation's job yet
  std::atomic<bool> stop{false};
  int spin_us = 100;   // (a driver at 10-20 Hz: ~1 ms of spinning per second and worker; a 1 kHz replay keeps them awake)
  // A worker only ever runs the job of a generation that was published AFTER it started: it is born
  // with `seen` = the generation current at that moment (read under the mutex).  A worker born with
  // seen = 0 into a pool whose gen was already > 0 used to wake at once and run the PREVIOUS upload's
  // job -- a lambda over a dead stack frame (tests/cpp/sanitize_host.cpp).
  void ensure(unsigned n) {
    std::lock_guard<std::mutex> lk(m);
    while (th.size() < n) {
      const unsigned long born = gen.load(std::memory_order_relaxed);
      try {
      th.emplace_back([this, born] {
        unsigned long seen = born;
        for (;;) {
          const auto t0 = std::chrono::steady_clock::now();
          unsigned spins = 0;
          while (gen.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
            _mm_pause();
            if ((++spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) {
              std::unique_lock<std::mutex> lk(m);
              cv.wait(lk, [&] { return stop.load(std::memory_order_relaxed) || gen.load(std::memory_order_acquire) != seen; });
              break;
            }
          }
          if (stop.load(std::memory_order_relaxed)) return;
          seen = gen.load(std::memory_order_acquire);
          if (job) job();
          if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(m);
            done_cv.notify_all();
          }
        }
      });
      } catch (const std::system_error&) {
        break;   // no more threads to be had (a process / thread limit): the hand-off goes on with the workers it has
      }          // -- none at all is fine too, the calling thread repacks every piece itself -- and nothing crosses the C-ABI
    }
  }
  // every worker runs f once (f claims pieces from a shared counter); returns at once.  Must be
  // followed by wait() before the next run() and before f's captures go out of scope.
  void run(std::function<void()> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f);
    pending.store((int)th.size(), std::memory_order_relaxed);  // every existing worker sees this generation exactly once
    gen.fetch_add(1, std::memory_order_release);
    cv.notify_all();
  }
  void wait() {
    for (unsigned spins = 0; pending.load(std::memory_order_acquire) != 0 && spins < 20000; ++spins) _mm_pause();
    std::unique_lock<std::mutex> lk(m);
    done_cv.wait(lk, [&] { return pending.load(std::memory_order_acquire) == 0; });
    job = nullptr;  // no callable outlives the frame it captured
  }
  ~RepackPool() {
    {
      std::lock_guard<std::mutex> lk(m);
      stop.store(true);
      cv.notify_all();
    }
    for (auto& t : th) t.join();
  }
};

// ---- AoS / SoA -> [x | y | z] --------------------------------------------------------------------------------------
// Points [lo, hi) of a strided host cloud (`base` + i * stride holds x, y, z as three floats) to three dense arrays;
// dx / dy / dz point at the destination of point `lo`.  16-byte loads where a point's 16 bytes are known to lie inside
// the caller's buffer: every point but the LAST of the cloud (`n_total`) when stride >= 16, whole groups of four points
// when stride == 12.  Everything else goes scalar.  Reads exactly the bytes [base, base + (n_total - 1) * stride + 12).
inline void repack_strided(const char* base, size_t stride, size_t n_total, size_t lo, size_t hi, float* dx, float* dy, float* dz) {
  size_t i = lo;
  if (stride >= 16) {
    const size_t vhi = std::min(hi, n_total ? n_total - 1 : 0);
    for (; i + 4 <= vhi; i += 4) {
      const char* p = base + i * stride;
      __m128 r0 = _mm_loadu_ps(reinterpret_cast<const float*>(p));
      __m128 r1 = _mm_loadu_ps(reinterpret_cast<const float*>(p + stride));
      __m128 r2 = _mm_loadu_ps(reinterpret_cast<const float*>(p + 2 * stride));
      __m128 r3 = _mm_loadu_ps(reinterpret_cast<const float*>(p + 3 * stride));
      _MM_TRANSPOSE4_PS(r0, r1, r2, r3);
      _mm_storeu_ps(dx + (i - lo), r0);
      _mm_storeu_ps(dy + (i - lo), r1);
      _mm_storeu_ps(dz + (i - lo), r2);
    }
  } else if (stride == 12) {
    for (; i + 4 <= hi; i += 4) {
      const float* p = reinterpret_cast<const float*>(base + i * 12);
      const __m128 a = _mm_loadu_ps(p);      // x0 y0 z0 x1
      const __m128 b = _mm_loadu_ps(p + 4);  // y1 z1 x2 y2
      const __m128 c = _mm_loadu_ps(p + 8);  // z2 x3 y3 z3
      const __m128 tx = _mm_shuffle_ps(b, c, _MM_SHUFFLE(1, 1, 2, 2));   // x2 x2 x3 x3
      const __m128 ty0 = _mm_shuffle_ps(a, b, _MM_SHUFFLE(0, 0, 1, 1));  // y0 y0 y1 y1
      const __m128 ty1 = _mm_shuffle_ps(b, c, _MM_SHUFFLE(2, 2, 3, 3));  // y2 y2 y3 y3
      const __m128 tz = _mm_shuffle_ps(a, b, _MM_SHUFFLE(1, 1, 2, 2));   // z0 z0 z1 z1
      _mm_storeu_ps(dx + (i - lo), _mm_shuffle_ps(a, tx, _MM_SHUFFLE(2, 0, 3, 0)));
      _mm_storeu_ps(dy + (i - lo), _mm_shuffle_ps(ty0, ty1, _MM_SHUFFLE(2, 0, 2, 0)));
      _mm_storeu_ps(dz + (i - lo), _mm_shuffle_ps(tz, c, _MM_SHUFFLE(3, 0, 2, 0)));
    }
  }
  for (; i < hi; ++i) {
    const float* p = reinterpret_cast<const float*>(base + i * stride);
    dx[i - lo] = p[0];
    dy[i - lo] = p[1];
    dz[i - lo] = p[2];
  }
}

// One cloud on its way into pinned staging.  Staging holds the cloud CHUNK BY CHUNK as [x | y | z] of the chunk's
// points -- every segment `seg` floats long, the chunk's point count rounded up to 4, so that all three start 16-byte
// aligned -- and a chunk crosses PCIe as one unit; a chunk is repacked as several PIECES claimed from a shared counter
// by the pool's workers and by the calling thread.  Chunk c starts at float 3 * c * chunk; capacity: stage_floats(n).
struct StageJob {
  const char* aos = nullptr;   // strided cloud (stride bytes apart), or
  size_t stride = 0;
  const float* x = nullptr;    // three dense arrays
  const float* y = nullptr;
  const float* z = nullptr;
  size_t n = 0;
  float* stage = nullptr;      // stage_floats(n) floats
  size_t chunk = 131072;       // points per transfer; a multiple of 4
  size_t piece = 8192;         // points per claim; divides chunk
  size_t nchunks = 0, npieces = 0;
  std::atomic<size_t> next{0};
  std::unique_ptr<std::atomic<int>[]> done;   // pieces finished, per chunk

  void prepare() {
    nchunks = (n + chunk - 1) / chunk;
    npieces = (n + piece - 1) / piece;
    done.reset(new std::atomic<int>[nchunks ? nchunks : 1]);
    for (size_t c = 0; c < nchunks; ++c) done[c].store(0, std::memory_order_relaxed);
    next.store(0, std::memory_order_relaxed);
  }
  static size_t stage_floats(size_t n) { return 3 * ((n + 3) & ~(size_t)3); }
  size_t seg(size_t c) const { return (chunk_hi(c) - chunk_lo(c) + 3) & ~(size_t)3; }
  size_t chunk_lo(size_t c) const { return c * chunk; }
  size_t chunk_hi(size_t c) const { return std::min(n, (c + 1) * chunk); }
  int pieces_of(size_t c) const { return (int)((chunk_hi(c) - chunk_lo(c) + piece - 1) / piece); }
  // claims and repacks one piece; false when none is left
  bool work_one() {
    const size_t p = next.fetch_add(1, std::memory_order_relaxed);
    if (p >= npieces) return false;
    const size_t lo = p * piece, hi = std::min(n, lo + piece);
    const size_t c = lo / chunk, c0 = chunk_lo(c), len = seg(c);
    float* b = stage + 3 * c0;
    if (aos) {
      repack_strided(aos, stride, n, lo, hi, b + (lo - c0), b + len + (lo - c0), b + 2 * len + (lo - c0));
    } else {
      std::memcpy(b + (lo - c0), x + lo, (hi - lo) * sizeof(float));
      std::memcpy(b + len + (lo - c0), y + lo, (hi - lo) * sizeof(float));
      std::memcpy(b + 2 * len + (lo - c0), z + lo, (hi - lo) * sizeof(float));
    }
    done[c].fetch_add(1, std::memory_order_release);
    return true;
  }
};

// Repacks the whole cloud of `job` and calls on_chunk(c, lo, hi) for every chunk, in order, as soon as it is complete
// (the caller starts the chunk's transfer there).  The calling thread repacks pieces whenever the next chunk is not ready.
// Returns when every piece is done and no worker touches `job` or the caller's cloud any more: the caller's memory is
// consumed.  `workers` = 0: everything on the calling thread.
template <class OnChunk>
inline void stage_cloud(RepackPool* pool, unsigned workers, StageJob& job, OnChunk&& on_chunk) {
  job.prepare();
  const bool use_pool = pool && workers > 0 && job.npieces > 1;
  if (use_pool) {
    pool->ensure(workers);
    StageJob* j = &job;
    pool->run([j] { while (j->work_one()) {} });  // (workers beyond `workers`, left from a larger hand-off, just help)
  }
  for (size_t c = 0; c < job.nchunks; ++c) {
    const int want = job.pieces_of(c);
    while (job.done[c].load(std::memory_order_acquire) < want)
      if (!job.work_one()) _mm_pause();
    on_chunk(c, job.chunk_lo(c), job.chunk_hi(c));
  }
  if (use_pool) pool->wait();   // the job and its captures live on the caller's frame
}

}  // namespace ndt
