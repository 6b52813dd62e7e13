// ndt_comm.cpp -- RCCL / shared-memory / hook reduction of evaluation partials.
#include "ndt_comm.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
// How long a rank waits for the others inside one all-reduce before it reports NDT_ERR_COMM
// (NDT_COMM_TIMEOUT_S, default 120 s: a peer that is paged out or being debugged is not an error).
std::chrono::seconds wait_limit() {
  static const long s = [] {
    const char* e = std::getenv("NDT_COMM_TIMEOUT_S");
    const long v = e ? std::atol(e) : 0;
    return v > 0 ? v : 120L;
  }();
  return std::chrono::seconds(s);
}
}  // namespace

namespace ndt {

namespace {

constexpr int kMaxRanks = 64;
constexpr uint64_t kMagic = 0x4e44545f53484d31ull;  // "NDT_SHM1"

// One cache line per rank for the sequence words so ranks do not false-share.
struct alignas(64) SeqWord {
  std::atomic<uint64_t> v;
  char pad[56];
};

struct ShmSeg {
  std::atomic<uint64_t> magic;
  int nranks;
  char pad[52];
  SeqWord seq[kMaxRanks];
  // attach handshake: rank r writes a value unique to this attempt, only a LIVE rank 0 echoes it.
  // A segment left behind by a crashed run (same name, magic set, large sequence words) never
  // answers, so nobody can start summing its garbage.
  SeqWord attach[kMaxRanks];
  SeqWord ack[kMaxRanks];
  double slot[2][kMaxRanks][NDT_EVAL_WORDS];
};

}  // namespace

int Reducer::unique_id(void* out128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return NDT_ERR_COMM;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(out128, &id, 128);
  return NDT_OK;
}

int Reducer::library_info(char* path_buf, size_t cap) {
  int v = 0;
  if (ncclGetVersion(&v) != ncclSuccess) return NDT_ERR_COMM;
  if (path_buf && cap) {
    Dl_info info;
    const char* name = "?";
    if (dladdr(reinterpret_cast<void*>(&ncclGetVersion), &info) && info.dli_fname) name = info.dli_fname;
    snprintf(path_buf, cap, "%s", name);
  }
  return v;
}

int Reducer::rank_count() const {
  if (mode_ == NDT_REDUCE_NONE) return 1;
  if (mode_ == NDT_REDUCE_RCCL) {
    int n = 0;
    if (ncclCommCount(static_cast<ncclComm_t>(nccl_comm_), &n) != ncclSuccess) return NDT_ERR_COMM;
    return n;
  }
  return nranks_;
}

int Reducer::init_rccl(const void* id128, int rank, int nranks, std::string* err) {
  destroy();
  if (nranks < 1 || rank < 0 || rank >= nranks) return NDT_ERR_INVALID_ARG;
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclComm_t comm = nullptr;
  ncclResult_t rc = ncclCommInitRank(&comm, nranks, id, rank);
  if (rc != ncclSuccess) {
    if (err) *err = std::string("ncclCommInitRank: ") + ncclGetErrorString(rc);
    return NDT_ERR_COMM;
  }
  nccl_comm_ = comm;
  mode_ = NDT_REDUCE_RCCL;
  rank_ = rank;
  nranks_ = nranks;
  return NDT_OK;
}

int Reducer::init_shm(const char* name, int rank, int nranks, std::string* err) {
  destroy();
  if (!name || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks)
    return NDT_ERR_INVALID_ARG;
  const size_t bytes = sizeof(ShmSeg);
  const auto t_start = std::chrono::steady_clock::now();
  auto expired = [&](int seconds) { return std::chrono::steady_clock::now() - t_start > std::chrono::seconds(seconds); };
  void* p = MAP_FAILED;
  if (rank == 0) {
    shm_unlink(name);
    int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
      if (err) *err = std::string("shm_open/ftruncate failed for ") + name;
      if (fd >= 0) close(fd);
      return NDT_ERR_COMM;
    }
    p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
      if (err) *err = "mmap of shm segment failed";
      return NDT_ERR_COMM;
    }
    ShmSeg* seg = static_cast<ShmSeg*>(p);  // fresh pages: all zero
    seg->nranks = nranks;
    seg->magic.store(kMagic, std::memory_order_release);
    for (int r = 1; r < nranks; ++r) {  // echo every rank's attach word
      uint64_t v;
      while ((v = seg->attach[r].v.load(std::memory_order_acquire)) == 0) {
        if (expired(60)) {
          munmap(p, bytes);
          shm_unlink(name);
          if (err) *err = "timed out waiting for the other ranks to attach to the shm segment";
          return NDT_ERR_COMM;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
      seg->ack[r].v.store(v, std::memory_order_release);
    }
  } else {
    // unique to this init call (pid, rank, clock): no segment of an earlier run can hold it
    const uint64_t mine = (((uint64_t)getpid() << 32) ^ ((uint64_t)rank << 24) ^
                           (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ 0x9e3779b97f4a7c15ull) | 1ull;
    for (;;) {
      if (expired(60)) {
        if (err) *err = std::string("timed out attaching to shm segment ") + name;
        return NDT_ERR_COMM;
      }
      int fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) {
        if (fd >= 0) close(fd);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        continue;
      }
      p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      close(fd);
      if (p == MAP_FAILED) {
        if (err) *err = "mmap of shm segment failed";
        return NDT_ERR_COMM;
      }
      ShmSeg* seg = static_cast<ShmSeg*>(p);
      bool live = false;
      const auto t_try = std::chrono::steady_clock::now();
      while (std::chrono::steady_clock::now() - t_try < std::chrono::milliseconds(500)) {
        if (seg->magic.load(std::memory_order_acquire) == kMagic && seg->nranks == nranks) {
          seg->attach[rank].v.store(mine, std::memory_order_release);
          if (seg->ack[rank].v.load(std::memory_order_acquire) == mine) { live = true; break; }
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
      if (live) break;
      munmap(p, bytes);  // a stale segment, or rank 0 is not there yet: open the name again
      p = MAP_FAILED;
    }
  }
  shm_ = p;
  shm_bytes_ = bytes;
  shm_name_ = name;
  shm_round_ = 0;
  mode_ = NDT_REDUCE_SHM;
  rank_ = rank;
  nranks_ = nranks;
  return NDT_OK;
}

namespace {
std::string hip_err(const char* what, hipError_t e) { return std::string(what) + ": " + hipGetErrorString(e); }
}  // namespace

int Reducer::p2p_handle(void* out64, std::string* err) {
  static_assert(sizeof(hipIpcMemHandle_t) == NDT_P2P_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
  if (!out64) return NDT_ERR_INVALID_ARG;
  if (mode_ != NDT_REDUCE_NONE) destroy();
  if (!xarea_) {
    // fine-grained (uncached at the device's L2): peers write it over xGMI while a kernel polls it
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, XCHG_AREA_BYTES, hipDeviceMallocFinegrained);
    if (e != hipSuccess || !p) {
      (void)hipGetLastError();
      if (err) *err = hip_err("hipExtMallocWithFlags(fine-grained exchange area)", e);
      return NDT_ERR_COMM;
    }
    xarea_ = p;
  }
  hipError_t e = hipMemset(xarea_, 0, XCHG_AREA_BYTES);  // round tags start at 1: zeroed rows never match
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t hd;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&hd, xarea_);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (err) *err = hip_err("hipIpcGetMemHandle(exchange area)", e);
    (void)hipFree(xarea_);
    xarea_ = nullptr;
    return NDT_ERR_COMM;
  }
  std::memcpy(out64, &hd, sizeof(hd));
  return NDT_OK;
}

int Reducer::init_p2p(const void* handles, int rank, int nranks, std::string* err) {
  if (!handles || nranks < 1 || nranks > XCHG_MAX_RANKS || rank < 0 || rank >= nranks) return NDT_ERR_INVALID_ARG;
  if (!xarea_) {
    if (err) *err = "ndt_comm_init_p2p before ndt_comm_p2p_handle";
    return NDT_ERR_INVALID_ARG;
  }
  // Peer access (xGMI): kernels of this device will store into the other ranks' memory.  A transport that cannot work
  // is REFUSED here, with a reason (NDT_ERR_UNSUPPORTED): an area that cannot be opened, or one on a device this one
  // has no peer access to, would otherwise surface as a 20 ms kernel-side time-out per evaluation.
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto close_opened = [&](int upto) {
    for (int q = 0; q < upto; ++q)
      if (q != rank && xpeer_[q]) { (void)hipIpcCloseMemHandle(xpeer_[q]); xpeer_[q] = nullptr; }
  };
  XchgInfo info;
  std::memset(&info, 0, sizeof(info));
  info.rank = rank;
  info.nranks = nranks;
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { xpeer_[r] = xarea_; continue; }
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, static_cast<const char*>(handles) + (size_t)r * NDT_P2P_HANDLE_BYTES, sizeof(hd));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess || !p) {
      (void)hipGetLastError();
      if (err) *err = "p2p unavailable: " + hip_err(("hipIpcOpenMemHandle(exchange area of rank " + std::to_string(r) + ")").c_str(), e) +
                      " -- are all ranks' devices visible to every rank (no per-rank HIP_VISIBLE_DEVICES)?";
      close_opened(r);
      return NDT_ERR_UNSUPPORTED;
    }
    xpeer_[r] = p;
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof(attr));
    if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.device != dev) {
      int can = 0;
      hipError_t ea = hipDeviceCanAccessPeer(&can, dev, attr.device);
      if (ea != hipSuccess || !can) {
        (void)hipGetLastError();
        if (err) *err = "p2p unavailable: device " + std::to_string(dev) + " has no peer access to device " +
                        std::to_string(attr.device) + " (rank " + std::to_string(r) + ")";
        close_opened(r + 1);
        return NDT_ERR_UNSUPPORTED;
      }
      hipError_t ep = hipDeviceEnablePeerAccess(attr.device, 0);
      if (ep != hipSuccess && ep != hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        if (err) *err = "p2p unavailable: " + hip_err("hipDeviceEnablePeerAccess", ep);
        close_opened(r + 1);
        return NDT_ERR_UNSUPPORTED;
      }
      (void)hipGetLastError();
    } else {
      (void)hipGetLastError();
    }
  }
  for (int r = 0; r < nranks; ++r) info.area[r] = (unsigned long long)reinterpret_cast<uintptr_t>(xpeer_[r]);
  hipError_t e = hipSuccess;
  if (!xstats_) e = hipMalloc(&xstats_, 4 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(xstats_, 0, 4 * sizeof(unsigned long long));
  info.stats = (unsigned long long)reinterpret_cast<uintptr_t>(xstats_);
  if (e == hipSuccess && !xinfo_dev_) e = hipMalloc(&xinfo_dev_, sizeof(XchgInfo));
  if (e == hipSuccess) e = hipMemcpy(xinfo_dev_, &info, sizeof(info), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (err) *err = hip_err("exchange info upload", e);
    for (int r = 0; r < nranks; ++r)
      if (r != rank && xpeer_[r]) { (void)hipIpcCloseMemHandle(xpeer_[r]); xpeer_[r] = nullptr; }
    return NDT_ERR_COMM;
  }
  mode_ = NDT_REDUCE_P2P;
  rank_ = rank;
  nranks_ = nranks;
  xround_ = 0;
  xbround_ = 0;
  return NDT_OK;
}

#ifdef __HIPCC__
namespace {
struct RowArg { unsigned long long w[2 * NDT_EVAL_WORDS]; };
// The host leg of the exchange writes a row exactly as the derivative kernel does: lane v stores slot v {round, value}
// with ONE 16-byte system-scope store into every rank's area -- a reader sees a slot entirely old or entirely new.  (A
// 512-byte hipMemcpy into a peer's mapping promises nothing about the order in which its bytes land: ADVICE r03.)
__global__ void __launch_bounds__(64) k_publish_row(XchgInfo info, unsigned long long round, RowArg row) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int v = threadIdx.x;
  if (v >= NDT_EVAL_WORDS) return;
  u32x4 d;
  d.x = (unsigned int)row.w[2 * v]; d.y = (unsigned int)(row.w[2 * v] >> 32);
  d.z = (unsigned int)row.w[2 * v + 1]; d.w = (unsigned int)(row.w[2 * v + 1] >> 32);
  for (int r = 0; r < info.nranks; ++r) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(info.area[r]), 0, 0xFFFFFFFFu, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, xchg_slot_offset(round, info.rank, v), 0, 17 /* sc0 sc1: system scope */);
  }
}
}  // namespace
#endif

int Reducer::p2p_stats(unsigned long long out[4], bool reset, std::string* err) {
  if (mode_ != NDT_REDUCE_P2P || !xstats_ || !out) return NDT_ERR_INVALID_ARG;
  hipError_t e = hipMemcpy(out, xstats_, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset) e = hipMemset(xstats_, 0, 4 * sizeof(unsigned long long));
  if (e != hipSuccess) {
    if (err) *err = hip_err("reading the exchange counters", e);
    return NDT_ERR_COMM;
  }
  return NDT_OK;
}

#ifdef __HIPCC__
namespace {
// value of word v that rank w writes under tag t: every bit depends on every input (splitmix64)
__host__ __device__ inline unsigned long long selftest_pattern(unsigned long long t, int w, int v) {
  unsigned long long z = t * 0x9E3779B97F4A7C15ull + (unsigned long long)(w * 64 + v) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// One wave per rank; lane v < 32 owns word v.  Round k (tag = base + k): the own row goes to every rank's area, then the
// rows of all ranks are polled in the own area -- the loads and stores of xchg_allsum (ndt_derivs.hip), the same
// offsets, the same two generations (a rank can be one round ahead of the slowest at most).  A slot whose tag is this
// round's and whose value is not the pattern was seen half-written.
__global__ void __launch_bounds__(64) k_xchg_selftest(XchgInfo info, unsigned long long base, int rounds, unsigned long long* out) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int v = threadIdx.x, n = info.nranks, me = info.rank;
  unsigned long long torn = 0ull, done = 0ull, late = 0ull, longest = 0ull;
  const __amdgpu_buffer_rsrc_t own = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(info.area[me]), 0, 0xFFFFFFFFu, 0x00020000);
  for (int k = 1; k <= rounds; ++k) {   // (wave-uniform control flow: every lane takes the same exits)
    const unsigned long long tag = base + (unsigned long long)k;
    if (v < NDT_EVAL_WORDS) {
      const unsigned long long bits = selftest_pattern(tag, me, v);
      u32x4 d;
      d.x = (unsigned int)tag; d.y = (unsigned int)(tag >> 32); d.z = (unsigned int)bits; d.w = (unsigned int)(bits >> 32);
      for (int r = 0; r < n; ++r) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(info.area[r]), 0, 0xFFFFFFFFu, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, xchg_slot_offset(tag, me, v), 0, 17 /* sc0 sc1 */);
      }
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool timed_out = false;
    for (int r = 0; r < n; ++r) {
      for (;;) {
        asm volatile("" ::: "memory");
        u32x4 q;
        q.x = (unsigned int)tag; q.y = (unsigned int)(tag >> 32); q.z = 0u; q.w = 0u;
        if (v < NDT_EVAL_WORDS) q = __builtin_amdgcn_raw_buffer_load_b128(own, xchg_slot_offset(tag, r, v), 0, 17);
        const bool here = q.x == (unsigned int)tag && q.y == (unsigned int)(tag >> 32);
        if (__ballot(here) == ~0ull) {
          const unsigned long long got = ((unsigned long long)q.w << 32) | q.z;
          if (v < NDT_EVAL_WORDS && got != selftest_pattern(tag, r, v)) ++torn;
          break;
        }
        if (__builtin_amdgcn_s_memrealtime() - t0 > XCHG_TIMEOUT_TICKS * 50ull) { timed_out = true; break; }   // 1 s
        __builtin_amdgcn_s_sleep(1);
      }
      if (timed_out) break;
    }
    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
    longest = dt > longest ? dt : longest;
    if (timed_out) { ++late; break; }
    ++done;
  }
  // lane sums of the torn slots; the other counters are wave-uniform
  for (int off = 32; off > 0; off >>= 1) torn += __shfl_xor(torn, off);
  if (v == 0) { out[0] = done; out[1] = torn; out[2] = late; out[3] = longest; }
}
}  // namespace
#endif

int Reducer::p2p_selftest(int rounds, unsigned long long out[4], std::string* err) {
  if (mode_ != NDT_REDUCE_P2P || rounds < 1 || !out) return NDT_ERR_INVALID_ARG;
#ifdef __HIPCC__
  XchgInfo info;
  std::memset(&info, 0, sizeof(info));
  info.rank = rank_;
  info.nranks = nranks_;
  for (int r = 0; r < nranks_; ++r) info.area[r] = (unsigned long long)reinterpret_cast<uintptr_t>(xpeer_[r]);
  unsigned long long* d_out = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), 4 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(d_out, 0, 4 * sizeof(unsigned long long));
  if (e == hipSuccess) {
    // tags far above any evaluation count, a fresh range per pass; the slots they leave behind never match a real round.
    // EVEN base: round k uses generation k & 1 on every rank
    const unsigned long long base = (1ull << 62) + (xtests_++ << 32);
    hipLaunchKernelGGL(k_xchg_selftest, dim3(1), dim3(64), 0, nullptr, info, base, rounds, d_out);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
  }
  if (e == hipSuccess) e = hipMemcpy(out, d_out, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (d_out) (void)hipFree(d_out);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (err) *err = hip_err("exchange-area integrity pass", e);
    return NDT_ERR_COMM;
  }
  return NDT_OK;
#else
  (void)err;
  return NDT_ERR_UNSUPPORTED;
#endif
}

// one row {round, value} x n of the own rank into every rank's area, from the host (batched evaluations)
int Reducer::p2p_publish_from_host(uint64_t round, const double* words, int n, std::string* err) {
  unsigned long long row[2 * NDT_EVAL_WORDS];
  for (int v = 0; v < NDT_EVAL_WORDS; ++v) {
    row[2 * v] = round;
    double w = v < n ? words[v] : 0.0;
    std::memcpy(&row[2 * v + 1], &w, sizeof(double));
  }
#ifdef __HIPCC__
  {
    XchgInfo info;
    std::memset(&info, 0, sizeof(info));
    info.rank = rank_;
    info.nranks = nranks_;
    for (int r = 0; r < nranks_; ++r) info.area[r] = (unsigned long long)reinterpret_cast<uintptr_t>(xpeer_[r]);
    RowArg ra;
    std::memcpy(ra.w, row, sizeof(row));
    hipLaunchKernelGGL(k_publish_row, dim3(1), dim3(64), 0, nullptr, info, (unsigned long long)round, ra);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
      if (err) *err = hip_err("peer-write from the host (publish kernel)", e);
      return NDT_ERR_COMM;
    }
    return NDT_OK;
  }
#endif
  for (int r = 0; r < nranks_; ++r) {
    hipError_t e = hipMemcpy(static_cast<char*>(xpeer_[r]) + xchg_slot_offset(round, rank_, 0), row, sizeof(row), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      if (err) *err = hip_err("peer-write from the host", e);
      return NDT_ERR_COMM;
    }
  }
  return NDT_OK;
}

int Reducer::p2p_finish_on_host(uint64_t round, double* words, int n, std::string* err) {
  if (mode_ != NDT_REDUCE_P2P || n > NDT_EVAL_WORDS) return NDT_ERR_INVALID_ARG;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned long long rows[XCHG_MAX_RANKS][2 * NDT_EVAL_WORDS];
  for (;;) {
    // the generation's rows of ranks 0 .. nranks-1 are contiguous
    hipError_t e = hipMemcpy(rows, static_cast<const char*>(xarea_) + xchg_slot_offset(round, 0, 0),
                             (size_t)nranks_ * sizeof(rows[0]), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      if (err) *err = hip_err("reading the exchange area", e);
      return NDT_ERR_COMM;
    }
    bool ok = true;
    for (int r = 0; r < nranks_ && ok; ++r)
      for (int v = 0; v < NDT_EVAL_WORDS && ok; ++v) ok = rows[r][2 * v] == round;
    if (ok) break;
    if (std::chrono::steady_clock::now() - t0 > wait_limit()) {
      if (err) *err = "peer-write all-reduce timed out waiting for a rank";
      return NDT_ERR_COMM;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  for (int v = 0; v < n; ++v) {
    double s = 0.0;
    for (int r = 0; r < nranks_; ++r) {
      double w;
      std::memcpy(&w, &rows[r][2 * v + 1], sizeof(double));
      s += w;
    }
    words[v] = s;
  }
  return NDT_OK;
}

#ifdef __HIPCC__
namespace {
// block k publishes pose k's row: lane v stores slot v {round, value} into every rank's batch region
__global__ void __launch_bounds__(64) k_publish_batch(XchgInfo info, unsigned long long round, const double* __restrict__ vals) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int v = threadIdx.x, k = blockIdx.x;
  if (v >= NDT_EVAL_WORDS) return;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(vals[(size_t)k * NDT_EVAL_WORDS + v]);
  u32x4 d;
  d.x = (unsigned int)round; d.y = (unsigned int)(round >> 32); d.z = (unsigned int)bits; d.w = (unsigned int)(bits >> 32);
  for (int r = 0; r < info.nranks; ++r) {
    char* base = reinterpret_cast<char*>(info.area[r]) + xchg_batch_offset(round, k, info.rank, v);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 16u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(d, rs, 0, 0, 17 /* sc0 sc1: system scope */);
  }
}
}  // namespace
#endif

int Reducer::allreduce_host_batch(double* words, int K, std::string* err) {
  if (K <= 0) return NDT_OK;
#ifdef __HIPCC__
  if (mode_ == NDT_REDUCE_P2P) {
    if (!xstage_) {
      hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&xstage_), (size_t)XCHG_BATCH_MAX * NDT_EVAL_WORDS * sizeof(double), hipHostMallocMapped);
      if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&xback_), (size_t)XCHG_BATCH_MAX * XCHG_MAX_RANKS * 2 * NDT_EVAL_WORDS * sizeof(unsigned long long), 0);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        if (err) *err = hip_err("pinned staging of the batched exchange", e);
        return NDT_ERR_COMM;
      }
    }
    XchgInfo info;
    std::memset(&info, 0, sizeof(info));
    info.rank = rank_;
    info.nranks = nranks_;
    for (int r = 0; r < nranks_; ++r) info.area[r] = (unsigned long long)reinterpret_cast<uintptr_t>(xpeer_[r]);
    for (int k0 = 0; k0 < K; k0 += XCHG_BATCH_MAX) {
      const int kb = std::min(XCHG_BATCH_MAX, K - k0);
      const uint64_t round = ++xbround_;
      std::memcpy(xstage_, words + (size_t)k0 * NDT_EVAL_WORDS, (size_t)kb * NDT_EVAL_WORDS * sizeof(double));
      double* stage_dev = nullptr;
      hipError_t e = hipHostGetDevicePointer(reinterpret_cast<void**>(&stage_dev), xstage_, 0);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_publish_batch, dim3((unsigned)kb), dim3(64), 0, nullptr, info, (unsigned long long)round, (const double*)stage_dev);
        e = hipGetLastError();
      }
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
      if (e != hipSuccess) {
        if (err) *err = hip_err("batched peer-write from the host", e);
        return NDT_ERR_COMM;
      }
      // gather: the rows of poses 0 .. kb-1, ranks 0 .. nranks-1 of this generation of the OWN area (one strided copy per poll)
      const size_t row_bytes = (size_t)2 * NDT_EVAL_WORDS * sizeof(unsigned long long);   // 512
      const size_t width = (size_t)nranks_ * row_bytes, pitch = (size_t)XCHG_MAX_RANKS * row_bytes;
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        e = hipMemcpy2D(xback_, width, static_cast<const char*>(xarea_) + xchg_batch_offset(round, 0, 0, 0), pitch, width, (size_t)kb,
                        hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
          if (err) *err = hip_err("reading the batched exchange area", e);
          return NDT_ERR_COMM;
        }
        bool ok = true;
        for (size_t i = 0; i < (size_t)kb * nranks_ * NDT_EVAL_WORDS && ok; ++i) ok = xback_[2 * i] == round;
        if (ok) break;
        if (std::chrono::steady_clock::now() - t0 > wait_limit()) {
          if (err) *err = "peer-write all-reduce (batched) timed out waiting for a rank";
          return NDT_ERR_COMM;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      for (int k = 0; k < kb; ++k)
        for (int v = 0; v < NDT_EVAL_WORDS; ++v) {
          double s = 0.0;
          for (int r = 0; r < nranks_; ++r) {   // rank order: every rank obtains bit-identical sums
            double w;
            std::memcpy(&w, &xback_[(((size_t)k * nranks_ + r) * NDT_EVAL_WORDS + v) * 2 + 1], sizeof(double));
            s += w;
          }
          words[(size_t)(k0 + k) * NDT_EVAL_WORDS + v] = s;
        }
    }
    return NDT_OK;
  }
#endif
  for (int k = 0; k < K; ++k) {
    int rc = allreduce_host(words + (size_t)k * NDT_EVAL_WORDS, NDT_EVAL_WORDS, err);
    if (rc) return rc;
  }
  return NDT_OK;
}

int Reducer::init_hook(ndt_allreduce_fn fn, void* ctx, int rank, int nranks) {
  destroy();
  if (!fn || nranks < 1 || rank < 0 || rank >= nranks) return NDT_ERR_INVALID_ARG;
  hook_ = fn;
  hook_ctx_ = ctx;
  mode_ = NDT_REDUCE_HOOK;
  rank_ = rank;
  nranks_ = nranks;
  return NDT_OK;
}

void Reducer::destroy() {
  if (nccl_comm_) {
    ncclCommDestroy(static_cast<ncclComm_t>(nccl_comm_));
    nccl_comm_ = nullptr;
  }
  if (shm_) {
    munmap(shm_, shm_bytes_);
    if (rank_ == 0) shm_unlink(shm_name_.c_str());
    shm_ = nullptr;
  }
  if (mode_ == NDT_REDUCE_P2P || xarea_) {
    for (int r = 0; r < XCHG_MAX_RANKS; ++r) {
      if (xpeer_[r] && xpeer_[r] != xarea_) (void)hipIpcCloseMemHandle(xpeer_[r]);
      xpeer_[r] = nullptr;
    }
    if (xinfo_dev_) { (void)hipFree(xinfo_dev_); xinfo_dev_ = nullptr; }
    if (xstats_) { (void)hipFree(xstats_); xstats_ = nullptr; }
    if (xarea_) { (void)hipFree(xarea_); xarea_ = nullptr; }
    if (xstage_) { (void)hipHostFree(xstage_); xstage_ = nullptr; }
    if (xback_) { (void)hipHostFree(xback_); xback_ = nullptr; }
    xround_ = 0;
    xbround_ = 0;
  }
  hook_ = nullptr;
  hook_ctx_ = nullptr;
  mode_ = NDT_REDUCE_NONE;
  rank_ = 0;
  nranks_ = 1;
}

int Reducer::allreduce_device(double* d_words, int n, hipStream_t s, std::string* err) {
  if (mode_ != NDT_REDUCE_RCCL) return NDT_ERR_INVALID_ARG;
  ncclResult_t rc = ncclAllReduce(d_words, d_words, (size_t)n, ncclDouble, ncclSum,
                                  static_cast<ncclComm_t>(nccl_comm_), s);
  if (rc != ncclSuccess) {
    if (err) *err = std::string("ncclAllReduce: ") + ncclGetErrorString(rc);
    return NDT_ERR_COMM;
  }
  return NDT_OK;
}

int Reducer::allreduce_host(double* words, int n, std::string* err) {
  if (mode_ == NDT_REDUCE_NONE) return NDT_OK;
  if (mode_ == NDT_REDUCE_HOOK) {
    if (hook_(hook_ctx_, words, n) != 0) {
      if (err) *err = "all-reduce hook reported failure";
      return NDT_ERR_COMM;
    }
    return NDT_OK;
  }
  if (mode_ == NDT_REDUCE_P2P) {
    // host-side round of the same exchange (batched evaluations; single-pose launches exchange inside
    // the kernel): publish the row into every rank's area, gather the own area, add in rank order
    if (n > NDT_EVAL_WORDS) return NDT_ERR_INVALID_ARG;
    const uint64_t round = ++xround_;
    int rc = p2p_publish_from_host(round, words, n, err);
    if (rc) return rc;
    return p2p_finish_on_host(round, words, n, err);
  }
  if (mode_ != NDT_REDUCE_SHM || n > NDT_EVAL_WORDS) return NDT_ERR_INVALID_ARG;
  // Every rank publishes its partial in the slot of this round's parity, bumps
  // its sequence word, waits for all sequence words, then sums the slots in
  // rank order (so every rank obtains bit-identical sums).  A rank can be at
  // most one round ahead of the slowest, hence two slot generations suffice.
  ShmSeg* seg = static_cast<ShmSeg*>(shm_);
  const uint64_t round = ++shm_round_;
  const int gen = (int)(round & 1);
  std::memcpy(seg->slot[gen][rank_], words, sizeof(double) * n);
  seg->seq[rank_].v.store(round, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < nranks_; ++r) {
    int spins = 0;
    while (seg->seq[r].v.load(std::memory_order_acquire) < round) {
      if (++spins > 4096) {
        spins = 0;
        if (std::chrono::steady_clock::now() - t0 > wait_limit()) {
          if (err) *err = "shared-memory all-reduce timed out";
          return NDT_ERR_COMM;
        }
        std::this_thread::yield();
      }
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int r = 0; r < nranks_; ++r) s += seg->slot[gen][r][i];
    words[i] = s;
  }
  return NDT_OK;
}

}  // namespace ndt
