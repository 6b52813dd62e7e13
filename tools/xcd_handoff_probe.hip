// How long does a tagged 16-byte slot take from one compute unit to another -- on the SAME XCD and on ANOTHER one?
// (Round 5: k_derivatives' rows go from 241 point blocks to the summing blocks through such slots; the stamps put a
// poll trip at ~1 us in the diagnostic build.  This probe prices the hop itself, what a summing block PER XCD could gain, and a trip on an idle device.)
//
// Two workgroups of one wave play ping-pong through two slots {tag, value}: A stores round r into slot 0, B polls
// slot 0 until it carries r and stores r into slot 1, A polls slot 1.  One-way latency = round trip / 2.  The partner
// is chosen by its XCC_ID (read from the hardware register, not assumed from the workgroup id): same XCD / another XCD.
// Stores are device scope (sc1), as k_derivatives' row stores; the polls are issued three ways: sc1 (device scope, what the
// engine uses), sc0 sc1 (system scope) and sc0 only (workgroup scope: may be served by the XCD's L2).  Every wait is
// bounded (20 ms); a pair that times out is reported as such -- with sc0 polls across XCDs that is the expected outcome
// (the poll may be served from a stale line), and it is why the engine could only use them for rows of its own XCD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000);
}

__global__ void k_where(unsigned int* xcc, unsigned int* hwid) {
  if (threadIdx.x == 0) {
    xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xf;   // HW_REG_XCC_ID
    hwid[blockIdx.x] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID
  }
}

// AUX: 16 = sc1 (device scope), 17 = sc0 sc1 (system scope), 1 = sc0 (workgroup scope)
template <int AUX>
__global__ void k_pingpong(unsigned long long* slots /* 2 x 128 B apart */, int a, int b, int rounds, unsigned long long base,
                           unsigned long long* out /* [0] ticks of A's loop, [1] rounds completed, [2] poll trips of A */) {
  const int me = (int)blockIdx.x == a ? 0 : ((int)blockIdx.x == b ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  const __amdgpu_buffer_rsrc_t r = rsrc(slots);
  const unsigned int mine = me == 0 ? 0u : 128u, theirs = me == 0 ? 128u : 0u;
  unsigned long long trips = 0ull;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int done = 0;
  for (int k = 1; k <= rounds; ++k) {
    const unsigned long long tag = base + (unsigned long long)k;
    u32x4 d;
    d.x = (unsigned int)tag; d.y = (unsigned int)(tag >> 32); d.z = (unsigned int)k; d.w = 0u;
    if (me == 0) __builtin_amdgcn_raw_buffer_store_b128(d, r, mine, 0, 16);
    bool ok = false;
    const unsigned long long tw = __builtin_amdgcn_s_memrealtime();
    for (;;) {
      asm volatile("" ::: "memory");
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(r, theirs, 0, AUX);
      ++trips;
      if (q.x == (unsigned int)tag && q.y == (unsigned int)(tag >> 32)) { ok = true; break; }
      if (__builtin_amdgcn_s_memrealtime() - tw > 2000000ull) break;   // 20 ms: every wave reaches an exit
    }
    if (!ok) break;
    if (me == 1) __builtin_amdgcn_raw_buffer_store_b128(d, r, mine, 0, 16);
    done = k;
  }
  if (me == 0) {
    out[0] = __builtin_amdgcn_s_memrealtime() - t0;
    out[1] = (unsigned long long)done;
    out[2] = trips;
  }
}

// One poll TRIP of a summing block on an idle device: `waves` waves, every lane `per_lane` 16-byte device-scope loads in flight
// (lane l of a wave reads slot l of `per_lane` rows of 512 bytes: with 8-lane columns a wave-instruction touches 8 lines, with
// 32-lane columns 2 rows = 8 lines as well), then one wait -- as sum_rows does.  Ticks per trip over `trips` trips.
__global__ void k_polltrip(const unsigned long long* rows, int per_lane, int lanes_per_col, int ncols_total, int trips, unsigned long long* out) {
  const __amdgpu_buffer_rsrc_t r = rsrc(rows);
  const int col = (int)threadIdx.x / lanes_per_col, v = (int)threadIdx.x % lanes_per_col;
  unsigned int acc = 0u;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
    asm volatile("" ::: "memory");
    u32x4 q[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < per_lane) q[k] = __builtin_amdgcn_raw_buffer_load_b128(r, ((unsigned int)(col + k * ncols_total) * 32u + (unsigned int)v) * 16u, 0, 16);
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < per_lane) acc += q[k].x;
    __builtin_amdgcn_s_sleep(1);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int AUX>
static int run(const char* what, unsigned long long* slots, unsigned long long* out, int nblocks, int a, int b, int rounds,
               unsigned long long* base) {
  CHECK(hipMemset(out, 0, 3 * sizeof(unsigned long long)));
  hipLaunchKernelGGL(k_pingpong<AUX>, dim3(nblocks), dim3(64), 0, 0, slots, a, b, rounds, *base, out);
  *base += (unsigned long long)rounds + 16ull;
  CHECK(hipDeviceSynchronize());
  unsigned long long h[3];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  if ((int)h[1] < rounds) printf("  %-34s TIMED OUT after %llu of %d rounds\n", what, h[1], rounds);
  else printf("  %-34s one way %.2f us  (%.1f polls per hop)\n", what, 0.01 * (double)h[0] / (2.0 * rounds), (double)h[2] / rounds);
  return 0;
}

int main() {
  const int nblocks = 64, rounds = 2000;
  unsigned int *xcc = nullptr, *hwid = nullptr;
  unsigned long long *slots = nullptr, *out = nullptr;
  CHECK(hipMalloc(&xcc, nblocks * 4)); CHECK(hipMalloc(&hwid, nblocks * 4));
  CHECK(hipMalloc(&slots, 4096)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(slots, 0, 4096));
  hipLaunchKernelGGL(k_where, dim3(nblocks), dim3(64), 0, 0, xcc, hwid);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned int> x(nblocks), hw(nblocks);
  CHECK(hipMemcpy(x.data(), xcc, nblocks * 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hw.data(), hwid, nblocks * 4, hipMemcpyDeviceToHost));
  printf("workgroup -> XCC_ID of a %d-block launch:", nblocks);
  for (int i = 0; i < 24; ++i) printf(" %u", x[i]);
  printf(" ...\n");
  // (placement of a LATER launch of the same shape repeats this one's on an otherwise idle device: round-robin over the XCDs)
  int same = -1, other = -1;
  for (int i = 1; i < nblocks && (same < 0 || other < 0); ++i) {
    if (same < 0 && x[i] == x[0] && hw[i] != hw[0]) same = i;
    if (other < 0 && x[i] != x[0]) other = i;
  }
  printf("workgroup 0 on XCD %u; partner on the same XCD: workgroup %d; on another XCD: workgroup %d (XCD %u)\n", x[0], same, other,
         other >= 0 ? x[other] : 0u);
  unsigned long long base = 1ull << 32;
  for (int rep = 0; rep < 2; ++rep) {
    if (same > 0) {
      printf("same XCD (workgroups 0 and %d):\n", same);
      if (run<16>("polls sc1 (device scope)", slots, out, nblocks, 0, same, rounds, &base)) return 1;
      if (run<17>("polls sc0 sc1 (system scope)", slots, out, nblocks, 0, same, rounds, &base)) return 1;
      if (run<1>("polls sc0 (workgroup scope)", slots, out, nblocks, 0, same, rounds, &base)) return 1;
    }
    if (other > 0) {
      printf("another XCD (workgroups 0 and %d):\n", other);
      if (run<16>("polls sc1 (device scope)", slots, out, nblocks, 0, other, rounds, &base)) return 1;
      if (run<17>("polls sc0 sc1 (system scope)", slots, out, nblocks, 0, other, rounds, &base)) return 1;
      if (run<1>("polls sc0 (workgroup scope)", slots, out, nblocks, 0, other, rounds, &base)) return 1;
    }
  }
  // poll trips on an idle device
  unsigned long long* rows = nullptr;
  CHECK(hipMalloc(&rows, 1 << 20));
  CHECK(hipMemset(rows, 0, 1 << 20));
  printf("one poll trip of ONE block on an idle device (device-scope 16-byte loads, all in flight, one wait), us per trip:\n");
  struct Shape { const char* what; int threads, per_lane, lanes_per_col, ncols; } shapes[] = {
    {"832 threads, 32-lane columns, 10 rows per lane (one summing block, 241 rows: 964 lines)", 832, 10, 32, 26},
    {"208 threads,  8-lane columns, 10 rows per lane (a quarter of the words, 241 rows: 241 lines)", 208, 10, 8, 26},
    {"208 threads,  8-lane columns,  4 rows per lane", 208, 4, 8, 26},
    {"208 threads,  8-lane columns,  1 row per lane", 208, 1, 8, 26},
    {"832 threads,  8-lane columns,  3 rows per lane (the same 241 lines over four times the lanes)", 832, 3, 8, 104},
    {" 64 threads,  8-lane columns,  1 row per lane (8 lines)", 64, 1, 8, 8},
    {" 64 threads,  8-lane columns, 16 rows per lane (128 lines)", 64, 16, 8, 8},
  };
  for (int rep = 0; rep < 2; ++rep)
    for (const Shape& sh : shapes) {
      CHECK(hipMemset(out, 0, 16));
      hipLaunchKernelGGL(k_polltrip, dim3(1), dim3(sh.threads), 0, 0, rows, sh.per_lane, sh.lanes_per_col, sh.ncols, 2000, out);
      CHECK(hipDeviceSynchronize());
      unsigned long long h[2];
      CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
      printf("  %-96s %.3f\n", sh.what, 0.01 * (double)h[0] / 2000.0);
    }
  return 0;
}
