// ndt_target.hip -- target voxel-grid build on gfx950 (MI355X).
//
// What it computes is the reference's VoxelGridCovariance::applyFilter
// (ref: extern/svn_ndt/include/voxel_grid_covariance_impl.hpp:77-379): bounds,
// integer grid, per-voxel point count / mean / 3x3 covariance, eigenvalue
// inflation, inverse covariance, validity filtering.  How it computes it is
// MI355X-first and shares nothing with the reference's single-threaded
// hash-map loop:
//   1. bounds        : one streaming pass, wave-shuffle min/max, one row per block; the block that
//                      draws the last ticket folds the rows and derives the grid geometry and the
//                      sort plan ON THE DEVICE (BuildGeom) -- no host round trip in mid-build; the
//                      same launch resets the cells the previous build published
//   2. cell keys     : one streaming pass (f32 floor, bit-compatible with the ref)
//   3. stable LSD radix sort of (cell, point index), hand-written (count / scan /
//                      scatter per 8-bit digit, wave-ballot ranking) -- points of one
//                      voxel become contiguous and stay in input order (deterministic sums)
//   4. run detection : run tails find their head through a wave ballot; leaf slots by
//                      count (+ scan by the last block) / emit (ascending cell order)
//   5. leaf sums     : 8 lanes per voxel gather + reduce sum(x), sum(x x^T) in f64
//   6. leaf finalise : one thread per voxel: 3x3 Jacobi eigen-solve / inflation /
//                      inverse; publishes an 80-byte VoxelRecord and the dense
//                      cell -> leaf index.
// Compiled with -ffp-contract=off: f32 index arithmetic must round as written.
#include "ndt_kernels.h"
#include "ndt_tuning.h"

#include <climits>
#include <cstdlib>
#include <cstring>

namespace ndt {

namespace {

__device__ __forceinline__ int encode_ordered(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}

__device__ __forceinline__ bool finite3(float a, float b, float c) {
  return isfinite(a) && isfinite(b) && isfinite(c);
}

constexpr int BOUNDS_BLOCKS = 512;

#ifdef NDT_STAMPS
// diagnostic build only: 100 MHz wall-clock stamps of thread 0 of every block of the fused build
// kernels (slot 0-2: sort pass 0-2, 3: run search), written to a side buffer no other code reads
__device__ unsigned long long g_bstamps[6 * 512 * 8];  // slots 4 / 5: k_bucket_pass / k_bucket_leaves
#define NDT_BSTAMP(slot, k)                                                                  \
  do {                                                                                       \
    if (threadIdx.x == 0 && blockIdx.x < 512)                                                \
      g_bstamps[((slot) * 512 + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
// per-WAVE stamps of k_bucket_leaves' blocks 0..31 (16 waves each), in the rows of slot 0 (the sort passes' slot:
// they do not run in a bucketed build)
#define NDT_WSTAMP(k)                                                                                   \
  do {                                                                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 32)                                                     \
      g_bstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();   \
  } while (0)
#else
#define NDT_BSTAMP(slot, k) do { } while (0)
#define NDT_WSTAMP(k) do { } while (0)
#endif

__device__ __forceinline__ float decode_ordered_dev(int enc) {
  return __int_as_float(enc >= 0 ? enc : enc ^ 0x7fffffff);
}

// Agent-scope accesses for the "last block finishes the job" hand-offs of the build kernels:
// a block publishes its partial with agent-scope stores by ONE thread, that thread waits for
// them to be acknowledged and takes a relaxed ticket, and the block that draws the last ticket
// reads the partials with agent-scope loads (cdna_hip_programming.md Guideline 16).
__device__ __forceinline__ void st_agent(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_agent(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool last_ticket(unsigned int* ticket, unsigned int nblocks) {
  // the partial was written with agent-scope (write-through) stores by THIS thread: once they are
  // acknowledged (vmcnt drained) a relaxed ticket is enough -- no release fence, whose L2
  // write-back cost ~7 us per kernel here -- and the reader uses agent-scope loads
  __builtin_amdgcn_s_waitcnt(0);
  const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return t == nblocks - 1u;
}

// Grid geometry in f32 exactly as the reference computes it on the host
// (ref: voxel_grid_covariance_impl.hpp:108-140), plus the sort plan.  planned_passes > 0: the
// host has already enqueued that many digit passes (sized for `cell_capacity` cells).
__device__ void derive_geometry(const int mnmx[6], int n_finite, float leaf, float inv_leaf, long long cell_capacity,
                                int planned_passes, BuildGeom* out) {
  BuildGeom b;
  GridGeom& g = b.g;
  g.leaf = leaf;
  g.inv_leaf = inv_leaf;
  b.n_finite = n_finite;
  b.status = BG_OK;
  b.bits = 1;
  b.passes = planned_passes > 0 ? planned_passes : 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { b.width[i] = 0; b.shift[i] = 0; }
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.min_b[a] = 0; g.div_b[a] = 0; g.lo[a] = 0.0f; g.hi[a] = 0.0f; b.max_b[a] = 0; }
  g.mul1 = g.mul2 = g.ncells = 0;
  if (n_finite == 0) {
    b.status = BG_NO_FINITE;
  } else {
    float mn[3], mx[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = decode_ordered_dev(mnmx[a]); mx[a] = decode_ordered_dev(mnmx[3 + a]); }
    long long d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = (long long)((mx[a] - mn[a]) * inv_leaf) + 1;
    const long long lim = 2147483647ll;
    if (d[0] < 0 || d[1] < 0 || d[2] < 0 || d[0] > lim || d[1] > lim || d[2] > lim || d[0] * d[1] > lim ||
        d[0] * d[1] * d[2] > lim) {
      b.status = BG_OVERFLOW;
    } else {
      long long ncells = 1;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        g.min_b[a] = (int)floorf(mn[a] * inv_leaf);
        b.max_b[a] = (int)floorf(mx[a] * inv_leaf);
        g.div_b[a] = b.max_b[a] - g.min_b[a] + 1;
        g.lo[a] = (float)g.min_b[a] * leaf;
        g.hi[a] = (float)(b.max_b[a] + 1) * leaf;
        ncells *= g.div_b[a];
      }
      if (ncells >= lim) {
        b.status = BG_OVERFLOW;
      } else {
        g.mul1 = g.div_b[0];
        g.mul2 = g.div_b[0] * g.div_b[1];
        g.ncells = (int)ncells;
        if (ncells > cell_capacity) b.status = BG_CAPACITY;
        int bits = 1;
        while (bits < 32 && (1ull << bits) <= (unsigned long long)ncells) ++bits;  // the sentinel key is `ncells`
        b.bits = bits;
        const int need = (bits + 7) / 8;
        if (planned_passes <= 0) b.passes = need;
        else if (need > planned_passes && b.status == BG_OK) b.status = BG_PASSES;
        const int base = bits / b.passes, rem = bits % b.passes;
        int sh = 0;
        for (int i = 0; i < b.passes && i < 4; ++i) {
          b.width[i] = base + (i < rem ? 1 : 0);
          b.shift[i] = sh;
          sh += b.width[i];
        }
      }
    }
  }
  *out = b;
}

// ref: pcl::getMinMax3D at voxel_grid_covariance_impl.hpp:103 (non-finite skipped).
// One row of 8 ints per block {min xyz, max xyz, #finite, 0}; no atomics on the values (every
// block contending on the same 7 words cost 0.65 ms for 1M points).  The block that draws the
// last ticket folds the <= 512 rows, derives the geometry and zeroes the leaf counters; every
// block also resets its share of the cells the PREVIOUS build published (the dense grid is
// filled with -1 once per allocation, not per build).
template <int U>
__global__ void __launch_bounds__(256) k_bounds(const float* __restrict__ x, const float* __restrict__ y,
                                               const float* __restrict__ z, size_t n, int* __restrict__ rows,
                                               unsigned int* __restrict__ ticket, float leaf, float inv_leaf,
                                               long long cell_capacity, int planned_passes,
                                               BuildGeom* __restrict__ gd, BuildGeom* __restrict__ gd_host,
                                               const LeafStats* __restrict__ old_stats, int dirty_slots,
                                               int* __restrict__ cell2leaf, size_t c2l_cap, int* __restrict__ nleaf) {
  __shared__ int lds[4][8];
  __shared__ int s_last;
  for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < dirty_slots; slot += gridDim.x * blockDim.x) {
    const int cell = old_stats[slot].cell;
    if (cell >= 0 && (size_t)cell < c2l_cap) cell2leaf[cell] = -1;
  }
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX};
  int mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  int cnt = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  // U strided points per trip: 3 U loads in flight instead of 3 (the pass is latency-bound)
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += U * stride) {
    float a[U], b[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      const size_t j = i < n ? i : i0;
      a[u] = x[j]; b[u] = y[j]; c[u] = z[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (i0 + u * stride >= n || !finite3(a[u], b[u], c[u])) continue;
      int ea = encode_ordered(a[u]), eb = encode_ordered(b[u]), ec = encode_ordered(c[u]);
      mn[0] = min(mn[0], ea); mx[0] = max(mx[0], ea);
      mn[1] = min(mn[1], eb); mx[1] = max(mx[1], eb);
      mn[2] = min(mn[2], ec); mx[2] = max(mx[2], ec);
      ++cnt;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], off));
      mx[a] = max(mx[a], __shfl_xor(mx[a], off));
    }
    cnt += __shfl_xor(cnt, off);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lds[wave][a] = mn[a]; lds[wave][3 + a] = mx[a]; }
    lds[wave][6] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 0; t < 7; ++t) {
      int v = lds[0][t];
      for (int w = 1; w < 4; ++w) {
        const int o = lds[w][t];
        v = t < 3 ? min(v, o) : (t < 6 ? max(v, o) : v + o);
      }
      st_agent(rows + blockIdx.x * 8 + t, v);
    }
    s_last = last_ticket(ticket, gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  // ---- the last block: fold the rows, derive the geometry ----
  int fm[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
  int fc = 0;
  for (int r = threadIdx.x; r < (int)gridDim.x; r += blockDim.x) {
    const int* p = rows + r * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      fm[a] = min(fm[a], ld_agent(p + a));
      fm[3 + a] = max(fm[3 + a], ld_agent(p + 3 + a));
    }
    fc += ld_agent(p + 6);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      fm[a] = min(fm[a], __shfl_xor(fm[a], off));
      fm[3 + a] = max(fm[3 + a], __shfl_xor(fm[3 + a], off));
    }
    fc += __shfl_xor(fc, off);
  }
  __syncthreads();  // lds is reused
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 6; ++a) lds[wave][a] = fm[a];
    lds[wave][6] = fc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int out[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      int v = lds[0][a];
      for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, lds[w][a]) : max(v, lds[w][a]);
      out[a] = v;
    }
    const int total = lds[0][6] + lds[1][6] + lds[2][6] + lds[3][6];
    derive_geometry(out, total, leaf, inv_leaf, cell_capacity, planned_passes, gd);
    *gd_host = *gd;
    nleaf[0] = 0;
    nleaf[1] = 0;
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next build
  }
}

// ref: voxel_grid_covariance_impl.hpp:222-225 -- floor in f32, f32 subtraction
// of min_b, truncation; 1-D index with divb_mul = (1, dx, dx*dy).
__device__ __forceinline__ int cell_of(float px, float py, float pz, const GridGeom& g) {
  int i0 = (int)(floorf(px * g.inv_leaf) - (float)g.min_b[0]);
  int i1 = (int)(floorf(py * g.inv_leaf) - (float)g.min_b[1]);
  int i2 = (int)(floorf(pz * g.inv_leaf) - (float)g.min_b[2]);
  return i0 + i1 * g.mul1 + i2 * g.mul2;
}

// ---- stable LSD radix sort of (cell key, point index) ---------------------------------
// One tile = 2048 consecutive pairs per 256-thread block; wave w owns the contiguous
// quarter [w*512, (w+1)*512) of the tile and walks it in 8 rounds of 64, so "tile order"
// is (wave, round, lane).  Per digit pass: count (per-tile histogram, bin-major), scan (one
// block per bin over the tiles + bin totals), scatter (rank of a pair among the equal
// digits before it in the tile: equal-digit lanes of a round find each other with 8
// ballots, rounds chain through a per-wave LDS counter, waves through a 4-entry prefix).
// No atomics on global memory, no look-back chain between tiles: every launch is a
// streaming pass, and the result does not depend on scheduling.
constexpr int SORT_THREADS = 256;
constexpr int SORT_ROUNDS = 8;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_TILE = SORT_THREADS * SORT_ROUNDS;
constexpr int SORT_BINS = 256;
// (Round 2 tried to let scatter pass p count the tile histogram of pass p + 1 with integer atomics
// on a global table -- one launch per pass less.  A million device-scope atomicAdds spread over
// 125 k counters took 88-248 us per pass: across eight XCDs they are served at the memory side.
// Reverted; the per-tile LDS histogram of k_sort_count costs 6 us.)

__device__ __forceinline__ int sort_index(int tile, int wave, int round, int lane) {
  return tile * SORT_TILE + wave * (SORT_TILE / SORT_WAVES) + round * 64 + lane;
}


// Cell key per point (+ the packed float4 copy the per-voxel gather reads) and, in the same
// pass, the tile histograms of the first sort digit.
__global__ void __launch_bounds__(SORT_THREADS) k_cell_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int n,
                                                           const BuildGeom* __restrict__ gd,
                                                           uint32_t* __restrict__ keys, float4* __restrict__ xyz4,
                                                           int ntiles, int* __restrict__ hist) {
  __shared__ int h[SORT_BINS];
  if (gd->status != BG_OK) return;  // uniform: nothing below runs on a refused geometry
  const GridGeom g = gd->g;
  const uint32_t digit_mask = (1u << gd->width[0]) - 1u;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a[SORT_ROUNDS], b[SORT_ROUNDS], c[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {  // all loads of the tile in flight before any use
    const int i = sort_index(blockIdx.x, wave, r, lane);
    const int j = i < n ? i : 0;
    a[r] = x[j]; b[r] = y[j]; c[r] = z[j];
  }
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int i = sort_index(blockIdx.x, wave, r, lane);
    if (i >= n) break;
    // packed copy for the per-voxel gather: one 16-byte line per point instead of three
    xyz4[i] = make_float4(a[r], b[r], c[r], 0.0f);
    uint32_t key = (uint32_t)g.ncells;  // sentinel sorts behind every real cell
    if (finite3(a[r], b[r], c[r])) {
      int idx = cell_of(a[r], b[r], c[r], g);
      if (idx >= 0 && idx < g.ncells) key = (uint32_t)idx;
    }
    keys[i] = key;
    atomicAdd(&h[key & digit_mask], 1);
  }
  __syncthreads();
  hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(SORT_THREADS) k_sort_count(const uint32_t* __restrict__ keys, int n, int pass,
                                                            const BuildGeom* __restrict__ gd, int ntiles,
                                                            int* __restrict__ hist) {
  __shared__ int h[SORT_BINS];
  if (gd->status != BG_OK) return;
  const int shift = gd->shift[pass];
  const uint32_t digit_mask = (1u << gd->width[pass]) - 1u;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t key[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int i = sort_index(blockIdx.x, wave, r, lane);
    key[r] = keys[i < n ? i : 0];
  }
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r)
    if (sort_index(blockIdx.x, wave, r, lane) < n) atomicAdd(&h[(key[r] >> shift) & digit_mask], 1);
  __syncthreads();
  hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off);
    if (lane >= off) v += t;
  }
  return v;
}

// one block per bin: exclusive scan of that bin's counts over the tiles (in place) + total
__global__ void __launch_bounds__(SORT_THREADS) k_sort_scan(int* __restrict__ hist, int ntiles, int* __restrict__ totals) {
  __shared__ int wsum[SORT_WAVES];
  __shared__ int carry;
  int* row = hist + (size_t)blockIdx.x * ntiles;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < ntiles; base += SORT_THREADS) {
    const int i = base + threadIdx.x;
    const int v = i < ntiles ? row[i] : 0;
    const int incl = wave_inclusive_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < ntiles) row[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == SORT_THREADS - 1) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// FIRST: the values are the identity permutation and are synthesised instead of read
template <bool FIRST>
__global__ void __launch_bounds__(SORT_THREADS) k_sort_scatter(const uint32_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in, int n, int pass,
                                                              const BuildGeom* __restrict__ gd, int ntiles,
                                                              const int* __restrict__ hist,
                                                              const int* __restrict__ totals,
                                                              uint32_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out) {
  __shared__ int cnt[SORT_WAVES][SORT_BINS];
  __shared__ int gbase[SORT_BINS];
  __shared__ int wsum[SORT_WAVES];
  if (gd->status != BG_OK) return;
  const int shift = gd->shift[pass];
  const uint32_t digit_mask = (1u << gd->width[pass]) - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {  // first output slot of every digit for this tile: bins before it + same bin in earlier tiles
    const int t = totals[threadIdx.x];
    const int incl = wave_inclusive_scan(t, lane);
    if (lane == 63) wsum[wave] = incl;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) cnt[w][threadIdx.x] = 0;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    gbase[threadIdx.x] = before + incl - t + hist[threadIdx.x * ntiles + blockIdx.x];
  }
  uint32_t key[SORT_ROUNDS], val[SORT_ROUNDS];
  int rank[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int i = sort_index(blockIdx.x, wave, r, lane);
    key[r] = i < n ? keys_in[i] : 0u;
    val[r] = FIRST ? (uint32_t)i : (i < n ? vals_in[i] : 0u);
  }
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const bool valid = sort_index(blockIdx.x, wave, r, lane) < n;
    const uint32_t d = (key[r] >> shift) & digit_mask;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int before = cnt[wave][d];
    const int lower = __popcll(same & lt_mask);
    rank[r] = before + lower;
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == 0) cnt[wave][d] = before + __popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  {
    int run = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const int t = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += t;
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    if (sort_index(blockIdx.x, wave, r, lane) >= n) break;
    const uint32_t d = (key[r] >> shift) & digit_mask;
    const int pos = gbase[d] + cnt[wave][d] + rank[r];
    keys_out[pos] = key[r];
    vals_out[pos] = val[r];
  }
}

// ---- one launch per digit ("fused pass") -------------------------------------------------
// count + scan + scatter of a digit in ONE launch (three launches at a 4-5 us floor each before):
// a block ranks its tile, publishes the tile's 256 digit counts as self-validating words
// {16-bit launch tag | 16-bit count} with agent-scope stores, and then reads the WHOLE
// tiles x 256 table back with agent-scope loads, re-reading any word whose tag is not this
// launch's yet: the column sum over all tiles gives the digit totals, the part of it over the
// tiles before its own the digit's offset.  No scan kernel, no look-back chain (every block does
// the same fixed amount of reading), no atomics, and the outcome does not depend on arrival
// order.  The table read grows with tiles^2, so the tile is 8192 pairs (1024 threads, wave w owns
// the contiguous 512 pairs [w*512, (w+1)*512) in 8 rounds of 64, so tile order is (wave, round,
// lane) as in the classic pass): 123 tiles x 1 KB for 1M points.  Every block waits for every
// other one, so all of them must be resident at once: the host uses this path for at most
// FUSED_MAX_TILES tiles (one block per CU) and the classic three-launch pass beyond.  A block that
// has waited FUSED_TIMEOUT_TICKS (other work holding the CUs) gives up: it marks the build
// BG_SPIN in *gd and *gd_host, every later build kernel returns at once, and the host repeats the
// build with classic passes.
constexpr int FUSED_THREADS = 1024;
constexpr int FUSED_WAVES = FUSED_THREADS / 64;
constexpr int FUSED_TILE = FUSED_THREADS * SORT_ROUNDS;          // 8192 pairs: clouds up to 256 tiles = 2 M points
constexpr int FUSED_BIG_ROUNDS = 2 * SORT_ROUNDS;
constexpr int FUSED_BIG_TILE = FUSED_THREADS * FUSED_BIG_ROUNDS;  // 16384 pairs (128 KB of LDS staging): up to 4 M points
constexpr int FUSED_MAX_TILES = 256;
constexpr int FUSED_COLS = FUSED_THREADS / 64;  // tile rows read per trip (one per wave)
constexpr unsigned long long FUSED_TIMEOUT_TICKS = 5000000ull;  // 50 ms of the 100 MHz clock

template <int ROUNDS>
__device__ __forceinline__ int fused_index(int tile, int wave, int round, int lane) {
  return tile * (FUSED_THREADS * ROUNDS) + wave * (ROUNDS * 64) + round * 64 + lane;
}

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// FROM_POINTS: pass 0 straight from the cloud -- the cell key is computed here (k_cell_keys is
// not launched), the packed float4 copy is written, and the values are the identity permutation.
template <bool FROM_POINTS, int ROUNDS>
__global__ void __launch_bounds__(FUSED_THREADS) k_sort_pass(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ z, float4* __restrict__ xyz4,
                                                            const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in, int n, int pass,
                                                            BuildGeom* __restrict__ gd, BuildGeom* __restrict__ gd_host,
                                                            int ntiles, uint32_t* __restrict__ table, uint32_t tag,
                                                            int mute_tile /* test seam: this tile never publishes */,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out) {
  constexpr int TILE = FUSED_THREADS * ROUNDS;
  __shared__ int cnt[FUSED_WAVES][SORT_BINS];
  // the column partial sums, and later -- once they have been folded -- the tile in sorted order
  __shared__ uint32_t scratch[2 * TILE];
  int (*part_total)[SORT_BINS] = reinterpret_cast<int (*)[SORT_BINS]>(scratch);
  int (*part_before)[SORT_BINS] = reinterpret_cast<int (*)[SORT_BINS]>(scratch + FUSED_COLS * SORT_BINS);
  uint32_t* stage_k = scratch;
  uint32_t* stage_v = scratch + TILE;
  __shared__ int gbase[SORT_BINS];
  __shared__ int lbase[SORT_BINS];
  __shared__ int wsum[2 * SORT_BINS / 64];
  __shared__ int s_fail;
  if (gd->status != BG_OK) return;  // uniform over the grid: no block waits for one that left here
  const int shift = gd->shift[pass];
  const uint32_t digit_mask = (1u << gd->width[pass]) - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = blockIdx.x;
  NDT_BSTAMP(pass, 0);
  // the 16-round tile keeps 48 registers of (key, value, rank) per thread: its values are fetched only
  // after the ranking (the loads land while the block waits for the table)
  constexpr bool LATE_VALS = ROUNDS > 8;
  uint32_t key[ROUNDS], val[ROUNDS];
  int rank[ROUNDS];
  if (FROM_POINTS) {
    const GridGeom g = gd->g;
    float a[ROUNDS], b[ROUNDS], c[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {  // all loads of the tile in flight before any use
      const int i = fused_index<ROUNDS>(tile, wave, r, lane);
      const int j = i < n ? i : 0;
      a[r] = x[j]; b[r] = y[j]; c[r] = z[j];
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int i = fused_index<ROUNDS>(tile, wave, r, lane);
      key[r] = (uint32_t)g.ncells;  // sentinel sorts behind every real cell
      if (!LATE_VALS) val[r] = (uint32_t)i;
      if (i < n) {
        xyz4[i] = make_float4(a[r], b[r], c[r], 0.0f);  // one 16-byte line per point for the per-voxel gather
        if (finite3(a[r], b[r], c[r])) {
          const int idx = cell_of(a[r], b[r], c[r], g);
          if (idx >= 0 && idx < g.ncells) key[r] = (uint32_t)idx;
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int i = fused_index<ROUNDS>(tile, wave, r, lane);
      key[r] = i < n ? keys_in[i] : 0u;
      if (!LATE_VALS) val[r] = i < n ? vals_in[i] : 0u;
    }
  }
  for (int d = threadIdx.x; d < FUSED_WAVES * SORT_BINS; d += FUSED_THREADS) (&cnt[0][0])[d] = 0;
  if (threadIdx.x == 0) s_fail = 0;
  __syncthreads();
  NDT_BSTAMP(pass, 1);  // keys in registers (thread 0's at least: its loads were consumed)
  // (Publishing the tile counts BEFORE the ranking, from an LDS-atomic histogram, was tried so that
  // the table would be complete by the time the ranks are: the atomics of a wave whose 64 pairs
  // share a digit serialise -- +0.9 / +3.8 / +6.2 us on the three passes of C3, whose top digit is
  // nearly constant -- against 1.2 us saved.  profiles/r02_build_stamps_early_publish.txt)
  // rank among the equal digits before it in the wave's 512 pairs (8 ballots per round)
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const bool valid = fused_index<ROUNDS>(tile, wave, r, lane) < n;
    const uint32_t d = (key[r] >> shift) & digit_mask;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int before = cnt[wave][d];
    const int lower = __popcll(same & lt_mask);
    rank[r] = before + lower;
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == 0) cnt[wave][d] = before + __popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  if (LATE_VALS) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const int i = fused_index<ROUNDS>(tile, wave, r, lane);
      val[r] = FROM_POINTS ? (uint32_t)i : (i < n ? vals_in[i] : 0u);
    }
  }
  __syncthreads();
  NDT_BSTAMP(pass, 2);  // ranked
  // per-digit: waves -> exclusive prefix inside the tile, tile count published
  int my_count = 0;  // digit threadIdx.x in this tile
  if (threadIdx.x < SORT_BINS) {
    int run = 0;
#pragma unroll
    for (int w = 0; w < FUSED_WAVES; ++w) {
      const int t = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += t;
    }
    my_count = run;
    if (tile != mute_tile)
      __hip_atomic_store(table + (size_t)tile * SORT_BINS + threadIdx.x, (tag << 16) | (uint32_t)run, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  // the whole table, 16 bytes (four digits) per lane and one tile row per wave and trip
  {
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(table, 0, 0xFFFFFFFFu, 0x00020000);
    int tot[4] = {0, 0, 0, 0}, bef[4] = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    constexpr int BATCH = ROUNDS <= 8 ? 8 : 2;  // (the 16-round variant holds 48 registers of pairs: fewer table words in flight)
    for (int r0 = wave; r0 < ntiles; r0 += BATCH * FUSED_COLS) {
      u32x4_t w[BATCH];
      for (;;) {
        asm volatile("" ::: "memory");  // the loads below must be re-issued on every trip
        bool ok = true;
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int row = r0 + k * FUSED_COLS;
          if (row < ntiles)
            w[k] = __builtin_amdgcn_raw_buffer_load_b128(rt, ((unsigned int)row * SORT_BINS + 4u * lane) * 4u, 0, 16 /* sc1 */);
          else
            w[k].x = w[k].y = w[k].z = w[k].w = tag << 16;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k)
          ok = ok && (w[k].x >> 16) == tag && (w[k].y >> 16) == tag && (w[k].z >> 16) == tag && (w[k].w >> 16) == tag;
        if (ok) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > FUSED_TIMEOUT_TICKS) { s_fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const int row = r0 + k * FUSED_COLS;
        const int c0 = w[k].x & 0xffff, c1 = w[k].y & 0xffff, c2 = w[k].z & 0xffff, c3 = w[k].w & 0xffff;
        tot[0] += c0; tot[1] += c1; tot[2] += c2; tot[3] += c3;
        if (row < tile) { bef[0] += c0; bef[1] += c1; bef[2] += c2; bef[3] += c3; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { part_total[wave][4 * lane + q] = tot[q]; part_before[wave][4 * lane + q] = bef[q]; }
  }
  __syncthreads();
  NDT_BSTAMP(pass, 3);  // table read
  if (s_fail) {
    if (threadIdx.x == 0) { gd->status = BG_SPIN; gd_host->status = BG_SPIN; }
    return;
  }
  int g_excl = 0, l_excl = 0;  // digit threadIdx.x: first output slot over all tiles / inside this tile
  if (threadIdx.x < SORT_BINS) {
    int t = 0, b = 0;
#pragma unroll
    for (int c = 0; c < FUSED_COLS; ++c) { t += part_total[c][threadIdx.x]; b += part_before[c][threadIdx.x]; }
    const int incl = wave_inclusive_scan(t, lane);
    const int lincl = wave_inclusive_scan(my_count, lane);
    if (lane == 63) { wsum[wave] = incl; wsum[4 + wave] = lincl; }
    g_excl = incl - t + b;
    l_excl = lincl - my_count;
  }
  __syncthreads();  // part_total / part_before are dead from here on: `scratch` becomes the staging tile
  if (threadIdx.x < SORT_BINS) {
    int before = 0, lbefore = 0;
    for (int w = 0; w < wave; ++w) { before += wsum[w]; lbefore += wsum[4 + w]; }
    gbase[threadIdx.x] = before + g_excl;
    lbase[threadIdx.x] = lbefore + l_excl;
  }
  __syncthreads();
  // the tile in sorted order through LDS, so that the global stores of a wave fall into a few
  // contiguous runs instead of 64 separate places (a scattered wave store occupies the address
  // path for 64 cycles; 256 of them per CU were 7 us of a 16 us launch)
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (fused_index<ROUNDS>(tile, wave, r, lane) >= n) break;
    const uint32_t d = (key[r] >> shift) & digit_mask;
    const int p = lbase[d] + cnt[wave][d] + rank[r];
    stage_k[p] = key[r];
    stage_v[p] = val[r];
  }
  __syncthreads();
  NDT_BSTAMP(pass, 4);  // staged
  const int tile_n = min(TILE, n - tile * TILE);
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int j = r * FUSED_THREADS + (int)threadIdx.x;
    if (j >= tile_n) break;
    const uint32_t k = stage_k[j];
    const uint32_t d = (k >> shift) & digit_mask;
    const int pos = gbase[d] + (j - lbase[d]);
    keys_out[pos] = k;
    vals_out[pos] = stage_v[j];
  }
#ifdef NDT_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  NDT_BSTAMP(pass, 5);  // thread 0's stores acknowledged
#endif
}

// Runs of equal cell key in the sorted array.  A thread owns RUN_KEYS consecutive keys (16-byte
// loads), a block one tile of 256 * RUN_KEYS keys.  Every run TAIL needs its head: the thread's
// own last head, else the last head of a lower lane (wave max-scan), else of a lower wave
// (LDS), and only for a run that began before the tile a backward gallop + bisection in
// global memory.  Runs with at least min_pts points become leaves
// (ref: voxel_grid_covariance_impl.hpp:270-273).  Leaf slots are handed out by a
// count / scan / emit triple instead of a global atomic counter (one contended address
// served ~90 adds/us and cost 0.1 ms): slots come out in ascending cell order, identically
// on every run.

// KEYS consecutive keys per thread (8 or 16): the count launch pays one ticket per block, and a
// contended ticket is served at ~12 ns, so fewer, fatter blocks are cheaper (489 -> 245 for 1M).
// MODE 0 / 1: the count launch and the emit launch of the classic pair.  MODE 2: both in ONE
// launch -- a block publishes its leaf count as a self-validating word {16-bit launch tag | count}
// and adds up the words of the blocks BEFORE it, re-reading any that is not tagged yet.  A block
// only ever waits for lower-numbered blocks, which were dispatched before it, so this needs no
// co-residency; the time-out (BG_SPIN, as in k_sort_pass) is a second line of defence.
constexpr int RUNS_COUNT = 0, RUNS_EMIT = 1, RUNS_FUSED = 2;
template <int MODE, int RUN_KEYS>
__global__ void __launch_bounds__(256) k_runs(const uint32_t* __restrict__ keys, int n,
                                             BuildGeom* __restrict__ gd, BuildGeom* __restrict__ gd_host, int min_pts,
                                             int* __restrict__ block_counts, int* __restrict__ block_offsets,
                                             unsigned int* __restrict__ ticket, uint32_t* __restrict__ run_tags,
                                             uint32_t tag, int* __restrict__ nleaf_out,
                                             int* __restrict__ leaf_start, int* __restrict__ leaf_cnt) {
  constexpr bool EMIT = MODE != RUNS_COUNT;
  __shared__ int wave_head[4];
  __shared__ int wave_total[4];
  __shared__ int s_last;
  __shared__ int s_head0;
  constexpr int RUN_TILE = 256 * RUN_KEYS;
  if (gd->status != BG_OK) return;
  NDT_BSTAMP(3, 0);
  const int ncells = gd->g.ncells;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * RUN_TILE + threadIdx.x * RUN_KEYS;
  const uint32_t sentinel = 0xFFFFFFFFu;
  uint32_t k[RUN_KEYS];
  if (s0 + RUN_KEYS <= n) {  // keys is 16-byte aligned and s0 a multiple of 8
#pragma unroll
    for (int q = 0; q < RUN_KEYS / 4; ++q) {
      const uint4 a = *reinterpret_cast<const uint4*>(keys + s0 + 4 * q);
      k[4 * q + 0] = a.x; k[4 * q + 1] = a.y; k[4 * q + 2] = a.z; k[4 * q + 3] = a.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < RUN_KEYS; ++j) k[j] = s0 + j < n ? keys[s0 + j] : sentinel;
  }
  // At most one run of the tile began before it: the one that holds the tile's first key.  Wave 0
  // finds its head with two 64-wide probes per 4096 keys of run length (a gallop + bisection by the
  // one lane that owns the tail was ~20 dependent loads for the map's most crowded voxel, and that
  // straggler set the launch's duration: 10.6 us against a median block's 4.3).
  if (wave == 0) {
    const int b = blockIdx.x * RUN_TILE;
    int head0 = b;
    if (b > 0 && b < n) {
      const uint32_t K0 = keys[b];
      if (K0 < (uint32_t)ncells && keys[b - 1] == K0) {
        int hi = b;  // holds K0
        for (;;) {
          const long long jc = (long long)hi - (long long)(lane + 1) * 64;
          const bool differs = jc < 0 || keys[jc] != K0;
          const unsigned long long m = __ballot(differs);
          if (m == 0ull) { hi -= 64 * 64; continue; }  // all 64 probes still in the run: further back
          const int l = __ffsll((long long)m) - 1;      // nearest probe outside the run
          const int lo_excl = hi - (l + 1) * 64;        // outside (or < 0); lo_excl + 64 is inside
          const int jf = lo_excl + 1 + lane;
          const bool inside = jf >= 0 && keys[jf] == K0;
          const unsigned long long m2 = __ballot(inside);
          head0 = lo_excl + 1 + (__ffsll((long long)m2) - 1);
          break;
        }
      }
    }
    if (lane == 0) s_head0 = head0;
  }
  uint32_t prev = __shfl_up(k[RUN_KEYS - 1], 1), next = __shfl_down(k[0], 1);
  if (lane == 0) prev = s0 > 0 && s0 - 1 < n ? keys[s0 - 1] : sentinel;
  if (lane == 63) next = s0 + RUN_KEYS < n ? keys[s0 + RUN_KEYS] : sentinel;

  // last head position inside this thread (-1: none), and the same for everything before it
  int own_head = -1;
#pragma unroll
  for (int j = 0; j < RUN_KEYS; ++j) {
    const uint32_t before = j == 0 ? prev : k[j - 1];
    const bool valid = s0 + j < n && k[j] < (uint32_t)ncells;
    if (valid && (s0 + j == 0 || before != k[j])) own_head = s0 + j;
  }
  int incl = own_head;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl = max(incl, t);
  }
  int head_before = __shfl_up(incl, 1);
  if (lane == 0) head_before = -1;
  if (lane == 63) wave_head[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; ++w) head_before = max(head_before, wave_head[w]);

  // walk the keys again: count (and emit) the leaf tails
  int cur_head = head_before;
  int nleaf = 0;
  int starts[RUN_KEYS], counts[RUN_KEYS];
#pragma unroll
  for (int j = 0; j < RUN_KEYS; ++j) {
    const int s = s0 + j;
    const uint32_t before = j == 0 ? prev : k[j - 1];
    const uint32_t after = j == RUN_KEYS - 1 ? next : k[j + 1];
    const bool valid = s < n && k[j] < (uint32_t)ncells;
    if (valid && (s == 0 || before != k[j])) cur_head = s;
    const bool tail = valid && (s == n - 1 || after != k[j]);
    counts[j] = 0;
    starts[j] = 0;
    if (tail) {
      int start = cur_head;
      if (start < 0) start = s_head0;  // the run began before this tile: it is the tile's carry-in run
      const int cnt = s - start + 1;
      if (cnt >= min_pts) {
        starts[j] = start;
        counts[j] = cnt;
        ++nleaf;
      }
    }
  }
  // slots in position order: lower lanes, lower waves, lower blocks first
  int lincl = nleaf;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(lincl, off);
    if (lane >= off) lincl += t;
  }
  if (lane == 63) wave_total[wave] = lincl;
  __syncthreads();
  if (!EMIT) {
    // count pass: publish the block's leaf count; the block that draws the last ticket turns the
    // counts into exclusive offsets (a separate one-block scan kernel cost a launch: 4.8 us for
    // 489 values) and writes the total
    if (threadIdx.x == 0) {
      st_agent(block_counts + blockIdx.x, wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3]);
      s_last = last_ticket(ticket, gridDim.x) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __shared__ int wsum[4];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int nblocks = (int)gridDim.x;
    for (int base = 0; base < nblocks; base += 256) {
      const int i = base + (int)threadIdx.x;
      const int v = i < nblocks ? ld_agent(block_counts + i) : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int before = carry;
      for (int w = 0; w < wave; ++w) before += wsum[w];
      if (i < nblocks) block_offsets[i] = before + incl - v;
      __syncthreads();
      if (threadIdx.x == 255) carry = before + incl;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      nleaf_out[0] = carry;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  int block_off;
  NDT_BSTAMP(3, 1);  // counted
  if (MODE == RUNS_FUSED) {
    __shared__ int red[4];
    __shared__ int s_fail;
    const int mine = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    if (threadIdx.x == 0) {
      s_fail = 0;
      __hip_atomic_store(run_tags + blockIdx.x, (tag << 16) | (uint32_t)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    constexpr int POLL_BATCH = 8;  // words of a thread in flight together (one memory round trip for 2048 blocks)
    for (int i0 = threadIdx.x; i0 < (int)blockIdx.x; i0 += 256 * POLL_BATCH) {
      uint32_t w[POLL_BATCH];
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int u = 0; u < POLL_BATCH; ++u) {
          const int i = i0 + 256 * u;
          w[u] = i < (int)blockIdx.x ? __hip_atomic_load(run_tags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag << 16);
        }
#pragma unroll
        for (int u = 0; u < POLL_BATCH; ++u) ok = ok && (w[u] >> 16) == tag;
        if (ok) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > FUSED_TIMEOUT_TICKS) {
          s_fail = 1;
#pragma unroll
          for (int u = 0; u < POLL_BATCH; ++u) w[u] = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int u = 0; u < POLL_BATCH; ++u) acc += (int)(w[u] & 0xffffu);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (s_fail) {
      if (threadIdx.x == 0) { gd->status = BG_SPIN; gd_host->status = BG_SPIN; }
      return;
    }
    block_off = red[0] + red[1] + red[2] + red[3];
    NDT_BSTAMP(3, 2);  // offset known
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) nleaf_out[0] = block_off + mine;
  } else {
    block_off = block_offsets[blockIdx.x];
  }
  if (nleaf == 0) return;
  int slot = block_off + lincl - nleaf;
  for (int w = 0; w < wave; ++w) slot += wave_total[w];
#pragma unroll
  for (int j = 0; j < RUN_KEYS; ++j) {
    if (counts[j] > 0) {
      leaf_start[slot] = starts[j];
      leaf_cnt[slot] = counts[j];
      ++slot;
    }
  }
#ifdef NDT_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  NDT_BSTAMP(3, 3);
#endif
}

// one Jacobi rotation of the symmetric 3x3 A (full storage) in the (P,Q) plane
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double A[9], double V[9]) {
  double apq = A[3 * P + Q];
  if (apq == 0.0) return;
  double theta = (A[3 * Q + Q] - A[3 * P + P]) / (2.0 * apq);
  double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double akp = A[3 * k + P], akq = A[3 * k + Q];
    A[3 * k + P] = c * akp - s * akq;
    A[3 * k + Q] = s * akp + c * akq;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double apk = A[3 * P + k], aqk = A[3 * Q + k];
    A[3 * P + k] = c * apk - s * aqk;
    A[3 * Q + k] = s * apk + c * aqk;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    double vkp = V[3 * k + P], vkq = V[3 * k + Q];
    V[3 * k + P] = c * vkp - s * vkq;
    V[3 * k + Q] = s * vkp + c * vkq;
  }
}

#define NDT_SWAP_COL(a, b)                                          \
  {                                                                 \
    double td = d[a]; d[a] = d[b]; d[b] = td;                       \
    _Pragma("unroll") for (int k = 0; k < 3; ++k) {                 \
      double tv = V[3 * k + a]; V[3 * k + a] = V[3 * k + b]; V[3 * k + b] = tv; \
    }                                                               \
  }

constexpr int LANES_PER_LEAF = 8;
constexpr int SUMS_BLOCKS_MAX = 2048;

// ref: voxel_grid_covariance_impl.hpp:236-239 -- per-voxel sum(x) and sum(x x^T) in f64.
// A wave owns 8 leaves.  The first LEAF_HEAD points of a leaf are gathered by its own 8 lanes
// (stable sort => ascending input order) and added by a fixed 3-step xor tree; whatever a
// crowded leaf holds beyond that is gathered by all 64 lanes of the wave, leaf after leaf, and
// added by a fixed 6-step tree -- otherwise the most crowded voxel of the map (1000+ points on 8
// lanes) sets the kernel's duration.  Every association is fixed: the sums are reproducible.
#ifndef NDT_LEAF_HEAD
#define NDT_LEAF_HEAD 64   // (128 measured in round 3: no difference, profiles/r03_build_stamps_waves.txt)
#endif
constexpr int LEAF_HEAD = NDT_LEAF_HEAD;
#ifndef NDT_SUMS_UNROLL
#define NDT_SUMS_UNROLL 4
#endif
constexpr int SUMS_UNROLL = NDT_SUMS_UNROLL;  // dependent (index -> point) gathers in flight per lane

struct Moments {
  double s[3], ss[6];
};

__device__ __forceinline__ void moments_add(Moments& m, const float4& p, bool live) {
  const double a = live ? (double)p.x : 0.0, b = live ? (double)p.y : 0.0, c = live ? (double)p.z : 0.0;
  m.s[0] += a; m.s[1] += b; m.s[2] += c;
  m.ss[0] += a * a; m.ss[1] += a * b; m.ss[2] += a * c;
  m.ss[3] += b * b; m.ss[4] += b * c; m.ss[5] += c * c;
}

template <int WIDTH>
__device__ __forceinline__ void moments_xor_tree(Moments& m) {
#pragma unroll
  for (int off = 1; off < WIDTH; off <<= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) m.s[a] += __shfl_xor(m.s[a], off);
#pragma unroll
    for (int a = 0; a < 6; ++a) m.ss[a] += __shfl_xor(m.ss[a], off);
  }
}

__device__ __forceinline__ bool finalize_one(int slot, const uint32_t* __restrict__ keys,
                                             const int* __restrict__ leaf_start, const int* __restrict__ leaf_cnt,
                                             const double* __restrict__ sums, FinalizeParams fp,
                                             VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                             LeafStats* __restrict__ stats, int* __restrict__ cell2leaf);
__device__ __forceinline__ bool finalize_leaf(int slot, int cell, int cnt, const double* __restrict__ in, FinalizeParams fp,
                                              VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                              LeafStats* __restrict__ stats, int* __restrict__ cell2leaf);

// ref: voxel_grid_covariance_impl.hpp:265-343 -- one thread per leaf: mean, covariance,
// eigen-decomposition, eigenvalue inflation, inverse, validity checks.  (Fusing this into the
// 8 lanes that sum a leaf was tried in round 2: 48 us against 20 + 12 -- the Jacobi state on top
// of the moments spills, and 8 of 64 lanes do distinct work.)
// The number of accepted leaves is summed without atomics: a count per block, and the block that
// draws the last ticket adds the counts and hands both leaf counters to the host through pinned
// memory (the D2H copy they used to take was a 4.4 us launch of its own).  Contended adds on one
// address are served at ~12 ns each, so the grid is as few blocks as the leaves need: with
// 64-thread blocks and one atomicAdd per wave this kernel spent 327 tickets + 327 adds on 21 k leaves.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_leaf_finalize(const uint32_t* __restrict__ keys,
                                                           int* __restrict__ nleaf_p,
                                                           const int* __restrict__ leaf_start,
                                                           const int* __restrict__ leaf_cnt,
                                                           const double* __restrict__ sums, FinalizeParams fp,
                                                           VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                                           LeafStats* __restrict__ stats,
                                                           int* __restrict__ cell2leaf, int* __restrict__ block_ok,
                                                           unsigned int* __restrict__ ticket,
                                                           int* __restrict__ nleaf_host, int done_tag) {
  constexpr int WAVES = THREADS / 64;
  __shared__ int s_ok[WAVES];
  __shared__ int s_last;
  const int slot = blockIdx.x * THREADS + threadIdx.x;
  const int nl = nleaf_p[0];
  // the grid is sized for the worst case (n / min_points leaves); only the blocks that hold a
  // leaf -- block 0 always -- take part in the ticket
  const int live_blocks = max(1, (nl + THREADS - 1) / THREADS);
  if ((int)blockIdx.x >= live_blocks) return;
  bool ok = false;
  if (slot < nl) ok = finalize_one(slot, keys, leaf_start, leaf_cnt, sums, fp, rec, cent, stats, cell2leaf);
  const int wave_ok = __popcll(__ballot(ok));
  if ((threadIdx.x & 63) == 0) s_ok[threadIdx.x >> 6] = wave_ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t += s_ok[w];
    st_agent(block_ok + blockIdx.x, t);
    s_last = last_ticket(ticket, (unsigned int)live_blocks) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  int sum = 0;
  for (int i = threadIdx.x; i < live_blocks; i += THREADS) sum += ld_agent(block_ok + i);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  if ((threadIdx.x & 63) == 0) s_ok[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) total += s_ok[w];
    nleaf_p[1] = total;  // leaves that passed every check
    // one 16-byte store {slots, accepted, build tag, 0}: the host may poll the tag instead of waiting
    // for the stream (nleaf_host is 16-byte aligned)
    *reinterpret_cast<int4*>(nleaf_host) = make_int4(nl, total, done_tag, 0);
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ bool finalize_one(int slot, const uint32_t* __restrict__ keys,
                                             const int* __restrict__ leaf_start, const int* __restrict__ leaf_cnt,
                                             const double* __restrict__ sums, FinalizeParams fp,
                                             VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                             LeafStats* __restrict__ stats, int* __restrict__ cell2leaf) {
  const int start = leaf_start[slot], cnt = leaf_cnt[slot];
  return finalize_leaf(slot, (int)keys[start], cnt, sums + (size_t)slot * 9, fp, rec, cent, stats, cell2leaf);
}

// one leaf: count, cell, and its nine moment sums -> statistics, record, dense index entry
__device__ __forceinline__ bool finalize_leaf(int slot, int cell, int cnt, const double* __restrict__ in, FinalizeParams fp,
                                              VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                              LeafStats* __restrict__ stats, int* __restrict__ cell2leaf) {
  const double s[3] = {in[0], in[1], in[2]};
  const double ss[6] = {in[3], in[4], in[5], in[6], in[7], in[8]};
  const double n = (double)cnt;
  double mean[3] = {s[0] / n, s[1] / n, s[2] / n};  // ref :278
  double C[9];
  const int tri[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
  if (fp.cov_mode == 0) {
    // ref :287-291
    const double k = n / (n - 1.0);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        C[3 * a + b] = ((ss[tri[3 * a + b]] / n) - (mean[a] * mean[b])) * k;
  } else {
    const double k = (n - 1.0) / n;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        C[3 * a + b] = ((ss[tri[3 * a + b]] - 2.0 * (s[a] * mean[b])) / n + mean[a] * mean[b]) * k;
  }

  LeafStats L;
  L.cell = cell;
  L.count = cnt;
#pragma unroll
  for (int a = 0; a < 3; ++a) L.mean[a] = mean[a];

  // eigen-decomposition (ref :298-300), cyclic Jacobi in f64
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
  for (int a = 0; a < 9; ++a) A[a] = C[a];
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    jacobi_rot<0, 1>(A, V);
    jacobi_rot<0, 2>(A, V);
    jacobi_rot<1, 2>(A, V);
  }
  double d[3] = {A[0], A[4], A[8]};
  if (d[0] > d[1]) NDT_SWAP_COL(0, 1);
  if (d[1] > d[2]) NDT_SWAP_COL(1, 2);
  if (d[0] > d[1]) NDT_SWAP_COL(0, 1);

  bool ok = !(d[0] < 0 || d[1] < 0 || d[2] < 1e-12);  // ref :303-309
  // ref :311-331
  const double floor_ev = fmax(1e-12, d[2] * fp.eig_ratio);
  bool recompose = false;
  if (d[0] < floor_ev) { d[0] = floor_ev; recompose = true; }
  if (d[1] < floor_ev) { d[1] = floor_ev; recompose = true; }
  if (recompose) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        double acc = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc += V[3 * a + k] * d[k] * V[3 * b + k];
        C[3 * a + b] = acc;
      }
  }
  // inverse by cofactors (ref :334)
  double c00 = C[4] * C[8] - C[5] * C[7];
  double c01 = C[5] * C[6] - C[3] * C[8];
  double c02 = C[3] * C[7] - C[4] * C[6];
  double det = C[0] * c00 + C[1] * c01 + C[2] * c02;
  double id = 1.0 / det;
  double I[9];
  I[0] = c00 * id;
  I[1] = (C[2] * C[7] - C[1] * C[8]) * id;
  I[2] = (C[1] * C[5] - C[2] * C[4]) * id;
  I[3] = c01 * id;
  I[4] = (C[0] * C[8] - C[2] * C[6]) * id;
  I[5] = (C[2] * C[3] - C[0] * C[5]) * id;
  I[6] = c02 * id;
  I[7] = (C[1] * C[6] - C[0] * C[7]) * id;
  I[8] = (C[0] * C[4] - C[1] * C[3]) * id;
  double amax = 0;
#pragma unroll
  for (int a = 0; a < 9; ++a) {
    if (!isfinite(I[a])) ok = false;
    amax = fmax(amax, fabs(I[a]));
  }
  if (amax > 1e12) ok = false;  // ref :337-343

#pragma unroll
  for (int a = 0; a < 9; ++a) { L.cov[a] = C[a]; L.icov[a] = I[a]; L.evecs[a] = V[a]; }
#pragma unroll
  for (int a = 0; a < 3; ++a) L.evals[a] = d[a];
  if (!ok) L.count = -cnt;
  stats[slot] = L;
  // Every slot gets a finite record: the derivative kernel reads record 0 for an absent
  // neighbour (masked by f = 0, but 0 * NaN would still poison the sums), and slot 0 may well be
  // a rejected leaf.  Only accepted leaves are reachable through the cell -> leaf grid.
  VoxelRecord r;
  r.mean[0] = ok ? mean[0] : 0.0; r.mean[1] = ok ? mean[1] : 0.0; r.mean[2] = ok ? mean[2] : 0.0;
  r.icov[0] = ok ? I[0] : 0.0; r.icov[1] = ok ? I[1] : 0.0; r.icov[2] = ok ? I[2] : 0.0;
  r.icov[3] = ok ? I[4] : 0.0; r.icov[4] = ok ? I[5] : 0.0; r.icov[5] = ok ? I[8] : 0.0;
  r.pad = (double)cnt;
  rec[slot] = r;
  // the centroid as the radius search sees it (f32-rounded leaf mean, ref: voxel_grid_covariance_impl.hpp:420-422),
  // 16 bytes per leaf: the KDTREE / multi-grid neighbourhoods test 27 cells per point against it; w = "no further leaf
  // in this cell" (the multi-grid union chains leaves through it)
  cent[slot] = make_float4((float)r.mean[0], (float)r.mean[1], (float)r.mean[2], __int_as_float(-1));
  if (ok) cell2leaf[cell] = slot;
  return ok;
}

__global__ void __launch_bounds__(256) k_leaf_sums(const float4* __restrict__ xyz4, const uint32_t* __restrict__ vals,
                                                  const int* __restrict__ nleaf_p,
                                                  const int* __restrict__ leaf_start,
                                                  const int* __restrict__ leaf_cnt, double* __restrict__ sums) {
  const int nleaf = nleaf_p[0];
  const int lane = threadIdx.x & 63;
  const int sub = lane & (LANES_PER_LEAF - 1);
  const int per_block = 256 / LANES_PER_LEAF;
  // Wave W takes the leaves W, W + rows, W + 2 rows, ... (rows = ceil(nleaf / 8)): crowded voxels
  // are neighbours in cell order (a wall, the road), and eight of them in one wave would
  // serialise.  The loop bound is wave-uniform: all 8 leaves of a wave step together.
  const int rows = (nleaf + 7) >> 3;
  for (int row = blockIdx.x * (per_block / 8) + (threadIdx.x >> 6); row < rows; row += gridDim.x * (per_block / 8)) {
    const int slot = row + (lane >> 3) * rows;
    const bool have = slot < nleaf;
    const int start = have ? leaf_start[slot] : 0, cnt = have ? leaf_cnt[slot] : 0;
    Moments m{};
    // four gathers in flight per lane (index load -> point load is a dependent pair; eight were
    // measured: 21.5 us against 20.3);
    // the adds stay in point order, masked lanes add exact zeros
    const int head = cnt < LEAF_HEAD ? cnt : LEAF_HEAD;
    for (int j0 = sub; j0 < head; j0 += SUMS_UNROLL * LANES_PER_LEAF) {
      float4 p[SUMS_UNROLL];
      bool live[SUMS_UNROLL];
#pragma unroll
      for (int u = 0; u < SUMS_UNROLL; ++u) {
        const int j = j0 + u * LANES_PER_LEAF;
        live[u] = j < head;
        p[u] = xyz4[vals[start + (live[u] ? j : 0)]];
      }
#pragma unroll
      for (int u = 0; u < SUMS_UNROLL; ++u) moments_add(m, p[u], live[u]);
    }
    moments_xor_tree<LANES_PER_LEAF>(m);
    // crowded leaves: the whole wave gathers the rest
    unsigned long long crowded = __ballot(cnt > LEAF_HEAD);
    while (crowded) {
      const int src = __ffsll((long long)crowded) - 1;  // first lane of that leaf's group
      crowded &= ~(0xFFull << (src & ~7));
      const int bstart = __shfl(start, src), bcnt = __shfl(cnt, src);
      Moments t{};
      for (int j0 = LEAF_HEAD + lane; j0 < bcnt; j0 += SUMS_UNROLL * 64) {
        float4 p[SUMS_UNROLL];
        bool live[SUMS_UNROLL];
#pragma unroll
        for (int u = 0; u < SUMS_UNROLL; ++u) {
          const int j = j0 + u * 64;
          live[u] = j < bcnt;
          p[u] = xyz4[vals[bstart + (live[u] ? j : LEAF_HEAD)]];
        }
#pragma unroll
        for (int u = 0; u < SUMS_UNROLL; ++u) moments_add(t, p[u], live[u]);
      }
      moments_xor_tree<64>(t);
      if ((lane >> 3) == (src >> 3)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) m.s[a] += t.s[a];
#pragma unroll
        for (int a = 0; a < 6; ++a) m.ss[a] += t.ss[a];
      }
    }
    if (!have) continue;
    // the 8 lanes hold identical sums; lane k writes word k, lane 0 also word 8
    double* o = sums + (size_t)slot * 9;
    const double mine = sub == 0 ? m.s[0] : sub == 1 ? m.s[1] : sub == 2 ? m.s[2] : sub == 3 ? m.ss[0]
                      : sub == 4 ? m.ss[1] : sub == 5 ? m.ss[2] : sub == 6 ? m.ss[3] : m.ss[4];
    o[sub] = mine;
    if (sub == 0) o[8] = m.ss[5];
  }
}

// ---- bucketed build: TWO launches for the whole voxel grid (round 3) ----------------------------------
// The sort-based pipeline above moves (key, index) pairs through HBM three times and then gathers the
// points at random (2.5x the algorithmic traffic, eight dependent launches, profiles/r02_pmc_summary.txt).
// Grouping points by voxel needs no global ORDER of the voxels, only (a) all points of a voxel in one
// place and (b) their input order kept (the f64 sums must be bit-reproducible: every rank of a multi-GPU
// job builds its own table and all must agree).  So:
//   k_bucket_pass    one pass over the cloud: bounds rows (the geometry is derived later), and a STABLE
//                    partition of the points themselves into 256 buckets by a hash of the point's absolute
//                    voxel coordinates (floor(p / leaf): no grid geometry needed, so this launch does not
//                    wait for the bounds) -- the ranking, the self-validating tile-count table and the
//                    LDS-staged scatter of k_sort_pass, carrying 16-byte points instead of (key, index).
//                    A hash of single voxels balances the buckets whatever the scene looks like (C3: 178
//                    occupied voxels per bucket, largest bucket 1.6x the mean).
//   k_bucket_leaves  one 1024-thread block per bucket, everything in LDS: derive the geometry (every block
//                    folds the bounds rows: same f32 arithmetic, same result), cell keys, stable LSD radix
//                    sort of (key, local index) in LDS, run search, per-voxel sums gathered from LDS in
//                    input order (no random HBM gathers), Jacobi / inflation / inverse, records + dense
//                    index.  Leaf slots are handed out by one atomic add per block: slot ORDER differs
//                    from run to run, nothing observable depends on it (exports sort by cell; a point's
//                    pairs are added in neighbour order, not slot order).
// Traffic: 12 MB read + 16 MB written, then 16 MB read + the leaves (C3) -- below SURVEY 8d's algorithmic
// 61 MB.  Steady state only (every buffer exists, the dense grid is clean); clouds whose largest bucket
// would not fit a block's LDS (BK_MAXP points: more than ~1.3 M points, or one voxel holding thousands),
// coordinates beyond 2^23 voxels, devices with fewer CUs than tiles, and first builds take the sort-based
// pipeline, which stays in the file as the fallback (status BG_BUCKET).
constexpr int BK_BUCKETS = 256;
constexpr int BK_THREADS = 1024;
constexpr int BK_WAVES = BK_THREADS / 64;
constexpr int BK_ROUNDS = 8;
constexpr int BK_MAXP = BK_THREADS * BK_ROUNDS;   // 8192 points per bucket block
constexpr int BK_MAX_LEAVES = BK_MAXP / 3;        // min_points_per_voxel >= 3
constexpr float BK_COORD_LIMIT = 8388608.0f;      // 2^23 voxels: below it floor(p / leaf) - min_b is exact in f32

// points of one voxel have identical floor(p * inv_leaf) triples, hence one bucket
__device__ __forceinline__ uint32_t bucket_of(float fx, float fy, float fz) {
  const uint32_t i = (uint32_t)__float2int_rz(fx), j = (uint32_t)__float2int_rz(fy), k = (uint32_t)__float2int_rz(fz);
  uint32_t h = (i * 0x9E3779B1u) ^ (j * 0x85EBCA77u) ^ (k * 0xC2B2AE3Du);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h >> 24;
}

// ROUNDS: points per thread = tile size / 1024 (1, 2, 4, 8: bucket_rounds_for(n) keeps the launch at <= BK_MAX_TILES tiles
// and, for the clouds the drivers hand over, at one tile per compute unit or more).
//
// Round 5: the partition is PER TILE.  A tile goes out in bucket order into its own window of pts_out
// ([tile * TILE, tile * TILE + tile_n): one contiguous, coalesced store of the staged tile) and says where each bucket's
// points lie in the column table tab[bucket][tile] = {count : 16 | first : 16}; the bucket's block of the next launch
// gathers its ~ntiles segments in tile order -- the same points in the same (input) order as the cloud-wide partition
// of rounds 3-4, so the leaves are bit-identical.  What went away: the tagged tile-count table every tile published and
// then polled for all the others (5.5 us of the launch's 23: write -> visible -> read of 123 KB by every block, and
// the launch needed every tile resident at once, BG_SPIN when it was not), the cloud-wide bucket offsets and the
// scattered 16-byte stores.  The launch waits for nothing now: any grid size, any residency.
constexpr int BK_MAX_TILES = 256;   // = threads that scan a bucket's column in k_bucket_leaves
template <int ROUNDS>
__global__ void __launch_bounds__(BK_THREADS) k_bucket_pass(const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ z, int n, float inv_leaf,
                                                           int tile0 /* first tile of this launch (the host hand-off launches the pass chunk by chunk) */,
                                                           uint32_t* __restrict__ tab,
                                                           const LeafStats* __restrict__ old_stats, int dirty_slots,
                                                           int* __restrict__ cell2leaf, size_t c2l_cap,
                                                           int* __restrict__ bnd, int* __restrict__ d_nleaf,
                                                           float4* __restrict__ pts_out) {
  constexpr int TILE = BK_THREADS * ROUNDS;
  __shared__ int cnt[BK_WAVES][SORT_BINS];
  __shared__ float4 stage[TILE];   // the tile in bucket order
  __shared__ int lbase[SORT_BINS];
  __shared__ int wsum[SORT_BINS / 64];
  __shared__ int wrow[BK_WAVES][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tile = tile0 + (int)blockIdx.x;
  // the cells the PREVIOUS build published (the dense grid is filled with -1 once per allocation)
  for (int slot = blockIdx.x * BK_THREADS + threadIdx.x; slot < dirty_slots; slot += gridDim.x * BK_THREADS) {
    const int cell = old_stats[slot].cell;
    if (cell >= 0 && (size_t)cell < c2l_cap) cell2leaf[cell] = -1;
  }
  if (tile == 0 && threadIdx.x == 0) { d_nleaf[0] = 0; d_nleaf[1] = 0; d_nleaf[2] = 0; }
  NDT_BSTAMP(4, 0);
  float a[ROUNDS], b[ROUNDS], c[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {  // all loads of the tile in flight before any use
    const int i = fused_index<ROUNDS>(tile, wave, r, lane);
    const int j = i < n ? i : 0;
    a[r] = x[j]; b[r] = y[j]; c[r] = z[j];
  }
  for (int d = threadIdx.x; d < BK_WAVES * SORT_BINS; d += BK_THREADS) (&cnt[0][0])[d] = 0;
  uint32_t dig[ROUNDS];
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN}, nfin = 0;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int i = fused_index<ROUNDS>(tile, wave, r, lane);
    dig[r] = 0u;  // non-finite points ride in bucket 0; k_bucket_leaves drops them
    if (i < n && finite3(a[r], b[r], c[r])) {
      dig[r] = bucket_of(floorf(a[r] * inv_leaf), floorf(b[r] * inv_leaf), floorf(c[r] * inv_leaf));
      const int ea = encode_ordered(a[r]), eb = encode_ordered(b[r]), ec = encode_ordered(c[r]);
      mn[0] = min(mn[0], ea); mx[0] = max(mx[0], ea);
      mn[1] = min(mn[1], eb); mx[1] = max(mx[1], eb);
      mn[2] = min(mn[2], ec); mx[2] = max(mx[2], ec);
      ++nfin;
    }
  }
  // ref: pcl::getMinMax3D at voxel_grid_covariance_impl.hpp:103 -- this tile's share of the bounds
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      mn[q] = min(mn[q], __shfl_xor(mn[q], off));
      mx[q] = max(mx[q], __shfl_xor(mx[q], off));
    }
    nfin += __shfl_xor(nfin, off);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { wrow[wave][q] = mn[q]; wrow[wave][3 + q] = mx[q]; }
    wrow[wave][6] = nfin;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int t = threadIdx.x;
    int v = wrow[0][t];
    for (int w = 1; w < BK_WAVES; ++w) {
      const int o = wrow[w][t];
      v = t < 3 ? min(v, o) : (t < 6 ? max(v, o) : v + o);
    }
    // Seven words for the whole cloud {min xyz, max xyz, #finite} (order-encoded ints), folded by integer atomics
    // -- 7 per tile, spread over the launch: the next launch reads them with scalar loads and every wave derives the
    // grid geometry in registers, where round 3's first version had every block fold 123 rows through LDS (5 us).
    // Exact and order-free (min / max / integer add).  k_bucket_leaves' last block puts them back to neutral.
    if (t < 3) atomicMin(&bnd[t], v);
    else if (t < 6) atomicMax(&bnd[t], v);
    else atomicAdd(&bnd[6], v);
  }
  NDT_BSTAMP(4, 1);  // points loaded, bounds row written
  // stable rank among the equal buckets before it in the wave's 64 * ROUNDS points (8 ballots per round)
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  int rank[ROUNDS];
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const bool valid = fused_index<ROUNDS>(tile, wave, r, lane) < n;
    const uint32_t d = dig[r];
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool bit = (d >> q) & 1u;
      const unsigned long long bal = __ballot(bit);
      same &= bit ? bal : ~bal;
    }
    const int before = cnt[wave][d];
    const int lower = __popcll(same & lt_mask);
    rank[r] = before + lower;
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == 0) cnt[wave][d] = before + __popcll(same);
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  NDT_BSTAMP(4, 2);  // ranked
  int my_count = 0, l_excl = 0;  // bucket threadIdx.x in this tile: its points, and the tile's points in the buckets before it
  if (threadIdx.x < SORT_BINS) {
    int run = 0;
#pragma unroll
    for (int w = 0; w < BK_WAVES; ++w) {
      const int t = cnt[w][threadIdx.x];
      cnt[w][threadIdx.x] = run;
      run += t;
    }
    my_count = run;
    const int lincl = wave_inclusive_scan(my_count, lane);
    if (lane == 63) wsum[wave] = lincl;
    l_excl = lincl - my_count;
  }
  __syncthreads();
  if (threadIdx.x < SORT_BINS) {
    int lbefore = 0;
    for (int w = 0; w < wave; ++w) lbefore += wsum[w];
    const int first = lbefore + l_excl;   // <= TILE <= 8192: 16 bits each
    lbase[threadIdx.x] = first;
    tab[(size_t)threadIdx.x * BK_MAX_TILES + tile] = ((uint32_t)my_count << 16) | (uint32_t)first;
  }
  __syncthreads();
  NDT_BSTAMP(4, 3);  // bucket offsets of the tile known, column entries issued
  // the tile in bucket order through LDS: the global stores below are one contiguous run
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    if (fused_index<ROUNDS>(tile, wave, r, lane) >= n) break;
    const uint32_t d = dig[r];
    stage[lbase[d] + cnt[wave][d] + rank[r]] = make_float4(a[r], b[r], c[r], __uint_as_float(d));
  }
  __syncthreads();
  NDT_BSTAMP(4, 4);  // staged
  const int tile_n = min(TILE, n - tile * TILE);
  float4* __restrict__ window = pts_out + (size_t)tile * TILE;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    const int j = r * BK_THREADS + (int)threadIdx.x;
    if (j >= tile_n) break;
    window[j] = stage[j];   // (non-temporal stores measured: 54.6 against 54.5 us per build, nothing)
  }
#ifdef NDT_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  NDT_BSTAMP(4, 5);  // thread 0's stores acknowledged
#endif
}

// What one bucket block keeps of a leaf between the run search and the finalize phase (aliases the hash
// table and the sorted ids, which are dead by then)
struct BucketLeaf {
  int cell;
  unsigned short start, cnt;
};

constexpr int BK_TAB = 4096;             // LDS hash table of the distinct voxels of a bucket (C3: ~180 per bucket)
constexpr int BK_MAX_DISTINCT = 3584;    // beyond it (a cloud of isolated points) the bucket declines
constexpr uint32_t BK_EMPTY = 0xFFFFFFFEu;

// Grid geometry in f32 exactly as derive_geometry() (ref: voxel_grid_covariance_impl.hpp:108-140), written field by
// field into LDS: no stack object (derive_geometry's indexed plan arrays live in scratch), no sort plan.
__device__ __forceinline__ void derive_geometry_lean(const int mnmx[6], int n_finite, float leaf, float inv_leaf,
                                                     long long cell_capacity, BuildGeom* out) {
  GridGeom& g = out->g;
  g.leaf = leaf;
  g.inv_leaf = inv_leaf;
  out->n_finite = n_finite;
  out->bits = 1;
  out->passes = 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) { out->width[i] = 0; out->shift[i] = 0; }
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.min_b[a] = 0; g.div_b[a] = 0; g.lo[a] = 0.0f; g.hi[a] = 0.0f; out->max_b[a] = 0; }
  g.mul1 = g.mul2 = g.ncells = 0;
  int status = BG_OK;
  if (n_finite == 0) {
    status = BG_NO_FINITE;
  } else {
    float mn[3], mx[3];
    long long d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      mn[a] = decode_ordered_dev(mnmx[a]);
      mx[a] = decode_ordered_dev(mnmx[3 + a]);
      d[a] = (long long)((mx[a] - mn[a]) * inv_leaf) + 1;
    }
    const long long lim = 2147483647ll;
    if (d[0] < 0 || d[1] < 0 || d[2] < 0 || d[0] > lim || d[1] > lim || d[2] > lim || d[0] * d[1] > lim ||
        d[0] * d[1] * d[2] > lim) {
      status = BG_OVERFLOW;
    } else {
      long long ncells = 1;
      int mnb[3], mxb[3], dv[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        mnb[a] = (int)floorf(mn[a] * inv_leaf);
        mxb[a] = (int)floorf(mx[a] * inv_leaf);
        dv[a] = mxb[a] - mnb[a] + 1;
        ncells *= dv[a];
      }
      if (ncells >= lim) {
        status = BG_OVERFLOW;
      } else {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          g.min_b[a] = mnb[a];
          out->max_b[a] = mxb[a];
          g.div_b[a] = dv[a];
          g.lo[a] = (float)mnb[a] * leaf;
          g.hi[a] = (float)(mxb[a] + 1) * leaf;
        }
        g.mul1 = dv[0];
        g.mul2 = dv[0] * dv[1];
        g.ncells = (int)ncells;
        if (ncells > cell_capacity) status = BG_CAPACITY;
      }
    }
  }
  out->status = status;
}

// v + (v of the partner lane) for the three steps of an 8-lane tree and the fourth of a 16-lane one, through DPP
// (no LDS crossbar round trip): quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.  Each
// step pairs a lane with one whose partial sum covers the OTHER half of the group, like the xor tree: the two
// operands of every addition are the same, so the sums are bit-identical to moments_xor_tree's.
template <int CTRL>
__device__ __forceinline__ double dpp_partner(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, 0xF, 0xF, true);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, 0xF, 0xF, true);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int WIDTH>
__device__ __forceinline__ void moments_tree_dpp(Moments& m) {
#define NDT_TREE_STEP(EXPR)                                   \
  {                                                           \
    _Pragma("unroll") for (int a = 0; a < 3; ++a) m.s[a] += EXPR(m.s[a]);   \
    _Pragma("unroll") for (int a = 0; a < 6; ++a) m.ss[a] += EXPR(m.ss[a]); \
  }
  NDT_TREE_STEP(dpp_partner<0xB1>)
  NDT_TREE_STEP(dpp_partner<0x4E>)
  NDT_TREE_STEP(dpp_partner<0x141>)
  if (WIDTH == 64) {
    NDT_TREE_STEP(dpp_partner<0x140>)
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) m.s[a] += __shfl_xor(m.s[a], off);
#pragma unroll
      for (int a = 0; a < 6; ++a) m.ss[a] += __shfl_xor(m.ss[a], off);
    }
  }
#undef NDT_TREE_STEP
}

// the bounds words between two builds: neutral elements of min / max / add
__device__ __forceinline__ void reset_bounds_words(int* __restrict__ bnd) {
#pragma unroll
  for (int k = 0; k < 3; ++k) { bnd[k] = INT_MAX; bnd[3 + k] = INT_MIN; }
  bnd[6] = 0;
  bnd[7] = 0;
}

__global__ void __launch_bounds__(BK_THREADS) k_bucket_leaves(const float4* __restrict__ pts, const uint32_t* __restrict__ coltab,
                                                             int ntiles, int tile_shift,
                                                             int* __restrict__ bnd, float leaf,
                                                             float inv_leaf, long long cell_capacity, int min_pts,
                                                             FinalizeParams fp, BuildGeom* __restrict__ gd,
                                                             BuildGeom* __restrict__ gd_host, int* __restrict__ d_nleaf,
                                                             unsigned long long* __restrict__ tail, double* __restrict__ sums,
                                                             VoxelRecord* __restrict__ rec, float4* __restrict__ cent,
                                                             LeafStats* __restrict__ stats,
                                                             int* __restrict__ cell2leaf, int max_leaves,
                                                             int* __restrict__ nleaf_host, int done_tag) {
  __shared__ float px[BK_MAXP], py[BK_MAXP], pz[BK_MAXP];
  // region A, three lives: {hash table of the bucket's distinct cells + their dense ids} -> {ids in sorted order +
  // the per-wave digit counters} -> {leaf list}
  __shared__ __align__(16) unsigned char region_a[BK_TAB * 4 + BK_TAB * 2];
  uint32_t* tab = reinterpret_cast<uint32_t*>(region_a);
  unsigned short* did = reinterpret_cast<unsigned short*>(region_a + BK_TAB * 4);
  unsigned short* sid = reinterpret_cast<unsigned short*>(region_a);                                   // BK_MAXP ids
  unsigned short (*cnt)[SORT_BINS] = reinterpret_cast<unsigned short (*)[SORT_BINS]>(region_a + BK_TAB * 4);  // 16 x 256
  BucketLeaf* leaves = reinterpret_cast<BucketLeaf*>(region_a);
  static_assert(BK_MAXP * 2 <= BK_TAB * 4 && BK_WAVES * SORT_BINS * 2 <= BK_TAB * 2, "sorted ids / counters alias the table");
  static_assert(sizeof(BucketLeaf) * BK_MAX_LEAVES <= sizeof(region_a), "leaf list aliases region A");
  __shared__ unsigned short sidx[BK_MAXP];
  __shared__ int dbase[SORT_BINS];
  __shared__ uint32_t idcell[SORT_BINS];   // the cell behind a dense id (buckets of <= 256 distinct cells: one digit pass)
  __shared__ int wsum[BK_WAVES];
  __shared__ int wave_head[BK_WAVES], wave_total[BK_WAVES];
  __shared__ int s_base, s_ok, s_decline;
  // where the bucket's points lie: position p of the bucket (input order) is pts[t_src[t] + p] for the last tile t
  // with t_first[t] <= p (t_first: first position a tile contributes; INT_MAX beyond the launch's tiles)
  __shared__ int t_first[BK_MAX_TILES + 1], t_src[BK_MAX_TILES];
  __shared__ int csum[BK_MAX_TILES / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bucket = blockIdx.x;
  NDT_BSTAMP(5, 0);
  // ---- everything this block needs before its points is requested now: the bounds words and its column of the
  // partition's table (the launch starts on cold caches: every dependent round trip is ~2 us) ----
  uint32_t col = 0u;
  if ((int)threadIdx.x < ntiles) col = coltab[(size_t)bucket * BK_MAX_TILES + threadIdx.x];
  int bw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) bw[k] = bnd[k];   // uniform: scalar loads
  int m_all;
  {
    const int mine = (int)(col >> 16);
    const int incl = wave_inclusive_scan(mine, lane);
    if (lane == 63 && wave < BK_MAX_TILES / 64) csum[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    m_all = 0;
#pragma unroll
    for (int w = 0; w < BK_MAX_TILES / 64; ++w) {
      if (w < wave) before += csum[w];
      m_all += csum[w];
    }
    m_all = __builtin_amdgcn_readfirstlane(m_all);   // (uniform: the round counts below stay scalar)
    if (threadIdx.x < BK_MAX_TILES) {
      const bool live = (int)threadIdx.x < ntiles;
      t_first[threadIdx.x] = live ? before : INT_MAX;
      t_src[threadIdx.x] = ((int)threadIdx.x << tile_shift) + (int)(col & 0xffffu) - before;
    }
    if (threadIdx.x == 0) t_first[BK_MAX_TILES] = INT_MAX;
    __syncthreads();
  }
  NDT_BSTAMP(5, 7);  // column scanned: the bucket knows where its points lie
  // a bucket beyond a block's LDS (a voxel holding thousands of points, a cloud beyond ~1.3 M): this block takes no
  // point and says so in the launch's tail word; the host clears the grid and repeats the build sort-based
  const bool oversize = m_all > BK_MAXP;
  const int m = oversize ? 0 : m_all;
  // Position p = wave * C + round * 64 + lane: a wave owns C consecutive positions, so "tile order" is (wave, round,
  // lane) as in the sort passes above; C is the smallest multiple of 64 that spreads the bucket over all 16 waves
  // (a bucket of 3906 points: 4 rounds on every wave instead of 8 rounds on half of them).
  const int C = ((m + BK_WAVES * 64 - 1) / (BK_WAVES * 64)) * 64, R = C >> 6;
  float4 q[BK_ROUNDS];
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; ++r) {
    const int p = wave * C + r * 64 + lane;
    q[r] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (r < R && p < m) {
      int t = 0;
#pragma unroll
      for (int step = BK_MAX_TILES / 2; step >= 1; step >>= 1)
        if (t_first[t + step] <= p) t += step;
      q[r] = pts[t_src[t] + p];
    }
  }
  // ---- geometry: every wave derives the same BuildGeom from the seven bounds words, in registers ---------------
  // (same f32 arithmetic as voxel_grid_covariance_impl.hpp:108-140; no LDS, no barrier)
  // (only the grid geometry stays live through the kernel; the block that ends the launch derives the whole
  // BuildGeom once more from the eight words for the host -- kept in registers it cost 100 bytes of scratch)
  auto derive = [&](BuildGeom* out) {
    derive_geometry_lean(bw, bw[6], leaf, inv_leaf, cell_capacity, out);
    if (out->status == BG_OK) {
      // exactness of floor(p / leaf) - min_b in f32 (one voxel, one bucket), and a bucket must fit a block
      bool big = false;
#pragma unroll
      for (int k = 0; k < 6; ++k) big = big || !(fabsf(floorf(decode_ordered_dev(bw[k]) * inv_leaf)) < BK_COORD_LIMIT);
      if (big) out->status = BG_BUCKET;
    }
  };
  GridGeom g;
  int status;
  {
    BuildGeom lg;
    derive(&lg);
    g = lg.g;
    status = lg.status;
  }
  if (threadIdx.x == 0) { s_ok = 0; s_decline = 0; }
  for (int t = threadIdx.x; t < BK_TAB; t += BK_THREADS) tab[t] = BK_EMPTY;
  __syncthreads();
  if (status != BG_OK) {
    // refused (the host repeats the build another way, or reports the error): nothing was written
    if (bucket == 0 && threadIdx.x == 0) {
      BuildGeom lg;
      derive(&lg);
      *gd = lg;
      *gd_host = lg;
      reset_bounds_words(bnd);
      nleaf_host[0] = 0;
      nleaf_host[1] = 0;
      // the verdict first, the tag the host polls for last (gd_host and nleaf_host are __restrict__: without
      // the release the compiler and the memory system are free to let the tag overtake the verdict)
      __hip_atomic_store(nleaf_host + 2, done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  NDT_BSTAMP(5, 1);  // geometry derived
  // ---- LDS copy of the points; cell key (ref: voxel_grid_covariance_impl.hpp:222-225); the distinct cells of the
  // bucket go into an LDS hash table (a few hundred of them): sorting by the table's DENSE ids takes one or two
  // narrow digit passes instead of three 8-bit passes over the 24-bit cell index -------------------------------
  unsigned short tslot[BK_ROUNDS];
  uint32_t ckey[BK_ROUNDS], cold[BK_ROUNDS];
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; ++r) {
    const int p = wave * C + r * 64 + lane;
    tslot[r] = 0;
    ckey[r] = BK_EMPTY;   // "no point here"
    if (r < R && p < m) {
      px[p] = q[r].x; py[p] = q[r].y; pz[p] = q[r].z;
      uint32_t key = (uint32_t)g.ncells;  // sentinel: never a leaf
      if (finite3(q[r].x, q[r].y, q[r].z)) {
        const int c = cell_of(q[r].x, q[r].y, q[r].z, g);
        if (c >= 0 && c < g.ncells) key = (uint32_t)c;
      }
      ckey[r] = key;
      tslot[r] = (unsigned short)((key * 0x9E3779B1u) >> (32 - 12));
    }
  }
  // first probe of every round in flight together (an LDS atomic with return is ~120 cycles; one after the other
  // they were a chain of R of them), collisions resolved afterwards
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; ++r) {
    cold[r] = BK_EMPTY;
    if (r < R && ckey[r] != BK_EMPTY) cold[r] = atomicCAS(&tab[tslot[r]], BK_EMPTY, ckey[r]);
  }
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; ++r) {
    if (r < R && ckey[r] != BK_EMPTY && cold[r] != BK_EMPTY && cold[r] != ckey[r]) {
      uint32_t hsl = tslot[r];
      for (int probe = 1; probe < BK_TAB; ++probe) {
        hsl = (hsl + 1) & (BK_TAB - 1);
        const uint32_t old = atomicCAS(&tab[hsl], BK_EMPTY, ckey[r]);
        if (old == BK_EMPTY || old == ckey[r]) break;
        if (probe == BK_TAB - 1) s_decline = 1;  // table full
      }
      tslot[r] = (unsigned short)hsl;
    }
  }
  NDT_BSTAMP(5, 2);  // points in LDS, cells in the table
  __syncthreads();
  NDT_BSTAMP(1, 0);  // table complete (barrier passed)
  // dense ids: exclusive scan of the table's occupancy (4 slots per thread)
  int ndistinct;
  {
    const uint4 t4 = *reinterpret_cast<const uint4*>(tab + threadIdx.x * 4);
    const int o0 = t4.x != BK_EMPTY, o1 = t4.y != BK_EMPTY, o2 = t4.z != BK_EMPTY, o3 = t4.w != BK_EMPTY;
    const int mine = o0 + o1 + o2 + o3;
    const int incl = wave_inclusive_scan(mine, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = incl - mine, total = 0;
    for (int w = 0; w < BK_WAVES; ++w) {
      if (w < wave) before += wsum[w];
      total += wsum[w];
    }
    ndistinct = total;
    unsigned short* dd = did + threadIdx.x * 4;
    dd[0] = (unsigned short)before;
    dd[1] = (unsigned short)(before + o0);
    dd[2] = (unsigned short)(before + o0 + o1);
    dd[3] = (unsigned short)(before + o0 + o1 + o2);
    if (total <= SORT_BINS) {   // (uniform) id -> cell, for the leaf list that comes straight out of the digit counters
      if (o0) idcell[before] = t4.x;
      if (o1) idcell[before + o0] = t4.y;
      if (o2) idcell[before + o0 + o1] = t4.z;
      if (o3) idcell[before + o0 + o1 + o2] = t4.w;
    }
    __syncthreads();
  }
  NDT_BSTAMP(1, 1);  // dense ids written
  const bool decline = oversize || s_decline != 0 || ndistinct > BK_MAX_DISTINCT;   // uniform over the block
  uint32_t key[BK_ROUNDS];   // the dense id of the point's cell from here on
  unsigned short idx[BK_ROUNDS];
#pragma unroll
  for (int r = 0; r < BK_ROUNDS; ++r) {
    const int p = wave * C + r * 64 + lane;
    key[r] = (r < R && p < m) ? (uint32_t)did[tslot[r]] : 0u;
    idx[r] = (unsigned short)p;
  }
  __syncthreads();   // table and ids are dead: region A becomes {sorted ids, digit counters}
  NDT_BSTAMP(1, 2);  // ids fetched, region A free
  // ---- stable LSD radix sort of (dense id, position) in LDS: ceil(log2(ndistinct) / 8) passes ---------------
  int idbits = 1;
  while ((1 << idbits) < ndistinct) ++idbits;
  const int npass = decline ? 0 : (idbits + 7) / 8;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // One digit pass (<= 256 distinct cells, the usual bucket): the digit IS the dense id, so the pass's counters are the
  // runs -- id d starts at dbase[d] and holds its column total -- and the leaf list is a scan over 256 flags under the
  // scatter, where the general path below walks the 8192 sorted positions once more (two barriers, ~3 us).  Same leaves
  // in the same (id) order with the same slots.
  const bool one_pass = npass == 1;
  int id_count = 0, id_start = 0, id_flag = 0, id_incl = 0;
  uint32_t id_cell = 0u;
  for (int pass = 0; pass < npass; ++pass) {
    const int width = (idbits + npass - 1 - pass) / npass;           // e.g. 11 bits: 6 + 5
    const int shift = pass == 0 ? 0 : (idbits + npass - 1) / npass;  // (two passes at most: ndistinct <= 3584)
    const uint32_t digit_mask = (1u << width) - 1u;
    for (int d = lane; d < SORT_BINS; d += 64) cnt[wave][d] = 0;
    __builtin_amdgcn_wave_barrier();
    int rank[BK_ROUNDS];
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; ++r) {
      rank[r] = 0;
      if (r < R) {
        const bool valid = wave * C + r * 64 + lane < m;
        const uint32_t d = (key[r] >> shift) & digit_mask;
        unsigned long long same = __ballot(valid);
        for (int k = 0; k < width; ++k) {
          const bool bit = (d >> k) & 1u;
          const unsigned long long bal = __ballot(bit);
          same &= bit ? bal : ~bal;
        }
        const int before = cnt[wave][d];
        const int lower = __popcll(same & lt_mask);
        rank[r] = before + lower;
        __builtin_amdgcn_wave_barrier();
        if (valid && lower == 0) cnt[wave][d] = (unsigned short)(before + __popcll(same));
        __builtin_amdgcn_wave_barrier();
      }
    }
    __syncthreads();
    NDT_BSTAMP(1, 3);  // ranked
    if (threadIdx.x < SORT_BINS) {
      int run = 0;
#pragma unroll
      for (int w = 0; w < BK_WAVES; ++w) {
        const int t = cnt[w][threadIdx.x];
        cnt[w][threadIdx.x] = (unsigned short)run;
        run += t;
      }
      const int incl = wave_inclusive_scan(run, lane);
      if (lane == 63) wsum[wave] = incl;
      dbase[threadIdx.x] = incl - run;
      id_count = run;
    }
    __syncthreads();
    if (threadIdx.x < SORT_BINS) {
      int before = 0;
      for (int w = 0; w < wave; ++w) before += wsum[w];
      dbase[threadIdx.x] += before;
      id_start = dbase[threadIdx.x];
      if (one_pass) {
        // ref :270-273: a run of >= min_pts points whose cell lies in the box (the points that classify nowhere share
        // the sentinel key g.ncells) is a leaf
        if ((int)threadIdx.x < ndistinct && id_count >= min_pts) {
          id_cell = idcell[threadIdx.x];
          id_flag = id_cell < (uint32_t)g.ncells ? 1 : 0;
        }
        id_incl = wave_inclusive_scan(id_flag, lane);
        if (lane == 63) wave_total[wave] = id_incl;
      }
    }
    __syncthreads();
    NDT_BSTAMP(1, 4);  // digit bases known
#pragma unroll
    for (int r = 0; r < BK_ROUNDS; ++r) {
      if (r < R && wave * C + r * 64 + lane < m) {
        const uint32_t d = (key[r] >> shift) & digit_mask;
        const int o = dbase[d] + cnt[wave][d] + rank[r];
        if (!one_pass) sid[o] = (unsigned short)key[r];   // (the leaf list lies where the sorted ids would)
        sidx[o] = idx[r];
      }
    }
    if (one_pass && threadIdx.x < SORT_BINS && id_flag) {
      int li = id_incl - 1;
      for (int w = 0; w < wave; ++w) li += wave_total[w];
      BucketLeaf L;
      L.cell = (int)id_cell;
      L.start = (unsigned short)id_start;
      L.cnt = (unsigned short)id_count;
      leaves[li] = L;
    }
    __syncthreads();
    if (pass + 1 < npass) {
#pragma unroll
      for (int r = 0; r < BK_ROUNDS; ++r) {
        const int p = wave * C + r * 64 + lane;
        if (r < R && p < m) { key[r] = sid[p]; idx[r] = sidx[p]; }
      }
      __syncthreads();
    }
  }
  NDT_BSTAMP(5, 3);  // sorted
  int nl = 0;
  if (one_pass) {
    nl = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    if (threadIdx.x == 0) s_base = nl > 0 ? atomicAdd(&d_nleaf[0], nl) : 0;
  } else {
  // ---- runs of equal id: thread t owns the 8 consecutive sorted positions [8 t, 8 t + 8) ------------
  const int ms = decline ? 0 : m;
  const int s0 = (int)threadIdx.x * BK_ROUNDS;
  uint32_t k[BK_ROUNDS];
#pragma unroll
  for (int j = 0; j < BK_ROUNDS; ++j) k[j] = s0 + j < ms ? (uint32_t)sid[s0 + j] : 0xFFFFFFFFu;
  const uint32_t prev = (s0 > 0 && s0 - 1 < ms) ? (uint32_t)sid[s0 - 1] : 0xFFFFFFFFu;
  const uint32_t next = s0 + BK_ROUNDS < ms ? (uint32_t)sid[s0 + BK_ROUNDS] : 0xFFFFFFFFu;
  int own_head = -1;
#pragma unroll
  for (int j = 0; j < BK_ROUNDS; ++j) {
    const uint32_t before = j == 0 ? prev : k[j - 1];
    if (s0 + j < ms && (s0 + j == 0 || before != k[j])) own_head = s0 + j;
  }
  int incl = own_head;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl = max(incl, t);
  }
  int head_before = __shfl_up(incl, 1);
  if (lane == 0) head_before = -1;
  if (lane == 63) wave_head[wave] = incl;
  __syncthreads();   // (also: every thread has its ids in registers -- region A may be overwritten below)
  for (int w = 0; w < wave; ++w) head_before = max(head_before, wave_head[w]);
  int cur_head = head_before, nleaf = 0;
  int starts[BK_ROUNDS], counts[BK_ROUNDS], cells[BK_ROUNDS];
#pragma unroll
  for (int j = 0; j < BK_ROUNDS; ++j) {
    const int s = s0 + j;
    const uint32_t before = j == 0 ? prev : k[j - 1];
    const uint32_t after = j == BK_ROUNDS - 1 ? next : k[j + 1];
    const bool valid = s < ms;
    if (valid && (s == 0 || before != k[j])) cur_head = s;
    const bool tail = valid && (s == ms - 1 || after != k[j]);
    counts[j] = 0;
    starts[j] = 0;
    cells[j] = 0;
    if (tail) {
      const int c = s - cur_head + 1;
      if (c >= min_pts) {   // ref :270-273
        // the run's cell, from its first point (every point of a run has the same one); the run of the points
        // that classify nowhere (non-finite, outside the box) is no leaf
        const int f = sidx[cur_head];
        const float fx = px[f], fy = py[f], fz = pz[f];
        if (finite3(fx, fy, fz)) {
          const int cell = cell_of(fx, fy, fz, g);
          if (cell >= 0 && cell < g.ncells) { starts[j] = cur_head; counts[j] = c; cells[j] = cell; ++nleaf; }
        }
      }
    }
  }
  int lincl = wave_inclusive_scan(nleaf, lane);
  if (lane == 63) wave_total[wave] = lincl;
  __syncthreads();
  int li0 = lincl - nleaf;
  for (int w = 0; w < BK_WAVES; ++w) {
    if (w < wave) li0 += wave_total[w];
    nl += wave_total[w];
  }
#pragma unroll
  for (int j = 0; j < BK_ROUNDS; ++j)
    if (counts[j] > 0) {
      BucketLeaf L;
      L.cell = cells[j];
      L.start = (unsigned short)starts[j];
      L.cnt = (unsigned short)counts[j];
      leaves[li0++] = L;
    }
  // this block's leaf slots: the atomic's round trip runs under the first batch of sums
  if (threadIdx.x == 0) s_base = nl > 0 ? atomicAdd(&d_nleaf[0], nl) : 0;
  __syncthreads();   // leaf list complete (s_base is read behind the next barrier only)
  }  // (general run search)
  NDT_BSTAMP(5, 4);  // runs found
  NDT_WSTAMP(0);
  // ---- per-voxel sums (ref :236-239), from LDS, in input order: 8 lanes per leaf, crowded leaves by the wave ----
  int slot0 = 0;
  {
    const int sub = lane & (LANES_PER_LEAF - 1);
    for (int l0 = 0; l0 < nl; l0 += BK_THREADS / LANES_PER_LEAF) {
      const int li = l0 + ((int)threadIdx.x >> 3);
      const bool have = li < nl;
      const int start = have ? (int)leaves[li].start : 0, c = have ? (int)leaves[li].cnt : 0;
      Moments mo{};
      const int head = c < LEAF_HEAD ? c : LEAF_HEAD;
      // (same association as k_leaf_sums: a lane adds its points in ascending order, masked lanes add exact zeros)
      for (int j0 = sub; j0 < head; j0 += SUMS_UNROLL * LANES_PER_LEAF) {
        float4 pp[SUMS_UNROLL];
        bool live[SUMS_UNROLL];
#pragma unroll
        for (int u = 0; u < SUMS_UNROLL; ++u) {
          const int j = j0 + u * LANES_PER_LEAF;
          live[u] = j < head;
          const int f = sidx[start + (live[u] ? j : 0)];
          pp[u] = make_float4(px[f], py[f], pz[f], 0.0f);
        }
#pragma unroll
        for (int u = 0; u < SUMS_UNROLL; ++u) moments_add(mo, pp[u], live[u]);
      }
      moments_tree_dpp<LANES_PER_LEAF>(mo);
      NDT_WSTAMP(1);
      unsigned long long crowded = __ballot(c > LEAF_HEAD);
      while (crowded) {
        const int src = __ffsll((long long)crowded) - 1;
        crowded &= ~(0xFFull << (src & ~7));
        const int bstart = __shfl(start, src), bcnt = __shfl(c, src);
        Moments t{};
        for (int j0 = LEAF_HEAD + lane; j0 < bcnt; j0 += SUMS_UNROLL * 64) {
          float4 pp[SUMS_UNROLL];
          bool live[SUMS_UNROLL];
#pragma unroll
          for (int u = 0; u < SUMS_UNROLL; ++u) {
            const int j = j0 + u * 64;
            live[u] = j < bcnt;
            const int f = sidx[bstart + (live[u] ? j : LEAF_HEAD)];
            pp[u] = make_float4(px[f], py[f], pz[f], 0.0f);
          }
#pragma unroll
          for (int u = 0; u < SUMS_UNROLL; ++u) moments_add(t, pp[u], live[u]);
        }
        moments_tree_dpp<64>(t);
        if ((lane >> 3) == (src >> 3)) {
#pragma unroll
          for (int a = 0; a < 3; ++a) mo.s[a] += t.s[a];
#pragma unroll
          for (int a = 0; a < 6; ++a) mo.ss[a] += t.ss[a];
        }
      }
      NDT_WSTAMP(2);
      if (l0 == 0) {   // uniform: the slot base has arrived by now
        __syncthreads();
        slot0 = s_base;
      }
      if (!have) continue;
      double* o = sums + (size_t)(slot0 + li) * 9;
      const double mine = sub == 0 ? mo.s[0] : sub == 1 ? mo.s[1] : sub == 2 ? mo.s[2] : sub == 3 ? mo.ss[0]
                        : sub == 4 ? mo.ss[1] : sub == 5 ? mo.ss[2] : sub == 6 ? mo.ss[3] : mo.ss[4];
      o[sub] = mine;
      if (sub == 0) o[8] = mo.ss[5];
    }
  }
  NDT_WSTAMP(3);
  __syncthreads();   // the sums of this block's leaves are visible to the whole block
  NDT_WSTAMP(4);
  NDT_BSTAMP(5, 5);  // sums written
  // ---- ref :265-343: one thread per leaf ---------------------------------------------------------------
  int ok_here = 0;
  for (int li = threadIdx.x; li < nl; li += BK_THREADS) {
    const int slot = slot0 + li;
    if (slot < max_leaves && finalize_leaf(slot, leaves[li].cell, (int)leaves[li].cnt, sums + (size_t)slot * 9, fp, rec, cent, stats, cell2leaf))
      ++ok_here;
  }
  NDT_WSTAMP(5);  // this wave's leaves finalised, stores issued
  const int wave_ok = __popcll(__ballot(ok_here == 1)) + 2 * __popcll(__ballot(ok_here == 2)) + 3 * __popcll(__ballot(ok_here >= 3));
  if (lane == 0 && wave_ok) atomicAdd(&s_ok, wave_ok);
  __syncthreads();
  NDT_BSTAMP(5, 6);  // statistics written
  if (threadIdx.x == 0) {
    // ONE 64-bit atomic ends the block: its ticket and its contributions to the launch's totals travel together, so
    // the block that draws the last ticket holds every block's counts in the returned value -- {ticket : 11 | buckets
    // that declined : 9 | accepted leaves : 22 | leaf slots : 22}.  (Until round 3: an add for the accepted leaves,
    // wait, the ticket, wait, three loads of the totals -- three round trips at the very end of the build.)
    const unsigned long long add = (1ull << 53) | ((unsigned long long)(decline ? 1u : 0u) << 44) |
                                   ((unsigned long long)(unsigned int)s_ok << 22) | (unsigned long long)(unsigned int)nl;
    const unsigned long long prev = __hip_atomic_fetch_add(tail, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 53) == (unsigned long long)gridDim.x - 1ull) {
      const unsigned long long total = prev + add;
      const int slots = (int)(total & 0x3fffffull), valid = (int)((total >> 22) & 0x3fffffull);
      // a bucket with more distinct cells than its table holds declined AFTER other blocks had published leaves:
      // the host repeats the build sort-based on a cleared grid
      BuildGeom lg;
      derive(&lg);
      if (((total >> 44) & 0x1ffull) != 0ull) lg.status = BG_BUCKET;
      *gd = lg;
      *gd_host = lg;
      reset_bounds_words(bnd);   // every block read them at its entry, long ago: neutral again for the next build
      d_nleaf[1] = valid;
      nleaf_host[0] = slots;
      nleaf_host[1] = valid;
      // geometry and counts first, the tag the host polls for last (see above)
      __hip_atomic_store(nleaf_host + 2, done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(tail, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Sliding-window target assembly: one archived body-frame scan moved into the
// map frame by its current pose and appended to the target arrays.  The reference does this
// on the host with a DOUBLE 4x4 (gtsam Pose3::matrix()) through pcl::transformPointCloud
// (ref: run/pipeline_ligo_tc.cpp:519-526, run/pipeline.cpp:554-556): f64 products summed
// left to right, rounded to f32 once.
struct Affine64 {
  double R[9];
  double t[3];
};

__global__ void __launch_bounds__(256) k_transform_append(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, size_t n, Affine64 T,
                                                         float* __restrict__ ox, float* __restrict__ oy,
                                                         float* __restrict__ oz) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = (double)x[i], b = (double)y[i], c = (double)z[i];
  ox[i] = (float)(T.R[0] * a + T.R[1] * b + T.R[2] * c + T.t[0]);
  oy[i] = (float)(T.R[3] * a + T.R[4] * b + T.R[5] * c + T.t[1]);
  oz[i] = (float)(T.R[6] * a + T.R[7] * b + T.R[8] * c + T.t[2]);
}

}  // namespace

namespace {

// ---- source ordering ("source points pre-sorted by target voxel key once per
// align so a wavefront touches few distinct voxels") -------------------------------------------
// Key of a source point = the 8 x 8 x 4-voxel block of the target grid its image under the
// initial guess falls into (points outside the box are clamped onto it).  A stable sort by that
// 16-bit-ish key makes the 64 points of a wave share a few hundred voxel records at most.  It
// only pays when the voxel table does not sit in L2 anyway: on C3-wide (3.2e5 leaves, 26 MB of
// records) the derivative kernel drops from 32.3 to 18.0 us per evaluation and its HBM fetch
// from 137 MB to the algorithmic 51 MB; on C3 (1.7 MB table) the scan order is already coherent.
// Any permutation of the source is the same source: only the f64 summation order changes.
struct SourceBlocks {
  int nbx, nby, nbz;
  int bits;
};
__host__ __device__ inline SourceBlocks source_blocks(const GridGeom& g) {
  SourceBlocks b;
  b.nbx = (g.div_b[0] + 7) >> 3;
  b.nby = (g.div_b[1] + 7) >> 3;
  b.nbz = (g.div_b[2] + 3) >> 2;
  const long long nb = (long long)b.nbx * b.nby * b.nbz;
  int bits = 1;
  while (bits < 31 && (1ll << bits) < nb) ++bits;
  b.bits = bits;
  return b;
}

__global__ void __launch_bounds__(SORT_THREADS) k_source_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ z, int n, GridGeom g,
                                                             PoseConsts P, BuildGeom* __restrict__ plan_out,
                                                             uint32_t* __restrict__ keys, int ntiles,
                                                             int* __restrict__ hist) {
  __shared__ int h[SORT_BINS];
  const SourceBlocks sb = source_blocks(g);
  // the sort plan of these keys, for the passes that follow (every block derives the same one)
  BuildGeom plan;
  plan.bits = sb.bits;
  plan.passes = (sb.bits + 7) / 8;
  {
    const int base = sb.bits / plan.passes, rem = sb.bits % plan.passes;
    int sh = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      plan.width[i] = i < plan.passes ? base + (i < rem ? 1 : 0) : 0;
      plan.shift[i] = sh;
      sh += plan.width[i];
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    plan_out->bits = plan.bits;
    plan_out->passes = plan.passes;
#pragma unroll
    for (int i = 0; i < 4; ++i) { plan_out->width[i] = plan.width[i]; plan_out->shift[i] = plan.shift[i]; }
    plan_out->status = BG_OK;
  }
  const uint32_t digit_mask = (1u << plan.width[0]) - 1u;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int i = sort_index(blockIdx.x, wave, r, lane);
    if (i >= n) break;
    const float px = x[i], py = y[i], pz = z[i];
    const float xt = P.R[0] * px + (P.R[1] * py + (P.R[2] * pz + P.t[0]));
    const float yt = P.R[3] * px + (P.R[4] * py + (P.R[5] * pz + P.t[1]));
    const float zt = P.R[6] * px + (P.R[7] * py + (P.R[8] * pz + P.t[2]));
    uint32_t key = 0u;
    if (finite3(xt, yt, zt)) {
      // clamped in float first: a far-away point must not overflow the int conversion
      const float fx = fminf(fmaxf(floorf(xt * g.inv_leaf) - (float)g.min_b[0], 0.0f), (float)(g.div_b[0] - 1));
      const float fy = fminf(fmaxf(floorf(yt * g.inv_leaf) - (float)g.min_b[1], 0.0f), (float)(g.div_b[1] - 1));
      const float fz = fminf(fmaxf(floorf(zt * g.inv_leaf) - (float)g.min_b[2], 0.0f), (float)(g.div_b[2] - 1));
      key = (uint32_t)(((int)fx >> 3) + ((int)fy >> 3) * sb.nbx + ((int)fz >> 2) * sb.nbx * sb.nby);
    }
    keys[i] = key;
    atomicAdd(&h[key & digit_mask], 1);
  }
  __syncthreads();
  hist[threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_gather_soa(const uint32_t* __restrict__ perm, const float* __restrict__ x,
                                                   const float* __restrict__ y, const float* __restrict__ z, int n,
                                                   float* __restrict__ ox, float* __restrict__ oy,
                                                   float* __restrict__ oz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = perm[i];
  ox[i] = x[j]; oy[i] = y[j]; oz[i] = z[j];
}

}  // namespace

int sort_tiles(size_t n);

int source_sort_passes(const GridGeom& g) { return (source_blocks(g).bits + 7) / 8; }

// keys + first histogram, `passes` sort passes, gather: the source in block order of the target
// grid under the transform P.  keys/vals a,b: n uint32 each; temp: sort_temp_bytes(n); plan: one
// BuildGeom in device memory (scratch).
hipError_t sort_source_by_blocks(const float* x, const float* y, const float* z, size_t n, const GridGeom& g,
                                 const PoseConsts& P, BuildGeom* plan, void* temp, uint32_t* keys_a,
                                 uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, float* ox, float* oy,
                                 float* oz, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int ntiles = sort_tiles(n);
  hipLaunchKernelGGL(k_source_keys, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, x, y, z, (int)n, g, P, plan, keys_a,
                     ntiles, static_cast<int*>(temp));
  bool in_b = false;
  hipError_t e = sort_pairs(temp, keys_a, keys_b, vals_a, vals_b, n, source_sort_passes(g), plan, s, &in_b);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gather_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in_b ? vals_b : vals_a, x, y, z,
                     (int)n, ox, oy, oz);
  return hipGetLastError();
}

// Host hand-off: one chunk of a cloud, laid out in MAPPED PINNED HOST memory as [x(seg) | y(seg) | z(seg)] (seg = the
// chunk's point count rounded up to 4), is pulled over PCIe straight into the three SoA arrays.  A kernel per chunk
// instead of a copy-engine transfer per chunk plus a conversion kernel: the copy engine pays ~10 us per transfer (eight
// 1.5 MB copies: 41 GB/s, one 12 MB copy: 55), a 64-block pull kernel per chunk reaches 49 GB/s and needs no second
// pass over the data (tools/h2d_pull_probe.hip, profiles/r04_h2d_pull_probe.txt).  Few blocks on purpose: PCIe needs
// ~100 KB in flight, not the machine.
__global__ void __launch_bounds__(256) k_pull_chunk(const float* __restrict__ src, size_t len, size_t seg,
                                                   float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
  const size_t n4 = len >> 2, stride = (size_t)gridDim.x * blockDim.x, t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool vec = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)z | (uintptr_t)src) & 15) == 0;
  float* const dst[3] = {x, y, z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float* s = src + (size_t)a * seg;
    float* d = dst[a];
    if (vec) {
      for (size_t i = t; i < n4; i += stride) reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
      if (t < (len & 3)) d[(n4 << 2) + t] = s[(n4 << 2) + t];
    } else {
      for (size_t i = t; i < len; i += stride) d[i] = s[i];
    }
  }
}

// pcl::VoxelGrid centroids (ref: run/pipeline_ins_map_distribution.cpp:324-340; PCL's published algorithm: every field
// averaged over the points of an occupied voxel, output in ascending voxel index): run r of the sorted keys =
// points vals_sorted[start .. start + cnt) -> one output point.  float sums in sorted order (the radix sort is stable:
// input order within a voxel) and a float division, as PCL's CentroidPoint accumulates -- PCL's own order within a voxel
// is whatever its unstable index sort left, so agreement with it is to rounding, with a stable-sort restatement exact.
__global__ void __launch_bounds__(256) k_voxel_centroids(const float4* __restrict__ xyz4, const float* __restrict__ inten,
                                                        const uint32_t* __restrict__ vals_sorted, const int* __restrict__ d_nleaf,
                                                        const int* __restrict__ leaf_start, const int* __restrict__ leaf_cnt,
                                                        int cap, float* __restrict__ ox, float* __restrict__ oy,
                                                        float* __restrict__ oz, float* __restrict__ oi) {
  const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= d_nleaf[0] || r >= cap) return;
  const int start = leaf_start[r], cnt = leaf_cnt[r];
  float sx = 0.0f, sy = 0.0f, sz = 0.0f, si = 0.0f;
  for (int j = 0; j < cnt; ++j) {
    const uint32_t p = vals_sorted[start + j];
    const float4 q = xyz4[p];
    sx += q.x; sy += q.y; sz += q.z;
    if (inten) si += inten[p];
  }
  const float nf = (float)cnt;
  ox[r] = sx / nf; oy[r] = sy / nf; oz[r] = sz / nf;
  if (oi) oi[r] = inten ? si / nf : 0.0f;
}

// multi-grid union table: cell2leaf[cells[i]] = slots[i] for the first leaf of every occupied cell
__global__ void __launch_bounds__(256) k_scatter_heads(const int* __restrict__ cells, const int* __restrict__ slots,
                                                      int n, int* __restrict__ cell2leaf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cell2leaf[cells[i]] = slots[i];
}

__global__ void __launch_bounds__(256) k_copy_soa(const float* __restrict__ x, const float* __restrict__ y,
                                                 const float* __restrict__ z, size_t n, float* __restrict__ ox,
                                                 float* __restrict__ oy, float* __restrict__ oz) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 4 <= n && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)z | (uintptr_t)ox | (uintptr_t)oy | (uintptr_t)oz) & 15) == 0)) {
    *reinterpret_cast<float4*>(ox + i) = *reinterpret_cast<const float4*>(x + i);
    *reinterpret_cast<float4*>(oy + i) = *reinterpret_cast<const float4*>(y + i);
    *reinterpret_cast<float4*>(oz + i) = *reinterpret_cast<const float4*>(z + i);
  } else {
    for (size_t j = i; j < n && j < i + 4; ++j) { ox[j] = x[j]; oy[j] = y[j]; oz[j] = z[j]; }
  }
}

void launch_copy_soa(const float* x, const float* y, const float* z, size_t n, float* ox, float* oy, float* oz,
                     hipStream_t s) {
  if (n == 0) return;
  const size_t threads = (n + 3) / 4;
  hipLaunchKernelGGL(k_copy_soa, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, x, y, z, n, ox, oy, oz);
}

void launch_transform_append(const float* x, const float* y, const float* z, size_t n,
                             const double pose_colmajor[16], float* ox, float* oy, float* oz,
                             hipStream_t s) {
  if (n == 0) return;
  Affine64 T;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T.R[3 * i + j] = pose_colmajor[4 * j + i];
    T.t[i] = pose_colmajor[12 + i];
  }
  hipLaunchKernelGGL(k_transform_append, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, z, n, T,
                     ox, oy, oz);
}

// Launch shapes of the build kernels that end in a "last block finishes the job" ticket: A/B switches of ndt_tuning
// (include/ndt_hip.h; profiles/r02_build_tickets.txt), none of them read from the environment by the production library.
struct BuildTuning {
  int bounds_blocks;     // <= BOUNDS_BLOCKS
  int bounds_unroll;     // 4 | 8 (16 measured: no change)
  int finalize_threads;  // 64 | 256
  int fused_sort;        // 0 | 1: one launch per sort digit where the cloud allows it
};
BuildTuning build_tuning() {
  const ndt_tuning& t = tuning();
  BuildTuning b;
  b.bounds_blocks = t.bounds_blocks;
  if (b.bounds_blocks < 1 || b.bounds_blocks > BOUNDS_BLOCKS) b.bounds_blocks = BOUNDS_BLOCKS;
  b.bounds_unroll = t.bounds_unroll == 4 ? 4 : 8;
  b.finalize_threads = t.finalize_threads == 64 ? 64 : 256;
  b.fused_sort = t.fused_sort != 0 ? 1 : 0;
  return b;
}

int bounds_rows(size_t n) {
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)build_tuning().bounds_blocks) blocks = build_tuning().bounds_blocks;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

int sort_passes_for_cells(long long ncells) {
  int bits = 1;
  while (bits < 32 && (1ull << bits) <= (unsigned long long)ncells) ++bits;
  return (bits + 7) / 8;
}

void launch_bounds_geometry(const float* x, const float* y, const float* z, size_t n, float leaf, float inv_leaf,
                            long long cell_capacity, int planned_passes, int* rows, unsigned int* ticket,
                            BuildGeom* gd, BuildGeom* gd_host, const LeafStats* old_stats, int dirty_slots,
                            int* cell2leaf, size_t c2l_cap, int* d_nleaf, hipStream_t s) {
  if (build_tuning().bounds_unroll == 8)
    hipLaunchKernelGGL(k_bounds<8>, dim3((unsigned)bounds_rows(n)), dim3(256), 0, s, x, y, z, n, rows, ticket, leaf,
                       inv_leaf, cell_capacity, planned_passes, gd, gd_host, old_stats, dirty_slots, cell2leaf, c2l_cap,
                       d_nleaf);
  else
    hipLaunchKernelGGL(k_bounds<4>, dim3((unsigned)bounds_rows(n)), dim3(256), 0, s, x, y, z, n, rows, ticket, leaf,
                       inv_leaf, cell_capacity, planned_passes, gd, gd_host, old_stats, dirty_slots, cell2leaf, c2l_cap,
                       d_nleaf);
}

int sort_tiles(size_t n) { return (int)((n + SORT_TILE - 1) / SORT_TILE); }

// scratch of the sort: the bin-major tile histograms + the bin totals
size_t sort_temp_bytes(size_t n) { return ((size_t)SORT_BINS * sort_tiles(n) + SORT_BINS) * sizeof(int); }

void fill_sort_plan(BuildGeom* b, int end_bit) {
  if (end_bit < 1) end_bit = 1;
  if (end_bit > 32) end_bit = 32;
  b->bits = end_bit;
  b->passes = (end_bit + 7) / 8;
  const int base = end_bit / b->passes, rem = end_bit % b->passes;
  int sh = 0;
  for (int i = 0; i < 4; ++i) {
    b->width[i] = i < b->passes ? base + (i < rem ? 1 : 0) : 0;
    b->shift[i] = sh;
    sh += b->width[i];
  }
}

void launch_cell_keys(const float* x, const float* y, const float* z, size_t n, const BuildGeom* gd,
                      uint32_t* keys, float* xyz4, void* sort_temp, hipStream_t s) {
  if (n == 0) return;
  const int ntiles = sort_tiles(n);
  hipLaunchKernelGGL(k_cell_keys, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, x, y, z, (int)n, gd, keys,
                     reinterpret_cast<float4*>(xyz4), ntiles, static_cast<int*>(sort_temp));
}

void launch_sort_first_count(const uint32_t* keys, size_t n, const BuildGeom* gd, void* sort_temp, hipStream_t s) {
  if (n == 0) return;
  const int ntiles = sort_tiles(n);
  hipLaunchKernelGGL(k_sort_count, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, keys, (int)n, 0, gd, ntiles,
                     static_cast<int*>(sort_temp));
}

// Sorts (keys_a, identity) by the key bits of the plan in *gd (device memory), stable, in
// `passes` digit passes.  The first digit's tile histograms must already be in `temp`
// (launch_cell_keys).  The passes ping-pong between the a and b buffers; *result_in_b says
// where the sorted pairs ended up.
hipError_t sort_pairs(void* temp, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b,
                      size_t n, int passes, const BuildGeom* gd, hipStream_t s, bool* result_in_b) {
  *result_in_b = false;
  if (n == 0) return hipSuccess;
  const int ntiles = sort_tiles(n);
  int* hist = static_cast<int*>(temp);
  int* totals = hist + (size_t)SORT_BINS * ntiles;
  uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
  for (int p = 0; p < passes; ++p) {
    if (p > 0)
      hipLaunchKernelGGL(k_sort_count, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, kin, (int)n, p, gd, ntiles, hist);
    hipLaunchKernelGGL(k_sort_scan, dim3(SORT_BINS), dim3(SORT_THREADS), 0, s, hist, ntiles, totals);
    if (p == 0)
      hipLaunchKernelGGL(k_sort_scatter<true>, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, kin, vin, (int)n, p, gd,
                         ntiles, hist, totals, kout, vout);
    else
      hipLaunchKernelGGL(k_sort_scatter<false>, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, s, kin, vin, (int)n, p, gd,
                         ntiles, hist, totals, kout, vout);
    uint32_t* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
  }
  *result_in_b = (passes & 1) != 0;
  return hipGetLastError();
}

// sized for the smaller tile, whatever the tuning
// ---- fused passes (one launch per digit) ----
int fused_tiles(size_t n, int tile) { return (int)((n + tile - 1) / tile); }
// the tile size a cloud gets on a device with `compute_units` CUs (one tile per CU at most): 0 = none fits
int fused_tile_for(size_t n, int compute_units) {
  if (n == 0) return 0;
  const int cap = compute_units < FUSED_MAX_TILES ? compute_units : FUSED_MAX_TILES;
  if (fused_tiles(n, FUSED_TILE) <= cap) return FUSED_TILE;
  if (fused_tiles(n, FUSED_BIG_TILE) <= cap) return FUSED_BIG_TILE;
  return 0;
}
bool fused_build_enabled() { return build_tuning().fused_sort != 0; }
// every block of a fused pass waits for all the others: one tile per compute unit at most (a
// partitioned device -- CPX mode: 32 CUs -- takes the classic passes for anything above 256 k points)
bool fused_sort_fits(size_t n, int compute_units) {
  return fused_tile_for(n, compute_units) != 0;
}
size_t fused_table_words() { return (size_t)FUSED_MAX_TILES * SORT_BINS; }

// Cell keys + stable sort by cell in `passes` launches, straight from the cloud.  `tile`: fused_tile_for();
// `table` (fused_table_words() words, zeroed at allocation and whenever *seq wraps) carries the tagged
// tile counts; *seq is the engine's launch tag counter.  The result lands in (keys_b, vals_b) for an
// odd number of passes, (keys_a, vals_a) otherwise.
template <int ROUNDS>
static void launch_sort_pass(bool from_points, int ntiles, hipStream_t s, const float* x, const float* y, const float* z,
                             float* xyz4, const uint32_t* kin, const uint32_t* vin, int n, int p, BuildGeom* gd,
                             BuildGeom* gd_host, uint32_t* table, uint32_t tag, int mute_tile, uint32_t* kout,
                             uint32_t* vout) {
  if (from_points)
    hipLaunchKernelGGL((k_sort_pass<true, ROUNDS>), dim3((unsigned)ntiles), dim3(FUSED_THREADS), 0, s, x, y, z,
                       reinterpret_cast<float4*>(xyz4), (const uint32_t*)nullptr, (const uint32_t*)nullptr, n, p, gd,
                       gd_host, ntiles, table, tag, mute_tile, kout, vout);
  else
    hipLaunchKernelGGL((k_sort_pass<false, ROUNDS>), dim3((unsigned)ntiles), dim3(FUSED_THREADS), 0, s,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (float4*)nullptr, kin, vin, n,
                       p, gd, gd_host, ntiles, table, tag, mute_tile, kout, vout);
}

hipError_t sort_cloud_fused(const float* x, const float* y, const float* z, size_t n, int tile, BuildGeom* gd,
                            BuildGeom* gd_host, float* xyz4, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                            uint32_t* vals_b, int passes, uint32_t* table, uint32_t* seq, hipStream_t s,
                            bool* result_in_b) {
  *result_in_b = false;
  if (n == 0) return hipSuccess;
  if (tile != FUSED_TILE && tile != FUSED_BIG_TILE) return hipErrorInvalidValue;
  const int ntiles = fused_tiles(n, tile);
#ifdef NDT_TEST_SEAMS
  // test seam (libndt_hip_seams.so only): NDT_DEBUG_FUSED_MUTE_TILE=<t> makes tile t withhold its counts, so every block times out
  static const int mute_tile = [] { const char* e = getenv("NDT_DEBUG_FUSED_MUTE_TILE"); return e && *e ? atoi(e) : -1; }();
#else
  constexpr int mute_tile = -1;
#endif
  uint32_t *kin = keys_a, *kout = keys_b, *vin = vals_a, *vout = vals_b;
  for (int p = 0; p < passes; ++p) {
    uint32_t tag = (*seq + 1u) & 0xffffu;
    if (tag == 0u) {  // wrapped: forget every old tag before tag 1 is handed out again
      hipError_t e = hipMemsetAsync(table, 0, fused_table_words() * sizeof(uint32_t), s);
      if (e != hipSuccess) return e;
      tag = 1u;
    }
    *seq = tag;
    // (pass 0 reads the cloud and writes into the "b" buffers, so that the ping-pong below stays uniform)
    if (tile == FUSED_TILE)
      launch_sort_pass<SORT_ROUNDS>(p == 0, ntiles, s, x, y, z, xyz4, kin, vin, (int)n, p, gd, gd_host, table, tag, mute_tile, kout, vout);
    else
      launch_sort_pass<FUSED_BIG_ROUNDS>(p == 0, ntiles, s, x, y, z, xyz4, kin, vin, (int)n, p, gd, gd_host, table, tag, mute_tile, kout, vout);
    uint32_t* t = kin; kin = kout; kout = t;
    t = vin; vin = vout; vout = t;
  }
  *result_in_b = (passes & 1) != 0;
  return hipGetLastError();
}

// ---- bucketed build (two launches) ----
bool bucket_build_enabled() { return tuning().bucket_build != 0; }
// points per thread of the partition launch: the smallest tile of >= 2048 points that keeps the launch at <= BK_MAX_TILES tiles
// (a 131 k-point scan: 64 tiles of 2048 points, the 1 M-point map: 245 tiles of 4096); ndt_tuning::bucket_tile forces a size that fits
static int bucket_rounds_for(size_t n) {
  const int forced = tuning().bucket_tile;
  if (forced != 0 && (n + (size_t)forced - 1) / (size_t)forced <= (size_t)BK_MAX_TILES) return forced / BK_THREADS;
  // (1024-point tiles are compiled and measured: C2's 131 k points, 128 tiles, 40.4 us against 38.8 with 64 tiles of 2048 --
  // a bucket's segments of 4 points are half a cache line)
  for (int r = 2; r < BK_ROUNDS; r *= 2)
    if ((n + (size_t)(BK_THREADS * r) - 1) / (size_t)(BK_THREADS * r) <= (size_t)BK_MAX_TILES) return r;
  return BK_ROUNDS;
}
int bucket_build_tiles(size_t n) {
  const size_t tile = (size_t)BK_THREADS * bucket_rounds_for(n);
  return (int)((n + tile - 1) / tile);
}
size_t bucket_table_words() { return (size_t)BK_BUCKETS * BK_MAX_TILES; }
void bucket_bounds_neutral(int out[8]) {
  for (int k = 0; k < 3; ++k) { out[k] = INT_MAX; out[3 + k] = INT_MIN; }
  out[6] = 0;
  out[7] = 0;
}
// The hash spreads voxels, not points: the largest bucket of a lidar map is ~1.6x the mean (crowded voxels
// near the sensor), and a bucket must fit BK_MAXP points -- so clouds up to 256 * BK_MAXP / 1.6 points are
// worth trying (a bucket that overflows anyway costs one declined launch pair, BG_BUCKET).  Neither launch waits
// for sibling blocks (round 5), so the device's size does not matter.
bool bucket_build_fits(size_t n, int compute_units) {
  (void)compute_units;
  return n != 0 && n <= (size_t)BK_BUCKETS * BK_MAXP * 5 / 8;
}

template <int ROUNDS>
static void launch_bucket_pass(int tile0, int ntiles, hipStream_t s, const float* x, const float* y, const float* z, int n, float inv_leaf,
                               uint32_t* tab, const LeafStats* old_stats, int dirty_slots, int* cell2leaf, size_t c2l_cap,
                               int* bnd, int* d_nleaf, float4* pts4) {
  hipLaunchKernelGGL(k_bucket_pass<ROUNDS>, dim3((unsigned)ntiles), dim3(BK_THREADS), 0, s, x, y, z, n, inv_leaf, tile0, tab, old_stats,
                     dirty_slots, cell2leaf, c2l_cap, bnd, d_nleaf, pts4);
}

size_t bucket_tile_points(size_t n) { return (size_t)BK_THREADS * bucket_rounds_for(n); }

hipError_t launch_bucket_pass_tiles(const float* x, const float* y, const float* z, size_t n, float inv_leaf, uint32_t* tab,
                                    const LeafStats* old_stats, int dirty_slots, int* cell2leaf, size_t c2l_cap, int* bnd,
                                    int* d_nleaf, float* pts4, int tile_first, int tile_end, hipStream_t s) {
  if (n == 0 || !bucket_build_fits(n, 0) || tile_first < 0 || tile_end > bucket_build_tiles(n) || tile_end > BK_MAX_TILES)
    return hipErrorInvalidValue;
  if (tile_end <= tile_first) return hipSuccess;
  float4* p4 = reinterpret_cast<float4*>(pts4);
  const int nt = tile_end - tile_first;
  const int dirty = tile_first == 0 ? dirty_slots : 0;   // the launch that holds tile 0 resets the previous build's cells
  switch (bucket_rounds_for(n)) {
    case 1: launch_bucket_pass<1>(tile_first, nt, s, x, y, z, (int)n, inv_leaf, tab, old_stats, dirty, cell2leaf, c2l_cap, bnd, d_nleaf, p4); break;
    case 2: launch_bucket_pass<2>(tile_first, nt, s, x, y, z, (int)n, inv_leaf, tab, old_stats, dirty, cell2leaf, c2l_cap, bnd, d_nleaf, p4); break;
    case 4: launch_bucket_pass<4>(tile_first, nt, s, x, y, z, (int)n, inv_leaf, tab, old_stats, dirty, cell2leaf, c2l_cap, bnd, d_nleaf, p4); break;
    default: launch_bucket_pass<8>(tile_first, nt, s, x, y, z, (int)n, inv_leaf, tab, old_stats, dirty, cell2leaf, c2l_cap, bnd, d_nleaf, p4); break;
  }
  return hipGetLastError();
}

hipError_t launch_bucket_leaves(size_t n, float leaf, float inv_leaf, long long cell_capacity, int min_pts, FinalizeParams fp,
                                BuildGeom* gd, BuildGeom* gd_host, const uint32_t* tab, int* cell2leaf, int* bnd, int* d_nleaf,
                                unsigned int* ticket, const float* pts4, double* sums, VoxelRecord* rec, float* cent4,
                                LeafStats* stats, int max_leaves, int* nleaf_host, int done_tag, hipStream_t s) {
  if (n == 0 || !bucket_build_fits(n, 0)) return hipErrorInvalidValue;
  const int ntiles = bucket_build_tiles(n);
  if (ntiles > BK_MAX_TILES) return hipErrorInvalidValue;
  int tile_shift = 10;
  while (((size_t)1 << tile_shift) < bucket_tile_points(n)) ++tile_shift;
  hipLaunchKernelGGL(k_bucket_leaves, dim3(BK_BUCKETS), dim3(BK_THREADS), 0, s, reinterpret_cast<const float4*>(pts4),
                     tab, ntiles, tile_shift, bnd, leaf, inv_leaf, cell_capacity, min_pts, fp, gd, gd_host, d_nleaf,
                     reinterpret_cast<unsigned long long*>(ticket) /* 8-byte aligned, zero between builds */, sums,
                     rec, reinterpret_cast<float4*>(cent4), stats, cell2leaf, max_leaves, nleaf_host, done_tag);
  return hipGetLastError();
}

hipError_t launch_bucket_build(const float* x, const float* y, const float* z, size_t n, float leaf, float inv_leaf,
                               long long cell_capacity, int min_pts, FinalizeParams fp, BuildGeom* gd, BuildGeom* gd_host,
                               uint32_t* tab, const LeafStats* old_stats, int dirty_slots, int* cell2leaf,
                               size_t c2l_cap, int* bnd, int* d_nleaf, unsigned int* ticket, float* pts4,
                               double* sums, VoxelRecord* rec, float* cent4, LeafStats* stats, int max_leaves, int* nleaf_host,
                               int done_tag, hipStream_t s) {
  hipError_t e = launch_bucket_pass_tiles(x, y, z, n, inv_leaf, tab, old_stats, dirty_slots, cell2leaf, c2l_cap, bnd, d_nleaf, pts4,
                                          0, bucket_build_tiles(n), s);
  if (e != hipSuccess) return e;
  return launch_bucket_leaves(n, leaf, inv_leaf, cell_capacity, min_pts, fp, gd, gd_host, tab, cell2leaf, bnd, d_nleaf, ticket, pts4,
                              sums, rec, cent4, stats, max_leaves, nleaf_host, done_tag, s);
}

int runs_blocks(size_t n) { return (int)((n + 256 * 8 - 1) / (256 * 8)); }

size_t run_tag_words(size_t n) { return (size_t)runs_blocks(n); }

// run_tags != nullptr: count and emit in one launch (k_runs<RUNS_FUSED>); *seq is the tag counter of
// that buffer (run_tag_words(n) words, zero at allocation)
hipError_t launch_find_runs(const uint32_t* keys_sorted, size_t n, BuildGeom* gd, BuildGeom* gd_host, int min_pts,
                            int* d_nleaf, int* block_counts, int* block_offsets, unsigned int* ticket, uint32_t* run_tags,
                            size_t run_tags_cap, uint32_t* seq, int* leaf_start, int* leaf_cnt, hipStream_t s) {
  if (n == 0) return hipSuccess;
  uint32_t tag = 0;
  if (run_tags) {
    tag = (*seq + 1u) & 0xffffu;
    if (tag == 0u) {
      hipError_t e = hipMemsetAsync(run_tags, 0, run_tags_cap * sizeof(uint32_t), s);
      if (e != hipSuccess) return e;
      tag = 1u;
    }
    *seq = tag;
  }
  const int blocks = runs_blocks(n);
  if (run_tags) {
    hipLaunchKernelGGL((k_runs<RUNS_FUSED, 8>), dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, gd, gd_host, min_pts,
                       block_counts, block_offsets, ticket, run_tags, tag, d_nleaf, leaf_start, leaf_cnt);
  } else {
    hipLaunchKernelGGL((k_runs<RUNS_COUNT, 8>), dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, gd, gd_host, min_pts,
                       block_counts, block_offsets, ticket, run_tags, tag, d_nleaf, leaf_start, leaf_cnt);
    hipLaunchKernelGGL((k_runs<RUNS_EMIT, 8>), dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, gd, gd_host, min_pts,
                       block_counts, block_offsets, ticket, run_tags, tag, d_nleaf, leaf_start, leaf_cnt);
  }
  return hipGetLastError();
}

int build_read_stamps(unsigned long long* out) {
#ifdef NDT_STAMPS
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(unsigned long long) * 6 * 512 * 8) == hipSuccess ? 6 * 512 : -1;
#else
  (void)out;
  return 0;
#endif
}

void launch_pull_chunk(const float* stage_dev, size_t len, size_t seg, float* x, float* y, float* z, hipStream_t s) {
  if (len == 0) return;
  const unsigned blocks = (unsigned)std::min<size_t>(64, (len / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(k_pull_chunk, dim3(blocks), dim3(256), 0, s, stage_dev, len, seg, x, y, z);
}

void launch_voxel_centroids(const float* xyz4, const float* intensity, const uint32_t* vals_sorted, const int* d_nleaf,
                            const int* leaf_start, const int* leaf_cnt, size_t max_runs, size_t cap, float* ox, float* oy,
                            float* oz, float* oi, hipStream_t s) {
  const size_t m = std::min(max_runs, cap);
  if (m == 0) return;
  hipLaunchKernelGGL(k_voxel_centroids, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(xyz4),
                     intensity, vals_sorted, d_nleaf, leaf_start, leaf_cnt, (int)std::min<size_t>(cap, 0x7fffffff), ox, oy, oz, oi);
}

void launch_scatter_heads(const int* cells, const int* slots, size_t n, int* cell2leaf, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_scatter_heads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, cells, slots, (int)n, cell2leaf);
}

int finalize_blocks(int max_leaves) { return (max_leaves + 63) / 64; }

void launch_finalize_leaves(const float* xyz4, const uint32_t* keys_sorted, const uint32_t* vals_sorted,
                            int* d_nleaf, const int* leaf_start, const int* leaf_cnt, int max_leaves,
                            FinalizeParams fp, double* sums, VoxelRecord* rec, float* cent4, LeafStats* stats, int* cell2leaf,
                            int* block_ok, unsigned int* ticket, int* nleaf_host, int done_tag, hipStream_t s) {
  if (max_leaves <= 0) return;
  size_t blocks = ((size_t)max_leaves * LANES_PER_LEAF + 255) / 256;
  if (blocks > (size_t)SUMS_BLOCKS_MAX) blocks = SUMS_BLOCKS_MAX;
  hipLaunchKernelGGL(k_leaf_sums, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(xyz4),
                     vals_sorted, d_nleaf, leaf_start, leaf_cnt, sums);
  if (build_tuning().finalize_threads == 256)
    hipLaunchKernelGGL(k_leaf_finalize<256>, dim3((unsigned)((max_leaves + 255) / 256)), dim3(256), 0, s, keys_sorted,
                       d_nleaf, leaf_start, leaf_cnt, sums, fp, rec, reinterpret_cast<float4*>(cent4), stats, cell2leaf, block_ok, ticket, nleaf_host, done_tag);
  else
    hipLaunchKernelGGL(k_leaf_finalize<64>, dim3((unsigned)((max_leaves + 63) / 64)), dim3(64), 0, s, keys_sorted,
                       d_nleaf, leaf_start, leaf_cnt, sums, fp, rec, reinterpret_cast<float4*>(cent4), stats, cell2leaf, block_ok, ticket, nleaf_host, done_tag);
}

}  // namespace ndt
