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

#include <climits>
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
  // four strided points per trip: 12 loads in flight instead of 3 (the pass is latency-bound)
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    float a[4], b[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t i = i0 + u * stride;
      const size_t j = i < n ? i : i0;
      a[u] = x[j]; b[u] = y[j]; c[u] = z[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
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

// Runs of equal cell key in the sorted array.  A thread owns RUN_KEYS consecutive keys (two
// 16-byte loads), a block one 2048-key tile.  Every run TAIL needs its head: the thread's
// own last head, else the last head of a lower lane (wave max-scan), else of a lower wave
// (LDS), and only for a run that began before the tile a backward gallop + bisection in
// global memory.  Runs with at least min_pts points become leaves
// (ref: voxel_grid_covariance_impl.hpp:270-273).  Leaf slots are handed out by a
// count / scan / emit triple instead of a global atomic counter (one contended address
// served ~90 adds/us and cost 0.1 ms): slots come out in ascending cell order, identically
// on every run.
constexpr int RUN_KEYS = 8;
constexpr int RUN_TILE = 256 * RUN_KEYS;

template <bool EMIT>
__global__ void __launch_bounds__(256) k_runs(const uint32_t* __restrict__ keys, int n,
                                             const BuildGeom* __restrict__ gd, int min_pts,
                                             int* __restrict__ block_counts, int* __restrict__ block_offsets,
                                             unsigned int* __restrict__ ticket, int* __restrict__ nleaf_out,
                                             int* __restrict__ leaf_start, int* __restrict__ leaf_cnt) {
  __shared__ int wave_head[4];
  __shared__ int wave_total[4];
  __shared__ int s_last;
  if (gd->status != BG_OK) return;
  const int ncells = gd->g.ncells;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * RUN_TILE + threadIdx.x * RUN_KEYS;
  const uint32_t sentinel = 0xFFFFFFFFu;
  uint32_t k[RUN_KEYS];
  if (s0 + RUN_KEYS <= n) {  // keys is 16-byte aligned and s0 a multiple of 8
    const uint4 a = *reinterpret_cast<const uint4*>(keys + s0);
    const uint4 b = *reinterpret_cast<const uint4*>(keys + s0 + 4);
    k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < RUN_KEYS; ++j) k[j] = s0 + j < n ? keys[s0 + j] : sentinel;
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
      if (start < 0) {
        // the run began before this tile: keys[tile_base] == key; walk back
        int hi = blockIdx.x * RUN_TILE;  // known to hold key
        int step = 1, lo;
        for (;;) {
          int nx = hi - step;
          if (nx < 0) { lo = -1; break; }
          if (keys[nx] != k[j]) { lo = nx; break; }
          hi = nx;
          step <<= 1;
        }
        while (hi - lo > 1) {  // keys[lo] != key (or lo == -1), keys[hi] == key
          int mid = (lo + hi) >> 1;
          if (keys[mid] == k[j]) hi = mid; else lo = mid;
        }
        start = hi;
      }
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
  if (nleaf == 0) return;
  int slot = block_offsets[blockIdx.x] + lincl - nleaf;
  for (int w = 0; w < wave; ++w) slot += wave_total[w];
#pragma unroll
  for (int j = 0; j < RUN_KEYS; ++j) {
    if (counts[j] > 0) {
      leaf_start[slot] = starts[j];
      leaf_cnt[slot] = counts[j];
      ++slot;
    }
  }
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
constexpr int LEAF_HEAD = 64;
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

__device__ __forceinline__ void finalize_one(int slot, const uint32_t* __restrict__ keys, int* __restrict__ nleaf_p,
                                             const int* __restrict__ leaf_start, const int* __restrict__ leaf_cnt,
                                             const double* __restrict__ sums, FinalizeParams fp,
                                             VoxelRecord* __restrict__ rec, LeafStats* __restrict__ stats,
                                             int* __restrict__ cell2leaf);

// one-wave blocks: every lane's stores / atomics are acknowledged, then lane 0 takes the ticket
__device__ __forceinline__ bool last_ticket_wave(unsigned int* ticket, unsigned int nblocks) {
  int last = 0;
  if ((threadIdx.x & 63) == 0) last = last_ticket(ticket, nblocks) ? 1 : 0;  // drains vmcnt first
  else __builtin_amdgcn_s_waitcnt(0);
  return __builtin_amdgcn_readfirstlane(last) != 0;
}

// ref: voxel_grid_covariance_impl.hpp:265-343 -- one thread per leaf: mean, covariance,
// eigen-decomposition, eigenvalue inflation, inverse, validity checks.  (Fusing this into the
// 8 lanes that sum a leaf was tried in round 2: 48 us against 20 + 12 -- the Jacobi state on top
// of the moments spills, and 8 of 64 lanes do distinct work.)
__global__ void __launch_bounds__(64) k_leaf_finalize(const uint32_t* __restrict__ keys,
                                                      int* __restrict__ nleaf_p,
                                                      const int* __restrict__ leaf_start,
                                                      const int* __restrict__ leaf_cnt,
                                                      const double* __restrict__ sums, FinalizeParams fp,
                                                      VoxelRecord* __restrict__ rec, LeafStats* __restrict__ stats,
                                                      int* __restrict__ cell2leaf, unsigned int* __restrict__ ticket,
                                                      int* __restrict__ nleaf_host) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  const int nl = nleaf_p[0];
  // the grid is sized for the worst case (n / min_points leaves); only the blocks that hold a
  // leaf -- block 0 always -- take part in the ticket (one contended address serves ~90 adds/us)
  const int live_blocks = max(1, (nl + (int)blockDim.x - 1) / (int)blockDim.x);
  if ((int)blockIdx.x >= live_blocks) return;
  if (slot < nl) finalize_one(slot, keys, nleaf_p, leaf_start, leaf_cnt, sums, fp, rec, stats, cell2leaf);
  // the block (one wave) that draws the last ticket hands the two leaf counters to the host
  // through pinned memory: the D2H copy they used to take was a 4.4 us launch of its own
  if (last_ticket_wave(ticket, (unsigned int)live_blocks)) {
    if (threadIdx.x < 2) nleaf_host[threadIdx.x] = ld_agent(nleaf_p + threadIdx.x);
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ void finalize_one(int slot, const uint32_t* __restrict__ keys, int* __restrict__ nleaf_p,
                                             const int* __restrict__ leaf_start, const int* __restrict__ leaf_cnt,
                                             const double* __restrict__ sums, FinalizeParams fp,
                                             VoxelRecord* __restrict__ rec, LeafStats* __restrict__ stats,
                                             int* __restrict__ cell2leaf) {
  const int start = leaf_start[slot], cnt = leaf_cnt[slot];
  const double* in = sums + (size_t)slot * 9;
  const double s[3] = {in[0], in[1], in[2]};
  const double ss[6] = {in[3], in[4], in[5], in[6], in[7], in[8]};
  const int cell = (int)keys[start];
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
  if (ok) {
    cell2leaf[cell] = slot;
    atomicAdd(nleaf_p + 1, 1);  // leaves that passed every check
  }
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

// three SoA arrays in one launch (three hipMemcpyAsync cost three dispatches)
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

int bounds_rows(size_t n) {
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)BOUNDS_BLOCKS) blocks = BOUNDS_BLOCKS;
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
  hipLaunchKernelGGL(k_bounds, dim3((unsigned)bounds_rows(n)), dim3(256), 0, s, x, y, z, n, rows, ticket, leaf, inv_leaf,
                     cell_capacity, planned_passes, gd, gd_host, old_stats, dirty_slots, cell2leaf, c2l_cap, d_nleaf);
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

int runs_blocks(size_t n) { return (int)((n + RUN_TILE - 1) / RUN_TILE); }

void launch_find_runs(const uint32_t* keys_sorted, size_t n, const BuildGeom* gd, int min_pts, int* d_nleaf,
                      int* block_counts, int* block_offsets, unsigned int* ticket, int* leaf_start, int* leaf_cnt,
                      hipStream_t s) {
  if (n == 0) return;
  const int blocks = runs_blocks(n);
  hipLaunchKernelGGL(k_runs<false>, dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, gd, min_pts, block_counts,
                     block_offsets, ticket, d_nleaf, leaf_start, leaf_cnt);
  hipLaunchKernelGGL(k_runs<true>, dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, gd, min_pts, block_counts,
                     block_offsets, ticket, d_nleaf, leaf_start, leaf_cnt);
}

void launch_finalize_leaves(const float* xyz4, const uint32_t* keys_sorted, const uint32_t* vals_sorted,
                            int* d_nleaf, const int* leaf_start, const int* leaf_cnt, int max_leaves,
                            FinalizeParams fp, double* sums, VoxelRecord* rec, LeafStats* stats, int* cell2leaf,
                            unsigned int* ticket, int* nleaf_host, hipStream_t s) {
  if (max_leaves <= 0) return;
  size_t blocks = ((size_t)max_leaves * LANES_PER_LEAF + 255) / 256;
  if (blocks > (size_t)SUMS_BLOCKS_MAX) blocks = SUMS_BLOCKS_MAX;
  hipLaunchKernelGGL(k_leaf_sums, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(xyz4),
                     vals_sorted, d_nleaf, leaf_start, leaf_cnt, sums);
  hipLaunchKernelGGL(k_leaf_finalize, dim3((unsigned)((max_leaves + 63) / 64)), dim3(64), 0, s, keys_sorted, d_nleaf,
                     leaf_start, leaf_cnt, sums, fp, rec, stats, cell2leaf, ticket, nleaf_host);
}

}  // namespace ndt
