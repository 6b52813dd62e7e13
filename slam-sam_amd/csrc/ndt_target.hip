// ndt_target.hip -- target voxel-grid build on gfx950 (MI355X).
//
// What it computes is the reference's VoxelGridCovariance::applyFilter
// (ref: extern/svn_ndt/include/voxel_grid_covariance_impl.hpp:77-379): bounds,
// integer grid, per-voxel point count / mean / 3x3 covariance, eigenvalue
// inflation, inverse covariance, validity filtering.  How it computes it is
// MI355X-first and shares nothing with the reference's single-threaded
// hash-map loop:
//   1. bounds        : one streaming pass, wave-shuffle min/max, 6 int atomics/block
//   2. cell keys     : one streaming pass (f32 floor, bit-compatible with the ref)
//   3. stable LSD radix sort of (cell, point index) -- points of one voxel become
//                      contiguous and stay in input order (deterministic sums)
//   4. run detection : run tails find their head through a wave ballot; leaf slots by
//                      count / scan / emit (ascending cell order, no atomics)
//   5. leaf sums     : 8 lanes per voxel gather + reduce sum(x), sum(x x^T) in f64
//   6. leaf finalise : one thread per voxel: 3x3 Jacobi eigen-solve / inflation /
//                      inverse; publishes an 80-byte VoxelRecord and the dense
//                      cell -> leaf index.
// Compiled with -ffp-contract=off: f32 index arithmetic must round as written.
#include "ndt_kernels.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <climits>

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

// ref: pcl::getMinMax3D at voxel_grid_covariance_impl.hpp:103 (non-finite skipped).
// One row of 8 ints per block {min xyz, max xyz, #finite, 0} written straight into
// device-mapped pinned host memory; the host (which needs the bounds anyway to size
// the grid) folds the <= 512 rows.  No atomics: every block contending on the same
// 7 words cost 0.65 ms for 1M points.
__global__ void __launch_bounds__(256) k_bounds(const float* __restrict__ x, const float* __restrict__ y,
                                               const float* __restrict__ z, size_t n, int* __restrict__ rows) {
  __shared__ int lds[4][8];
  int mn[3] = {INT_MAX, INT_MAX, INT_MAX};
  int mx[3] = {INT_MIN, INT_MIN, INT_MIN};
  int cnt = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float a = x[i], b = y[i], c = z[i];
    if (!finite3(a, b, c)) continue;
    int ea = encode_ordered(a), eb = encode_ordered(b), ec = encode_ordered(c);
    mn[0] = min(mn[0], ea); mx[0] = max(mx[0], ea);
    mn[1] = min(mn[1], eb); mx[1] = max(mx[1], eb);
    mn[2] = min(mn[2], ec); mx[2] = max(mx[2], ec);
    ++cnt;
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
  if (threadIdx.x < 8) {
    const int t = threadIdx.x;
    int v = lds[0][t];
    for (int w = 1; w < 4; ++w) {
      int o = lds[w][t];
      v = t < 3 ? min(v, o) : (t < 6 ? max(v, o) : v + o);
    }
    rows[blockIdx.x * 8 + t] = t == 7 ? 0 : v;
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

__global__ void __launch_bounds__(256) k_cell_keys(const float* __restrict__ x, const float* __restrict__ y,
                                                  const float* __restrict__ z, size_t n, GridGeom g,
                                                  uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                  float4* __restrict__ xyz4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y[i], c = z[i];
  // packed copy for the per-voxel gather: one 16-byte line per point instead of three
  xyz4[i] = make_float4(a, b, c, 0.0f);
  uint32_t key = (uint32_t)g.ncells;  // sentinel sorts behind every real cell
  if (finite3(a, b, c)) {
    int idx = cell_of(a, b, c, g);
    if (idx >= 0 && idx < g.ncells) key = (uint32_t)idx;
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// Runs of equal cell key in the sorted array.  Each run TAIL finds its head: inside
// the wave through a ballot of head flags (no memory traffic), and only for runs
// that began in an earlier wave by a backward gallop + bisection.  Runs with at
// least min_pts points become leaves (ref: voxel_grid_covariance_impl.hpp:270-273).
// Leaf slots are handed out by a count / scan / emit triple instead of a global
// atomic counter (one contended address served ~90 adds/us and cost 0.1 ms): slots
// come out in ascending cell order, identically on every run.
template <bool EMIT>
__global__ void __launch_bounds__(256) k_runs(const uint32_t* __restrict__ keys, int n, int ncells, int min_pts,
                                             int* __restrict__ block_counts,
                                             const int* __restrict__ block_offsets,
                                             int* __restrict__ leaf_start, int* __restrict__ leaf_cnt) {
  __shared__ int wave_total[4];
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sentinel = 0xFFFFFFFFu;
  const uint32_t key = s < n ? keys[s] : sentinel;
  uint32_t prev = __shfl_up(key, 1), next = __shfl_down(key, 1);
  if (lane == 0) prev = s > 0 && s - 1 < n ? keys[s - 1] : sentinel;
  if (lane == 63) next = s + 1 < n ? keys[s + 1] : sentinel;
  const bool valid = s < n && key < (uint32_t)ncells;
  const bool head = valid && (s == 0 || prev != key);
  const bool tail = valid && (s == n - 1 || next != key);
  const unsigned long long heads = __ballot(head);
  int start = s;
  if (tail) {
    const unsigned long long below = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    if (below) {
      start = s - (lane - (63 - __clzll((long long)below)));
    } else {
      // the run began before this wave: keys[wave_base] == key; walk back
      int hi = s - lane;  // known to hold key
      int step = 1, lo;
      for (;;) {
        int nx = hi - step;
        if (nx < 0) { lo = -1; break; }
        if (keys[nx] != key) { lo = nx; break; }
        hi = nx;
        step <<= 1;
      }
      while (hi - lo > 1) {  // keys[lo] != key (or lo == -1), keys[hi] == key
        int mid = (lo + hi) >> 1;
        if (keys[mid] == key) hi = mid; else lo = mid;
      }
      start = hi;
    }
  }
  const int cnt = s - start + 1;
  const bool leaf = tail && cnt >= min_pts;
  const unsigned long long leaves = __ballot(leaf);
  if (lane == 0) wave_total[wave] = __popcll(leaves);
  __syncthreads();
  if (!EMIT) {
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    return;
  }
  if (!leaf) return;
  int slot = block_offsets[blockIdx.x] + __popcll(leaves & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) slot += wave_total[w];
  leaf_start[slot] = start;
  leaf_cnt[slot] = cnt;
}

// exclusive scan of the per-block leaf counts (one block; a few thousand values)
__global__ void __launch_bounds__(1024) k_scan_counts(const int* __restrict__ counts, int nblocks,
                                                     int* __restrict__ offsets, int* __restrict__ total) {
  __shared__ int wsum[16];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < nblocks ? counts[i] : 0;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int t = __shfl_up(incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = carry;
    for (int w = 0; w < wave; ++w) before += wsum[w];
    if (i < nblocks) offsets[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[0] = carry;
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
// 8 lanes per leaf gather its points (stable sort => ascending input order) and a fixed
// 3-step xor tree adds the 8 partial sums: deterministic.
__global__ void __launch_bounds__(256) k_leaf_sums(const float4* __restrict__ xyz4, const uint32_t* __restrict__ vals,
                                                  const int* __restrict__ nleaf_p,
                                                  const int* __restrict__ leaf_start,
                                                  const int* __restrict__ leaf_cnt, double* __restrict__ sums) {
  const int nleaf = nleaf_p[0];
  const int sub = threadIdx.x & (LANES_PER_LEAF - 1);
  const int per_block = 256 / LANES_PER_LEAF;
  for (int slot = blockIdx.x * per_block + threadIdx.x / LANES_PER_LEAF; slot < nleaf;
       slot += gridDim.x * per_block) {
    const int start = leaf_start[slot], cnt = leaf_cnt[slot];
    double s[3] = {0, 0, 0}, ss[6] = {0, 0, 0, 0, 0, 0};
    // four gathers in flight per lane (index load -> point load is a dependent pair);
    // the adds stay in point order, masked lanes add exact zeros
    for (int j0 = sub; j0 < cnt; j0 += 4 * LANES_PER_LEAF) {
      float4 p[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * LANES_PER_LEAF;
        live[u] = j < cnt;
        p[u] = xyz4[vals[start + (live[u] ? j : 0)]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = live[u] ? (double)p[u].x : 0.0, b = live[u] ? (double)p[u].y : 0.0,
                     c = live[u] ? (double)p[u].z : 0.0;
        s[0] += a; s[1] += b; s[2] += c;
        ss[0] += a * a; ss[1] += a * b; ss[2] += a * c;
        ss[3] += b * b; ss[4] += b * c; ss[5] += c * c;
      }
    }
#pragma unroll
    for (int off = 1; off < LANES_PER_LEAF; off <<= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) s[a] += __shfl_xor(s[a], off);
#pragma unroll
      for (int a = 0; a < 6; ++a) ss[a] += __shfl_xor(ss[a], off);
    }
    // the 8 lanes hold identical sums; lane k writes word k, lane 0 also word 8
    double* o = sums + (size_t)slot * 9;
    const double mine = sub == 0 ? s[0] : sub == 1 ? s[1] : sub == 2 ? s[2] : sub == 3 ? ss[0]
                      : sub == 4 ? ss[1] : sub == 5 ? ss[2] : sub == 6 ? ss[3] : ss[4];
    o[sub] = mine;
    if (sub == 0) o[8] = ss[5];
  }
}

// ref: voxel_grid_covariance_impl.hpp:265-343 -- one thread per leaf: mean, covariance,
// eigen-decomposition, eigenvalue inflation, inverse, validity checks.
__global__ void __launch_bounds__(256) k_leaf_finalize(const uint32_t* __restrict__ keys,
                                                      int* __restrict__ nleaf_p,
                                                      const int* __restrict__ leaf_start,
                                                      const int* __restrict__ leaf_cnt,
                                                      const double* __restrict__ sums, FinalizeParams fp,
                                                      VoxelRecord* __restrict__ rec, LeafStats* __restrict__ stats,
                                                      int* __restrict__ cell2leaf) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= nleaf_p[0]) return;
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
  if (ok) {
    VoxelRecord r;
    r.mean[0] = mean[0]; r.mean[1] = mean[1]; r.mean[2] = mean[2];
    r.icov[0] = I[0]; r.icov[1] = I[1]; r.icov[2] = I[2];
    r.icov[3] = I[4]; r.icov[4] = I[5]; r.icov[5] = I[8];
    r.pad = (double)cnt;
    rec[slot] = r;
    cell2leaf[cell] = slot;
    atomicAdd(nleaf_p + 1, 1);  // leaves that passed every check
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

float decode_ordered(int enc) {
  int i = enc >= 0 ? enc : enc ^ 0x7fffffff;
  float f;
  memcpy(&f, &i, sizeof(f));
  return f;
}

int bounds_rows(size_t n) {
  size_t blocks = (n + 255) / 256;
  if (blocks > (size_t)BOUNDS_BLOCKS) blocks = BOUNDS_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

void launch_bounds(const float* x, const float* y, const float* z, size_t n, int* rows, hipStream_t s) {
  hipLaunchKernelGGL(k_bounds, dim3((unsigned)bounds_rows(n)), dim3(256), 0, s, x, y, z, n, rows);
}

void fold_bounds(const int* rows, int nrows, int out[8]) {
  for (int a = 0; a < 3; ++a) { out[a] = INT_MAX; out[3 + a] = INT_MIN; }
  out[6] = out[7] = 0;
  for (int r = 0; r < nrows; ++r) {
    const int* p = rows + 8 * r;
    for (int a = 0; a < 3; ++a) {
      out[a] = p[a] < out[a] ? p[a] : out[a];
      out[3 + a] = p[3 + a] > out[3 + a] ? p[3 + a] : out[3 + a];
    }
    out[6] += p[6];
  }
}

void launch_cell_keys(const float* x, const float* y, const float* z, size_t n, const GridGeom& g,
                      uint32_t* keys, uint32_t* vals, float* xyz4, hipStream_t s) {
  if (n == 0) return;
  size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_cell_keys, dim3((unsigned)blocks), dim3(256), 0, s, x, y, z, n, g, keys, vals,
                     reinterpret_cast<float4*>(xyz4));
}

// rocPRIM's default switches to a merge sort below 1M items (10 merge passes, ~165 us for the
// 1M-point map); the onesweep LSD radix sort needs ceil(bits/8) passes over the keys only.
#ifndef NDT_SORT_MERGE_LIMIT
#define NDT_SORT_MERGE_LIMIT 16384
#endif
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                              rocprim::default_config, NDT_SORT_MERGE_LIMIT>;

size_t sort_temp_bytes(size_t n) {
  size_t bytes = 0;
  uint32_t* p = nullptr;
  (void)rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, p, p, p, p, n, 0, 32, (hipStream_t)0);
  return bytes;
}

hipError_t sort_pairs(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                      const uint32_t* vals_in, uint32_t* vals_out, size_t n, int end_bit,
                      hipStream_t s) {
  return rocprim::radix_sort_pairs<SortConfig>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out,
                                               n, 0, (unsigned)end_bit, s);
}

int runs_blocks(size_t n) { return (int)((n + 255) / 256); }

void launch_find_runs(const uint32_t* keys_sorted, size_t n, int ncells, int min_pts, int* d_nleaf,
                      int* block_counts, int* block_offsets, int* leaf_start, int* leaf_cnt,
                      hipStream_t s) {
  if (n == 0) return;
  const int blocks = runs_blocks(n);
  hipLaunchKernelGGL(k_runs<false>, dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, ncells, min_pts,
                     block_counts, (const int*)nullptr, leaf_start, leaf_cnt);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, block_counts, blocks, block_offsets, d_nleaf);
  hipLaunchKernelGGL(k_runs<true>, dim3(blocks), dim3(256), 0, s, keys_sorted, (int)n, ncells, min_pts,
                     block_counts, block_offsets, leaf_start, leaf_cnt);
}

void launch_finalize_leaves(const float* xyz4,
                            const uint32_t* keys_sorted, const uint32_t* vals_sorted,
                            int* d_nleaf, const int* leaf_start, const int* leaf_cnt,
                            int max_leaves, FinalizeParams fp, double* sums, VoxelRecord* rec,
                            LeafStats* stats, int* cell2leaf, hipStream_t s) {
  if (max_leaves <= 0) return;
  size_t blocks = ((size_t)max_leaves * LANES_PER_LEAF + 255) / 256;
  if (blocks > (size_t)SUMS_BLOCKS_MAX) blocks = SUMS_BLOCKS_MAX;
  hipLaunchKernelGGL(k_leaf_sums, dim3((unsigned)blocks), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(xyz4), vals_sorted, d_nleaf,
                     leaf_start, leaf_cnt, sums);
  hipLaunchKernelGGL(k_leaf_finalize, dim3((unsigned)((max_leaves + 255) / 256)), dim3(256), 0, s,
                     keys_sorted, d_nleaf, leaf_start, leaf_cnt, sums, fp, rec, stats, cell2leaf);
}

}  // namespace ndt
