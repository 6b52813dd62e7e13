// ndt_derivs.hip -- NDT score / gradient / Hessian accumulation on gfx950.
//
// Computes what the reference's computeParticleDerivatives (= pclomp
// computeDerivatives) computes (ref: extern/svn_ndt/include/svn_ndt_impl.hpp:
// 518-668, per-pair update :401-513, point derivatives :339-396), for one pose
// or a batch of poses, but organised for CDNA4:
//   * SoA f32 source, one thread per source point, the rigid transform fused in
//     (the reference materialises a transformed cloud, :761);
//   * neighbour lookup = <= 7 loads from a dense int32 cell->leaf grid followed
//     by <= 7 80-byte record loads, all issued before any arithmetic;
//   * algebra refactored: with J = [I | A(x)] every per-pair term of the
//     gradient and Hessian is J^T (.) J of a 3-vector / symmetric 3x3, so a
//     pair only accumulates  w += f v,  S += f (C - d2 v v^T)  (v = C (x'-mu));
//     the 6-vector / 6x6 expansion happens once per POINT, not per pair;
//   * 31 f64 accumulators per thread; recursive-halving reduce-scatter across the
//     wave (32 cross-lane moves, not 32*6), LDS across the 4 waves, one partial row
//     per block; the last-arriving block adds the rows in fixed order inside the
//     same launch (tagged 16-byte slots, no fences), so results are bit-reproducible.
// No MFMA: 3x3 / 6x6 work is not a dense contraction.
// Compiled with -ffp-contract=off; the transform and the index arithmetic must
// round exactly as written to classify points into the same voxels as the ref.
#include "ndt_kernels.h"
#include "ndt_tuning.h"

#include <hip/hip_ext.h>

#include <cstddef>
#include <algorithm>
#include <cstdlib>

namespace ndt {

namespace {

constexpr int MAX_BLOCK = 1024;  // block size is chosen per launch (derivs_block_threads), <= 16 waves

// Classifying a point into a voxel (ref: voxel_grid_covariance_impl.hpp:46-71 for the f32 bounds
// test, voxel_grid_covariance.h:297-300 for the index):
//   in   = lo <= p < hi on every axis (f32);   i_a = (int)(floorf(p_a * inv_leaf) - (float)min_b_a);
//   idx  = i0 + i1 * mul1 + i2 * mul2.
// The reference looks idx up in its hash map whatever its value (f32 rounding at an upper face
// can alias into the next row); an idx outside [0, ncells) can never be a stored key there, so
// it is a miss here too -- same outcome.  point_pairs() evaluates this for the point and its
// six face-offset copies with the per-axis pieces shared.

struct PairAcc {
  double w[3];
  double S[6];
  double score, best;
  int npairs;
};

// ref: svn_ndt_impl.hpp:401-447 (guards, exp, factor); accumulation refactored
// MODE: 0 = score + gradient only, 1 = full analytic Hessian, 2 = Gauss-Newton Hessian,
// 3 = score only (ndt_score_transform: transform probability / NVTL, no derivatives).
// A template parameter (not a run-time flag) so each pair is one straight-line block and
// the compiler is free to keep several records in flight.
template <int MODE>
__device__ __forceinline__ void pair_update(PairAcc& a, const VoxelRecord& r, float xt, float yt,
                                            float zt, const EvalConsts& ec, bool present) {
#pragma clang fp contract(fast)  // f64 accumulation may fuse; only the f32 transform / index math may not
  double x0 = (double)xt - r.mean[0], x1 = (double)yt - r.mean[1], x2 = (double)zt - r.mean[2];
  double v0 = r.icov[0] * x0 + r.icov[1] * x1 + r.icov[2] * x2;
  double v1 = r.icov[1] * x0 + r.icov[3] * x1 + r.icov[4] * x2;
  double v2 = r.icov[2] * x0 + r.icov[4] * x1 + r.icov[5] * x2;
  double q = x0 * v0 + x1 * v1 + x2 * v2;
  a.npairs += present ? 1 : 0;
  // The reference's guards (ref :421-447) as predicates instead of early returns, so the 7
  // unrolled pairs do not each split the wave three more times:
  //   q non-finite or < -1e-9 -> no contribution;  d2*q/2 > 50 -> no contribution;
  //   |d1*d2*e| < 1e-15 -> score only.
  // (ordered compares: a NaN or -Inf q fails the first, a +Inf q the second -- no separate finiteness test; a masked
  // pair's e may be anything finite or not: it is SELECTED away, never multiplied away)
  bool ok = present && q >= -1e-9;
  asm("v_max_f64 %0, %1, 0" : "=v"(q) : "v"(q));   // q = max(q, 0), NaN -> 0 (fmax() costs a canonicalising v_max in front)
  const double earg = ec.d2 * q * 0.5;
  ok = ok && !(earg > 50.0);
  const double e = exp(-earg);   // (computed for every lane and selected below: as `ok ? exp(..) : 0` it became a branch with
                                 // the polynomial's constants re-materialised in vector registers inside it)
  const double sc = ok ? -ec.d1 * e : 0.0;
  a.score += sc;
  a.best = fmax(a.best, sc);
  double f = ec.d1 * ec.d2 * e;
  const bool dok = ok && fabs(f) >= 1e-15;
  f = dok ? f : 0.0;
  // v is finite whenever it gets here: the caller replaces a non-finite transformed point by
  // the origin (it has no neighbours, so every f is 0) and stored records are finite, so
  // f * v is an exact zero for a masked pair without zeroing v itself
  if (MODE != 3) { a.w[0] += f * v0; a.w[1] += f * v1; a.w[2] += f * v2; }
  if (MODE == 1 || MODE == 2) {
    a.S[0] += f * r.icov[0]; a.S[1] += f * r.icov[1]; a.S[2] += f * r.icov[2];
    a.S[3] += f * r.icov[3]; a.S[4] += f * r.icov[4]; a.S[5] += f * r.icov[5];
    if (MODE == 1) {
      const double fd = f * ec.d2;
      a.S[0] -= fd * v0 * v0; a.S[1] -= fd * v0 * v1; a.S[2] -= fd * v0 * v2;
      a.S[3] -= fd * v1 * v1; a.S[4] -= fd * v1 * v2; a.S[5] -= fd * v2 * v2;
    }
  }
}

// The running sums of a point are where the code says they are after every pair: without this the compiler is free to
// postpone  S += f * icov  of all seven pairs to the end of the pair phase (it did, for the Gauss-Newton kernels, once S
// was first used in another basic block: 42 f64 of seven records stayed live and the kernel spilled).  No instructions.
template <int MODE>
__device__ __forceinline__ void pin_sums(PairAcc& a) {
  if (MODE == 1 || MODE == 2)
    asm volatile("" : "+v"(a.S[0]), "+v"(a.S[1]), "+v"(a.S[2]), "+v"(a.S[3]), "+v"(a.S[4]), "+v"(a.S[5]));
  if (MODE != 3) asm volatile("" : "+v"(a.w[0]), "+v"(a.w[1]), "+v"(a.w[2]));
}

__device__ __forceinline__ float dot3f(const float* m, float x, float y, float z) {
  return m[0] * x + m[1] * y + m[2] * z;
}

// Per-POINT expansion of the pair sums into the 6-gradient / 21 Hessian words:
// g = J^T w,  H = J^T S J (+ w . d2x'/dp_i dp_j), J = [I | A(x)] (ref :339-396, :449-494).
struct AngleTables {
  float jang[24];  // 8x3
  float hang[45];  // 15x3
};

// Where an expansion takes a point's pair sums from: the wave's own registers, or another wave's hand-over area in LDS.
// The LDS source fetches a word WHEN it is used (volatile: neither hoisted to the top nor merged with the other use of
// the same word) -- a finishing wave holds 32 running sums in registers across the items it expands, and 22 more
// registers of pair sums fetched up front made a third of the kernel's instantiations spill.
struct RegSource {
  const PairAcc& a;
  __device__ __forceinline__ double score() const { return a.score; }
  __device__ __forceinline__ double best() const { return a.best; }
  __device__ __forceinline__ int npairs() const { return a.npairs; }
  __device__ __forceinline__ double w(int k) const { return a.w[k]; }
  __device__ __forceinline__ double S(int k) const { return a.S[k]; }
};

// The point-dependent factors of an expansion -- the eight non-trivial entries of A(x) (ref :339-363) and, for the full
// Hessian, the fifteen second-derivative products (ref :369-394) -- computed from the angle tables, or fetched from LDS
// where a wave has computed them AHEAD of its pair phase (precompute_geo: the finishing wave that expands last).
struct TableGeo {
  const float* T;
  float x, y, z;
  __device__ __forceinline__ double A(int k) const { return (double)dot3f(T + 3 * k, x, y, z); }
  __device__ __forceinline__ double h(int k) const { return (double)dot3f(T + 24 + 3 * k, x, y, z); }
};
struct PreGeo {
  const float* pre;   // [23][64] f32 (the products ARE f32: the expansion widens them): A(0..7), h(0..14), lane-minor
  int lane;
  __device__ __forceinline__ double A(int k) const { return (double)pre[k * 64 + lane]; }
  __device__ __forceinline__ double h(int k) const { return (double)pre[(8 + k) * 64 + lane]; }
};
constexpr int PRE_WORDS = 23;
constexpr int PRE_BYTES = PRE_WORDS * 64 * 4;   // 5 888: fits in a wave's hand-over region (static_assert below)
// (word(k): table word k as a wave-uniform value.  Eight products at a time: the wave is in front of its pair phase,
// with the grid geometry, R|t and every pointer live in scalar registers -- all 69 words at once would spill them)
template <int MODE, class Word>
__device__ __forceinline__ void precompute_geo(float* pre, int lane, const Word& word, float x, float y, float z) {
  if (MODE == 3) return;
#pragma unroll
  for (int k0 = 0; k0 < (MODE == 1 ? PRE_WORDS : 8); k0 += 8) {
#pragma unroll
    for (int k = k0; k < k0 + 8 && k < (MODE == 1 ? PRE_WORDS : 8); ++k) {
      const float m[3] = {word(3 * k), word(3 * k + 1), word(3 * k + 2)};
      pre[k * 64 + lane] = dot3f(m, x, y, z);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ACCUM: the words are ADDED to acc (a finishing wave expands several waves' points, see k_derivatives).
// T: the 69 angle-table words (jang[24] then hang[45]) -- wave-uniform values the finishing waves hold in SCALAR
// registers (angle_tables_to_sgprs), so that an expansion has no LDS reads in its dependency chains: a finishing wave
// works on the last item of its SIMD alone, with nobody to hide a round trip behind.
// Every product-sum is written as an explicit fma chain and contraction is OFF here: the pre-launched, the ordinary and
// the batched instantiations must round identically (their results are compared bit for bit), which "the compiler fuses
// what it likes" does not promise across separately compiled instantiations.
template <int MODE, bool ACCUM, class Src, class Geo>
__device__ __forceinline__ void expand_point(double acc[EV_WORDS], const Src& src, const Geo& geo) {
#pragma clang fp contract(off)
#define NDT_PUT(k, v) do { if (ACCUM) acc[(k)] += (v); else acc[(k)] = (v); } while (0)
  // a b (+ acc)            /  a b + c d (+ acc)  /  a b + c d + e f (+ acc)
#define NDT_T1(k, a, b) do { acc[(k)] = ACCUM ? __builtin_fma((a), (b), acc[(k)]) : (a) * (b); } while (0)
#define NDT_T2(k, a, b, c, d) do { acc[(k)] = __builtin_fma((a), (b), ACCUM ? __builtin_fma((c), (d), acc[(k)]) : (c) * (d)); } while (0)
#define NDT_T3(k, a, b, c, d, e, f) \
  do { acc[(k)] = __builtin_fma((a), (b), __builtin_fma((c), (d), ACCUM ? __builtin_fma((e), (f), acc[(k)]) : (e) * (f))); } while (0)
  // ... always added to a word that exists already
#define NDT_A1(k, a, b) do { acc[(k)] = __builtin_fma((a), (b), acc[(k)]); } while (0)
#define NDT_A2(k, a, b, c, d) do { acc[(k)] = __builtin_fma((a), (b), __builtin_fma((c), (d), acc[(k)])); } while (0)
#define NDT_A3(k, a, b, c, d, e, f) \
  do { acc[(k)] = __builtin_fma((a), (b), __builtin_fma((c), (d), __builtin_fma((e), (f), acc[(k)]))); } while (0)
  // Written for every thread: a point without neighbours (ref :592) has w = S = 0 and
  // contributes exact zeros, so no divergent skip (and no per-path zero-fill) is needed.
  {
    const int np = src.npairs();
    NDT_PUT(EV_SCORE, src.score());
    NDT_PUT(EV_NVTL, src.best());
    NDT_PUT(EV_NWITH, np > 0 ? 1.0 : 0.0);
    NDT_PUT(EV_NPAIRS, (double)np);
  }
  if (!ACCUM) acc[31] = 0.0;
  if (MODE == 3) {
    if (!ACCUM) {
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[EV_G + k] = 0.0;
    }
    return;
  }

  // point Jacobian, angular block A (3x3) from the ORIGINAL point (ref :339-363)
  const double A10 = geo.A(0), A20 = geo.A(1), A01 = geo.A(2), A11 = geo.A(3), A21 = geo.A(4), A02 = geo.A(5), A12 = geo.A(6),
               A22 = geo.A(7);
  {
    const double w0 = src.w(0), w1 = src.w(1), w2 = src.w(2);
    NDT_PUT(EV_G + 0, w0);
    NDT_PUT(EV_G + 1, w1);
    NDT_PUT(EV_G + 2, w2);
    NDT_T2(EV_G + 3, A10, w1, A20, w2);
    NDT_T3(EV_G + 4, A01, w0, A11, w1, A21, w2);
    NDT_T3(EV_G + 5, A02, w0, A12, w1, A22, w2);
  }
  if (MODE == 0) {
    if (!ACCUM) {
#pragma unroll
      for (int k = 0; k < 21; ++k) acc[EV_H + k] = 0.0;
    }
    return;
  }

  // B = S A row by row, every row consumed at once -- H's translation x rotation block takes it as it is, the rotation
  // block A^T S A = A^T B collects A(i, .) x B(i, .) -- so that three words of B are live at a time, not nine.  A(0, 0) = 0.
  {
    const double Sxx = src.S(0), Sxy = src.S(1), Sxz = src.S(2);
    const double B00 = __builtin_fma(Sxy, A10, Sxz * A20), B01 = __builtin_fma(Sxx, A01, __builtin_fma(Sxy, A11, Sxz * A21)),
                 B02 = __builtin_fma(Sxx, A02, __builtin_fma(Sxy, A12, Sxz * A22));
    NDT_PUT(EV_H + 0, Sxx); NDT_PUT(EV_H + 1, Sxy); NDT_PUT(EV_H + 2, Sxz);
    NDT_PUT(EV_H + 3, B00); NDT_PUT(EV_H + 4, B01); NDT_PUT(EV_H + 5, B02);
    NDT_T1(EV_H + 18, A01, B01); NDT_T1(EV_H + 19, A01, B02); NDT_T1(EV_H + 20, A02, B02);
  }
  {
    const double Sxy = src.S(1), Syy = src.S(3), Syz = src.S(4);
    const double B10 = __builtin_fma(Syy, A10, Syz * A20), B11 = __builtin_fma(Sxy, A01, __builtin_fma(Syy, A11, Syz * A21)),
                 B12 = __builtin_fma(Sxy, A02, __builtin_fma(Syy, A12, Syz * A22));
    NDT_PUT(EV_H + 6, Syy); NDT_PUT(EV_H + 7, Syz);
    NDT_PUT(EV_H + 8, B10); NDT_PUT(EV_H + 9, B11); NDT_PUT(EV_H + 10, B12);
    NDT_T1(EV_H + 15, A10, B10); NDT_T1(EV_H + 16, A10, B11); NDT_T1(EV_H + 17, A10, B12);
    NDT_A1(EV_H + 18, A11, B11); NDT_A1(EV_H + 19, A11, B12); NDT_A1(EV_H + 20, A12, B12);
  }
  {
    const double Sxz = src.S(2), Syz = src.S(4), Szz = src.S(5);
    const double B20 = __builtin_fma(Syz, A10, Szz * A20), B21 = __builtin_fma(Sxz, A01, __builtin_fma(Syz, A11, Szz * A21)),
                 B22 = __builtin_fma(Sxz, A02, __builtin_fma(Syz, A12, Szz * A22));
    NDT_PUT(EV_H + 11, Szz);
    NDT_PUT(EV_H + 12, B20); NDT_PUT(EV_H + 13, B21); NDT_PUT(EV_H + 14, B22);
    NDT_A1(EV_H + 15, A20, B20); NDT_A1(EV_H + 16, A20, B21); NDT_A1(EV_H + 17, A20, B22);
    NDT_A1(EV_H + 18, A21, B21); NDT_A1(EV_H + 19, A21, B22); NDT_A1(EV_H + 20, A22, B22);
  }
  if (MODE == 1) {
    // + second-derivative term, full Hessian only (ref :369-394 layout of the 15 rows; term 3 of :479-489)
    const double w0 = src.w(0), w1 = src.w(1), w2 = src.w(2);
    NDT_A2(EV_H + 15, w1, geo.h(0), w2, geo.h(1));
    NDT_A2(EV_H + 16, w1, geo.h(2), w2, geo.h(3));
    NDT_A2(EV_H + 17, w1, geo.h(4), w2, geo.h(5));
    NDT_A3(EV_H + 18, w0, geo.h(6), w1, geo.h(7), w2, geo.h(8));
    NDT_A3(EV_H + 19, w0, geo.h(9), w1, geo.h(10), w2, geo.h(11));
    NDT_A3(EV_H + 20, w0, geo.h(12), w1, geo.h(13), w2, geo.h(14));
  }
#undef NDT_A3
#undef NDT_A2
#undef NDT_A1
#undef NDT_T3
#undef NDT_T2
#undef NDT_T1
#undef NDT_PUT
}

// The 69 table words out of LDS into scalar registers: lane k fetches word k (and word 64 + k), then one v_readlane per word.
__device__ __forceinline__ void angle_tables_to_sgprs(float T[69], const AngleTables& tab, int lane) {
  const float* w = tab.jang;   // runs on into hang[]: the two arrays are contiguous
  const float a = w[lane], b = w[64 + (lane < 5 ? lane : 0)];
#pragma unroll
  for (int k = 0; k < 64; ++k) T[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), k));
#pragma unroll
  for (int k = 0; k < 5; ++k) T[64 + k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), k));
}

struct RigidRT {
  float R[9];
  float t[3];
};

// One voxel record into registers, from the 80-byte f64 table or the 48-byte packed one (`packed` is a compile-time
// constant at every call site).  The pair arithmetic is the same f64 code either way.
__device__ __forceinline__ VoxelRecord fetch_record(const VoxelRecord* __restrict__ rec, int i, bool packed) {
  if (packed) {
    const PackedRecord p = reinterpret_cast<const PackedRecord*>(rec)[i];
    VoxelRecord r;
    r.mean[0] = p.mean[0]; r.mean[1] = p.mean[1]; r.mean[2] = p.mean[2];
#pragma unroll
    for (int k = 0; k < 6; ++k) r.icov[k] = (double)p.icov[k];
    r.pad = -1.0;
    return r;
  }
  return rec[i];
}

// Phase 1 of a point: transform, neighbour lookup, pair sums.  Needs only R|t.
// D7: DIRECT7 (centre + 6 face neighbours); otherwise DIRECT1 (the point's own voxel only).
// PACKED: the record table is PackedRecord[] (compile time here: as a run-time choice the seven pipelined fetches
// of DIRECT7 merged both formats' registers and spilled 48-140 bytes per lane)
// mid(): called once the cell lookups are on their way -- work that does not depend on them (precompute_geo) runs in
// their shadow.  Every lane takes part: one without a point (active = false) has no neighbours.
template <int MODE, bool D7, bool PACKED, class Mid>
__device__ __forceinline__ void point_pairs(PairAcc& a, float x, float y, float z, const GridGeom& g,
                                            const int* __restrict__ cell2leaf,
                                            const VoxelRecord* __restrict__ rec, const RigidRT& P,
                                            const EvalConsts& ec, bool active, const Mid& mid) {
  a.w[0] = a.w[1] = a.w[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) a.S[k] = 0.0;
  a.score = 0.0; a.best = 0.0; a.npairs = 0;
  // x' = r0*x + (r1*y + (r2*z + t)), f32, unfused (transformPointCloud, ref :761)
  float xt = P.R[0] * x + (P.R[1] * y + (P.R[2] * z + P.t[0]));
  float yt = P.R[3] * x + (P.R[4] * y + (P.R[5] * z + P.t[1]));
  float zt = P.R[6] * x + (P.R[7] * y + (P.R[8] * z + P.t[2]));
  const bool finite = active && isfinite(xt) && isfinite(yt) && isfinite(zt);  // ref :573: such points are skipped
  if (!finite) { xt = 0.0f; yt = 0.0f; zt = 0.0f; }  // keeps NaN / Inf out of the (masked) pair arithmetic

  // ref: voxel_grid_covariance_impl.hpp:560-600 -- neighbours are found by offsetting the
  // POINT by +-leaf in f32 and re-classifying it (see above).  The seven classifications share
  // their per-axis pieces: a probe differs from the centre in one coordinate only, so only that
  // axis' bounds test, floor and index are recomputed -- same f32 operations on the same
  // operands as seven independent classifications, a third of the instructions.
  const float w = g.leaf;
  const float cx[3] = {xt, xt + w, xt - w}, cy[3] = {yt, yt + w, yt - w}, cz[3] = {zt, zt + w, zt - w};
  bool inx[3], iny[3], inz[3];
  unsigned int ix[3], iy[3], iz[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    inx[k] = cx[k] >= g.lo[0] && cx[k] < g.hi[0];
    iny[k] = cy[k] >= g.lo[1] && cy[k] < g.hi[1];
    inz[k] = cz[k] >= g.lo[2] && cz[k] < g.hi[2];
    // (out-of-range coordinates convert to saturated garbage that the bounds flags mask;
    // unsigned arithmetic so that garbage may wrap)
    ix[k] = (unsigned int)(int)(floorf(cx[k] * g.inv_leaf) - (float)g.min_b[0]);
    iy[k] = (unsigned int)(int)(floorf(cy[k] * g.inv_leaf) - (float)g.min_b[1]) * (unsigned int)g.mul1;
    iz[k] = (unsigned int)(int)(floorf(cz[k] * g.inv_leaf) - (float)g.min_b[2]) * (unsigned int)g.mul2;
  }
  int cell[7];
  {
    const int idx0 = (int)(ix[0] + iy[0] + iz[0]);
    cell[0] = (inx[0] && iny[0] && inz[0] && idx0 >= 0 && idx0 < g.ncells) ? idx0 : -1;
  }
  if (D7) {
#pragma unroll
    for (int k = 1; k < 3; ++k) {
      const int ax = (int)(ix[k] + iy[0] + iz[0]), ay = (int)(ix[0] + iy[k] + iz[0]), az = (int)(ix[0] + iy[0] + iz[k]);
      // an idx outside [0, ncells) can never be a stored key of the reference's hash map: a miss
      cell[k] = (inx[k] && iny[0] && inz[0] && ax >= 0 && ax < g.ncells) ? ax : -1;
      cell[2 + k] = (inx[0] && iny[k] && inz[0] && ay >= 0 && ay < g.ncells) ? ay : -1;
      cell[4 + k] = (inx[0] && iny[0] && inz[k] && az >= 0 && az < g.ncells) ? az : -1;
    }
  } else {
#pragma unroll
    for (int k = 1; k < 7; ++k) cell[k] = -1;
  }
  if (!D7) {  // DIRECT1 (ref: getNeighborhoodAtPoint1, voxel_grid_covariance_impl.hpp:604-615)
    const int slot0 = (finite && cell[0] >= 0) ? cell2leaf[cell[0]] : -1;
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
    const VoxelRecord r0 = fetch_record(rec, slot0 >= 0 ? slot0 : 0, PACKED);
    pair_update<MODE>(a, r0, xt, yt, zt, ec, slot0 >= 0);
    return;
  }
  int slot[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) slot[k] = (finite && cell[k] >= 0) ? cell2leaf[cell[k]] : -1;
  __builtin_amdgcn_sched_barrier(0);
  mid();
  __builtin_amdgcn_sched_barrier(0);
  // Fully predicated: an absent neighbour reads record 0 (a wave-wide broadcast) and is
  // masked out, so the unrolled pairs carry no exec-mask splits and no accumulator merges.
  // Records are fetched in two batches (4 + 3) so a point pays two L2 round trips for its
  // neighbours instead of seven, within a 128-VGPR budget.
  // Software-pipelined by hand: THREE records in flight, each further one requested as soon as
  // one has been consumed (four were in flight in rounds 1-2: with the final sum's retry state on top, the
  // full-Hessian and Gauss-Newton instantiations then spilled 2-3 VGPRs; interleaved A/B on one box,
  // profiles/r03_step_ab_summer_depth.txt: no difference in wall time per evaluation, 0.7 us less per
  // ordinary launch by HIP events).  The scheduling fences keep the compiler from sinking the loads
  // back to their first use (it otherwise serialises seven L2 round trips per point).
  const VoxelRecord r0 = fetch_record(rec, slot[0] >= 0 ? slot[0] : 0, PACKED);
  const VoxelRecord r1 = fetch_record(rec, slot[1] >= 0 ? slot[1] : 0, PACKED);
  const VoxelRecord r2 = fetch_record(rec, slot[2] >= 0 ? slot[2] : 0, PACKED);
  __builtin_amdgcn_sched_barrier(0);
  pair_update<MODE>(a, r0, xt, yt, zt, ec, slot[0] >= 0);
  pin_sums<MODE>(a);
  const VoxelRecord r3 = fetch_record(rec, slot[3] >= 0 ? slot[3] : 0, PACKED);
  __builtin_amdgcn_sched_barrier(0);
  pair_update<MODE>(a, r1, xt, yt, zt, ec, slot[1] >= 0);
  pin_sums<MODE>(a);
  const VoxelRecord r4 = fetch_record(rec, slot[4] >= 0 ? slot[4] : 0, PACKED);
  __builtin_amdgcn_sched_barrier(0);
  pair_update<MODE>(a, r2, xt, yt, zt, ec, slot[2] >= 0);
  pin_sums<MODE>(a);
  const VoxelRecord r5 = fetch_record(rec, slot[5] >= 0 ? slot[5] : 0, PACKED);
  __builtin_amdgcn_sched_barrier(0);
  pair_update<MODE>(a, r3, xt, yt, zt, ec, slot[3] >= 0);
  pin_sums<MODE>(a);
  const VoxelRecord r6 = fetch_record(rec, slot[6] >= 0 ? slot[6] : 0, PACKED);
  __builtin_amdgcn_sched_barrier(0);
  pair_update<MODE>(a, r4, xt, yt, zt, ec, slot[4] >= 0);
  pin_sums<MODE>(a);
  pair_update<MODE>(a, r5, xt, yt, zt, ec, slot[5] >= 0);
  pin_sums<MODE>(a);
  pair_update<MODE>(a, r6, xt, yt, zt, ec, slot[6] >= 0);
  pin_sums<MODE>(a);
}

// FLANN's L2_Simple in f32, accumulated x, y, z, strict `<` (ref: radiusSearch,
// voxel_grid_covariance_impl.hpp:505-554; centroids are the f32-rounded leaf means, :420-422).
// Deliberately outside the FMA-contracted regions.
__device__ __forceinline__ bool kd_within(const VoxelRecord& r, float xt, float yt, float zt, float r2) {
  const float ex = xt - (float)r.mean[0], ey = yt - (float)r.mean[1], ez = zt - (float)r.mean[2];
  float d = ex * ex;
  d = d + ey * ey;
  d = d + ez * ez;
  return d < r2;
}

// KDTREE neighbourhood: every valid voxel whose centroid lies within one leaf size of the
// point.  Such a centroid can only be in the 3x3x3 cells around the point's cell, so the
// kd-tree becomes 27 index probes + a distance test.  A point has 3-4 such neighbours out of
// 27 cells, so the candidates are compacted first: every lane writes the leaves of its
// occupied cells to its own LDS column (cell order, so the summation order is fixed), and the
// wave then runs max-over-lanes(count) predicated pair updates instead of 27.
constexpr int KD_CELLS = 27;

// RADIUS = false is DIRECT26 [RECALLED] (pclomp getNeighborhoodAtPoint): the same 27-cell
// enumeration in integer index space, every valid leaf found is a neighbour.
// CHAIN (multi-grid target, [RECALLED] tier4 MultiGridNormalDistributionsTransform): the table is the
// union of several separately voxelised grids on the same absolute lattice, a cell may hold one leaf
// per grid; cell2leaf gives the first, VoxelRecord::pad of a leaf the slot of the next one in the
// same cell (-1: none).  The chain is walked inside the trip loop (wave-uniform exit), so a point's
// pairs are added in (cell, grid) order.
template <int MODE, bool RADIUS, bool CHAIN>
__device__ __forceinline__ void point_pairs_kd(PairAcc& a, float x, float y, float z, const GridGeom& g,
                                               const int* __restrict__ cell2leaf,
                                               const VoxelRecord* __restrict__ rec, const float4* __restrict__ cent,
                                               const RigidRT& P, const EvalConsts& ec, int* __restrict__ lds_list /* the wave's own KD_CELLS x 64 ints */,
                                               bool active) {
  a.w[0] = a.w[1] = a.w[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) a.S[k] = 0.0;
  a.score = 0.0; a.best = 0.0; a.npairs = 0;
  float xt = P.R[0] * x + (P.R[1] * y + (P.R[2] * z + P.t[0]));
  float yt = P.R[3] * x + (P.R[4] * y + (P.R[5] * z + P.t[1]));
  float zt = P.R[6] * x + (P.R[7] * y + (P.R[8] * z + P.t[2]));
  const bool finite = active && isfinite(xt) && isfinite(yt) && isfinite(zt);
  if (!finite) { xt = 0.0f; yt = 0.0f; zt = 0.0f; }
  // the point's own (possibly out-of-box) cell, same f32 arithmetic as the grid build
  const int i0 = finite ? (int)(floorf(xt * g.inv_leaf) - (float)g.min_b[0]) : -4;
  const int i1 = finite ? (int)(floorf(yt * g.inv_leaf) - (float)g.min_b[1]) : -4;
  const int i2 = finite ? (int)(floorf(zt * g.inv_leaf) - (float)g.min_b[2]) : -4;
  // The 27 cells are nine rows of three x-adjacent cells: ONE 12-byte load per row (nine loads in flight instead of
  // 27; round 3).  At the ends of the grid a row starts one or two ints outside it -- the index grid is readable four
  // ints beyond either end (IndexGrid, ndt_engine.h) and those lanes are masked.  Cell n = (dx+1) + 3 (dy+1) + 9 (dz+1),
  // as before: the listing order, hence the summation order, is unchanged.
  int slot[KD_CELLS];
  {
    // (unsigned compares: one instruction per range test; bitwise `&` on the flags: the short-circuit form compiles to an
    // exec-mask split per row; the row offsets differ by +-mul1 / +-mul2 from ONE base: two integer multiplies, not 18)
    const unsigned int d0 = (unsigned int)g.div_b[0], d1 = (unsigned int)g.div_b[1], d2 = (unsigned int)g.div_b[2];
    const bool xin0 = (unsigned int)(i0 - 1) < d0, xin1 = (unsigned int)i0 < d0, xin2 = (unsigned int)(i0 + 1) < d0;
    const bool xany = xin0 | xin1 | xin2;
    const bool yin[3] = {(unsigned int)(i1 - 1) < d1, (unsigned int)i1 < d1, (unsigned int)(i1 + 1) < d1};
    const bool zin[3] = {(unsigned int)(i2 - 1) < d2, (unsigned int)i2 < d2, (unsigned int)(i2 + 1) < d2};
    const int base0 = (i0 - 1) + i1 * g.mul1 + i2 * g.mul2;
    struct alignas(4) Int3 { int a, b, c; };
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      const bool rowok = xany & yin[r % 3] & zin[r / 3];
      const int base = rowok ? base0 + ((r % 3) - 1) * g.mul1 + ((r / 3) - 1) * g.mul2 : 0;
      const Int3 v = *reinterpret_cast<const Int3*>(cell2leaf + base);
      slot[3 * r + 0] = (rowok & xin0) ? v.a : -1;
      slot[3 * r + 1] = (rowok & xin1) ? v.b : -1;
      slot[3 * r + 2] = (rowok & xin2) ? v.c : -1;
    }
  }
  // column `lane` of the wave's own lds_list[KD_CELLS][64]: conflict-free, and private to the wave (it becomes the
  // wave's hand-over area to the finishing waves once the pairs are done, see k_derivatives)
  constexpr int stride = 64;
  const int kd_lane = (int)(threadIdx.x & 63u);
  int count = 0;
  bool filtered = false;  // the LDS column holds leaves that already passed the distance test
  if (RADIUS && CHAIN) {
    // multi-grid union: as for KDTREE below, with the chains walked at listing time (a lane follows
    // the `pad` links of its own cells; only cells that several grids share have any).  A lane with
    // more than KD_CELLS centroids within the radius (seven or more grids on top of each other) makes
    // its wave take the unfiltered path below instead.
    bool overflow = false;
#pragma unroll
    for (int n0 = 0; n0 < KD_CELLS; n0 += 9) {
      float4 m[9];   // f32 centroid + chain link: 16 bytes per leaf instead of 32 of its 80-byte record
#pragma unroll
      for (int q = 0; q < 9; ++q) m[q] = cent[slot[n0 + q] >= 0 ? slot[n0 + q] : 0];
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        if (slot[n0 + q] < 0) continue;
        int sl = slot[n0 + q];
        float4 c = m[q];
        for (;;) {
          const float ex = xt - c.x, ey = yt - c.y, ez = zt - c.z;
          float d = ex * ex;
          d = d + ey * ey;
          d = d + ez * ez;
          if (d < ec.kd_radius2) {
            if (count < KD_CELLS) { lds_list[count * stride + kd_lane] = sl; ++count; }
            else overflow = true;
          }
          sl = __float_as_int(c.w);
          if (sl < 0) break;
          c = cent[sl];
        }
      }
    }
    filtered = __ballot(overflow) == 0ull;  // wave-uniform
    if (!filtered) count = 0;
  }
  if (RADIUS && CHAIN && filtered) {
    // (listed above)
  } else if (RADIUS && !CHAIN) {
    // KDTREE: the distance test comes BEFORE the listing, on the centroids alone (24 bytes per occupied
    // cell, nine cells in flight at a time): a point has 9-11 occupied cells around it but only 3-4
    // centroids within one leaf size, and the wave runs max-over-lanes(listed) full pair updates --
    // 7 instead of 15 on C3.  Same f32 test on the same operands, same (cell) order: the same pairs.
    constexpr int KB = 9;   // centroids in flight per lane (14 = two round trips instead of three: no faster, profiles/r04_kd_batch_ab.txt)
#pragma unroll
    for (int n0 = 0; n0 < KD_CELLS; n0 += KB) {
      float4 m[KB];   // the f32 centroids (one 16-byte load per cell; two loads of the record's f64 mean before round 3)
#pragma unroll
      for (int q = 0; q < KB; ++q) m[q] = cent[(n0 + q < KD_CELLS && slot[n0 + q < KD_CELLS ? n0 + q : 0] >= 0) ? slot[n0 + q < KD_CELLS ? n0 + q : 0] : 0];
#pragma unroll
      for (int q = 0; q < KB; ++q) {
        if (n0 + q >= KD_CELLS) break;
        const float ex = xt - m[q].x, ey = yt - m[q].y, ez = zt - m[q].z;
        float d = ex * ex;
        d = d + ey * ey;
        d = d + ez * ez;
        // (no branch: the slot is stored at position `count` whether it is a neighbour or not -- the next cell overwrites
        // a position that was not taken, and positions >= count are never read; KDTREE -0.2 ... -0.5 us per launch)
        const int in = (int)((unsigned int)(slot[n0 + q] >= 0) & (unsigned int)(d < ec.kd_radius2));
        lds_list[count * stride + kd_lane] = slot[n0 + q];
        count += in;
      }
    }
  } else {
#pragma unroll
    for (int n = 0; n < KD_CELLS; ++n) {
      if (slot[n] >= 0) {   // (DIRECT26 keeps the branch: 27 unconditional stores cost it 0.5 us, profiles/r04_kd_mask_ab.txt)
        lds_list[count * stride + kd_lane] = slot[n];
        ++count;
      }
    }
  }
  // wave-uniform trip count (the lanes of a wave only read their own column: no barrier needed)
  // (count <= KD_CELLS = 27: five ballots find the maximum bit by bit and leave it in a SCALAR register; the butterfly of
  // six shuffles that stood here shared its lane-address arithmetic with the wave reduction at the end of the kernel,
  // and the compiler kept those addresses in registers -- or spilled them -- across the whole pair loop)
  int trips = 0;
#pragma unroll
  for (int b = 4; b >= 0; --b) {
    const int c = trips | (1 << b);
    if (__ballot(count >= c) != 0ull) trips = c;
  }
  // (Requesting the record of trip j + 1 before trip j is worked on was measured: KDTREE 33.7 against
  // 32.0 us, DIRECT26 32.9 against 30.2 -- the other waves of the SIMD already hide a trip's round trip.)
  for (int j = 0; j < trips; ++j) {
    const bool have = j < count;
    if (!CHAIN) {
      const int sl = have ? lds_list[j * stride + kd_lane] : 0;
      const VoxelRecord r = rec[sl];
      pair_update<MODE>(a, r, xt, yt, zt, ec, have);  // (KDTREE: only centroids within the radius were listed)
    } else if (RADIUS && filtered) {
      const int sl = have ? lds_list[j * stride + kd_lane] : 0;
      const VoxelRecord r = rec[sl];
      pair_update<MODE>(a, r, xt, yt, zt, ec, have);
    } else {
      int sl = have ? lds_list[j * stride + kd_lane] : -1;
      while (__ballot(sl >= 0) != 0ull) {  // every lane of the wave leaves together
        const bool live = sl >= 0;
        const VoxelRecord r = rec[live ? sl : 0];
        const bool present = live && (!RADIUS || kd_within(r, xt, yt, zt, ec.kd_radius2));
        pair_update<MODE>(a, r, xt, yt, zt, ec, present);
        pin_sums<MODE>(a);
        sl = live ? (int)r.pad : -1;
      }
    }
  }
}

// Recursive-halving reduce-scatter of 32 f64 words over the wave: after 6 steps lane
// l holds the wave sum of word l>>1.  The two widest steps use gfx950's
// v_permlane32_swap / v_permlane16_swap (exchange the upper half of one register with
// the lower half of another in one instruction, no select needed); the last four
// (8 -> 1 words) stay inside a row of 16 lanes and go through DPP.  ~100 instructions instead of 6 x 32 shuffles.
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

template <bool SWAP32>
__device__ __forceinline__ double swap_add(double a, double b) {
  // a: word i, b: word i + N/2.  Returns (word i summed over the exchanged half) in the
  // lower lanes / even rows and (word i + N/2 ...) in the upper lanes / odd rows.
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a);
  const unsigned long long ub = (unsigned long long)__double_as_longlong(b);
  uint2v lo, hi;
  if (SWAP32) {
    lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  } else {
    lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  }
  const double na = __longlong_as_double((long long)(((unsigned long long)hi[0] << 32) | lo[0]));
  const double nb = __longlong_as_double((long long)(((unsigned long long)hi[1] << 32) | lo[1]));
  return na + nb;
}

// In-row steps: the partner's word comes through DPP (two v_mov_b32_dpp per f64, no LDS round trip -- a finishing wave
// runs this alone on its SIMD at the end of a block, where four ds_bpermute latencies in a row were a third of the
// reduction's time).  CTRL: row_ror:8 (lane ^ 8), row_half_mirror (lane -> 7 - lane within 8: ANY pairing of the two
// halves serves a reduce-scatter), quad_perm [2,3,0,1] and [1,0,3,2] (lane ^ 2, lane ^ 1).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)u, CTRL, 0xf, 0xf, false);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int N, int CTRL>
__device__ __forceinline__ void rs_step(double* a, bool upper) {
#pragma unroll
  for (int i = 0; i < N / 2; ++i) {
    const double send = upper ? a[i] : a[i + N / 2];
    const double keep = upper ? a[i + N / 2] : a[i];
    a[i] = keep + dpp_f64<CTRL>(send);
  }
}

__device__ __forceinline__ void wave_reduce_scatter32(double* acc, int lane) {
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = swap_add<true>(acc[i], acc[i + 16]);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = swap_add<false>(acc[i], acc[i + 8]);
  rs_step<8, 0x128>(acc, (lane & 8) != 0);
  rs_step<4, 0x141>(acc, (lane & 4) != 0);
  rs_step<2, 0x4E>(acc, (lane & 2) != 0);
  acc[0] += dpp_f64<0xB1>(acc[0]);
}

// ---- finishing waves (round 5) ----------------------------------------------------------------------------------
// Per-wave stamps (profiles/r05_stamps_per_wave.txt) showed what the 200 k-point launch waits for: a block's 13 waves
// sit 4 + 3 + 3 + 3 on the SIMDs of its compute unit, every wave carried its points through pairs (~690 VALU
// instructions), expansion (~220) and reduce-scatter (~140), and the block's row waited for the FOURTH wave of the full
// SIMD -- 4 x 1.75 us of VALU issue behind the first memory round trips.  Now a wave hands its per-point pair sums
// (w, S, score, best) to LDS and only FOUR finishing waves per block, one per SIMD under the hardware's round-robin
// placement (wave i on SIMD i mod 4), expand and reduce: their own points straight from registers, the other waves'
// from LDS, accumulated per lane in a fixed order, then ONE reduce-scatter each.
//   * A SIMD that holds one wave more than the others finishes its pair phases last whatever happens: its finishing
//     wave is its LAST wave, which expands its own points only -- the tail behind the last pair phase is one expansion
//     from registers and one reduce-scatter.  Its earlier waves' points go to the other SIMDs' finishing waves.
//   * The other SIMDs' finishing waves are their FIRST (oldest) waves: they are through their own pairs first, and
//     the hardware issues oldest-first, so their expansions interleave with the younger waves' pair phases instead of
//     queueing up behind them (a finishing wave that is alone on its SIMD issues one dependent instruction at a time:
//     measured 0.9 us per expansion against 0.4 interleaved).
// The item -> finishing wave table (EvalConsts::item_owner / fin_waves, derivs_item_owners()) balances pair phases
// against expansions: 13 waves = SIMD loads (4 P + 1 E, 3 P + 4 E, 3 P + 4 E, 3 P + 4 E) instead of 4 x (P + E + R) on
// the full one, and 4 reduce-scatters per block instead of 13.  The placement assumption costs speed, never correctness,
// when it does not hold: tables and order of additions are fixed per block shape, so sums stay bit-reproducible.
// A wave's LDS region: [64 x {x, y, z, npairs}] first -- the point goes there BEFORE the pair phase, so that it is not
// held in registers through it (the kernel sits at the 128-VGPR edge) -- then the 11 f64 words per lane; in the 27-cell
// modes the candidate list lies where the f64 words will go (it is dead when they are written).
constexpr int PART_F64 = 11;                                   // score, best, w[3], S[6]
constexpr int PART_XYZ_BYTES = 64 * 16;
constexpr int PART_BYTES = PART_XYZ_BYTES + PART_F64 * 64 * 8; // 6656 bytes per wave
static_assert(PRE_BYTES <= PART_BYTES, "the precomputed factors live in the finishing wave's own region");
__host__ __device__ constexpr int wave_region_bytes(bool kd) {
  return kd ? (PART_XYZ_BYTES + KD_CELLS * 64 * 4 > PART_BYTES ? PART_XYZ_BYTES + KD_CELLS * 64 * 4 : PART_BYTES) : PART_BYTES;
}

__device__ __forceinline__ double* region_part(char* region) { return reinterpret_cast<double*>(region + PART_XYZ_BYTES); }
__device__ __forceinline__ int* region_npairs(char* region, int lane) { return reinterpret_cast<int*>(region) + 4 * lane + 3; }

// before the pair phase: the point as the expansion wants it -- a non-finite one (it never has neighbours, ref :573, so
// w = S = 0) as the origin, so that it expands to exact zeros
__device__ __forceinline__ void store_point(char* region, int lane, float x, float y, float z) {
  const bool fin = isfinite(x) && isfinite(y) && isfinite(z);
  reinterpret_cast<float4*>(region)[lane] = make_float4(fin ? x : 0.0f, fin ? y : 0.0f, fin ? z : 0.0f, 0.0f);
}
__device__ __forceinline__ void load_point(const char* region, int lane, float& x, float& y, float& z) {
  const float4 q = reinterpret_cast<const float4*>(region)[lane];
  x = q.x; y = q.y; z = q.z;
}

template <int MODE>
__device__ __forceinline__ void store_partials(double* d /* the region's f64 words */, int* np /* this lane's pair count */, int lane, const PairAcc& a) {
  d[0 * 64 + lane] = a.score;
  d[1 * 64 + lane] = a.best;
  if (MODE != 3) {
#pragma unroll
    for (int k = 0; k < 3; ++k) d[(2 + k) * 64 + lane] = a.w[k];
  }
  if (MODE == 1 || MODE == 2) {
#pragma unroll
    for (int k = 0; k < 6; ++k) d[(5 + k) * 64 + lane] = a.S[k];
  }
  *np = a.npairs;
}

// (see RegSource)
struct LdsSource {
  const double* part;   // the region's f64 words
  const int* np;        // this lane's pair count
  int lane;
  __device__ __forceinline__ double f64(int k) const { return part[k * 64 + lane]; }
  __device__ __forceinline__ double score() const { return f64(0); }
  __device__ __forceinline__ double best() const { return f64(1); }
  __device__ __forceinline__ int npairs() const { return *np; }
  __device__ __forceinline__ double w(int k) const { return f64(2 + k); }
  __device__ __forceinline__ double S(int k) const { return f64(5 + k); }
};

// A wave has published its item (its pair sums, and -- waves 0 and 1 of an ordinary or batched launch -- its share of
// the angle tables): LDS traffic of one wave is executed in order, the flag goes out behind the data.
// The flag is a 64-bit TAG of (launch, block, wave, process): LDS is not initialised, and no barrier stands between a
// wave clearing its own flag -- the first thing it does -- and a finishing wave looking at it, microseconds later, behind
// its own pair phase (a block barrier at the start made every wave wait for the last one to be launched, 0.44 us, and
// start its memory round trips in lock-step with the others).  Should a wave ever be launched that late, the word its
// flag holds is what an earlier workgroup left there: equal to this tag with probability 2^-64.
__device__ __forceinline__ unsigned long long item_tag(unsigned long long base, int wave) { return base ^ (unsigned long long)wave; }
__device__ __forceinline__ void publish_item(unsigned long long* s_item, unsigned long long base, int wave, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) __hip_atomic_store(&s_item[wave], item_tag(base, wave), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The waited-for wave belongs to the same workgroup (resident by construction) and publishes unconditionally; the trip
// bound (~0.5 s) only keeps a wave from spinning for ever should that ever be broken.  Returns false on the bound.
__device__ __forceinline__ bool wait_item(unsigned long long* s_item, unsigned long long base, int it) {
  const unsigned long long want = item_tag(base, it);
  for (int trip = 0; trip < (1 << 22); ++trip) {
    if (__hip_atomic_load(&s_item[it], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == want) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

#ifdef NDT_STAMPS
// diagnostic build only: 100 MHz wall-clock stamps of wave 0 / lane 0 of every block,
// written to a side buffer no other code reads (cdna_hip_programming.md section 7)
__device__ unsigned long long g_stamps[4096 * 8];
__device__ unsigned int g_hwid[4096 * 2];
__device__ unsigned long long g_mstamps[4096 * 2];  // pre-launched launches that computed: {entry, pose seen}
#define NDT_STAMP(k)                                                                   \
  do {                                                                                 \
    if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                    \
      __builtin_amdgcn_s_waitcnt(0);                                                   \
      g_stamps[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();               \
    }                                                                                  \
  } while (0)
// ... and per WAVE (round 5): lane 0 of every wave of the first WS_BLOCKS blocks keeps WS_N stamps in LDS (a ds_write
// per stamp, no vector-memory traffic, no wait on outstanding loads or stores unless the stamp asks for it) and
// flushes them to the side buffer once, behind the block's row store.  Stamps: 0 entry, 1 xyz loaded, 2 pairs done,
// 3 angle tables visible (barrier), 4 expanded, 5 wave reduce-scatter done, 6 cross-wave barrier passed,
// 7 row store issued (wave 0) / leaving (others), 8 row store acknowledged (wave 0).  The summing block: 5 polling
// starts, 6 this wave has all its slots, 7 barrier passed, 8 result store issued, 9 acknowledged; and per poll trip of
// its wave 0 {time, lanes still missing a slot}.
constexpr int WS_BLOCKS = 512, WS_WAVES = 16, WS_N = 10, WS_TRIPS = 64;
__device__ unsigned long long g_wstamps[WS_BLOCKS * WS_WAVES * WS_N];
__device__ unsigned int g_whwid[WS_BLOCKS * WS_WAVES];
__device__ unsigned long long g_sumtrips[WS_TRIPS * 2 + 1];
__device__ __forceinline__ unsigned long long* ws_row() {
  __shared__ unsigned long long s_ws[WS_WAVES][WS_N];
  return s_ws[threadIdx.x >> 6];
}
#define NDT_WSTAMP(k)                                                                  \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if ((threadIdx.x & 63u) == 0 && blockIdx.y == 0) ws_row()[(k)] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#define NDT_WSTAMP_DRAINED(k) do { __builtin_amdgcn_s_waitcnt(0); NDT_WSTAMP(k); } while (0)
__device__ __forceinline__ void ws_flush() {
  if ((threadIdx.x & 63u) == 0 && blockIdx.y == 0 && blockIdx.x < WS_BLOCKS) {
    const unsigned long long* r = ws_row();
    unsigned long long* o = g_wstamps + ((size_t)blockIdx.x * WS_WAVES + (threadIdx.x >> 6)) * WS_N;
#pragma unroll
    for (int k = 0; k < WS_N; ++k) o[k] = r[k];
    g_whwid[blockIdx.x * WS_WAVES + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
}
__device__ __forceinline__ void ws_clear() {
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long* r = ws_row();
#pragma unroll
    for (int k = 0; k < WS_N; ++k) r[k] = 0ull;
  }
}
#else
#define NDT_STAMP(k) do {} while (0)
#define NDT_WSTAMP(k) do {} while (0)
#define NDT_WSTAMP_DRAINED(k) do {} while (0)
#endif

constexpr int NGROUPS = 32;          // second-level fan-in (only for grids above SINGLE_LEVEL_MAX rows)
// (rows one block adds directly: ndt_tuning::deriv_single_level_max, default 2048)
constexpr int COUNTERS_PER_POSE = 1 + NGROUPS;
constexpr int MAX_WAVES = MAX_BLOCK / 64;

// Cross-block hand-off without fences and without waiting for a store to be acknowledged
// (cdna_hip_programming.md Guideline 16, "every store sc1 ... every load sc1" form, taken one
// step further).  A partial row is 32 slots of 16 bytes {tag, value}: the tag is the launch's
// process-wide unique sequence number, and a slot is written by ONE 16-byte store of one lane
// and read by ONE 16-byte load of one lane, so a reader sees a slot either entirely old (a tag
// of an earlier launch) or entirely new.  The storing wave takes its ticket right after issuing
// the stores; the block that draws the last ticket therefore knows every row has been
// *issued*, reads the rows with agent-scope loads and simply re-reads a slot whose tag is not
// this launch's yet.  The finished evaluation goes to pinned host memory in the same slot
// format and the host polls the 32 tags.  Against the previous protocol (drain the row stores,
// then ticket; drain the result stores, then a flag word) this removes two store-acknowledge
// round trips (~1 us each) from every evaluation's critical path; a release/acquire fence pair
// per block had cost 2.5x the kernel.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROW_WORDS = 2 * EV_WORDS;   // 32 x {tag, value}
constexpr int AUX_AGENT = 16;             // sc1
constexpr int AUX_SYSTEM = 17;            // sc0 sc1
constexpr int SUM_BATCH = 16;             // rows per thread and memory round trip (64 VGPRs in flight; 20 spilled once the retry mask joined them)
// 2 s of the 100 MHz clock.  The summing block waits for blocks of its OWN launch that may not have been dispatched
// yet: on a device shared with other processes (whose waiting kernels hold compute units for up to 20 ms each) that
// took longer than the 100 ms this limit stood at for a while -- four ranks rehearsing on one device lost rows.
constexpr unsigned long long SUM_TIMEOUT_TICKS = 200000000ull;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t slots_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFFu, 0x00020000);
}
__device__ __forceinline__ void store_slot(__amdgpu_buffer_rsrc_t r, unsigned int byte_off, unsigned long long tag,
                                           double v, bool system_scope) {
  const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
  u32x4 d;
  d.x = (unsigned int)tag; d.y = (unsigned int)(tag >> 32); d.z = (unsigned int)vb; d.w = (unsigned int)(vb >> 32);
  if (system_scope) __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, AUX_SYSTEM);
  else __builtin_amdgcn_raw_buffer_store_b128(d, r, byte_off, 0, AUX_AGENT);
}

// wave 0, lane 0, right after wave 0 issued the block's row stores
__device__ __forceinline__ int ticket_is_last(unsigned int* counter, unsigned int expected) {
  const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (t == expected - 1u) ? 1 : 0;
}

// Fixed-order sum of the tagged rows [first, end) of `rows` (byte offsets from its base):
// thread (c = tid>>5, v = tid&31) takes slot v of rows first+c, first+c+ncols, ... (ncols =
// blockDim/32); SUM_BATCH loads are issued before the first add so one memory round trip covers
// SUM_BATCH*ncols rows.  The column sums are then added in order and the result is written to
// dst as tagged slots (group rows, or the host's result buffer) or as 32 plain doubles.
// Cross-rank step of the final sum (XchgInfo in ndt_device.h): lane v < 32 holds word v of this rank's
// sum.  Publish it into every rank's area, gather the N rows of the own area, add in rank order.
// Returns the global word; *late is set when a peer's row had not arrived in time.
__device__ __forceinline__ double xchg_allsum(const XchgInfo* __restrict__ xi, unsigned long long round, int v,
                                              double mine, bool* late) {
  const int me = xi->rank, n = xi->nranks;
  for (int r = 0; r < n; ++r)
    store_slot(slots_rsrc(reinterpret_cast<const void*>(xi->area[r])), xchg_slot_offset(round, me, v), round, mine, true);
  const __amdgpu_buffer_rsrc_t own = slots_rsrc(reinterpret_cast<const void*>(xi->area[me]));
  const unsigned int tag_lo = (unsigned int)round, tag_hi = (unsigned int)(round >> 32);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
  constexpr int XB = 8;  // rows in flight per lane: one memory round trip for a node of 8
  for (int r0 = 0; r0 < n; r0 += XB) {
    u32x4 q[XB];
    for (;;) {
      asm volatile("" ::: "memory");  // re-issued on every trip
      bool ok = true;
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        if (r0 + k < n) q[k] = __builtin_amdgcn_raw_buffer_load_b128(own, xchg_slot_offset(round, r0 + k, v), 0, AUX_SYSTEM);
        else { q[k].x = tag_lo; q[k].y = tag_hi; q[k].z = 0u; q[k].w = 0u; }
      }
#pragma unroll
      for (int k = 0; k < XB; ++k) ok = ok && q[k].x == tag_lo && q[k].y == tag_hi;
      if (ok) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > XCHG_TIMEOUT_TICKS) { *late = true; break; }  // every lane reaches an exit
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < XB; ++k)
      if (r0 + k < n) s += __longlong_as_double((long long)(((unsigned long long)q[k].w << 32) | q[k].z));
  }
  if (v == 0 && xi->stats != 0ull) {   // (fire and forget: nobody waits for these)
    unsigned long long* st = reinterpret_cast<unsigned long long*>(xi->stats);
    const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t0;
    __hip_atomic_fetch_add(st + 0, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(st + 1, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(st + 2, dt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (*late) __hip_atomic_fetch_add(st + 3, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return s;
}

// WLOG / word0 (round 5, "split summing blocks"): the block adds the 1 << WLOG words from word0 on -- all 32 (WLOG = 5), or the
// eight of ONE 128-byte line of every row (WLOG = 3), with eight lanes per column instead of 32.  The COLUMNS are the same
// either way (blockDim / 32 of them, whatever WLOG), and so are the order of a column's additions, the pairing of the
// columns and the tree over the pairs: a word's sum does not depend on which block, or how many lanes beside it, added it.
template <int WLOG>
__device__ __forceinline__ void sum_rows(__amdgpu_buffer_rsrc_t rows, unsigned int rows_off, int first, int end,
                                         unsigned long long seq, double (*lds_c)[EV_WORDS],
                                         __amdgpu_buffer_rsrc_t dst, unsigned int dst_off, bool dst_system,
                                         double* plain_dst, int* s_fail, const XchgInfo* __restrict__ xi = nullptr,
                                         unsigned long long xround = 0ull, int word0 = 0) {
  constexpr int W = 1 << WLOG;
  const int ncols = (int)blockDim.x >> 5;
  const int v = (int)(threadIdx.x & (unsigned)(W - 1)) + word0, c = (int)(threadIdx.x >> WLOG);
  const unsigned int tag_lo = (unsigned int)seq, tag_hi = (unsigned int)(seq >> 32);
  double s = 0.0;
  for (int b0 = first + c; c < ncols && b0 < end; b0 += SUM_BATCH * ncols) {
    u32x4 t[SUM_BATCH];
    // Slots that carry this launch's tag are kept; only the MISSING ones are requested again.  (Re-reading the whole
    // batch on every trip -- 241 rows x 32 slots = 123 KB per trip for the 200 k-point scan -- made a trip ~1.5 us,
    // and the sum trailed the last row by two trips: round-3 stamps, profiles/r03_stamps_prelaunch.txt.)
    unsigned int missing = 0u;
#pragma unroll
    for (int k = 0; k < SUM_BATCH; ++k) {
      t[k].x = tag_lo; t[k].y = tag_hi; t[k].z = 0u; t[k].w = 0u;   // (rows beyond `end` add an exact zero)
      if (b0 + k * ncols < end) missing |= 1u << k;
    }
    const unsigned long long t_wait0 = __builtin_amdgcn_s_memrealtime();
#ifdef NDT_STAMPS
    int ws_trip = 0;
#endif
    for (;;) {
      // EVERY load of a slot sits behind this clobber, inside the loop: with a copy of the first batch in front of
      // the loop the compiler took the in-loop load of slot 0 for the same value and never re-read it (the
      // summing block then waited 100 ms for its own row)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < SUM_BATCH; ++k)
        if (missing & (1u << k))
          t[k] = __builtin_amdgcn_raw_buffer_load_b128(rows, rows_off + ((unsigned int)(b0 + k * ncols) * EV_WORDS + v) * 16u, 0, AUX_AGENT);
#pragma unroll
      for (int k = 0; k < SUM_BATCH; ++k)
        if (t[k].x == tag_lo && t[k].y == tag_hi) missing &= ~(1u << k);
#ifdef NDT_STAMPS
      if (threadIdx.x < 64 && blockIdx.y == 0 && b0 == first + c) {   // wave 0 of the summing block, per poll trip
        const unsigned long long still = __ballot(missing != 0u);
        if (threadIdx.x == 0 && ws_trip < WS_TRIPS) {
          g_sumtrips[2 * ws_trip] = __builtin_amdgcn_s_memrealtime();
          g_sumtrips[2 * ws_trip + 1] = (unsigned long long)__popcll(still);
          g_sumtrips[2 * WS_TRIPS] = (unsigned long long)(ws_trip + 1);
        }
        ++ws_trip;
      }
#endif
      if (missing == 0u) break;
      if (__builtin_amdgcn_s_memrealtime() - t_wait0 > SUM_TIMEOUT_TICKS) {
        // a row that never comes (its block left without computing: a pre-launched grid whose blocks
        // timed out unevenly); exit anyway and say so: word 31 of an evaluation is 0 by construction,
        // the host re-evaluates a pre-launched pose and turns anything else into NDT_ERR_HIP
        *s_fail = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < SUM_BATCH; ++k)
      s += (missing & (1u << k)) ? 0.0 : __longlong_as_double((long long)(((unsigned long long)t[k].w << 32) | t[k].z));
  }
  NDT_WSTAMP(6);
  // the two columns of a wave first (lanes v and v + 32 hold the same word), then the waves' sums as a fixed tree of
  // depth 4 -- a chain of up to 32 dependent LDS reads and adds stood here (0.6 us behind the last row, stamps of round 5)
  s += __shfl_xor(s, W);   // (the neighbouring column: lanes v and v + 32 of a wave, or -- eight lanes per column -- v and v + 8)
  if ((threadIdx.x & (unsigned)W) == 0 && c < ncols) lds_c[c >> 1][v] = s;
  __syncthreads();
  NDT_WSTAMP(7);
  if (threadIdx.x < (unsigned)W) {
    const int nwv = ((int)blockDim.x + 63) >> 6;
    double q[MAX_WAVES];
#pragma unroll
    for (int k = 0; k < MAX_WAVES; ++k) q[k] = k < nwv ? lds_c[k][v] : 0.0;
#pragma unroll
    for (int w = 1; w < MAX_WAVES; w <<= 1)
#pragma unroll
      for (int k = 0; k + w < MAX_WAVES; k += 2 * w) q[k] += q[k + w];
    double t = q[0];
    if (v == EV_WORDS - 1 && *s_fail) t += 1.0;
    // (a block that adds a part of the words and gave up waiting cannot raise word 31 unless it is its own: its words
    // go out as NaN, which the host reads as a lost row -- evaluate() in ndt_evaluate.hip)
    if (W < EV_WORDS && v != EV_WORDS - 1 && *s_fail) t = __longlong_as_double(0x7ff8000000000000ll);
    if (W == EV_WORDS && xi != nullptr && !*s_fail) {
      // (a local sum that missed rows is NOT published: the host re-evaluates, peers wait for that)
      bool late = false;
      t = xchg_allsum(xi, xround, v, t, &late);
      if (__ballot(late) != 0ull && v == EV_WORDS - 1) t = 3.0;
    }
    if (plain_dst) plain_dst[v] = t;
    else store_slot(dst, dst_off + (unsigned int)v * 16u, seq, t, dst_system);
  }
  NDT_WSTAMP(8);
}

// Block sum of the 32 accumulator words -> one tagged row per block; the rows are added INSIDE
// the same launch: the block that draws the last ticket adds all rows in fixed order and
// writes the evaluation (grids above 2048 rows go through 32 group rows first).  No second
// kernel, no float atomics, and the summation tree does not depend on arrival order:
// results are bit-reproducible.  host_slots != nullptr: the evaluation is written as 32 tagged
// slots into pinned host memory for the host to poll; otherwise 32 plain doubles go to `out`.
// (the finishing wave of SIMD `fin`: its 64 lanes' sums of the 32 words, into lds_w[fin])
__device__ __forceinline__ void finish_wave_sums(double acc[EV_WORDS], int fin, int lane, double (*lds_w)[EV_WORDS]) {
  wave_reduce_scatter32(acc, lane);
  if ((lane & 1) == 0) lds_w[fin][lane >> 1] = acc[0];
}
__device__ __forceinline__ void block_reduce_finish(double (*lds_w)[EV_WORDS] /* the finishing waves' sums (finish_wave_sums) */,
                                                    int nfin, double* __restrict__ rows,
                                                    double* __restrict__ group_rows,
                                                    unsigned int* __restrict__ counters,
                                                    double* __restrict__ out,
                                                    unsigned long long* host_slots, unsigned long long seq,
                                                    int single_level_max, bool fixed_summer,
                                                    const XchgInfo* __restrict__ xi, unsigned long long xround,
                                                    int my_row /* this block's row */, int nb /* rows = computing blocks */,
                                                    bool dedicated /* a block without points adds the rows */, int mute_row = 0,
                                                    bool split_words = false /* rows 0 .. 3's blocks add eight words each */) {
  // (the summing stage's scratch lies over the waves' regions: every wave is past the barrier below by then)
  extern __shared__ int lds_dyn[];
  double (*lds_c)[EV_WORDS] = reinterpret_cast<double (*)[EV_WORDS]>(lds_dyn);
  __shared__ int s_last;
  __shared__ int s_fail;
  if (threadIdx.x == 0) s_fail = 0;
  NDT_WSTAMP(5);
  __syncthreads();
  NDT_WSTAMP(6);
  const __amdgpu_buffer_rsrc_t rrows = slots_rsrc(rows);
  if (threadIdx.x < EV_WORDS) {
    double sum = 0.0;
    for (int f = 0; f < nfin; ++f) sum += lds_w[f][threadIdx.x];
#ifdef NDT_TEST_SEAMS
    if (mute_row != my_row + 1)   // (seam: this block's row never arrives)
#endif
    store_slot(rrows, ((unsigned int)my_row * EV_WORDS + threadIdx.x) * 16u, seq, sum, false);
  }
  NDT_WSTAMP(7);
  NDT_STAMP(4);
#ifdef NDT_STAMPS
  if ((threadIdx.x >> 6) == 0) NDT_WSTAMP_DRAINED(8);
  ws_flush();
#endif
  if (dedicated) return;  // the summing block (summer_finish) polls the rows; this block is done
  int ngroups = 1;
  const __amdgpu_buffer_rsrc_t rgroups = slots_rsrc(group_rows);
  if (nb > single_level_max) {
    const int gsize = (nb + NGROUPS - 1) / NGROUPS;  // blocks per group
    ngroups = (nb + gsize - 1) / gsize;              // <= NGROUPS
    const int grp = my_row / gsize;
    const int first = grp * gsize, end = min(first + gsize, nb);
    // the row stores above were issued by wave 0, the wave that takes the ticket
    if (threadIdx.x == 0) s_last = ticket_is_last(counters + 1 + grp, (unsigned int)(end - first));
    __syncthreads();
    if (!s_last) return;
    sum_rows<5>(rrows, 0u, first, end, seq, lds_c, rgroups, (unsigned int)grp * EV_WORDS * 16u, false, nullptr, &s_fail);
    __syncthreads();
  }
  const bool two_level = nb > single_level_max;
  const int nrows = two_level ? ngroups : nb;
  if (fixed_summer && !two_level) {
    // Variant without tickets: the rows validate themselves (tag == this launch's sequence
    // number), so a FIXED block can add them -- block 0, the first one dispatched -- by polling
    // the tags; every other block is done once its row store is issued.  Takes the ticket's
    // atomic round trip out of the launch's critical path.  Block 0 depends on the others, never
    // the other way round, so a grid larger than the machine cannot dead-lock on it.
    if (split_words) {
      // (round 5) ... or the blocks of rows 0 .. SUMMER_SPLIT - 1, eight words -- one 128-byte line of every row -- each: a
      // grid whose point blocks fill the machine (the 128 x 1024 scan) has no unit for dedicated summing blocks, and one block
      // polling 1024 lines per trip behind its own points was the evaluation's tail.  Same columns, order and tree per word.
      if (my_row >= SUMMER_SPLIT) return;
      NDT_STAMP(5);
      sum_rows<3>(rrows, 0u, 0, nrows, seq, lds_c, slots_rsrc(host_slots), 0u, true, host_slots ? nullptr : out, &s_fail, nullptr, 0ull,
                  (EV_WORDS / SUMMER_SPLIT) * my_row);
      NDT_STAMP(6);
      NDT_STAMP(7);
      return;
    }
    if (my_row != 0) return;
    NDT_STAMP(5);
  } else {
    if (threadIdx.x == 0) s_last = ticket_is_last(counters, (unsigned int)nrows);
    NDT_STAMP(5);
    __syncthreads();
    if (!s_last) return;
  }
  sum_rows<5>(two_level ? rgroups : rrows, 0u, 0, nrows, seq, lds_c, slots_rsrc(host_slots), 0u, true,
              host_slots ? nullptr : out, &s_fail, xi, xround);
  NDT_STAMP(6);
  if (threadIdx.x <= ngroups && !(fixed_summer && !two_level))  // leave the tickets at zero for the next launch
    __hip_atomic_store(counters + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  NDT_STAMP(7);
}

// The dedicated summing block (EvalConsts::dedicated_summer): block 0 of the grid owns NO points.  It starts polling
// the rows' tags at once and keeps every slot that has arrived, so when the last computing block's row lands only
// that row is still to be fetched -- where a computing block that doubles as the summer first finishes its own
// points and then pays a whole memory round trip for all the rows (2.6 us behind the last row for the 200 k-point
// scan, profiles/r03_stamps_prelaunch.txt).  One spare compute unit (the scan's grid is 241 of 256).
// Round 5: SEVERAL summing blocks (EvalConsts::dedicated_summer = SUMMER_SPLIT).  Each polls ONE 128-byte line -- eight
// words -- of every row and writes its eight result slots itself.  Same columns, same order, same tree per word
// (sum_rows): the bits do not change.
__device__ __forceinline__ void summer_finish(double* __restrict__ rows, double* __restrict__ out,
                                              unsigned long long* host_slots, unsigned long long seq, int nrows,
                                              const XchgInfo* __restrict__ xi, unsigned long long xround, int nsummers) {
  extern __shared__ int lds_dyn[];   // (the summing block has no regions)
  double (*lds_c)[EV_WORDS] = reinterpret_cast<double (*)[EV_WORDS]>(lds_dyn);
  __shared__ int s_fail;
  if (threadIdx.x == 0) s_fail = 0;
  __syncthreads();
  NDT_STAMP(5);
  NDT_WSTAMP(5);
  if (nsummers == 4)
    sum_rows<3>(slots_rsrc(rows), 0u, 0, nrows, seq, lds_c, slots_rsrc(host_slots), 0u, true, host_slots ? nullptr : out, &s_fail,
                nullptr, 0ull, (EV_WORDS / 4) * (int)blockIdx.x);
  else if (nsummers == 8)
    sum_rows<2>(slots_rsrc(rows), 0u, 0, nrows, seq, lds_c, slots_rsrc(host_slots), 0u, true, host_slots ? nullptr : out, &s_fail,
                nullptr, 0ull, (EV_WORDS / 8) * (int)blockIdx.x);
  else
    sum_rows<5>(slots_rsrc(rows), 0u, 0, nrows, seq, lds_c, slots_rsrc(host_slots), 0u, true, host_slots ? nullptr : out, &s_fail,
                xi, xround);
  NDT_STAMP(6);
  NDT_STAMP(7);
#ifdef NDT_STAMPS
  NDT_WSTAMP_DRAINED(9);
  ws_flush();
#endif
}

// The explicit arguments of k_derivatives as the kernel-argument segment lays them out
// (natural alignment in declaration order, first argument at offset 0): the single-pose kernel
// reads its angle tables from the segment as memory.
struct DerivKernArgs {
  const float* sx;
  const float* sy;
  const float* sz;
  int n;
  GridGeom g;
  const int* cell2leaf;
  const VoxelRecord* rec;
  const float4* cent;
  PoseConsts pose;
  const PoseConsts* poses;
  EvalConsts ec;
  double* partials;
  unsigned int* counters;
  double* out;
  unsigned long long* flag;
  unsigned long long seq;
  const PoseMailbox* mbox;
  const XchgInfo* xinfo;
  unsigned long long xround;
  unsigned int* arrive_ctr;
  unsigned long long* arrived_host;
  const BuildGeom* geom_dev;
};
constexpr unsigned int KERNARG_TABLES_OFFSET = offsetof(DerivKernArgs, pose) + offsetof(PoseConsts, jang);
static_assert(offsetof(PoseConsts, hang) == offsetof(PoseConsts, jang) + 24 * sizeof(float) &&
              offsetof(AngleTables, hang) == 24 * sizeof(float), "jang / hang must be contiguous");

// NB: neighbourhood -- 0 DIRECT1, 1 DIRECT7, 2 KDTREE, 3 DIRECT26, 4 multi-grid union; 5 / 6: DIRECT1 / DIRECT7 on the
// 48-byte packed record table (the 27-cell neighbourhoods always read the 80-byte records)
// MBOX (single-pose only): a pre-launched evaluation -- the pose is not in the kernel arguments
// (it did not exist yet when the launch was enqueued) but arrives in *mbox, see PoseMailbox.
template <bool BATCH, int MODE, int NB, bool MBOX>
__global__ void __launch_bounds__(MAX_BLOCK)
k_derivatives(const float* __restrict__ sx, const float* __restrict__ sy, const float* __restrict__ sz, int n,
              GridGeom g_arg, const int* __restrict__ cell2leaf, const VoxelRecord* __restrict__ rec,
              const float4* __restrict__ cent, PoseConsts pose_arg, const PoseConsts* __restrict__ poses, EvalConsts ec,
              double* __restrict__ partials, unsigned int* __restrict__ counters, double* __restrict__ out,
              unsigned long long* flag, unsigned long long seq, const PoseMailbox* mbox,
              const XchgInfo* __restrict__ xinfo, unsigned long long xround,
              unsigned int* __restrict__ arrive_ctr, unsigned long long* arrived_host,
              const BuildGeom* __restrict__ geom_dev) {
  // geom_dev != nullptr: the launch was enqueued BEHIND the voxel-grid build that produces its grid, before the host
  // knew the geometry (ndt_evaluate.hip, evaluate(): the first evaluation of an align that follows a deferred build).  The
  // build's last block has left the geometry and its verdict in device memory; after a refused build every block leaves
  // at once -- nobody waits for anybody, the host repeats the evaluation the ordinary way.
  GridGeom g = g_arg;
  if (!BATCH && !MBOX && geom_dev != nullptr) {   // uniform
    if (geom_dev->status != BG_OK) return;
    g = geom_dev->g;
  }
  // R|t (12 dwords) stay in scalar registers; the 69 angle-table words are only needed
  // after the pair loop, so they are parked in LDS (81 live SGPRs would spill) and the
  // barrier that publishes them sits behind the memory-latency part of the kernel.
  NDT_STAMP(0);
#ifdef NDT_STAMPS
  ws_clear();
  NDT_WSTAMP(0);
  if (threadIdx.x == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {
    g_hwid[blockIdx.x * 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    g_hwid[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // HW_REG_XCC_ID
  }
#endif
  __shared__ AngleTables tab;
  __shared__ unsigned long long s_item[MAX_WAVES];   // a wave's pair sums (and table words) are in LDS: see "finishing waves"
  if ((threadIdx.x & 63u) == 0) s_item[threadIdx.x >> 6] = 0ull;   // (own flag, first thing)
  // (top bit set: never the zero just written; the low four bits are the wave's)
  const unsigned long long item_base =
      ((((seq * 0x9E3779B97F4A7C15ull) ^ ec.item_salt ^ ((unsigned long long)(blockIdx.y * gridDim.x + blockIdx.x) << 8))) | (1ull << 63)) & ~15ull;
  RigidRT rt;
  // The 69 table words are fetched by 69 lanes as one vector load each -- from the pose array,
  // or straight out of the kernel-argument segment -- at the very start, and written to LDS
  // only behind the pair phase: a wave-0 prologue of 69 scalar-to-vector moves and four
  // serialised s_load waits used to sit in front of every block's first point load.
  float tab_word = 0.0f;
  // exactly one source point per thread: the 32 accumulator words are only live from the
  // per-point expansion to the block reduction, not across the pair loop
  // (with a dedicated summing block the points start at block 1; block 0's threads own none)
  const bool summing_block = (int)blockIdx.x < ec.dedicated_summer;   // (0, 1, 4 or 8 of them, in front of the point blocks)
  // (position among the row's point blocks; the XCD of position 0 follows from the block's LINEAR workgroup id;
  // xcd_chunk: ndt_device.h)
  const int pos = (int)blockIdx.x - ec.dedicated_summer, npos = (int)gridDim.x - ec.dedicated_summer;
  const int chunk = (ec.xcd_count > 1 && !summing_block)
                        ? xcd_chunk(pos, npos, (int)(((unsigned int)blockIdx.y * gridDim.x + (unsigned int)ec.dedicated_summer) % (unsigned int)ec.xcd_count),
                                    ec.xcd_count, ec.xcd_stripe)
                        : pos;
  const int i = summing_block ? n : chunk * (int)blockDim.x + threadIdx.x;
  float x = 0.0f, y = 0.0f, z = 0.0f;
  if (MBOX) {
    // a pre-launched kernel has nothing to do until its pose arrives: its point does not depend
    // on the pose, so it can be fetched now (one memory round trip off the critical path)
    if (ec.mbox_preload && i < n) { x = sx[i]; y = sy[i]; z = sz[i]; }
    // (Pulling the whole record table into every XCD's L2 while waiting -- the L2s are cold at every
    // launch -- was measured too: 16.2-16.6 us per evaluation against 16.2-16.45,
    // profiles/r02_mailbox_prefetch_ab.txt.  Not kept.)
    __shared__ float s_rt[12];
    __shared__ int s_go;
    // "every block of this launch is resident": the block that arrives last says so in pinned host memory.  The
    // host only enqueues the NEXT evaluation's kernel on the other stream (where it can take compute units as
    // this launch's blocks leave, instead of waiting for the launch to end) once this launch needs no more
    // compute units -- otherwise the two could wait for each other.  Off the critical path: nothing waits for it.
    if (arrive_ctr != nullptr && threadIdx.x == 0) {
      const unsigned int t = __hip_atomic_fetch_add(arrive_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == gridDim.x * gridDim.y - 1u) {
        __hip_atomic_store(arrive_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(arrived_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (ec.mbox_tagged) {
      // the pose as 82 self-validating 8-byte granules {tag, word} (PoseMailbox): lane k < 41 of wave 0
      // watches granules 2k and 2k + 1 with one 16-byte load; when every tag is this launch's the
      // words are already in registers.  Granule 81 is the control word (lane 40's second).
      if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool mine = lane < MBOX_GRANULES / 2;
        const __amdgpu_buffer_rsrc_t rm = slots_rsrc(mbox);
        const unsigned int tag = mbox_tag32(seq);
        u32x4 v;
        v.x = v.y = v.z = v.w = 0u;
        int go = -1;  // timed out
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
          asm volatile("" ::: "memory");  // re-issued on every trip
          if (mine) v = __builtin_amdgcn_raw_buffer_load_b128(rm, (unsigned int)offsetof(PoseMailbox, gran) + lane * 16u, 0, AUX_SYSTEM);
          const bool ok = !mine || (v.x == tag && v.z == tag);
          const bool quit = lane == MBOX_GRANULES / 2 - 1 && v.z == tag && v.w == MBOX_CTRL_QUIT;
          if (__ballot(quit) != 0ull) { go = 0; break; }
          if (__ballot(ok) == ~0ull) { go = 1; break; }
          if (__builtin_amdgcn_s_memrealtime() - t0 > MBOX_TIMEOUT_TICKS) break;  // every wave reaches an exit
          __builtin_amdgcn_s_sleep(1);
        }
        if (go == 1 && mine) {
          const int w0 = 2 * lane, w1 = 2 * lane + 1;
          if (w0 < 12) s_rt[w0] = __uint_as_float(v.y); else tab.jang[w0 - 12] = __uint_as_float(v.y);  // runs on into hang[]
          if (w1 < 12) s_rt[w1] = __uint_as_float(v.w); else if (w1 < 81) tab.jang[w1 - 12] = __uint_as_float(v.w);
        }
        if (lane == 0) s_go = go;
      }
      __syncthreads();
    } else {
      if (threadIdx.x == 0) {
        int go = -1;  // timed out
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
          const unsigned long long v = __hip_atomic_load(&mbox->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v == seq) { go = 1; break; }
          if (v == (seq | MBOX_QUIT)) { go = 0; break; }
          if (__builtin_amdgcn_s_memrealtime() - t0 > MBOX_TIMEOUT_TICKS) break;  // every wave reaches an exit
          __builtin_amdgcn_s_sleep(1);
        }
        s_go = go;
      }
      __syncthreads();
    }
    const int go = s_go;
#ifdef NDT_STAMPS
    if (go == 1 && threadIdx.x == 0 && blockIdx.x < 4096) {
      g_mstamps[blockIdx.x * 2] = g_stamps[blockIdx.x * 8];   // (this launch's entry stamp)
      g_mstamps[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (go <= 0) {
      // quit: nothing to do.  timed out: say so in the result slots (word 31 = 2), the host
      // evaluates this pose through an ordinary launch instead
      if (go < 0 && flag != nullptr) {
        if (ec.doubling_split) {   // the blocks of rows 0 .. 3 add eight words each: each says so in its own slots (see below)
          if (chunk < SUMMER_SPLIT && (int)threadIdx.x < EV_WORDS / SUMMER_SPLIT) {
            const unsigned int w = (unsigned int)((EV_WORDS / SUMMER_SPLIT) * chunk) + threadIdx.x;
            store_slot(slots_rsrc(flag), w * 16u, seq, w == (unsigned int)EV_FAIL ? 2.0 : __longlong_as_double(0x7ff8000000000000ll), true);
          }
        } else if (ec.dedicated_summer <= 1) {
          if (blockIdx.x == 0 && threadIdx.x < EV_WORDS)
            store_slot(slots_rsrc(flag), threadIdx.x * 16u, seq, threadIdx.x == EV_FAIL ? 2.0 : 0.0, true);
        } else if (summing_block && (int)threadIdx.x < EV_WORDS / ec.dedicated_summer) {
          // every summing block says so in its OWN slots: word 31's owner raises it, the others' words go out as NaN
          // (which the host reads as "nothing usable": should a sibling have caught the pose and published, the
          // evaluation is still repeated)
          const unsigned int w = (unsigned int)(EV_WORDS / ec.dedicated_summer) * blockIdx.x + threadIdx.x;
          store_slot(slots_rsrc(flag), w * 16u, seq, w == (unsigned int)EV_FAIL ? 2.0 : __longlong_as_double(0x7ff8000000000000ll), true);
        }
      }
      return;
    }
    if (!ec.mbox_tagged) {
      if (threadIdx.x < 81) {  // the pose was published before seq (PCIe keeps posted writes in order)
        const unsigned int w = __hip_atomic_load(&mbox->words[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x < 12) s_rt[threadIdx.x] = __uint_as_float(w);
        else tab.jang[threadIdx.x - 12] = __uint_as_float(w);  // runs on into hang[]
      }
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) rt.R[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_rt[k])));
#pragma unroll
    for (int k = 0; k < 3; ++k) rt.t[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_rt[9 + k])));
  } else if (BATCH) {
    const PoseConsts& pg = poses[blockIdx.y];
#pragma unroll
    for (int k = 0; k < 9; ++k) rt.R[k] = pg.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) rt.t[k] = pg.t[k];
    if (threadIdx.x < 69) tab_word = pg.jang[threadIdx.x];  // jang[24] and hang[45] are contiguous
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) rt.R[k] = pose_arg.R[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) rt.t[k] = pose_arg.t[k];
    if (threadIdx.x < 69) {
      const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();  // constant -> generic address space
      tab_word = reinterpret_cast<const float*>(ka + KERNARG_TABLES_OFFSET)[threadIdx.x];
    }
  }
  if (summing_block) {   // uniform
    double* sbase = partials + (size_t)blockIdx.y * (gridDim.x + NGROUPS) * ROW_WORDS;
    summer_finish(sbase + (size_t)NGROUPS * ROW_WORDS, out + (size_t)blockIdx.y * EV_WORDS,
                  flag ? flag + (size_t)blockIdx.y * ROW_WORDS : nullptr, seq, (int)gridDim.x - ec.dedicated_summer,
                  BATCH ? nullptr : xinfo, xround, ec.dedicated_summer);
    return;
  }
  PairAcc a;
  a.w[0] = a.w[1] = a.w[2] = 0.0;
#pragma unroll
  for (int k = 0; k < 6; ++k) a.S[k] = 0.0;
  a.score = 0.0; a.best = 0.0; a.npairs = 0;
  extern __shared__ int lds_dyn[];   // one region per wave: the KD candidate list first, the hand-over of its pair sums afterwards
  constexpr bool KD = NB >= 2 && NB <= 4;
  // (the wave's number as a scalar: nothing but the thread id and the pair sums stays in vector registers through the
  // pair phase -- the kernel sits at the 128-VGPR edge there)
  const int lane = (int)(threadIdx.x & 63u), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* const regions = reinterpret_cast<char*>(lds_dyn);
  if (!(MBOX && ec.mbox_preload) && i < n) { x = sx[i]; y = sy[i]; z = sz[i]; }
  // A finishing wave expands its OWN points first, with its SIMD's other waves still in their pair phase or -- the last
  // wave of the fullest SIMD, which expands nobody else -- at the very end of the block with the SIMD to itself.  The
  // factors of an expansion that depend on the pose and the point only, not on the pairs (A(x), and the
  // second-derivative products), are computed HERE, while the SIMD waits for the first cell lookups, and parked in the
  // wave's own region (a finishing wave hands nothing over and needs its point for nothing else; the 27-cell modes
  // keep their candidate list there and expand from the tables as before).
  // (EVERY wave doing so for whoever expands its points -- the finishing waves then need neither tables nor points --
  // was measured too, with regions of 11.5 KB: 0.6 % slower, profiles/r05_pre_all_ab.txt.  The stretch in which the
  // SIMDs "wait" is not idle enough to take 92 more instructions per wave for nothing.)
  const int nw = (int)(blockDim.x >> 6);
  constexpr bool PRE = !KD && MODE != 3;
  const bool finishing = (int)((ec.fin_waves >> (4 * (wave & 3))) & 15u) == wave;   // wave-uniform
  constexpr int region_bytes = wave_region_bytes(KD);
  char* const my_region = regions + wave * region_bytes;
  NDT_STAMP(1);
  NDT_WSTAMP_DRAINED(1);
  float* const pre = reinterpret_cast<float*>(my_region);
  const bool precomputes = PRE && finishing;   // wave-uniform
  if (!precomputes) store_point(my_region, lane, x, y, z);
  auto precompute = [&]() {   // (in the shadow of the cell lookups: point_pairs)
    if (!precomputes) return;
    // (a non-finite point -- it never has neighbours, ref :573, so w = S = 0 -- as the origin: it expands to exact zeros)
    const bool pfin = isfinite(x) && isfinite(y) && isfinite(z);
    const float px = pfin ? x : 0.0f, py = pfin ? y : 0.0f, pz = pfin ? z : 0.0f;
    if (MBOX) {          // out of LDS, lane k holding words k and 64 + k (as angle_tables_to_sgprs)
      const float* w = tab.jang;
      const float wa = w[lane], wb = w[64 + (lane < 5 ? lane : 0)];
      precompute_geo<MODE>(pre, lane, [&](int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(k < 64 ? wa : wb), k & 63)); },
                           px, py, pz);
    } else if (BATCH) {  // out of the pose array
      const PoseConsts& pg = poses[blockIdx.y];
      precompute_geo<MODE>(pre, lane, [&](int k) { return pg.jang[k]; }, px, py, pz);   // jang[24] and hang[45] are contiguous
    } else {             // out of the kernel arguments
      precompute_geo<MODE>(pre, lane, [&](int k) { return k < 24 ? pose_arg.jang[k < 24 ? k : 0] : pose_arg.hang[k >= 24 ? k - 24 : 0]; }, px, py, pz);
    }
  };
  // (the table words were requested before the point: they have arrived with it; a pre-launched kernel has had its
  // tables in LDS since it was released)
  if (!MBOX && threadIdx.x < 69) tab.jang[threadIdx.x] = tab_word;  // runs on into hang[]: the two arrays are contiguous
  if (KD) {
    // every lane takes part (wave-wide trip count): lanes beyond n run with nothing to add
    point_pairs_kd<MODE, NB == 2 || NB == 4, NB == 4>(a, x, y, z, g, cell2leaf, rec, cent, rt, ec,
                                                      reinterpret_cast<int*>(my_region + PART_XYZ_BYTES), i < n);
  } else {
    point_pairs<MODE, NB == 1 || NB == 6, NB >= 5>(a, x, y, z, g, cell2leaf, rec, rt, ec, i < n, precompute);
  }
  NDT_STAMP(2);
  NDT_WSTAMP(2);
  // ---- hand-over to the finishing waves (see above) ----
  // (the lane number once more, from the exec mask this time: a value derived from the thread id in front of the pair
  // phase would be kept in a register through it -- or spilled around it, as the multi-grid kernels did)
  const int lane2 = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int nfin = nw < 4 ? nw : 4;
  const int simd = wave & 3;                                   // (the SIMD this wave is taken to sit on)
  if (!finishing) store_partials<MODE>(region_part(my_region), region_npairs(my_region, lane2), lane2, a);
  publish_item(s_item, item_base, wave, lane2);
  NDT_WSTAMP(3);
  __shared__ double lds_w[4][EV_WORDS];
  if (finishing) {   // wave-uniform
    double acc[EV_WORDS];
    bool ok = true;
    if (!MBOX) {    // the angle tables are in LDS once waves 0 and 1 have published
      ok = wait_item(s_item, item_base, 0) && ok;
      if (nw > 1) ok = wait_item(s_item, item_base, 1) && ok;
    }
    // its own points: the pair sums straight from registers
    if (PRE) expand_point<MODE, false>(acc, RegSource{a}, PreGeo{pre, lane2});
    const unsigned int owners = ec.item_owner, fins = ec.fin_waves;
    if (!PRE || wave != ec.lone_wave) {   // (lone_wave: the finishing wave that has nobody else's)
      float T[69];
      angle_tables_to_sgprs(T, tab, lane2);
      if (!PRE) {
        load_point(my_region, lane2, x, y, z);
        expand_point<MODE, false>(acc, RegSource{a}, TableGeo{T, x, y, z});
      }
      for (int it = 0; it < nw; ++it) {                           // the other waves' in a fixed order
        if ((int)((owners >> (2 * it)) & 3u) != simd || (int)((fins >> (4 * (it & 3))) & 15u) == it) continue;
        ok = wait_item(s_item, item_base, it) && ok;
        char* region = regions + it * region_bytes;
        float qx, qy, qz;
        load_point(region, lane2, qx, qy, qz);
        expand_point<MODE, true>(acc, LdsSource{region_part(region), region_npairs(region, lane2), lane2}, TableGeo{T, qx, qy, qz});
      }
    }
    if (!ok && lane2 == 0) acc[EV_FAIL] += 1.0;   // (never seen: a sibling wave that did not publish)
    NDT_WSTAMP(4);
    finish_wave_sums(acc, simd, lane2, lds_w);
  }
  NDT_STAMP(3);
#ifdef NDT_STAMPS
  if (!finishing) NDT_WSTAMP(4);
#endif
  double* base = partials + (size_t)blockIdx.y * (gridDim.x + NGROUPS) * ROW_WORDS;
  block_reduce_finish(lds_w, nfin, base + (size_t)NGROUPS * ROW_WORDS, base, counters + blockIdx.y * COUNTERS_PER_POSE,
                      out + (size_t)blockIdx.y * EV_WORDS,
                      flag ? flag + (size_t)blockIdx.y * ROW_WORDS : nullptr,  // pose y's 32 host slots
                      seq, ec.single_level_max,
                      ec.fixed_summer != 0, BATCH ? nullptr : xinfo, xround, chunk,
                      (int)gridDim.x - ec.dedicated_summer, ec.dedicated_summer != 0, ec.mute_row, !BATCH && ec.doubling_split != 0);
}

__global__ void __launch_bounds__(256) k_transform(const float* __restrict__ sx, const float* __restrict__ sy,
                                                  const float* __restrict__ sz, int n, PoseConsts P,
                                                  float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = sx[i], y = sy[i], z = sz[i];
  out[3 * i + 0] = P.R[0] * x + (P.R[1] * y + (P.R[2] * z + P.t[0]));
  out[3 * i + 1] = P.R[3] * x + (P.R[4] * y + (P.R[5] * z + P.t[1]));
  out[3 * i + 2] = P.R[6] * x + (P.R[7] * y + (P.R[8] * z + P.t[2]));
}

}  // namespace

size_t derivs_partials_words(size_t n_src, int K, int cus) {
  return (size_t)K * ((size_t)derivs_grid_blocks(n_src, K, cus) + NGROUPS) * ROW_WORDS;
}
int derivs_counters_per_pose() { return COUNTERS_PER_POSE; }

int derivs_read_stamps(unsigned long long* out, int nblocks) {
#ifdef NDT_STAMPS
  if (nblocks > 4096) nblocks = 4096;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * nblocks) != hipSuccess) return -1;
  // hw ids follow the stamps: nblocks x 2 uint32 packed into nblocks uint64
  if (hipMemcpyFromSymbol(out + (size_t)8 * nblocks, HIP_SYMBOL(g_hwid), sizeof(unsigned int) * 2 * nblocks) != hipSuccess) return -1;
  // ... and {entry, pose seen} of the last pre-launched launch that computed: 2 x nblocks uint64
  return hipMemcpyFromSymbol(out + (size_t)9 * nblocks, HIP_SYMBOL(g_mstamps), sizeof(unsigned long long) * 2 * nblocks) == hipSuccess ? nblocks : -1;
#else
  (void)out; (void)nblocks;
  return 0;
#endif
}

// per-wave stamps of the last launch: out = [nblocks][16 waves][10 stamps] u64, then [nblocks][16] hw ids packed as u32
// pairs, then the summing block's poll trips: 64 x {time, lanes still missing} + the trip count
int derivs_read_wave_stamps(unsigned long long* out, int nblocks) {
#ifdef NDT_STAMPS
  if (nblocks > WS_BLOCKS) nblocks = WS_BLOCKS;
  const size_t nw = (size_t)nblocks * WS_WAVES;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * nw * WS_N) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + nw * WS_N, HIP_SYMBOL(g_whwid), sizeof(unsigned int) * nw) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + nw * WS_N + nw / 2, HIP_SYMBOL(g_sumtrips), sizeof(unsigned long long) * (2 * WS_TRIPS + 1)) != hipSuccess) return -1;
  return nblocks;
#else
  (void)out; (void)nblocks;
  return 0;
#endif
}

// Threads per block.  512 (8 waves) measured best on MI355X from 50k to 4M points: with
// 126 VGPRs two such blocks fill a CU (4 waves/SIMD); smaller blocks multiply the partial
// rows of the in-kernel final sum, larger ones make every wave wait on wider barriers
// (sweep 128..832 in profiles/r01_block_sweep.txt).  One exception, single-pose launches of
// 131k..262k points (the 200k-point scan of the headline workload): the block is sized so that
// the grid is at most one block per CU (256 CUs) -- 200 000 points: 832 threads, 241 blocks --
// which keeps every CU at <= 4 waves/SIMD and leaves <= 256 rows for the final sum
// (profiles/r02_block_sweep.txt: 16.6 us at 832, 17.2 at 1024, 20.3 at 512).
// ndt_tuning::deriv_block overrides it for tuning.
// (the compute units of the handle's device come in as `cus`: two engines on differently partitioned devices in one
// process must not share a block shape)
static inline int cus_or_default(int cus) { return cus > 0 ? cus : 256; }

namespace {
// (ndt_tuning, include/ndt_hip.h: A/B switches, none of them read from the environment by the production library)
int deriv_single_level_max() { return tuning().deriv_single_level_max; }
int deriv_fixed_summer() { return tuning().deriv_summer; }       // 0 = ticket + last block, 1 = a fixed block polls the rows
bool deriv_dedicated_enabled() { return tuning().deriv_dedicated != 0; }
}  // namespace

int derivs_block_threads(size_t n_src, int K, int cus) {
  const int g_compute_units = cus_or_default(cus);
  const int forced = tuning().deriv_block;  // multiple of 64, 64..1024 (checked by ndt_set_tuning)
  if (forced) return forced;
  // (one compute unit is left to the dedicated summing block)
  const size_t kCUs = (size_t)std::max(2, g_compute_units - (deriv_dedicated_enabled() && deriv_fixed_summer() ? 1 : 0));
  // (up to 512 points per compute unit the 512-thread shape stays: two such blocks share a CU, and a scan of exactly
  // 128 x 1024 points keeps the same partition -- hence the same sums, bit for bit -- single-pose and batched)
  if (K == 1 && n_src > (size_t)512 * (size_t)g_compute_units && n_src <= (size_t)MAX_BLOCK * kCUs) {
    const size_t per_cu = (n_src + kCUs - 1) / kCUs;
    return (int)(((per_cu + 63) / 64) * 64);
  }
  // Small sources -- a rank's share of the scan in a multi-GPU job, a sparse scan -- spread over more compute units in
  // smaller blocks: about 196 blocks of 256 .. 512 threads (25 k points: 98 blocks of 256 = 8.2 us per launch against
  // 9.9 with 49 of 512; 50 k: 9.2 against 10.1; 100 k stays at 196 x 512, where 256-thread blocks would put two on
  // a compute unit: 13.8 against 10.4; profiles/r03_small_source_blocks.txt).  For every K, so that a batched launch
  // keeps the partition -- hence the sums, bit for bit -- of the single-pose one.
  if (n_src <= (size_t)196 * 512) {
    const size_t per = (n_src + 195) / 196;
    const size_t bt = ((per + 63) / 64) * 64;
    return (int)std::min<size_t>(512, std::max<size_t>(256, bt));
  }
  return 512;
}

// blocks that own points
static int derivs_point_blocks(size_t n_src, int K, int cus) {
  const size_t bt = (size_t)derivs_block_threads(n_src, K, cus);
  size_t blocks = (n_src + bt - 1) / bt;  // one point per thread
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// 1: block 0 of the grid owns no points and adds the rows (single-level grids with the fixed summer)
static int derivs_dedicated_summer(size_t n_src, int K, int cus) {
  if (!deriv_dedicated_enabled() || deriv_fixed_summer() == 0) return 0;
  const int pb = derivs_point_blocks(n_src, K, cus);
  if (pb > deriv_single_level_max()) return 0;
  // SUMMER_SPLIT of them where the point blocks leave that many compute units (the 200 k-point scan: 241 + 4 of 256),
  // single-pose launches only: each polls one 128-byte line of every row (summer_finish)
  {
    const int want = tuning().deriv_summer_split == 1 ? SUMMER_SPLIT : tuning().deriv_summer_split;   // (4 | 8: A/B)
    if (K == 1 && want > 1 && pb + want <= cus_or_default(cus)) return want;
  }
  // The summing block needs a compute unit of its own.  A single-pose grid whose point blocks fill the machine EXACTLY
  // -- 256 blocks of 512 threads: a 128 x 1024 Ouster scan, the size of C2 and of every C5 frame -- would push one
  // compute unit to two point blocks (4 waves per SIMD instead of 2), and the launch waits for that unit: 15.5 us against
  // 11.2 with block 0 doubling as the summer (profiles/r04_c2_block_shapes.txt).  Same rows, same order: the same bits.
  const int g_compute_units = cus_or_default(cus);
  if (K == 1 && pb <= g_compute_units && pb + 1 > g_compute_units) return 0;
  return 1;
}

// Which finishing wave expands which wave's points (k_derivatives, "finishing waves").  The hardware places the waves
// of a block round-robin on the four SIMDs of its compute unit (observed on every block of every launch stamped,
// profiles/r05_stamps_per_wave.txt), so wave i is taken to sit on SIMD i mod 4.
//   fin_waves: 4 bits per SIMD -- its finishing wave (15: none): the LAST wave of a SIMD that holds one wave more than
//              others do (it finishes its pairs last anyway and expands itself from registers), the FIRST wave elsewhere
//              (it is through its pairs first and works the others' items off as they arrive);
//   owners:    2 bits per wave -- the SIMD whose finishing wave expands that wave's points: its own SIMD's first; then
//              items move off the fullest SIMD, the earliest-finishing ones first, for as long as that lowers the largest
//              load, counted in VALU instructions (pair phase + LDS hand-over, expansion, reduce-scatter).
// 13 waves: wave 12 expands itself, waves 1 / 2 / 3 take four items each.  Depends on the block shape only: fixed
// tables, a fixed order of additions.
//   lone:      a finishing wave that is left with no item but its own (13 waves: wave 12), -1 if there is none: it
//              does not fetch the angle tables behind its pair phase.
void derivs_item_owners(int threads, unsigned int* owners_out, unsigned int* fin_waves_out, int* lone_out) {
  const int nw = threads / 64, nfin = nw < 4 ? nw : 4;
  int fin_of_simd[4] = {15, 15, 15, 15}, p[4] = {0, 0, 0, 0}, e[4] = {0, 0, 0, 0}, owner[MAX_WAVES];
  for (int i = 0; i < nw; ++i) { ++p[i % 4]; ++e[i % 4]; owner[i] = i % 4; }
  const int pmin = nw >= 4 ? nw / 4 : 1;
  for (int q = 0; q < nfin; ++q) fin_of_simd[q] = (nw >= 4 && p[q] > pmin) ? q + 4 * (p[q] - 1) : q;
  constexpr int P = 710, E = 245, R = 140;
  for (;;) {
    int m = 0, k = 0, load[4];
    for (int q = 0; q < nfin; ++q) load[q] = p[q] * P + e[q] * E + R;
    for (int q = 1; q < nfin; ++q) { if (load[q] > load[m]) m = q; if (load[q] < load[k]) k = q; }
    if (load[k] + E >= load[m]) break;
    int it = -1;
    for (int i = 0; i < nw && it < 0; ++i)
      if (owner[i] == m && i != fin_of_simd[i % 4]) it = i;
    if (it < 0) break;
    owner[it] = k;
    --e[m];
    ++e[k];
  }
  unsigned int ob = 0u, fb = 0u;
  for (int i = 0; i < nw; ++i) ob |= (unsigned int)owner[i] << (2 * i);
  for (int q = 0; q < 4; ++q) fb |= (unsigned int)fin_of_simd[q] << (4 * q);
  *owners_out = ob;
  *fin_waves_out = fb;
  if (lone_out != nullptr) {
    *lone_out = -1;
    for (int q = nfin - 1; q >= 0; --q)
      if (e[q] == 1) *lone_out = fin_of_simd[q];   // (the lowest SIMD if several: the others go through an empty item loop)
  }
}

int derivs_grid_blocks(size_t n_src, int K, int cus) { return derivs_point_blocks(n_src, K, cus) + derivs_dedicated_summer(n_src, K, cus); }

void launch_derivatives(const float* sx, const float* sy, const float* sz, size_t n_src,
                        const GridGeom& g, const int* cell2leaf, const VoxelRecord* rec, const float* cent4,
                        const PoseConsts& pose, const PoseConsts* d_poses, int K,
                        const EvalConsts& ec, double* d_partials, unsigned int* d_counters,
                        double* d_out, hipStream_t s, unsigned long long* d_flag,
                        unsigned long long seq, const PoseMailbox* d_mbox, const XchgInfo* d_xinfo,
                        unsigned long long xround, unsigned int* d_arrive_ctr, unsigned long long* d_arrived_host,
                        hipEvent_t ev_start, hipEvent_t ev_stop, const BuildGeom* d_geom) {
  const int cus = cus_or_default(ec.compute_units);
  int blocks = derivs_grid_blocks(n_src, d_poses ? K : 1, cus);
  const int threads = derivs_block_threads(n_src, d_poses ? K : 1, cus);
  const int mode = ec.score_only ? 3 : (!ec.need_hessian ? 0 : (ec.gauss_newton ? 2 : 1));
  EvalConsts ecl = ec;
  ecl.single_level_max = deriv_single_level_max();
  ecl.fixed_summer = deriv_fixed_summer();
  ecl.dedicated_summer = derivs_dedicated_summer(n_src, d_poses ? K : 1, cus);
  if (ecl.dedicated_summer > 1 && d_xinfo != nullptr) {   // the in-kernel cross-rank exchange is one block's (xchg_allsum)
    blocks -= ecl.dedicated_summer - 1;
    ecl.dedicated_summer = 1;
  }
  ecl.doubling_split = 0;
  if (ec.safe_sum) {
    // Same rows in the same order, added by the block that draws the last ticket: by then every row has been issued,
    // so nothing in the launch waits for a block that is not resident (a device shared with other processes).
    ecl.fixed_summer = 0;
    ecl.dedicated_summer = 0;
    blocks = derivs_point_blocks(n_src, d_poses ? K : 1, cus);
  }
  // XCD-aware chunk assignment (xcd_chunk, ndt_device.h).  gfx950 has 32 compute units per XCD: 8 XCDs on a whole MI355X,
  // one in a CPX partition (nothing to do there).  Grids that are resident at once (at most one block per compute unit)
  // take the whole row as one stripe.  ndt_tuning::deriv_xcd: 0 = off, 1 = resident single-pose grids only (default), 2 = also
  // larger grids and batched launches, in stripes of one residency round (blocks per CU from the block size: 16 waves
  // per CU).  Measured in round 4 (profiles/r04_xcd_stripes_ab.txt): the stripes LOSE -- C3 400 k / 800 k points 25.8 /
  // 40.7 us against 24.1 / 39.1 with chunk = block id, C3-wide 30.0 / 48.9 against 28.1 / 46.1, the SVN Stage-1 launch
  // (20 poses x 131 k points) 87.7 against 83.3 -- so they stay a knob.
  const int xcd_mode = tuning().deriv_xcd;
  const int nxcd = std::max(1, cus / 32);
  const int point_blocks = blocks - ecl.dedicated_summer;
  // no unit to spare for summing blocks (the 128 x 1024 scan): the blocks of rows 0 .. 3 add eight words each
  if (!d_poses && d_xinfo == nullptr && !ec.safe_sum && ecl.dedicated_summer == 0 && ecl.fixed_summer != 0 && tuning().deriv_summer_split != 0 &&
      point_blocks >= SUMMER_SPLIT && point_blocks <= ecl.single_level_max)
    ecl.doubling_split = 1;
  const int per_cu = std::max(1, 1024 / threads);
  const bool resident = blocks <= cus * per_cu + 1 && !d_poses;
  ecl.xcd_count = 0;
  ecl.xcd_stripe = 0;
  if (nxcd > 1 && point_blocks >= 2 * nxcd && xcd_mode != 0) {
    if (!d_poses && blocks <= cus + 1) {
      ecl.xcd_count = nxcd;                 // the whole grid at once: one stripe (round 3's case)
    } else if (xcd_mode >= 2) {
      ecl.xcd_count = nxcd;
      ecl.xcd_stripe = resident ? 0 : std::max(1, (cus / nxcd) * per_cu);
    }
  }
  // DIRECT7 / DIRECT1 only: the union's leaves are chained through VoxelRecord::pad, and in the 27-cell neighbourhoods
  // the format (as a run-time flag) cost more than the shorter fetch gave back (KDTREE 21.9 -> 22.5 us, DIRECT26 29.1 -> 29.7)
  if (ec.multigrid || ec.kdtree || ec.direct26) ecl.packed = 0;
  const int nb = ec.multigrid ? 4 : (ec.kdtree ? 2 : (ec.direct26 ? 3 : (ec.direct7 ? (ecl.packed ? 6 : 1) : (ecl.packed ? 5 : 0))));
  derivs_item_owners(threads, &ecl.item_owner, &ecl.fin_waves, &ecl.lone_wave);
  // one LDS region per wave: the 27-cell modes' candidate list, then the wave's hand-over to the finishing waves
  // (the summing stage's 4 KB of scratch lie over them)
  size_t dyn_lds = std::max<size_t>((size_t)(threads / 64) * (size_t)wave_region_bytes(nb >= 2 && nb <= 4), (size_t)MAX_WAVES * EV_WORDS * sizeof(double));
  // ONE block of a single-pose launch per compute unit.  Two 8-wave blocks fit a unit, and the hardware deals workgroups to the
  // shader engines in turn whatever the engines' number of (harvested) compute units: of the 256 blocks of a 128 x 1024 scan two
  // landed beside a sibling on every box stamped, took twice as long, and the evaluation waited 2.7 us for their rows
  // (tools/stamps_prelaunch.py, profiles/r05_c2_stragglers.txt).  A block that asks for more than half a unit's LDS has the unit to
  // itself; the kernel pre-launched for the next evaluation then moves in as this one's blocks leave, as it always has for the
  // 13-wave blocks of the 200 k-point scan.
  if (!d_poses && ec.own_units && d_xinfo == nullptr && tuning().deriv_one_block_per_cu != 0 && blocks <= cus && dyn_lds <= (size_t)80 * 1024)
    dyn_lds = (size_t)80 * 1024 + 512;
  // ev_start / ev_stop: events attached to THIS dispatch (hipExtLaunchKernel): they carry the kernel's own begin and
  // end timestamps, what rocprofv3 reports -- events recorded around the launch include ~2.4 us of dispatch
#define NDT_LAUNCH2(B, M, NBH, MB, GY, FLAG, SEQ)                                                            \
  do {                                                                                                       \
    if (ev_start != nullptr)                                                                                 \
      hipExtLaunchKernelGGL((k_derivatives<B, M, NBH, MB>), dim3(blocks, GY), dim3(threads), (std::uint32_t)dyn_lds, s, ev_start, ev_stop, 0u, sx, sy, sz,  \
                            (int)n_src, g, cell2leaf, rec, reinterpret_cast<const float4*>(cent4), pose, d_poses, ecl, d_partials, d_counters, d_out,         \
                            FLAG, SEQ, d_mbox, d_xinfo, xround, d_arrive_ctr, d_arrived_host, d_geom);        \
    else                                                                                                     \
      hipLaunchKernelGGL((k_derivatives<B, M, NBH, MB>), dim3(blocks, GY), dim3(threads), dyn_lds, s, sx, sy, sz,  \
                         (int)n_src, g, cell2leaf, rec, reinterpret_cast<const float4*>(cent4), pose, d_poses, ecl, d_partials, d_counters, d_out,         \
                         FLAG, SEQ, d_mbox, d_xinfo, xround, d_arrive_ctr, d_arrived_host, d_geom);           \
  } while (0)
#define NDT_LAUNCH(B, M, NBH, GY, FLAG, SEQ)                                  \
  do {                                                                         \
    if (!B && d_mbox != nullptr) NDT_LAUNCH2(false, M, NBH, true, GY, FLAG, SEQ);  \
    else NDT_LAUNCH2(B, M, NBH, false, GY, FLAG, SEQ);                         \
  } while (0)
#define NDT_LAUNCH_MODE(B, NBH, GY, FLAG, SEQ)                 \
  do {                                                          \
    if (mode == 0) NDT_LAUNCH(B, 0, NBH, GY, FLAG, SEQ);        \
    else if (mode == 1) NDT_LAUNCH(B, 1, NBH, GY, FLAG, SEQ);   \
    else if (mode == 2) NDT_LAUNCH(B, 2, NBH, GY, FLAG, SEQ);   \
    else NDT_LAUNCH(B, 3, NBH, GY, FLAG, SEQ);                  \
  } while (0)
#define NDT_LAUNCH_NB(B, GY, FLAG, SEQ)                        \
  do {                                                          \
    if (nb == 0) NDT_LAUNCH_MODE(B, 0, GY, FLAG, SEQ);          \
    else if (nb == 1) NDT_LAUNCH_MODE(B, 1, GY, FLAG, SEQ);     \
    else if (nb == 2) NDT_LAUNCH_MODE(B, 2, GY, FLAG, SEQ);     \
    else if (nb == 3) NDT_LAUNCH_MODE(B, 3, GY, FLAG, SEQ);     \
    else if (nb == 4) NDT_LAUNCH_MODE(B, 4, GY, FLAG, SEQ);     \
    else if (nb == 5) NDT_LAUNCH_MODE(B, 5, GY, FLAG, SEQ);     \
    else NDT_LAUNCH_MODE(B, 6, GY, FLAG, SEQ);                  \
  } while (0)
  if (d_poses) NDT_LAUNCH_NB(true, K, d_flag, seq);
  else NDT_LAUNCH_NB(false, 1, d_flag, seq);
#undef NDT_LAUNCH_NB
#undef NDT_LAUNCH_MODE
#undef NDT_LAUNCH
#undef NDT_LAUNCH2
}

namespace {
__global__ void __launch_bounds__(256) k_pack_records(const VoxelRecord* __restrict__ rec, PackedRecord* __restrict__ out, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const VoxelRecord r = rec[i];
  PackedRecord p;
  p.mean[0] = r.mean[0]; p.mean[1] = r.mean[1]; p.mean[2] = r.mean[2];
#pragma unroll
  for (int k = 0; k < 6; ++k) p.icov[k] = (float)r.icov[k];  // round to nearest even, like the reference's c_inv4 cast
  out[i] = p;
}
}  // namespace

void launch_pack_records(const VoxelRecord* rec, PackedRecord* out, size_t n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rec, out, (int)n);
}

void launch_transform(const float* sx, const float* sy, const float* sz, size_t n,
                      const PoseConsts& pose, float* out_xyz, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sx, sy, sz,
                     (int)n, pose, out_xyz);
}

}  // namespace ndt
