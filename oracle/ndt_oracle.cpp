// ndt_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see ndt_oracle.h for the parity
// status and the list of reference files this restates).  Dependency-free
// float64 CPU statement of the reference's NDT path.  Compile with
// -ffp-contract=off so f32 expressions round exactly as written.
//
// All "ref:" citations are relative to /root/reference.
#include "ndt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <unordered_map>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// small dense helpers (row-major 3x3 / 6x6, f64)
// ---------------------------------------------------------------------------

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations; eigenvalues
// ascending, eigenvectors as columns of V (row-major).  Stands in for
// Eigen::SelfAdjointEigenSolver (ref: voxel_grid_covariance_impl.hpp:298-300).
void sym_eig3(const double A_in[9], double evals[3], double V[9]) {
  double A[9];
  std::memcpy(A, A_in, sizeof(A));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double diag = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < 2; ++p) {
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[3 * p + q];
        if (apq == 0.0) continue;
        double app = A[3 * p + p], aqq = A[3 * q + q];
        double theta = (aqq - app) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) /
                   (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A * G
          double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- G^T * A
          double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
    }
  }
  int order[3] = {0, 1, 2};
  double d[3] = {A[0], A[4], A[8]};
  std::sort(order, order + 3, [&](int a, int b) { return d[a] < d[b]; });
  double Vs[9];
  for (int j = 0; j < 3; ++j) {
    evals[j] = d[order[j]];
    for (int k = 0; k < 3; ++k) Vs[3 * k + j] = V[3 * k + order[j]];
  }
  std::memcpy(V, Vs, sizeof(Vs));
}

// 3x3 inverse by cofactors (what Eigen does for fixed 3x3; ref :334).
void inv3(const double m[9], double out[9]) {
  double c00 = m[4] * m[8] - m[5] * m[7];
  double c01 = m[5] * m[6] - m[3] * m[8];
  double c02 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  double id = 1.0 / det;
  out[0] = c00 * id;
  out[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  out[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  out[3] = c01 * id;
  out[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  out[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  out[6] = c02 * id;
  out[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  out[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// x = pinv(A) b through a one-sided Jacobi SVD of the 6x6 matrix; singular
// values <= 6*eps*sigma_max are dropped (Eigen::JacobiSVD::solve default).
void svd_solve6(const double A_in[36], const double b[6], double x[6]) {
  double U[36], V[36];
  std::memcpy(U, A_in, sizeof(U));
  for (int i = 0; i < 36; ++i) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 5; ++p) {
      for (int q = p + 1; q < 6; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < 6; ++k) {
          alpha += U[6 * k + p] * U[6 * k + p];
          beta += U[6 * k + q] * U[6 * k + q];
          gamma += U[6 * k + p] * U[6 * k + q];
        }
        if (gamma == 0.0 || std::fabs(gamma) <= 1e-300 + 1e-15 * std::sqrt(alpha * beta)) continue;
        rotated = true;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0.0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int k = 0; k < 6; ++k) {
          double up = U[6 * k + p], uq = U[6 * k + q];
          U[6 * k + p] = c * up - s * uq;
          U[6 * k + q] = s * up + c * uq;
          double vp = V[6 * k + p], vq = V[6 * k + q];
          V[6 * k + p] = c * vp - s * vq;
          V[6 * k + q] = s * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
  double sig[6], smax = 0;
  for (int j = 0; j < 6; ++j) {
    double n2 = 0;
    for (int k = 0; k < 6; ++k) n2 += U[6 * k + j] * U[6 * k + j];
    sig[j] = std::sqrt(n2);
    smax = std::max(smax, sig[j]);
  }
  double thr = std::max(smax * 6.0 * std::numeric_limits<double>::epsilon(),
                        std::numeric_limits<double>::min());
  for (int i = 0; i < 6; ++i) x[i] = 0.0;
  for (int j = 0; j < 6; ++j) {
    if (!(sig[j] > thr)) continue;
    double utb = 0;
    for (int k = 0; k < 6; ++k) utb += U[6 * k + j] * b[k];  // U col j is sig*u_j
    double coef = utb / (sig[j] * sig[j]);
    for (int i = 0; i < 6; ++i) x[i] += V[6 * i + j] * coef;
  }
}

inline const float* pt_at(const float* base, size_t stride, size_t i) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * stride);
}

inline bool finite3(const float* p) {
  return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
}

}  // namespace

// ---------------------------------------------------------------------------
// voxel grid
// ---------------------------------------------------------------------------
struct oracle_leaf {
  int64_t cell;
  int n;
  double mean[3], cov[9], icov[9], evecs[9], evals[3];
};

struct oracle_grid {
  float leaf, inv_leaf;
  int min_b[3], max_b[3], div_b[3], divb_mul[3];
  int min_points;
  int64_t n_cells_hit;
  bool built;
  std::vector<oracle_leaf> leaves;                // ascending cell index
  std::unordered_map<int64_t, int64_t> cell2rank; // valid leaves only
};

extern "C" void oracle_default_params(oracle_params* p) {
  std::memset(p, 0, sizeof(*p));
  p->resolution = 1.0f;
  p->outlier_ratio = 0.55;
  p->step_size = 0.1;
  p->trans_epsilon = 0.01;
  p->max_iterations = 35;
  p->search_method = ORACLE_DIRECT7;
  p->min_points_per_voxel = 6;
  p->eig_inflation_ratio = 0.01;
  p->hessian_mode = ORACLE_HESSIAN_FULL;
  p->cov_mode = ORACLE_COV_SVN;
  p->pair_mode = ORACLE_PAIR_SVN;
  p->add_ridge = 0;
  p->use_line_search = 1;
  p->num_threads = 1;
  p->use_regularization = 0;
  p->regularization_scale_factor = 0.0f;
  p->symmetrize_hessian = 0;
}

// ref: voxel_grid_covariance_impl.hpp:222-225 -- f32 floor, f32 subtraction of
// min_b, truncation to int.
static inline int cell_coord(float v, float inv_leaf, int min_b) {
  return static_cast<int>(std::floor(v * inv_leaf) - static_cast<float>(min_b));
}

// ref: voxel_grid_covariance_impl.hpp:77-379 (applyFilter, unfiltered pass)
extern "C" oracle_grid* oracle_grid_build(const float* xyz, size_t n, size_t stride,
                                          const oracle_params* prm) {
  oracle_grid* g = new oracle_grid();
  g->built = false;
  g->leaf = prm->resolution;
  g->inv_leaf = 1.0f / prm->resolution;  // pcl::VoxelGrid::setLeafSize, f32 division
  g->min_points = std::max(3, prm->min_points_per_voxel);  // ref: voxel_grid_covariance.h:176-184
  g->n_cells_hit = 0;
  for (int a = 0; a < 3; ++a) g->min_b[a] = g->max_b[a] = g->div_b[a] = g->divb_mul[a] = 0;
  if (n == 0 || !(prm->resolution > 0.0f)) return g;

  // ref :103 pcl::getMinMax3D (non-finite points skipped)
  float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
                 std::numeric_limits<float>::max()};
  float mx[3] = {-mn[0], -mn[1], -mn[2]};
  size_t n_finite = 0;
  for (size_t i = 0; i < n; ++i) {
    const float* p = pt_at(xyz, stride, i);
    if (!finite3(p)) continue;
    ++n_finite;
    for (int a = 0; a < 3; ++a) {
      mn[a] = std::min(mn[a], p[a]);
      mx[a] = std::max(mx[a], p[a]);
    }
  }
  if (n_finite == 0) return g;

  // ref :108-125 overflow guard
  int64_t d[3];
  for (int a = 0; a < 3; ++a) d[a] = static_cast<int64_t>((mx[a] - mn[a]) * g->inv_leaf) + 1;
  const int64_t lim = std::numeric_limits<int32_t>::max();
  if (d[0] < 0 || d[1] < 0 || d[2] < 0 || d[0] > lim || d[1] > lim || d[2] > lim ||
      d[0] * d[1] > lim || d[0] * d[1] * d[2] > lim)
    return g;

  // ref :129-140
  for (int a = 0; a < 3; ++a) {
    g->min_b[a] = static_cast<int>(std::floor(mn[a] * g->inv_leaf));
    g->max_b[a] = static_cast<int>(std::floor(mx[a] * g->inv_leaf));
    g->div_b[a] = g->max_b[a] - g->min_b[a] + 1;
  }
  g->divb_mul[0] = 1;
  g->divb_mul[1] = g->div_b[0];
  g->divb_mul[2] = g->div_b[0] * g->div_b[1];

  // pass 1, ref :218-248: running sums in f64, in point order
  struct acc { int n; double s[3]; double ss[9]; };
  std::unordered_map<int64_t, acc> cells;
  cells.reserve(n_finite / 4 + 16);
  for (size_t i = 0; i < n; ++i) {
    const float* p = pt_at(xyz, stride, i);
    if (!finite3(p)) continue;
    int i0 = cell_coord(p[0], g->inv_leaf, g->min_b[0]);
    int i1 = cell_coord(p[1], g->inv_leaf, g->min_b[1]);
    int i2 = cell_coord(p[2], g->inv_leaf, g->min_b[2]);
    int64_t idx = static_cast<int64_t>(i0 * g->divb_mul[0] + i1 * g->divb_mul[1] + i2 * g->divb_mul[2]);
    auto it = cells.find(idx);
    if (it == cells.end()) {
      acc z;
      std::memset(&z, 0, sizeof(z));
      it = cells.emplace(idx, z).first;
    }
    acc& c = it->second;
    double q[3] = {p[0], p[1], p[2]};
    for (int a = 0; a < 3; ++a) c.s[a] += q[a];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) c.ss[3 * a + b] += q[a] * q[b];
    c.n++;
  }
  g->n_cells_hit = static_cast<int64_t>(cells.size());

  std::vector<int64_t> keys;
  keys.reserve(cells.size());
  for (auto& kv : cells) keys.push_back(kv.first);
  std::sort(keys.begin(), keys.end());

  // pass 2, ref :265-373
  for (int64_t key : keys) {
    const acc& c = cells[key];
    if (c.n < g->min_points) continue;  // ref :270-273
    oracle_leaf L;
    L.cell = key;
    L.n = c.n;
    const double cnt = static_cast<double>(c.n);
    for (int a = 0; a < 3; ++a) L.mean[a] = c.s[a] / cnt;  // ref :278
    if (prm->cov_mode == ORACLE_COV_SVN) {
      // ref :287-291  cov = ss/n - mu mu^T, then * n/(n-1)
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          L.cov[3 * a + b] = (c.ss[3 * a + b] / cnt) - (L.mean[a] * L.mean[b]);
      const double k = cnt / (cnt - 1.0);
      for (int a = 0; a < 9; ++a) L.cov[a] *= k;
    } else {
      // [RECALLED] PCL: (ss - 2 s mu^T)/n + mu mu^T, then * (n-1)/n
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
          L.cov[3 * a + b] = (c.ss[3 * a + b] - 2.0 * (c.s[a] * L.mean[b])) / cnt + L.mean[a] * L.mean[b];
      const double k = (cnt - 1.0) / cnt;
      for (int a = 0; a < 9; ++a) L.cov[a] *= k;
    }
    // ref :298-309
    double ev[3];
    sym_eig3(L.cov, ev, L.evecs);
    const double min_thr = 1e-12;
    if (ev[0] < 0 || ev[1] < 0 || ev[2] < min_thr) continue;
    // ref :311-331
    const double floor_ev = std::max(min_thr, ev[2] * prm->eig_inflation_ratio);
    bool recompose = false;
    if (ev[0] < floor_ev) { ev[0] = floor_ev; recompose = true; }
    if (ev[1] < floor_ev) { ev[1] = floor_ev; recompose = true; }
    for (int a = 0; a < 3; ++a) L.evals[a] = ev[a];
    if (recompose) {
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
          double s = 0;
          for (int k = 0; k < 3; ++k) s += L.evecs[3 * a + k] * ev[k] * L.evecs[3 * b + k];
          L.cov[3 * a + b] = s;
        }
    }
    // ref :334-343
    inv3(L.cov, L.icov);
    bool ok = true;
    double amax = 0;
    for (int a = 0; a < 9; ++a) {
      if (!std::isfinite(L.icov[a])) ok = false;
      amax = std::max(amax, std::fabs(L.icov[a]));
    }
    if (!ok || amax > 1e12) continue;
    g->cell2rank.emplace(key, static_cast<int64_t>(g->leaves.size()));
    g->leaves.push_back(L);
  }
  g->built = true;
  return g;
}

extern "C" void oracle_grid_free(oracle_grid* g) { delete g; }

extern "C" void oracle_grid_get_info(const oracle_grid* g, oracle_grid_info* out) {
  for (int a = 0; a < 3; ++a) {
    out->min_b[a] = g->min_b[a];
    out->max_b[a] = g->max_b[a];
    out->div_b[a] = g->div_b[a];
  }
  out->leaf = g->leaf;
  out->inv_leaf = g->inv_leaf;
  out->n_leaves = static_cast<int64_t>(g->leaves.size());
  out->n_cells_hit = g->n_cells_hit;
}

extern "C" void oracle_grid_export(const oracle_grid* g, int64_t* cell, int32_t* count,
                                   double* mean3, double* cov9, double* icov9,
                                   double* evecs9, double* evals3) {
  for (size_t i = 0; i < g->leaves.size(); ++i) {
    const oracle_leaf& L = g->leaves[i];
    if (cell) cell[i] = L.cell;
    if (count) count[i] = L.n;
    if (mean3) std::memcpy(mean3 + 3 * i, L.mean, sizeof(L.mean));
    if (cov9) std::memcpy(cov9 + 9 * i, L.cov, sizeof(L.cov));
    if (icov9) std::memcpy(icov9 + 9 * i, L.icov, sizeof(L.icov));
    if (evecs9) std::memcpy(evecs9 + 9 * i, L.evecs, sizeof(L.evecs));
    if (evals3) std::memcpy(evals3 + 3 * i, L.evals, sizeof(L.evals));
  }
}

// ref: voxel_grid_covariance_impl.hpp:46-71 (bounds in f32) +
//      voxel_grid_covariance.h:280-304 (index) + :262-270 (validity)
static int64_t leaf_rank_at(const oracle_grid* g, float x, float y, float z) {
  if (!g->built) return -1;
  const float p[3] = {x, y, z};
  for (int a = 0; a < 3; ++a) {
    float lo = static_cast<float>(g->min_b[a]) * g->leaf;
    float hi = static_cast<float>(g->max_b[a] + 1) * g->leaf;
    if (!(p[a] >= lo && p[a] < hi)) return -1;
  }
  int i0 = cell_coord(x, g->inv_leaf, g->min_b[0]);
  int i1 = cell_coord(y, g->inv_leaf, g->min_b[1]);
  int i2 = cell_coord(z, g->inv_leaf, g->min_b[2]);
  int64_t idx = static_cast<int64_t>(i0 * g->divb_mul[0] + i1 * g->divb_mul[1] + i2 * g->divb_mul[2]);
  auto it = g->cell2rank.find(idx);
  return it == g->cell2rank.end() ? -1 : it->second;
}

// ref: voxel_grid_covariance_impl.hpp:560-600 (DIRECT7: centre, +x, -x, +y, -y,
// +z, -z, found by offsetting the POINT by the leaf size in f32) and :604-615.
// ref: voxel_grid_covariance_impl.hpp:505-554 (radiusSearch over the centroid cloud of
// :386-435); FLANN's L2_Simple accumulates (dx^2 + dy^2) + dz^2 in f32 and keeps dist < r^2.
static int kd_neighbors(const oracle_grid* g, const float p[3], int64_t out_rank[27]) {
  if (!g->built) return 0;
  const float r2 = static_cast<float>(static_cast<double>(g->leaf) * static_cast<double>(g->leaf));
  const int c[3] = {cell_coord(p[0], g->inv_leaf, g->min_b[0]), cell_coord(p[1], g->inv_leaf, g->min_b[1]),
                    cell_coord(p[2], g->inv_leaf, g->min_b[2])};
  struct Hit { float d; int64_t r; };
  Hit hits[27];
  int n = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int i0 = c[0] + dx, i1 = c[1] + dy, i2 = c[2] + dz;
        if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= g->div_b[0] || i1 >= g->div_b[1] || i2 >= g->div_b[2]) continue;
        const int64_t idx = static_cast<int64_t>(i0 * g->divb_mul[0] + i1 * g->divb_mul[1] + i2 * g->divb_mul[2]);
        auto it = g->cell2rank.find(idx);
        if (it == g->cell2rank.end()) continue;
        const oracle_leaf& L = g->leaves[it->second];
        const float ex = p[0] - static_cast<float>(L.mean[0]), ey = p[1] - static_cast<float>(L.mean[1]),
                    ez = p[2] - static_cast<float>(L.mean[2]);
        float d = 0.0f;
        d += ex * ex; d += ey * ey; d += ez * ez;
        if (d < r2) hits[n++] = Hit{d, it->second};
      }
  std::sort(hits, hits + n, [](const Hit& a, const Hit& b) { return a.d < b.d || (a.d == b.d && a.r < b.r); });
  for (int k = 0; k < n; ++k) out_rank[k] = hits[k].r;
  return n;
}

// DIRECT26 [RECALLED]: pclomp's getNeighborhoodAtPoint -- the 3x3x3 block of cells around the
// point's own cell, enumerated in INTEGER index space (ijk = floor(p * inv_leaf), displacement
// kept when min_b <= ijk + d <= max_b), every valid leaf found there is a neighbour (27 cells,
// the centre included; "26" counts the surrounding ones).  The reference tree only carries the
// enum value (run/pipeline.cpp:471-480) and a commented-out stub (svn_ndt_impl.hpp:581-583);
// the enumeration order is x fastest, then y, then z.
static int direct26_neighbors(const oracle_grid* g, const float p[3], int64_t out_rank[27]) {
  if (!g->built) return 0;
  const int c[3] = {cell_coord(p[0], g->inv_leaf, g->min_b[0]), cell_coord(p[1], g->inv_leaf, g->min_b[1]),
                    cell_coord(p[2], g->inv_leaf, g->min_b[2])};
  int n = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int i0 = c[0] + dx, i1 = c[1] + dy, i2 = c[2] + dz;
        if (i0 < 0 || i1 < 0 || i2 < 0 || i0 >= g->div_b[0] || i1 >= g->div_b[1] || i2 >= g->div_b[2]) continue;
        const int64_t idx = static_cast<int64_t>(i0 * g->divb_mul[0] + i1 * g->divb_mul[1] + i2 * g->divb_mul[2]);
        auto it = g->cell2rank.find(idx);
        if (it != g->cell2rank.end()) out_rank[n++] = it->second;
      }
  return n;
}

extern "C" int oracle_grid_neighbors(const oracle_grid* g, const float p[3], int method,
                                     int64_t out_rank[27]) {
  if (method == ORACLE_KDTREE) return kd_neighbors(g, p, out_rank);
  if (method == ORACLE_DIRECT26) return direct26_neighbors(g, p, out_rank);
  int n = 0;
  int64_t r = leaf_rank_at(g, p[0], p[1], p[2]);
  if (r >= 0) out_rank[n++] = r;
  if (method == ORACLE_DIRECT1) return n;
  const float w = g->leaf;
  const float q[6][3] = {{p[0] + w, p[1], p[2]}, {p[0] - w, p[1], p[2]},
                         {p[0], p[1] + w, p[2]}, {p[0], p[1] - w, p[2]},
                         {p[0], p[1], p[2] + w}, {p[0], p[1], p[2] - w}};
  for (int k = 0; k < 6; ++k) {
    r = leaf_rank_at(g, q[k][0], q[k][1], q[k][2]);
    if (r >= 0) out_rank[n++] = r;
  }
  return n;
}

// ---------------------------------------------------------------------------
// NDT math
// ---------------------------------------------------------------------------

// ref: svn_ndt_impl.hpp:80-131
extern "C" void oracle_gauss_constants(double resolution, double outlier_ratio, double out[3]) {
  double c1 = 10.0 * (1.0 - outlier_ratio);
  double c2 = outlier_ratio / std::pow(resolution, 3);
  const double eps = 1e-9;
  if (c1 <= eps) c1 = eps;
  if (c2 <= eps) c2 = eps;
  double d3 = -std::log(c2);
  double d1 = -std::log(c1 + c2) - d3;
  double d2 = 1.0;
  if (std::fabs(d1) >= eps) {
    double inner = c1 * std::exp(-0.5) + c2;
    if (inner > eps) {
      double outer = (-std::log(inner) - d3) / d1;
      if (outer > eps) d2 = -2.0 * std::log(outer);
    }
  }
  if (!std::isfinite(d1) || !std::isfinite(d2) || !std::isfinite(d3)) {
    d1 = 1.0; d2 = 1.0; d3 = 0.0;
  }
  out[0] = d1; out[1] = d2; out[2] = d3;
}

// ref: svn_ndt_impl.hpp:255-334 (Magnusson 2009 eq. 6.19 / 6.21), R = Rx*Ry*Rz
extern "C" void oracle_angle_tables(const double p[6], float j[24], float h[45]) {
  double cx, cy, cz, sx, sy, sz;
  const double aeps = 1e-7;
  if (std::fabs(p[3]) < aeps) { sx = 0; cx = 1; } else { sx = std::sin(p[3]); cx = std::cos(p[3]); }
  if (std::fabs(p[4]) < aeps) { sy = 0; cy = 1; } else { sy = std::sin(p[4]); cy = std::cos(p[4]); }
  if (std::fabs(p[5]) < aeps) { sz = 0; cz = 1; } else { sz = std::sin(p[5]); cz = std::cos(p[5]); }
  const double J[8][3] = {
      {-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy},
      {cx * sz + sx * sy * cz, cx * cz - sx * sy * sz, -sx * cy},
      {-sy * cz, sy * sz, cy},
      {sx * cy * cz, -sx * cy * sz, sx * sy},
      {-cx * cy * cz, cx * cy * sz, -cx * sy},
      {-cy * sz, -cy * cz, 0.0},
      {cx * cz - sx * sy * sz, -cx * sz - sx * sy * cz, 0.0},
      {sx * cz + cx * sy * sz, cx * sy * cz - sx * sz, 0.0}};
  const double H[15][3] = {
      {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy},   // a2
      {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy},  // a3
      {cx * cy * cz, -cx * cy * sz, cx * sy},                        // b2
      {sx * cy * cz, -sx * cy * sz, sx * sy},                        // b3
      {-sx * cz - cx * sy * sz, sx * sz - cx * sy * cz, 0.0},        // c2
      {cx * cz - sx * sy * sz, -sx * sy * cz - cx * sz, 0.0},        // c3
      {-cy * cz, cy * sz, sy},                                       // d1
      {-sx * sy * cz, sx * sy * sz, sx * cy},                        // d2
      {cx * sy * cz, -cx * sy * sz, -cx * cy},                       // d3
      {sy * sz, sy * cz, 0.0},                                       // e1
      {-sx * cy * sz, -sx * cy * cz, 0.0},                           // e2
      {cx * cy * sz, cx * cy * cz, 0.0},                             // e3
      {-cy * cz, cy * sz, 0.0},                                      // f1
      {-cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0.0},       // f2
      {-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0.0}};      // f3
  for (int r = 0; r < 8; ++r)
    for (int c = 0; c < 3; ++c) j[3 * r + c] = static_cast<float>(J[r][c]);
  for (int r = 0; r < 15; ++r)
    for (int c = 0; c < 3; ++c) h[3 * r + c] = static_cast<float>(H[r][c]);
}

// p -> 4x4 f32.  pclomp builds Translation<float> * AngleAxis<float>(roll,X) *
// AngleAxis<float>(pitch,Y) * AngleAxis<float>(yaw,Z) [RECALLED]; restated as f32
// trig and f32 products (Rx*Ry)*Rz.
extern "C" void oracle_pose_to_matrix(const double p[6], float T[16]) {
  float r = static_cast<float>(p[3]), pi = static_cast<float>(p[4]), y = static_cast<float>(p[5]);
  float sx = std::sin(r), cx = std::cos(r);
  float sy = std::sin(pi), cy = std::cos(pi);
  float sz = std::sin(y), cz = std::cos(y);
  float Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
  float Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  float Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  float A[9], R[9];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      float s = 0;
      for (int m = 0; m < 3; ++m) s += Rx[3 * i + m] * Ry[3 * m + k];
      A[3 * i + k] = s;
    }
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      float s = 0;
      for (int m = 0; m < 3; ++m) s += A[3 * i + m] * Rz[3 * m + k];
      R[3 * i + k] = s;
    }
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) T[4 * c + rr] = (rr == c) ? 1.0f : 0.0f;
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) T[4 * k + i] = R[3 * i + k];
  T[12] = static_cast<float>(p[0]);
  T[13] = static_cast<float>(p[1]);
  T[14] = static_cast<float>(p[2]);
}

// 4x4 f32 -> p.  Angles by Eigen's Matrix3f::eulerAngles(0,1,2) (published Eigen
// 3.3/3.4 algorithm, incl. its [0,pi] first-angle range) [RECALLED call:
// eig_transformation.rotation().eulerAngles(0,1,2) in pclomp]; the polar
// decomposition inside Transform::rotation() is skipped (input is a rotation).
extern "C" void oracle_matrix_to_pose(const float T[16], double p[6]) {
  auto m = [&](int r, int c) { return T[4 * c + r]; };
  const float PI_F = 3.14159265358979323846f;
  float res[3];
  res[0] = std::atan2(m(1, 2), m(2, 2));
  float c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
  if (res[0] > 0.0f) {  // even permutation branch
    res[0] -= PI_F;
    res[1] = std::atan2(-m(0, 2), -c2);
  } else {
    res[1] = std::atan2(-m(0, 2), c2);
  }
  float s1 = std::sin(res[0]), c1 = std::cos(res[0]);
  res[2] = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
  p[0] = T[12]; p[1] = T[13]; p[2] = T[14];
  p[3] = -res[0]; p[4] = -res[1]; p[5] = -res[2];
}

namespace {

struct PointDeriv {
  float J[3][6];      // point Jacobian rows x,y,z (ref: svn_ndt_impl.hpp:339-363, :603-604)
  float Hp[6][6][3];  // second derivatives d2x'/dp_i dp_j (ref :369-394)
};

// ref: svn_ndt_impl.hpp:339-396
void point_derivatives(const float x[3], const float jang[24], const float hang[45],
                       bool need_h, PointDeriv& d) {
  std::memset(&d, 0, sizeof(d));
  d.J[0][0] = d.J[1][1] = d.J[2][2] = 1.0f;
  float xj[8];
  for (int r = 0; r < 8; ++r)
    xj[r] = jang[3 * r] * x[0] + jang[3 * r + 1] * x[1] + jang[3 * r + 2] * x[2];
  d.J[1][3] = xj[0]; d.J[2][3] = xj[1];
  d.J[0][4] = xj[2]; d.J[1][4] = xj[3]; d.J[2][4] = xj[4];
  d.J[0][5] = xj[5]; d.J[1][5] = xj[6]; d.J[2][5] = xj[7];
  if (!need_h) return;
  float xh[15];
  for (int r = 0; r < 15; ++r)
    xh[r] = hang[3 * r] * x[0] + hang[3 * r + 1] * x[1] + hang[3 * r + 2] * x[2];
  d.Hp[3][3][1] = xh[0]; d.Hp[3][3][2] = xh[1];
  d.Hp[3][4][1] = xh[2]; d.Hp[3][4][2] = xh[3];
  d.Hp[4][3][1] = xh[2]; d.Hp[4][3][2] = xh[3];
  d.Hp[3][5][1] = xh[4]; d.Hp[3][5][2] = xh[5];
  d.Hp[5][3][1] = xh[4]; d.Hp[5][3][2] = xh[5];
  d.Hp[4][4][0] = xh[6]; d.Hp[4][4][1] = xh[7]; d.Hp[4][4][2] = xh[8];
  d.Hp[4][5][0] = xh[9]; d.Hp[4][5][1] = xh[10]; d.Hp[4][5][2] = xh[11];
  d.Hp[5][4][0] = xh[9]; d.Hp[5][4][1] = xh[10]; d.Hp[5][4][2] = xh[11];
  d.Hp[5][5][0] = xh[12]; d.Hp[5][5][1] = xh[13]; d.Hp[5][5][2] = xh[14];
}

// ref: svn_ndt_impl.hpp:401-513 (vendored per-pair update).  F = float is the reference's
// arithmetic (x_trans4 / c_inv4 / point_gradient4 are float there); F = double carries the same
// formulas in f64 (ORACLE_PAIR_SVN_F64, a test seam).
template <typename F>
double update_pair_svn_t(double g[6], double H[36], const PointDeriv& d, const double xr[3],
                       const double ci[9], double d1, double d2, bool need_h, bool gauss_newton) {
  double cx[3];
  for (int a = 0; a < 3; ++a) cx[a] = ci[3 * a] * xr[0] + ci[3 * a + 1] * xr[1] + ci[3 * a + 2] * xr[2];
  double mahal = xr[0] * cx[0] + xr[1] * cx[1] + xr[2] * cx[2];
  if (!std::isfinite(mahal) || mahal < -1e-9) return 0.0;
  if (mahal < 0.0) mahal = 0.0;
  double earg = d2 * mahal * 0.5;
  if (earg > 50.0) return 0.0;
  double e = std::exp(-earg);
  double score_inc = -d1 * e;
  double factor = d1 * d2 * e;
  if (!std::isfinite(factor) || std::fabs(factor) < 1e-15) return score_inc;

  F x4[3] = {static_cast<F>(xr[0]), static_cast<F>(xr[1]), static_cast<F>(xr[2])};
  F c4[9];
  for (int a = 0; a < 9; ++a) c4[a] = static_cast<F>(ci[a]);
  F tv[3][6];  // C^-1 * J
  for (int a = 0; a < 3; ++a)
    for (int j = 0; j < 6; ++j) {
      F s = 0;
      for (int k = 0; k < 3; ++k) s += c4[3 * a + k] * static_cast<F>(d.J[k][j]);
      tv[a][j] = s;
    }
  F gc[6];  // (x-mu)^T C^-1 J
  for (int j = 0; j < 6; ++j) gc[j] = x4[0] * tv[0][j] + x4[1] * tv[1][j] + x4[2] * tv[2][j];
  double ginc[6];
  bool gfin = true;
  for (int j = 0; j < 6; ++j) {
    ginc[j] = factor * static_cast<double>(gc[j]);
    gfin = gfin && std::isfinite(ginc[j]);
  }
  if (gfin)
    for (int j = 0; j < 6; ++j) g[j] += ginc[j];

  if (need_h) {
    double hc[36];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        F s = 0;
        for (int k = 0; k < 3; ++k) s += static_cast<F>(d.J[k][i]) * tv[k][j];
        hc[6 * i + j] = static_cast<double>(s);  // term2 = J^T C^-1 J
      }
    if (!gauss_newton) {
      F xc[3];
      for (int k = 0; k < 3; ++k) xc[k] = x4[0] * c4[k] + x4[1] * c4[3 + k] + x4[2] * c4[6 + k];
      for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
          F t3 = xc[0] * static_cast<F>(d.Hp[i][j][0]) + xc[1] * static_cast<F>(d.Hp[i][j][1]) +
                 xc[2] * static_cast<F>(d.Hp[i][j][2]);
          double t1 = -d2 * (static_cast<double>(gc[i]) * static_cast<double>(gc[j]));
          double add_ij = t1 + static_cast<double>(t3);
          hc[6 * i + j] += add_ij;
          if (j != i) hc[6 * j + i] += add_ij;
        }
    }
    bool hfin = true;
    for (int a = 0; a < 36; ++a) {
      hc[a] *= factor;
      hfin = hfin && std::isfinite(hc[a]);
    }
    if (hfin)
      for (int a = 0; a < 36; ++a) H[a] += hc[a];
  }
  return score_inc;
}

double update_pair_svn(double g[6], double H[36], const PointDeriv& d, const double xr[3],
                       const double ci[9], double d1, double d2, bool need_h, bool gauss_newton) {
  return update_pair_svn_t<float>(g, H, d, xr, ci, d1, d2, need_h, gauss_newton);
}

// [RECALLED] upstream pclomp updateDerivatives: everything in f32, guard on
// d2*e outside [0,1] or NaN; full analytic Hessian only.
double update_pair_pclomp(double g[6], double H[36], const PointDeriv& d, const double xr[3],
                          const double ci[9], double d1, double d2, bool need_h) {
  float x4[3] = {static_cast<float>(xr[0]), static_cast<float>(xr[1]), static_cast<float>(xr[2])};
  float c4[9];
  for (int a = 0; a < 9; ++a) c4[a] = static_cast<float>(ci[a]);
  float gd2 = static_cast<float>(d2);
  float xc[3];
  for (int k = 0; k < 3; ++k) xc[k] = x4[0] * c4[k] + x4[1] * c4[3 + k] + x4[2] * c4[6 + k];
  float q = x4[0] * xc[0] + x4[1] * xc[1] + x4[2] * xc[2];
  float e = std::exp(-gd2 * q * 0.5f);
  float score_inc = static_cast<float>(-d1 * e);
  e = gd2 * e;
  if (e > 1 || e < 0 || e != e) return 0.0;
  e = static_cast<float>(e * d1);
  float tv[3][6];
  for (int a = 0; a < 3; ++a)
    for (int j = 0; j < 6; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += c4[3 * a + k] * d.J[k][j];
      tv[a][j] = s;
    }
  float gc[6];
  for (int j = 0; j < 6; ++j) gc[j] = x4[0] * tv[0][j] + x4[1] * tv[1][j] + x4[2] * tv[2][j];
  for (int j = 0; j < 6; ++j) g[j] += static_cast<double>(e * gc[j]);
  if (need_h) {
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        float jcj = 0;
        for (int k = 0; k < 3; ++k) jcj += d.J[k][j] * tv[k][i];
        float t3 = xc[0] * d.Hp[i][j][0] + xc[1] * d.Hp[i][j][1] + xc[2] * d.Hp[i][j][2];
        H[6 * i + j] += static_cast<double>(e * (-gd2 * gc[i] * gc[j] + t3 + jcj));
      }
  }
  return static_cast<double>(score_inc);
}

struct Accum {
  double score = 0;
  double g[6] = {0, 0, 0, 0, 0, 0};
  double H[36];
  double nvtl = 0;
  int64_t n_with = 0, n_pairs = 0;
  Accum() { std::memset(H, 0, sizeof(H)); }
};

}  // namespace

// ref: svn_ndt_impl.hpp:518-668.  The source is transformed in f32 as
// x' = r00*x + (r01*y + (r02*z + t)) (PCL's SSE transformer order [RECALLED],
// pcl::transformPointCloud at :761), never fused.
extern "C" void oracle_derivatives(const oracle_grid* g, const float* src, size_t n,
                                   size_t stride, const float T[16], const double pose6[6],
                                   const oracle_params* prm, int compute_hessian,
                                   oracle_derivs* out) {
  double gd[3];
  oracle_gauss_constants(static_cast<double>(prm->resolution), prm->outlier_ratio, gd);
  const double d1 = gd[0], d2 = gd[1];
  float jang[24], hang[45];
  oracle_angle_tables(pose6, jang, hang);
  const bool need_h = compute_hessian != 0;
  const bool gn = prm->hessian_mode == ORACLE_HESSIAN_GAUSS_NEWTON;
  const bool need_hp = need_h && !(gn && prm->pair_mode != ORACLE_PAIR_PCLOMP_RECALLED);

  int nthreads = std::max(1, prm->num_threads);
  std::vector<Accum> accs(nthreads);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
  {
    int tid = 0, nt = 1;
#ifdef _OPENMP
    tid = omp_get_thread_num();
    nt = omp_get_num_threads();
#endif
    size_t lo = n * static_cast<size_t>(tid) / nt, hi = n * static_cast<size_t>(tid + 1) / nt;
    Accum& A = accs[tid];
    PointDeriv pd;
    for (size_t i = lo; i < hi; ++i) {
      const float* x = pt_at(src, stride, i);
      float xt[3];
      for (int a = 0; a < 3; ++a)
        xt[a] = T[a] * x[0] + (T[4 + a] * x[1] + (T[8 + a] * x[2] + T[12 + a]));
      if (!finite3(xt)) continue;  // ref :573
      int64_t nb[27];
      int nn = oracle_grid_neighbors(g, xt, prm->search_method, nb);
      if (nn == 0) continue;  // ref :592
      point_derivatives(x, jang, hang, need_hp, pd);
      double ps = 0, pg[6] = {0, 0, 0, 0, 0, 0}, pH[36], best = 0;
      std::memset(pH, 0, sizeof(pH));
      for (int k = 0; k < nn; ++k) {
        const oracle_leaf& L = g->leaves[nb[k]];
        double xr[3] = {static_cast<double>(xt[0]) - L.mean[0],
                        static_cast<double>(xt[1]) - L.mean[1],
                        static_cast<double>(xt[2]) - L.mean[2]};
        double s = prm->pair_mode == ORACLE_PAIR_SVN
                       ? update_pair_svn(pg, pH, pd, xr, L.icov, d1, d2, need_h, gn)
                       : prm->pair_mode == ORACLE_PAIR_SVN_F64
                             ? update_pair_svn_t<double>(pg, pH, pd, xr, L.icov, d1, d2, need_h, gn)
                             : update_pair_pclomp(pg, pH, pd, xr, L.icov, d1, d2, need_h);
        ps += s;
        best = std::max(best, s);
      }
      A.score += ps;
      for (int a = 0; a < 6; ++a) A.g[a] += pg[a];
      for (int a = 0; a < 36; ++a) A.H[a] += pH[a];
      A.nvtl += best;
      A.n_with += 1;
      A.n_pairs += nn;
    }
  }
  std::memset(out, 0, sizeof(*out));
  for (int t = 0; t < nthreads; ++t) {
    out->score += accs[t].score;
    for (int a = 0; a < 6; ++a) out->gradient[a] += accs[t].g[a];
    for (int a = 0; a < 36; ++a) out->hessian[a] += accs[t].H[a];
    out->nvtl_sum += accs[t].nvtl;
    out->n_with_neighbors += accs[t].n_with;
    out->n_pairs += accs[t].n_pairs;
  }
  if (need_h && prm->symmetrize_hessian)
    for (int i = 0; i < 6; ++i)
      for (int j = i + 1; j < 6; ++j) out->hessian[6 * j + i] = out->hessian[6 * i + j];
  if (need_h && prm->add_ridge)  // ref :650-653
    for (int a = 0; a < 6; ++a) out->hessian[7 * a] += 1e-6;

  // [RECALLED] tier4 longitudinal regularisation, f32 arithmetic
  if (prm->use_regularization) {
    const float k = prm->regularization_scale_factor;
    const float dx = prm->regularization_pose[12] - static_cast<float>(pose6[0]);
    const float dy = prm->regularization_pose[13] - static_cast<float>(pose6[1]);
    const float sy = static_cast<float>(std::sin(pose6[5]));
    const float cy = static_cast<float>(std::cos(pose6[5]));
    const float lon = dy * sy + dx * cy;
    const float w = static_cast<float>(out->n_pairs);
    out->score += static_cast<double>(-k * w * lon * lon);
    out->gradient[0] += static_cast<double>(k * w * 2.0f * cy * lon);
    out->gradient[1] += static_cast<double>(k * w * 2.0f * sy * lon);
    if (need_h) {
      out->hessian[0] += static_cast<double>(-k * w * 2.0f * cy * cy);
      out->hessian[1] += static_cast<double>(-k * w * 2.0f * cy * sy);
      out->hessian[6] += static_cast<double>(-k * w * 2.0f * cy * sy);
      out->hessian[7] += static_cast<double>(-k * w * 2.0f * sy * sy);
    }
  }
  // ref :656-663 non-finite guards
  bool gfin = true, hfin = true;
  for (int a = 0; a < 6; ++a) gfin = gfin && std::isfinite(out->gradient[a]);
  for (int a = 0; a < 36; ++a) hfin = hfin && std::isfinite(out->hessian[a]);
  if (!gfin)
    for (int a = 0; a < 6; ++a) out->gradient[a] = 0;
  if (need_h && !hfin)
    for (int a = 0; a < 36; ++a) out->hessian[a] = (a % 7 == 0) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------
// Newton + More-Thuente (published algorithm; pclomp call sites
// ref: run/pipeline.cpp:557-568, test_svn_ndt.cpp:144-179)
// ---------------------------------------------------------------------------
namespace {

struct LineFn {
  const oracle_grid* g;
  const float* src;
  size_t n, stride;
  const oracle_params* prm;
  int n_evals = 0;
  float T[16];
  oracle_derivs last;

  void eval(const double p[6], bool need_h) {
    oracle_pose_to_matrix(p, T);
    oracle_derivatives(g, src, n, stride, T, p, prm, need_h ? 1 : 0, &last);
    ++n_evals;
  }
};

double cubic_min(double a, double fa, double ga, double b, double fb, double gb) {
  // minimiser of the cubic through (a,fa,ga),(b,fb,gb): Sun & Yuan 2.4.52/2.4.56
  double z = 3 * (fb - fa) / (b - a) - gb - ga;
  double w = std::sqrt(z * z - gb * ga);
  return a + (b - a) * (w - ga - z) / (gb - ga + 2 * w);
}

// More & Thuente 1994, "Trial value selection", cases 1-4
double trial_value(double a_l, double f_l, double g_l, double a_u, double f_u, double g_u,
                   double a_t, double f_t, double g_t) {
  if (f_t > f_l) {
    double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    double a_q = a_l - 0.5 * (a_l - a_t) * g_l / (g_l - (f_l - f_t) / (a_l - a_t));
    return (std::fabs(a_c - a_l) < std::fabs(a_q - a_l)) ? a_c : 0.5 * (a_q + a_c);
  }
  if (g_t * g_l < 0) {
    double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    return (std::fabs(a_c - a_t) >= std::fabs(a_s - a_t)) ? a_c : a_s;
  }
  if (std::fabs(g_t) <= std::fabs(g_l)) {
    double a_c = cubic_min(a_l, f_l, g_l, a_t, f_t, g_t);
    double a_s = a_l - (a_l - a_t) / (g_l - g_t) * g_l;
    double nxt = (std::fabs(a_c - a_t) < std::fabs(a_s - a_t)) ? a_c : a_s;
    return (a_t > a_l) ? std::min(a_t + 0.66 * (a_u - a_t), nxt)
                       : std::max(a_t + 0.66 * (a_u - a_t), nxt);
  }
  return cubic_min(a_u, f_u, g_u, a_t, f_t, g_t);
}

// More & Thuente 1994, "Updating algorithm" (cases U1-U3); true = interval collapsed
bool update_interval(double& a_l, double& f_l, double& g_l, double& a_u, double& f_u,
                     double& g_u, double a_t, double f_t, double g_t) {
  if (f_t > f_l) { a_u = a_t; f_u = f_t; g_u = g_t; return false; }
  if (g_t * (a_l - a_t) > 0) { a_l = a_t; f_l = f_t; g_l = g_t; return false; }
  if (g_t * (a_l - a_t) < 0) {
    a_u = a_l; f_u = f_l; g_u = g_l;
    a_l = a_t; f_l = f_t; g_l = g_t;
    return false;
  }
  return true;
}

// step length along dir from x; on return score/grad/hess hold the state at
// x + a*dir.  phi = -score (minimisation form).
double step_length_mt(LineFn& fn, const double x[6], double dir[6], double step_init,
                      double step_max, double step_min, double& score, double grad[6],
                      double hess[36], double x_t[6]) {
  double phi_0 = -score;
  double d_phi_0 = 0;
  for (int i = 0; i < 6; ++i) d_phi_0 -= grad[i] * dir[i];
  if (d_phi_0 >= 0) {
    if (d_phi_0 == 0) {
      for (int i = 0; i < 6; ++i) x_t[i] = x[i];
      return 0;
    }
    d_phi_0 = -d_phi_0;
    for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
  }
  const int max_step_iterations = 10;
  int step_iterations = 0;
  const double mu = 1e-4, nu = 0.9;
  double a_l = 0, a_u = 0;
  double f_l = 0, g_l = d_phi_0 - mu * d_phi_0;  // psi(0), psi'(0)
  double f_u = 0, g_u = g_l;
  bool interval_converged = (step_max - step_min) < 0, open_interval = true;
  double a_t = std::max(std::min(step_init, step_max), step_min);
  for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
  fn.eval(x_t, true);
  auto take = [&]() {
    score = fn.last.score;
    std::memcpy(grad, fn.last.gradient, sizeof(double) * 6);
  };
  take();
  std::memcpy(hess, fn.last.hessian, sizeof(double) * 36);
  double phi_t = -score, d_phi_t = 0;
  for (int i = 0; i < 6; ++i) d_phi_t -= grad[i] * dir[i];
  double psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
  double d_psi_t = d_phi_t - mu * d_phi_0;

  while (fn.prm->use_line_search && !interval_converged && step_iterations < max_step_iterations &&
         !(psi_t <= 0 && d_phi_t <= -nu * d_phi_0)) {
    a_t = open_interval ? trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                        : trial_value(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    a_t = std::max(std::min(a_t, step_max), step_min);
    for (int i = 0; i < 6; ++i) x_t[i] = x[i] + dir[i] * a_t;
    fn.eval(x_t, false);
    take();
    phi_t = -score;
    d_phi_t = 0;
    for (int i = 0; i < 6; ++i) d_phi_t -= grad[i] * dir[i];
    psi_t = phi_t - phi_0 - mu * d_phi_0 * a_t;
    d_psi_t = d_phi_t - mu * d_phi_0;
    if (open_interval && (psi_t <= 0 && d_psi_t >= 0)) {
      open_interval = false;
      f_l += phi_0 - mu * d_phi_0 * a_l;
      g_l += mu * d_phi_0;
      f_u += phi_0 - mu * d_phi_0 * a_u;
      g_u += mu * d_phi_0;
    }
    interval_converged =
        open_interval ? update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, psi_t, d_psi_t)
                      : update_interval(a_l, f_l, g_l, a_u, f_u, g_u, a_t, phi_t, d_phi_t);
    ++step_iterations;
  }
  if (step_iterations) {  // Hessian at the accepted point
    fn.eval(x_t, true);
    std::memcpy(hess, fn.last.hessian, sizeof(double) * 36);
  }
  return a_t;
}

}  // namespace

extern "C" void oracle_align(const oracle_grid* g, const float* src, size_t n, size_t stride,
                             const float guess[16], const oracle_params* prm,
                             oracle_result* out) {
  std::memset(out, 0, sizeof(*out));
  LineFn fn;
  fn.g = g; fn.src = src; fn.n = n; fn.stride = stride; fn.prm = prm;
  double p[6];
  oracle_matrix_to_pose(guess, p);
  std::memcpy(out->final_transformation, guess, sizeof(float) * 16);

  // first evaluation uses the guess matrix itself (pclomp transforms the
  // output cloud by `guess` before the loop [RECALLED])
  std::memcpy(fn.T, guess, sizeof(float) * 16);
  oracle_derivatives(g, src, n, stride, guess, p, prm, 1, &fn.last);
  fn.n_evals = 1;
  double score = fn.last.score, grad[6], hess[36];
  std::memcpy(grad, fn.last.gradient, sizeof(grad));
  std::memcpy(hess, fn.last.hessian, sizeof(hess));

  int iters = 0;
  bool converged = false;
  while (!converged) {
    double neg_g[6], dp[6];
    for (int i = 0; i < 6; ++i) neg_g[i] = -grad[i];
    svd_solve6(hess, neg_g, dp);
    double norm = 0;
    for (int i = 0; i < 6; ++i) norm += dp[i] * dp[i];
    norm = std::sqrt(norm);
    if (norm == 0 || norm != norm) {
      converged = (norm == norm);
      break;
    }
    for (int i = 0; i < 6; ++i) dp[i] /= norm;
    double x_t[6];
    double a = step_length_mt(fn, p, dp, norm, prm->step_size, prm->trans_epsilon / 2, score,
                              grad, hess, x_t);
    for (int i = 0; i < 6; ++i) p[i] += dp[i] * a;
    std::memcpy(out->final_transformation, fn.T, sizeof(float) * 16);
    if (out->n_logged < 128) {
      std::memcpy(out->log_pose[out->n_logged], p, sizeof(p));
      out->log_step[out->n_logged] = a;
      out->log_score[out->n_logged] = score;
      out->n_logged++;
    }
    if (iters > prm->max_iterations || (iters && std::fabs(a) < prm->trans_epsilon)) converged = true;
    ++iters;
  }
  out->converged = converged ? 1 : 0;
  out->iterations = iters;
  out->n_evaluations = fn.n_evals;
  std::memcpy(out->final_pose, p, sizeof(p));
  std::memcpy(out->hessian, hess, sizeof(hess));
  out->score = score;
  out->transform_probability = n ? score / static_cast<double>(n) : 0.0;
  out->nvtl = fn.last.n_with_neighbors ? fn.last.nvtl_sum / static_cast<double>(fn.last.n_with_neighbors) : 0.0;
}

// ---------------------------------------------------------------------------
// reference test fixture, ref: extern/svn_ndt/test/test_svn_ndt.cpp:44-83,104-111
// ---------------------------------------------------------------------------
namespace {
void so3_exp(const double w[3], double R[9]) {
  double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double K2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = s;
    }
  double a = th < 1e-10 ? 1.0 : std::sin(th) / th;
  double b = th < 1e-10 ? 0.5 : (1 - std::cos(th)) / (th * th);
  for (int i = 0; i < 9; ++i) R[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * K[i] + b * K2[i];
}
}  // namespace

extern "C" size_t oracle_two_plane_fixture(float* src, float* tgt, double gt16[16],
                                           double guess16[16]) {
  // ground truth: Rot3::Yaw(0.2618)*Rot3::Pitch(0.0873), t=(0.5,0,0.3)  (:104-106)
  const double yaw = 0.2618, pitch = 0.0873;
  const double cz = std::cos(yaw), sz = std::sin(yaw), cy = std::cos(pitch), sy = std::sin(pitch);
  const double Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
  const double Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy};
  double R[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += Rz[3 * i + k] * Ry[3 * k + j];
      R[3 * i + j] = s;
    }
  const double t[3] = {0.5, 0.0, 0.3};
  for (int i = 0; i < 16; ++i) gt16[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) gt16[4 * j + i] = R[3 * i + j];
    gt16[12 + i] = t[i];
  }
  // initial guess = gt.retract(-delta), delta=[rot 0.05,-0.02,0.04 | trans 0.02,-0.01,0.03]
  // (:110-111); GTSAM's Pose3 retract taken as the full SE(3) exponential.
  const double w[3] = {-0.05, 0.02, -0.04}, v[3] = {-0.02, 0.01, -0.03};
  double dR[9];
  so3_exp(w, dR);
  double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, K2[9], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += K[3 * i + k] * K[3 * k + j];
      K2[3 * i + j] = s;
    }
  double b = (1 - std::cos(th)) / (th * th), c = (th - std::sin(th)) / (th * th * th);
  for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + b * K[i] + c * K2[i];
  double dt[3];
  for (int i = 0; i < 3; ++i) dt[i] = V[3 * i] * v[0] + V[3 * i + 1] * v[1] + V[3 * i + 2] * v[2];
  for (int i = 0; i < 16; ++i) guess16[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += R[3 * i + k] * dR[3 * k + j];
      guess16[4 * j + i] = s;
    }
    guess16[12 + i] = R[3 * i] * dt[0] + R[3 * i + 1] * dt[1] + R[3 * i + 2] * dt[2] + t[i];
  }

  size_t n = 0;
  for (double x = -10.0; x <= 10.0; x += 0.15)
    for (double y = -10.0; y <= 10.0; y += 0.15) {
      src[3 * n] = static_cast<float>(x); src[3 * n + 1] = static_cast<float>(y); src[3 * n + 2] = 0.0f;
      ++n;
    }
  for (double x = -10.0; x <= 10.0; x += 0.15)
    for (double z = -10.0; z <= 10.0; z += 0.15) {
      src[3 * n] = static_cast<float>(x); src[3 * n + 1] = 0.0f; src[3 * n + 2] = static_cast<float>(z);
      ++n;
    }
  std::default_random_engine gen(1337);
  std::normal_distribution<double> noise(0.0, 0.02);
  for (size_t i = 0; i < n; ++i) {
    double ps[3] = {src[3 * i], src[3 * i + 1], src[3 * i + 2]};
    for (int a = 0; a < 3; ++a) {
      double pt = R[3 * a] * ps[0] + R[3 * a + 1] * ps[1] + R[3 * a + 2] * ps[2] + t[a];
      tgt[3 * i + a] = static_cast<float>(pt + noise(gen));
    }
  }
  return n;
}

// ---------------------------------------------------------------------------
// SVN-NDT outer loop, ref: extern/svn_ndt/include/svn_ndt_impl.hpp:675-964
// ---------------------------------------------------------------------------
namespace {

struct P3 {  // gtsam::Pose3 stand-in: row-major R, t
  double R[9];
  double t[3];
};

P3 p3_from16(const double T[16]) {
  P3 p;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) p.R[3 * i + j] = T[4 * j + i];
    p.t[i] = T[12 + i];
  }
  return p;
}
void p3_to16(const P3& p, double T[16]) {
  for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * j + i] = p.R[3 * i + j];
    T[12 + i] = p.t[i];
  }
}
P3 p3_compose(const P3& a, const P3& b) {
  P3 c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a.R[3 * i + k] * b.R[3 * k + j];
      c.R[3 * i + j] = s;
    }
    c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
  }
  return c;
}
P3 p3_between(const P3& a, const P3& b) {  // a^-1 * b
  P3 inv;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) inv.R[3 * i + j] = a.R[3 * j + i];
  for (int i = 0; i < 3; ++i)
    inv.t[i] = -(inv.R[3 * i] * a.t[0] + inv.R[3 * i + 1] * a.t[1] + inv.R[3 * i + 2] * a.t[2]);
  return p3_compose(inv, b);
}
// gtsam::Pose3::Expmap, xi = [omega, v]
P3 p3_expmap(const double xi[6]) {
  P3 p;
  so3_exp(xi, p.R);
  const double* w = xi;
  const double* v = xi + 3;
  double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > 1e-20) {
    double wv = w[0] * v[0] + w[1] * v[1] + w[2] * v[2];
    double c[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    for (int i = 0; i < 3; ++i) {
      double Rc = p.R[3 * i] * c[0] + p.R[3 * i + 1] * c[1] + p.R[3 * i + 2] * c[2];
      p.t[i] = (c[i] - Rc + w[i] * wv) / th2;
    }
  } else {
    for (int i = 0; i < 3; ++i) p.t[i] = v[i];
  }
  return p;
}
// gtsam::SO3 / Pose3 Logmap
void p3_logmap(const P3& p, double xi[6]) {
  const double* R = p.R;
  double tr = R[0] + R[4] + R[8];
  double w[3];
  if (tr + 1.0 < 1e-10) {  // rotation by ~pi
    const double PI = 3.14159265358979323846;
    if (std::fabs(R[8] + 1.0) > 1e-5) {
      double k = PI / std::sqrt(2.0 + 2.0 * R[8]);
      w[0] = k * R[2]; w[1] = k * R[5]; w[2] = k * (1.0 + R[8]);
    } else if (std::fabs(R[4] + 1.0) > 1e-5) {
      double k = PI / std::sqrt(2.0 + 2.0 * R[4]);
      w[0] = k * R[1]; w[1] = k * (1.0 + R[4]); w[2] = k * R[7];
    } else {
      double k = PI / std::sqrt(2.0 + 2.0 * R[0]);
      w[0] = k * (1.0 + R[0]); w[1] = k * R[3]; w[2] = k * R[6];
    }
  } else {
    double mag;
    double tr3 = tr - 3.0;
    if (tr3 < -1e-6) {
      double th = std::acos(std::fmin(1.0, std::fmax(-1.0, (tr - 1.0) / 2.0)));
      mag = th / (2.0 * std::sin(th));
    } else {
      mag = 0.5 - tr3 / 12.0 + tr3 * tr3 / 60.0;
    }
    w[0] = mag * (R[7] - R[5]); w[1] = mag * (R[2] - R[6]); w[2] = mag * (R[3] - R[1]);
  }
  double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  xi[0] = w[0]; xi[1] = w[1]; xi[2] = w[2];
  if (th < 1e-10) {
    xi[3] = p.t[0]; xi[4] = p.t[1]; xi[5] = p.t[2];
    return;
  }
  double W[3] = {w[0] / th, w[1] / th, w[2] / th};
  double Tan = std::tan(0.5 * th);
  auto cross = [](const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  double WT[3], WWT[3];
  cross(W, p.t, WT);
  cross(W, WT, WWT);
  for (int i = 0; i < 3; ++i) xi[3 + i] = p.t[i] - (0.5 * th) * WT[i] + (1.0 - th / (2.0 * Tan)) * WWT[i];
}

// Solve A x = b (6x6) by LU with partial pivoting; false if singular / non-finite.
// (the reference uses Eigen::LDLT, ref :834-839)
bool solve6_lu(const double Ain[36], const double bin[6], double x[6]) {
  double A[36], b[6];
  std::memcpy(A, Ain, sizeof(A));
  std::memcpy(b, bin, sizeof(b));
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (std::fabs(A[6 * r + c]) > std::fabs(A[6 * piv + c])) piv = r;
    if (!(std::fabs(A[6 * piv + c]) > 0) || !std::isfinite(A[6 * piv + c])) return false;
    if (piv != c) {
      for (int k = 0; k < 6; ++k) std::swap(A[6 * c + k], A[6 * piv + k]);
      std::swap(b[c], b[piv]);
    }
    for (int r = c + 1; r < 6; ++r) {
      double f = A[6 * r + c] / A[6 * c + c];
      for (int k = c; k < 6; ++k) A[6 * r + k] -= f * A[6 * c + k];
      b[r] -= f * b[c];
    }
  }
  for (int r = 5; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < 6; ++k) s -= A[6 * r + k] * x[k];
    x[r] = s / A[6 * r + r];
  }
  for (int i = 0; i < 6; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

// symmetric 6x6 eigen-decomposition (Jacobi): A = Q diag(ev) Q^T, Q row-major
void sym_eig6(const double Ain[36], double ev[6], double Q[36]) {
  double A[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) A[6 * i + j] = 0.5 * (Ain[6 * i + j] + Ain[6 * j + i]);
  for (int i = 0; i < 36; ++i) Q[i] = (i % 7 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) (i == j ? dg : off) += A[6 * i + j] * A[6 * i + j];
    if (off <= 1e-30 * dg || off == 0.0) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (A[6 * p + q] == 0.0) continue;
        double th = (A[6 * q + q] - A[6 * p + p]) / (2.0 * A[6 * p + q]);
        double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {
          double u = A[6 * k + p], v = A[6 * k + q];
          A[6 * k + p] = c * u - s * v; A[6 * k + q] = s * u + c * v;
        }
        for (int k = 0; k < 6; ++k) {
          double u = A[6 * p + k], v = A[6 * q + k];
          A[6 * p + k] = c * u - s * v; A[6 * q + k] = s * u + c * v;
        }
        for (int k = 0; k < 6; ++k) {
          double u = Q[6 * k + p], v = Q[6 * k + q];
          Q[6 * k + p] = c * u - s * v; Q[6 * k + q] = s * u + c * v;
        }
      }
  }
  for (int i = 0; i < 6; ++i) ev[i] = A[7 * i];
}

}  // namespace

extern "C" void oracle_se3_expmap(const double xi[6], double T16[16]) { p3_to16(p3_expmap(xi), T16); }
extern "C" void oracle_se3_logmap(const double T16[16], double xi[6]) { p3_logmap(p3_from16(T16), xi); }

extern "C" void oracle_svn_sample_particles(const double prior16[16], int K, uint64_t seed,
                                            double* particles) {
  const double sig[6] = {0.01, 0.01, 0.02, 0.05, 0.05, 0.05};  // ref :709
  std::mt19937_64 gen(seed);
  std::normal_distribution<double> nd(0.0, 1.0);
  const P3 prior = p3_from16(prior16);
  for (int k = 0; k < K; ++k) {
    double xi[6];
    for (int i = 0; i < 6; ++i) xi[i] = sig[i] * nd(gen);
    p3_to16(p3_compose(prior, p3_expmap(xi)), particles + 16 * (size_t)k);  // prior.retract(sample), ref :715
  }
}

extern "C" void oracle_svn_align(const oracle_grid* g, const float* src, size_t n, size_t stride,
                                 const double prior16[16], double* particles16,
                                 const oracle_params* prm, const oracle_svn_params* svn,
                                 oracle_svn_result* out) {
  std::memset(out, 0, sizeof(*out));
  const int K = svn->particle_count;
  const P3 prior = p3_from16(prior16);
  p3_to16(prior, out->final_pose);
  for (int i = 0; i < 6; ++i) out->final_covariance[7 * i] = 1.0;
  if (K <= 0 || n == 0 || g->leaves.empty()) return;  // ref :682-702: prior, identity covariance

  std::vector<P3> part(K);
  for (int k = 0; k < K; ++k) part[k] = p3_from16(particles16 + 16 * (size_t)k);
  std::vector<oracle_derivs> D(K);
  std::vector<double> upd(6 * (size_t)K);
  P3 mean_cur = prior, mean_last = prior;

  for (int iter = 0; iter < svn->max_iterations; ++iter) {
    mean_last = mean_cur;
    // Stage 1 (ref :758-781)
    for (int k = 0; k < K; ++k) {
      double T64[16];
      p3_to16(part[k], T64);
      float T[16];
      for (int i = 0; i < 16; ++i) T[i] = static_cast<float>(T64[i]);
      const double* R = part[k].R;
      // gtsam Rot3::rpy(): R = Rz(yaw) Ry(pitch) Rx(roll); fed as is into the Rx*Ry*Rz tables (:765-767)
      double p6[6] = {part[k].t[0], part[k].t[1], part[k].t[2], std::atan2(R[7], R[8]),
                      std::atan2(-R[6], std::sqrt(R[7] * R[7] + R[8] * R[8])), std::atan2(R[3], R[0])};
      oracle_derivatives(g, src, n, stride, T, p6, prm, 1, &D[k]);
    }
    // Stage 2 (ref :789-839), GTSAM order [rot, trans]
    for (int k = 0; k < K; ++k) {
      double phi[6] = {0, 0, 0, 0, 0, 0}, Ht[36];
      std::memset(Ht, 0, sizeof(Ht));
      for (int l = 0; l < K; ++l) {
        double d[6];
        p3_logmap(p3_between(part[l], part[k]), d);
        double sq = 0;
        for (int i = 0; i < 6; ++i) sq += d[i] * d[i];
        double kv, kg[6];
        if (svn->kernel_bandwidth <= 1e-12) {  // ref :218-220, :233-235
          kv = (sq < 1e-18) ? 1.0 : 0.0;
          for (int i = 0; i < 6; ++i) kg[i] = 0.0;
        } else {
          kv = std::exp(-sq / svn->kernel_bandwidth);
          for (int i = 0; i < 6; ++i) kg[i] = kv * (-2.0 / svn->kernel_bandwidth) * d[i];
        }
        bool fin = std::isfinite(kv);
        for (int i = 0; i < 6; ++i) fin = fin && std::isfinite(kg[i]);
        if (!fin) continue;
        double gl[6], Hl[36];
        for (int i = 0; i < 3; ++i) { gl[i] = D[l].gradient[3 + i]; gl[3 + i] = D[l].gradient[i]; }
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) Hl[6 * i + j] = D[l].hessian[6 * ((i + 3) % 6) + ((j + 3) % 6)];
        bool gfin = true, hfin = true;
        for (int i = 0; i < 6; ++i) gfin = gfin && std::isfinite(gl[i]);
        for (int i = 0; i < 36; ++i) hfin = hfin && std::isfinite(Hl[i]);
        for (int i = 0; i < 6; ++i) phi[i] += (gfin ? kv * gl[i] : 0.0) + kg[i];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) Ht[6 * i + j] += (hfin ? kv * kv * Hl[6 * i + j] : 0.0) + kg[i] * kg[j];
      }
      for (int i = 0; i < 6; ++i) phi[i] /= static_cast<double>(K);
      for (int i = 0; i < 36; ++i) Ht[i] /= static_cast<double>(K);
      for (int i = 0; i < 6; ++i) Ht[7 * i] += 1e-6;
      double rhs[6], u[6];
      for (int i = 0; i < 6; ++i) rhs[i] = -phi[i];
      bool ok = solve6_lu(Ht, rhs, u);
      for (int i = 0; i < 6; ++i) upd[6 * (size_t)k + i] = ok ? u[i] : 0.0;
    }
    // Stage 3 (ref :848-855)
    for (int k = 0; k < K; ++k) {
      double sc[6];
      bool fin = true;
      for (int i = 0; i < 6; ++i) { sc[i] = svn->step_size * upd[6 * (size_t)k + i]; fin = fin && std::isfinite(sc[i]); }
      if (!fin) continue;
      part[k] = p3_compose(part[k], p3_expmap(sc));
    }
    // mean in the prior's tangent space (ref :865-870), convergence on its update (:877,:893)
    double mxi[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
      double d[6];
      p3_logmap(p3_between(prior, part[k]), d);
      for (int i = 0; i < 6; ++i) mxi[i] += d[i];
    }
    for (int i = 0; i < 6; ++i) mxi[i] /= static_cast<double>(K);
    mean_cur = p3_compose(prior, p3_expmap(mxi));
    out->iterations = iter + 1;
    double d[6], nrm = 0;
    p3_logmap(p3_between(mean_last, mean_cur), d);
    for (int i = 0; i < 6; ++i) nrm += d[i] * d[i];
    nrm = std::sqrt(nrm);
    if (out->n_logged < 128) out->log_mean_update[out->n_logged++] = nrm;
    if (nrm < svn->stop_threshold) { out->converged = 1; break; }
  }
  p3_to16(mean_cur, out->final_pose);
  // sample covariance in the tangent space at the mean (ref :908-930), eigenvalue floor 1e-9 (:932-949)
  double C[36];
  std::memset(C, 0, sizeof(C));
  if (K > 1) {
    std::vector<double> tv(6 * (size_t)K);
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; ++k) {
      p3_logmap(p3_between(mean_cur, part[k]), &tv[6 * (size_t)k]);
      for (int i = 0; i < 6; ++i) m[i] += tv[6 * (size_t)k + i];
    }
    for (int i = 0; i < 6; ++i) m[i] /= static_cast<double>(K);
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) C[6 * i + j] += (tv[6 * (size_t)k + i] - m[i]) * (tv[6 * (size_t)k + j] - m[j]);
    for (int i = 0; i < 36; ++i) C[i] /= static_cast<double>(K - 1);
  } else {
    const double sig[6] = {0.01, 0.01, 0.02, 0.05, 0.05, 0.05};
    for (int i = 0; i < 6; ++i) C[7 * i] = 1e-6 * sig[i] * sig[i];
  }
  double ev[6], Q[36];
  sym_eig6(C, ev, Q);
  bool clamp = false;
  for (int i = 0; i < 6; ++i)
    if (ev[i] < 1e-9) { ev[i] = 1e-9; clamp = true; }
  if (clamp)
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        double s = 0;
        for (int k = 0; k < 6; ++k) s += Q[6 * i + k] * ev[k] * Q[6 * j + k];
        C[6 * i + j] = s;
      }
  std::memcpy(out->final_covariance, C, sizeof(C));
  for (int k = 0; k < K; ++k) p3_to16(part[k], particles16 + 16 * (size_t)k);
}
