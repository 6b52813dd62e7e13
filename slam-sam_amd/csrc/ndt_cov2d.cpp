// ndt_cov2d.cpp -- the 2-D (x, y) covariance estimators of tier4 ndt_omp.
//
// [RECALLED] Their source (src/estimate_covariance/estimate_covariance.cpp) is in the un-vendored
// submodule; the reference tree only names the file in its build (ref: CMakeLists.txt:40) and no
// driver calls it.  Restated from the published algorithm of that file as recalled:
//   * Laplace approximation: cov_xy = -(H[0:2, 0:2])^-1 of the result's Hessian;
//   * poses to search: offsets rotated onto the principal axes of that covariance (angle of the
//     eigenvector of its SMALLER eigenvalue) and added to the result's translation;
//   * MULTI_NDT: re-align from every pose; unbiased sample covariance of the (x, y) of the main
//     result and the re-aligned results;
//   * MULTI_NDT_SCORE: no re-alignment -- the nearest-voxel transformation likelihood of the
//     source at every pose (here: ONE batched score-only launch for all poses), weights
//     softmax(score / temperature), weighted mean and covariance.
// Host code over the C-ABI; the GPU work is ndt_align / ndt_score_transforms.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/ndt_hip.h"

namespace {

bool inv2(const double m[4], double out[4]) {
  const double det = m[0] * m[3] - m[1] * m[2];
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  out[0] = m[3] / det; out[1] = -m[1] / det; out[2] = -m[2] / det; out[3] = m[0] / det;
  return std::isfinite(out[0]) && std::isfinite(out[1]) && std::isfinite(out[2]) && std::isfinite(out[3]);
}

}  // namespace

extern "C" {

int ndt_xy_covariance_laplace(const double hessian36[36], double cov_xy[4]) {
  if (!hessian36 || !cov_xy) return NDT_ERR_INVALID_ARG;
  const double hxy[4] = {hessian36[0], hessian36[1], hessian36[6], hessian36[7]};
  double inv[4];
  if (!inv2(hxy, inv)) return NDT_ERR_INVALID_ARG;
  for (int i = 0; i < 4; ++i) cov_xy[i] = -inv[i];
  return NDT_OK;
}

int ndt_propose_poses_to_search(const ndt_result* r, const double* offsets_x, const double* offsets_y, int n,
                                float* poses16) {
  if (!r || !offsets_x || !offsets_y || !poses16 || n <= 0) return NDT_ERR_INVALID_ARG;
  double cov[4];
  int rc = ndt_xy_covariance_laplace(r->hessian, cov);
  if (rc) return rc;
  // eigenvector of the smaller eigenvalue of the symmetric 2x2 (SelfAdjointEigenSolver orders
  // ascending and col(0) is taken), its angle, the rotation by that angle
  const double a = cov[0], b = 0.5 * (cov[1] + cov[2]), d = cov[3];
  const double tr = a + d, df = a - d;
  const double root = std::sqrt(0.25 * df * df + b * b);
  const double l0 = 0.5 * tr - root;  // smaller eigenvalue
  double vx, vy;
  if (std::fabs(b) > 1e-300) { vx = b; vy = l0 - a; }
  else if (a <= d) { vx = 1.0; vy = 0.0; }
  else { vx = 0.0; vy = 1.0; }
  const double th = std::atan2(vy, vx);
  const double c = std::cos(th), s = std::sin(th);
  for (int i = 0; i < n; ++i) {
    float* T = poses16 + 16 * (size_t)i;
    std::memcpy(T, r->final_transformation, sizeof(float) * 16);
    const double ox = c * offsets_x[i] - s * offsets_y[i], oy = s * offsets_x[i] + c * offsets_y[i];
    T[12] += (float)ox;
    T[13] += (float)oy;
  }
  return NDT_OK;
}

int ndt_xy_covariance_multi_ndt(ndt_handle* h, const ndt_result* main_result, const float* poses16, int n,
                                double mean_xy[2], double cov_xy[4]) {
  if (!h || !main_result || !poses16 || n <= 0 || !mean_xy || !cov_xy) return NDT_ERR_INVALID_ARG;
  std::vector<double> px((size_t)n + 1), py((size_t)n + 1);
  px[0] = main_result->final_transformation[12];
  py[0] = main_result->final_transformation[13];
  for (int i = 0; i < n; ++i) {
    ndt_result sub;
    int rc = ndt_align(h, poses16 + 16 * (size_t)i, &sub);
    if (rc) return rc;
    px[(size_t)i + 1] = sub.final_transformation[12];
    py[(size_t)i + 1] = sub.final_transformation[13];
  }
  const int m = n + 1;
  double mx = 0, my = 0;
  for (int i = 0; i < m; ++i) { mx += px[i]; my += py[i]; }
  mx /= m; my /= m;
  double c[4] = {0, 0, 0, 0};
  for (int i = 0; i < m; ++i) {
    const double dx = px[i] - mx, dy = py[i] - my;
    c[0] += dx * dx; c[1] += dx * dy; c[2] += dy * dx; c[3] += dy * dy;
  }
  for (int i = 0; i < 4; ++i) cov_xy[i] = c[i] / (double)(m - 1);  // unbiased
  mean_xy[0] = mx; mean_xy[1] = my;
  return NDT_OK;
}

int ndt_xy_covariance_multi_ndt_score(ndt_handle* h, const ndt_result* main_result, const float* poses16, int n,
                                      double temperature, double mean_xy[2], double cov_xy[4]) {
  if (!h || !main_result || !poses16 || n <= 0 || !mean_xy || !cov_xy || !(temperature > 0.0)) return NDT_ERR_INVALID_ARG;
  std::vector<ndt_score> sc((size_t)n);
  int rc = ndt_score_transforms(h, poses16, n, sc.data());  // one launch for all poses
  if (rc) return rc;
  const int m = n + 1;
  std::vector<double> px((size_t)m), py((size_t)m), w((size_t)m);
  px[0] = main_result->final_transformation[12];
  py[0] = main_result->final_transformation[13];
  w[0] = main_result->nearest_voxel_transformation_likelihood;
  for (int i = 0; i < n; ++i) {
    px[(size_t)i + 1] = poses16[16 * (size_t)i + 12];
    py[(size_t)i + 1] = poses16[16 * (size_t)i + 13];
    w[(size_t)i + 1] = sc[(size_t)i].nearest_voxel_transformation_likelihood;
  }
  double wmax = w[0];
  for (int i = 1; i < m; ++i) wmax = std::fmax(wmax, w[i]);
  double sum = 0;
  for (int i = 0; i < m; ++i) { w[i] = std::exp((w[i] - wmax) / temperature); sum += w[i]; }
  double mx = 0, my = 0;
  for (int i = 0; i < m; ++i) { w[i] /= sum; mx += w[i] * px[i]; my += w[i] * py[i]; }
  double c[4] = {0, 0, 0, 0};
  for (int i = 0; i < m; ++i) {
    const double dx = px[i] - mx, dy = py[i] - my;
    c[0] += w[i] * dx * dx; c[1] += w[i] * dx * dy; c[2] += w[i] * dy * dx; c[3] += w[i] * dy * dy;
  }
  std::memcpy(cov_xy, c, sizeof(c));
  mean_xy[0] = mx; mean_xy[1] = my;
  return NDT_OK;
}

}  // extern "C"
