// ndt_newton.cpp -- host Newton / More-Thuente driver of align().
//
// The reference's loop lives in the un-vendored tier4/ndt_omp submodule
// (extern/ndt_omp is empty); this is written from the published algorithm --
// Magnusson 2009, Algorithm 2 (Newton step on the 6-vector pose, maximising the
// NDT score) and More & Thuente 1994 (line search with sufficient-decrease
// constant 1e-4, curvature constant 0.9, at most 10 trial steps) -- and from
// the reference's call sites: run/pipeline.cpp:464-481,557-568 (setters,
// align, getFinalTransformation, getResult().iteration_num/.hessian) and
// extern/svn_ndt/test/test_svn_ndt.cpp:144-179 (setStepSize, setMaximumIterations,
// computeTransformation, hasConverged, getFinalNumIteration).
#include "ndt_newton.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <cstring>
#include <limits>

#include "ndt_device.h"

namespace ndt {

// ---------------------------------------------------------------------------
// pose <-> matrix, constants
// ---------------------------------------------------------------------------
void pose_to_matrix(const double p[6], float T[16]) {
  const float ar = (float)p[3], ap = (float)p[4], ay = (float)p[5];
  const float sr = sinf(ar), cr = cosf(ar), sp = sinf(ap), cp = cosf(ap), sy = sinf(ay), cy = cosf(ay);
  // M = Rx * Ry (f32), then R = M * Rz (f32)
  const float M[3][3] = {{cp, 0.0f, sp}, {sr * sp, cr, -sr * cp}, {-cr * sp, sr, cr * cp}};
  float R[3][3];
  for (int i = 0; i < 3; ++i) {
    R[i][0] = M[i][0] * cy + M[i][1] * sy;
    R[i][1] = M[i][1] * cy - M[i][0] * sy;
    R[i][2] = M[i][2];
  }
  std::memset(T, 0, sizeof(float) * 16);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[4 * j + i] = R[i][j];
  T[12] = (float)p[0];
  T[13] = (float)p[1];
  T[14] = (float)p[2];
  T[15] = 1.0f;
}

void matrix_to_pose(const float T[16], double p[6]) {
  // Euler angles of R = Rx*Ry*Rz in the range convention of
  // Eigen::Matrix3f::eulerAngles(0,1,2): first raw angle folded into [-pi, 0].
  const float m00 = T[0], m01 = T[4], m02 = T[8], m10 = T[1], m11 = T[5], m12 = T[9], m20 = T[2],
              m21 = T[6], m22 = T[10];
  const float kPi = 3.14159265358979323846f;
  float a0 = atan2f(m12, m22);
  const float c2 = sqrtf(m00 * m00 + m01 * m01);
  float a1;
  if (a0 > 0.0f) {
    a0 -= kPi;
    a1 = atan2f(-m02, -c2);
  } else {
    a1 = atan2f(-m02, c2);
  }
  const float s1 = sinf(a0), c1 = cosf(a0);
  const float a2 = atan2f(s1 * m20 - c1 * m10, c1 * m11 - s1 * m21);
  p[0] = T[12]; p[1] = T[13]; p[2] = T[14];
  p[3] = -a0; p[4] = -a1; p[5] = -a2;
}

void gauss_constants(double resolution, double outlier_ratio, double* d1, double* d2) {
  // ref: svn_ndt_impl.hpp:90-130
  const double tiny = 1e-9;
  double c1 = 10.0 * (1.0 - outlier_ratio);
  double c2 = outlier_ratio / (resolution * resolution * resolution);
  if (c1 <= tiny) c1 = tiny;
  if (c2 <= tiny) c2 = tiny;
  const double d3 = -std::log(c2);
  double a = -std::log(c1 + c2) - d3;
  double b = 1.0;
  if (std::fabs(a) >= tiny) {
    const double inner = c1 * std::exp(-0.5) + c2;
    if (inner > tiny) {
      const double outer = (-std::log(inner) - d3) / a;
      if (outer > tiny) b = -2.0 * std::log(outer);
    }
  }
  if (!std::isfinite(a) || !std::isfinite(b) || !std::isfinite(d3)) { a = 1.0; b = 1.0; }
  *d1 = a;
  *d2 = b;
}

void angle_tables(const double p[6], float jang[24], float hang[45]) {
  // ref: svn_ndt_impl.hpp:260-331.  Row 6 of the second-derivative table keeps the
  // reference's (+sy) third component although d2(x')/dpitch^2 has -sy there.
  double sx, cx, sy, cy, sz, cz;
  const double small = 1e-7;
  if (std::fabs(p[3]) < small) { sx = 0; cx = 1; } else { sx = std::sin(p[3]); cx = std::cos(p[3]); }
  if (std::fabs(p[4]) < small) { sy = 0; cy = 1; } else { sy = std::sin(p[4]); cy = std::cos(p[4]); }
  if (std::fabs(p[5]) < small) { sz = 0; cz = 1; } else { sz = std::sin(p[5]); cz = std::cos(p[5]); }
  const double j[24] = {
      -sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy,
      cx * sz + sx * sy * cz,  cx * cz - sx * sy * sz,  -sx * cy,
      -sy * cz,                sy * sz,                 cy,
      sx * cy * cz,            -sx * cy * sz,           sx * sy,
      -cx * cy * cz,           cx * cy * sz,            -cx * sy,
      -cy * sz,                -cy * cz,                0,
      cx * cz - sx * sy * sz,  -cx * sz - sx * sy * cz, 0,
      sx * cz + cx * sy * sz,  cx * sy * cz - sx * sz,  0};
  const double h[45] = {
      -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, sx * cy,
      -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, -cx * cy,
      cx * cy * cz,            -cx * cy * sz,           cx * sy,
      sx * cy * cz,            -sx * cy * sz,           sx * sy,
      -sx * cz - cx * sy * sz, sx * sz - cx * sy * cz,  0,
      cx * cz - sx * sy * sz,  -sx * sy * cz - cx * sz, 0,
      -cy * cz,                cy * sz,                 sy,
      -sx * sy * cz,           sx * sy * sz,            sx * cy,
      cx * sy * cz,            -cx * sy * sz,           -cx * cy,
      sy * sz,                 sy * cz,                 0,
      -sx * cy * sz,           -sx * cy * cz,           0,
      cx * cy * sz,            cx * cy * cz,            0,
      -cy * cz,                cy * sz,                 0,
      -cx * sz - sx * sy * cz, -cx * cz + sx * sy * sz, 0,
      -sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0};
  for (int i = 0; i < 24; ++i) jang[i] = (float)j[i];
  for (int i = 0; i < 45; ++i) hang[i] = (float)h[i];
}

void unpack_eval(const double* w, Eval* e) {
  e->score = w[EV_SCORE];
  for (int i = 0; i < 6; ++i) e->g[i] = w[EV_G + i];
  int k = EV_H;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      e->H[6 * i + j] = w[k];
      e->H[6 * j + i] = w[k];
      ++k;
    }
  e->nvtl_sum = w[EV_NVTL];
  e->n_with = w[EV_NWITH];
  e->n_pairs = w[EV_NPAIRS];
}

void finish_eval(const ndt_params& prm, const float* reg_pose, const double p[6], bool need_h,
                 Eval* e) {
  if (need_h && prm.add_ridge)  // ref: svn_ndt_impl.hpp:650-653
    for (int i = 0; i < 6; ++i) e->H[7 * i] += 1e-6;
  if (reg_pose) {
    // tier4 ndt_omp longitudinal regularisation (setRegularizationPose, ref:
    // run/pipeline_ligo_tc.cpp:293,531): penalise the offset between the pose and
    // the regularisation pose along the vehicle's heading, weighted by the number
    // of (point, voxel) pairs; f32 arithmetic as upstream [recalled].
    const float k = prm.regularization_scale_factor;
    const float dx = reg_pose[12] - (float)p[0];
    const float dy = reg_pose[13] - (float)p[1];
    const float sy = (float)std::sin(p[5]), cy = (float)std::cos(p[5]);
    const float lon = dy * sy + dx * cy;
    const float wgt = (float)e->n_pairs;
    e->score += (double)(-k * wgt * lon * lon);
    e->g[0] += (double)(k * wgt * 2.0f * cy * lon);
    e->g[1] += (double)(k * wgt * 2.0f * sy * lon);
    if (need_h) {
      e->H[0] += (double)(-k * wgt * 2.0f * cy * cy);
      e->H[1] += (double)(-k * wgt * 2.0f * cy * sy);
      e->H[6] += (double)(-k * wgt * 2.0f * cy * sy);
      e->H[7] += (double)(-k * wgt * 2.0f * sy * sy);
    }
  }
  // ref: svn_ndt_impl.hpp:656-663
  bool gok = true, hok = true;
  for (int i = 0; i < 6; ++i) gok = gok && std::isfinite(e->g[i]);
  for (int i = 0; i < 36; ++i) hok = hok && std::isfinite(e->H[i]);
  if (!gok) std::memset(e->g, 0, sizeof(e->g));
  if (need_h && !hok)
    for (int i = 0; i < 36; ++i) e->H[i] = (i % 7 == 0) ? 1.0 : 0.0;
}

namespace {

// Minimum-norm solution of the symmetric 6x6 system H x = b through the
// eigen-decomposition H = Q L Q^T (two-sided Jacobi); for a symmetric matrix
// this is the SVD pseudo-inverse the reference obtains from
// Eigen::JacobiSVD(H).solve(b), with the same rank threshold 6*eps*|l|max.
// Fast path, taken in the regular case of a comfortably negative-definite H (the Hessian of a
// maximised score near its optimum): Cholesky of -H, 0.1 us instead of the 2-3 us of the
// eigen-decomposition -- which sits on the critical path of every Newton iteration, between
// one evaluation's result and the next launch.  With pivots within 1e-10 of each other no
// singular value is anywhere near the pseudo-inverse's rank threshold, so both routes return
// H^-1 b up to rounding; anything else (indefinite, ill-conditioned, non-finite) takes the
// eigen-decomposition below.
bool solve_negdef6(const double A[6][6], const double b[6], double x[6]) {
  double L[6][6];
  double dmin = std::numeric_limits<double>::infinity(), dmax = 0.0;
  for (int j = 0; j < 6; ++j) {
    double d = -A[j][j];
    for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    dmin = std::fmin(dmin, d);
    dmax = std::fmax(dmax, d);
    const double l = std::sqrt(d), inv = 1.0 / l;
    L[j][j] = l;
    for (int i = j + 1; i < 6; ++i) {
      double v = -A[i][j];
      for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
      L[i][j] = v * inv;
    }
  }
  if (!(dmin > 1e-10 * dmax)) return false;
  double y[6];
  for (int i = 0; i < 6; ++i) {  // L y = -b
    double v = -b[i];
    for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
    y[i] = v / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {  // L^T x = y
    double v = y[i];
    for (int k = i + 1; k < 6; ++k) v -= L[k][i] * x[k];
    x[i] = v / L[i][i];
  }
  for (int i = 0; i < 6; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

void solve_sym6(const double Hin[36], const double b[6], double x[6]) {
  double A[6][6], Q[6][6];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      A[i][j] = 0.5 * (Hin[6 * i + j] + Hin[6 * j + i]);
      Q[i][j] = (i == j) ? 1.0 : 0.0;
    }
  if (solve_negdef6(A, b, x)) return;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, dg = 0;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) (i == j ? dg : off) += A[i][j] * A[i][j];
    if (off <= 1e-30 * dg || off == 0.0) break;
    for (int p = 0; p < 5; ++p)
      for (int q = p + 1; q < 6; ++q) {
        if (A[p][q] == 0.0) continue;
        const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 6; ++k) {
          const double u = A[k][p], v = A[k][q];
          A[k][p] = c * u - s * v;
          A[k][q] = s * u + c * v;
        }
        for (int k = 0; k < 6; ++k) {
          const double u = A[p][k], v = A[q][k];
          A[p][k] = c * u - s * v;
          A[q][k] = s * u + c * v;
        }
        for (int k = 0; k < 6; ++k) {
          const double u = Q[k][p], v = Q[k][q];
          Q[k][p] = c * u - s * v;
          Q[k][q] = s * u + c * v;
        }
      }
  }
  double lmax = 0;
  for (int i = 0; i < 6; ++i) lmax = std::fmax(lmax, std::fabs(A[i][i]));
  const double cut = std::fmax(lmax * 6.0 * std::numeric_limits<double>::epsilon(),
                               std::numeric_limits<double>::min());
  for (int i = 0; i < 6; ++i) x[i] = 0.0;
  for (int j = 0; j < 6; ++j) {
    if (!(std::fabs(A[j][j]) > cut)) continue;
    double qb = 0;
    for (int k = 0; k < 6; ++k) qb += Q[k][j] * b[k];
    qb /= A[j][j];
    for (int i = 0; i < 6; ++i) x[i] += Q[i][j] * qb;
  }
}

// state of one bracket end of the More-Thuente search
struct End {
  double a, f, g;
};

double cubic_step(const End& u, double a_t, double f_t, double g_t) {
  // minimiser of the cubic interpolating (u.a, u.f, u.g) and (a_t, f_t, g_t)
  const double z = 3.0 * (f_t - u.f) / (a_t - u.a) - g_t - u.g;
  const double w = std::sqrt(z * z - g_t * u.g);
  return u.a + (a_t - u.a) * (w - u.g - z) / (g_t - u.g + 2.0 * w);
}

double secant_step(const End& l, double a_t, double g_t) {
  return l.a - (l.a - a_t) / (l.g - g_t) * l.g;
}

// "Trial value selection" of More & Thuente 1994, section 4
double next_trial(const End& l, const End& u, double a_t, double f_t, double g_t) {
  if (f_t > l.f) {  // case 1: higher value -> minimum bracketed
    const double ac = cubic_step(l, a_t, f_t, g_t);
    const double aq = l.a - 0.5 * (l.a - a_t) * l.g / (l.g - (l.f - f_t) / (l.a - a_t));
    return std::fabs(ac - l.a) < std::fabs(aq - l.a) ? ac : 0.5 * (aq + ac);
  }
  if (g_t * l.g < 0) {  // case 2: derivative changes sign
    const double ac = cubic_step(l, a_t, f_t, g_t), as = secant_step(l, a_t, g_t);
    return std::fabs(ac - a_t) >= std::fabs(as - a_t) ? ac : as;
  }
  if (std::fabs(g_t) <= std::fabs(l.g)) {  // case 3: derivative shrinks
    const double ac = cubic_step(l, a_t, f_t, g_t), as = secant_step(l, a_t, g_t);
    const double pick = std::fabs(ac - a_t) < std::fabs(as - a_t) ? ac : as;
    const double lim = a_t + 0.66 * (u.a - a_t);
    return a_t > l.a ? std::fmin(lim, pick) : std::fmax(lim, pick);
  }
  return cubic_step(u, a_t, f_t, g_t);  // case 4
}

// "Updating algorithm"; returns true when the interval has collapsed
bool shrink(End& l, End& u, double a_t, double f_t, double g_t) {
  if (f_t > l.f) { u = {a_t, f_t, g_t}; return false; }
  const double dir = g_t * (l.a - a_t);
  if (dir > 0) { l = {a_t, f_t, g_t}; return false; }
  if (dir < 0) { u = l; l = {a_t, f_t, g_t}; return false; }
  return true;
}

class Solver {
 public:
  Solver(const ndt_params& prm, const EvalFn& fn, bool h_in_trials) : prm_(prm), fn_(fn), h_in_trials_(h_in_trials) {}

  int evaluate(const double p[6], bool need_h) {
    pose_to_matrix(p, T_);
    return evaluate_with(p, T_, need_h);
  }
  int evaluate_with(const double p[6], const float T[16], bool need_h) {
    if (T != T_) std::memcpy(T_, T, sizeof(T_));
    // The evaluator is a pure, bit-reproducible function of (p, T): a repeated request -- the
    // More-Thuente loop re-tries a step clamped to its lower bound up to 10 times, and the
    // reference's loop re-evaluates at the accepted step to get the Hessian -- is answered
    // from the last result instead of a launch.  Same numbers, fewer evaluations.
    if (memo_ && have_last_ && (last_h_ || !need_h) && std::memcmp(last_p_, p, sizeof(last_p_)) == 0 &&
        std::memcmp(last_T_, T_, sizeof(last_T_)) == 0) {
      cur_ = last_;
      ++n_reused_;
      return 0;
    }
    ++n_evals_;
    const int rc = fn_(p, T_, need_h, &cur_);
    have_last_ = rc == 0;
    if (have_last_) {
      std::memcpy(last_p_, p, sizeof(last_p_));
      std::memcpy(last_T_, T_, sizeof(last_T_));
      last_h_ = need_h;
      last_ = cur_;
    }
    return rc;
  }

  // Step length along `dir` from `x`.  phi(a) = -score(x + a dir).  On return
  // cur_ holds score/gradient at the accepted point and H its Hessian.
  int line_search(const double x[6], double dir[6], double a_init, double a_max, double a_min,
                  double score0, const double g0[6], double H[36], double* a_out) {
    const double phi0 = -score0;
    double dphi0 = 0;
    for (int i = 0; i < 6; ++i) dphi0 -= g0[i] * dir[i];
    if (dphi0 >= 0) {
      if (dphi0 == 0) { *a_out = 0; return 0; }
      dphi0 = -dphi0;  // not an ascent direction of the score: search the other way
      for (int i = 0; i < 6; ++i) dir[i] = -dir[i];
    }
    const double mu = 1e-4, nu = 0.9;
    const int max_trials = 10;
    End lo{0, 0, dphi0 - mu * dphi0}, up = lo;  // in terms of psi while the interval is open
    bool collapsed = (a_max - a_min) < 0, open = true;
    double a = std::fmax(std::fmin(a_init, a_max), a_min);
    double xt[6];
    auto probe = [&](bool need_h, double* phi, double* dphi) -> int {
      for (int i = 0; i < 6; ++i) xt[i] = x[i] + dir[i] * a;
      int rc = evaluate(xt, need_h);
      *phi = -cur_.score;
      *dphi = 0;
      for (int i = 0; i < 6; ++i) *dphi -= cur_.g[i] * dir[i];
      return rc;
    };
    double phi, dphi;
    int rc = probe(true, &phi, &dphi);
    if (rc) return rc;
    std::memcpy(H, cur_.H, sizeof(double) * 36);
    double psi = phi - phi0 - mu * dphi0 * a, dpsi = dphi - mu * dphi0;
    int trials = 0;
    while (prm_.use_line_search && !collapsed && trials < max_trials &&
           !(psi <= 0 && dphi <= -nu * dphi0)) {
      a = open ? next_trial(lo, up, a, psi, dpsi) : next_trial(lo, up, a, phi, dphi);
      a = std::fmax(std::fmin(a, a_max), a_min);
      rc = probe(h_in_trials_, &phi, &dphi);
      if (rc) return rc;
      psi = phi - phi0 - mu * dphi0 * a;
      dpsi = dphi - mu * dphi0;
      if (open && psi <= 0 && dpsi >= 0) {
        open = false;  // switch the bracket from psi to phi
        lo.f += phi0 - mu * dphi0 * lo.a; lo.g += mu * dphi0;
        up.f += phi0 - mu * dphi0 * up.a; up.g += mu * dphi0;
      }
      collapsed = open ? shrink(lo, up, a, psi, dpsi) : shrink(lo, up, a, phi, dphi);
      ++trials;
    }
    if (trials && h_in_trials_) {
      std::memcpy(H, cur_.H, sizeof(double) * 36);  // the last trial IS the accepted point
    } else if (trials) {  // the trial evaluations skipped the Hessian: get it at the accepted point
      const double s = cur_.score;
      double g[6];
      std::memcpy(g, cur_.g, sizeof(g));
      rc = evaluate(xt, true);
      if (rc) return rc;
      std::memcpy(H, cur_.H, sizeof(double) * 36);
      cur_.score = s;
      std::memcpy(cur_.g, g, sizeof(g));
    }
    *a_out = a;
    return 0;
  }

  const ndt_params& prm_;
  const EvalFn& fn_;
  bool h_in_trials_;
  Eval cur_;
  float T_[16];
  int n_evals_ = 0;
  // memo of the last evaluation (see evaluate_with)
  bool memo_ = false, have_last_ = false, last_h_ = false;
  double last_p_[6];
  float last_T_[16];
  Eval last_;
  int n_reused_ = 0;
};

}  // namespace

int newton_align(const ndt_params& prm, int64_t n_source_total, const float guess[16],
                 const EvalFn& fn, ndt_result* out, bool hessian_in_trials, IterHistory* history) {
  const auto t0 = std::chrono::steady_clock::now();
  if (history) history->clear();
  auto record = [&](const float* T, const Eval& e) {
    if (history)
      history->push(T, n_source_total > 0 ? e.score / (double)n_source_total : 0.0, e.n_with > 0 ? e.nvtl_sum / e.n_with : 0.0);
  };
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out->final_transformation, guess, sizeof(float) * 16);
  Solver sv(prm, fn, hessian_in_trials);
  sv.memo_ = hessian_in_trials;  // the product path; the plain driver keeps the reference's evaluation count
  double p[6];
  matrix_to_pose(guess, p);
  // the first evaluation transforms the source by the guess matrix itself
  int rc = sv.evaluate_with(p, guess, true);
  if (rc) return rc;
  double score = sv.cur_.score, g[6], H[36];
  std::memcpy(g, sv.cur_.g, sizeof(g));
  std::memcpy(H, sv.cur_.H, sizeof(H));
  record(guess, sv.cur_);

  int iters = 0;
  bool converged = false;
  for (;;) {
    double rhs[6], dp[6];
    for (int i = 0; i < 6; ++i) rhs[i] = -g[i];
    solve_sym6(H, rhs, dp);
    double len = 0;
    for (int i = 0; i < 6; ++i) len += dp[i] * dp[i];
    len = std::sqrt(len);
    if (len == 0 || len != len) {  // zero or NaN step: stop (converged only if not NaN)
      converged = (len == len);
      break;
    }
    for (int i = 0; i < 6; ++i) dp[i] /= len;
    double a = 0;
    rc = sv.line_search(p, dp, len, prm.step_size, prm.trans_epsilon / 2, score, g, H, &a);
    if (rc) return rc;
    score = sv.cur_.score;
    std::memcpy(g, sv.cur_.g, sizeof(g));
    for (int i = 0; i < 6; ++i) p[i] += dp[i] * a;
    std::memcpy(out->final_transformation, sv.T_, sizeof(float) * 16);
    record(sv.T_, sv.cur_);
    const bool stop = iters > prm.max_iterations || (iters && std::fabs(a) < prm.trans_epsilon);
    ++iters;
    if (stop) { converged = true; break; }
  }
  out->converged = converged ? 1 : 0;
  out->iterations = iters;
  out->n_evaluations = sv.n_evals_;
  out->n_evaluations_reused = sv.n_reused_;
  std::memcpy(out->final_pose, p, sizeof(p));
  std::memcpy(out->hessian, H, sizeof(H));
  out->score = score;
  out->transform_probability = n_source_total > 0 ? score / (double)n_source_total : 0.0;
  out->nearest_voxel_transformation_likelihood =
      sv.cur_.n_with > 0 ? sv.cur_.nvtl_sum / sv.cur_.n_with : 0.0;
  out->n_pairs = (int64_t)sv.cur_.n_pairs;
  out->n_points_with_neighbors = (int64_t)sv.cur_.n_with;
  out->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

bool result_covariance(const double H[36], double eps, bool gtsam_order, double cov[36]) {
  // Gauss-Jordan with partial pivoting on [H + eps I | -I]
  double a[6][12];
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      a[i][j] = H[6 * i + j] + (i == j ? eps : 0.0);
      a[i][6 + j] = i == j ? -1.0 : 0.0;
      if (!std::isfinite(a[i][j])) return false;
    }
  }
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    if (a[piv][c] == 0.0) return false;
    if (piv != c)
      for (int j = 0; j < 12; ++j) std::swap(a[piv][j], a[c][j]);
    const double inv = 1.0 / a[c][c];
    for (int j = 0; j < 12; ++j) a[c][j] *= inv;
    for (int r = 0; r < 6; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      if (f == 0.0) continue;
      for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 6; ++i) {
    for (int j = 0; j < 6; ++j) {
      const double v = a[i][6 + j];
      if (!std::isfinite(v)) return false;
      // GTSAM order: C_rr top-left, C_tt bottom-right, the cross blocks stay where they are
      // (exactly what reorderCovarianceForGTSAM does -- it does not transpose them)
      const int bi = i / 3, bj = j / 3;
      const int oi = gtsam_order && bi == bj ? (i + 3) % 6 : i;
      const int oj = gtsam_order && bi == bj ? (j + 3) % 6 : j;
      cov[6 * oi + oj] = v;
    }
  }
  return true;
}

}  // namespace ndt
