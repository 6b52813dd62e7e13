// ndt_se3.h -- minimal SE(3) algebra for the SVN-NDT outer loop (host side, f64).
// Stands in for the gtsam::Pose3 operations the reference uses
// (ref: extern/svn_ndt/include/svn_ndt_impl.hpp:214-244,715,765-767,853,865-877,915):
// Expmap / Logmap with tangent order [omega, v], compose, between, retract (= compose with
// Expmap, GTSAM's default GTSAM_POSE3_EXPMAP chart) and Rot3::rpy().
#pragma once

#include <cmath>

namespace ndt {
namespace se3 {

struct Pose {
  double R[3][3];
  double t[3];
};

inline Pose identity() {
  Pose p{};
  p.R[0][0] = p.R[1][1] = p.R[2][2] = 1.0;
  return p;
}

inline Pose from_colmajor(const double T[16]) {
  Pose p;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) p.R[i][j] = T[4 * j + i];
    p.t[i] = T[12 + i];
  }
  return p;
}

inline void to_colmajor(const Pose& p, double T[16]) {
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) T[4 * c + r] = (r == c) ? 1.0 : 0.0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[4 * j + i] = p.R[i][j];
    T[12 + i] = p.t[i];
  }
}

inline Pose compose(const Pose& a, const Pose& b) {
  Pose c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c.R[i][j] = a.R[i][0] * b.R[0][j] + a.R[i][1] * b.R[1][j] + a.R[i][2] * b.R[2][j];
    c.t[i] = a.R[i][0] * b.t[0] + a.R[i][1] * b.t[1] + a.R[i][2] * b.t[2] + a.t[i];
  }
  return c;
}

inline Pose inverse(const Pose& a) {
  Pose c;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c.R[i][j] = a.R[j][i];
    c.t[i] = -(a.R[0][i] * a.t[0] + a.R[1][i] * a.t[1] + a.R[2][i] * a.t[2]);
  }
  return c;
}

inline Pose between(const Pose& a, const Pose& b) { return compose(inverse(a), b); }

// coefficients of Rodrigues' formula with series fall-backs
inline void abc(double th2, double* A, double* B, double* Cc) {
  const double th = std::sqrt(th2);
  if (th2 < 1e-12) {
    *A = 1.0 - th2 / 6.0;
    *B = 0.5 - th2 / 24.0;
    *Cc = 1.0 / 6.0 - th2 / 120.0;
  } else {
    *A = std::sin(th) / th;
    *B = (1.0 - std::cos(th)) / th2;
    *Cc = (th - std::sin(th)) / (th2 * th);
  }
}

// Exp: xi = [omega(3), v(3)] -> Pose, R = I + A K + B K^2, t = (I + B K + C K^2) v
inline Pose expmap(const double xi[6]) {
  const double wx = xi[0], wy = xi[1], wz = xi[2];
  const double th2 = wx * wx + wy * wy + wz * wz;
  double A, B, Cc;
  abc(th2, &A, &B, &Cc);
  const double K[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
  double K2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
  Pose p;
  for (int i = 0; i < 3; ++i) {
    double ti = 0;
    for (int j = 0; j < 3; ++j) {
      const double I = (i == j) ? 1.0 : 0.0;
      p.R[i][j] = I + A * K[i][j] + B * K2[i][j];
      ti += (I + B * K[i][j] + Cc * K2[i][j]) * xi[3 + j];
    }
    p.t[i] = ti;
  }
  return p;
}

// Log of a rotation matrix -> rotation vector
inline void so3_log(const double R[3][3], double w[3]) {
  const double tr = R[0][0] + R[1][1] + R[2][2];
  const double ax = R[2][1] - R[1][2], ay = R[0][2] - R[2][0], az = R[1][0] - R[0][1];
  const double s = 0.5 * std::sqrt(ax * ax + ay * ay + az * az);  // sin(theta)
  const double c = 0.5 * (tr - 1.0);                              // cos(theta)
  const double th = std::atan2(s, c);
  if (s > 1e-6) {  // generic: axis from the antisymmetric part
    const double k = th / (2.0 * s);
    w[0] = k * ax; w[1] = k * ay; w[2] = k * az;
  } else if (c > 0.0) {  // theta ~ 0: w = (1/2 + theta^2/12) * vee(R - R^T)
    const double k = 0.5 + th * th / 12.0;
    w[0] = k * ax; w[1] = k * ay; w[2] = k * az;
  } else {  // theta ~ pi: axis from the symmetric part R + I = 2 n n^T (up to O(pi - theta))
    int m = 0;
    if (R[1][1] > R[m][m]) m = 1;
    if (R[2][2] > R[m][m]) m = 2;
    double n[3];
    const double d = std::sqrt(std::fmax(0.0, 0.5 * (R[m][m] + 1.0)));
    for (int i = 0; i < 3; ++i) n[i] = 0.25 * (R[i][m] + R[m][i]) / d;
    n[m] = d;
    // orient the axis consistently with the (tiny) antisymmetric part
    if (n[0] * ax + n[1] * ay + n[2] * az < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
    w[0] = th * n[0]; w[1] = th * n[1]; w[2] = th * n[2];
  }
}

// Log: Pose -> xi = [omega, v], v = V^-1 t with V^-1 = I - K/2 + D K^2
inline void logmap(const Pose& p, double xi[6]) {
  double w[3];
  so3_log(p.R, w);
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double th = std::sqrt(th2);
  double D;
  if (th < 1e-5) D = 1.0 / 12.0 + th2 / 720.0;
  else D = (1.0 - th * std::cos(0.5 * th) / (2.0 * std::sin(0.5 * th))) / th2;
  const double K[3][3] = {{0, -w[2], w[1]}, {w[2], 0, -w[0]}, {-w[1], w[0], 0}};
  double K2[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
  for (int i = 0; i < 3; ++i) {
    xi[i] = w[i];
    double v = 0;
    for (int j = 0; j < 3; ++j) v += (((i == j) ? 1.0 : 0.0) - 0.5 * K[i][j] + D * K2[i][j]) * p.t[j];
    xi[3 + i] = v;
  }
}

inline Pose retract(const Pose& p, const double xi[6]) { return compose(p, expmap(xi)); }

// gtsam::Rot3::rpy(): R = Rz(yaw) Ry(pitch) Rx(roll)
inline void rpy(const Pose& p, double out[3]) {
  out[0] = std::atan2(p.R[2][1], p.R[2][2]);
  out[1] = std::atan2(-p.R[2][0], std::sqrt(p.R[2][1] * p.R[2][1] + p.R[2][2] * p.R[2][2]));
  out[2] = std::atan2(p.R[1][0], p.R[0][0]);
}

}  // namespace se3
}  // namespace ndt
