// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// g2o SE3Quat arithmetic shared by the LocalBundleAdjustment and PoseOptimization restatements
// (Thirdparty/g2o/g2o/types/se3quat.h, types_six_dof_expmap.h:73-76) with Eigen 3.4's quaternion <-> matrix conversions.
#pragma once
#include <cmath>
#include <cstring>

namespace gfso_se3 {

struct Pose {
  double q[4];  // x, y, z, w
  double t[3];
};

// Eigen QuaternionBase::_transformVector: v + w * (2 q x v) + q x (2 q x v)
inline void quat_rotate(const double* q, const double* v, double* o) {
  const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
inline void quat_to_R(const double* q, double* R /*row-major 3x3*/) {  // Eigen toRotationMatrix
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
               tzz = tz * z;
  R[0] = 1 - (tyy + tzz);
  R[1] = txy - twz;
  R[2] = txz + twy;
  R[3] = txy + twz;
  R[4] = 1 - (txx + tzz);
  R[5] = tyz - twx;
  R[6] = txz - twy;
  R[7] = tyz + twx;
  R[8] = 1 - (txx + tyy);
}
inline void R_to_quat(const double* m /*row-major*/, double* q) {  // Eigen quaternionbase_assign_impl<Other,3,3>
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
    q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
    q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
  }
}
inline void normalize_rotation(double* q) {  // SE3Quat::normalizeRotation, types/se3quat.h:280-285
  if (q[3] < 0)
    for (int i = 0; i < 4; i++) q[i] *= -1;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

// SE3Quat::exp (types/se3quat.h:223-257) followed by exp(update) * estimate (types_six_dof_expmap.h:73-76)
inline void pose_oplus(Pose& P, const double* u) {
  const double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double O[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};  // row-major skew
  double O2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
    std::memcpy(V, R, sizeof(R));
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
                 c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 9; i++) {
      R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
      V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
    }
  }
  Pose E;
  R_to_quat(R, E.q);
  for (int r = 0; r < 3; r++) E.t[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
  normalize_rotation(E.q);  // SE3Quat(q, t) ctor
  // result = E * P (se3quat.h:101-107): t = E.t + E.r * P.t ; r = E.r * P.r ; normalizeRotation
  double rt[3];
  quat_rotate(E.q, P.t, rt);
  const double* a = E.q;
  const double* b = P.q;
  double q[4];
  q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  for (int i = 0; i < 3; i++) P.t[i] = E.t[i] + rt[i];
  std::memcpy(P.q, q, sizeof(q));
  normalize_rotation(P.q);
}

inline void map_point(const Pose& T, const double* X, double* o) {  // SE3Quat::map (se3quat.h:217-220)
  quat_rotate(T.q, X, o);
  o[0] += T.t[0];
  o[1] += T.t[1];
  o[2] += T.t[2];
}

// RobustKernelHuber::robustify (core/robust_kernel_impl.cpp:78-91): rho[0], rho[1]
inline void huber(double e, double delta, double* rho0, double* rho1) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    *rho0 = e;
    *rho1 = 1.0;
  } else {
    const double sqrte = std::sqrt(e);
    *rho0 = 2 * sqrte * delta - dsqr;
    *rho1 = delta / sqrte;
  }
}

}  // namespace gfso_se3
