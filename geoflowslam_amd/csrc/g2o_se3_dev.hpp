// g2o SE3Quat / robust-kernel arithmetic shared by the LocalBundleAdjustment and PoseOptimization kernels (device side).
// Restates Thirdparty/g2o/g2o/types/se3quat.h, types_six_dof_expmap.h:73-76 and core/robust_kernel_impl.cpp:78-91 with
// Eigen 3.4's quaternion <-> rotation-matrix conversions; double precision like the reference.
#pragma once
#include <hip/hip_runtime.h>

#include "glibc_math.hpp"

namespace gfs_se3 {

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* o) {  // Eigen _transformVector
  const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {  // row-major
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
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
__device__ __forceinline__ void normalize_rotation(double* q) {  // SE3Quat::normalizeRotation
  if (q[3] < 0)
    for (int i = 0; i < 4; i++) q[i] *= -1;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}
__device__ inline void R_to_quat(const double* m, double* q) {
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    // (the three cases spelled out with constant indices: an array indexed by i, j, k at run time would live in scratch memory)
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > (i == 0 ? m[0] : m[4])) i = 2;
    if (i == 0) {  // j = 1, k = 2
      t = sqrt(m[0] - m[4] - m[8] + 1.0);
      q[0] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[7] - m[5]) * t;
      q[1] = (m[3] + m[1]) * t;
      q[2] = (m[6] + m[2]) * t;
    } else if (i == 1) {  // j = 2, k = 0
      t = sqrt(m[4] - m[8] - m[0] + 1.0);
      q[1] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[2] - m[6]) * t;
      q[2] = (m[7] + m[5]) * t;
      q[0] = (m[1] + m[3]) * t;
    } else {  // j = 0, k = 1
      t = sqrt(m[8] - m[0] - m[4] + 1.0);
      q[2] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[3] - m[1]) * t;
      q[0] = (m[2] + m[6]) * t;
      q[1] = (m[5] + m[7]) * t;
    }
  }
}
// VertexSE3Expmap::oplusImpl: estimate <- SE3Quat::exp(update) * estimate (types/se3quat.h:223-257, 101-107)
__device__ inline void pose_oplus(const double* q_in, const double* t_in, const double* u, double* q_out, double* t_out) {
  const double om[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) {
      R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i];
      V[i] = R[i];
    }
  } else {
    // the reference's libm calls, with glibc's bits (glibc_math.hpp)
    const double sn = gfs_glibc::sin(theta), cn = gfs_glibc::cos(theta);
    const double a = sn / theta, b = (1 - cn) / (theta * theta), c = (theta - sn) / gfs_glibc::pow3(theta);
    for (int i = 0; i < 9; i++) {
      R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
      V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * O[i] + c * O2[i];
    }
  }
  double eq[4], et[3];
  R_to_quat(R, eq);
  for (int r = 0; r < 3; r++) et[r] = V[3 * r] * ups[0] + V[3 * r + 1] * ups[1] + V[3 * r + 2] * ups[2];
  normalize_rotation(eq);
  double rt[3];
  quat_rotate(eq, t_in, rt);
  const double* a = eq;
  const double* b = q_in;
  double q[4];
  q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  normalize_rotation(q);
  for (int i = 0; i < 3; i++) t_out[i] = et[i] + rt[i];
  for (int i = 0; i < 4; i++) q_out[i] = q[i];
}

__device__ __forceinline__ void huber(double e, double delta, double* rho0, double* rho1) {  // robust_kernel_impl.cpp:78-91
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    *rho0 = e;
    *rho1 = 1.0;
  } else {
    const double sq = sqrt(e);
    *rho0 = 2 * sq * delta - dsqr;
    *rho1 = delta / sq;
  }
}

}  // namespace gfs_se3
