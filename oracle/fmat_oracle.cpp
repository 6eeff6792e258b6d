// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of cv::findFundamentalMat(points1, points2, cv::FM_RANSAC, threshold, confidence, mask) as the reference calls
// it: the F check of the optical-flow matcher (src/ORBmatcher.cc:2399-2405, 2463-2469; also :236) and
// Tracking::EstimatePoseByOF (src/Tracking.cc:1973-1974).  OpenCV (calib3d/src/fundam.cpp, ptsetreg.cpp; 4.5.4 semantics) is NOT in
// the container; restated from the published sources from memory:
//   * n >= 15 points -> RANSACPointSetRegistrator(FMEstimatorCallback, modelPoints = 7, threshold, confidence, maxIters = 1000);
//     8 .. 14 points -> LMeDSPointSetRegistrator(same callback, 7, confidence): iteration count fixed by a 45 % outlier
//     assumption, the model with the least MEDIAN error wins, inliers = errors within (2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median))^2,
//   * cv::RNG rng((uint64)-1): state = (uint32)state * 4164903690 + (state >> 32); uniform(0, n) = next() % n,
//   * getSubset: 7 draws, a draw is repeated while the index is already in the subset; checkSubset rejects a subset whose last
//     point is collinear with two earlier ones in either image (haveCollinearPoints); up to 10000 attempts,
//   * every model of the 7-point solution (1..3) is scored with the symmetric epipolar distance (computeError, double arithmetic,
//     stored as float) against threshold^2; a model replaces the best when goodCount > max(best, 6), and then
//     niters = RANSACUpdateNumIters(confidence, outlier ratio, 7, niters); the mask is the inlier set of the best model.
//
// TWO DELIBERATE DIFFERENCES (applied identically by the HIP path): the solutions of the 7-point problem do not depend
// on which basis of the 2-D null space is used nor on how the cubic is solved, so instead of OpenCV's Jacobi SVD + trigonometric
// cv::solveCubic (whose last bits depend on the libm in use) the null space is taken by Gauss-Jordan elimination with full
// pivoting and the real roots of det(l*F1 + (1-l)*F2) = 0 by bracketing + bisection + Newton, all in +, -, *, /, sqrt: the CPU and
// the GPU then produce the same bits.  The fundamental matrices agree with OpenCV's to rounding noise; their ORDER within one
// subset (ascending l here) can differ, which matters only when two models of the same subset tie on the inlier count.
// So that the effect of those two choices is MEASURED and not asserted, gfso_fmat_set_solver(1) switches seven_point() to OpenCV's own
// internals, restated from the 4.5.4 sources: cv::SVDecomp(A, W, U, Vt, MODIFY_A | FULL_UV) on the 7 x 9 system = JacobiSVDImpl_
// (core/src/lapack.cpp) on the rows of A -- one-sided Jacobi sweeps until no pair rotates, singular values sorted descending, and
// the two null vectors produced the way FULL_UV completes the basis: a +-1/9 vector drawn from cv::RNG(0x12345678), projected
// off the earlier rows twice and normalised --, then cv::solveCubic (core/src/mathfuncs.cpp: acos / cos for three real roots, pow
// for one), roots in its order.  tests/test_fmat_solver_variants.py runs RANSAC both ways over many problems and compares masks.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "gfs_oracle.h"

namespace gfs_fmat {

int g_solver_mode = 0;  // 0: Gauss-Jordan null space + bisection (what the HIP path implements), 1: OpenCV's Jacobi SVD + solveCubic

// cv::SVDecomp(A 7x9, FULL_UV): rows 7 and 8 of Vt.  `a` holds A on entry (rows 0 .. 6) and Vt on return (9 rows).
void opencv_svd_null_space(double a[9][9]) {
  const int m = 9, n = 7, n1 = 9;
  const double minval = 2.2250738585072014e-308, eps = 2.220446049250313e-16 * 10;
  double W[9];
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) sd += a[i][k] * a[i][k];
    W[i] = sd;
  }
  const int max_iter = 30;  // std::max(m, 30)
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        double *Ai = a[i], *Aj = a[j];
        double aa = W[i], p = 0, bb = W[j];
        for (int k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (std::abs(p) <= eps * std::sqrt(aa * bb)) continue;
        p *= 2;
        const double beta = aa - bb, gamma = hypot(p, beta);
        double c, sn;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          sn = std::sqrt(delta / gamma);
          c = p / (gamma * sn * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
          sn = p / (gamma * c * 2);
        }
        aa = bb = 0;
        for (int k = 0; k < m; k++) {
          const double t0 = c * Ai[k] + sn * Aj[k];
          const double t1 = -sn * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          aa += t0 * t0;
          bb += t1 * t1;
        }
        W[i] = aa;
        W[j] = bb;
        changed = true;
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) sd += a[i][k] * a[i][k];
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; k++) std::swap(a[i][k], a[j][k]);
    }
  }
  uint64_t state = 0x12345678;  // RNG rng(0x12345678)
  auto next = [&]() {
    state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
    return (unsigned)state;
  };
  for (int i = 0; i < n1; i++) {
    double sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (int k = 0; k < m; k++) a[i][k] = (next() & 256) != 0 ? val0 : -val0;
      for (int iter = 0; iter < 2; iter++)
        for (int j = 0; j < i; j++) {
          sd = 0;
          for (int k = 0; k < m; k++) sd += a[i][k] * a[j][k];
          double asum = 0;
          for (int k = 0; k < m; k++) {
            const double t = a[i][k] - sd * a[j][k];
            a[i][k] = t;
            asum += std::abs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (int k = 0; k < m; k++) a[i][k] *= asum;
        }
      sd = 0;
      for (int k = 0; k < m; k++) sd += a[i][k] * a[i][k];
      sd = std::sqrt(sd);
    }
    const double sc = sd > minval ? 1 / sd : 0.;
    for (int k = 0; k < m; k++) a[i][k] *= sc;
  }
}

// cv::solveCubic(coeffs 1x4, roots): returns the number of roots, in OpenCV's order
int opencv_solve_cubic(const double* cd, double* roots) {
  double a0 = cd[0], a1 = cd[1], a2 = cd[2], a3 = cd[3];
  double x0 = 0., x1 = 0., x2 = 0.;
  int n = 0;
  if (a0 == 0) {
    if (a1 == 0) {
      if (a2 == 0)
        n = a3 == 0 ? -1 : 0;
      else {
        x0 = -a3 / a2;
        n = 1;
      }
    } else {
      double d = a2 * a2 - 4 * a1 * a3;
      if (d >= 0) {
        d = std::sqrt(d);
        const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
        if (std::fabs(q1) > std::fabs(q2)) {
          x0 = q1 / a1;
          x1 = a3 / q1;
        } else {
          x0 = q2 / a1;
          x1 = a3 / q2;
        }
        n = d > 0 ? 2 : 1;
      }
    }
  } else {
    a0 = 1. / a0;
    a1 *= a0;
    a2 *= a0;
    a3 *= a0;
    const double Q = (a1 * a1 - 3 * a2) * (1. / 9);
    const double R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1. / 54);
    const double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    if (d > 0) {
      const double theta = std::acos(R / std::sqrt(Qcubed));
      const double sqrtQ = std::sqrt(Q);
      const double t0 = -2 * sqrtQ, t1 = theta * (1. / 3), t2 = a1 * (1. / 3);
      x0 = t0 * std::cos(t1) - t2;
      x1 = t0 * std::cos(t1 + (2. * 3.1415926535897932384626433832795 / 3)) - t2;
      x2 = t0 * std::cos(t1 + (4. * 3.1415926535897932384626433832795 / 3)) - t2;
      n = 3;
    } else if (d == 0) {
      if (R >= 0) {
        x0 = -2 * std::pow(R, 1. / 3) - a1 / 3;
        x1 = std::pow(R, 1. / 3) - a1 / 3;
      } else {
        x0 = 2 * std::pow(-R, 1. / 3) - a1 / 3;
        x1 = -std::pow(-R, 1. / 3) - a1 / 3;
      }
      x2 = 0;
      n = x0 == x1 ? 1 : 2;
      x1 = x0 == x1 ? 0 : x1;
    } else {
      d = std::sqrt(-d);
      double e = std::pow(d + std::fabs(R), 1. / 3);
      if (R > 0) e = -e;
      x0 = (e + Q / e) - a1 * (1. / 3);
      n = 1;
    }
  }
  roots[0] = x0;
  roots[1] = x1;
  roots[2] = x2;
  return n;
}

struct Rng {  // cv::RNG((uint64)-1)
  uint64_t state = 0xffffffffffffffffull;
  unsigned next() {
    state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32);
    return (unsigned)state;
  }
  int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

bool have_collinear_points(const float* m, const int* idx, int count) {  // modelest / ptsetreg.cpp haveCollinearPoints
  const int i = count - 1;
  for (int j = 0; j < i; j++) {
    const double dx1 = m[2 * idx[j]] - m[2 * idx[i]], dy1 = m[2 * idx[j] + 1] - m[2 * idx[i] + 1];  // float differences
    for (int k = 0; k < j; k++) {
      const double dx2 = m[2 * idx[k]] - m[2 * idx[i]], dy2 = m[2 * idx[k] + 1] - m[2 * idx[i] + 1];
      if (std::fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2)))
        return true;
    }
  }
  return false;
}

bool get_subset(Rng& rng, const float* m1, const float* m2, int count, int* idx, int max_attempts) {  // ...::getSubset
  for (int iters = 0; iters < max_attempts; ++iters) {
    for (int i = 0; i < 7; ++i) {
      int v;
      for (;;) {
        v = rng.uniform(0, count);
        bool dup = false;
        for (int j = 0; j < i; j++) dup |= idx[j] == v;
        if (!dup) break;
      }
      idx[i] = v;
    }
    if (!have_collinear_points(m1, idx, 7) && !have_collinear_points(m2, idx, 7)) return true;  // FMEstimatorCallback::checkSubset
  }
  return false;
}

inline double cubic_eval(double B, double C, double D, double x) { return ((x + B) * x + C) * x + D; }
// the root of the monic cubic inside [lo, hi], p(lo) and p(hi) on different sides of zero: bisection until the interval collapses
inline double cubic_bisect(double B, double C, double D, double lo, double hi) {
  const bool neg_lo = cubic_eval(B, C, D, lo) < 0;
  for (int it = 0; it < 4096; it++) {  // an interval of doubles collapses in < 2200 halvings; the cap only guards against NaN
    const double mid = 0.5 * (lo + hi);
    if (mid == lo || mid == hi) break;
    if ((cubic_eval(B, C, D, mid) < 0) == neg_lo)
      lo = mid;
    else
      hi = mid;
  }
  return hi;
}
// real roots of c0 x^3 + c1 x^2 + c2 x + c3, ascending; sign changes only (a double root that merely touches zero is not one)
int cubic_real_roots(const double* c, double* r) {
  if (c[0] == 0) {
    if (c[1] == 0) {
      if (c[2] == 0) return 0;
      r[0] = -c[3] / c[2];
      return 1;
    }
    const double d = c[2] * c[2] - 4 * c[1] * c[3];
    if (d < 0) return 0;
    const double sq = std::sqrt(d);
    const double x0 = (-c[2] - sq) / (2 * c[1]), x1 = (-c[2] + sq) / (2 * c[1]);
    r[0] = x0 < x1 ? x0 : x1;
    r[1] = x0 < x1 ? x1 : x0;
    return d > 0 ? 2 : 1;
  }
  const double inv = 1. / c[0];
  const double B = c[1] * inv, C = c[2] * inv, D = c[3] * inv;
  double bound = std::fabs(B);
  if (std::fabs(C) > bound) bound = std::fabs(C);
  if (std::fabs(D) > bound) bound = std::fabs(D);
  bound += 1.0;
  if (!(bound < 1.0e300)) return 0;  // a vanishing leading coefficient blew the monic form up (or NaN): no usable root
  double knots[4];
  int nk = 0;
  knots[nk++] = -bound;
  const double disc = B * B - 3 * C;
  if (disc > 0) {
    const double sq = std::sqrt(disc);
    knots[nk++] = (-B - sq) / 3;
    knots[nk++] = (-B + sq) / 3;
  }
  knots[nk++] = bound;
  int n = 0;
  for (int k = 0; k + 1 < nk; k++) {
    const double lo = knots[k], hi = knots[k + 1];
    if (!(lo < hi)) continue;
    if ((cubic_eval(B, C, D, lo) < 0) != (cubic_eval(B, C, D, hi) < 0)) r[n++] = cubic_bisect(B, C, D, lo, hi);
  }
  return n;
}

// FMEstimatorCallback::runKernel for 7 points (run7Point): up to three 3x3 matrices, row-major, in Fm
int seven_point(const float* m1, const float* m2, const int* idx, double* Fm) {
  double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
  for (int i = 0; i < 7; i++) {
    c1x += m1[2 * idx[i]];
    c1y += m1[2 * idx[i] + 1];
    c2x += m2[2 * idx[i]];
    c2y += m2[2 * idx[i] + 1];
  }
  const double t = 1. / 7;
  c1x *= t;
  c1y *= t;
  c2x *= t;
  c2y *= t;
  double s1 = 0, s2 = 0;
  for (int i = 0; i < 7; i++) {
    const double ax = m1[2 * idx[i]] - c1x, ay = m1[2 * idx[i] + 1] - c1y, bx = m2[2 * idx[i]] - c2x, by = m2[2 * idx[i] + 1] - c2y;
    s1 += std::sqrt(ax * ax + ay * ay);
    s2 += std::sqrt(bx * bx + by * by);
  }
  s1 *= t;
  s2 *= t;
  if (s1 < 1.1920928955078125e-07 || s2 < 1.1920928955078125e-07) return 0;
  s1 = std::sqrt(2.) / s1;
  s2 = std::sqrt(2.) / s2;
  double a[7][9];
  for (int i = 0; i < 7; i++) {
    const double x0 = (m1[2 * idx[i]] - c1x) * s1, y0 = (m1[2 * idx[i] + 1] - c1y) * s1;
    const double x1 = (m2[2 * idx[i]] - c2x) * s2, y1 = (m2[2 * idx[i] + 1] - c2y) * s2;
    a[i][0] = x1 * x0;
    a[i][1] = x1 * y0;
    a[i][2] = x1;
    a[i][3] = y1 * x0;
    a[i][4] = y1 * y0;
    a[i][5] = y1;
    a[i][6] = x0;
    a[i][7] = y0;
    a[i][8] = 1;
  }
  double f1[9], f2[9];
  if (g_solver_mode == 1) {  // OpenCV's own: SVDecomp(A, W, U, Vt, MODIFY_A | FULL_UV); f1 = Vt row 7, f2 = Vt row 8
    double at[9][9];
    for (int i = 0; i < 7; i++)
      for (int j = 0; j < 9; j++) at[i][j] = a[i][j];
    opencv_svd_null_space(at);
    for (int j = 0; j < 9; j++) {
      f1[j] = at[7][j];
      f2[j] = at[8][j];
    }
  } else {
    // null space of the 7 x 9 system: Gauss-Jordan with full pivoting; perm = column order, the last two columns stay free
    int perm[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
    for (int r = 0; r < 7; r++) {
      int pi = r, pj = r;
      double pv = -1;
      for (int i = r; i < 7; i++)
        for (int j = r; j < 9; j++)
          if (std::fabs(a[i][j]) > pv) {
            pv = std::fabs(a[i][j]);
            pi = i;
            pj = j;
          }
      if (!(pv > 0)) return 0;
      for (int j = 0; j < 9; j++) {
        const double tmp = a[r][j];
        a[r][j] = a[pi][j];
        a[pi][j] = tmp;
      }
      for (int i = 0; i < 7; i++) {
        const double tmp = a[i][r];
        a[i][r] = a[i][pj];
        a[i][pj] = tmp;
      }
      {
        const int tmp = perm[r];
        perm[r] = perm[pj];
        perm[pj] = tmp;
      }
      const double ip = 1. / a[r][r];
      for (int j = 0; j < 9; j++) a[r][j] *= ip;
      for (int i = 0; i < 7; i++) {
        if (i == r) continue;
        const double f = a[i][r];
        for (int j = 0; j < 9; j++) a[i][j] -= f * a[r][j];
      }
    }
    for (int j = 0; j < 9; j++) f1[j] = f2[j] = 0;
    for (int r = 0; r < 7; r++) {
      f1[perm[r]] = -a[r][7];
      f2[perm[r]] = -a[r][8];
    }
    f1[perm[7]] = 1;
    f2[perm[8]] = 1;
  }
  // det(l * f1 + (1 - l) * f2) = 0 as in run7Point: f1 := f1 - f2, polynomial c[0] l^3 + c[1] l^2 + c[2] l + c[3]
  for (int i = 0; i < 9; i++) f1[i] -= f2[i];
  double c[4];
  double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
  c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
  c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
         f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
         f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
  t0 = f1[4] * f1[8] - f1[5] * f1[7];
  t1 = f1[3] * f1[8] - f1[5] * f1[6];
  t2 = f1[3] * f1[7] - f1[4] * f1[6];
  c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
         f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
         f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
  c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
  double roots[3];
  const int n = g_solver_mode == 1 ? opencv_solve_cubic(c, roots) : cubic_real_roots(c, roots);
  if (n < 1 || n > 3) return n < 0 ? 0 : n;
  for (int k = 0; k < n; k++) {
    double* F = Fm + 9 * k;
    double lambda = roots[k], mu = 1.;
    const double s = f1[8] * roots[k] + f2[8];
    if (std::fabs(s) > 2.220446049250313e-16) {
      mu = 1. / s;
      lambda *= mu;
      F[8] = 1.;
    } else {
      F[8] = 0.;
    }
    for (int i = 0; i < 8; i++) F[i] = f1[i] * lambda + f2[i] * mu;
    // de-normalise: F = T2^T F T1 with T = [s 0 -s*cx; 0 s -s*cy; 0 0 1]
    double G[9];  // F * T1
    for (int rr = 0; rr < 3; rr++) {
      G[3 * rr] = F[3 * rr] * s1;
      G[3 * rr + 1] = F[3 * rr + 1] * s1;
      G[3 * rr + 2] = F[3 * rr] * (-s1 * c1x) + F[3 * rr + 1] * (-s1 * c1y) + F[3 * rr + 2];
    }
    for (int cc = 0; cc < 3; cc++) {  // T2^T * G
      F[cc] = s2 * G[cc];
      F[3 + cc] = s2 * G[3 + cc];
      F[6 + cc] = (-s2 * c2x) * G[cc] + (-s2 * c2y) * G[3 + cc] + G[6 + cc];
    }
    if (std::fabs(F[8]) > 1.1920928955078125e-07) {
      const double sc = 1. / F[8];
      for (int i = 0; i < 9; i++) F[i] *= sc;
    }
  }
  return n;
}

inline float epipolar_error(const double* F, const float* p1, const float* p2) {  // FMEstimatorCallback::computeError
  double a = F[0] * p1[0] + F[1] * p1[1] + F[2];
  double b = F[3] * p1[0] + F[4] * p1[1] + F[5];
  double c = F[6] * p1[0] + F[7] * p1[1] + F[8];
  const double s2 = 1. / (a * a + b * b);
  const double d2 = p2[0] * a + p2[1] * b + c;
  a = F[0] * p2[0] + F[3] * p2[1] + F[6];
  b = F[1] * p2[0] + F[4] * p2[1] + F[7];
  c = F[2] * p2[0] + F[5] * p2[1] + F[8];
  const double s1 = 1. / (a * a + b * b);
  const double d1 = p1[0] * a + p1[1] * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (float)(e1 > e2 ? e1 : e2);
}

int update_num_iters(double p, double ep, int max_iters) {  // RANSACUpdateNumIters(p, ep, 7, maxIters)
  p = p < 0 ? 0 : (p > 1 ? 1 : p);
  ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
  double num = 1 - p > 2.2250738585072014e-308 ? 1 - p : 2.2250738585072014e-308;
  double denom = 1 - std::pow(1 - ep, 7);
  if (denom < 2.2250738585072014e-308) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace gfs_fmat

extern "C" void gfso_fmat_set_solver(int mode) { gfs_fmat::g_solver_mode = mode == 1 ? 1 : 0; }

extern "C" int gfso_fundamental_ransac(const float* pts1, const float* pts2, int n, double threshold, double confidence, int max_iters,
                                       uint8_t* mask, double* F_out, int* iterations_run) {
  using namespace gfs_fmat;
  if (n < 8) return -2;  // 7 points: the 7-point solutions themselves; fewer: no result (neither is restated)
  if (threshold <= 0) threshold = 3;
  if (confidence < 2.220446049250313e-16 || confidence > 1 - 2.220446049250313e-16) confidence = 0.99;
  if (n < 15) {
    // findFundamentalMat: "(method & ~3) == FM_RANSAC && npoints >= 15" -> RANSAC, else LMeDSPointSetRegistrator(cb, 7, confidence)
    // ::run — least median of the errors instead of a consensus count, threshold unused
    Rng rng;
    int niters = update_num_iters(confidence, 0.45, 1000);
    if (niters < 3) niters = 3;
    double min_median = 1.7976931348623157e308;
    double best[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int iters = 0;
    for (int iter = 0; iter < niters; iter++) {
      int idx[7];
      if (!get_subset(rng, pts1, pts2, n, idx, 1000)) {
        if (iter == 0) return 0;
        break;
      }
      iters++;
      double Fm[27];
      const int nm = seven_point(pts1, pts2, idx, Fm);
      for (int m = 0; m < nm; m++) {
        float e[16];
        for (int i = 0; i < n; i++) e[i] = epipolar_error(Fm + 9 * m, pts1 + 2 * i, pts2 + 2 * i);
        for (int i = 1; i < n; i++) {  // the (n / 2)-th smallest error: what std::nth_element leaves at position n / 2
          const float v = e[i];
          int j = i;
          for (; j > 0 && e[j - 1] > v; j--) e[j] = e[j - 1];
          e[j] = v;
        }
        const double median = e[n / 2];
        if (median < min_median) {
          min_median = median;
          std::memcpy(best, Fm + 9 * m, sizeof(best));
        }
      }
    }
    if (iterations_run) *iterations_run = iters;
    if (!(min_median < 1.7976931348623157e308)) {
      for (int i = 0; i < n; i++) mask[i] = 0;
      return 0;
    }
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 7)) * std::sqrt(min_median);
    if (sigma < 0.001) sigma = 0.001;
    const float ts = (float)(sigma * sigma);
    int count = 0;
    for (int i = 0; i < n; i++) count += mask[i] = epipolar_error(best, pts1 + 2 * i, pts2 + 2 * i) <= ts;
    if (F_out) {
      if (count >= 7)
        std::memcpy(F_out, best, sizeof(best));  // "result = count >= modelPoints": fewer inliers -> an empty matrix is returned
      else
        std::memset(F_out, 0, sizeof(best));
    }
    return count;
  }
  Rng rng;
  int niters = max_iters > 1 ? max_iters : 1;
  int max_good = 0, iters = 0;
  double best[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  const float t2 = (float)(threshold * threshold);  // findInliers compares float errors with a float threshold
  for (int iter = 0; iter < niters; iter++) {
    int idx[7];
    if (!get_subset(rng, pts1, pts2, n, idx, 10000)) {
      if (iter == 0) return 0;
      break;
    }
    iters++;
    double Fm[27];
    const int nm = seven_point(pts1, pts2, idx, Fm);
    for (int m = 0; m < nm; m++) {
      int good = 0;
      for (int i = 0; i < n; i++) good += epipolar_error(Fm + 9 * m, pts1 + 2 * i, pts2 + 2 * i) <= t2;
      if (good > (max_good > 6 ? max_good : 6)) {
        std::memcpy(best, Fm + 9 * m, sizeof(best));
        max_good = good;
        niters = update_num_iters(confidence, (double)(n - good) / n, niters);
      }
    }
  }
  if (iterations_run) *iterations_run = iters;
  if (max_good <= 0) {
    for (int i = 0; i < n; i++) mask[i] = 0;
    return 0;
  }
  for (int i = 0; i < n; i++) mask[i] = epipolar_error(best, pts1 + 2 * i, pts2 + 2 * i) <= t2;
  if (F_out) std::memcpy(F_out, best, sizeof(best));
  return max_good;
}
