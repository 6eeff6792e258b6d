// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of Optimizer::PoseOptimization (reference src/Optimizer.cc:763-1098), conventional-SLAM branch
// (pFrame->mpCamera2 == nullptr): one VertexSE3Expmap, unary edges EdgeSE3ProjectXYZOnlyPose (mono,
// include/OptimizableTypes.h + src/OptimizableTypes.cpp:49-63, Pinhole camera src/CameraModels/Pinhole.cpp:35-41,71-81) and
// g2o::EdgeStereoSE3ProjectXYZOnlyPose (Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:205-236,cpp:339-404}), Huber
// kernels, BlockSolver_6_3 + LinearSolverDense (Eigen::LDLT, solvers/linear_solver_dense.h:64-112) +
// OptimizationAlgorithmLevenberg (core/optimization_algorithm_levenberg.cpp:61-168).  Four rounds of optimize(10), every
// round restarts from the frame's pose, outliers are re-classified after each round (:966-1073); the optimised pose is
// computed but never written back (:1078-1097).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "g2o_se3.hpp"
#include "gfs_oracle.h"

using namespace gfso_se3;

namespace {

// Eigen::LDLT<MatrixXd> on a 6x6 (LDLT.h, ldlt_inplace<Lower>::unblocked): diagonal pivoting by the largest |a_ii|,
// in-place on the lower triangle; returns false when the factorisation reports a sign other than positive
// (LinearSolverDense::solve checks _cholesky.isPositive()).
bool ldlt6_solve_positive(const double H[36] /*row-major symmetric*/, const double b[6], double x[6]) {
  double A[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) A[i][j] = H[6 * i + j];
  int tr[6];
  int sign = 0;  // 0 zero, +1 positive semi-def, -1 negative semi-def, 2 indefinite
  for (int k = 0; k < 6; k++) {
    int p = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++)
      if (std::fabs(A[i][i]) > best) {
        best = std::fabs(A[i][i]);
        p = i;
      }
    tr[k] = p;
    if (p != k) {  // symmetric transposition restricted to the lower triangle
      for (int j = 0; j < k; j++) std::swap(A[k][j], A[p][j]);
      for (int i = p + 1; i < 6; i++) std::swap(A[i][k], A[i][p]);
      std::swap(A[k][k], A[p][p]);
      for (int i = k + 1; i < p; i++) std::swap(A[i][k], A[p][i]);
    }
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; j++) temp[j] = A[j][j] * A[k][j];
      double acc = 0;
      for (int j = 0; j < k; j++) acc += A[k][j] * temp[j];
      A[k][k] -= acc;
      for (int i = k + 1; i < 6; i++) {
        double a2 = 0;
        for (int j = 0; j < k; j++) a2 += A[i][j] * temp[j];
        A[i][k] -= a2;
      }
    }
    const double akk = A[k][k];
    if (std::fabs(akk) > 0)
      for (int i = k + 1; i < 6; i++) A[i][k] /= akk;
    if (sign == 1) {
      if (akk < 0) sign = 2;
    } else if (sign == -1) {
      if (akk > 0) sign = 2;
    } else if (sign == 0) {
      if (akk > 0) sign = 1;
      else if (akk < 0) sign = -1;
    }
  }
  if (sign != 1) return false;
  // solve: x = P^T L^-T D^+ L^-1 P b
  double y[6];
  for (int i = 0; i < 6; i++) y[i] = b[i];
  for (int k = 0; k < 6; k++) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  const double tol = std::numeric_limits<double>::min();  // LDLT::_solve_impl: pseudo-inverse of D
  for (int i = 0; i < 6; i++) y[i] = std::fabs(A[i][i]) > tol ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  for (int k = 5; k >= 0; k--) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < 6; i++) x[i] = y[i];
  return true;
}

struct Ctx {
  const gfso_pose_problem* p;
  Pose T;
  std::vector<double> err;   // 3 per edge
  std::vector<double> chi2;  // per edge: errorᵀ Ω error of the last computeError
  std::vector<int> level;    // 0 active, 1 outlier
  bool robust = true;
};

// computeError of one edge at the current estimate
void edge_error(const Ctx& C, int e, double* r) {
  const gfso_pose_problem& p = *C.p;
  double xc[3];
  map_point(C.T, p.xw + 3 * e, xc);
  const double* obs = p.obs + 3 * e;
  if (p.stereo[e]) {  // cam_project, types_six_dof_expmap.cpp:339-346: float invz, double bf
    const float invz = (float)(1.0 / xc[2]);
    const double u = xc[0] * (double)invz * p.fx + p.cx, v = xc[1] * (double)invz * p.fy + p.cy;
    r[0] = obs[0] - u;
    r[1] = obs[1] - v;
    r[2] = obs[2] - (u - p.bf * (double)invz);
  } else {  // Pinhole::project(Vector3d): float parameters widened to double
    r[0] = obs[0] - (p.fx * xc[0] / xc[2] + p.cx);
    r[1] = obs[1] - (p.fy * xc[1] / xc[2] + p.cy);
    r[2] = 0;
  }
}
double edge_chi2(const Ctx& C, int e, const double* r) {  // BaseEdge::chi2: errorᵀ information error, information = I * invSigma2
  const double w = (double)C.p->inv_sigma2[e];
  return C.p->stereo[e] ? (r[0] * w * r[0] + r[1] * w * r[1] + r[2] * w * r[2]) : (r[0] * w * r[0] + r[1] * w * r[1]);
}
void compute_active_errors(Ctx& C) {
  for (int e = 0; e < C.p->n_obs; e++)
    if (C.level[e] == 0) {
      edge_error(C, e, &C.err[3 * e]);
      C.chi2[e] = edge_chi2(C, e, &C.err[3 * e]);
    }
}
double delta_of(const Ctx& C, int e) { return C.p->stereo[e] ? (double)(float)std::sqrt(7.815) : (double)(float)std::sqrt(5.991); }
double active_robust_chi2(const Ctx& C) {  // SparseOptimizer::activeRobustChi2 (core/sparse_optimizer.cpp:104-122)
  double chi = 0;
  for (int e = 0; e < C.p->n_obs; e++)
    if (C.level[e] == 0) {
      if (C.robust) {
        double r0, r1;
        huber(C.chi2[e], delta_of(C, e), &r0, &r1);
        chi += r0;
      } else {
        chi += C.chi2[e];
      }
    }
  return chi;
}
// linearizeOplus + BaseUnaryEdge::constructQuadraticForm (core/base_unary_edge.hpp:43-72) summed in edge-id order
void build_system(const Ctx& C, double H[36], double b[6]) {
  const gfso_pose_problem& p = *C.p;
  for (int i = 0; i < 36; i++) H[i] = 0;
  for (int i = 0; i < 6; i++) b[i] = 0;
  for (int e = 0; e < p.n_obs; e++) {
    if (C.level[e] != 0) continue;
    double xc[3];
    map_point(C.T, p.xw + 3 * e, xc);
    const double x = xc[0], y = xc[1], z = xc[2];
    double J[18];  // rows x 6, row-major
    int rows;
    if (p.stereo[e]) {  // types_six_dof_expmap.cpp:375-404
      rows = 3;
      const double invz = 1.0 / z, invz_2 = invz * invz;
      J[0] = x * y * invz_2 * p.fx;
      J[1] = -(1 + (x * x * invz_2)) * p.fx;
      J[2] = y * invz * p.fx;
      J[3] = -invz * p.fx;
      J[4] = 0;
      J[5] = x * invz_2 * p.fx;
      J[6] = (1 + y * y * invz_2) * p.fy;
      J[7] = -x * y * invz_2 * p.fy;
      J[8] = -x * invz * p.fy;
      J[9] = 0;
      J[10] = -invz * p.fy;
      J[11] = y * invz_2 * p.fy;
      J[12] = J[0] - p.bf * y * invz_2;
      J[13] = J[1] + p.bf * x * invz_2;
      J[14] = J[2];
      J[15] = J[3];
      J[16] = 0;
      J[17] = J[5] - p.bf * invz_2;
    } else {  // src/OptimizableTypes.cpp:49-63: -projectJac(xyz) * SE3deriv
      rows = 2;
      const double pj[6] = {p.fx / z, 0, -p.fx * x / (z * z), 0, p.fy / z, -p.fy * y / (z * z)};
      const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
      for (int r = 0; r < 2; r++)
        for (int c = 0; c < 6; c++) J[6 * r + c] = -(pj[3 * r] * D[c] + pj[3 * r + 1] * D[6 + c] + pj[3 * r + 2] * D[12 + c]);
    }
    const double w = (double)p.inv_sigma2[e];
    double rho1 = 1.0;
    if (C.robust) {
      double r0;
      huber(C.chi2[e], delta_of(C, e), &r0, &rho1);
    }
    const double* r = &C.err[3 * e];
    // b -= rho1 * Jᵀ Ω e ; H += Jᵀ (rho1 Ω) J   (robustInformation = rho[1] * information, base_edge.h), in the association
    // Eigen gives the two expressions of base_unary_edge.hpp:62-63: ((rho1 Aᵀ) Ω) e and (Aᵀ weightedOmega) A, i.e.
    // H(a, c) = Σ_k (J_ka w') J_kc (the lower triangle, which LDLT reads, is NOT the mirror of the upper one bit for bit).
    // Whether a given Eigen release hoists the scalar rho1 out of the first product cannot be checked here (no Eigen): see header.
    for (int a = 0; a < 6; a++) {
      double s = 0;
      for (int k = 0; k < rows; k++) s += ((rho1 * J[6 * k + a]) * w) * r[k];
      b[a] -= s;
      for (int c = 0; c < 6; c++) {
        double h = 0;
        for (int k = 0; k < rows; k++) h += (J[6 * k + a] * (rho1 * w)) * J[6 * k + c];
        H[6 * a + c] += h;
      }
    }
  }
}

}  // namespace

extern "C" int gfso_pose_optimization(const gfso_pose_problem* p, gfso_pose_solution* s) {
  const int n = p->n_obs;
  for (int e = 0; e < n; e++) s->outlier[e] = 0;  // :819 / :851
  s->n_inliers = 0;
  s->rounds_run = 0;
  s->iterations_run = 0;
  s->avg_reproj_error = 0.f;
  Ctx C;
  C.p = p;
  std::memcpy(C.T.q, p->q, 32);
  std::memcpy(C.T.t, p->t, 24);
  normalize_rotation(C.T.q);  // SE3Quat(q, t) constructor
  std::memcpy(s->q, C.T.q, 32);
  std::memcpy(s->t, C.T.t, 24);
  if (n < 3) return 0;  // nInitialCorrespondences < 3 (:958)
  C.err.assign(3 * (size_t)n, 0.0);
  C.chi2.assign(n, 0.0);
  C.level.assign(n, 0);
  const Pose T0 = C.T;
  const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
  int nBad = 0, nGood = 0;
  const double tau = 1e-5, goodStepUpperScale = 2. / 3., goodStepLowerScale = 1. / 3.;
  const int maxTrialsAfterFailure = 10;
  for (int it = 0; it < p->n_rounds; it++) {
    C.T = T0;  // vSE3->setEstimate(pFrame->GetPose()) — the frame pose is never updated (:968-970)
    C.robust = it <= 2;  // setRobustKernel(0) at the end of round 2 (:1001,:1057)
    // ---- optimizer.initializeOptimization(0); optimizer.optimize(its[it]) ----
    int n_active = 0;
    for (int e = 0; e < n; e++) n_active += C.level[e] == 0;
    double currentLambda = -1, ni = 2;
    int nBadLm = 0;
    if (n_active > 0) {  // optimize() returns at once on an empty active set (core/sparse_optimizer.cpp:360-364)
      for (int iteration = 0; iteration < p->its; iteration++) {
        compute_active_errors(C);
        double currentChi = active_robust_chi2(C);
        double tempChi = currentChi;
        const double iniChi = currentChi;
        double H[36], b[6];
        build_system(C, H, b);
        if (iteration == 0) {
          double maxDiagonal = 0;
          for (int a = 0; a < 6; a++) maxDiagonal = std::max(std::fabs(H[7 * a]), maxDiagonal);
          currentLambda = tau * maxDiagonal;
          ni = 2;
          nBadLm = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
          const Pose backup = C.T;  // push()
          double Hl[36], x[6];
          std::memcpy(Hl, H, sizeof(Hl));
          for (int a = 0; a < 6; a++) Hl[7 * a] += currentLambda;
          const bool ok2 = ldlt6_solve_positive(Hl, b, x);
          if (ok2) pose_oplus(C.T, x);
          compute_active_errors(C);
          tempChi = active_robust_chi2(C);
          if (!ok2) tempChi = std::numeric_limits<double>::max();
          rho = (currentChi - tempChi);
          double scale = 0;
          if (ok2)
            for (int a = 0; a < 6; a++) scale += x[a] * (currentLambda * x[a] + b[a]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && std::isfinite(tempChi)) {
            double alpha = 1. - std::pow((2 * rho - 1), 3);
            alpha = std::min(alpha, goodStepUpperScale);
            const double scaleFactor = std::max(goodStepLowerScale, alpha);
            currentLambda *= scaleFactor;
            ni = 2;
            currentChi = tempChi;
          } else {
            currentLambda *= ni;
            ni *= 2;
            C.T = backup;  // pop(): the estimate is restored, the edges keep the errors of the rejected trial
          }
          qmax++;
        } while (rho < 0 && qmax < maxTrialsAfterFailure);
        s->iterations_run++;
        if (qmax == maxTrialsAfterFailure || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi)
          nBadLm++;
        else
          nBadLm = 0;
        if (nBadLm >= 3) break;
      }
    }
    // ---- classification (:972-1060): mono edges first, then stereo edges, each in creation order ----
    nBad = 0;
    float avg = 0.0f;
    for (int pass = 0; pass < 2; pass++)
      for (int e = 0; e < n; e++) {
        if ((p->stereo[e] != 0) != (pass == 1)) continue;
        if (s->outlier[e]) {  // e->computeError() for edges that were not active
          edge_error(C, e, &C.err[3 * e]);
          C.chi2[e] = edge_chi2(C, e, &C.err[3 * e]);
        }
        const float chi2 = (float)C.chi2[e];
        if (chi2 > (pass ? chi2Stereo : chi2Mono)) {
          s->outlier[e] = 1;
          C.level[e] = 1;
          nBad++;
        } else {
          avg += chi2;
          s->outlier[e] = 0;
          C.level[e] = 0;
          nGood++;
        }
      }
    avg /= (float)nGood;  // nGood is never reset between the rounds (:965,:993,:1053)
    s->avg_reproj_error = avg;
    s->rounds_run = it + 1;
    if (n < 10) break;  // optimizer.edges().size() < 10 (:1073)
  }
  for (int e = 0; e < n; e++) s->chi2[e] = C.chi2[e];
  std::memcpy(s->q, C.T.q, 32);
  std::memcpy(s->t, C.T.t, 24);
  s->n_inliers = n - nBad;
  return s->n_inliers;
}
