// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of the numeric core of Optimizer::LocalBundleAdjustment (reference src/Optimizer.cc:1588-2040):
// the g2o graph optimisation `optimizer.optimize(10)` with BlockSolver_6_3 + OptimizationAlgorithmLevenberg +
// LinearSolverEigen on VertexSE3Expmap / VertexSBAPointXYZ vertices and EdgeSE3ProjectXYZ (mono) /
// EdgeStereoSE3ProjectXYZ (RGB-D "stereo") edges with Huber kernels.  Each function cites the g2o file:line
// (Thirdparty/g2o/g2o/...) it follows.  Dense per-problem algebra (the reduced system is <= ~30 poses).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "g2o_se3.hpp"
#include "gfs_oracle.h"

namespace {

using namespace gfso_se3;

struct Problem {
  const gfso_lba_problem* p;
  std::vector<Pose> poses;
  std::vector<double> points;  // 3 per point
  std::vector<int> free_index;  // pose -> index among free poses or -1
  int n_free = 0;
  std::vector<double> err;   // 3 per edge (mono uses 2)
  std::vector<double> chi2;  // per edge, last computeActiveErrors
};


// computeError: EdgeStereoSE3ProjectXYZ (types_six_dof_expmap.h:157-162 + cam_project .cpp:190-197, float invz)
//               EdgeSE3ProjectXYZ (include/OptimizableTypes.h:108-115 + Pinhole::project, src/CameraModels/Pinhole.cpp:35-41)
void compute_active_errors(Problem& S) {
  const gfso_lba_problem& p = *S.p;
  for (int e = 0; e < p.n_edges; e++) {
    double xc[3];
    map_point(S.poses[p.edge_pose[e]], &S.points[3 * p.edge_point[e]], xc);
    const double* obs = p.edge_obs + 3 * e;
    double* r = &S.err[3 * e];
    if (p.edge_stereo[e]) {
      const float invz = (float)(1.0f / xc[2]);
      const double u = xc[0] * invz * p.fx + p.cx, v = xc[1] * invz * p.fy + p.cy;
      const float bf = (float)p.bf;
      const double ur = u - bf * invz;
      r[0] = obs[0] - u;
      r[1] = obs[1] - v;
      r[2] = obs[2] - ur;
      S.chi2[e] = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * p.edge_inv_sigma2[e];
    } else {
      const double u = p.fx * xc[0] / xc[2] + p.cx, v = p.fy * xc[1] / xc[2] + p.cy;
      r[0] = obs[0] - u;
      r[1] = obs[1] - v;
      r[2] = 0;
      S.chi2[e] = (r[0] * r[0] + r[1] * r[1]) * p.edge_inv_sigma2[e];
    }
  }
}

inline void huber(double e, double delta, double* rho) {  // RobustKernelHuber::robustify, core/robust_kernel_impl.cpp:78-91
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e;
    rho[1] = 1.;
    rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

double active_robust_chi2(const Problem& S) {  // core/sparse_optimizer.cpp:100-114
  const gfso_lba_problem& p = *S.p;
  double chi = 0;
  for (int e = 0; e < p.n_edges; e++) {
    double rho[3];
    huber(S.chi2[e], p.edge_stereo[e] ? p.huber_stereo : p.huber_mono, rho);
    chi += rho[0];
  }
  return chi;
}

// linearizeOplus (types_six_dof_expmap.cpp:228-275 stereo; src/OptimizableTypes.cpp:134-154 mono):
// Ji = d err / d point (D x 3), Jj = d err / d pose (D x 6, [omega, upsilon]); row-major
void edge_jacobians(const Problem& S, int e, double* Ji, double* Jj, int* D) {
  const gfso_lba_problem& p = *S.p;
  const Pose& T = S.poses[p.edge_pose[e]];
  double xc[3], R[9];
  map_point(T, &S.points[3 * p.edge_point[e]], xc);
  quat_to_R(T.q, R);
  const double x = xc[0], y = xc[1], z = xc[2];
  const double fx = p.fx, fy = p.fy, bf = p.bf;
  if (p.edge_stereo[e]) {
    *D = 3;
    const double z_2 = z * z;
    for (int c = 0; c < 3; c++) {
      Ji[0 * 3 + c] = -fx * R[0 * 3 + c] / z + fx * x * R[2 * 3 + c] / z_2;
      Ji[1 * 3 + c] = -fy * R[1 * 3 + c] / z + fy * y * R[2 * 3 + c] / z_2;
      Ji[2 * 3 + c] = Ji[0 * 3 + c] - bf * R[2 * 3 + c] / z_2;
    }
    Jj[0] = x * y / z_2 * fx;
    Jj[1] = -(1 + (x * x / z_2)) * fx;
    Jj[2] = y / z * fx;
    Jj[3] = -1. / z * fx;
    Jj[4] = 0;
    Jj[5] = x / z_2 * fx;
    Jj[6] = (1 + y * y / z_2) * fy;
    Jj[7] = -x * y / z_2 * fy;
    Jj[8] = -x / z * fy;
    Jj[9] = 0;
    Jj[10] = -1. / z * fy;
    Jj[11] = y / z_2 * fy;
    Jj[12] = Jj[0] - bf * y / z_2;
    Jj[13] = Jj[1] + bf * x / z_2;
    Jj[14] = Jj[2];
    Jj[15] = Jj[3];
    Jj[16] = 0;
    Jj[17] = Jj[5] - bf / z_2;
  } else {
    *D = 2;
    // -Pinhole::projectJac (src/CameraModels/Pinhole.cpp:71-81)
    const double pj[6] = {-(fx / z), -0.0, -(-fx * x / (z * z)), -0.0, -(fy / z), -(-fy * y / (z * z))};
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 3; c++) Ji[r * 3 + c] = pj[r * 3] * R[c] + pj[r * 3 + 1] * R[3 + c] + pj[r * 3 + 2] * R[6 + c];
    const double sd[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
    for (int r = 0; r < 2; r++)
      for (int c = 0; c < 6; c++) Jj[r * 6 + c] = pj[r * 3] * sd[c] + pj[r * 3 + 1] * sd[6 + c] + pj[r * 3 + 2] * sd[12 + c];
    for (int c = 0; c < 3; c++) Ji[6 + c] = 0;
    for (int c = 0; c < 6; c++) Jj[12 + c] = 0;
  }
}

struct System {
  int nf, np, ne;
  std::vector<double> Hpp;  // nf x 36 (row-major 6x6)
  std::vector<double> Hll;  // np x 9
  std::vector<double> Hpl;  // ne x 18: (6 pose rows) x (3 point cols), zero for fixed poses
  std::vector<double> bp, bl;
};

// BlockSolver::buildSystem (core/block_solver.hpp:502-558) + BaseBinaryEdge::constructQuadraticForm
// (core/base_binary_edge.hpp:55-120) with the Huber-weighted information (core/base_edge.h:96-102).
void build_system(const Problem& S, System& A) {
  const gfso_lba_problem& p = *S.p;
  A.nf = S.n_free;
  A.np = p.n_points;
  A.ne = p.n_edges;
  A.Hpp.assign((size_t)A.nf * 36, 0);
  A.Hll.assign((size_t)A.np * 9, 0);
  A.Hpl.assign((size_t)A.ne * 18, 0);
  A.bp.assign((size_t)A.nf * 6, 0);
  A.bl.assign((size_t)A.np * 3, 0);
  for (int e = 0; e < p.n_edges; e++) {
    double Ji[9], Jj[18];
    int D;
    edge_jacobians(S, e, Ji, Jj, &D);
    double rho[3];
    huber(S.chi2[e], p.edge_stereo[e] ? p.huber_stereo : p.huber_mono, rho);
    const double w = rho[1] * p.edge_inv_sigma2[e];  // weightedOmega = rho[1] * information
    const double* r = &S.err[3 * e];
    double omega_r[3];
    for (int k = 0; k < 3; k++) omega_r[k] = -(p.edge_inv_sigma2[e] * r[k]) * rho[1];
    const int pt = p.edge_point[e], fi = S.free_index[p.edge_pose[e]];
    for (int a = 0; a < 3; a++) {
      for (int k = 0; k < D; k++) A.bl[3 * pt + a] += Ji[k * 3 + a] * omega_r[k];
      for (int b = 0; b < 3; b++) {
        double s = 0;
        for (int k = 0; k < D; k++) s += Ji[k * 3 + a] * w * Ji[k * 3 + b];
        A.Hll[9 * pt + 3 * a + b] += s;
      }
    }
    if (fi >= 0) {
      for (int a = 0; a < 6; a++) {
        for (int k = 0; k < D; k++) A.bp[6 * fi + a] += Jj[k * 6 + a] * omega_r[k];
        for (int b = 0; b < 6; b++) {
          double s = 0;
          for (int k = 0; k < D; k++) s += Jj[k * 6 + a] * w * Jj[k * 6 + b];
          A.Hpp[36 * fi + 6 * a + b] += s;
        }
        for (int b = 0; b < 3; b++) {
          double s = 0;
          for (int k = 0; k < D; k++) s += Jj[k * 6 + a] * w * Ji[k * 3 + b];
          A.Hpl[18 * e + 3 * a + b] += s;
        }
      }
    }
  }
}

void inv3_rowmajor(const double* a, double* r) {
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double id = 1.0 / det;
  r[0] = c00 * id;
  r[1] = (a[2] * a[7] - a[1] * a[8]) * id;
  r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
  r[3] = c01 * id;
  r[4] = (a[0] * a[8] - a[2] * a[6]) * id;
  r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
  r[6] = c02 * id;
  r[7] = (a[1] * a[6] - a[0] * a[7]) * id;
  r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
}

// dense LDL^T without pivoting on the reduced pose system (LinearSolverEigen / SimplicialLDLT,
// solvers/linear_solver_eigen.h:94-120); false iff a zero pivot appears (Eigen: NumericalIssue)
bool ldlt_solve(std::vector<double>& H, int n, const std::vector<double>& b, std::vector<double>& x) {
  std::vector<double> d(n);
  for (int j = 0; j < n; j++) {
    double dj = H[(size_t)j * n + j];
    for (int k = 0; k < j; k++) dj -= H[(size_t)j * n + k] * H[(size_t)j * n + k] * d[k];
    if (dj == 0 || !std::isfinite(dj)) return false;
    d[j] = dj;
    for (int i = j + 1; i < n; i++) {
      double v = H[(size_t)i * n + j];
      for (int k = 0; k < j; k++) v -= H[(size_t)i * n + k] * H[(size_t)j * n + k] * d[k];
      H[(size_t)i * n + j] = v / dj;
    }
  }
  x = b;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < i; k++) x[i] -= H[(size_t)i * n + k] * x[k];
  for (int i = 0; i < n; i++) x[i] /= d[i];
  for (int i = n - 1; i >= 0; i--)
    for (int k = i + 1; k < n; k++) x[i] -= H[(size_t)k * n + i] * x[k];
  return true;
}

// BlockSolver::solve with Schur complement (core/block_solver.hpp:354-487); lambda already on the diagonals
bool solve_schur(const Problem& S, const System& A, double lambda, std::vector<double>& xp, std::vector<double>& xl) {
  const gfso_lba_problem& p = *S.p;
  const int nf = A.nf, n = 6 * nf;
  std::vector<double> Hs((size_t)n * n, 0), bs(A.bp);
  for (int i = 0; i < nf; i++)
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++) Hs[(size_t)(6 * i + a) * n + 6 * i + b] = A.Hpp[36 * i + 6 * a + b] + (a == b ? lambda : 0.0);
  // edges grouped by landmark
  std::vector<std::vector<int>> by_point(p.n_points);
  for (int e = 0; e < p.n_edges; e++)
    if (S.free_index[p.edge_pose[e]] >= 0) by_point[p.edge_point[e]].push_back(e);
  std::vector<double> Dinv((size_t)p.n_points * 9);
  for (int l = 0; l < p.n_points; l++) {
    double D[9];
    for (int k = 0; k < 9; k++) D[k] = A.Hll[9 * l + k] + (k % 4 == 0 ? lambda : 0.0);
    double* Di = &Dinv[9 * l];
    inv3_rowmajor(D, Di);
    double db[3];
    for (int a = 0; a < 3; a++) db[a] = Di[3 * a] * A.bl[3 * l] + Di[3 * a + 1] * A.bl[3 * l + 1] + Di[3 * a + 2] * A.bl[3 * l + 2];
    for (int e1 : by_point[l]) {
      const int i1 = S.free_index[p.edge_pose[e1]];
      const double* Bi = &A.Hpl[18 * e1];
      double BD[18];
      for (int a = 0; a < 6; a++)
        for (int b = 0; b < 3; b++) BD[3 * a + b] = Bi[3 * a] * Di[b] + Bi[3 * a + 1] * Di[3 + b] + Bi[3 * a + 2] * Di[6 + b];
      for (int a = 0; a < 6; a++) bs[6 * i1 + a] -= Bi[3 * a] * db[0] + Bi[3 * a + 1] * db[1] + Bi[3 * a + 2] * db[2];
      for (int e2 : by_point[l]) {
        const int i2 = S.free_index[p.edge_pose[e2]];
        const double* Bj = &A.Hpl[18 * e2];
        for (int a = 0; a < 6; a++)
          for (int b = 0; b < 6; b++)
            Hs[(size_t)(6 * i1 + a) * n + 6 * i2 + b] -= BD[3 * a] * Bj[3 * b] + BD[3 * a + 1] * Bj[3 * b + 1] + BD[3 * a + 2] * Bj[3 * b + 2];
      }
    }
  }
  if (!ldlt_solve(Hs, n, bs, xp)) return false;
  xl.assign((size_t)p.n_points * 3, 0);
  std::vector<double> cl(A.bl);
  for (int e = 0; e < p.n_edges; e++) {  // cl = bl - Hpl^T xp
    const int fi = S.free_index[p.edge_pose[e]];
    if (fi < 0) continue;
    const double* B = &A.Hpl[18 * e];
    for (int b = 0; b < 3; b++)
      for (int a = 0; a < 6; a++) cl[3 * p.edge_point[e] + b] -= B[3 * a + b] * xp[6 * fi + a];
  }
  for (int l = 0; l < p.n_points; l++)
    for (int a = 0; a < 3; a++)
      xl[3 * l + a] = Dinv[9 * l + 3 * a] * cl[3 * l] + Dinv[9 * l + 3 * a + 1] * cl[3 * l + 1] + Dinv[9 * l + 3 * a + 2] * cl[3 * l + 2];
  return true;
}

void init_problem(Problem& S, const gfso_lba_problem* p) {
  S.p = p;
  S.poses.resize(p->n_poses);
  S.free_index.assign(p->n_poses, -1);
  S.n_free = 0;
  for (int i = 0; i < p->n_poses; i++) {
    std::memcpy(S.poses[i].q, p->pose_q + 4 * i, 32);
    std::memcpy(S.poses[i].t, p->pose_t + 3 * i, 24);
    normalize_rotation(S.poses[i].q);  // SE3Quat(q, t) ctor (src/Optimizer.cc:1693-1694)
    if (!p->pose_fixed[i]) S.free_index[i] = S.n_free++;
  }
  S.points.assign(p->points, p->points + 3 * (size_t)p->n_points);
  S.err.assign((size_t)p->n_edges * 3, 0);
  S.chi2.assign(p->n_edges, 0);
}

}  // namespace

extern "C" {

double gfso_lba_linearize(const gfso_lba_problem* p, double* Hpp, double* Hll, double* Hpl, double* bp, double* bl,
                          double* edge_chi2) {
  Problem S;
  init_problem(S, p);
  compute_active_errors(S);
  System A;
  build_system(S, A);
  // column-major outputs to match the C ABI of the product (gfs_lba_linearize)
  for (int i = 0; i < A.nf; i++)
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 6; b++)
        if (Hpp) Hpp[36 * i + a + 6 * b] = A.Hpp[36 * i + 6 * a + b];
  for (int l = 0; l < A.np; l++)
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        if (Hll) Hll[9 * l + a + 3 * b] = A.Hll[9 * l + 3 * a + b];
  for (int e = 0; e < A.ne; e++)
    for (int a = 0; a < 6; a++)
      for (int b = 0; b < 3; b++)
        if (Hpl) Hpl[18 * e + a + 6 * b] = A.Hpl[18 * e + 3 * a + b];
  if (bp) std::memcpy(bp, A.bp.data(), A.bp.size() * 8);
  if (bl) std::memcpy(bl, A.bl.data(), A.bl.size() * 8);
  if (edge_chi2) std::memcpy(edge_chi2, S.chi2.data(), S.chi2.size() * 8);
  return active_robust_chi2(S);
}

// SparseOptimizer::optimize (core/sparse_optimizer.cpp:354-419) + OptimizationAlgorithmLevenberg::solve
// (core/optimization_algorithm_levenberg.cpp:61-168)
int gfso_lba_solve(const gfso_lba_problem* p, gfso_lba_solution* s) {
  Problem S;
  init_problem(S, p);
  System A;
  const double tau = 1e-5, goodStepUpperScale = 2. / 3., goodStepLowerScale = 1. / 3.;
  const int maxTrialsAfterFailure = 10;
  double currentLambda = -1, ni = 2;
  int nBad = 0, iters = 0;
  double lastChi = 0;
  for (int iteration = 0; iteration < p->iterations; iteration++) {
    compute_active_errors(S);
    double currentChi = active_robust_chi2(S);
    double tempChi = currentChi;
    const double iniChi = currentChi;
    build_system(S, A);
    if (iteration == 0) {  // computeLambdaInit: tau * max |diag(H)| over all optimised vertices
      double maxDiagonal = 0;
      for (int i = 0; i < A.nf; i++)
        for (int a = 0; a < 6; a++) maxDiagonal = std::max(std::fabs(A.Hpp[36 * i + 7 * a]), maxDiagonal);
      for (int l = 0; l < A.np; l++)
        for (int a = 0; a < 3; a++) maxDiagonal = std::max(std::fabs(A.Hll[9 * l + 4 * a]), maxDiagonal);
      currentLambda = tau * maxDiagonal;
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const std::vector<Pose> poses_backup = S.poses;  // _optimizer->push()
      const std::vector<double> points_backup = S.points;
      std::vector<double> xp, xl;
      const bool ok2 = solve_schur(S, A, currentLambda, xp, xl);
      if (ok2) {  // _optimizer->update(x)
        for (int i = 0; i < p->n_poses; i++)
          if (S.free_index[i] >= 0) pose_oplus(S.poses[i], &xp[6 * S.free_index[i]]);
        for (size_t k = 0; k < S.points.size(); k++) S.points[k] += xl[k];
      }
      compute_active_errors(S);
      tempChi = active_robust_chi2(S);
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = 0;  // computeScale: sum x (lambda x + b)
      if (ok2) {
        for (size_t j = 0; j < xp.size(); j++) scale += xp[j] * (currentLambda * xp[j] + A.bp[j]);
        for (size_t j = 0; j < xl.size(); j++) scale += xl[j] * (currentLambda * xl[j] + A.bl[j]);
      }
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
        S.poses = poses_backup;  // _optimizer->pop()
        S.points = points_backup;
      }
      qmax++;
    } while (rho < 0 && qmax < maxTrialsAfterFailure);
    iters++;
    lastChi = currentChi;
    if (qmax == maxTrialsAfterFailure || rho == 0) break;  // Terminate
    if ((iniChi - currentChi) * 1e3 < iniChi)
      nBad++;
    else
      nBad = 0;
    if (nBad >= 3) break;
  }
  if (p->iterations <= 0) compute_active_errors(S);
  for (int i = 0; i < p->n_poses; i++) {
    std::memcpy(s->pose_q + 4 * i, S.poses[i].q, 32);
    std::memcpy(s->pose_t + 3 * i, S.poses[i].t, 24);
  }
  std::memcpy(s->points, S.points.data(), S.points.size() * 8);
  for (int e = 0; e < p->n_edges; e++) {
    if (s->edge_chi2) s->edge_chi2[e] = S.chi2[e];  // e->chi2(): from the last computeActiveErrors
    if (s->edge_depth_positive) {                   // isDepthPositive(): from the final estimates
      double xc[3];
      map_point(S.poses[p->edge_pose[e]], &S.points[3 * p->edge_point[e]], xc);
      s->edge_depth_positive[e] = xc[2] > 0.0;
    }
  }
  s->iterations_run = iters;
  s->final_chi2 = lastChi;
  s->final_lambda = currentLambda;
  return 0;
}

}  // extern "C"
