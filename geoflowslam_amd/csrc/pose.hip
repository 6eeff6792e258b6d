// Optimizer::PoseOptimization (reference src/Optimizer.cc:763-1098) on MI355X: motion-only bundle adjustment of a
// batch of frames, conventional-SLAM branch (pFrame->mpCamera2 == nullptr).
//
// One 256-thread workgroup owns one frame and runs its whole schedule on the device: 4 rounds x optimize(10) of the
// g2o Levenberg-Marquardt loop (core/optimization_algorithm_levenberg.cpp:61-168) over one VertexSE3Expmap with unary
// edges (EdgeSE3ProjectXYZOnlyPose: src/OptimizableTypes.cpp:49-63; g2o::EdgeStereoSE3ProjectXYZOnlyPose:
// Thirdparty/g2o/g2o/types/types_six_dof_expmap.cpp:339-404), Huber kernels, a dense 6x6 system solved with Eigen::LDLT
// semantics (solvers/linear_solver_dense.h:64-112), and the outlier re-classification between rounds (:972-1073).
// Edges are spread over the threads, but every sum over the edges -- activeRobustChi2 (core/sparse_optimizer.cpp:104-122) and
// the 21 + 6 entries of the normal equations (core/base_unary_edge.hpp:43-72, one edge after the other into the vertex's
// block) -- is added in g2o's order, the edge order: the threads park their terms in an LDS slab and one lane per quantity
// adds them up sequentially.  Together with glibc's sin / cos / pow (glibc_math.hpp) every double of the solve has the bits
// the sequential CPU restatement (oracle/pose_oracle.cpp) produces; the sign of a gain ratio at a converged state (one more LM
// iteration or not) depends on exactly that.  The 2- / 3-term products of an edge are associated the way Eigen evaluates
// base_unary_edge.hpp:62-63 -- (A' weightedOmega) A and ((rho1 A') Omega) e, left to right -- but no Eigen / g2o build exists in
// this image to pin that against: "bit-identical" is a statement about the restatement, the bar against the reference is 1e-5.  The scalar LM bookkeeping runs on thread 0 and is broadcast through LDS.
// Quirks kept: every round restarts from the frame's pose, chi2 values are compared as floats, nGood is never reset,
// the stereo projection uses a float 1/z, and the optimised pose is returned but meant to be discarded.
#include <memory>
#include <mutex>

#include "g2o_se3_dev.hpp"
#include "gfs_common.hpp"
#include "wave_reduce.hpp"

using namespace gfs_se3;

namespace {

constexpr int kPoseThreads = 256;
constexpr int kSys = 27;  // 21 (upper triangle of H) + 6 (b)

struct PoseFrame {
  double q[4], t[3];
  double fx, fy, cx, cy, bf;
  int n_obs, n_rounds, its, pad;
};
struct PoseOut {
  double q[4], t[3];
  float avg;
  int n_inliers, rounds_run, iterations_run;
};

struct EdgeView {
  const double* xw;
  const double* obs;
  const float* w;
  const uint8_t* stereo;
};

__device__ __forceinline__ void map3(const double* q, const double* t, const double* X, double* o) {  // SE3Quat::map
  quat_rotate(q, X, o);
  o[0] += t[0];
  o[1] += t[1];
  o[2] += t[2];
}

// computeError of edge e at pose (q, t)
__device__ __forceinline__ void pose_edge_error(const PoseFrame& F, const EdgeView& E, int e, const double* q, const double* t,
                                                double* r) {
  double xc[3];
  map3(q, t, E.xw + 3 * e, xc);
  const double* obs = E.obs + 3 * e;
  if (E.stereo[e]) {  // cam_project (types_six_dof_expmap.cpp:339-346): float invz, double bf
    const float invz = (float)(1.0 / xc[2]);
    const double u = xc[0] * (double)invz * F.fx + F.cx, v = xc[1] * (double)invz * F.fy + F.cy;
    r[0] = obs[0] - u;
    r[1] = obs[1] - v;
    r[2] = obs[2] - (u - F.bf * (double)invz);
  } else {  // Pinhole::project(Vector3d), src/CameraModels/Pinhole.cpp:35-41
    r[0] = obs[0] - (F.fx * xc[0] / xc[2] + F.cx);
    r[1] = obs[1] - (F.fy * xc[1] / xc[2] + F.cy);
    r[2] = 0;
  }
}
__device__ __forceinline__ double pose_edge_chi2(const EdgeView& E, int e, const double* r) {
  const double w = (double)E.w[e];
  return E.stereo[e] ? (r[0] * w * r[0] + r[1] * w * r[1] + r[2] * w * r[2]) : (r[0] * w * r[0] + r[1] * w * r[1]);
}

// deterministic block sum (256 threads): wave shuffle tree, then the four waves in order; result in every thread
__device__ double block_sum256(double v, double* s4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
  __syncthreads();
  if (lane == 0) s4[wave] = v;
  __syncthreads();
  return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

// Ordered sums: kChunk edges per pass (one per thread) park their 27 terms in an LDS slab, row q = quantity q with an odd
// row stride (the 27 summing lanes then read different banks); lane q adds its row to its running value in index order.
constexpr int kChunk = kPoseThreads;
constexpr int kSlabStride = kChunk + 1;
constexpr int kSlabDoubles = kSys * kSlabStride;  // also the capacity of one pass of the chi2 sum

__device__ __forceinline__ double ordered_sum(const double* __restrict__ v, int cnt, double s) {
  int j = 0;
  for (; j + 8 <= cnt; j += 8) {
    const double a0 = v[j], a1 = v[j + 1], a2 = v[j + 2], a3 = v[j + 3], a4 = v[j + 4], a5 = v[j + 5], a6 = v[j + 6], a7 = v[j + 7];
    s += a0;
    s += a1;
    s += a2;
    s += a3;
    s += a4;
    s += a5;
    s += a6;
    s += a7;
  }
  for (; j < cnt; j++) s += v[j];
  return s;
}

// Eigen::LDLT<MatrixXd>::compute + isPositive + solve on a 6x6 (see oracle/pose_oracle.cpp for the line-by-line restatement)
// The pivot search and the symmetric transpositions index the matrix at run time: private arrays would live in scratch memory (a
// round trip to the L1 per access, on the one lane everybody waits for), so the caller hands in LDS: A 6x6 (holds H on entry),
// y 6, tr 6.
__device__ __forceinline__ bool ldlt6_solve_positive(double (*A)[6], const double* b, double* x, double* y, int* tr) {
  int sign = 0;
  for (int k = 0; k < 6; k++) {
    int p = k;
    double best = fabs(A[k][k]);
    for (int i = k + 1; i < 6; i++)
      if (fabs(A[i][i]) > best) {
        best = fabs(A[i][i]);
        p = i;
      }
    tr[k] = p;
    if (p != k) {
      for (int j = 0; j < k; j++) {
        const double tmp = A[k][j];
        A[k][j] = A[p][j];
        A[p][j] = tmp;
      }
      for (int i = p + 1; i < 6; i++) {
        const double tmp = A[i][k];
        A[i][k] = A[i][p];
        A[i][p] = tmp;
      }
      {
        const double tmp = A[k][k];
        A[k][k] = A[p][p];
        A[p][p] = tmp;
      }
      for (int i = k + 1; i < p; i++) {
        const double tmp = A[i][k];
        A[i][k] = A[p][i];
        A[p][i] = tmp;
      }
    }
    if (k > 0) {
      double* temp = y;  // (y is not in use yet)
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
    if (fabs(akk) > 0)
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
  for (int i = 0; i < 6; i++) y[i] = b[i];
  for (int k = 0; k < 6; k++) {
    const double tmp = y[k];
    y[k] = y[tr[k]];
    y[tr[k]] = tmp;
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < 6; i++) y[i] = fabs(A[i][i]) > 2.2250738585072014e-308 ? y[i] / A[i][i] : 0.0;
  for (int i = 5; i >= 0; i--)
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  for (int k = 5; k >= 0; k--) {
    const double tmp = y[k];
    y[k] = y[tr[k]];
    y[tr[k]] = tmp;
  }
  for (int i = 0; i < 6; i++) x[i] = y[i];
  return true;
}

__global__ __launch_bounds__(kPoseThreads) void k_pose_opt(const PoseFrame* __restrict__ frames, const double* __restrict__ xw_all,
                                                           const double* __restrict__ obs_all, const float* __restrict__ w_all,
                                                           const uint8_t* __restrict__ stereo_all, int stride,
                                                           uint8_t* __restrict__ outlier_all, double* __restrict__ chi2_all,
                                                           double* __restrict__ err_all, uint8_t* __restrict__ level_all,
                                                           PoseOut* __restrict__ outs) {
  __shared__ double s4[4];
  __shared__ double s_slab[kSlabDoubles];
  __shared__ double s_T[7], s_Tb[7];  // current estimate, backup (push / pop)
  __shared__ double s_sys[kSys];
  __shared__ double s_x[6];
  __shared__ double s_A[6][6], s_y[6];
  __shared__ int s_tr[6];
  __shared__ int s_flag[2];
  const int f = blockIdx.x, tid = threadIdx.x;
  const PoseFrame F = frames[f];
  const int n = F.n_obs;
  EdgeView E{xw_all + (size_t)f * stride * 3, obs_all + (size_t)f * stride * 3, w_all + (size_t)f * stride,
             stereo_all + (size_t)f * stride};
  uint8_t* outlier = outlier_all + (size_t)f * stride;
  double* chi2 = chi2_all + (size_t)f * stride;
  double* err = err_all + (size_t)f * stride * 3;
  uint8_t* level = level_all + (size_t)f * stride;
  const double dMono = (double)(float)sqrt(5.991), dStereo = (double)(float)sqrt(7.815);  // deltaMono / deltaStereo are floats (:807-808)
  double q0[4] = {F.q[0], F.q[1], F.q[2], F.q[3]};
  normalize_rotation(q0);  // SE3Quat(q, t) constructor
  for (int e = tid; e < n; e += kPoseThreads) {
    outlier[e] = 0;
    level[e] = 0;
    chi2[e] = 0;
  }
  if (tid < 4) s_T[tid] = q0[tid];
  if (tid < 3) s_T[4 + tid] = F.t[tid];
  __syncthreads();
  PoseOut O;
  for (int k = 0; k < 4; k++) O.q[k] = q0[k];
  for (int k = 0; k < 3; k++) O.t[k] = F.t[k];
  O.avg = 0.f;
  O.n_inliers = 0;
  O.rounds_run = 0;
  O.iterations_run = 0;
  if (n < 3) {  // nInitialCorrespondences < 3 -> return 0 (:958)
    if (tid == 0) outs[f] = O;
    return;
  }
  // errors + chi2 of the active edges at the current estimate; returns activeRobustChi2, summed in edge order (thread 0 only)
  auto compute_active = [&](bool robust) {
    double T[7];
    for (int k = 0; k < 7; k++) T[k] = s_T[k];
    double chi = 0;
    for (int base = 0; base < n; base += kSlabDoubles) {
      const int cnt = min(kSlabDoubles, n - base);
      for (int e = base + tid; e < base + cnt; e += kPoseThreads) {
        double term = 0;  // an edge that is not active adds nothing (x + 0 = x)
        if (!level[e]) {
          double r[3];
          pose_edge_error(F, E, e, T, T + 4, r);
          const double c = pose_edge_chi2(E, e, r);
          err[3 * e] = r[0];
          err[3 * e + 1] = r[1];
          err[3 * e + 2] = r[2];
          chi2[e] = c;
          term = c;
          if (robust) {
            double r1;
            huber(c, E.stereo[e] ? dStereo : dMono, &term, &r1);
          }
        }
        s_slab[e - base] = term;
      }
      __syncthreads();
      if (tid == 0) chi = ordered_sum(s_slab, cnt, chi);
      __syncthreads();
    }
    return chi;
  };
  int nBad = 0, nGood = 0;
  for (int it = 0; it < F.n_rounds; it++) {
    const bool robust = it <= 2;  // setRobustKernel(0) at the end of round 2
    if (tid < 4) s_T[tid] = q0[tid];  // setEstimate(pFrame->GetPose()): the frame pose never changes
    if (tid < 3) s_T[4 + tid] = F.t[tid];
    int local_active = 0;
    for (int e = tid; e < n; e += kPoseThreads) local_active += level[e] == 0;
    __syncthreads();
    const int n_active = (int)block_sum256((double)local_active, s4);
    double currentLambda = -1, ni = 2;  // thread 0 only
    int nBadLm = 0;
    for (int iteration = 0; iteration < F.its && n_active > 0; iteration++) {
      double currentChi = compute_active(robust);
      const double iniChi = currentChi;
      // ---- buildSystem: linearizeOplus + constructQuadraticForm, summed over the threads' edges
      {
        double T[7];
        for (int k = 0; k < 7; k++) T[k] = s_T[k];
        double run = 0;  // threads 0 .. 26: quantity tid, added up edge by edge
        for (int base = 0; base < n; base += kChunk) {
          const int e = base + tid;
          double acc[kSys];
#pragma unroll
          for (int k = 0; k < kSys; k++) acc[k] = 0;
          if (e < n && !level[e]) {
            double xc[3];
            map3(T, T + 4, E.xw + 3 * e, xc);
            const double x = xc[0], y = xc[1], z = xc[2];
            double J[18];
            int rows;
            if (E.stereo[e]) {  // types_six_dof_expmap.cpp:375-404
              rows = 3;
              const double invz = 1.0 / z, invz_2 = invz * invz;
              J[0] = x * y * invz_2 * F.fx;
              J[1] = -(1 + (x * x * invz_2)) * F.fx;
              J[2] = y * invz * F.fx;
              J[3] = -invz * F.fx;
              J[4] = 0;
              J[5] = x * invz_2 * F.fx;
              J[6] = (1 + y * y * invz_2) * F.fy;
              J[7] = -x * y * invz_2 * F.fy;
              J[8] = -x * invz * F.fy;
              J[9] = 0;
              J[10] = -invz * F.fy;
              J[11] = y * invz_2 * F.fy;
              J[12] = J[0] - F.bf * y * invz_2;
              J[13] = J[1] + F.bf * x * invz_2;
              J[14] = J[2];
              J[15] = J[3];
              J[16] = 0;
              J[17] = J[5] - F.bf * invz_2;
            } else {  // src/OptimizableTypes.cpp:49-63: -projectJac(xyz) * SE3deriv
              rows = 2;
              const double pj[6] = {F.fx / z, 0, -F.fx * x / (z * z), 0, F.fy / z, -F.fy * y / (z * z)};
              const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
              for (int r = 0; r < 2; r++)
                for (int c = 0; c < 6; c++) J[6 * r + c] = -(pj[3 * r] * D[c] + pj[3 * r + 1] * D[6 + c] + pj[3 * r + 2] * D[12 + c]);
              for (int c = 0; c < 6; c++) J[12 + c] = 0;
            }
            const double w = (double)E.w[e];
            double rho1 = 1.0;
            if (robust) {
              double r0;
              huber(chi2[e], E.stereo[e] ? dStereo : dMono, &r0, &rho1);
            }
            const double r[3] = {err[3 * e], err[3 * e + 1], err[3 * e + 2]};
            // the lower triangle, row a / column c <= a: the entries Eigen's LDLT reads (packed a (a + 1) / 2 + c)
            // (every index below is a compile-time constant: arrays indexed at run time would live in scratch memory.  The third
            //  row is added by a select, not as a zero term, so that a mono edge sums exactly its two terms)
            const bool three = rows == 3;
            int o = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
              double sb = 0;
              sb += ((rho1 * J[a]) * w) * r[0];
              sb += ((rho1 * J[6 + a]) * w) * r[1];
              const double sb3 = sb + ((rho1 * J[12 + a]) * w) * r[2];
              sb = three ? sb3 : sb;
              acc[21 + a] = -sb;  // b -= ((rho1 A') Omega) e, Eigen's left-to-right association of base_unary_edge.hpp:62
#pragma unroll
              for (int c = 0; c <= a; c++) {
                double hh = 0;
                hh += (J[a] * (rho1 * w)) * J[c];  // (A' weightedOmega) A: H(a, c) = sum_k (J_ka w') J_kc, base_unary_edge.hpp:63
                hh += (J[6 + a] * (rho1 * w)) * J[6 + c];
                const double hh3 = hh + (J[12 + a] * (rho1 * w)) * J[12 + c];
                acc[o++] = three ? hh3 : hh;
              }
            }
          }
#pragma unroll
          for (int k = 0; k < kSys; k++) s_slab[k * kSlabStride + tid] = acc[k];
          __syncthreads();
          if (tid < kSys) run = ordered_sum(s_slab + tid * kSlabStride, min(kChunk, n - base), run);
          __syncthreads();
        }
        if (tid < kSys) s_sys[tid] = run;
        __syncthreads();
      }
      if (tid == 0 && iteration == 0) {  // computeLambdaInit: tau * max |diag(H)|
        double maxDiagonal = 0;
        for (int a = 0; a < 6; a++) maxDiagonal = fmax(fabs(s_sys[a * (a + 1) / 2 + a]), maxDiagonal);
        currentLambda = 1e-5 * maxDiagonal;
        ni = 2;
        nBadLm = 0;
      }
      double rho = 0;
      int qmax = 0;
      bool again = true;
      while (again) {
        if (tid == 0) {
          for (int k = 0; k < 7; k++) s_Tb[k] = s_T[k];  // push()
          double x[6];
          int o = 0;
          for (int a = 0; a < 6; a++)
            for (int c = 0; c <= a; c++) {
              s_A[a][c] = s_sys[o];
              s_A[c][a] = s_sys[o];
              o++;
            }
          for (int a = 0; a < 6; a++) s_A[a][a] += currentLambda;
          const bool ok2 = ldlt6_solve_positive(s_A, s_sys + 21, x, s_y, s_tr);
          if (ok2) {
            double qn[4], tn[3];
            pose_oplus(s_T, s_T + 4, x, qn, tn);
            for (int k = 0; k < 4; k++) s_T[k] = qn[k];
            for (int k = 0; k < 3; k++) s_T[4 + k] = tn[k];
          }
          for (int a = 0; a < 6; a++) s_x[a] = ok2 ? x[a] : 0.0;
          s_flag[0] = ok2 ? 1 : 0;
        }
        __syncthreads();
        double tempChi = compute_active(robust);
        if (tid == 0) {
          const bool ok2 = s_flag[0] != 0;
          if (!ok2) tempChi = 1.79769313486231570e308;
          rho = currentChi - tempChi;
          double scale = 0;
          if (ok2)
            for (int a = 0; a < 6; a++) scale += s_x[a] * (currentLambda * s_x[a] + s_sys[21 + a]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - gfs_glibc::pow3(2 * rho - 1);
            alpha = fmin(alpha, 2. / 3.);
            const double scaleFactor = fmax(1. / 3., alpha);
            currentLambda *= scaleFactor;
            ni = 2;
            currentChi = tempChi;
          } else {
            currentLambda *= ni;
            ni *= 2;
            for (int k = 0; k < 7; k++) s_T[k] = s_Tb[k];  // pop(): estimate restored, edge errors stay those of the trial
          }
          qmax++;
          s_flag[1] = (rho < 0 && qmax < 10) ? 1 : 0;
        }
        __syncthreads();
        again = s_flag[1] != 0;
        __syncthreads();
      }
      if (tid == 0) {
        O.iterations_run++;
        int stop = 0;
        if (qmax == 10 || rho == 0) stop = 1;
        if (!stop) {
          if ((iniChi - currentChi) * 1e3 < iniChi) nBadLm++;
          else nBadLm = 0;
          if (nBadLm >= 3) stop = 1;
        }
        s_flag[0] = stop;
      }
      __syncthreads();
      const int stop = s_flag[0];
      __syncthreads();
      if (stop) break;
    }
    // ---- classification (:972-1060).  Outlier edges of the previous round are re-evaluated at the final estimate
    //      (parallel); the float accumulation runs on thread 0 in the reference's order (mono list, then stereo list).
    {
      double T[7];
      for (int k = 0; k < 7; k++) T[k] = s_T[k];
      for (int e = tid; e < n; e += kPoseThreads)
        if (outlier[e]) {
          double r[3];
          pose_edge_error(F, E, e, T, T + 4, r);
          err[3 * e] = r[0];
          err[3 * e + 1] = r[1];
          err[3 * e + 2] = r[2];
          chi2[e] = pose_edge_chi2(E, e, r);
        }
    }
    __syncthreads();
    {
      // every edge is classified by its own thread; what is order dependent -- the float sum of the inliers' chi2, mono list first,
      // then the stereo list, each in creation order (:972-1060) -- is added up by one lane from terms parked in LDS (an edge that
      // is not an inlier of the list at hand parks +0, which changes nothing)
      float* s_term = reinterpret_cast<float*>(s_slab);
      constexpr int kTerms = 2 * kSlabDoubles;
      int bad_local = 0, good_local = 0;
      float avg = 0.0f;  // thread 0
      for (int pass = 0; pass < 2; pass++)
        for (int base = 0; base < n; base += kTerms) {
          const int cnt = min(kTerms, n - base);
          for (int e = base + tid; e < base + cnt; e += kPoseThreads) {
            float term = 0.0f;
            if ((E.stereo[e] != 0) == (pass == 1)) {
              const float c = (float)chi2[e];
              const bool out = c > (pass ? 7.815f : 5.991f);
              outlier[e] = out ? 1 : 0;
              level[e] = out ? 1 : 0;
              bad_local += out ? 1 : 0;
              good_local += out ? 0 : 1;
              if (!out) term = c;
            }
            s_term[e - base] = term;
          }
          __syncthreads();
          if (tid == 0) {
            int j = 0;
            for (; j + 8 <= cnt; j += 8) {
              const float a0 = s_term[j], a1 = s_term[j + 1], a2 = s_term[j + 2], a3 = s_term[j + 3], a4 = s_term[j + 4], a5 = s_term[j + 5],
                          a6 = s_term[j + 6], a7 = s_term[j + 7];
              avg += a0;
              avg += a1;
              avg += a2;
              avg += a3;
              avg += a4;
              avg += a5;
              avg += a6;
              avg += a7;
            }
            for (; j < cnt; j++) avg += s_term[j];
          }
          __syncthreads();
        }
      nBad = (int)block_sum256((double)bad_local, s4);
      nGood += (int)block_sum256((double)good_local, s4);  // nGood is never reset between the rounds
      if (tid == 0) {
        avg /= (float)nGood;
        O.avg = avg;
        O.rounds_run = it + 1;
      }
    }
    __syncthreads();
    if (n < 10) break;  // optimizer.edges().size() < 10 (:1073)
  }
  if (tid == 0) {
    for (int k = 0; k < 4; k++) O.q[k] = s_T[k];
    for (int k = 0; k < 3; k++) O.t[k] = s_T[4 + k];
    O.n_inliers = n - nBad;
    outs[f] = O;
  }
}

__global__ void k_test_glibc_math(const double* __restrict__ x, int n, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = gfs_glibc::sin(x[i]);
  out[n + i] = gfs_glibc::cos(x[i]);
  out[2 * (size_t)n + i] = gfs_glibc::pow3(x[i]);
}

}  // namespace

struct gfs_pose {
  int device, max_obs, max_batch;
  hipStream_t stream;
  std::mutex mu;
  gfs::DevBuf<PoseFrame> d_frames;
  gfs::DevBuf<PoseOut> d_out;
  gfs::DevBuf<double> d_xw, d_obs, d_chi2, d_err;
  gfs::DevBuf<float> d_w;
  gfs::DevBuf<uint8_t> d_stereo, d_outlier, d_level;
  gfs::PinBuf<PoseFrame> h_frames;
  gfs::PinBuf<PoseOut> h_out;
  gfs::PinBuf<double> h_xw, h_obs, h_chi2;
  gfs::PinBuf<float> h_w;
  gfs::PinBuf<uint8_t> h_stereo, h_outlier;
};

extern "C" {

int gfs_pose_create(int device, int max_obs, int max_batch, gfs_pose** out) {
  GFS_REQUIRE(out && max_obs > 0 && max_batch > 0, GFS_ERR_INVALID_ARG, "gfs_pose_create: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_pose> h(new gfs_pose);
  h->device = device;
  h->max_obs = max_obs;
  h->max_batch = max_batch;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t E = (size_t)max_obs * max_batch, B = max_batch;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_frames.alloc(B));
  A(h->d_out.alloc(B));
  A(h->d_xw.alloc(E * 3));
  A(h->d_obs.alloc(E * 3));
  A(h->d_chi2.alloc(E));
  A(h->d_err.alloc(E * 3));
  A(h->d_w.alloc(E));
  A(h->d_stereo.alloc(E));
  A(h->d_outlier.alloc(E));
  A(h->d_level.alloc(E));
  A(h->h_frames.alloc(B));
  A(h->h_out.alloc(B));
  A(h->h_xw.alloc(E * 3));
  A(h->h_obs.alloc(E * 3));
  A(h->h_chi2.alloc(E));
  A(h->h_w.alloc(E));
  A(h->h_stereo.alloc(E));
  A(h->h_outlier.alloc(E));
#undef A
  if (rc) {
    (void)hipStreamDestroy(h->stream);
    return rc;
  }
  *out = h.release();
  return GFS_OK;
}

void gfs_pose_destroy(gfs_pose* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_pose_optimize(gfs_pose* h, const gfs_pose_problem* problems, int B, gfs_pose_solution* solutions) {
  GFS_REQUIRE(h && problems && solutions && B > 0, GFS_ERR_INVALID_ARG, "gfs_pose_optimize: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_pose_optimize: batch %d exceeds capacity %d", B, h->max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int S = h->max_obs;
  for (int f = 0; f < B; f++) {
    const gfs_pose_problem& p = problems[f];
    GFS_REQUIRE(p.n_obs >= 0 && p.n_obs <= S, GFS_ERR_CAPACITY, "gfs_pose_optimize: frame %d has %d observations (capacity %d)", f,
                p.n_obs, S);
    GFS_REQUIRE(p.n_obs == 0 || (p.xw && p.obs && p.inv_sigma2 && p.stereo), GFS_ERR_INVALID_ARG,
                "gfs_pose_optimize: frame %d has NULL observation arrays", f);
    GFS_REQUIRE(p.n_obs == 0 || (solutions[f].outlier && solutions[f].chi2), GFS_ERR_INVALID_ARG,
                "gfs_pose_optimize: frame %d has NULL output arrays", f);
    PoseFrame& F = h->h_frames.p[f];
    for (int k = 0; k < 4; k++) F.q[k] = p.q[k];
    for (int k = 0; k < 3; k++) F.t[k] = p.t[k];
    F.fx = p.fx;
    F.fy = p.fy;
    F.cx = p.cx;
    F.cy = p.cy;
    F.bf = p.bf;
    F.n_obs = p.n_obs;
    F.n_rounds = p.n_rounds;
    F.its = p.its;
    F.pad = 0;
    if (p.n_obs > 0) {
      memcpy(h->h_xw.p + (size_t)f * S * 3, p.xw, (size_t)p.n_obs * 24);
      memcpy(h->h_obs.p + (size_t)f * S * 3, p.obs, (size_t)p.n_obs * 24);
      memcpy(h->h_w.p + (size_t)f * S, p.inv_sigma2, (size_t)p.n_obs * 4);
      memcpy(h->h_stereo.p + (size_t)f * S, p.stereo, (size_t)p.n_obs);
    }
  }
  hipStream_t s = h->stream;
  const size_t E = (size_t)S * B;
  GFS_HIP(hipMemcpyAsync(h->d_frames.p, h->h_frames.p, B * sizeof(PoseFrame), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xw.p, h->h_xw.p, E * 24, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_obs.p, h->h_obs.p, E * 24, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_w.p, h->h_w.p, E * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_stereo.p, h->h_stereo.p, E, hipMemcpyHostToDevice, s));
  GFS_LAUNCH("k_pose_opt", k_pose_opt, dim3(B), dim3(kPoseThreads), 0, s, h->d_frames.p, h->d_xw.p, h->d_obs.p, h->d_w.p,
             h->d_stereo.p, S, h->d_outlier.p, h->d_chi2.p, h->d_err.p, h->d_level.p, h->d_out.p);
  GFS_HIP(hipMemcpyAsync(h->h_out.p, h->d_out.p, B * sizeof(PoseOut), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_outlier.p, h->d_outlier.p, E, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_chi2.p, h->d_chi2.p, E * 8, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < B; f++) {
    const PoseOut& O = h->h_out.p[f];
    gfs_pose_solution& r = solutions[f];
    const int n = problems[f].n_obs;
    if (n > 0) {
      memcpy(r.outlier, h->h_outlier.p + (size_t)f * S, n);
      memcpy(r.chi2, h->h_chi2.p + (size_t)f * S, (size_t)n * 8);
    }
    for (int k = 0; k < 4; k++) r.q[k] = O.q[k];
    for (int k = 0; k < 3; k++) r.t[k] = O.t[k];
    r.avg_reproj_error = O.avg;
    r.n_inliers = O.n_inliers;
    r.rounds_run = O.rounds_run;
    r.iterations_run = O.iterations_run;
  }
  return GFS_OK;
}

// Test hook: the restated glibc functions evaluated on the device (tests/test_gpu_glibc_math.py).
int gfs_test_glibc_math(int device, const double* x, int n, double* sin_out, double* cos_out, double* pow3_out) {
  GFS_REQUIRE(x && sin_out && cos_out && pow3_out && n >= 0, GFS_ERR_INVALID_ARG, "gfs_test_glibc_math: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  if (n == 0) return GFS_OK;
  GFS_HIP(hipSetDevice(device));
  gfs::DevBuf<double> d_x, d_o;
  int rc = d_x.alloc(n);
  if (!rc) rc = d_o.alloc((size_t)3 * n);
  if (rc) return rc;
  GFS_HIP(hipMemcpy(d_x.p, x, (size_t)n * 8, hipMemcpyHostToDevice));
  GFS_LAUNCH("k_test_glibc_math", k_test_glibc_math, dim3(gfs::div_up(n, 256)), dim3(256), 0, (hipStream_t)0, d_x.p, n, d_o.p);
  GFS_HIP(hipMemcpy(sin_out, d_o.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  GFS_HIP(hipMemcpy(cos_out, d_o.p + n, (size_t)n * 8, hipMemcpyDeviceToHost));
  GFS_HIP(hipMemcpy(pow3_out, d_o.p + 2 * (size_t)n, (size_t)n * 8, hipMemcpyDeviceToHost));
  return GFS_OK;
}

}  // extern "C"
