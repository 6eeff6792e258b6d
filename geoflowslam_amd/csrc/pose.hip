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

// computeError of an edge at pose (q, t) and its chi2
// An edge as its thread keeps it: the inputs (read once) and the state g2o keeps per edge (error vector, chi2, level) plus mvbOutlier.
struct EdgeReg {
  double xw[3], obs[3], err[3], chi2, w;
  int stereo, level, outlier;
};
__device__ __forceinline__ void pose_edge_error(const PoseFrame& F, const EdgeReg& R, const double* q, const double* t, double* r) {
  double xc[3];
  map3(q, t, R.xw, xc);
  if (R.stereo) {  // cam_project (types_six_dof_expmap.cpp:339-346): float invz, double bf
    const float invz = (float)(1.0 / xc[2]);
    const double u = xc[0] * (double)invz * F.fx + F.cx, v = xc[1] * (double)invz * F.fy + F.cy;
    r[0] = R.obs[0] - u;
    r[1] = R.obs[1] - v;
    r[2] = R.obs[2] - (u - F.bf * (double)invz);
  } else {  // Pinhole::project(Vector3d), src/CameraModels/Pinhole.cpp:35-41
    r[0] = R.obs[0] - (F.fx * xc[0] / xc[2] + F.cx);
    r[1] = R.obs[1] - (F.fy * xc[1] / xc[2] + F.cy);
    r[2] = 0;
  }
}
__device__ __forceinline__ double pose_edge_chi2(const EdgeReg& R, const double* r) {
  return R.stereo ? (r[0] * R.w * r[0] + r[1] * R.w * r[1] + r[2] * R.w * r[2]) : (r[0] * R.w * r[0] + r[1] * R.w * r[1]);
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

template <typename T>
__device__ __forceinline__ T ordered_sum(const T* __restrict__ v, int cnt, T s) {
  // one dependent chain of cnt additions; the terms of the NEXT batch are fetched from LDS while this batch is added (a plain loop
  // waits for its eight reads, adds, and only then asks for the next eight: 250 cycles a batch instead of the 80 the adds take)
  constexpr int kB = 8;
  T a[kB];
  int j = 0;
  if (cnt >= kB) {
#pragma unroll
    for (int u = 0; u < kB; u++) a[u] = v[u];
    for (; j + 2 * kB <= cnt; j += kB) {
      T b[kB];
#pragma unroll
      for (int u = 0; u < kB; u++) b[u] = v[j + kB + u];
#pragma unroll
      for (int u = 0; u < kB; u++) s += a[u];
#pragma unroll
      for (int u = 0; u < kB; u++) a[u] = b[u];
    }
#pragma unroll
    for (int u = 0; u < kB; u++) s += a[u];
    j += kB;
  }
  for (; j < cnt; j++) s += v[j];
  return s;
}

// Eigen::LDLT<MatrixXd>::compute + isPositive + solve on a 6x6 (see oracle/pose_oracle.cpp for the line-by-line restatement).
// The pivot search and the symmetric transpositions index the matrix at run time.  A private array indexed at run time lives in
// scratch memory, an LDS copy costs a round trip per access on the one lane everybody waits for (6.5 us a solve, a fifth of the
// kernel): here every index is a compile-time constant -- the loops over k, i, j are unrolled and the run-time pivot p is matched
// against its (at most five) possible values, each with its own statically indexed swaps -- so the 21 entries of the lower
// triangle, y and the transpositions stay in registers.  The arithmetic, operation for operation, is the restatement's.
template <int K, int PC>
__device__ __forceinline__ void ldlt6_transpose(double (&A)[6][6]) {  // symmetric transposition k <-> p restricted to the lower triangle
#pragma unroll
  for (int j = 0; j < K; j++) {
    const double tmp = A[K][j];
    A[K][j] = A[PC][j];
    A[PC][j] = tmp;
  }
#pragma unroll
  for (int i = PC + 1; i < 6; i++) {
    const double tmp = A[i][K];
    A[i][K] = A[i][PC];
    A[i][PC] = tmp;
  }
  {
    const double tmp = A[K][K];
    A[K][K] = A[PC][PC];
    A[PC][PC] = tmp;
  }
#pragma unroll
  for (int i = K + 1; i < PC; i++) {
    const double tmp = A[i][K];
    A[i][K] = A[PC][i];
    A[PC][i] = tmp;
  }
}
template <int K>
__device__ __forceinline__ void ldlt6_step(double (&A)[6][6], int (&tr)[6], int& sign) {
  int p = K;
  double best = fabs(A[K][K]);
#pragma unroll
  for (int i = K + 1; i < 6; i++)
    if (fabs(A[i][i]) > best) {
      best = fabs(A[i][i]);
      p = i;
    }
  tr[K] = p;
  if constexpr (K + 1 < 6) { if (p == K + 1) ldlt6_transpose<K, K + 1 < 6 ? K + 1 : 5>(A); }
  if constexpr (K + 2 < 6) { if (p == K + 2) ldlt6_transpose<K, K + 2 < 6 ? K + 2 : 5>(A); }
  if constexpr (K + 3 < 6) { if (p == K + 3) ldlt6_transpose<K, K + 3 < 6 ? K + 3 : 5>(A); }
  if constexpr (K + 4 < 6) { if (p == K + 4) ldlt6_transpose<K, K + 4 < 6 ? K + 4 : 5>(A); }
  if constexpr (K + 5 < 6) { if (p == K + 5) ldlt6_transpose<K, K + 5 < 6 ? K + 5 : 5>(A); }
  if constexpr (K > 0) {
    double temp[K];
#pragma unroll
    for (int j = 0; j < K; j++) temp[j] = A[j][j] * A[K][j];
    double acc = 0;
#pragma unroll
    for (int j = 0; j < K; j++) acc += A[K][j] * temp[j];
    A[K][K] -= acc;
#pragma unroll
    for (int i = K + 1; i < 6; i++) {
      double a2 = 0;
#pragma unroll
      for (int j = 0; j < K; j++) a2 += A[i][j] * temp[j];
      A[i][K] -= a2;
    }
  }
  const double akk = A[K][K];
  if (fabs(akk) > 0) {
#pragma unroll
    for (int i = K + 1; i < 6; i++) A[i][K] /= akk;
  }
  if (sign == 1) {
    if (akk < 0) sign = 2;
  } else if (sign == -1) {
    if (akk > 0) sign = 2;
  } else if (sign == 0) {
    if (akk > 0) sign = 1;
    else if (akk < 0) sign = -1;
  }
}
template <int K>
__device__ __forceinline__ void ldlt6_swap_y(double (&y)[6], int p) {  // y[K] <-> y[p], p >= K
  if constexpr (K + 1 < 6) { if (p == K + 1) { const double t = y[K]; y[K] = y[K + 1 < 6 ? K + 1 : 5]; y[K + 1 < 6 ? K + 1 : 5] = t; } }
  if constexpr (K + 2 < 6) { if (p == K + 2) { const double t = y[K]; y[K] = y[K + 2 < 6 ? K + 2 : 5]; y[K + 2 < 6 ? K + 2 : 5] = t; } }
  if constexpr (K + 3 < 6) { if (p == K + 3) { const double t = y[K]; y[K] = y[K + 3 < 6 ? K + 3 : 5]; y[K + 3 < 6 ? K + 3 : 5] = t; } }
  if constexpr (K + 4 < 6) { if (p == K + 4) { const double t = y[K]; y[K] = y[K + 4 < 6 ? K + 4 : 5]; y[K + 4 < 6 ? K + 4 : 5] = t; } }
  if constexpr (K + 5 < 6) { if (p == K + 5) { const double t = y[K]; y[K] = y[K + 5 < 6 ? K + 5 : 5]; y[K + 5 < 6 ? K + 5 : 5] = t; } }
}
// H: the 21 entries of the lower triangle, packed a (a + 1) / 2 + c; lambda is added to the diagonal
__device__ __forceinline__ bool ldlt6_solve_positive(const double* H21, double lambda, const double* b, double* x) {
  double A[6][6];
  {
    int o = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c <= a; c++) {
        A[a][c] = H21[o];
        A[c][a] = H21[o];
        o++;
      }
  }
#pragma unroll
  for (int a = 0; a < 6; a++) A[a][a] += lambda;
  int tr[6], sign = 0;
  ldlt6_step<0>(A, tr, sign);
  ldlt6_step<1>(A, tr, sign);
  ldlt6_step<2>(A, tr, sign);
  ldlt6_step<3>(A, tr, sign);
  ldlt6_step<4>(A, tr, sign);
  ldlt6_step<5>(A, tr, sign);
  if (sign != 1) return false;
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = b[i];
  ldlt6_swap_y<0>(y, tr[0]);
  ldlt6_swap_y<1>(y, tr[1]);
  ldlt6_swap_y<2>(y, tr[2]);
  ldlt6_swap_y<3>(y, tr[3]);
  ldlt6_swap_y<4>(y, tr[4]);
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] = fabs(A[i][i]) > 2.2250738585072014e-308 ? y[i] / A[i][i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; i--)
#pragma unroll
    for (int j = i + 1; j < 6; j++) y[i] -= A[j][i] * y[j];
  ldlt6_swap_y<4>(y, tr[4]);
  ldlt6_swap_y<3>(y, tr[3]);
  ldlt6_swap_y<2>(y, tr[2]);
  ldlt6_swap_y<1>(y, tr[1]);
  ldlt6_swap_y<0>(y, tr[0]);
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = y[i];
  return true;
}

// kTree = false: every sum over the edges in edge order on one lane (the bits of the sequential restatement; a chain of n dependent
//   additions, 21 cycles each, three times an LM iteration).
// kTree = true (the default of the handle): the same terms added by a tree of FIXED shape -- a thread adds its own edges (e = tid,
//   tid + 256, ...) in index order, then the 256 partial sums are folded by the wave shuffle tree and the four waves in order.  The
//   shape depends on nothing but the number of edges, so a frame gives the same bits alone or inside any batch; against the
//   restatement the sums differ in their last bits (relative 1e-16), the bar on the pose is 1e-5, and an outlier flag can only
//   differ where an edge's chi2 sits within rounding of its threshold (tests/test_gpu_pose.py proves that for every flip).

// The 6x6 solve of the tree-sum default.  H + lambda I of an LM trial is symmetric positive definite unless the trial is hopeless, and
// for such a matrix Eigen's diagonal pivoting only re-orders the rounding: an UN-pivoted LDL^T (as k_gicp_solve uses for the same
// reason) gives the solution to rounding with ~150 instead of ~1 000 instructions on the one lane everybody waits for (no pivot
// search, no transpositions, six reciprocals instead of 21 divisions).  "Not positive" (LinearSolverDense: the trial is rejected) =
// a pivot that is not > 0 -- the same matrices, up to those within rounding of singular.
__device__ __forceinline__ bool ldlt6_solve_spd_fast(const double* H21, double lambda, const double* b, double* x) {
#pragma clang fp contract(fast)
  double L[6][6], W[6][6], rd[6];  // L unit lower, W = L D, rd = 1 / d
  {
    int o = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c <= a; c++) L[a][c] = H21[o++];
  }
#pragma unroll
  for (int a = 0; a < 6; a++) L[a][a] += lambda;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = L[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) dj -= L[j][k] * W[j][k];
    ok = ok && dj > 0.0;
    rd[j] = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double v = L[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) v -= L[i][k] * W[j][k];
      W[i][j] = v;
      L[i][j] = v * rd[j];
    }
  }
  if (!ok) return false;
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    y[i] = b[i];
#pragma unroll
    for (int j = 0; j < i; j++) y[i] -= L[i][j] * y[j];
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= rd[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
#pragma unroll
    for (int j = i + 1; j < 6; j++) y[i] -= L[j][i] * y[j];
    x[i] = y[i];
  }
  return true;
}

// VertexSE3Expmap::oplusImpl as pose_oplus (g2o_se3_dev.hpp), for the tree-sum default: the device library's sin / cos and a plain
// cube instead of the bit-for-bit restatement of glibc's (a few hundred instructions on the one lane everybody waits for), fused
// multiply-adds.  Same formulas, results within rounding.
__device__ __forceinline__ void pose_oplus_fast(const double* q_in, const double* t_in, const double* u, double* q_out, double* t_out) {
#pragma clang fp contract(fast)
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
    double sn, cn;
    sincos(theta, &sn, &cn);
    const double a = sn / theta, b = (1 - cn) / (theta * theta), c = (theta - sn) / (theta * theta * theta);
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

#ifdef GFS_POSE_TIMING
#define PT_INIT long long pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt_last = clock64();
#define PT(k) { const long long _n = clock64(); pt_acc[k] += _n - pt_last; pt_last = _n; }
#define PT_END if (tid == 0 && f == 0) printf("POSET setup=%lld active=%lld build=%lld solve=%lld trial=%lld decide=%lld classify=%lld its=%d\n", pt_acc[0], pt_acc[1], pt_acc[2], pt_acc[3], pt_acc[4], pt_acc[5], pt_acc[6], O.iterations_run);
#else
#define PT_INIT
#define PT(k)
#define PT_END
#endif
template <bool kTree>
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
  __shared__ int s_flag[3];
  const int f = blockIdx.x, tid = threadIdx.x;
  PT_INIT
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
  // Edge e belongs to thread e % 256 in every pass of the kernel.  The thread's first two edges (frames of up to 512 observations:
  // the usual case) live in registers for the whole solve -- inputs, error vector, chi2, level, outlier flag; without that every one
  // of the ~50 passes over the edges starts with a round trip to memory for 56 bytes an edge and ends with another for the state.
  // Edges beyond them go through global memory (load, pass body, store).
  auto load_edge = [&](int e, bool with_state) {
    EdgeReg R;
    for (int k = 0; k < 3; k++) {
      R.xw[k] = E.xw[3 * e + k];
      R.obs[k] = E.obs[3 * e + k];
      R.err[k] = with_state ? err[3 * e + k] : 0.0;
    }
    R.w = (double)E.w[e];
    R.stereo = E.stereo[e];
    R.chi2 = with_state ? chi2[e] : 0.0;
    R.level = with_state ? level[e] : 0;
    R.outlier = with_state ? outlier[e] : 0;
    return R;
  };
  auto store_edge = [&](int e, const EdgeReg& R) {
    for (int k = 0; k < 3; k++) err[3 * e + k] = R.err[k];
    chi2[e] = R.chi2;
    level[e] = (uint8_t)R.level;
    outlier[e] = (uint8_t)R.outlier;
  };
  EdgeReg R0 = load_edge(min(tid, n - 1 < 0 ? 0 : n - 1), false), R1 = load_edge(min(tid + kPoseThreads, n - 1 < 0 ? 0 : n - 1), false);
  auto with_edge = [&](int e, auto&& body) {  // e = tid + 256 k: k is the same in every thread of a pass
    const int k = (e - tid) / kPoseThreads;
    if (k == 0) {
      body(R0);
    } else if (k == 1) {
      body(R1);
    } else {
      EdgeReg R = load_edge(e, true);
      body(R);
      store_edge(e, R);
    }
  };
  for (int e = tid + 2 * kPoseThreads; e < n; e += kPoseThreads) {
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
    if (tid < n) {
      outlier[tid] = 0;
      chi2[tid] = 0;
    }
    if (tid == 0) outs[f] = O;
    return;
  }
  // errors + chi2 of the active edges at the current estimate; returns activeRobustChi2, summed in edge order (thread 0 only)
  auto compute_active = [&](bool robust) {
    double T[7];
    for (int k = 0; k < 7; k++) T[k] = s_T[k];
    double chi = 0;
    if constexpr (kTree) {
      double mine = 0;
      for (int e = tid; e < n; e += kPoseThreads) {
        with_edge(e, [&](EdgeReg& R) {
          if (!R.level) {
            double r[3];
            pose_edge_error(F, R, T, T + 4, r);
            const double c = pose_edge_chi2(R, r);
            R.err[0] = r[0];
            R.err[1] = r[1];
            R.err[2] = r[2];
            R.chi2 = c;
            double term = c;
            if (robust) {
              double r1;
              huber(c, R.stereo ? dStereo : dMono, &term, &r1);
            }
            mine += term;
          }
        });
      }
      return block_sum256(mine, s4);
    }
    for (int base = 0; base < n; base += kSlabDoubles) {
      const int cnt = min(kSlabDoubles, n - base);
      for (int e = base + tid; e < base + cnt; e += kPoseThreads) {
        double term = 0;  // an edge that is not active adds nothing (x + 0 = x)
        with_edge(e, [&](EdgeReg& R) {
          if (!R.level) {
            double r[3];
            pose_edge_error(F, R, T, T + 4, r);
            const double c = pose_edge_chi2(R, r);
            R.err[0] = r[0];
            R.err[1] = r[1];
            R.err[2] = r[2];
            R.chi2 = c;
            term = c;
            if (robust) {
              double r1;
              huber(c, R.stereo ? dStereo : dMono, &term, &r1);
            }
          }
        });
        s_slab[e - base] = term;
      }
      __syncthreads();
      if (tid == 0) chi = ordered_sum(s_slab, cnt, chi);
      __syncthreads();
    }
    return chi;
  };
  int nBad = 0, nGood = 0;
  PT(0)
  for (int it = 0; it < F.n_rounds; it++) {
    const bool robust = it <= 2;  // setRobustKernel(0) at the end of round 2
    if (tid < 4) s_T[tid] = q0[tid];  // setEstimate(pFrame->GetPose()): the frame pose never changes
    if (tid < 3) s_T[4 + tid] = F.t[tid];
    int local_active = 0;
    for (int e = tid; e < n; e += kPoseThreads) with_edge(e, [&](EdgeReg& R) { local_active += R.level == 0; });
    __syncthreads();
    const int n_active = (int)block_sum256((double)local_active, s4);
    double currentLambda = -1, ni = 2;  // thread 0 only
    int nBadLm = 0;
    // g2o computes the active errors at the top of every iteration; after an ACCEPTED trial they are what that trial has just left
    // in the edges, at the same estimate -- same values, same sum -- so the pass is run only after a rejected one (pop(): estimate
    // restored, the edges keep the trial's errors) and at the start of a round
    bool fresh = false;       // uniform
    double currentChi = 0;    // thread 0
    for (int iteration = 0; iteration < F.its && n_active > 0; iteration++) {
      PT(5)
      if (!fresh) currentChi = compute_active(robust);
      const double iniChi = currentChi;
      PT(1)
      // ---- buildSystem: linearizeOplus + constructQuadraticForm, summed over the threads' edges
      {
        double T[7];
        for (int k = 0; k < 7; k++) T[k] = s_T[k];
        double run = 0;  // threads 0 .. 26: quantity tid, added up edge by edge
        double tsum[kSys];  // kTree: the thread's own sums over its edges
#pragma unroll
        for (int k = 0; k < kSys; k++) tsum[k] = 0;
        for (int base = 0; base < n; base += kChunk) {
          const int e = base + tid;
          double acc[kSys];
#pragma unroll
          for (int k = 0; k < kSys; k++) acc[k] = 0;
          if (e < n) with_edge(e, [&](EdgeReg& R) {
            if (R.level) return;
            double xc[3];
            map3(T, T + 4, R.xw, xc);
            const double x = xc[0], y = xc[1], z = xc[2];
            double J[18];
            int rows;
            if (R.stereo) {  // types_six_dof_expmap.cpp:375-404
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
            const double w = R.w;
            double rho1 = 1.0;
            if (robust) {
              double r0;
              huber(R.chi2, R.stereo ? dStereo : dMono, &r0, &rho1);
            }
            const double r[3] = {R.err[0], R.err[1], R.err[2]};
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
          });
          if constexpr (kTree) {
#pragma unroll
            for (int k = 0; k < kSys; k++) tsum[k] += acc[k];
          } else {
#pragma unroll
            for (int k = 0; k < kSys; k++) s_slab[k * kSlabStride + tid] = acc[k];
            __syncthreads();
            if (tid < kSys) run = ordered_sum(s_slab + tid * kSlabStride, min(kChunk, n - base), run);
            __syncthreads();
          }
        }
        if constexpr (kTree) run = gfs_red::block_sum_many<kSys, kPoseThreads / 64>(tsum, s_slab);  // valid in threads 0 .. 26
        if (tid < kSys) s_sys[tid] = run;
        __syncthreads();
      }
      PT(2)
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
          double x[6], H21[21], b6[6];
#pragma unroll
          for (int k = 0; k < 21; k++) H21[k] = s_sys[k];
#pragma unroll
          for (int k = 0; k < 6; k++) b6[k] = s_sys[21 + k];
          const bool ok2 = kTree ? ldlt6_solve_spd_fast(H21, currentLambda, b6, x) : ldlt6_solve_positive(H21, currentLambda, b6, x);
          if (ok2) {
            double qn[4], tn[3];
            if constexpr (kTree) pose_oplus_fast(s_T, s_T + 4, x, qn, tn);
            else pose_oplus(s_T, s_T + 4, x, qn, tn);
            for (int k = 0; k < 4; k++) s_T[k] = qn[k];
            for (int k = 0; k < 3; k++) s_T[4 + k] = tn[k];
          }
          for (int a = 0; a < 6; a++) s_x[a] = ok2 ? x[a] : 0.0;
          s_flag[0] = ok2 ? 1 : 0;
        }
        __syncthreads();
        PT(3)
        double tempChi = compute_active(robust);
        PT(4)
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
            const double g3 = 2 * rho - 1;
            double alpha = 1. - (kTree ? g3 * g3 * g3 : gfs_glibc::pow3(g3));
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
          s_flag[2] = (rho > 0 && isfinite(tempChi)) ? 1 : 0;  // the trial was accepted
        }
        __syncthreads();
        again = s_flag[1] != 0;
        fresh = s_flag[2] != 0;
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
    PT(5)
    // ---- classification (:972-1060).  Outlier edges of the previous round are re-evaluated at the final estimate
    //      (parallel); the float accumulation runs on thread 0 in the reference's order (mono list, then stereo list).
    {
      double T[7];
      for (int k = 0; k < 7; k++) T[k] = s_T[k];
      for (int e = tid; e < n; e += kPoseThreads)
        with_edge(e, [&](EdgeReg& R) {
          if (!R.outlier) return;
          double r[3];
          pose_edge_error(F, R, T, T + 4, r);
          R.err[0] = r[0];
          R.err[1] = r[1];
          R.err[2] = r[2];
          R.chi2 = pose_edge_chi2(R, r);
        });
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
      double mine_avg = 0.0;  // kTree: the thread's inlier terms (floats, added exactly in double), mono list first, then the stereo list
      for (int pass = 0; pass < 2; pass++)
        for (int base = 0; base < n; base += kTerms) {
          const int cnt = min(kTerms, n - base);
          for (int e = base + tid; e < base + cnt; e += kPoseThreads) {
            float term = 0.0f;
            with_edge(e, [&](EdgeReg& R) {
              if ((R.stereo != 0) == (pass == 1)) {
                const float c = (float)R.chi2;
                const bool out = c > (pass ? 7.815f : 5.991f);
                R.outlier = out ? 1 : 0;
                R.level = out ? 1 : 0;
                bad_local += out ? 1 : 0;
                good_local += out ? 0 : 1;
                if (!out) term = c;
              }
            });
            if constexpr (kTree) mine_avg += (double)term;
            else s_term[e - base] = term;
          }
          if constexpr (!kTree) {
            __syncthreads();
            if (tid == 0) avg = ordered_sum(s_term, cnt, avg);
            __syncthreads();
          }
        }
      if constexpr (kTree) {  // (float terms, exactly representable in double: every partial sum in double, rounded once)
        avg = (float)block_sum256(mine_avg, s4);
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
    PT(6)
    if (n < 10) break;  // optimizer.edges().size() < 10 (:1073)
  }
  if (tid < n) store_edge(tid, R0);  // what the host reads back: mvbOutlier and the per-edge chi2
  if (tid + kPoseThreads < n) store_edge(tid + kPoseThreads, R1);
  if (tid == 0) {
    for (int k = 0; k < 4; k++) O.q[k] = s_T[k];
    for (int k = 0; k < 3; k++) O.t[k] = s_T[4 + k];
    O.n_inliers = n - nBad;
    outs[f] = O;
  }
  PT_END
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
  // One pinned arena that mirrors one device block for the inputs (frames | xw | obs | w | stereo) and one for the outputs
  // (out | chi2 | outlier): a call is ONE copy in, the kernel, ONE copy out (five + three copies before: ~10 us of host time and a
  // copy-engine round trip each, more than the kernel's share of a single frame).
  gfs::DevBuf<uint8_t> d_in, d_res;
  gfs::PinBuf<uint8_t> h_in, h_res;
  // layout of a call with B frames, arrays strided by `stride` (the call's largest observation count, rounded up) per frame
  struct Layout {
    size_t o_xw, o_obs, o_w, o_st, in_bytes, r_chi, r_outl, res_bytes;
  };
  Layout layout(size_t B, size_t stride) const {
    auto up = [](size_t v) { return gfs::align_up(v, 256); };
    const size_t E = stride * B;
    Layout L;
    L.o_xw = up(B * sizeof(PoseFrame));
    L.o_obs = L.o_xw + up(E * 24);
    L.o_w = L.o_obs + up(E * 24);
    L.o_st = L.o_w + up(E * 4);
    L.in_bytes = L.o_st + up(E);
    L.r_chi = up(B * sizeof(PoseOut));
    L.r_outl = L.r_chi + up(E * 8);
    L.res_bytes = L.r_outl + up(E);
    return L;
  }
  gfs::DevBuf<double> d_err;
  gfs::DevBuf<uint8_t> d_level;
  int sum_order = GFS_POSE_SUMS_EDGE_ORDER;  // the ABI default reproduces g2o's integer outputs; the tree is opt-in (gfs_abi.h)
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
  const size_t Smax = gfs::align_up((size_t)max_obs, 64), E = Smax * max_batch, B = max_batch;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  const gfs_pose::Layout L = h->layout(B, Smax);
  A(h->d_in.alloc(L.in_bytes));
  A(h->h_in.alloc(L.in_bytes));
  A(h->d_res.alloc(L.res_bytes));
  A(h->h_res.alloc(L.res_bytes));
  A(h->d_err.alloc(E * 3));
  A(h->d_level.alloc(E));
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

int gfs_pose_set_sum_order(gfs_pose* h, int order) {
  GFS_REQUIRE(h && (order == GFS_POSE_SUMS_TREE || order == GFS_POSE_SUMS_EDGE_ORDER), GFS_ERR_INVALID_ARG,
              "gfs_pose_set_sum_order: invalid argument");
  std::lock_guard<std::mutex> lk(h->mu);
  h->sum_order = order;
  return GFS_OK;
}

int gfs_pose_optimize(gfs_pose* h, const gfs_pose_problem* problems, int B, gfs_pose_solution* solutions) {
  GFS_REQUIRE(h && problems && solutions && B > 0, GFS_ERR_INVALID_ARG, "gfs_pose_optimize: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_pose_optimize: batch %d exceeds capacity %d", B, h->max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  int S = 64;  // stride of the per-frame arrays in this call
  for (int f = 0; f < B; f++) {
    GFS_REQUIRE(problems[f].n_obs >= 0 && problems[f].n_obs <= h->max_obs, GFS_ERR_CAPACITY,
                "gfs_pose_optimize: frame %d has %d observations (capacity %d)", f, problems[f].n_obs, h->max_obs);
    S = std::max(S, (int)gfs::align_up((size_t)problems[f].n_obs, 64));
  }
  S = std::min(S, (int)gfs::align_up((size_t)h->max_obs, 64));
  const gfs_pose::Layout L = h->layout((size_t)B, (size_t)S);
  PoseFrame* h_frames = reinterpret_cast<PoseFrame*>(h->h_in.p);
  double* h_xw = reinterpret_cast<double*>(h->h_in.p + L.o_xw);
  double* h_obs = reinterpret_cast<double*>(h->h_in.p + L.o_obs);
  float* h_w = reinterpret_cast<float*>(h->h_in.p + L.o_w);
  uint8_t* h_stereo = h->h_in.p + L.o_st;
  const PoseOut* h_out = reinterpret_cast<const PoseOut*>(h->h_res.p);
  const double* h_chi2 = reinterpret_cast<const double*>(h->h_res.p + L.r_chi);
  const uint8_t* h_outlier = h->h_res.p + L.r_outl;
  for (int f = 0; f < B; f++) {
    const gfs_pose_problem& p = problems[f];
    GFS_REQUIRE(p.n_obs == 0 || (p.xw && p.obs && p.inv_sigma2 && p.stereo), GFS_ERR_INVALID_ARG,
                "gfs_pose_optimize: frame %d has NULL observation arrays", f);
    GFS_REQUIRE(p.n_obs == 0 || (solutions[f].outlier && solutions[f].chi2), GFS_ERR_INVALID_ARG,
                "gfs_pose_optimize: frame %d has NULL output arrays", f);
    PoseFrame& F = h_frames[f];
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
      memcpy(h_xw + (size_t)f * S * 3, p.xw, (size_t)p.n_obs * 24);
      memcpy(h_obs + (size_t)f * S * 3, p.obs, (size_t)p.n_obs * 24);
      memcpy(h_w + (size_t)f * S, p.inv_sigma2, (size_t)p.n_obs * 4);
      memcpy(h_stereo + (size_t)f * S, p.stereo, (size_t)p.n_obs);
    }
  }
  hipStream_t s = h->stream;
  GFS_HIP(hipMemcpyAsync(h->d_in.p, h->h_in.p, L.in_bytes, hipMemcpyHostToDevice, s));
  auto launch = [&](auto kernel) -> int {
    GFS_LAUNCH("k_pose_opt", kernel, dim3(B), dim3(kPoseThreads), 0, s, reinterpret_cast<const PoseFrame*>(h->d_in.p),
               reinterpret_cast<const double*>(h->d_in.p + L.o_xw), reinterpret_cast<const double*>(h->d_in.p + L.o_obs),
               reinterpret_cast<const float*>(h->d_in.p + L.o_w), (const uint8_t*)(h->d_in.p + L.o_st), S, h->d_res.p + L.r_outl,
               reinterpret_cast<double*>(h->d_res.p + L.r_chi), h->d_err.p, h->d_level.p, reinterpret_cast<PoseOut*>(h->d_res.p));
    return GFS_OK;
  };
  const int rc_launch = h->sum_order == GFS_POSE_SUMS_EDGE_ORDER ? launch(k_pose_opt<false>) : launch(k_pose_opt<true>);
  if (rc_launch) return rc_launch;
  GFS_HIP(hipMemcpyAsync(h->h_res.p, h->d_res.p, L.res_bytes, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < B; f++) {
    const PoseOut& O = h_out[f];
    gfs_pose_solution& r = solutions[f];
    const int n = problems[f].n_obs;
    if (n > 0) {
      memcpy(r.outlier, h_outlier + (size_t)f * S, n);
      memcpy(r.chi2, h_chi2 + (size_t)f * S, (size_t)n * 8);
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
