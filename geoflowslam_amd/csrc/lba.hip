// Local bundle adjustment for gfx950 (MI355X): the numeric core of Optimizer::LocalBundleAdjustment
// (reference src/Optimizer.cc:1588-2040): g2o BlockSolver_6_3 + OptimizationAlgorithmLevenberg on
// VertexSE3Expmap / VertexSBAPointXYZ with EdgeSE3ProjectXYZ (mono) and EdgeStereoSE3ProjectXYZ (RGB-D) edges,
// Huber kernels, Schur complement on the landmarks, 10 LM iterations with up to 10 lambda trials each.
//
// MI355X design: a window (tens of free poses, a few thousand landmarks, ~50 k edges) is small and windows of one map cannot
// be sharded ("replicas only"), so the work is spread over the chip phase by phase: one launch per phase of
// a Levenberg-Marquardt trial (edge errors, landmark blocks with 16 lanes per landmark, pose blocks, Schur products with one
// workgroup per pose pair, the reduced solve, back-substitution / update, the LM decision), the LM state lives in HBM
// (LbaState) and the host only reads two mapped flags per trial.  The reduced pose system is factored by one workgroup with a
// blocked 6-wide LDL^T: in LDS while its packed triangle fits (<= 30 free poses, 130 KB), in place in HBM / L2 with the
// current panel staged in LDS for larger windows.  All sums are taken in a fixed order (deterministic, unlike atomics).
// GFS_LBA_SINGLE_WG=1 selects the first implementation instead (k_lba: the whole LM loop of a window in ONE workgroup with
// the reduced system in LDS, no host round trip; windows of <= 30 free poses only).
//
// Edge order: edges are stably re-ordered landmark-major on the host so each landmark's observations are
// contiguous (the reference builds them that way, src/Optimizer.cc:1816-1952); a second CSR lists each free
// pose's edges; edge_of[free pose][landmark] gives O(1) co-visibility lookups for the Schur products.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>

#include "g2o_se3_dev.hpp"
#include "gfs_common.hpp"
#include "wave_reduce.hpp"

namespace {

constexpr int kThreads = 1024;
constexpr int kFlagInts = 8;  // one slot of gfs_lba::h_flags: 4 ints + 2 doubles
constexpr int kMaxFreeLds = 30;  // up to this many free poses the reduced system (180 x 180 packed) is factored in LDS

// LM bookkeeping of the multi-kernel path, resident in HBM (the scalar logic of OptimizationAlgorithmLevenberg::solve)
struct LbaState {
  double lambda, ni, current_chi, ini_chi, temp_chi, rho, last_chi;
  int cur;                       // which of the two estimate buffers holds the accepted state
  int qmax, n_bad, iters;
  int solve_ok;                  // LinearSolver succeeded in the current trial
  int again, terminate;          // outputs of k_lba_decide for the host loop
  int err_at_cur;                // chi2 / err / part_chi already hold the accepted estimate's values (the accepted trial computed them)
  int phase, it;                 // batched entry: 0 = build next, 1 = trial next, 2 = done; index of the running LM iteration
  int spec_ok, pad_;             // k_lba_decide: the trial was accepted and the loop goes on -- the build group of the NEXT iteration, queued
                                 // behind it speculatively by gfs_lba_solve's host loop (gate = 1), may run (round 6)
};

// The descriptor's pointers carry the GLOBAL address space in device code: the batched kernels read their descriptor from an array
// in memory, and pointers that come out of memory are otherwise GENERIC — every access through them a FLAT instruction (64-bit
// VGPR addresses, both wait counters).  Host code (and the host functions as the device pass parses them) sees plain pointers and
// assigns through a cast to the member's type.
#if defined(__HIP_DEVICE_COMPILE__)
#define GFS_GLOBAL __attribute__((address_space(1)))
#else
#define GFS_GLOBAL
#endif
struct LbaDev {
  // problem (edges landmark-major)
  int n_poses, n_points, n_edges, n_free;
  const GFS_GLOBAL double* pose_q0;  // [n_poses][4]
  const GFS_GLOBAL double* pose_t0;  // [n_poses][3]
  const GFS_GLOBAL int* free_index;  // [n_poses] -> free slot or -1
  const GFS_GLOBAL int* free_pose;   // [n_free]  -> pose
  const GFS_GLOBAL double* points0;  // [n_points][3]
  const GFS_GLOBAL int* e_pose;
  const GFS_GLOBAL int* e_point;
  const GFS_GLOBAL double* e_obs;  // [n_edges][3]
  const GFS_GLOBAL double* e_w;    // inv_sigma2
  const GFS_GLOBAL unsigned char* e_stereo;
  const GFS_GLOBAL int* pt_begin;     // [n_points+1]
  const GFS_GLOBAL int* pose_begin;   // [n_free+1]
  const GFS_GLOBAL int* pose_edges;   // edge ids (landmark-major numbering), ascending, per free pose
  const GFS_GLOBAL int* edge_of;      // [n_free][n_points] edge id or -1 (the FIRST edge of a (pose, landmark) pair)
  const GFS_GLOBAL int* lm_wg;        // [n_lm_wg + 1] landmark ranges of the workgroups of b_build_landmarks (<= kMk edges each, or one landmark)
  int n_lm_wg;
  const GFS_GLOBAL int* e_dup;        // [n_edges] first edge of the same (pose, landmark) pair, -1 for a first edge; NULL: no duplicates
  double fx, fy, cx, cy, bf, huber_mono, huber_stereo;
  int iterations;
  // state / workspace
  GFS_GLOBAL double* q;    // [n_poses][4] current
  GFS_GLOBAL double* t;    // [n_poses][3]
  GFS_GLOBAL double* X;    // [n_points][3]
  GFS_GLOBAL double* q_try;
  GFS_GLOBAL double* t_try;
  GFS_GLOBAL double* X_try;
  GFS_GLOBAL double* chi2;   // [n_edges] last computeActiveErrors
  GFS_GLOBAL double* err;    // [n_edges][3]
  GFS_GLOBAL double* Hpl;    // [n_edges][18] row-major (6 pose rows x 3 point cols)
  GFS_GLOBAL double* Hll;    // [n_points][6] symmetric (xx,xy,xz,yy,yz,zz)
  GFS_GLOBAL double* bl;     // [n_points][3]
  GFS_GLOBAL double* Dinv;   // [n_points][6]
  GFS_GLOBAL double* Hpp;    // [n_free][21] upper triangle row-major
  GFS_GLOBAL double* bp;     // [n_free][6]
  GFS_GLOBAL double* xl;     // [n_points][3]
  GFS_GLOBAL double* xp;     // [n_free*6]
  volatile GFS_GLOBAL int* stop;  // host-mapped force-stop flag
  GFS_GLOBAL int* out_info;       // [0]=iterations_run [1]=failed flag
  GFS_GLOBAL double* out_stats;   // [0]=final chi2 [1]=final lambda
  int mode;            // 0 = full solve, 1 = linearise only
  // multi-kernel path (one launch per phase, all CUs): LM state, per-block partial sums, the reduced system in HBM
  GFS_GLOBAL struct LbaState* S;
  GFS_GLOBAL double* part_chi;    // [n_err_blocks] robust chi2 partial sums of the last k_lba_errors
  GFS_GLOBAL double* part_scale;  // [n_upd_blocks] computeScale partial sums of the last k_lba_update
  GFS_GLOBAL double* Hs;          // packed lower triangle of the Schur complement (6 n_free)^2 / 2
  GFS_GLOBAL double* bs;          // [6 n_free]
  int n_err_blocks, n_upd_blocks;
  GFS_GLOBAL double* out_pack;  // results in one block: chi2 [n_edges], q [n_poses][4], t [n_poses][3], X [n_points][3], chi2 / lambda, iterations / buffer
  // Schur products by landmark chunks: partial blocks [chunk][pose pair][36] and right-hand sides [chunk][free pose][6]
  GFS_GLOBAL double* schur_part;
  GFS_GLOBAL double* schur_part_b;
  int n_schur_chunks, n_pair_tiles, schur_sub;  // chunks of kSchurPts landmarks; tiles of kMk pose pairs; landmarks staged at a time
  int schur_mfma;  // 1: b_schur_mfma (n_schur_chunks chunks of kSchurMPts landmarks, n_pair_tiles = 128 x 128 blocks of the lower triangle)
};

using namespace gfs_se3;

// residual of one edge (computeError): types_six_dof_expmap.h:157-162 / .cpp:190-197 (float invz) and
// include/OptimizableTypes.h:108-115 + src/CameraModels/Pinhole.cpp:35-41
__device__ __forceinline__ void edge_residual(const LbaDev& D, int e, const double* q, const double* t, const double* X,
                                              double* xc, double* r) {
  const int pi = D.e_pose[e], li = D.e_point[e];
  quat_rotate(q + 4 * pi, X + 3 * li, xc);
  xc[0] += t[3 * pi];
  xc[1] += t[3 * pi + 1];
  xc[2] += t[3 * pi + 2];
  const double* obs = D.e_obs + 3 * e;
  if (D.e_stereo[e]) {
    const float invz = (float)(1.0 / xc[2]);
    const double u = xc[0] * (double)invz * D.fx + D.cx, v = xc[1] * (double)invz * D.fy + D.cy;
    const float bf = (float)D.bf;
    const double ur = u - (double)(bf * invz);
    r[0] = obs[0] - u;
    r[1] = obs[1] - v;
    r[2] = obs[2] - ur;
  } else {
    r[0] = obs[0] - (D.fx * xc[0] / xc[2] + D.cx);
    r[1] = obs[1] - (D.fy * xc[1] / xc[2] + D.cy);
    r[2] = 0;
  }
}

// linearizeOplus (types_six_dof_expmap.cpp:228-275; src/OptimizableTypes.cpp:134-154): Ji (3x3), Jj (3x6) row-major
__device__ __forceinline__ void edge_jacobians(const LbaDev& D, int e, const double* q, const double* xc, double* Ji, double* Jj) {
  double R[9];
  quat_to_R(q + 4 * D.e_pose[e], R);
  const double x = xc[0], y = xc[1], z = xc[2], fx = D.fx, fy = D.fy, bf = D.bf;
  if (D.e_stereo[e]) {
    const double z_2 = z * z;
    for (int c = 0; c < 3; c++) {
      Ji[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
      Ji[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
      Ji[6 + c] = Ji[c] - bf * R[6 + c] / z_2;
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

// deterministic block reductions (1024 threads): wave shuffle tree, then the 16 wave results in order
__device__ double block_sum(double v, double* s16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
  __syncthreads();
  if (lane == 0) s16[wave] = v;
  __syncthreads();
  double r = 0;
  for (int w = 0; w < 16; w++) r += s16[w];
  return r;
}
__device__ double block_max(double v, double* s16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) v = fmax(v, __shfl_down(v, ofs, 64));
  __syncthreads();
  if (lane == 0) s16[wave] = v;
  __syncthreads();
  double r = 0;
  for (int w = 0; w < 16; w++) r = fmax(r, s16[w]);
  return r;
}

// SparseOptimizer::computeActiveErrors + activeRobustChi2 (core/sparse_optimizer.cpp:100-114)
__device__ double compute_errors(const LbaDev& D, const double* q, const double* t, const double* X, double* s16) {
  double local = 0;
  for (int e = threadIdx.x; e < D.n_edges; e += kThreads) {
    double xc[3], r[3];
    edge_residual(D, e, q, t, X, xc, r);
    const double c = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * D.e_w[e];
    D.chi2[e] = c;
    D.err[3 * e] = r[0];
    D.err[3 * e + 1] = r[1];
    D.err[3 * e + 2] = r[2];
    double r0, r1;
    huber(c, D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
    local += r0;
  }
  return block_sum(local, s16);
}

// BlockSolver::buildSystem + BaseBinaryEdge::constructQuadraticForm (block_solver.hpp:502-558, base_binary_edge.hpp:55-120)
__device__ void build_system(const LbaDev& D, double* s16) {
  // landmark sweep: Hll, bl and the per-edge pose-landmark blocks
  for (int l = threadIdx.x; l < D.n_points; l += kThreads) {
    double H[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    for (int e = D.pt_begin[l]; e < D.pt_begin[l + 1]; e++) {
      double xc[3], r[3], Ji[9], Jj[18];
      edge_residual(D, e, D.q, D.t, D.X, xc, r);
      edge_jacobians(D, e, D.q, xc, Ji, Jj);
      double r0, r1;
      huber(D.chi2[e], D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
      const double w = r1 * D.e_w[e];
      double omr[3];
      for (int k = 0; k < 3; k++) omr[k] = -(D.e_w[e] * D.err[3 * e + k]) * r1;
      int o = 0;
      for (int a = 0; a < 3; a++) {
        b[a] += Ji[a] * omr[0] + Ji[3 + a] * omr[1] + Ji[6 + a] * omr[2];
        for (int c = a; c < 3; c++) H[o++] += Ji[a] * w * Ji[c] + Ji[3 + a] * w * Ji[3 + c] + Ji[6 + a] * w * Ji[6 + c];
      }
      if (D.free_index[D.e_pose[e]] >= 0) {
        double* B = D.Hpl + 18 * (size_t)e;
        for (int a = 0; a < 6; a++)
          for (int c = 0; c < 3; c++) B[3 * a + c] = Jj[a] * w * Ji[c] + Jj[6 + a] * w * Ji[3 + c] + Jj[12 + a] * w * Ji[6 + c];
      }
    }
    for (int k = 0; k < 6; k++) D.Hll[6 * (size_t)l + k] = H[k];
    for (int k = 0; k < 3; k++) D.bl[3 * (size_t)l + k] = b[k];
  }
  // pose sweep: one wave per free pose, lanes over its edges, fixed-order shuffle reduction
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int f = wave; f < D.n_free; f += 16) {
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; k++) acc[k] = 0;
    for (int i = D.pose_begin[f] + lane; i < D.pose_begin[f + 1]; i += 64) {
      const int e = D.pose_edges[i];
      double xc[3], r[3], Ji[9], Jj[18];
      edge_residual(D, e, D.q, D.t, D.X, xc, r);
      edge_jacobians(D, e, D.q, xc, Ji, Jj);
      double r0, r1;
      huber(D.chi2[e], D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
      const double w = r1 * D.e_w[e];
      double omr[3];
      for (int k = 0; k < 3; k++) omr[k] = -(D.e_w[e] * D.err[3 * e + k]) * r1;
      int o = 0;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = a; c < 6; c++) acc[o++] += Jj[a] * w * Jj[c] + Jj[6 + a] * w * Jj[6 + c] + Jj[12 + a] * w * Jj[12 + c];
#pragma unroll
      for (int a = 0; a < 6; a++) acc[21 + a] += Jj[a] * omr[0] + Jj[6 + a] * omr[1] + Jj[12 + a] * omr[2];
    }
#pragma unroll
    for (int k = 0; k < 27; k++) {
      double v = acc[k];
#pragma unroll
      for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
      if (lane == 0) {
        if (k < 21)
          D.Hpp[21 * f + k] = v;
        else
          D.bp[6 * f + (k - 21)] = v;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }  // packed lower, i >= j

__device__ __forceinline__ void inv3_sym(const double* h /*xx,xy,xz,yy,yz,zz*/, double lambda, double* o) {
  const double a = h[0] + lambda, b = h[1], c = h[2], d = h[3] + lambda, e = h[4], f = h[5] + lambda;
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  o[0] = c00 * id;
  o[1] = c01 * id;
  o[2] = c02 * id;
  o[3] = (a * f - c * c) * id;
  o[4] = (b * c - a * e) * id;
  o[5] = (a * d - b * b) * id;
}

// BlockSolver::solve (block_solver.hpp:354-487): Schur complement in LDS, LDL^T, landmark back-substitution.
// Returns false iff a zero / non-finite pivot appears (LinearSolverEigen reports failure).
__device__ bool solve_schur(const LbaDev& D, double lambda, double* Hs, double* bs, double* s16, int* s_flag) {
  const int n = 6 * D.n_free;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int l = threadIdx.x; l < D.n_points; l += kThreads) inv3_sym(D.Hll + 6 * (size_t)l, lambda, D.Dinv + 6 * (size_t)l);
  for (int k = threadIdx.x; k < n * (n + 1) / 2; k += kThreads) Hs[k] = 0;
  if (threadIdx.x == 0) *s_flag = 0;
  __syncthreads();
  // pose pairs (i1 >= i2), one wave per pair: Hs(i1,i2) = [i1==i2](Hpp + lambda I) - sum_l B_i1 Dinv_l B_i2^T
  const int npairs = D.n_free * (D.n_free + 1) / 2;
  for (int pr = wave; pr < npairs; pr += 16) {
    int i1 = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
    while (i1 * (i1 + 1) / 2 > pr) i1--;
    while ((i1 + 1) * (i1 + 2) / 2 <= pr) i1++;
    const int i2 = pr - i1 * (i1 + 1) / 2;
    double acc[36], accb[6];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) accb[k] = 0;
    const int* eo = D.edge_of + (size_t)i2 * D.n_points;
    for (int i = D.pose_begin[i1] + lane; i < D.pose_begin[i1 + 1]; i += 64) {
      const int e1 = D.pose_edges[i];
      const int l = D.e_point[e1];
      const int e2 = eo[l];
      if (e2 < 0) continue;
      const double* Bi = D.Hpl + 18 * (size_t)e1;
      const double* Bj = D.Hpl + 18 * (size_t)e2;
      const double* Di = D.Dinv + 6 * (size_t)l;
      const double d9[9] = {Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5]};
      double BD[18];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 3; c++) BD[3 * a + c] = Bi[3 * a] * d9[c] + Bi[3 * a + 1] * d9[3 + c] + Bi[3 * a + 2] * d9[6 + c];
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int c = 0; c < 6; c++) acc[6 * a + c] += BD[3 * a] * Bj[3 * c] + BD[3 * a + 1] * Bj[3 * c + 1] + BD[3 * a + 2] * Bj[3 * c + 2];
      if (i1 == i2) {
        const double* bl = D.bl + 3 * (size_t)l;
#pragma unroll
        for (int a = 0; a < 6; a++) accb[a] += BD[3 * a] * bl[0] + BD[3 * a + 1] * bl[1] + BD[3 * a + 2] * bl[2];
      }
    }
#pragma unroll
    for (int k = 0; k < 36; k++) {
      double v = acc[k];
#pragma unroll
      for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
      if (lane == 0) {
        const int a = k / 6, c = k % 6;
        if (i1 != i2) {
          Hs[tri(6 * i1 + a, 6 * i2 + c)] = -v;
        } else if (a >= c) {
          const int ua = c, uc = a;  // Hpp upper-triangle index of (c, a)
          const double hpp = D.Hpp[21 * i1 + (ua * 6 - ua * (ua - 1) / 2 + (uc - ua))];
          Hs[tri(6 * i1 + a, 6 * i1 + c)] = hpp + (a == c ? lambda : 0.0) - v;
        }
      }
    }
    if (i1 == i2) {
#pragma unroll
      for (int k = 0; k < 6; k++) {
        double v = accb[k];
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
        if (lane == 0) bs[6 * i1 + k] = D.bp[6 * i1 + k] - v;
      }
    }
  }
  __syncthreads();
  // LDL^T (right-looking) of the packed lower triangle; D on the diagonal, unit L below
  for (int j = 0; j < n; j++) {
    const double d = Hs[tri(j, j)];
    if (threadIdx.x == 0 && (d == 0.0 || !isfinite(d))) *s_flag = 1;
    __syncthreads();
    if (*s_flag) return false;
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) Hs[tri(i, j)] /= d;
    __syncthreads();
    const int m = n - j - 1;  // trailing size
    for (int k = threadIdx.x; k < m * (m + 1) / 2; k += kThreads) {
      int r = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
      while (r * (r + 1) / 2 > k) r--;
      while ((r + 1) * (r + 2) / 2 <= k) r++;
      const int c = k - r * (r + 1) / 2;
      const int ii = j + 1 + r, kk = j + 1 + c;
      Hs[tri(ii, kk)] -= Hs[tri(ii, j)] * Hs[tri(kk, j)] * d;
    }
    __syncthreads();
  }
  // forward, diagonal, backward substitution on bs (in place)
  for (int j = 0; j < n; j++) {
    const double xj = bs[j];
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) bs[i] -= Hs[tri(i, j)] * xj;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += kThreads) bs[i] /= Hs[tri(i, i)];
  __syncthreads();
  for (int j = n - 1; j >= 0; j--) {
    const double xj = bs[j];
    for (int i = threadIdx.x; i < j; i += kThreads) bs[i] -= Hs[tri(j, i)] * xj;
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += kThreads) D.xp[i] = bs[i];
  // landmarks: xl = Dinv (bl - Hpl^T xp)
  for (int l = threadIdx.x; l < D.n_points; l += kThreads) {
    double cl[3] = {D.bl[3 * (size_t)l], D.bl[3 * (size_t)l + 1], D.bl[3 * (size_t)l + 2]};
    for (int e = D.pt_begin[l]; e < D.pt_begin[l + 1]; e++) {
      const int f = D.free_index[D.e_pose[e]];
      if (f < 0) continue;
      const double* B = D.Hpl + 18 * (size_t)e;
      for (int c = 0; c < 3; c++)
        for (int a = 0; a < 6; a++) cl[c] -= B[3 * a + c] * bs[6 * f + a];
    }
    const double* Di = D.Dinv + 6 * (size_t)l;
    D.xl[3 * (size_t)l] = Di[0] * cl[0] + Di[1] * cl[1] + Di[2] * cl[2];
    D.xl[3 * (size_t)l + 1] = Di[1] * cl[0] + Di[3] * cl[1] + Di[4] * cl[2];
    D.xl[3 * (size_t)l + 2] = Di[2] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2];
  }
  __syncthreads();
  return true;
}

// SparseOptimizer::optimize + OptimizationAlgorithmLevenberg::solve
// (core/sparse_optimizer.cpp:354-419, core/optimization_algorithm_levenberg.cpp:61-168)
__global__ __launch_bounds__(kThreads) void k_lba(LbaDev D) {
  extern __shared__ __align__(16) double lds[];
  __shared__ double s16[16];
  __shared__ int s_flag;
  const int n = 6 * D.n_free;
  double* Hs = lds;
  double* bs = lds + (size_t)n * (n + 1) / 2;
  // load the initial estimates (SE3Quat ctor normalises the quaternion, src/Optimizer.cc:1693-1694)
  for (int i = threadIdx.x; i < D.n_poses; i += kThreads) {
    double q[4] = {D.pose_q0[4 * i], D.pose_q0[4 * i + 1], D.pose_q0[4 * i + 2], D.pose_q0[4 * i + 3]};
    normalize_rotation(q);
    for (int k = 0; k < 4; k++) D.q[4 * i + k] = D.q_try[4 * i + k] = q[k];
    for (int k = 0; k < 3; k++) D.t[3 * i + k] = D.t_try[3 * i + k] = D.pose_t0[3 * i + k];
  }
  for (int i = threadIdx.x; i < 3 * D.n_points; i += kThreads) D.X[i] = D.X_try[i] = D.points0[i];
  __syncthreads();
  const double tau = 1e-5, good_up = 2. / 3., good_lo = 1. / 3.;
  double lambda = -1, ni = 2, last_chi = 0;
  int n_bad = 0, iters = 0;
  if (D.mode == 1 || D.iterations <= 0) {
    const double chi = compute_errors(D, D.q, D.t, D.X, s16);
    if (D.mode == 1) build_system(D, s16);
    if (threadIdx.x == 0) {
      D.out_info[0] = 0;
      D.out_info[1] = 0;
      D.out_stats[0] = chi;
      D.out_stats[1] = 0;
    }
    return;
  }
  for (int iteration = 0; iteration < D.iterations; iteration++) {
    if (threadIdx.x == 0) s_flag = *D.stop;  // SparseOptimizer::terminate(): one reader, block-uniform decision
    __syncthreads();
    const int stop_now = s_flag;
    __syncthreads();
    if (stop_now) break;
    double current_chi = compute_errors(D, D.q, D.t, D.X, s16);
    const double ini_chi = current_chi;
    build_system(D, s16);
    if (iteration == 0) {  // computeLambdaInit
      double mx = 0;
      for (int i = threadIdx.x; i < D.n_free * 6; i += kThreads) {
        const int f = i / 6, a = i % 6;
        mx = fmax(mx, fabs(D.Hpp[21 * f + (a * 6 - a * (a - 1) / 2)]));
      }
      for (int i = threadIdx.x; i < D.n_points * 3; i += kThreads) {
        const int l = i / 3, a = i % 3;
        mx = fmax(mx, fabs(D.Hll[6 * (size_t)l + (a == 0 ? 0 : a == 1 ? 3 : 5)]));
      }
      lambda = tau * block_max(mx, s16);
      ni = 2;
      n_bad = 0;
    }
    double rho = 0;
    int qmax = 0;
    bool stopped = false;
    do {
      const bool ok2 = solve_schur(D, lambda, Hs, bs, s16, &s_flag);
      double temp_chi, scale = 0;
      if (ok2) {
        for (int f = threadIdx.x; f < D.n_free; f += kThreads) {
          const int p = D.free_pose[f];
          pose_oplus(D.q + 4 * p, D.t + 3 * p, D.xp + 6 * f, D.q_try + 4 * p, D.t_try + 3 * p);
        }
        for (int i = threadIdx.x; i < 3 * D.n_points; i += kThreads) D.X_try[i] = D.X[i] + D.xl[i];
        __syncthreads();
        temp_chi = compute_errors(D, D.q_try, D.t_try, D.X_try, s16);
        double loc = 0;  // computeScale
        for (int i = threadIdx.x; i < n; i += kThreads) {
          const int f = i / 6, a = i % 6;
          loc += D.xp[i] * (lambda * D.xp[i] + D.bp[6 * f + a]);
        }
        for (int i = threadIdx.x; i < 3 * D.n_points; i += kThreads) loc += D.xl[i] * (lambda * D.xl[i] + D.bl[i]);
        scale = block_sum(loc, s16);
      } else {
        temp_chi = compute_errors(D, D.q, D.t, D.X, s16);
        temp_chi = 1.79769313486231570e308;
      }
      rho = (current_chi - temp_chi) / (scale + 1e-3);
      if (rho > 0 && isfinite(temp_chi)) {
        double alpha = 1. - gfs_glibc::pow3(2 * rho - 1);
        alpha = fmin(alpha, good_up);
        lambda *= fmax(good_lo, alpha);
        ni = 2;
        current_chi = temp_chi;
        for (int f = threadIdx.x; f < D.n_free; f += kThreads) {  // discardTop: keep the trial
          const int p = D.free_pose[f];
          for (int k = 0; k < 4; k++) D.q[4 * p + k] = D.q_try[4 * p + k];
          for (int k = 0; k < 3; k++) D.t[3 * p + k] = D.t_try[3 * p + k];
        }
        for (int i = threadIdx.x; i < 3 * D.n_points; i += kThreads) D.X[i] = D.X_try[i];
      } else {
        lambda *= ni;
        ni *= 2;
      }
      if (threadIdx.x == 0) s_flag = *D.stop;
      __syncthreads();
      qmax++;
      stopped = s_flag != 0;
      __syncthreads();
    } while (rho < 0 && qmax < 10 && !stopped);
    iters++;
    last_chi = current_chi;
    if (qmax == 10 || rho == 0) break;
    if ((ini_chi - current_chi) * 1e3 < ini_chi)
      n_bad++;
    else
      n_bad = 0;
    if (n_bad >= 3) break;
  }
  if (threadIdx.x == 0) {
    D.out_info[0] = iters;
    D.out_info[1] = 0;
    D.out_stats[0] = last_chi;
    D.out_stats[1] = lambda;
  }
}

// ================================================================================================
// Multi-kernel path: the same algorithm, one launch per phase so that every phase uses the whole chip instead of one
// CU (the single-workgroup kernel above is latency-bound: 61 ms for the C5 window).  The scalar LM logic lives in
// LbaState (HBM) and runs in k_lba_begin / k_lba_decide; the host only enqueues the phases and reads two flags per trial.
// The two estimate buffers (q/t/X and q_try/t_try/X_try) swap roles when a step is accepted (S->cur).
// ================================================================================================
constexpr int kMk = 256;  // threads per workgroup of the wide kernels

__device__ __forceinline__ double* sel(double* a, double* b, int which) { return which ? b : a; }

// deterministic 256-thread block sum: wave shuffle tree, then the four waves in order; result in every thread
__device__ double block_sum256(double v, double* s4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) v += __shfl_down(v, ofs, 64);
  __syncthreads();
  if (lane == 0) s4[wave] = v;
  __syncthreads();
  return ((s4[0] + s4[1]) + s4[2]) + s4[3];
}

__device__ __forceinline__ void b_init(const LbaDev& D, const int bx, const int gdx) {
  const int g = bx * kMk + threadIdx.x, G = gdx * kMk;
  for (int i = g; i < D.n_poses; i += G) {
    double q[4] = {D.pose_q0[4 * i], D.pose_q0[4 * i + 1], D.pose_q0[4 * i + 2], D.pose_q0[4 * i + 3]};
    normalize_rotation(q);
    for (int k = 0; k < 4; k++) D.q[4 * i + k] = D.q_try[4 * i + k] = q[k];
    for (int k = 0; k < 3; k++) D.t[3 * i + k] = D.t_try[3 * i + k] = D.pose_t0[3 * i + k];
  }
  for (int i = g; i < 3 * D.n_points; i += G) D.X[i] = D.X_try[i] = D.points0[i];
  for (size_t i = g; i < (size_t)D.n_edges * 18; i += G) D.Hpl[i] = 0;  // the blocks of fixed-pose edges stay zero
  if (g == 0) {
    LbaState& S = *D.S;
    S.lambda = -1;
    S.ni = 2;
    S.current_chi = S.ini_chi = S.temp_chi = S.rho = S.last_chi = 0;
    S.cur = 0;
    S.qmax = S.n_bad = S.iters = 0;
    S.solve_ok = 1;
    S.again = S.terminate = 0;
    S.phase = S.it = 0;
    S.err_at_cur = 0;
    S.spec_ok = S.pad_ = 0;
  }
}
__global__ __launch_bounds__(kMk) void k_lba_init(LbaDev D) { b_init(D, blockIdx.x, gridDim.x); }


// computeActiveErrors on the accepted (trial = 0) or the trial estimate (trial = 1; falls back to the accepted one when the
// linear solve failed, like the reference which restores the estimate before recomputing the errors)
__device__ __forceinline__ void b_errors(const LbaDev& D, const int bx, const int gdx, int trial) {
  __shared__ double s4[4];
  const LbaState& S = *D.S;
  // computeActiveErrors at the top of an iteration that follows an accepted trial would recompute, at the same estimate, what that
  // trial's own evaluation left in chi2 / err / part_chi
  if (!trial && S.err_at_cur) return;
  const int which = (trial && S.solve_ok) ? (S.cur ^ 1) : S.cur;
  const double *q = sel(D.q, D.q_try, which), *t = sel(D.t, D.t_try, which), *X = sel(D.X, D.X_try, which);
  double local = 0;
  const int e = bx * kMk + threadIdx.x;
  if (e < D.n_edges) {
    double xc[3], r[3];
    edge_residual(D, e, q, t, X, xc, r);
    const double c = (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]) * D.e_w[e];
    D.chi2[e] = c;
    D.err[3 * e] = r[0];
    D.err[3 * e + 1] = r[1];
    D.err[3 * e + 2] = r[2];
    double r0, r1;
    huber(c, D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
    local = r0;
  }
  const double tot = block_sum256(local, s4);
  if (threadIdx.x == 0) D.part_chi[bx] = tot;
}
// gate = 1 (k_lba_errors / build_landmarks / build_poses / begin): a launch queued speculatively behind k_lba_decide -- it runs only if that
// trial was accepted and the loop goes on (LbaState::spec_ok); uniform per launch, in front of every barrier
__global__ __launch_bounds__(kMk) void k_lba_errors(LbaDev D, int trial, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_errors(D, blockIdx.x, gridDim.x, trial);
}


// buildSystem, landmark side: Hll, bl and the per-edge pose-landmark blocks.  16 lanes per landmark (its edges over the
// lanes, the 9 sums folded by xor-shuffles inside the group in a fixed order), 8 landmarks per 128-thread workgroup.
__device__ __forceinline__ void b_build_landmarks(const LbaDev& D, const int bx, const int gdx) {
  // One thread per EDGE: a workgroup takes a run of whole landmarks holding at most kMk edges (lm_wg, packed by the host; a
  // landmark with more than kMk edges gets a workgroup to itself and the threads stride over its edges).  The nine landmark
  // sums (Hll, bl) of an edge go through LDS and thread <-> (landmark, component) adds its landmark's edges in edge order.
  // (16 lanes per landmark, the round-1 mapping, left half the lanes idle: a landmark of this window has 16.5 edges on
  // average, so nearly every group ran a second, almost empty pass.)
  __shared__ double s_c[kMk][9];
  const LbaState& S = *D.S;
  const double *q = sel(D.q, D.q_try, S.cur), *t = sel(D.t, D.t_try, S.cur), *X = sel(D.X, D.X_try, S.cur);
  const int l0 = D.lm_wg[bx], l1 = D.lm_wg[bx + 1];
  const int e0 = D.pt_begin[l0], e1 = D.pt_begin[l1];
  double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // H (xx,xy,xz,yy,yz,zz), b
  for (int e = e0 + threadIdx.x; e < e1; e += kMk) {
    double xc[3], r[3], Ji[9], Jj[18];
    edge_residual(D, e, q, t, X, xc, r);
    edge_jacobians(D, e, q, xc, Ji, Jj);
    double r0, r1;
    huber(D.chi2[e], D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
    const double w = r1 * D.e_w[e];
    double omr[3];
    for (int k = 0; k < 3; k++) omr[k] = -(D.e_w[e] * D.err[3 * e + k]) * r1;
    int o = 0;
    for (int a = 0; a < 3; a++) {
      acc[6 + a] += Ji[a] * omr[0] + Ji[3 + a] * omr[1] + Ji[6 + a] * omr[2];
      for (int c = a; c < 3; c++) acc[o++] += Ji[a] * w * Ji[c] + Ji[3 + a] * w * Ji[3 + c] + Ji[6 + a] * w * Ji[6 + c];
    }
    if (D.free_index[D.e_pose[e]] >= 0) {
      double* B = D.Hpl + 18 * (size_t)e;
      for (int a = 0; a < 6; a++)
        for (int c = 0; c < 3; c++) B[3 * a + c] = Jj[a] * w * Ji[c] + Jj[6 + a] * w * Ji[3 + c] + Jj[12 + a] * w * Ji[6 + c];
    }
  }
#pragma unroll
  for (int k = 0; k < 9; k++) s_c[threadIdx.x][k] = acc[k];
  __syncthreads();
  const int nl = l1 - l0;
  for (int u = threadIdx.x; u < nl * 9; u += kMk) {
    const int l = l0 + u / 9, k = u - (u / 9) * 9;
    int s0 = D.pt_begin[l] - e0, s1 = D.pt_begin[l + 1] - e0;
    if (e1 - e0 > kMk) {  // one landmark with more edges than threads: every thread holds a partial sum of it
      s0 = 0;
      s1 = kMk;
    }
    double v = 0;
    for (int sl = s0; sl < s1; sl++) v += s_c[sl][k];
    if (k < 6)
      D.Hll[6 * (size_t)l + k] = v;
    else
      D.bl[3 * (size_t)l + (k - 6)] = v;
  }
  // several edges between one pose and one landmark (a rig seeing the point with two cameras): g2o adds their J_pose^T Omega J_point
  // into the one Hpl block of that vertex pair (core/block_solver.hpp:143-295 allocates it once).  The first edge's block takes
  // the sum, in edge order, and the later ones are cleared: everything downstream (Schur products, back-substitution) sums over
  // edges.  gfs_lba_linearize (mode 1) reports the per-edge blocks instead.
  if (D.e_dup && D.mode == 0) {
    __threadfence_block();
    __syncthreads();
    for (int l = l0 + threadIdx.x; l < l1; l += kMk) {
      for (int e = D.pt_begin[l]; e < D.pt_begin[l + 1]; e++) {
        const int first = D.e_dup[e];
        if (first < 0 || D.free_index[D.e_pose[e]] < 0) continue;
        double* B = D.Hpl + 18 * (size_t)e;
        double* A = D.Hpl + 18 * (size_t)first;
        for (int k = 0; k < 18; k++) {
          A[k] += B[k];
          B[k] = 0;
        }
      }
    }
  }
}
__global__ __launch_bounds__(kMk) void k_lba_build_landmarks(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_build_landmarks(D, blockIdx.x, gridDim.x);
}


// buildSystem, pose side: one workgroup per free pose, threads over its edges, fixed-order reduction
// (kPoseWg threads a pose: a free pose of the configs[4] window has ~2 500 edges, and an edge is a chain of dependent loads --
// index, then its data -- so the kernel's time is edges per THREAD: 10 at 256 threads, 27 us a launch; 2.5 at 1 024, round 5)
constexpr int kPoseWg = 1024;
__device__ __forceinline__ void b_build_poses(const LbaDev& D, const int bx, const int gdx) {
  __shared__ double s_buf[(kPoseWg / 64) * 32];
  const LbaState& S = *D.S;
  const double *q = sel(D.q, D.q_try, S.cur), *t = sel(D.t, D.t_try, S.cur), *X = sel(D.X, D.X_try, S.cur);
  const int f = bx;
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; k++) acc[k] = 0;
  for (int i = D.pose_begin[f] + threadIdx.x; i < D.pose_begin[f + 1]; i += kPoseWg) {
    const int e = D.pose_edges[i];
    double xc[3], r[3], Ji[9], Jj[18];
    edge_residual(D, e, q, t, X, xc, r);
    edge_jacobians(D, e, q, xc, Ji, Jj);
    double r0, r1;
    huber(D.chi2[e], D.e_stereo[e] ? D.huber_stereo : D.huber_mono, &r0, &r1);
    const double w = r1 * D.e_w[e];
    double omr[3];
    for (int k = 0; k < 3; k++) omr[k] = -(D.e_w[e] * D.err[3 * e + k]) * r1;
    int o = 0;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = a; c < 6; c++) acc[o++] += Jj[a] * w * Jj[c] + Jj[6 + a] * w * Jj[6 + c] + Jj[12 + a] * w * Jj[12 + c];
#pragma unroll
    for (int a = 0; a < 6; a++) acc[21 + a] += Jj[a] * omr[0] + Jj[6 + a] * omr[1] + Jj[12 + a] * omr[2];
  }
  const double v = gfs_red::block_sum_many<27, kPoseWg / 64>(acc, s_buf);
  if (threadIdx.x < 21)
    D.Hpp[21 * f + threadIdx.x] = v;
  else if (threadIdx.x < 27)
    D.bp[6 * f + (threadIdx.x - 21)] = v;
}
__global__ __launch_bounds__(kPoseWg) void k_lba_build_poses(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_build_poses(D, blockIdx.x, gridDim.x);
}


// start of an LM iteration: currentChi, and at iteration 0 computeLambdaInit (tau * max |diag H|)
__device__ __forceinline__ void b_begin(const LbaDev& D, const int bx, const int gdx, int iteration) {
  __shared__ double s16[16];
  LbaState& S = *D.S;
  double chi = 0;
  if (threadIdx.x == 0)
    for (int b = 0; b < D.n_err_blocks; b++) chi += D.part_chi[b];
  if (iteration == 0) {
    double mx = 0;
    for (int i = threadIdx.x; i < D.n_free * 6; i += kThreads) {
      const int f = i / 6, a = i % 6;
      mx = fmax(mx, fabs(D.Hpp[21 * f + (a * 6 - a * (a - 1) / 2)]));
    }
    for (int i = threadIdx.x; i < D.n_points * 3; i += kThreads) {
      const int l = i / 3, a = i % 3;
      mx = fmax(mx, fabs(D.Hll[6 * (size_t)l + (a == 0 ? 0 : a == 1 ? 3 : 5)]));
    }
    mx = block_max(mx, s16);
    if (threadIdx.x == 0) {
      S.lambda = 1e-5 * mx;
      S.ni = 2;
      S.n_bad = 0;
    }
  }
  if (threadIdx.x == 0) {
    S.current_chi = S.ini_chi = chi;
    S.rho = 0;
    S.qmax = 0;
    S.again = 0;
  }
}
__global__ __launch_bounds__(kThreads) void k_lba_begin(LbaDev D, int iteration, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_begin(D, blockIdx.x, gridDim.x, iteration);
}


__device__ __forceinline__ void b_dinv(const LbaDev& D, const int bx, const int gdx) {
  const int l = bx * kMk + threadIdx.x;
  if (l < D.n_points) inv3_sym(D.Hll + 6 * (size_t)l, D.S->lambda, D.Dinv + 6 * (size_t)l);
}
__global__ __launch_bounds__(kMk) void k_lba_dinv(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_dinv(D, blockIdx.x, gridDim.x);
}


// Schur complement, one workgroup per pose pair (i1 >= i2): Hs(i1,i2) = [i1==i2](Hpp + lambda I) - sum_l B_i1 Dinv_l B_i2^T
__device__ __forceinline__ void b_schur(const LbaDev& D, const int bx, const int gdx) {
  __shared__ double s_buf[(kMk / 64) * 32];
  const double lambda = D.S->lambda;
  const int pr = bx;
  int i1 = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
  while (i1 * (i1 + 1) / 2 > pr) i1--;
  while ((i1 + 1) * (i1 + 2) / 2 <= pr) i1++;
  const int i2 = pr - i1 * (i1 + 1) / 2;
  double acc[36], accb[6];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) accb[k] = 0;
  const int* eo = D.edge_of + (size_t)i2 * D.n_points;
  for (int i = D.pose_begin[i1] + threadIdx.x; i < D.pose_begin[i1 + 1]; i += kMk) {
    const int e1 = D.pose_edges[i];
    const int l = D.e_point[e1];
    const int e2 = eo[l];
    if (e2 < 0) continue;
    const double* Bi = D.Hpl + 18 * (size_t)e1;
    const double* Bj = D.Hpl + 18 * (size_t)e2;
    const double* Di = D.Dinv + 6 * (size_t)l;
    const double d9[9] = {Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5]};
    double BD[18];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c < 3; c++) BD[3 * a + c] = Bi[3 * a] * d9[c] + Bi[3 * a + 1] * d9[3 + c] + Bi[3 * a + 2] * d9[6 + c];
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
      for (int c = 0; c < 6; c++) acc[6 * a + c] += BD[3 * a] * Bj[3 * c] + BD[3 * a + 1] * Bj[3 * c + 1] + BD[3 * a + 2] * Bj[3 * c + 2];
    if (i1 == i2) {
      const double* bl = D.bl + 3 * (size_t)l;
#pragma unroll
      for (int a = 0; a < 6; a++) accb[a] += BD[3 * a] * bl[0] + BD[3 * a + 1] * bl[1] + BD[3 * a + 2] * bl[2];
    }
  }
  // 36 + 6 sums over the workgroup: two many-value reductions (32, then the remaining 4 + 6)
  double head[32], tail[10];
#pragma unroll
  for (int k = 0; k < 32; k++) head[k] = acc[k];
#pragma unroll
  for (int k = 0; k < 4; k++) tail[k] = acc[32 + k];
#pragma unroll
  for (int k = 0; k < 6; k++) tail[4 + k] = accb[k];
  const double v0 = gfs_red::block_sum_many<32, kMk / 64>(head, s_buf);
  const double v1 = gfs_red::block_sum_many<10, kMk / 64>(tail, s_buf);
  const int tk = threadIdx.x;
  // threads 0..31 own acc[0..31] (v0); threads 0..3 own acc[32..35] and threads 4..9 own accb[0..5] (v1)
  auto store_h = [&](int k, double val) {
    const int a = k / 6, c = k % 6;
    if (i1 != i2) {
      D.Hs[tri(6 * i1 + a, 6 * i2 + c)] = -val;
    } else if (a >= c) {
      const int ua = c, uc = a;  // Hpp upper-triangle index of (c, a)
      const double hpp = D.Hpp[21 * i1 + (ua * 6 - ua * (ua - 1) / 2 + (uc - ua))];
      D.Hs[tri(6 * i1 + a, 6 * i1 + c)] = hpp + (a == c ? lambda : 0.0) - val;
    }
  };
  if (tk < 32) store_h(tk, v0);
  if (tk < 4) store_h(32 + tk, v1);
  if (i1 == i2 && tk >= 4 && tk < 10) D.bs[6 * i1 + (tk - 4)] = D.bp[6 * i1 + (tk - 4)] - v1;
}
__global__ __launch_bounds__(kMk) void k_lba_schur(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_schur(D, blockIdx.x, gridDim.x);
}

// The same products taken landmark by landmark.  b_schur gives a workgroup to a pose pair, and every pair re-reads the 6x3
// blocks of its two poses for each common landmark: Hpl is read ~n_free times (L2 traffic, 0.8 MB per pair).  Here a workgroup
// takes a chunk of kSchurPts landmarks: the blocks B_f (6x3) of every free pose seeing a landmark, and B_f Dinv_l, are staged
// in LDS once (schur_sub landmarks at a time), and thread <-> pose pair (i1 >= i2) accumulates its 6x6 block (and, on the
// diagonal, B Dinv b_l) over the chunk's landmarks in index order.  Per-chunk partial blocks go to HBM; b_schur_reduce adds
// them in chunk order: the sums have a fixed order that does not depend on the launch (a window solved alone or in a batch
// gives the same bits).  Pose pairs beyond kMk: blockIdx.x = chunk * n_pair_tiles + tile.
constexpr int kSchurPts = 64;
constexpr int kSchurSlot = 38;  // doubles per staged (landmark, pose): B (18), B Dinv (18), + 2: 16 lanes' 16-byte reads of consecutive poses hit distinct banks

__device__ __forceinline__ void b_schur_chunks(const LbaDev& D, const int bx) {
  extern __shared__ __align__(16) double lds[];
  const int F = D.n_free, NP = D.n_points, SB = D.schur_sub;
  const int chunk = bx / D.n_pair_tiles, tile = bx - chunk * D.n_pair_tiles;
  double* s_blk = lds;                                   // [SB][F][kSchurSlot]
  double* s_bl = s_blk + (size_t)SB * F * kSchurSlot;    // [SB][4]
  int* s_edge = (int*)(s_bl + 4 * SB);                   // [SB][F] edge id or -1
  const int npairs = F * (F + 1) / 2;
  const int pr = tile * kMk + threadIdx.x;
  int i1 = 0, i2 = 0;
  if (pr < npairs) {
    i1 = (int)((sqrt(8.0 * pr + 1.0) - 1.0) * 0.5);
    while (i1 * (i1 + 1) / 2 > pr) i1--;
    while ((i1 + 1) * (i1 + 2) / 2 <= pr) i1++;
    i2 = pr - i1 * (i1 + 1) / 2;
  }
  double acc[36], accb[6];
#pragma unroll
  for (int k = 0; k < 36; k++) acc[k] = 0;
#pragma unroll
  for (int k = 0; k < 6; k++) accb[k] = 0;
  const int l_begin = chunk * kSchurPts, l_end = min(l_begin + kSchurPts, NP);
  for (int l0 = l_begin; l0 < l_end; l0 += SB) {
    const int nl = min(SB, l_end - l0);
    __syncthreads();  // the previous sub-batch has been consumed
    for (int u = threadIdx.x; u < nl * F; u += kMk) {
      const int p = u / F, f = u - p * F, l = l0 + p;
      const int e = D.edge_of[(size_t)f * NP + l];
      s_edge[p * F + f] = e;
      if (e >= 0) {
        const double* B = D.Hpl + 18 * (size_t)e;
        const double* Di = D.Dinv + 6 * (size_t)l;
        const double d9[9] = {Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5]};
        double* o = s_blk + (size_t)(p * F + f) * kSchurSlot;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const double b0 = B[3 * a], b1 = B[3 * a + 1], b2 = B[3 * a + 2];
          o[3 * a] = b0;
          o[3 * a + 1] = b1;
          o[3 * a + 2] = b2;
#pragma unroll
          for (int c = 0; c < 3; c++) o[18 + 3 * a + c] = b0 * d9[c] + b1 * d9[3 + c] + b2 * d9[6 + c];
        }
      }
    }
    if (threadIdx.x < nl * 3) s_bl[4 * (threadIdx.x / 3) + threadIdx.x % 3] = D.bl[3 * (size_t)l0 + threadIdx.x];
    __syncthreads();
    if (pr < npairs) {
      for (int p = 0; p < nl; p++) {
        if (s_edge[p * F + i1] < 0 || s_edge[p * F + i2] < 0) continue;
        const double* BD = s_blk + (size_t)(p * F + i1) * kSchurSlot + 18;
        const double* Bj = s_blk + (size_t)(p * F + i2) * kSchurSlot;
        double bd[18], bj[18];
#pragma unroll
        for (int k = 0; k < 18; k++) {
          bd[k] = BD[k];
          bj[k] = Bj[k];
        }
        // fused multiply-adds: this loop is the kernel's VALU time, and the order of the sums already differs from g2o's
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int c = 0; c < 6; c++)
            acc[6 * a + c] = fma(bd[3 * a + 2], bj[3 * c + 2], fma(bd[3 * a + 1], bj[3 * c + 1], fma(bd[3 * a], bj[3 * c], acc[6 * a + c])));
        if (i1 == i2) {
          const double* bl = s_bl + 4 * p;
#pragma unroll
          for (int a = 0; a < 6; a++) accb[a] = fma(bd[3 * a + 2], bl[2], fma(bd[3 * a + 1], bl[1], fma(bd[3 * a], bl[0], accb[a])));
        }
      }
    }
  }
  if (pr < npairs) {
    double* o = D.schur_part + ((size_t)chunk * npairs + pr) * 36;
#pragma unroll
    for (int k = 0; k < 36; k++) o[k] = acc[k];
    if (i1 == i2) {
      double* ob = D.schur_part_b + ((size_t)chunk * F + i1) * 6;
#pragma unroll
      for (int k = 0; k < 6; k++) ob[k] = accb[k];
    }
  }
}
__global__ __launch_bounds__(kMk) void k_lba_schur_chunks(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_schur_chunks(D, blockIdx.x);
}

// ---- the same sum on the matrix cores.  With W_l (6F x 3: the blocks B_f of landmark l stacked, zero where pose f does not see
// it) the Schur sum over landmarks is  S = sum_l (W_l Dinv_l) W_l^T = [WD_1 WD_2 ...] [W_1 W_2 ...]^T : one (6F x 3L)(3L x 6F)
// product, double precision, v_mfma_f64_16x16x4_f64.  The right-hand side rides along as one more row: row n = 6F of WD holds
// Dinv_l b_l, so that S[n][c] = sum_l B_c Dinv_l b_l.  A workgroup takes a chunk of kSchurMPts landmarks and one 128 x 128 block
// (bi >= bj) of S: kSchurSub landmarks (24 columns = 6 k-steps, nothing padded) are staged at a time in LDS, column-major with a
// row stride of 144 doubles (the four k-groups of a fragment read land in disjoint bank halves), each of the 4 waves owns a share
// of the block's 16 x 16 tiles and keeps them in registers across the chunk (diagonal block: tile rows w and 7 - w of the lower
// triangle, 9 tiles a wave; off-diagonal: rows 2w, 2w + 1, 16 tiles).  The edges of a run of landmarks are contiguous
// (landmark-major order), so the staging threads read e_pose / e_point / Hpl of edge e0 + tid directly, one sub-batch ahead of the
// products.  Per-chunk partial blocks go to HBM ([chunk][row][col], leading dimension 128 NB) and b_schur_reduce adds them in
// chunk order.  Operands are products of single roundings (fused multiply-add inside the matrix core), like b_schur_chunks.
// (round 6: 64 landmarks a workgroup instead of 128 -- a configs[4] window of 3 000 landmarks is 47 workgroups instead of 24 on the 256 CUs,
//  one window 2.13 - 2.16 -> 2.00 ms; 32: 2.02 ms; the batched path, whose launches fill the chip either way, is unchanged.  The partial
//  blocks are added in chunk order by the reduce kernel: the chunk size is part of the arithmetic, the same for one window and a batch)
#ifndef GFS_SCHUR_MPTS
#define GFS_SCHUR_MPTS 64
#endif
constexpr int kSchurMPts = GFS_SCHUR_MPTS;  // landmarks per workgroup
constexpr int kSchurSub = 8;     // landmarks staged at a time
constexpr int kSchurLd = 144;    // row stride (doubles) of a staged column
typedef double d4_t __attribute__((ext_vector_type(4)));

template <bool DIAG, int W>
__device__ __forceinline__ void schur_mfma_ksteps(const double* __restrict__ s_a, const double* __restrict__ s_b, d4_t (&acc0)[8],
                                                  d4_t (&acc1)[8]) {
  constexpr int r0 = DIAG ? W : 2 * W, r1 = DIAG ? 7 - W : 2 * W + 1;
  constexpr int nc0 = DIAG ? r0 + 1 : 8, nc1 = DIAG ? r1 + 1 : 8;
  const int lane = threadIdx.x & 63, row = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < 3 * kSchurSub / 4; ks++) {
    const double* ca = s_a + (4 * ks + kk) * kSchurLd + row;
    const double* cb = s_b + (4 * ks + kk) * kSchurLd + row;
    const double a0 = ca[16 * r0], a1 = ca[16 * r1];
    double b[8];
#pragma unroll
    for (int c = 0; c < 8; c++)
      if (c < nc1 || c < nc0) b[c] = cb[16 * c];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (c < nc0) acc0[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[c], acc0[c], 0, 0, 0);
      if (c < nc1) acc1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[c], acc1[c], 0, 0, 0);
    }
  }
}

template <bool DIAG, int W>
__device__ __forceinline__ void schur_mfma_store(double* __restrict__ out, int ld, const d4_t (&acc0)[8], const d4_t (&acc1)[8]) {
  constexpr int r0 = DIAG ? W : 2 * W, r1 = DIAG ? 7 - W : 2 * W + 1;
  constexpr int nc0 = DIAG ? r0 + 1 : 8, nc1 = DIAG ? r1 + 1 : 8;
  const int lane = threadIdx.x & 63, col = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int c = 0; c < 8; c++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (c < nc0) out[(size_t)(16 * r0 + rq + 4 * i) * ld + 16 * c + col] = acc0[c][i];
      if (c < nc1) out[(size_t)(16 * r1 + rq + 4 * i) * ld + 16 * c + col] = acc1[c][i];
    }
}

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. it
// would wait for the global loads of the NEXT sub-batch that are meant to stay in flight across the products
#define GFS_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <bool kOneBlock>
__device__ __forceinline__ void b_schur_mfma(const LbaDev& D, const int bx) {
  extern __shared__ __align__(16) double lds[];
  const int F = D.n_free, NP = D.n_points, n = 6 * F;
  const int NB = (n + 1 + 127) / 128, nbp = NB * (NB + 1) / 2, ld = 128 * NB;
  const int chunk = bx / nbp, bp = bx - chunk * nbp;
  int bi = (int)((sqrt(8.0 * bp + 1.0) - 1.0) * 0.5);
  while (bi * (bi + 1) / 2 > bp) bi--;
  while ((bi + 1) * (bi + 2) / 2 <= bp) bi++;
  const int bj = bp - bi * (bi + 1) / 2;
  double* s_a = lds;                                  // [24][kSchurLd]  rows of block bi of [WD; Dinv b]
  double* s_b = s_a + 3 * kSchurSub * kSchurLd;       // [24][kSchurLd]  rows of block bj of W
  double* s_dinv = s_b + 3 * kSchurSub * kSchurLd;    // [kSchurMPts][6]
  double* s_v = s_dinv + 6 * kSchurMPts;              // [kSchurMPts][3]  Dinv_l b_l
  int* s_ptb = (int*)(s_v + 3 * kSchurMPts);          // [kSchurMPts + 1]
  int* s_free = s_ptb + kSchurMPts + 1;               // [n_poses]
  const int tid = threadIdx.x, wave = tid >> 6;
  const int l_begin = chunk * kSchurMPts, l_end = min(l_begin + kSchurMPts, NP), nlm = l_end - l_begin;
  for (int i = tid; i < 6 * nlm; i += kMk) s_dinv[i] = D.Dinv[6 * (size_t)l_begin + i];
  for (int i = tid; i <= nlm; i += kMk) s_ptb[i] = D.pt_begin[l_begin + i];
  for (int i = tid; i < D.n_poses; i += kMk) s_free[i] = D.free_index[i];
  __syncthreads();
  for (int p = tid; p < nlm; p += kMk) {
    const double* Di = s_dinv + 6 * p;
    const double* bl = D.bl + 3 * (size_t)(l_begin + p);
    const double b0 = bl[0], b1 = bl[1], b2 = bl[2];
    s_v[3 * p] = fma(Di[2], b2, fma(Di[1], b1, Di[0] * b0));
    s_v[3 * p + 1] = fma(Di[4], b2, fma(Di[3], b1, Di[1] * b0));
    s_v[3 * p + 2] = fma(Di[5], b2, fma(Di[4], b1, Di[2] * b0));
  }
  d4_t acc0[8], acc1[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc0[c] = acc1[c] = d4_t{0, 0, 0, 0};
  // edge prefetch registers (one edge per thread and pass; a sub-batch has at most kSchurSub * n_poses edges)
  double Bn[18];
  int pose_n = -1, lm_n = 0;
  auto fetch = [&](int sb_l0, int pass) {
    pose_n = -1;
    if (sb_l0 >= l_end) return;
    const int e0 = s_ptb[sb_l0 - l_begin], e1 = s_ptb[min(sb_l0 + kSchurSub, l_end) - l_begin];
    const int e = e0 + pass * kMk + tid;
    if (e < e1 && !(D.e_dup && D.e_dup[e] >= 0)) {  // (a duplicate's block has been added to its first edge's)
      pose_n = D.e_pose[e];
      lm_n = D.e_point[e];
      const double* B = D.Hpl + 18 * (size_t)e;
#pragma unroll
      for (int k = 0; k < 18; k++) Bn[k] = B[k];
    }
  };
  const int a_lo = 128 * bi, b_lo = 128 * bj;
  fetch(l_begin, 0);
  for (int l0 = l_begin; l0 < l_end; l0 += kSchurSub) {
    GFS_LDS_BARRIER();  // the previous sub-batch has been consumed (and s_v is complete the first time round)
    for (int i = tid; i < 2 * 3 * kSchurSub * kSchurLd / 2; i += kMk) ((double2*)s_a)[i] = double2{0.0, 0.0};  // s_a and s_b are adjacent
    GFS_LDS_BARRIER();
    const int e0 = s_ptb[l0 - l_begin], e1 = s_ptb[min(l0 + kSchurSub, l_end) - l_begin];
    const int npass = (e1 - e0 + kMk - 1) / kMk;
    for (int pass = 0; pass < npass; pass++) {
      if (pass > 0) fetch(l0, pass);
      if (pose_n >= 0) {
        const int f = s_free[pose_n], p = lm_n - l0;
        if (f >= 0) {
          const double* Di = s_dinv + 6 * (lm_n - l_begin);
          const double d9[9] = {Di[0], Di[1], Di[2], Di[1], Di[3], Di[4], Di[2], Di[4], Di[5]};
#pragma unroll
          for (int a = 0; a < 6; a++) {
            const int rg = 6 * f + a;  // global row
            const double b0 = Bn[3 * a], b1 = Bn[3 * a + 1], b2 = Bn[3 * a + 2];
            if ((unsigned)(rg - b_lo) < 128u) {
#pragma unroll
              for (int k = 0; k < 3; k++) s_b[(3 * p + k) * kSchurLd + rg - b_lo] = Bn[3 * a + k];
            }
            if ((unsigned)(rg - a_lo) < 128u) {
#pragma unroll
              for (int k = 0; k < 3; k++) s_a[(3 * p + k) * kSchurLd + rg - a_lo] = fma(b2, d9[6 + k], fma(b1, d9[3 + k], b0 * d9[k]));
            }
          }
        }
      }
    }
    // the extra row: Dinv_l b_l at global row n
    if ((unsigned)(n - a_lo) < 128u && tid < 3 * kSchurSub && l0 + tid / 3 < l_end)
      s_a[tid * kSchurLd + n - a_lo] = s_v[3 * (l0 - l_begin) + tid];
    fetch(l0 + kSchurSub, 0);  // the next sub-batch's edges travel while the products run
    GFS_LDS_BARRIER();
    if (kOneBlock || bi == bj) {
      switch (wave) {
        case 0: schur_mfma_ksteps<true, 0>(s_a, s_b, acc0, acc1); break;
        case 1: schur_mfma_ksteps<true, 1>(s_a, s_b, acc0, acc1); break;
        case 2: schur_mfma_ksteps<true, 2>(s_a, s_b, acc0, acc1); break;
        default: schur_mfma_ksteps<true, 3>(s_a, s_b, acc0, acc1); break;
      }
    } else {
      switch (wave) {
        case 0: schur_mfma_ksteps<false, 0>(s_a, s_b, acc0, acc1); break;
        case 1: schur_mfma_ksteps<false, 1>(s_a, s_b, acc0, acc1); break;
        case 2: schur_mfma_ksteps<false, 2>(s_a, s_b, acc0, acc1); break;
        default: schur_mfma_ksteps<false, 3>(s_a, s_b, acc0, acc1); break;
      }
    }
  }
  double* out = D.schur_part + (size_t)chunk * ld * ld + (size_t)a_lo * ld + b_lo;
  if (kOneBlock || bi == bj) {
    switch (wave) {
      case 0: schur_mfma_store<true, 0>(out, ld, acc0, acc1); break;
      case 1: schur_mfma_store<true, 1>(out, ld, acc0, acc1); break;
      case 2: schur_mfma_store<true, 2>(out, ld, acc0, acc1); break;
      default: schur_mfma_store<true, 3>(out, ld, acc0, acc1); break;
    }
  } else {
    switch (wave) {
      case 0: schur_mfma_store<false, 0>(out, ld, acc0, acc1); break;
      case 1: schur_mfma_store<false, 1>(out, ld, acc0, acc1); break;
      case 2: schur_mfma_store<false, 2>(out, ld, acc0, acc1); break;
      default: schur_mfma_store<false, 3>(out, ld, acc0, acc1); break;
    }
  }
}
template <bool kOneBlock>
__global__ __launch_bounds__(kMk, kOneBlock ? 2 : 1) void k_lba_schur_mfma(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_schur_mfma<kOneBlock>(D, blockIdx.x);
}

// the reduction for b_schur_mfma's partial blocks: Hs entry (r, c), r >= c, and bs[c] = bp[c] - S[n][c]
__device__ __forceinline__ void b_schur_reduce_mfma(const LbaDev& D, const int bx) {
  const int F = D.n_free, n = 6 * F, ntri = n * (n + 1) / 2;
  const int NB = (n + 1 + 127) / 128, ld = 128 * NB;
  const int k = bx * kMk + threadIdx.x;
  if (k >= ntri + n) return;
  int r, c;
  if (k < ntri) {
    r = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > k) r--;
    while ((r + 1) * (r + 2) / 2 <= k) r++;
    c = k - r * (r + 1) / 2;
  } else {
    r = n;
    c = k - ntri;
  }
  const double* part = D.schur_part + (size_t)r * ld + c;
  double sum = 0;
  for (int ch = 0; ch < D.n_schur_chunks; ch++) sum += part[(size_t)ch * ld * ld];
  if (k >= ntri) {
    D.bs[c] = D.bp[c] - sum;
  } else {
    const int i1 = r / 6, a = r - 6 * i1, i2 = c / 6, cc = c - 6 * i2;
    if (i1 != i2) {
      D.Hs[k] = -sum;
    } else {
      const double hpp = D.Hpp[21 * i1 + (cc * 6 - cc * (cc - 1) / 2 + (a - cc))];
      D.Hs[k] = hpp + (a == cc ? D.S->lambda : 0.0) - sum;
    }
  }
}
__global__ __launch_bounds__(kMk) void k_lba_schur_reduce_mfma(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_schur_reduce_mfma(D, blockIdx.x);
}

// Hs = [diagonal block](Hpp + lambda I) - sum over chunks (in chunk order), bs = bp - sum: one thread per entry of the packed
// lower triangle, then one per entry of the right-hand side
__device__ __forceinline__ void b_schur_reduce(const LbaDev& D, const int bx) {
  const int F = D.n_free, n = 6 * F, ntri = n * (n + 1) / 2, npairs = F * (F + 1) / 2;
  const int k = bx * kMk + threadIdx.x;
  if (k < ntri) {
    int r = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > k) r--;
    while ((r + 1) * (r + 2) / 2 <= k) r++;
    const int c = k - r * (r + 1) / 2;
    const int i1 = r / 6, a = r - 6 * i1, i2 = c / 6, cc = c - 6 * i2;
    const double* part = D.schur_part + ((size_t)(i1 * (i1 + 1) / 2 + i2)) * 36 + 6 * a + cc;
    double sum = 0;
    for (int ch = 0; ch < D.n_schur_chunks; ch++) sum += part[(size_t)ch * npairs * 36];
    if (i1 != i2) {
      D.Hs[k] = -sum;
    } else {  // a >= cc: Hpp's upper-triangle entry (cc, a)
      const double hpp = D.Hpp[21 * i1 + (cc * 6 - cc * (cc - 1) / 2 + (a - cc))];
      D.Hs[k] = hpp + (a == cc ? D.S->lambda : 0.0) - sum;
    }
  } else if (k < ntri + n) {
    const int j = k - ntri;
    double sum = 0;
    for (int ch = 0; ch < D.n_schur_chunks; ch++) sum += D.schur_part_b[(size_t)ch * n + j];
    D.bs[j] = D.bp[j] - sum;
  }
}
__global__ __launch_bounds__(kMk) void k_lba_schur_reduce(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_schur_reduce(D, blockIdx.x);
}



// LDL^T + triangular solves of the reduced pose system by a single workgroup.  kLds: the packed triangle fits the 160 KB of
// LDS (n <= 180, i.e. up to 30 free poses) and is factored there.  Otherwise (larger windows) it is factored in place in HBM /
// L2; only the right-hand side and the current 6-column panel are held in LDS, so the trailing update reads and writes each
// element once.  Same arithmetic, same order of operations per entry in both variants.
// The reduced system in LDS (n <= 180): LDL^T by 6-wide block columns with the pivots' reciprocals and the scaled panel
// P = L D kept beside L, so that a panel row costs 15 fused multiply-adds + 6 products instead of 6 dependent divisions and the
// trailing update 6 fused multiply-adds an entry; the forward substitution rides along with the factorisation (the right-hand
// side is one more row of every panel), and the backward substitution is done by a single wave (n values, no workgroup
// barriers).  The reference solves this system with a sparse Cholesky (g2o LinearSolverEigen): there is no operation order to
// follow here, only a fixed one to keep.
__device__ __forceinline__ void b_solve_lds(const LbaDev& D) {
  extern __shared__ __align__(16) double lds[];
  __shared__ int s_flag;
  __shared__ double s_w[6][6], s_y[6];
  const int n = 6 * D.n_free;
  double* Hs = lds;                               // packed lower triangle, L in place
  double* bs = Hs + (size_t)n * (n + 1) / 2;      // [n]
  double* invd = bs + n;                          // [n] 1 / d
  double* pan = invd + n;                         // [n][6] P = L D of the current block column
  // (eight loads in flight: as a plain loop the 7 trips of a 120-row system are 7 dependent round trips)
  gfs::strided_batch<8>(D.Hs, (int)threadIdx.x, kThreads, n * (n + 1) / 2, [&](int k, double v) { Hs[k] = v; });
  for (int k = threadIdx.x; k < n; k += kThreads) bs[k] = D.bs[k];
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  bool ok = true;
  for (int jb = 0; jb < n; jb += 6) {
    if (threadIdx.x == 0) {  // (a) the 6x6 diagonal block and the block's share of the forward substitution, on one lane
      double a[6][6], y[6];
#pragma unroll
      for (int i = 0; i < 6; i++) {
        y[i] = bs[jb + i];
#pragma unroll
        for (int k = 0; k <= i; k++) a[i][k] = Hs[tri(jb + i, jb + k)];
      }
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const double d = a[c][c];
        if (d == 0.0 || !isfinite(d)) bad = true;
        const double r = bad ? 0.0 : 1.0 / d;
        invd[jb + c] = r;
#pragma unroll
        for (int i = c + 1; i < 6; i++) {
          const double p = a[i][c];  // = L(i,c) d
          s_w[i][c] = p;
          a[i][c] = p * r;
        }
#pragma unroll
        for (int k = c + 1; k < 6; k++)
#pragma unroll
          for (int i = k; i < 6; i++) a[i][k] = fma(-a[i][c], s_w[k][c], a[i][k]);
      }
      if (bad) s_flag = 1;
#pragma unroll
      for (int c = 0; c < 6; c++)
#pragma unroll
        for (int c2 = 0; c2 < c; c2++) y[c] = fma(-a[c][c2], y[c2], y[c]);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        bs[jb + i] = y[i];
        s_y[i] = y[i];
#pragma unroll
        for (int k = 0; k <= i; k++) Hs[tri(jb + i, jb + k)] = a[i][k];
      }
    }
    __syncthreads();
    if (s_flag) {
      ok = false;
      break;
    }
    for (int i = jb + 6 + threadIdx.x; i < n; i += kThreads) {  // (b) panel rows: P(i,c) and L(i,c) = P(i,c) / d_c, and bs[i]
      double l[6], acc = bs[i];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        double v = Hs[tri(i, jb + c)];
#pragma unroll
        for (int c2 = 0; c2 < c; c2++) v = fma(-l[c2], s_w[c][c2], v);
        pan[6 * (i - jb - 6) + c] = v;
        l[c] = v * invd[jb + c];
        Hs[tri(i, jb + c)] = l[c];
        acc = fma(-l[c], s_y[c], acc);
      }
      bs[i] = acc;
    }
    __syncthreads();
    const int m = n - jb - 6;  // (c) trailing size
    for (int k = threadIdx.x; k < m * (m + 1) / 2; k += kThreads) {
      int r = (int)((sqrtf(8.0f * (float)k + 1.0f) - 1.0f) * 0.5f);  // (a guess: the two loops settle it)
      while (r * (r + 1) / 2 > k) r--;
      while ((r + 1) * (r + 2) / 2 <= k) r++;
      const int c = k - r * (r + 1) / 2;
      const int ii = jb + 6 + r, kk = jb + 6 + c;
      double v = Hs[tri(ii, kk)];
#pragma unroll
      for (int c2 = 0; c2 < 6; c2++) v = fma(-pan[6 * r + c2], Hs[tri(kk, jb + c2)], v);
      Hs[tri(ii, kk)] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) D.S->solve_ok = ok ? 1 : 0;
  if (!ok || threadIdx.x >= 64) return;
  // one wave from here on: z = D^-1 y, then L^T x = z from the last block up (LDS accesses of one wave keep their order)
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += 64) bs[i] *= invd[i];
  for (int jb = n - 6; jb >= 0; jb -= 6) {
    if (lane == 0) {
      double y[6], l[6][6];
#pragma unroll
      for (int c = 0; c < 6; c++) {
        y[c] = bs[jb + c];
#pragma unroll
        for (int c2 = c + 1; c2 < 6; c2++) l[c2][c] = Hs[tri(jb + c2, jb + c)];
      }
#pragma unroll
      for (int c = 5; c >= 0; c--)
#pragma unroll
        for (int c2 = 5; c2 > c; c2--) y[c] = fma(-l[c2][c], y[c2], y[c]);
#pragma unroll
      for (int c = 0; c < 6; c++) bs[jb + c] = y[c];
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < jb; i += 64) {
      double v = bs[i];
#pragma unroll
      for (int c = 5; c >= 0; c--) v = fma(-Hs[tri(jb + c, i)], bs[jb + c], v);
      bs[i] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
  for (int i = lane; i < n; i += 64) D.xp[i] = bs[i];
}

template <bool kLds>
__device__ __forceinline__ void b_solve(const LbaDev& D) {
  extern __shared__ __align__(16) double lds[];
  __shared__ int s_flag;
  const int n = 6 * D.n_free;
  double* Hs = kLds ? lds : D.Hs;
  double* bs = kLds ? lds + (size_t)n * (n + 1) / 2 : lds;
  double* pan = lds + n;  // !kLds only: panel rows [n][6] followed by the 6 pivots
  if (kLds) gfs::strided_batch<8>(D.Hs, (int)threadIdx.x, kThreads, n * (n + 1) / 2, [&](int k, double v) { Hs[k] = v; });
  for (int k = threadIdx.x; k < n; k += kThreads) bs[k] = D.bs[k];
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  // LDL^T by 6-wide block columns (n = 6 n_free): every entry sees the same subtraction sequence as in the column-by-column
  // form, but a block step costs 3 barriers instead of 18:
  //   (a) the 6x6 diagonal block is factored by one lane, (b) every row below solves its 6 entries against it (rows are
  //   independent), (c) the trailing triangle takes the rank-6 update.
  bool ok = true;
  for (int jb = 0; jb < n && ok; jb += 6) {
    if (threadIdx.x == 0) {  // the 21 entries are pulled into registers first: one LDS latency instead of ~100 dependent ones
      double a[6][6];
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k <= i; k++) a[i][k] = Hs[tri(jb + i, jb + k)];
      bool bad = false;
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const double d = a[c][c];
        if (d == 0.0 || !isfinite(d)) bad = true;
        if (!bad) {
#pragma unroll
          for (int i = c + 1; i < 6; i++) a[i][c] /= d;
#pragma unroll
          for (int k = c + 1; k < 6; k++)
#pragma unroll
            for (int i = k; i < 6; i++) a[i][k] -= a[i][c] * a[k][c] * d;
        }
      }
      if (bad) s_flag = 1;
#pragma unroll
      for (int i = 0; i < 6; i++)
#pragma unroll
        for (int k = 0; k <= i; k++) Hs[tri(jb + i, jb + k)] = a[i][k];
    }
    __syncthreads();
    if (s_flag) {
      ok = false;
      break;
    }
    for (int i = jb + 6 + threadIdx.x; i < n; i += kThreads) {  // (b) panel rows
      for (int c = 0; c < 6; c++) {
        double v = Hs[tri(i, jb + c)];
        for (int c2 = 0; c2 < c; c2++) v -= Hs[tri(i, jb + c2)] * Hs[tri(jb + c, jb + c2)] * Hs[tri(jb + c2, jb + c2)];
        Hs[tri(i, jb + c)] = v / Hs[tri(jb + c, jb + c)];
      }
    }
    __syncthreads();
    const int m = n - jb - 6;  // (c) trailing size
    if (!kLds) {               // panel and pivots into LDS: the update then touches HBM once per element
      for (int k = threadIdx.x; k < 6 * m; k += kThreads) pan[k] = Hs[tri(jb + 6 + k / 6, jb + k % 6)];
      if (threadIdx.x < 6) pan[6 * m + threadIdx.x] = Hs[tri(jb + threadIdx.x, jb + threadIdx.x)];
      __syncthreads();
    }
    for (int k = threadIdx.x; k < m * (m + 1) / 2; k += kThreads) {
      int r = (int)((sqrt(8.0 * k + 1.0) - 1.0) * 0.5);
      while (r * (r + 1) / 2 > k) r--;
      while ((r + 1) * (r + 2) / 2 <= k) r++;
      const int c = k - r * (r + 1) / 2;
      const int ii = jb + 6 + r, kk = jb + 6 + c;
      double v = Hs[tri(ii, kk)];
      if (kLds) {
        for (int c2 = 0; c2 < 6; c2++) v -= Hs[tri(ii, jb + c2)] * Hs[tri(kk, jb + c2)] * Hs[tri(jb + c2, jb + c2)];
      } else {
        for (int c2 = 0; c2 < 6; c2++) v -= pan[6 * r + c2] * pan[6 * c + c2] * pan[6 * m + c2];
      }
      Hs[tri(ii, kk)] = v;
    }
    __syncthreads();
  }
  if (ok) {
    // forward substitution (unit lower), block by block: the 6 unknowns of a block on one lane, then all rows below
    for (int jb = 0; jb < n; jb += 6) {
      if (threadIdx.x == 0) {
        double y[6], l[6][6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          y[c] = bs[jb + c];
#pragma unroll
          for (int c2 = 0; c2 < c; c2++) l[c][c2] = Hs[tri(jb + c, jb + c2)];
        }
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
          for (int c2 = 0; c2 < c; c2++) y[c] -= l[c][c2] * y[c2];
#pragma unroll
        for (int c = 0; c < 6; c++) bs[jb + c] = y[c];
      }
      __syncthreads();
      for (int i = jb + 6 + threadIdx.x; i < n; i += kThreads) {
        double v = bs[i];
        for (int c = 0; c < 6; c++) v -= Hs[tri(i, jb + c)] * bs[jb + c];
        bs[i] = v;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += kThreads) bs[i] /= Hs[tri(i, i)];
    __syncthreads();
    // backward substitution with L^T, from the last block up
    for (int jb = n - 6; jb >= 0; jb -= 6) {
      if (threadIdx.x == 0) {
        double y[6], l[6][6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
          y[c] = bs[jb + c];
#pragma unroll
          for (int c2 = c + 1; c2 < 6; c2++) l[c2][c] = Hs[tri(jb + c2, jb + c)];
        }
#pragma unroll
        for (int c = 5; c >= 0; c--)
#pragma unroll
          for (int c2 = 5; c2 > c; c2--) y[c] -= l[c2][c] * y[c2];
#pragma unroll
        for (int c = 0; c < 6; c++) bs[jb + c] = y[c];
      }
      __syncthreads();
      for (int i = threadIdx.x; i < jb; i += kThreads) {
        double v = bs[i];
        for (int c = 5; c >= 0; c--) v -= Hs[tri(jb + c, i)] * bs[jb + c];
        bs[i] = v;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += kThreads) D.xp[i] = bs[i];
  }
  if (threadIdx.x == 0) D.S->solve_ok = ok ? 1 : 0;
}
template <bool kLds>
__global__ __launch_bounds__(kThreads) void k_lba_solve(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  if (kLds)
    b_solve_lds(D);
  else
    b_solve<false>(D);
}


// landmark back-substitution, update of the trial estimate, computeScale partial sums
__device__ __forceinline__ void b_update(const LbaDev& D, const int bx, const int gdx) {
  __shared__ double s4[4];
  const LbaState& S = *D.S;
  double loc = 0;
  if (S.solve_ok) {
    const int cur = S.cur;
    const double *q = sel(D.q, D.q_try, cur), *t = sel(D.t, D.t_try, cur), *X = sel(D.X, D.X_try, cur);
    double *qn = sel(D.q, D.q_try, cur ^ 1), *tn = sel(D.t, D.t_try, cur ^ 1), *Xn = sel(D.X, D.X_try, cur ^ 1);
    const double lambda = S.lambda;
    const int l = bx * kMk + threadIdx.x;
    if (l < D.n_points) {
      double cl[3] = {D.bl[3 * (size_t)l], D.bl[3 * (size_t)l + 1], D.bl[3 * (size_t)l + 2]};
      // (edges in order, the same subtractions as a plain loop; but a plain loop is one chain of three dependent round trips an
      //  edge -- pose index, free index, block -- for ~16 edges a landmark: four edges' indices and two edges' blocks are asked for
      //  together, round 5)
      const int e_end = D.pt_begin[l + 1];
      for (int eb = D.pt_begin[l]; eb < e_end; eb += 4) {
        int ep[4], f[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ep[u] = D.e_pose[min(eb + u, e_end - 1)];
#pragma unroll
        for (int u = 0; u < 4; u++) f[u] = eb + u < e_end ? D.free_index[ep[u]] : -1;
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
          double Bv[2][18], xv[2][6];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int e = min(eb + 2 * h2 + u, e_end - 1), ff = max(f[2 * h2 + u], 0);
#pragma unroll
            for (int k = 0; k < 18; k++) Bv[u][k] = D.Hpl[18 * (size_t)e + k];
#pragma unroll
            for (int a = 0; a < 6; a++) xv[u][a] = D.xp[6 * ff + a];
          }
#pragma unroll
          for (int u = 0; u < 2; u++) {
            if (f[2 * h2 + u] < 0) continue;
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
              for (int a = 0; a < 6; a++) cl[c] -= Bv[u][3 * a + c] * xv[u][a];
          }
        }
      }
      const double* Di = D.Dinv + 6 * (size_t)l;
      const double xl[3] = {Di[0] * cl[0] + Di[1] * cl[1] + Di[2] * cl[2], Di[1] * cl[0] + Di[3] * cl[1] + Di[4] * cl[2],
                            Di[2] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2]};
      for (int c = 0; c < 3; c++) {
        D.xl[3 * (size_t)l + c] = xl[c];
        Xn[3 * (size_t)l + c] = X[3 * (size_t)l + c] + xl[c];
        loc += xl[c] * (lambda * xl[c] + D.bl[3 * (size_t)l + c]);
      }
    }
    if (bx == 0) {
      for (int f = threadIdx.x; f < D.n_free; f += kMk) {
        const int p = D.free_pose[f];
        pose_oplus(q + 4 * p, t + 3 * p, D.xp + 6 * f, qn + 4 * p, tn + 3 * p);
        for (int a = 0; a < 6; a++) loc += D.xp[6 * f + a] * (lambda * D.xp[6 * f + a] + D.bp[6 * f + a]);
      }
    }
  }
  const double tot = block_sum256(loc, s4);
  if (threadIdx.x == 0) D.part_scale[bx] = tot;
}
__global__ __launch_bounds__(kMk) void k_lba_update(LbaDev D, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_update(D, blockIdx.x, gridDim.x);
}


// end of an LM trial (and, when the trial loop ends, of the iteration): rho test, lambda update, termination tests
__device__ __forceinline__ void b_decide(const LbaDev& D, const int bx, int force_end, int* __restrict__ host_flags,
                                         double* __restrict__ snap = nullptr) {
  if (threadIdx.x != 0 || bx != 0) return;
  LbaState& S = *D.S;
  if (!force_end) {
    double temp_chi = 0, scale = 0;
    for (int b = 0; b < D.n_err_blocks; b++) temp_chi += D.part_chi[b];
    for (int b = 0; b < D.n_upd_blocks; b++) scale += D.part_scale[b];
    if (!S.solve_ok) {
      temp_chi = 1.79769313486231570e308;
      scale = 0;
    }
    S.temp_chi = temp_chi;
    S.rho = (S.current_chi - temp_chi) / (scale + 1e-3);
    if (S.rho > 0 && isfinite(temp_chi)) {
      double alpha = 1. - gfs_glibc::pow3(2 * S.rho - 1);
      alpha = fmin(alpha, 2. / 3.);
      S.lambda *= fmax(1. / 3., alpha);
      S.ni = 2;
      S.current_chi = temp_chi;
      S.cur ^= 1;  // discardTop: the trial becomes the estimate
      S.err_at_cur = 1;
    } else {
      S.lambda *= S.ni;
      S.ni *= 2;
      S.err_at_cur = 0;  // (the arrays hold the rejected trial's values)
    }
    S.qmax++;
    S.again = (S.rho < 0 && S.qmax < 10) ? 1 : 0;
  } else {
    S.again = 0;  // the stop flag ended the trial loop
    S.err_at_cur = 0;
  }
  S.terminate = 0;
  if (!S.again) {
    S.iters++;
    S.last_chi = S.current_chi;
    if (S.qmax == 10 || S.rho == 0) {
      S.terminate = 1;
    } else {
      if ((S.ini_chi - S.current_chi) * 1e3 < S.ini_chi)
        S.n_bad++;
      else
        S.n_bad = 0;
      if (S.n_bad >= 3) S.terminate = 1;
    }
  }
  S.spec_ok = (!S.again && !S.terminate && !force_end) ? 1 : 0;
  host_flags[0] = S.again;
  host_flags[1] = S.terminate;
  host_flags[2] = S.cur;
  host_flags[3] = S.iters;
  if (snap) {  // what k_lba_finish reports if the loop ends HERE (gfs_lba_solve: an iteration running ahead of a stop flag is discarded)
    snap[0] = S.lambda;
    snap[1] = S.last_chi;
  }
}
// host_out: one slot of gfs_lba::h_flags -- {again, terminate, cur, iters} and, 16 bytes on, {lambda, last_chi}
__global__ void k_lba_decide(LbaDev D, int force_end, int* __restrict__ host_out, int gate) {
  if (gate && !D.S->spec_ok) return;
  b_decide(D, blockIdx.x, force_end, host_out, reinterpret_cast<double*>(host_out + 4));
}
// the state k_lba_decide left behind an earlier iteration, put back (an iteration that ran ahead of the caller's stop flag is discarded:
// its trial only wrote the OTHER estimate buffer); the errors are evaluated again at that estimate by the caller
__global__ void k_lba_restore(LbaDev D, int cur, int iters, double lambda, double last_chi) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  LbaState& S = *D.S;
  S.cur = cur;
  S.iters = iters;
  S.lambda = lambda;
  S.last_chi = last_chi;
  S.err_at_cur = 0;
  S.spec_ok = 0;
}


__device__ __forceinline__ void b_finish(const LbaDev& D, const int bx) {
  if (threadIdx.x != 0 || bx != 0) return;
  const LbaState& S = *D.S;
  D.out_info[0] = S.iters;
  D.out_info[1] = S.cur;
  D.out_stats[0] = D.mode == 1 ? S.current_chi : S.last_chi;
  D.out_stats[1] = D.mode == 1 ? 0.0 : S.lambda;
}
__global__ void k_lba_finish(LbaDev D) { b_finish(D, blockIdx.x); }

// the results of a window gathered into one block (one copy to the host instead of six): layout of LbaDev::out_pack
__device__ __forceinline__ void b_pack(const LbaDev& D, const int bx, const int gdx) {
  const int cur = D.out_info[1];  // which estimate buffer holds the accepted state (b_finish; 0 on the single-workgroup path)
  const double *q = sel(D.q, D.q_try, cur), *t = sel(D.t, D.t_try, cur), *X = sel(D.X, D.X_try, cur);
  const int E = D.n_edges, NQ = D.n_poses, NP = D.n_points;
  double* o = D.out_pack;
  const int g = bx * kMk + threadIdx.x, G = gdx * kMk;
  for (int i = g; i < E; i += G) o[i] = D.chi2[i];
  o += E;
  for (int i = g; i < 4 * NQ; i += G) o[i] = q[i];
  o += 4 * NQ;
  for (int i = g; i < 3 * NQ; i += G) o[i] = t[i];
  o += 3 * NQ;
  for (int i = g; i < 3 * NP; i += G) o[i] = X[i];
  o += 3 * NP;
  if (g == 0) {
    o[0] = D.out_stats[0];
    o[1] = D.out_stats[1];
    o[2] = (double)D.out_info[0];
    o[3] = (double)cur;
  }
}
__global__ __launch_bounds__(kMk) void k_lba_pack(LbaDev D) { b_pack(D, blockIdx.x, gridDim.x); }


// ------------------------------------------------------------------------------------------------
// Batched windows (gfs_lba_solve_batch): the same phase kernels with the window index in blockIdx.y.  Every window carries its
// own LM state machine (LbaState::phase: 0 = build the system next, 1 = a trial step next, 2 = done); the host launches
// "rounds" of [build group][trial group] over all windows, a kernel leaves at once when its window is in another phase or its
// block index is beyond that window's size.  Same bodies, same arithmetic: a window's result does not depend on the batch.
// ------------------------------------------------------------------------------------------------
#define GFS_LBAB_PROLOGUE(PHASE, NEED)                         \
  const LbaDev D = DD[blockIdx.y];                             \
  if (D.S->phase != (PHASE)) return;                           \
  const int need = (NEED);                                     \
  if ((int)blockIdx.x >= need) return;

__global__ __launch_bounds__(kMk) void kb_lba_init(const LbaDev* __restrict__ DD) {
  const LbaDev D = DD[blockIdx.y];
  b_init(D, blockIdx.x, gridDim.x);
}
__global__ __launch_bounds__(kMk) void kb_lba_errors(const LbaDev* __restrict__ DD, int trial) {
  GFS_LBAB_PROLOGUE(trial ? 1 : 0, D.n_err_blocks)
  b_errors(D, blockIdx.x, need, trial);
}
__global__ __launch_bounds__(kMk) void kb_lba_build_landmarks(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(0, D.n_lm_wg)
  b_build_landmarks(D, blockIdx.x, need);
}
__global__ __launch_bounds__(kPoseWg) void kb_lba_build_poses(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(0, D.n_free)
  b_build_poses(D, blockIdx.x, need);
}
__global__ __launch_bounds__(kThreads) void kb_lba_begin(const LbaDev* __restrict__ DD, int* __restrict__ n_done) {
  GFS_LBAB_PROLOGUE(0, 1)
  b_begin(D, blockIdx.x, need, D.S->it);
  __syncthreads();
  if (threadIdx.x == 0) {
    LbaState& S = *D.S;
    if (D.iterations <= 0) {
      // optimize(0): the errors have been evaluated, nothing else happens to this window (what gfs_lba_solve reports for it:
      // the chi2 of the initial estimate, lambda 0, no iteration)
      S.last_chi = S.current_chi;
      S.lambda = 0;
      S.phase = 2;
      atomicAdd(n_done, 1);
    } else {
      S.phase = 1;  // the build group is complete for this window: trials follow
    }
  }
}
__global__ __launch_bounds__(kMk) void kb_lba_dinv(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, D.n_upd_blocks)
  b_dinv(D, blockIdx.x, need);
}
__global__ __launch_bounds__(kMk) void kb_lba_schur(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, (D.schur_sub || D.schur_mfma) ? 0 : D.n_free * (D.n_free + 1) / 2)
  b_schur(D, blockIdx.x, need);
}
__global__ __launch_bounds__(kMk) void kb_lba_schur_chunks(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, D.schur_sub ? D.n_schur_chunks * D.n_pair_tiles : 0)
  b_schur_chunks(D, blockIdx.x);
}
__global__ __launch_bounds__(kMk) void kb_lba_schur_reduce(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, D.schur_sub ? (6 * D.n_free * (6 * D.n_free + 1) / 2 + 6 * D.n_free + kMk - 1) / kMk : 0)
  b_schur_reduce(D, blockIdx.x);
}
template <bool kOneBlock>
__global__ __launch_bounds__(kMk, kOneBlock ? 2 : 1) void kb_lba_schur_mfma(const LbaDev* __restrict__ DD) {
  // every window takes the instance gfs_lba_solve would launch for it (one 128 x 128 block of pose pairs or several), whatever
  // else is in the batch: the host launches both instances when the batch mixes the two kinds
  GFS_LBAB_PROLOGUE(1, D.schur_mfma && (D.n_pair_tiles == 1) == kOneBlock ? D.n_schur_chunks * D.n_pair_tiles : 0)
  b_schur_mfma<kOneBlock>(D, blockIdx.x);
}
__global__ __launch_bounds__(kMk) void kb_lba_schur_reduce_mfma(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, D.schur_mfma ? (6 * D.n_free * (6 * D.n_free + 1) / 2 + 6 * D.n_free + kMk - 1) / kMk : 0)
  b_schur_reduce_mfma(D, blockIdx.x);
}
template <bool kLds>
__global__ __launch_bounds__(kThreads) void kb_lba_solve(const LbaDev* __restrict__ DD) {
  // every window takes the variant gfs_lba_solve would give it (the two differ in arithmetic), whatever else is in the batch
  GFS_LBAB_PROLOGUE(1, (D.n_free <= kMaxFreeLds) == kLds ? 1 : 0)
  if (kLds)
    b_solve_lds(D);
  else
    b_solve<false>(D);
}
__global__ __launch_bounds__(kMk) void kb_lba_update(const LbaDev* __restrict__ DD) {
  GFS_LBAB_PROLOGUE(1, D.n_upd_blocks)
  b_update(D, blockIdx.x, need);
}
// end of a trial: the scalar LM logic of k_lba_decide, then the window's next phase (another trial, the next iteration's build,
// or done: optimize(iterations) ran out or the algorithm terminated)
__global__ void kb_lba_decide(const LbaDev* __restrict__ DD, int force_end, int* __restrict__ flags_all, int* __restrict__ n_done) {
  const LbaDev D = DD[blockIdx.x];
  if (threadIdx.x != 0 || D.S->phase != 1) return;
  b_decide(D, 0, force_end, flags_all + 4 * blockIdx.x);
  LbaState& S = *D.S;
  if (S.again) return;  // phase stays 1
  S.it++;
  if (S.terminate || S.it >= D.iterations || force_end) {
    S.phase = 2;
    atomicAdd(n_done, 1);
  } else {
    S.phase = 0;
  }
}
__global__ void kb_lba_finish(const LbaDev* __restrict__ DD) {
  const LbaDev D = DD[blockIdx.x];
  b_finish(D, 0);
}
__global__ __launch_bounds__(kMk) void kb_lba_pack(const LbaDev* __restrict__ DD) {
  const LbaDev D = DD[blockIdx.y];
  b_pack(D, blockIdx.x, gridDim.x);
}
#undef GFS_LBAB_PROLOGUE

// host -> device through the pinned arena (bump allocation; the arena outlives the asynchronous copies of one call)
}  // namespace

struct gfs_lba_batch;
extern "C" void gfs_lba_batch_destroy(gfs_lba_batch* b);
struct gfs_lba {
  gfs_lba_batch* self_batch = nullptr;  // one-window wrapper: gfs_lba_solve runs through the batched (device-driven) LM loop
  int device, max_poses, max_points, max_edges;
  hipStream_t stream;
  std::mutex mu;
  int* h_stop = nullptr;  // host-mapped
  int* h_flags = nullptr;  // host-mapped, two slots of kFlagInts ints: {again, terminate, cur, iters | lambda, last_chi (doubles)} written by k_lba_decide
  hipEvent_t ev_decide[2] = {nullptr, nullptr};  // behind decide i: event i & 1 -- the host waits for THIS, not for what is queued behind it
  LbaDev last_desc;        // the window of the last run()
  gfs::DevBuf<LbaState> d_state;
  gfs::PinBuf<unsigned char> h_stage;  // pinned staging arena for the problem upload (pageable copies cost ~2 ms each)
  gfs::DevBuf<double> d_part_chi, d_part_scale, d_Hs, d_bs;
  gfs::DevBuf<double> d_schur_part, d_schur_part_b;  // grown on demand (upload_and_fill)
  gfs::DevBuf<unsigned char> d_in;      // the window's inputs, same layout as h_stage (Stager)
  gfs::DevBuf<double> d_out;            // the window's results (b_pack)
  gfs::PinBuf<double> h_out;
  gfs::DevBuf<double> d_q, d_t, d_X, d_qt, d_tt, d_Xt, d_chi2, d_err, d_Hpl, d_Hll, d_bl, d_Dinv, d_Hpp, d_bp, d_xl, d_xp, d_stats;
  gfs::DevBuf<int> d_info;
};

namespace {

// Host preparation of a window, written straight into the handle's pinned arena (the device block d_in has the same layout, one
// copy moves it): landmark-major edge order, the CSR tables, the co-visibility table.
struct HostPrep {
  std::vector<int> order;  // landmark-major position -> original edge
  int n_free = 0;
  double *q0 = nullptr, *t0 = nullptr, *X0 = nullptr, *obs = nullptr, *w = nullptr;
  int *free_index = nullptr, *free_pose = nullptr, *e_pose = nullptr, *e_point = nullptr, *pt_begin = nullptr, *pose_begin = nullptr,
      *pose_edges = nullptr, *edge_of = nullptr;
  unsigned char* stereo = nullptr;
  int* e_dup = nullptr;  // NULL when the window has no duplicate (pose, landmark) edges
  int* lm_wg = nullptr;  // landmark ranges of b_build_landmarks' workgroups
  int n_lm_wg = 0;
  size_t used = 0;  // bytes of the arena in use
};

int prepare(gfs_lba* h, const gfs_lba_problem* p, HostPrep& P) {
  GFS_REQUIRE(p && p->n_poses >= 0 && p->n_points >= 0 && p->n_edges >= 0, GFS_ERR_INVALID_ARG, "gfs_lba: invalid problem");
  GFS_REQUIRE(p->n_poses <= h->max_poses && p->n_points <= h->max_points && p->n_edges <= h->max_edges, GFS_ERR_CAPACITY,
              "gfs_lba: problem (%d poses, %d points, %d edges) exceeds handle capacity (%d, %d, %d)", p->n_poses,
              p->n_points, p->n_edges, h->max_poses, h->max_points, h->max_edges);
  const int E = p->n_edges, NP = p->n_points, NQ = p->n_poses;
  int nf = 0;
  for (int i = 0; i < NQ; i++) nf += p->pose_fixed[i] ? 0 : 1;
  P.n_free = nf;
  {  // carve the arena
    unsigned char* base = h->h_stage.p;
    size_t at = 0;
    auto take = [&](size_t bytes) {
      unsigned char* r = base + at;
      at = (at + bytes + 63) & ~(size_t)63;
      return r;
    };
    P.q0 = (double*)take((size_t)NQ * 32);
    P.t0 = (double*)take((size_t)NQ * 24);
    P.X0 = (double*)take((size_t)NP * 24);
    P.obs = (double*)take((size_t)E * 24);
    P.w = (double*)take((size_t)E * 8);
    P.free_index = (int*)take((size_t)NQ * 4);
    P.free_pose = (int*)take((size_t)nf * 4);
    P.e_pose = (int*)take((size_t)E * 4);
    P.e_point = (int*)take((size_t)E * 4);
    P.pt_begin = (int*)take((size_t)(NP + 1) * 4);
    P.pose_begin = (int*)take((size_t)(nf + 1) * 4);
    P.pose_edges = (int*)take((size_t)E * 4);
    P.edge_of = (int*)take((size_t)nf * NP * 4);
    P.stereo = take((size_t)E);
    P.lm_wg = (int*)take((size_t)(NP + 2) * 4);
    P.used = at;  // (grows by the duplicate table below when the window has any)
    P.e_dup = (int*)take((size_t)E * 4);
    GFS_REQUIRE(at <= h->h_stage.n, GFS_ERR_CAPACITY, "gfs_lba: staging arena too small");
  }
  bool any_dup = false;
  if (NQ) memcpy(P.q0, p->pose_q, (size_t)NQ * 32);
  if (NQ) memcpy(P.t0, p->pose_t, (size_t)NQ * 24);
  if (NP) memcpy(P.X0, p->points, (size_t)NP * 24);
  nf = 0;
  for (int i = 0; i < NQ; i++) {
    P.free_index[i] = -1;
    if (!p->pose_fixed[i]) {
      P.free_index[i] = nf;
      P.free_pose[nf++] = i;
    }
  }
  std::fill(P.pt_begin, P.pt_begin + NP + 1, 0);
  for (int e = 0; e < E; e++) {
    GFS_REQUIRE(p->edge_point[e] >= 0 && p->edge_point[e] < NP && p->edge_pose[e] >= 0 && p->edge_pose[e] < NQ,
                GFS_ERR_INVALID_ARG, "gfs_lba: edge %d references an unknown vertex", e);
    P.pt_begin[p->edge_point[e] + 1]++;
  }
  for (int l = 0; l < NP; l++) P.pt_begin[l + 1] += P.pt_begin[l];
  {  // whole landmarks packed into workgroups of at most kMk edges (b_build_landmarks)
    int nw = 0, l = 0;
    while (l < NP) {
      P.lm_wg[nw++] = l;
      const int base = P.pt_begin[l];
      int l2 = l + 1;
      while (l2 < NP && P.pt_begin[l2 + 1] - base <= kMk) l2++;
      l = l2;
    }
    P.lm_wg[nw] = NP;
    P.n_lm_wg = nw;
  }
  P.order.assign(E, 0);
  {
    std::vector<int> pos(P.pt_begin, P.pt_begin + NP);
    for (int e = 0; e < E; e++) P.order[pos[p->edge_point[e]]++] = e;  // stable
  }
  std::fill(P.pose_begin, P.pose_begin + P.n_free + 1, 0);
  std::fill(P.edge_of, P.edge_of + (size_t)P.n_free * NP, -1);
  for (int k = 0; k < E; k++) {
    const int e = P.order[k];
    P.e_pose[k] = p->edge_pose[e];
    P.e_point[k] = p->edge_point[e];
    for (int c = 0; c < 3; c++) P.obs[3 * (size_t)k + c] = p->edge_obs[3 * (size_t)e + c];
    P.w[k] = p->edge_inv_sigma2[e];
    P.stereo[k] = p->edge_stereo[e] ? 1 : 0;
    const int f = P.free_index[P.e_pose[k]];
    P.e_dup[k] = -1;
    if (f >= 0) {
      P.pose_begin[f + 1]++;
      int& slot = P.edge_of[(size_t)f * NP + P.e_point[k]];
      if (slot < 0) {
        slot = k;
      } else {  // a second edge between this pose and this landmark
        P.e_dup[k] = slot;
        any_dup = true;
      }
    }
  }
  if (any_dup)
    P.used = (size_t)((unsigned char*)(P.e_dup + E) - h->h_stage.p);
  else
    P.e_dup = nullptr;
  for (int f = 0; f < P.n_free; f++) P.pose_begin[f + 1] += P.pose_begin[f];
  {
    std::vector<int> pos(P.pose_begin, P.pose_begin + P.n_free);
    for (int k = 0; k < E; k++) {
      const int f = P.free_index[P.e_pose[k]];
      if (f >= 0) P.pose_edges[pos[f]++] = k;
    }
  }
  return GFS_OK;
}

// the arena of a prepared window -> the device block (one copy, on stream s)
int upload(gfs_lba* h, const HostPrep& P, hipStream_t s) {
  if (P.used) GFS_HIP(hipMemcpyAsync(h->d_in.p, h->h_stage.p, P.used, hipMemcpyHostToDevice, s));
  return GFS_OK;
}

// describes an uploaded window for the kernels
int upload_and_fill(gfs_lba* h, const gfs_lba_problem* p, const HostPrep& P, int mode, hipStream_t s, LbaDev& D, bool do_upload = true) {
  const int E = p->n_edges, NP = p->n_points;
  int rc;
  if (do_upload && (rc = upload(h, P, s))) return rc;
  D = LbaDev{};
  auto dev = [&](const void* host) { return (const void*)(h->d_in.p + ((const unsigned char*)host - h->h_stage.p)); };
  D.pose_q0 = (decltype(D.pose_q0))((const double*)dev(P.q0));
  D.pose_t0 = (decltype(D.pose_t0))((const double*)dev(P.t0));
  D.points0 = (decltype(D.points0))((const double*)dev(P.X0));
  D.free_index = (decltype(D.free_index))((const int*)dev(P.free_index));
  D.free_pose = (decltype(D.free_pose))((const int*)dev(P.free_pose));
  D.e_pose = (decltype(D.e_pose))((const int*)dev(P.e_pose));
  D.e_point = (decltype(D.e_point))((const int*)dev(P.e_point));
  D.e_obs = (decltype(D.e_obs))((const double*)dev(P.obs));
  D.e_w = (decltype(D.e_w))((const double*)dev(P.w));
  D.pt_begin = (decltype(D.pt_begin))((const int*)dev(P.pt_begin));
  D.pose_begin = (decltype(D.pose_begin))((const int*)dev(P.pose_begin));
  D.pose_edges = (decltype(D.pose_edges))((const int*)dev(P.pose_edges));
  D.edge_of = (decltype(D.edge_of))((const int*)dev(P.edge_of));
  D.e_stereo = (decltype(D.e_stereo))((const unsigned char*)dev(P.stereo));
  D.e_dup = (decltype(D.e_dup))(P.e_dup ? (const int*)dev(P.e_dup) : nullptr);
  D.lm_wg = (decltype(D.lm_wg))((const int*)dev(P.lm_wg));
  D.n_lm_wg = P.n_lm_wg;
  D.n_poses = p->n_poses;
  D.n_points = NP;
  D.n_edges = E;
  D.n_free = P.n_free;
  D.fx = p->fx;
  D.fy = p->fy;
  D.cx = p->cx;
  D.cy = p->cy;
  D.bf = p->bf;
  D.huber_mono = p->huber_mono;
  D.huber_stereo = p->huber_stereo;
  D.iterations = p->iterations;
  D.q = (decltype(D.q))(h->d_q.p);
  D.t = (decltype(D.t))(h->d_t.p);
  D.X = (decltype(D.X))(h->d_X.p);
  D.q_try = (decltype(D.q_try))(h->d_qt.p);
  D.t_try = (decltype(D.t_try))(h->d_tt.p);
  D.X_try = (decltype(D.X_try))(h->d_Xt.p);
  D.chi2 = (decltype(D.chi2))(h->d_chi2.p);
  D.err = (decltype(D.err))(h->d_err.p);
  D.Hpl = (decltype(D.Hpl))(h->d_Hpl.p);
  D.Hll = (decltype(D.Hll))(h->d_Hll.p);
  D.bl = (decltype(D.bl))(h->d_bl.p);
  D.Dinv = (decltype(D.Dinv))(h->d_Dinv.p);
  D.Hpp = (decltype(D.Hpp))(h->d_Hpp.p);
  D.bp = (decltype(D.bp))(h->d_bp.p);
  D.xl = (decltype(D.xl))(h->d_xl.p);
  D.xp = (decltype(D.xp))(h->d_xp.p);
  int* d_stop = nullptr;
  GFS_HIP(hipHostGetDevicePointer((void**)&d_stop, h->h_stop, 0));
  *h->h_stop = 0;
  D.stop = (decltype(D.stop))(d_stop);
  D.out_info = (decltype(D.out_info))(h->d_info.p);
  D.out_stats = (decltype(D.out_stats))(h->d_stats.p);
  D.out_pack = (decltype(D.out_pack))(h->d_out.p);
  D.mode = mode;
  D.S = (decltype(D.S))(h->d_state.p);
  D.part_chi = (decltype(D.part_chi))(h->d_part_chi.p);
  D.part_scale = (decltype(D.part_scale))(h->d_part_scale.p);
  D.Hs = (decltype(D.Hs))(h->d_Hs.p);
  D.bs = (decltype(D.bs))(h->d_bs.p);
  D.n_err_blocks = gfs::div_up(std::max(E, 1), kMk);
  D.n_upd_blocks = gfs::div_up(std::max(NP, 1), kMk);
  // Schur products: on the matrix cores by landmark chunks (b_schur_mfma), GFS_LBA_SCHUR=chunks | pairs select the vector kernels
  // (b_schur_chunks; one workgroup per pose pair, which also takes the windows too wide for the others' staging)
  {
    const int F = P.n_free;
    static const char* mode = getenv("GFS_LBA_SCHUR");
    static const bool by_pairs = mode && !strcmp(mode, "pairs"), by_chunks = mode && !strcmp(mode, "chunks");
    size_t need = 0, need_b = 0;
    if (F > 0 && !by_pairs && !by_chunks) {
      const size_t NB = (6 * (size_t)F + 1 + 127) / 128, ld = 128 * NB, chunks = gfs::div_up(std::max(NP, 1), kSchurMPts);
      need = chunks * ld * ld;
      if (need * sizeof(double) <= ((size_t)2 << 30) && p->n_poses <= 4096) {
        D.schur_mfma = 1;
        D.n_schur_chunks = (int)chunks;
        D.n_pair_tiles = (int)(NB * (NB + 1) / 2);
      }
    }
    if (F > 0 && !by_pairs && !D.schur_mfma) {
      const size_t per_landmark = (size_t)F * (kSchurSlot * sizeof(double) + sizeof(int)) + 4 * sizeof(double);
      const int sub = (int)std::min<size_t>(8, (96 * 1024) / per_landmark);
      const size_t npairs = (size_t)F * (F + 1) / 2, chunks = gfs::div_up(std::max(NP, 1), kSchurPts);
      need = chunks * npairs * 36;
      need_b = chunks * (size_t)F * 6;
      if (sub >= 1 && need * sizeof(double) <= ((size_t)1 << 30)) {
        D.schur_sub = sub;
        D.n_schur_chunks = (int)chunks;
        D.n_pair_tiles = (int)gfs::div_up((int)npairs, kMk);
      }
    }
    if (D.schur_sub || D.schur_mfma) {
      if (h->d_schur_part.n < need) {
        GFS_HIP(hipStreamSynchronize(s));
        if ((rc = h->d_schur_part.alloc(need + need / 4))) return rc;
      }
      if (h->d_schur_part_b.n < need_b) {
        GFS_HIP(hipStreamSynchronize(s));
        if ((rc = h->d_schur_part_b.alloc(need_b + need_b / 4))) return rc;
      }
      D.schur_part = (decltype(D.schur_part))(h->d_schur_part.p);
      D.schur_part_b = (decltype(D.schur_part_b))(h->d_schur_part_b.p);
    }
  }
  return GFS_OK;
}

inline size_t schur_mfma_lds_bytes(const LbaDev& D) {
  return (size_t)(2 * 3 * kSchurSub * kSchurLd + 9 * kSchurMPts) * sizeof(double) + (size_t)(kSchurMPts + 1 + D.n_poses) * sizeof(int);
}
inline size_t schur_lds_bytes(const LbaDev& D) {
  return (size_t)D.schur_sub * ((size_t)D.n_free * (kSchurSlot * sizeof(double) + sizeof(int)) + 4 * sizeof(double));
}

// The dynamic-LDS ceiling of a kernel is one global setting per function: raising it per call to that call's need would let
// two host threads with different windows undercut each other.  Set once per device to the hardware limit instead.
int lba_raise_lds_limits(int device) {
  static std::mutex mu;
  static bool done[64] = {};
  std::lock_guard<std::mutex> lk(mu);
  if (device >= 0 && device < 64 && done[device]) return GFS_OK;
  const int lim = 160 * 1024 - 1024;
  const void* fns[] = {(const void*)k_lba, (const void*)k_lba_solve<true>, (const void*)k_lba_solve<false>, (const void*)kb_lba_solve<true>,
                       (const void*)kb_lba_solve<false>, (const void*)k_lba_schur_chunks, (const void*)kb_lba_schur_chunks,
                       (const void*)k_lba_schur_mfma<true>, (const void*)k_lba_schur_mfma<false>, (const void*)kb_lba_schur_mfma<true>,
                       (const void*)kb_lba_schur_mfma<false>};
  for (const void* f : fns) GFS_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lim));
  if (device >= 0 && device < 64) done[device] = true;
  return GFS_OK;
}

// The caller's force-stop flag, read LIVE every time the host loop looks at it: an int (gfs_lba_solve) or the C++ bool the
// reference hands in (Optimizer::LocalBundleAdjustment's pbStopFlag = &mbAbortBA, one byte: gfs_lba_solve_bool).
struct StopFlag {
  volatile const int* i = nullptr;
  volatile const unsigned char* b = nullptr;
  explicit operator bool() const { return i || b; }
  bool operator*() const { return (i && *i) || (b && *b); }
};

int run(gfs_lba* h, const gfs_lba_problem* p, const HostPrep& P, int mode, StopFlag stop) {
  hipStream_t s = h->stream;
  const int E = p->n_edges;
  LbaDev D;
  int rc = upload_and_fill(h, p, P, mode, s, D);
  if (rc) return rc;
  const int n = 6 * P.n_free;
  const bool in_lds = P.n_free <= kMaxFreeLds;
  const size_t lds = (in_lds ? (size_t)n * (n + 1) / 2 + 8 * n + 8 : (size_t)7 * n + 8) * sizeof(double);
  GFS_REQUIRE(lds <= 160 * 1024, GFS_ERR_CAPACITY, "gfs_lba: %d free poses exceed the solver's workspace", P.n_free);
  static const bool single_wg = getenv("GFS_LBA_SINGLE_WG") != nullptr;  // the round-1a kernel: whole solve in one workgroup
  h->last_desc = D;
  if (single_wg && in_lds) {
    GFS_REQUIRE(!D.e_dup, GFS_ERR_UNSUPPORTED, "gfs_lba: GFS_LBA_SINGLE_WG does not take several edges between one pose and one point");
    if (E) GFS_HIP(hipMemsetAsync(h->d_Hpl.p, 0, (size_t)E * 18 * sizeof(double), s));
    GFS_LAUNCH("k_lba", k_lba, dim3(1), dim3(kThreads), lds, s, D);
    // setForceStopFlag semantics (src/Optimizer.cc:1679): relay the caller's flag to the device-visible one
    if (stop) {
      while (hipStreamQuery(s) == hipErrorNotReady) {
        if (*stop) *h->h_stop = 1;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
    }
    GFS_HIP(hipStreamSynchronize(s));
    return GFS_OK;
  }
  // ---- multi-kernel path: one launch per phase, the host walks the LM control flow from two flags per trial
  static const bool timing = getenv("GFS_LBA_TIMING") != nullptr;
  if (timing) GFS_HIP(hipStreamSynchronize(s));
  const auto Ta = std::chrono::steady_clock::now();
  int* d_flags = nullptr;
  GFS_HIP(hipHostGetDevicePointer((void**)&d_flags, h->h_flags, 0));
  const dim3 g_err(D.n_err_blocks), g_lm(std::max(D.n_lm_wg, 1)), g_upd(D.n_upd_blocks);
  GFS_LAUNCH("k_lba_init", k_lba_init, dim3(64), dim3(kMk), 0, s, D);
  const bool lin_only = mode == 1 || p->iterations <= 0;
  if (lin_only) {
    GFS_LAUNCH("k_lba_errors", k_lba_errors, g_err, dim3(kMk), 0, s, D, 0, 0);
    if (mode == 1) {
      if (D.n_lm_wg > 0) GFS_LAUNCH("k_lba_build_landmarks", k_lba_build_landmarks, g_lm, dim3(kMk), 0, s, D, 0);
      if (P.n_free > 0) GFS_LAUNCH("k_lba_build_poses", k_lba_build_poses, dim3(P.n_free), dim3(kPoseWg), 0, s, D, 0);
    }
    GFS_LAUNCH("k_lba_begin", k_lba_begin, dim3(1), dim3(kThreads), 0, s, D, 1, 0);
    LbaDev Df = D;
    Df.mode = 1;
    GFS_LAUNCH("k_lba_finish", k_lba_finish, dim3(1), dim3(64), 0, s, Df);
    GFS_HIP(hipStreamSynchronize(s));
    return GFS_OK;
  }
  const int npairs = P.n_free * (P.n_free + 1) / 2;
  // ---- the LM loop.  Round 6: the host runs ONE ITERATION AHEAD of what it knows.  Behind the decide kernel of iteration k's trial it
  // queues the whole of iteration k + 1 -- build group and first trial -- GATED on the device by what that kernel decides
  // (LbaState::spec_ok: trial accepted, loop goes on; every kernel of a gated launch leaves at once otherwise), and only then waits
  // for decide k's flags (an event behind it, the flags in host-mapped memory).  An accepted trial -- the rule -- finds the GPU already
  // at work on the next iteration while the host wakes up and queues the one after: the ~24 us round trip and the ~40 us of launch
  // calls per iteration are off the critical path.  A rejected trial costs the launches of one gated iteration (~35 us) and is retried
  // ungated.  Same kernels in the same order per executed trial: bit-identical results.  The caller's stop flag is looked at where
  // g2o looks (top of an iteration, end of a rejected trial); if it is up when an iteration is already running ahead, that iteration
  // is DISCARDED: its trial wrote only the other estimate buffer, the scalar state decide k left is put back from its snapshot
  // (k_lba_restore) and the errors are evaluated again at that estimate -- what the loop without the look-ahead would have left.
  // GFS_LBA_SPECULATE=0: that loop (every launch once the flags are known).
  static const bool speculate = !(getenv("GFS_LBA_SPECULATE") && atoi(getenv("GFS_LBA_SPECULATE")) == 0);
  auto build_group = [&](int iteration, int gate) -> int {
    GFS_LAUNCH("k_lba_errors", k_lba_errors, g_err, dim3(kMk), 0, s, D, 0, gate);
    if (D.n_lm_wg > 0) GFS_LAUNCH("k_lba_build_landmarks", k_lba_build_landmarks, g_lm, dim3(kMk), 0, s, D, gate);
    if (P.n_free > 0) GFS_LAUNCH("k_lba_build_poses", k_lba_build_poses, dim3(P.n_free), dim3(kPoseWg), 0, s, D, gate);
    GFS_LAUNCH("k_lba_begin", k_lba_begin, dim3(1), dim3(kThreads), 0, s, D, iteration, gate);
    return GFS_OK;
  };
  int n_decides = 0;  // decide kernels queued so far: number i writes slot i & 1 of h_flags and is followed by event i & 1
  auto trial_group = [&](int gate) -> int {
    GFS_LAUNCH("k_lba_dinv", k_lba_dinv, g_upd, dim3(kMk), 0, s, D, gate);
    if (npairs > 0 && D.schur_mfma) {
      if (D.n_pair_tiles == 1)
        GFS_LAUNCH("k_lba_schur_mfma", k_lba_schur_mfma<true>, dim3(D.n_schur_chunks), dim3(kMk), schur_mfma_lds_bytes(D), s, D, gate);
      else
        GFS_LAUNCH("k_lba_schur_mfma", k_lba_schur_mfma<false>, dim3(D.n_schur_chunks * D.n_pair_tiles), dim3(kMk), schur_mfma_lds_bytes(D), s, D, gate);
      GFS_LAUNCH("k_lba_schur_reduce", k_lba_schur_reduce_mfma, dim3(gfs::div_up(n * (n + 1) / 2 + n, kMk)), dim3(kMk), 0, s, D, gate);
    } else if (npairs > 0 && D.schur_sub) {
      GFS_LAUNCH("k_lba_schur_chunks", k_lba_schur_chunks, dim3(D.n_schur_chunks * D.n_pair_tiles), dim3(kMk), schur_lds_bytes(D), s, D, gate);
      GFS_LAUNCH("k_lba_schur_reduce", k_lba_schur_reduce, dim3(gfs::div_up(n * (n + 1) / 2 + n, kMk)), dim3(kMk), 0, s, D, gate);
    } else if (npairs > 0) {
      GFS_LAUNCH("k_lba_schur", k_lba_schur, dim3(npairs), dim3(kMk), 0, s, D, gate);
    }
    if (in_lds)
      GFS_LAUNCH("k_lba_solve", k_lba_solve<true>, dim3(1), dim3(kThreads), lds, s, D, gate);
    else
      GFS_LAUNCH("k_lba_solve", k_lba_solve<false>, dim3(1), dim3(kThreads), lds, s, D, gate);
    GFS_LAUNCH("k_lba_update", k_lba_update, g_upd, dim3(kMk), 0, s, D, gate);
    GFS_LAUNCH("k_lba_errors", k_lba_errors, g_err, dim3(kMk), 0, s, D, 1, gate);
    GFS_LAUNCH("k_lba_decide", k_lba_decide, dim3(1), dim3(64), 0, s, D, 0, d_flags + kFlagInts * (n_decides & 1), gate);
    GFS_HIP(hipEventRecord(h->ev_decide[n_decides & 1], s));
    n_decides++;
    return GFS_OK;
  };
  auto flags_of = [&](int i) { return h->h_flags + kFlagInts * (i & 1); };
  if (!speculate) {
    for (int iteration = 0; iteration < p->iterations; iteration++) {
      if (stop && *stop) break;  // SparseOptimizer::terminate() at the top of the iteration
      if ((rc = build_group(iteration, 0))) return rc;
      bool terminate = false;
      for (;;) {
        if ((rc = trial_group(0))) return rc;
        GFS_HIP(hipStreamSynchronize(s));
        const int* f = flags_of(n_decides - 1);
        const bool again = f[0] != 0;
        terminate = f[1] != 0;
        if (!again) break;
        if (stop && *stop) {  // the stop flag ends the trial loop: close the iteration's bookkeeping on the device
          GFS_LAUNCH("k_lba_decide", k_lba_decide, dim3(1), dim3(64), 0, s, D, 1, d_flags + kFlagInts * (n_decides & 1), 0);
          n_decides++;
          GFS_HIP(hipStreamSynchronize(s));
          terminate = flags_of(n_decides - 1)[1] != 0;
          break;
        }
      }
      if (terminate) break;
    }
  } else if (!(stop && *stop)) {
    int iteration = 0;
    if ((rc = build_group(0, 0)) || (rc = trial_group(0))) return rc;
    int waiting = n_decides - 1;  // the decide whose verdict the host waits for next
    for (;;) {
      const bool ahead = iteration + 1 < p->iterations;
      if (ahead && ((rc = build_group(iteration + 1, 1)) || (rc = trial_group(1)))) return rc;  // gated on decide `waiting`
      GFS_HIP(hipEventSynchronize(h->ev_decide[waiting & 1]));
      const int* f = flags_of(waiting);
      const bool again = f[0] != 0, terminate = f[1] != 0;
      if (again) {  // rejected: what was queued ahead has left (or is leaving) untouched
        if (stop && *stop) {
          GFS_LAUNCH("k_lba_decide", k_lba_decide, dim3(1), dim3(64), 0, s, D, 1, d_flags + kFlagInts * (n_decides & 1), 0);
          n_decides++;
          break;
        }
        if ((rc = trial_group(0))) return rc;
        waiting = n_decides - 1;
        continue;
      }
      if (terminate || !ahead) break;
      iteration++;  // accepted, the loop goes on: iteration `iteration` is already running (decide waiting + 1)
      if (stop && *stop) {  // ... but g2o would not have started it: put decide `waiting`'s state back
        const int cur = f[2], iters = f[3];
        const double* snap = reinterpret_cast<const double*>(f + 4);
        const double lambda = snap[0], last_chi = snap[1];
        GFS_HIP(hipStreamSynchronize(s));  // (the iteration running ahead: let it finish, nothing reads the snapshot slot meanwhile)
        GFS_LAUNCH("k_lba_restore", k_lba_restore, dim3(1), dim3(64), 0, s, D, cur, iters, lambda, last_chi);
        GFS_LAUNCH("k_lba_errors", k_lba_errors, g_err, dim3(kMk), 0, s, D, 0, 0);
        break;
      }
      waiting = waiting + 1;
    }
  }
  GFS_LAUNCH("k_lba_finish", k_lba_finish, dim3(1), dim3(64), 0, s, D);
  if (timing) {
    GFS_HIP(hipStreamSynchronize(s));
    fprintf(stderr, "  run: LM loop %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - Ta).count());
  }
  return GFS_OK;
}

}  // namespace

extern "C" {

int gfs_lba_create(int device, int max_poses, int max_points, int max_edges, gfs_lba** out) {
  GFS_REQUIRE(out && max_poses > 0 && max_points > 0 && max_edges > 0, GFS_ERR_INVALID_ARG, "gfs_lba_create: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_lba> h(new gfs_lba);
  h->device = device;
  h->max_poses = max_poses;
  h->max_points = max_points;
  h->max_edges = max_edges;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  if (int rc0 = lba_raise_lds_limits(device)) return rc0;
  GFS_HIP(hipHostMalloc((void**)&h->h_stop, sizeof(int), hipHostMallocMapped));
  GFS_HIP(hipHostMalloc((void**)&h->h_flags, 2 * kFlagInts * sizeof(int), hipHostMallocMapped));
  for (int k = 0; k < 2; k++) GFS_HIP(hipEventCreateWithFlags(&h->ev_decide[k], hipEventDisableTiming));
  const size_t NP = max_points, E = max_edges, NQ = max_poses, F = max_poses;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_q.alloc(NQ * 4));
  A(h->d_t.alloc(NQ * 3));
  A(h->d_X.alloc(NP * 3));
  A(h->d_qt.alloc(NQ * 4));
  A(h->d_tt.alloc(NQ * 3));
  A(h->d_Xt.alloc(NP * 3));
  A(h->d_chi2.alloc(E));
  A(h->d_err.alloc(E * 3));
  A(h->d_Hpl.alloc(E * 18));
  A(h->d_Hll.alloc(NP * 6));
  A(h->d_bl.alloc(NP * 3));
  A(h->d_Dinv.alloc(NP * 6));
  A(h->d_Hpp.alloc(F * 21));
  A(h->d_bp.alloc(F * 6));
  A(h->d_xl.alloc(NP * 3));
  A(h->d_xp.alloc(F * 6));
  A(h->d_stats.alloc(2));
  A(h->d_info.alloc(2));
  A(h->d_state.alloc(1));
  A(h->h_stage.alloc(NQ * (32 + 24 + 4) + NP * (24 + 4) + E * (4 + 4 + 24 + 8 + 1 + 4 + 4) + F * (4 + 4 + NP * 4) + (NP + 2) * 4 + 4096));
  A(h->d_in.alloc(h->h_stage.n));
  A(h->d_out.alloc(E + 7 * NQ + 3 * NP + 4));
  A(h->h_out.alloc(E + 7 * NQ + 3 * NP + 4));
  A(h->d_part_chi.alloc(E / kMk + 2));
  A(h->d_part_scale.alloc(NP / kMk + 2));
  A(h->d_Hs.alloc((size_t)(6 * F) * (6 * F + 1) / 2 + 8));
  A(h->d_bs.alloc(6 * F + 8));
#undef A
  if (rc) return rc;
  *out = h.release();
  return GFS_OK;
}

void gfs_lba_destroy(gfs_lba* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->self_batch) gfs_lba_batch_destroy(h->self_batch);  // (a wrapper: it does not own this window)
  h->self_batch = nullptr;
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  if (h->h_stop) (void)hipHostFree(h->h_stop);
  if (h->h_flags) (void)hipHostFree(h->h_flags);
  for (int k = 0; k < 2; k++)
    if (h->ev_decide[k]) (void)hipEventDestroy(h->ev_decide[k]);
  delete h;
}

// result download of one window, in two halves so that a batch can queue all copies before one synchronisation: the packed
// block of b_pack comes back in one copy
struct LbaFetch {
  const double *chi, *q, *t, *X, *stats;
};
static int lba_fetch_issue(gfs_lba* h, const gfs_lba_problem* p, hipStream_t s, LbaFetch& F) {
  const size_t E = p->n_edges, NP = p->n_points, NQ = p->n_poses, total = E + 7 * NQ + 3 * NP + 4;
  GFS_REQUIRE(total <= h->h_out.n, GFS_ERR_CAPACITY, "gfs_lba: result arena too small");
  const double* base = h->h_out.p;
  F.chi = base;
  F.q = base + E;
  F.t = F.q + 4 * NQ;
  F.X = F.t + 3 * NQ;
  F.stats = F.X + 3 * NP;
  GFS_HIP(hipMemcpyAsync(h->h_out.p, h->d_out.p, total * sizeof(double), hipMemcpyDeviceToHost, s));
  return GFS_OK;
}
static void lba_fetch_finish(const gfs_lba_problem* p, const HostPrep& P, const LbaFetch& F, gfs_lba_solution* sol) {
  const int E = p->n_edges, NP = p->n_points;
  const double *chi = F.chi, *q = F.q, *t = F.t, *X = F.X, *stats = F.stats;
  if (p->n_poses && sol->pose_q) memcpy(sol->pose_q, q, (size_t)p->n_poses * 32);
  if (p->n_poses && sol->pose_t) memcpy(sol->pose_t, t, (size_t)p->n_poses * 24);
  if (NP && sol->points) memcpy(sol->points, X, (size_t)NP * 24);
  for (int k = 0; k < E; k++) {
    const int e = P.order[k];
    if (sol->edge_chi2) sol->edge_chi2[e] = chi[k];
    if (sol->edge_depth_positive) {  // isDepthPositive() at the final estimates (host side, trivial)
      const double* qq = &q[4 * (size_t)p->edge_pose[e]];
      const double* v = &X[3 * (size_t)p->edge_point[e]];
      const double ux = 2 * (qq[1] * v[2] - qq[2] * v[1]), uy = 2 * (qq[2] * v[0] - qq[0] * v[2]);
      const double uz = 2 * (qq[0] * v[1] - qq[1] * v[0]);
      const double z = v[2] + qq[3] * uz + (qq[0] * uy - qq[1] * ux) + t[3 * (size_t)p->edge_pose[e] + 2];
      sol->edge_depth_positive[e] = z > 0.0;
    }
  }
  sol->iterations_run = (int)stats[2];
  sol->final_chi2 = stats[0];
  sol->final_lambda = stats[1];
}

static int lba_solve_one_batched(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* sol, StopFlag stop);
static int lba_solve_impl(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* sol, StopFlag stop) {
  GFS_REQUIRE(h && p && sol, GFS_ERR_INVALID_ARG, "gfs_lba_solve: NULL argument");
  if (stop && *stop) {  // if (pbStopFlag) if (*pbStopFlag) return;  (src/Optimizer.cc:1955-1956)
    gfs::set_error("gfs_lba_solve: stop flag raised before optimisation");
    return GFS_ERR_STOPPED;
  }
  // GFS_LBA_SINGLE=batched: one window through the batched kernels, whose LM state machine lives on the device (the host queues
  // rounds ahead and polls one round behind instead of waiting for two flags after every trial).  Measured in round 4: 2.29 ms
  // against 2.23 ms for the host-driven loop below -- the descriptor indirection and the gated launches of the batched kernels
  // cost what the saved round trips return -- so it is not the default.  Same arithmetic, bit-identical results.
  static const bool batched_loop = getenv("GFS_LBA_SINGLE") && strcmp(getenv("GFS_LBA_SINGLE"), "batched") == 0 && !getenv("GFS_LBA_SINGLE_WG");
  if (batched_loop) return lba_solve_one_batched(h, p, sol, stop);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  static const bool timing = getenv("GFS_LBA_TIMING") != nullptr;
  const auto T0 = std::chrono::steady_clock::now();
  static thread_local HostPrep tl_prep;  // vectors keep their capacity between calls (no allocation on the hot path)
  HostPrep& P = tl_prep;
  int rc = prepare(h, p, P);
  if (rc) return rc;
  const auto T1 = std::chrono::steady_clock::now();
  rc = run(h, p, P, 0, stop);
  if (rc) return rc;
  const auto T2 = std::chrono::steady_clock::now();
  if (timing)
    fprintf(stderr, "gfs_lba_solve: prepare %.2f ms, upload + solve %.2f ms\n", std::chrono::duration<double, std::milli>(T1 - T0).count(),
            std::chrono::duration<double, std::milli>(T2 - T1).count());
  LbaFetch F;
  GFS_LAUNCH("k_lba_pack", k_lba_pack, dim3(16), dim3(kMk), 0, h->stream, h->last_desc);
  rc = lba_fetch_issue(h, p, h->stream, F);
  if (rc) return rc;
  GFS_HIP(hipStreamSynchronize(h->stream));
  lba_fetch_finish(p, P, F, sol);
  return GFS_OK;
}

// ---- batched windows -----------------------------------------------------------------------------------------------
struct gfs_lba_batch {
  int device = 0, max_windows = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  std::vector<gfs_lba*> win;       // one workspace per window (their kernels run together: window index = blockIdx.y)
  gfs::DevBuf<LbaDev> d_desc;
  gfs::PinBuf<LbaDev> h_desc;
  gfs::DevBuf<int> d_flags, d_done;
  int* h_done = nullptr;           // pinned copy targets of the done counter (two rounds in flight)
  hipEvent_t ev_round[2] = {nullptr, nullptr};
  bool owns_windows = true;        // false: the one-window wrapper of a gfs_lba (gfs_lba_solve runs through the batched kernels)
  gfs::PinBuf<int> h_cur;          // per window: which estimate buffer holds the result
  std::vector<HostPrep> prep;
};

int gfs_lba_batch_create(int device, int max_windows, int max_poses, int max_points, int max_edges, gfs_lba_batch** out) {
  GFS_REQUIRE(out && max_windows > 0, GFS_ERR_INVALID_ARG, "gfs_lba_batch_create: invalid argument");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_lba_batch> b(new gfs_lba_batch);
  b->device = device;
  b->max_windows = max_windows;
  GFS_HIP(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  for (int w = 0; w < max_windows; w++) {
    gfs_lba* h = nullptr;
    const int rc = gfs_lba_create(device, max_poses, max_points, max_edges, &h);
    if (rc) {
      gfs_lba_batch_destroy(b.release());
      return rc;
    }
    b->win.push_back(h);
  }
  int rc = 0;
  if ((rc = b->d_desc.alloc(max_windows)) || (rc = b->h_desc.alloc(max_windows)) || (rc = b->d_flags.alloc(4 * (size_t)max_windows)) ||
      (rc = b->d_done.alloc(1)) || (rc = b->h_cur.alloc(max_windows))) {
    gfs_lba_batch_destroy(b.release());  // the windows (tens of MB of HBM and pinned arenas each) and the stream go with it
    return rc;
  }
  if (hipHostMalloc((void**)&b->h_done, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) {
    b->h_done = nullptr;
    gfs::set_error("gfs_lba_batch_create: hipHostMalloc failed");
    gfs_lba_batch_destroy(b.release());
    return GFS_ERR_HIP;
  }
  for (int k = 0; k < 2; k++)
    if (hipEventCreateWithFlags(&b->ev_round[k], hipEventDisableTiming) != hipSuccess) {
      gfs::set_error("gfs_lba_batch_create: hipEventCreate failed");
      gfs_lba_batch_destroy(b.release());
      return GFS_ERR_HIP;
    }
  b->prep.resize(max_windows);
  *out = b.release();
  return GFS_OK;
}

void gfs_lba_batch_destroy(gfs_lba_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->owns_windows)
    for (gfs_lba* x : b->win) gfs_lba_destroy(x);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  for (int k = 0; k < 2; k++)
    if (b->ev_round[k]) (void)hipEventDestroy(b->ev_round[k]);
  if (b->h_done) (void)hipHostFree(b->h_done);
  delete b;
}

// n independent windows (replicas of the single-window solve: "LBA of one map does not shard") solved
// together: per window exactly the arithmetic of gfs_lba_solve, the launches shared by all of them.
static int lba_solve_batch_impl(gfs_lba_batch* b, const gfs_lba_problem* problems, gfs_lba_solution* solutions, int n, StopFlag stop) {
  GFS_REQUIRE(b && problems && solutions && n > 0, GFS_ERR_INVALID_ARG, "gfs_lba_solve_batch: invalid argument");
  GFS_REQUIRE(n <= b->max_windows, GFS_ERR_CAPACITY, "gfs_lba_solve_batch: %d windows exceed the handle's %d", n, b->max_windows);
  if (stop && *stop) {
    gfs::set_error("gfs_lba_solve_batch: stop flag raised before optimisation");
    return GFS_ERR_STOPPED;
  }
  std::lock_guard<std::mutex> lk(b->mu);
  GFS_HIP(hipSetDevice(b->device));
  hipStream_t s = b->stream;
  static const bool timing = getenv("GFS_LBA_TIMING") != nullptr;
  auto now = []() { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) {
    return std::chrono::duration<double, std::milli>(c - a).count();
  };
  const auto T0 = now();
  // ---- host preparation of every window (edge re-ordering, CSR tables), on a few threads
  std::vector<int> rcs((size_t)n, 0);
  std::vector<std::string> msgs((size_t)n);  // gfs_last_error() is per thread: a worker's message is carried back to the caller's
  {
    static const int cap = getenv("GFS_LBA_THREADS") ? atoi(getenv("GFS_LBA_THREADS")) : 32;
    const int nthreads = std::max(1, std::min(n, std::min(cap, (int)std::thread::hardware_concurrency())));
    auto work = [&](int t) {
      (void)hipSetDevice(b->device);
      for (int w = t; w < n; w += nthreads) {
        rcs[w] = prepare(b->win[w], &problems[w], b->prep[w]);
        if (!rcs[w]) rcs[w] = upload(b->win[w], b->prep[w], s);  // the copy leaves as soon as the window is ready
        if (rcs[w]) msgs[w] = gfs_last_error();
      }
    };
    if (nthreads == 1) {
      work(0);  // one window: no thread to start
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; t++) th.emplace_back(work, t);
      for (auto& x : th) x.join();
    }
  }
  for (int w = 0; w < n; w++)
    if (rcs[w]) {
      gfs::set_error("gfs_lba_solve_batch: window %d: %s", w, msgs[w].c_str());
      return rcs[w];
    }
  const auto T1 = now();
  int max_free = 0, max_err = 1, max_upd = 1, max_lm = 1, max_iter = 0, max_chunk_blocks = 0, max_pair_blocks = 0;
  size_t schur_lds = 0, mfma_lds = 0;
  int max_mfma_blocks = 0;
  bool any_one_block = false, any_multi_block = false;
  for (int w = 0; w < n; w++) {
    LbaDev D;
    const int rc = upload_and_fill(b->win[w], &problems[w], b->prep[w], 0, s, D, false);
    if (rc) return rc;
    {
      void* dstop = nullptr;
      GFS_HIP(hipHostGetDevicePointer(&dstop, b->win[w]->h_stop, 0));
      D.stop = (decltype(D.stop))dstop;
    }
    *b->win[w]->h_stop = 0;
    b->h_desc.p[w] = D;
    max_free = std::max(max_free, D.n_free);
    max_err = std::max(max_err, D.n_err_blocks);
    max_upd = std::max(max_upd, D.n_upd_blocks);
    max_lm = std::max(max_lm, D.n_lm_wg);
    max_iter = std::max(max_iter, problems[w].iterations);
    if (D.schur_mfma) {
      max_mfma_blocks = std::max(max_mfma_blocks, D.n_schur_chunks * D.n_pair_tiles);
      any_one_block = any_one_block || D.n_pair_tiles == 1;
      any_multi_block = any_multi_block || D.n_pair_tiles != 1;
      mfma_lds = std::max(mfma_lds, schur_mfma_lds_bytes(D));
    } else if (D.schur_sub) {
      max_chunk_blocks = std::max(max_chunk_blocks, D.n_schur_chunks * D.n_pair_tiles);
      schur_lds = std::max(schur_lds, schur_lds_bytes(D));
    } else {
      max_pair_blocks = std::max(max_pair_blocks, D.n_free * (D.n_free + 1) / 2);
    }
  }
  GFS_HIP(hipMemcpyAsync(b->d_desc.p, b->h_desc.p, (size_t)n * sizeof(LbaDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemsetAsync(b->d_done.p, 0, sizeof(int), s));
  const int nmax = 6 * max_free;
  int free_small = 0, free_large = 0;  // largest window solved in LDS / in HBM
  for (int w = 0; w < n; w++) {
    const int f = b->h_desc.p[w].n_free;
    if (f <= kMaxFreeLds)
      free_small = std::max(free_small, f);
    else
      free_large = std::max(free_large, f);
  }
  const size_t lds_small = ((size_t)(6 * free_small) * (6 * free_small + 1) / 2 + 8 * (6 * free_small) + 8) * sizeof(double);
  const size_t lds_large = ((size_t)7 * 6 * free_large + 8) * sizeof(double);
  GFS_REQUIRE(lds_large <= 160 * 1024 - 1024, GFS_ERR_CAPACITY, "gfs_lba_solve_batch: %d free poses exceed the solver's workspace", max_free);
  const LbaDev* DD = b->d_desc.p;
  const auto T2 = now();
  int rounds_run = 0;
  GFS_LAUNCH("kb_lba_init", kb_lba_init, dim3(64, n), dim3(kMk), 0, s, DD);
  const int max_rounds = std::max(max_iter, 0) * 11 + 1;  // (windows with iterations <= 0 leave in the first round, after the errors)
  bool stopped = false;
  for (int round = 0; round < max_rounds; round++) {
    GFS_LAUNCH("kb_lba_errors", kb_lba_errors, dim3(max_err, n), dim3(kMk), 0, s, DD, 0);
    GFS_LAUNCH("kb_lba_build_landmarks", kb_lba_build_landmarks, dim3(max_lm, n), dim3(kMk), 0, s, DD);
    if (max_free > 0) GFS_LAUNCH("kb_lba_build_poses", kb_lba_build_poses, dim3(max_free, n), dim3(kPoseWg), 0, s, DD);
    GFS_LAUNCH("kb_lba_begin", kb_lba_begin, dim3(1, n), dim3(kThreads), 0, s, DD, b->d_done.p);
    GFS_LAUNCH("kb_lba_dinv", kb_lba_dinv, dim3(max_upd, n), dim3(kMk), 0, s, DD);
    if (max_mfma_blocks > 0) {
      if (any_one_block) GFS_LAUNCH("kb_lba_schur_mfma", kb_lba_schur_mfma<true>, dim3(max_mfma_blocks, n), dim3(kMk), mfma_lds, s, DD);
      if (any_multi_block) GFS_LAUNCH("kb_lba_schur_mfma", kb_lba_schur_mfma<false>, dim3(max_mfma_blocks, n), dim3(kMk), mfma_lds, s, DD);
      GFS_LAUNCH("kb_lba_schur_reduce", kb_lba_schur_reduce_mfma, dim3(gfs::div_up(nmax * (nmax + 1) / 2 + nmax, kMk), n), dim3(kMk), 0, s, DD);
    }
    if (max_chunk_blocks > 0) {
      GFS_LAUNCH("kb_lba_schur_chunks", kb_lba_schur_chunks, dim3(max_chunk_blocks, n), dim3(kMk), schur_lds, s, DD);
      GFS_LAUNCH("kb_lba_schur_reduce", kb_lba_schur_reduce, dim3(gfs::div_up(nmax * (nmax + 1) / 2 + nmax, kMk), n), dim3(kMk), 0, s, DD);
    }
    if (max_pair_blocks > 0) GFS_LAUNCH("kb_lba_schur", kb_lba_schur, dim3(max_pair_blocks, n), dim3(kMk), 0, s, DD);
    bool any_small = false;
    for (int w = 0; w < n; w++) any_small = any_small || b->h_desc.p[w].n_free <= kMaxFreeLds;
    if (any_small) GFS_LAUNCH("kb_lba_solve", kb_lba_solve<true>, dim3(1, n), dim3(kThreads), lds_small, s, DD);
    if (free_large > 0) GFS_LAUNCH("kb_lba_solve", kb_lba_solve<false>, dim3(1, n), dim3(kThreads), lds_large, s, DD);
    GFS_LAUNCH("kb_lba_update", kb_lba_update, dim3(max_upd, n), dim3(kMk), 0, s, DD);
    GFS_LAUNCH("kb_lba_errors", kb_lba_errors, dim3(max_err, n), dim3(kMk), 0, s, DD, 1);
    const bool force_end = stop && *stop;  // setForceStopFlag: the running iteration is closed, nothing further starts
    GFS_LAUNCH("kb_lba_decide", kb_lba_decide, dim3(n), dim3(64), 0, s, DD, force_end ? 1 : 0, b->d_flags.p, b->d_done.p);
    // The done counter is polled ONE ROUND BEHIND the round just queued: the GPU never idles on the host round trip (a window's
    // state machine lives on the device, a round queued for windows that are all done finds every kernel leaving at once).
    GFS_HIP(hipMemcpyAsync(b->h_done + (round & 1), b->d_done.p, sizeof(int), hipMemcpyDeviceToHost, s));
    GFS_HIP(hipEventRecord(b->ev_round[round & 1], s));
    rounds_run++;
    static const bool poll_sync = getenv("GFS_LBA_POLL") && strcmp(getenv("GFS_LBA_POLL"), "sync") == 0;  // A/B: wait for every round
    if (force_end || poll_sync) {
      GFS_HIP(hipStreamSynchronize(s));
      if (poll_sync && !force_end) {
        if (b->h_done[round & 1] >= n) break;
        continue;
      }
      stopped = true;
      break;
    }
    if (round >= 1) {
      GFS_HIP(hipEventSynchronize(b->ev_round[(round - 1) & 1]));
      if (b->h_done[(round - 1) & 1] >= n) break;
    }
  }
  (void)stopped;
  const auto T3 = now();
  GFS_LAUNCH("kb_lba_finish", kb_lba_finish, dim3(n), dim3(64), 0, s, DD);
  GFS_LAUNCH("kb_lba_pack", kb_lba_pack, dim3(8, n), dim3(kMk), 0, s, DD);
  std::vector<LbaFetch> F((size_t)n);
  for (int w = 0; w < n; w++) {
    const int rc = lba_fetch_issue(b->win[w], &problems[w], s, F[w]);
    if (rc) return rc;
  }
  GFS_HIP(hipStreamSynchronize(s));
  const auto T4 = now();
  {
    const int nthreads = std::max(1, std::min(n, std::min(32, (int)std::thread::hardware_concurrency())));
    if (nthreads == 1) {
      for (int w = 0; w < n; w++) lba_fetch_finish(&problems[w], b->prep[w], F[w], &solutions[w]);
    } else {
      std::vector<std::thread> th;
      for (int t = 0; t < nthreads; t++)
        th.emplace_back([&, t]() {
          for (int w = t; w < n; w += nthreads) lba_fetch_finish(&problems[w], b->prep[w], F[w], &solutions[w]);
        });
      for (auto& x : th) x.join();
    }
  }
  if (timing)
    fprintf(stderr, "gfs_lba_solve_batch(%d): prepare %.2f, upload %.2f, %d rounds %.2f, download %.2f, scatter %.2f ms\n", n, ms(T0, T1),
            ms(T1, T2), rounds_run, ms(T2, T3), ms(T3, T4), ms(T4, now()));
  return GFS_OK;
}

// gfs_lba_solve through the batched path: a wrapper batch of one window around the handle itself (created on first use)
static int lba_solve_one_batched(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* sol, StopFlag stop) {
  {
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->self_batch) {
      GFS_HIP(hipSetDevice(h->device));
      std::unique_ptr<gfs_lba_batch> b(new gfs_lba_batch);
      b->device = h->device;
      b->max_windows = 1;
      b->owns_windows = false;
      b->win.push_back(h);
      int rc = 0;
      if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) rc = GFS_ERR_HIP;
      if (!rc && ((rc = b->d_desc.alloc(1)) || (rc = b->h_desc.alloc(1)) || (rc = b->d_flags.alloc(4)) || (rc = b->d_done.alloc(1)) ||
                  (rc = b->h_cur.alloc(1)))) {
      }
      if (!rc && hipHostMalloc((void**)&b->h_done, 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) rc = GFS_ERR_HIP;
      for (int k = 0; k < 2 && !rc; k++)
        if (hipEventCreateWithFlags(&b->ev_round[k], hipEventDisableTiming) != hipSuccess) rc = GFS_ERR_HIP;
      if (rc) {
        gfs::set_error("gfs_lba_solve: could not set up the one-window batch");
        gfs_lba_batch_destroy(b.release());
        return rc;
      }
      b->prep.resize(1);
      h->self_batch = b.release();
    }
  }
  const int rc = lba_solve_batch_impl(h->self_batch, p, sol, 1, stop);
  if (rc == GFS_ERR_STOPPED) gfs::set_error("gfs_lba_solve: stop flag raised before optimisation");
  return rc;
}

int gfs_lba_linearize(gfs_lba* h, const gfs_lba_problem* p, double* Hpp, double* Hll, double* Hpl, double* bp, double* bl,
                      double* edge_chi2, double* chi2) {
  GFS_REQUIRE(h && p, GFS_ERR_INVALID_ARG, "gfs_lba_linearize: NULL argument");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  static thread_local HostPrep tl_prep;  // vectors keep their capacity between calls (no allocation on the hot path)
  HostPrep& P = tl_prep;
  int rc = prepare(h, p, P);
  if (rc) return rc;
  rc = run(h, p, P, 1, StopFlag{});
  if (rc) return rc;
  const int E = p->n_edges, NP = p->n_points, F = P.n_free;
  std::vector<double> hpp((size_t)F * 21), hll((size_t)NP * 6), hpl((size_t)E * 18), chi(E);
  if (F) GFS_HIP(hipMemcpy(hpp.data(), h->d_Hpp.p, hpp.size() * 8, hipMemcpyDeviceToHost));
  if (NP) GFS_HIP(hipMemcpy(hll.data(), h->d_Hll.p, hll.size() * 8, hipMemcpyDeviceToHost));
  if (E) GFS_HIP(hipMemcpy(hpl.data(), h->d_Hpl.p, hpl.size() * 8, hipMemcpyDeviceToHost));
  if (E) GFS_HIP(hipMemcpy(chi.data(), h->d_chi2.p, chi.size() * 8, hipMemcpyDeviceToHost));
  if (bp && F) GFS_HIP(hipMemcpy(bp, h->d_bp.p, (size_t)F * 48, hipMemcpyDeviceToHost));
  if (bl && NP) GFS_HIP(hipMemcpy(bl, h->d_bl.p, (size_t)NP * 24, hipMemcpyDeviceToHost));
  double stats[2];
  GFS_HIP(hipMemcpy(stats, h->d_stats.p, sizeof(stats), hipMemcpyDeviceToHost));
  if (chi2) *chi2 = stats[0];
  if (Hpp)
    for (int f = 0; f < F; f++) {
      int o = 0;
      for (int a = 0; a < 6; a++)
        for (int c = a; c < 6; c++) {
          Hpp[36 * (size_t)f + a + 6 * c] = hpp[21 * (size_t)f + o];
          Hpp[36 * (size_t)f + c + 6 * a] = hpp[21 * (size_t)f + o];
          o++;
        }
    }
  if (Hll)
    for (int l = 0; l < NP; l++) {
      const double* s6 = &hll[6 * (size_t)l];
      double* o = Hll + 9 * (size_t)l;
      o[0] = s6[0];
      o[1] = o[3] = s6[1];
      o[2] = o[6] = s6[2];
      o[4] = s6[3];
      o[5] = o[7] = s6[4];
      o[8] = s6[5];
    }
  for (int k = 0; k < E; k++) {
    const int e = P.order[k];
    if (edge_chi2) edge_chi2[e] = chi[k];
    if (Hpl)
      for (int a = 0; a < 6; a++)
        for (int c = 0; c < 3; c++) Hpl[18 * (size_t)e + a + 6 * c] = hpl[18 * (size_t)k + 3 * a + c];
  }
  return GFS_OK;
}

int gfs_lba_solve(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* sol, volatile const int* stop) {
  StopFlag f;
  f.i = stop;
  return lba_solve_impl(h, p, sol, f);
}
int gfs_lba_solve_bool(gfs_lba* h, const gfs_lba_problem* p, gfs_lba_solution* sol, const volatile unsigned char* stop) {
  StopFlag f;
  f.b = stop;
  return lba_solve_impl(h, p, sol, f);
}
int gfs_lba_solve_batch(gfs_lba_batch* b, const gfs_lba_problem* problems, gfs_lba_solution* solutions, int n, volatile const int* stop) {
  StopFlag f;
  f.i = stop;
  return lba_solve_batch_impl(b, problems, solutions, n, f);
}

}  // extern "C"
