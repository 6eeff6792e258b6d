// cv::findFundamentalMat(points1, points2, cv::FM_RANSAC, threshold, confidence, mask) on MI355X, as the reference calls it in the
// optical-flow matcher (src/ORBmatcher.cc:2399-2405, 2463-2469, also :236) and Tracking::EstimatePoseByOF (src/Tracking.cc:1973-1974):
// the 7-point RANSAC of OpenCV's RANSACPointSetRegistrator (calib3d/src/ptsetreg.cpp, fundam.cpp; restated in
// oracle/fmat_oracle.cpp, which also lists the two deliberate differences: null space by Gauss-Jordan elimination and cubic roots by
// bisection, so that every step is +, -, *, /, sqrt and the device reproduces the oracle's bits).  15 or more points: RANSAC;
// 8 .. 14 points: the LMedS registrator cv::findFundamentalMat switches to (k_fmat_median instead of k_fmat_count).
//
// RANSAC looks sequential (the iteration budget shrinks whenever a better model is found) but the random subsets do not depend
// on the models: the cv::RNG stream is consumed the same way whatever is accepted.  So a round evaluates a CHUNK of iterations
// (16 first, then 64) of every problem at once — k_fmat_hyp: one lane draws the subsets (the only sequential part), a lane per
// iteration solves the 7-point problem (its 7 x 9 system in LDS); k_fmat_count: one wave per model counts its inliers over all
// points — and the host replays
// the accept / update-budget logic over the 64 results in order (its libm evaluates RANSACUpdateNumIters exactly like the
// oracle).  With a quarter or fewer outliers the first chunk is all it takes; harder problems continue chunk by chunk from the
// saved generator state.  Problems of a batch run in parallel.
#include <cmath>
#include <memory>
#include <mutex>

#include "gfs_common.hpp"

namespace {

using gfs::DevBuf;
using gfs::PinBuf;

constexpr int kFmChunk = 64;      // RANSAC iterations evaluated per launch
constexpr int kFmThreads = 256;
constexpr double kFltEps = 1.1920928955078125e-07, kDblEps = 2.220446049250313e-16;

struct FmProblem {
  int n;                         // points
  int active;                    // this problem still needs the current chunk
  unsigned long long rng;        // cv::RNG state before the chunk / after it
  float t2;                      // (float)(threshold^2); LMedS: (float)(sigma^2) once the best model is known
  int lmeds;                     // 8 .. 14 points: LMeDSPointSetRegistrator instead of RANSAC
  int max_attempts;              // getSubset: 10000 (RANSAC) / 1000 (LMedS)
};

__device__ __forceinline__ unsigned rng_next(unsigned long long& s) {  // cv::RNG::next
  s = (unsigned long long)(unsigned)s * 4164903690u + (unsigned)(s >> 32);
  return (unsigned)s;
}

__device__ __forceinline__ bool have_collinear_points(const float2 (&p)[7]) {  // the last of 7 points against every earlier pair
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const double dx1 = (double)(p[j].x - p[6].x), dy1 = (double)(p[j].y - p[6].y);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      if (k >= j) continue;
      const double dx2 = (double)(p[k].x - p[6].x), dy2 = (double)(p[k].y - p[6].y);
      if (fabs(dx2 * dy1 - dy2 * dx1) <= kFltEps * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) return true;
    }
  }
  return false;
}

// 7 distinct indices from the generator (the drawing half of getSubset)
__device__ __forceinline__ void draw_subset(unsigned long long& rng, int count, int (&idx)[7]) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    int v;
    for (;;) {
      v = (int)(rng_next(rng) % (unsigned)count);
      bool dup = false;
#pragma unroll
      for (int j = 0; j < 7; j++) dup |= j < i && idx[j] == v;
      if (!dup) break;
    }
    idx[i] = v;
  }
}
// FMEstimatorCallback::checkSubset: the 14 points are fetched together, the pair tests run on registers
__device__ __forceinline__ bool subset_ok(const float2* m1, const float2* m2, const int (&idx)[7]) {
  float2 a[7], b[7];
#pragma unroll
  for (int i = 0; i < 7; i++) {
    a[i] = m1[idx[i]];
    b[i] = m2[idx[i]];
  }
  return !have_collinear_points(a) && !have_collinear_points(b);
}

__device__ __forceinline__ double cubic_eval(double B, double C, double D, double x) { return ((x + B) * x + C) * x + D; }
__device__ double cubic_bisect(double B, double C, double D, double lo, double hi) {
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
__device__ int cubic_real_roots(const double* c, double* r) {
  if (c[0] == 0) {
    if (c[1] == 0) {
      if (c[2] == 0) return 0;
      r[0] = -c[3] / c[2];
      return 1;
    }
    const double d = c[2] * c[2] - 4 * c[1] * c[3];
    if (d < 0) return 0;
    const double sq = __dsqrt_rn(d);
    const double x0 = (-c[2] - sq) / (2 * c[1]), x1 = (-c[2] + sq) / (2 * c[1]);
    r[0] = x0 < x1 ? x0 : x1;
    r[1] = x0 < x1 ? x1 : x0;
    return d > 0 ? 2 : 1;
  }
  const double inv = 1. / c[0];
  const double B = c[1] * inv, C = c[2] * inv, D = c[3] * inv;
  double bound = fabs(B);
  if (fabs(C) > bound) bound = fabs(C);
  if (fabs(D) > bound) bound = fabs(D);
  bound += 1.0;
  if (!(bound < 1.0e300)) return 0;  // a vanishing leading coefficient blew the monic form up (or NaN): no usable root
  double knots[4];
  int nk = 0;
  knots[nk++] = -bound;  // Cauchy: every root lies in (-bound, bound)
  const double disc = B * B - 3 * C;
  if (disc > 0) {
    const double sq = __dsqrt_rn(disc);
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

// FMEstimatorCallback::runKernel for 7 points: up to three 3x3 matrices (row-major) in Fm
// `sa`: this lane's 7 x 9 system in LDS, element (i, j) at sa[(i * 9 + j) * kFmChunk] (lane-minor: conflict-free); the pivoting
// indexes it dynamically, which would push a register array into scratch memory.
__device__ int seven_point(const float2* m1, const float2* m2, const int* idx, double* Fm, double* sa) {
#define A_(i, j) sa[((i) * 9 + (j)) * kFmChunk]
  double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
  for (int i = 0; i < 7; i++) {
    c1x += (double)m1[idx[i]].x;
    c1y += (double)m1[idx[i]].y;
    c2x += (double)m2[idx[i]].x;
    c2y += (double)m2[idx[i]].y;
  }
  const double t = 1. / 7;
  c1x *= t;
  c1y *= t;
  c2x *= t;
  c2y *= t;
  double s1 = 0, s2 = 0;
  for (int i = 0; i < 7; i++) {
    const double ax = (double)m1[idx[i]].x - c1x, ay = (double)m1[idx[i]].y - c1y, bx = (double)m2[idx[i]].x - c2x,
                 by = (double)m2[idx[i]].y - c2y;
    s1 += __dsqrt_rn(ax * ax + ay * ay);
    s2 += __dsqrt_rn(bx * bx + by * by);
  }
  s1 *= t;
  s2 *= t;
  if (s1 < kFltEps || s2 < kFltEps) return 0;
  s1 = __dsqrt_rn(2.) / s1;
  s2 = __dsqrt_rn(2.) / s2;
  for (int i = 0; i < 7; i++) {
    const double x0 = ((double)m1[idx[i]].x - c1x) * s1, y0 = ((double)m1[idx[i]].y - c1y) * s1;
    const double x1 = ((double)m2[idx[i]].x - c2x) * s2, y1 = ((double)m2[idx[i]].y - c2y) * s2;
    A_(i, 0) = x1 * x0;
    A_(i, 1) = x1 * y0;
    A_(i, 2) = x1;
    A_(i, 3) = y1 * x0;
    A_(i, 4) = y1 * y0;
    A_(i, 5) = y1;
    A_(i, 6) = x0;
    A_(i, 7) = y0;
    A_(i, 8) = 1;
  }
  int perm[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
  for (int r = 0; r < 7; r++) {  // Gauss-Jordan with full pivoting
    int pi = r, pj = r;
    double pv = -1;
    for (int i = r; i < 7; i++)
      for (int j = r; j < 9; j++)
        if (fabs(A_(i, j)) > pv) {
          pv = fabs(A_(i, j));
          pi = i;
          pj = j;
        }
    if (!(pv > 0)) return 0;
    for (int j = 0; j < 9; j++) {
      const double tmp = A_(r, j);
      A_(r, j) = A_(pi, j);
      A_(pi, j) = tmp;
    }
    for (int i = 0; i < 7; i++) {
      const double tmp = A_(i, r);
      A_(i, r) = A_(i, pj);
      A_(i, pj) = tmp;
    }
    {
      const int tmp = perm[r];
      perm[r] = perm[pj];
      perm[pj] = tmp;
    }
    const double ip = 1. / A_(r, r);
    for (int j = 0; j < 9; j++) A_(r, j) *= ip;
    for (int i = 0; i < 7; i++) {
      if (i == r) continue;
      const double f = A_(i, r);
      for (int j = 0; j < 9; j++) A_(i, j) -= f * A_(r, j);
    }
  }
  double f1[9], f2[9];
  for (int j = 0; j < 9; j++) f1[j] = f2[j] = 0;
  for (int r = 0; r < 7; r++) {
    f1[perm[r]] = -A_(r, 7);
    f2[perm[r]] = -A_(r, 8);
  }
  f1[perm[7]] = 1;
  f2[perm[8]] = 1;
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
  const int n = cubic_real_roots(c, roots);
  for (int k = 0; k < n; k++) {
    double* F = Fm + 9 * k;
    double lambda = roots[k], mu = 1.;
    const double s = f1[8] * roots[k] + f2[8];
    if (fabs(s) > kDblEps) {
      mu = 1. / s;
      lambda *= mu;
      F[8] = 1.;
    } else {
      F[8] = 0.;
    }
    for (int i = 0; i < 8; i++) F[i] = f1[i] * lambda + f2[i] * mu;
    double G[9];
    for (int rr = 0; rr < 3; rr++) {
      G[3 * rr] = F[3 * rr] * s1;
      G[3 * rr + 1] = F[3 * rr + 1] * s1;
      G[3 * rr + 2] = F[3 * rr] * (-s1 * c1x) + F[3 * rr + 1] * (-s1 * c1y) + F[3 * rr + 2];
    }
    for (int cc = 0; cc < 3; cc++) {
      F[cc] = s2 * G[cc];
      F[3 + cc] = s2 * G[3 + cc];
      F[6 + cc] = (-s2 * c2x) * G[cc] + (-s2 * c2y) * G[3 + cc] + G[6 + cc];
    }
    if (fabs(F[8]) > kFltEps) {
      const double sc = 1. / F[8];
      for (int i = 0; i < 9; i++) F[i] *= sc;
    }
  }
  return n;
#undef A_
}

__device__ __forceinline__ float epipolar_error(const double* F, float2 p1, float2 p2) {  // FMEstimatorCallback::computeError
  const double x1 = (double)p1.x, y1 = (double)p1.y, x2 = (double)p2.x, y2 = (double)p2.y;
  double a = F[0] * x1 + F[1] * y1 + F[2];
  double b = F[3] * x1 + F[4] * y1 + F[5];
  double c = F[6] * x1 + F[7] * y1 + F[8];
  const double s2 = 1. / (a * a + b * b);
  const double d2 = x2 * a + y2 * b + c;
  a = F[0] * x2 + F[3] * y2 + F[6];
  b = F[1] * x2 + F[4] * y2 + F[7];
  c = F[2] * x2 + F[5] * y2 + F[8];
  const double s1 = 1. / (a * a + b * b);
  const double d1 = x1 * a + y1 * b + c;
  const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
  return (float)(e1 > e2 ? e1 : e2);
}

// Hypotheses of one chunk of K <= kFmChunk RANSAC iterations of every active problem: 64 threads per problem.
// pts1 / pts2: [B][stride] float2; models [B][kFmChunk][27], nmodels [B][kFmChunk] (-1 = getSubset failed).
__global__ void __launch_bounds__(kFmChunk) k_fmat_hyp(FmProblem* __restrict__ prob, const float2* __restrict__ pts1,
                                                       const float2* __restrict__ pts2, int stride, int K, double* __restrict__ models,
                                                       int* __restrict__ nmodels) {
  __shared__ int s_idx[kFmChunk][7];
  __shared__ int s_nm[kFmChunk];
  __shared__ unsigned long long s_state[kFmChunk + 1];
  __shared__ int s_bad, s_fail;
  __shared__ double s_a[63 * kFmChunk];
  const int b = blockIdx.x, tid = threadIdx.x;
  FmProblem P = prob[b];
  if (!P.active) return;
  const float2* m1 = pts1 + (size_t)b * stride;
  const float2* m2 = pts2 + (size_t)b * stride;
  double* Fm = models + (size_t)b * kFmChunk * 27;
  // getSubset for the K iterations.  Drawing is sequential but needs no data; checkSubset needs the points but is independent per
  // subset.  So lane 0 draws all subsets as if none were rejected, the lanes test them in parallel, and only if one is rejected
  // (collinear points: rare) lane 0 redraws that one the serial way and the subsets after it are drawn again.
  int start = 0;
  for (;;) {
    if (tid == 0) {
      unsigned long long rng = start == 0 ? P.rng : s_state[start];
      for (int k = start; k < K; k++) {
        int idx[7] = {0, 0, 0, 0, 0, 0, 0};
        draw_subset(rng, P.n, idx);
        for (int i = 0; i < 7; i++) s_idx[k][i] = idx[i];
        s_state[k + 1] = rng;
        s_nm[k] = 0;
      }
      s_bad = K;
    }
    __syncthreads();
    if (tid >= start && tid < K) {
      int idx[7];
      for (int i = 0; i < 7; i++) idx[i] = s_idx[tid][i];
      if (!subset_ok(m1, m2, idx)) atomicMin(&s_bad, tid);
    }
    __syncthreads();
    const int bad = s_bad;
    if (bad >= K) break;
    if (tid == 0) {  // attempts 2 .. 10000 of that getSubset call
      unsigned long long rng = s_state[bad + 1];
      bool ok = false;
      int idx[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int iters = 1; iters < P.max_attempts && !ok; ++iters) {
        draw_subset(rng, P.n, idx);
        ok = subset_ok(m1, m2, idx);
      }
      for (int i = 0; i < 7; i++) s_idx[bad][i] = idx[i];
      s_state[bad + 1] = rng;
      s_fail = ok ? 0 : 1;
      if (!ok)
        for (int k = bad; k < K; k++) s_nm[k] = -1;  // the RANSAC loop ends at this iteration
    }
    __syncthreads();
    if (s_fail) break;
    start = bad + 1;
    if (start >= K) break;
  }
  if (tid == 0) prob[b].rng = s_state[K];
  __syncthreads();
  if (tid < K) {
    int nm = s_nm[tid];
    if (nm == 0) {
      int idx[7];
      for (int i = 0; i < 7; i++) idx[i] = s_idx[tid][i];
      double F[27];
      nm = seven_point(m1, m2, idx, F, s_a + tid);
      for (int i = 0; i < 9 * nm; i++) Fm[(size_t)tid * 27 + i] = F[i];
    }
    nmodels[(size_t)b * kFmChunk + tid] = nm;
  }
}

// Inlier counts of the up to 3 K models of a chunk: one wave per (iteration, model), lanes over the points; good [B][kFmChunk][3].
__global__ void __launch_bounds__(kFmThreads) k_fmat_count(const FmProblem* __restrict__ prob, const float2* __restrict__ pts1,
                                                          const float2* __restrict__ pts2, int stride, int K,
                                                          const double* __restrict__ models, const int* __restrict__ nmodels,
                                                          int* __restrict__ good) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int q = blockIdx.x * (kFmThreads / 64) + (threadIdx.x >> 6);
  const FmProblem P = prob[b];
  if (!P.active || P.lmeds || q >= 3 * K) return;
  const int k = q / 3, m = q - 3 * k;
  if (m >= nmodels[(size_t)b * kFmChunk + k]) return;  // wave-uniform
  const float2* m1 = pts1 + (size_t)b * stride;
  const float2* m2 = pts2 + (size_t)b * stride;
  double F[9];
  for (int i = 0; i < 9; i++) F[i] = models[((size_t)b * kFmChunk + k) * 27 + 9 * m + i];
  int cnt = 0;
  for (int i = lane; i < P.n; i += 64) cnt += epipolar_error(F, m1[i], m2[i]) <= P.t2 ? 1 : 0;
  for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
  if (lane == 0) good[((size_t)b * kFmChunk + k) * 3 + m] = cnt;
}

// LMedS problems (8 .. 14 points): the median error of every model of the chunk, one thread per (iteration, model);
// med [B][kFmChunk][3] (what std::nth_element leaves at position n / 2: the (n / 2)-th smallest of the float errors)
__global__ void __launch_bounds__(kFmThreads) k_fmat_median(const FmProblem* __restrict__ prob, const float2* __restrict__ pts1,
                                                           const float2* __restrict__ pts2, int stride, int K,
                                                           const double* __restrict__ models, const int* __restrict__ nmodels,
                                                           float* __restrict__ med) {
  const int b = blockIdx.y, q = blockIdx.x * kFmThreads + threadIdx.x;
  const FmProblem P = prob[b];
  if (!P.active || !P.lmeds || q >= 3 * K) return;
  const int k = q / 3, m = q - 3 * k;
  if (m >= nmodels[(size_t)b * kFmChunk + k]) return;
  double F[9];
  for (int i = 0; i < 9; i++) F[i] = models[((size_t)b * kFmChunk + k) * 27 + 9 * m + i];
  float e[16];
  for (int i = 0; i < P.n; i++) e[i] = epipolar_error(F, pts1[(size_t)b * stride + i], pts2[(size_t)b * stride + i]);
  for (int i = 1; i < P.n; i++) {
    const float v = e[i];
    int j = i;
    for (; j > 0 && e[j - 1] > v; j--) e[j] = e[j - 1];
    e[j] = v;
  }
  med[((size_t)b * kFmChunk + k) * 3 + m] = e[P.n / 2];
}

// best[b] = the model accepted last in this chunk (keep[b] = iteration * 3 + model, or -1: nothing new for problem b)
__global__ void k_fmat_keep(const int* __restrict__ keep, const double* __restrict__ models, double* __restrict__ best, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B || keep[b] < 0) return;
  for (int i = 0; i < 9; i++) best[(size_t)b * 9 + i] = models[(size_t)b * kFmChunk * 27 + (size_t)keep[b] * 9 + i];
}

// mask[i] = inlier of the chosen model; sel [B] >= 0: a model exists, -1: none (mask 0), -2: problem passed through (mask 1);
// count [B] (zeroed by the caller) receives the number of ones
__global__ void __launch_bounds__(kFmThreads) k_fmat_mask(const FmProblem* __restrict__ prob, const int* __restrict__ sel,
                                                         const double* __restrict__ best, const float2* __restrict__ pts1,
                                                         const float2* __restrict__ pts2, int stride, uint8_t* __restrict__ mask,
                                                         int* __restrict__ count) {
  const int b = blockIdx.y, i = blockIdx.x * kFmThreads + threadIdx.x;
  const FmProblem P = prob[b];
  uint8_t v = 0;
  if (i < P.n) {
    if (sel[b] >= 0) {
      double F[9];
      for (int k = 0; k < 9; k++) F[k] = best[(size_t)b * 9 + k];
      v = epipolar_error(F, pts1[(size_t)b * stride + i], pts2[(size_t)b * stride + i]) <= P.t2 ? 1 : 0;
    } else if (sel[b] == -2) {
      v = 1;
    }
    mask[(size_t)b * stride + i] = v;
  }
  const unsigned long long ball = __ballot(v != 0);
  if ((threadIdx.x & 63) == 0 && ball) atomicAdd(count + b, __popcll(ball));
}

int update_num_iters(double p, double ep, int max_iters) {  // RANSACUpdateNumIters(p, ep, 7, maxIters), host libm like the oracle
  p = p < 0 ? 0 : (p > 1 ? 1 : p);
  ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
  double num = 1 - p > 2.2250738585072014e-308 ? 1 - p : 2.2250738585072014e-308;
  double denom = 1 - std::pow(1 - ep, 7);
  if (denom < 2.2250738585072014e-308) return 0;
  num = std::log(num);
  denom = std::log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::lrint(num / denom);
}

}  // namespace

struct gfs_fmat {
  int device = 0, max_points = 0, max_batch = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
  DevBuf<FmProblem> d_prob;
  DevBuf<float2> d_p1, d_p2;
  DevBuf<double> d_models, d_best;
  DevBuf<int> d_nm, d_good, d_sel, d_keep, d_cnt;
  PinBuf<int> h_cnt;
  DevBuf<float> d_med;
  PinBuf<float> h_med;
  DevBuf<uint8_t> d_mask;
  PinBuf<FmProblem> h_prob;
  PinBuf<float2> h_p1, h_p2;
  PinBuf<double> h_best;
  PinBuf<int> h_nm, h_good, h_sel, h_keep;
  PinBuf<uint8_t> h_mask;
};

extern "C" {

int gfs_fmat_create(int device, int max_points, int max_batch, gfs_fmat** out) {
  GFS_REQUIRE(out && max_points >= 15 && max_batch > 0, GFS_ERR_INVALID_ARG, "gfs_fmat_create: invalid argument");
  *out = nullptr;
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  auto h = std::make_unique<gfs_fmat>();
  h->device = device;
  h->max_points = max_points;
  h->max_batch = max_batch;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t B = max_batch, NP = B * max_points;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_prob.alloc(B));
  A(h->d_p1.alloc(NP));
  A(h->d_p2.alloc(NP));
  A(h->d_models.alloc(B * kFmChunk * 27));
  A(h->d_best.alloc(B * 9));
  A(h->d_nm.alloc(B * kFmChunk));
  A(h->d_good.alloc(B * kFmChunk * 3));
  A(h->d_sel.alloc(B));
  A(h->d_med.alloc(B * kFmChunk * 3));
  A(h->h_med.alloc(B * kFmChunk * 3));
  A(h->d_keep.alloc(B));
  A(h->d_cnt.alloc(B));
  A(h->h_cnt.alloc(B));
  A(h->d_mask.alloc(NP));
  A(h->h_prob.alloc(B));
  A(h->h_p1.alloc(NP));
  A(h->h_p2.alloc(NP));
  A(h->h_best.alloc(B * 9));
  A(h->h_nm.alloc(B * kFmChunk));
  A(h->h_good.alloc(B * kFmChunk * 3));
  A(h->h_sel.alloc(B));
  A(h->h_keep.alloc(B));
  A(h->h_mask.alloc(NP));
#undef A
  if (rc) {
    (void)hipStreamDestroy(h->stream);
    return rc;
  }
  *out = h.release();
  return GFS_OK;
}

void gfs_fmat_destroy(gfs_fmat* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

}  // extern "C"

namespace {
// The RANSAC / LMedS rounds on device-resident points d1 / d2 [B][S]; h->h_prob [b] = {n, active, ...} prepared by the caller
// (active = 0 and h_sel = -2: the problem is passed through with an all-ones mask).  Leaves the mask in d_mask [B][S] (device),
// the inlier counts in h->h_cnt and the best models in h->h_best.
int fmat_run(gfs_fmat* h, int B, int S, const float2* d1, const float2* d2, double confidence, int max_iters, uint8_t* d_mask,
             std::vector<int>& max_good) {
  hipStream_t s = h->stream;
  bool any_lmeds = false;
  int nmax = 1, remaining = 0;
  std::vector<int> n_points(B);
  for (int b = 0; b < B; b++) {
    n_points[b] = h->h_prob.p[b].n;
    any_lmeds |= h->h_prob.p[b].lmeds != 0 && h->h_prob.p[b].active;
    nmax = n_points[b] > nmax ? n_points[b] : nmax;
    remaining += h->h_prob.p[b].active ? 1 : 0;
  }
  // host replay state of RANSACPointSetRegistrator::run per problem
  std::vector<int> niters(B, max_iters > 1 ? max_iters : 1), iter(B, 0), done(B, 0), best_here(B, 0);
  max_good.assign(B, 0);
  for (int b = 0; b < B; b++) done[b] = h->h_prob.p[b].active ? 0 : 1;
  std::vector<double> min_median(B, 1.7976931348623157e308);
  for (int b = 0; b < B; b++)
    if (h->h_prob.p[b].lmeds) {  // LMeDSPointSetRegistrator::run: budget from a 45 % outlier assumption, at least 3
      niters[b] = update_num_iters(confidence, 0.45, 1000);
      if (niters[b] < 3) niters[b] = 3;
    }
  int K = 16;  // the first chunk is short: with a quarter of outliers or fewer the budget drops below 16 after the first good model
  while (remaining > 0) {
    GFS_HIP(hipMemcpyAsync(h->d_prob.p, h->h_prob.p, B * sizeof(FmProblem), hipMemcpyHostToDevice, s));
    GFS_LAUNCH("k_fmat_hyp", k_fmat_hyp, dim3(B), dim3(kFmChunk), 0, s, h->d_prob.p, d1,
               d2, S, K, h->d_models.p, h->d_nm.p);
    GFS_LAUNCH("k_fmat_count", k_fmat_count, dim3(gfs::div_up(3 * K, kFmThreads / 64), B), dim3(kFmThreads), 0, s,
               (const FmProblem*)h->d_prob.p, d1, d2, S, K, (const double*)h->d_models.p,
               (const int*)h->d_nm.p, h->d_good.p);
    if (any_lmeds) {
      GFS_LAUNCH("k_fmat_median", k_fmat_median, dim3(gfs::div_up(3 * K, kFmThreads), B), dim3(kFmThreads), 0, s,
                 (const FmProblem*)h->d_prob.p, d1, d2, S, K, (const double*)h->d_models.p,
                 (const int*)h->d_nm.p, h->d_med.p);
      GFS_HIP(hipMemcpyAsync(h->h_med.p, h->d_med.p, (size_t)B * kFmChunk * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    GFS_HIP(hipMemcpyAsync(h->h_prob.p, h->d_prob.p, B * sizeof(FmProblem), hipMemcpyDeviceToHost, s));
    GFS_HIP(hipMemcpyAsync(h->h_nm.p, h->d_nm.p, (size_t)B * kFmChunk * sizeof(int), hipMemcpyDeviceToHost, s));
    GFS_HIP(hipMemcpyAsync(h->h_good.p, h->d_good.p, (size_t)B * kFmChunk * 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    GFS_HIP(hipStreamSynchronize(s));
    std::fill(best_here.begin(), best_here.end(), 0);
    for (int b = 0; b < B; b++) {
      if (done[b]) continue;
      const int n = n_points[b];
      for (int k = 0; k < K && !done[b]; k++) {
        if (iter[b] >= niters[b]) {
          done[b] = 1;
          break;
        }
        const int nm = h->h_nm.p[(size_t)b * kFmChunk + k];
        if (nm < 0) {  // getSubset failed: the loop ends (with no model at all if it was the first iteration)
          done[b] = 1;
          break;
        }
        for (int m = 0; m < nm; m++) {
          if (h->h_prob.p[b].lmeds) {
            const double median = h->h_med.p[((size_t)b * kFmChunk + k) * 3 + m];
            if (median < min_median[b]) {
              min_median[b] = median;
              h->h_sel.p[b] = k * 3 + m;
              best_here[b] = 1;
            }
            continue;
          }
          const int g = h->h_good.p[((size_t)b * kFmChunk + k) * 3 + m];
          if (g > (max_good[b] > 6 ? max_good[b] : 6)) {
            max_good[b] = g;
            h->h_sel.p[b] = k * 3 + m;
            best_here[b] = 1;
            niters[b] = update_num_iters(confidence, (double)(n - g) / n, niters[b]);
          }
        }
        iter[b]++;
      }
      if (!done[b] && iter[b] >= niters[b]) done[b] = 1;
      if (done[b]) {
        h->h_prob.p[b].active = 0;
        remaining--;
      }
    }
    // the accepted models of this chunk stay on the device (the next chunk overwrites d_models)
    bool any_new = false;
    for (int b = 0; b < B; b++) {
      h->h_keep.p[b] = best_here[b] ? h->h_sel.p[b] : -1;
      any_new |= best_here[b] != 0;
    }
    if (any_new) {
      GFS_HIP(hipMemcpyAsync(h->d_keep.p, h->h_keep.p, B * sizeof(int), hipMemcpyHostToDevice, s));
      GFS_LAUNCH("k_fmat_keep", k_fmat_keep, dim3(gfs::div_up(B, 64)), dim3(64), 0, s, (const int*)h->d_keep.p,
                 (const double*)h->d_models.p, h->d_best.p, B);
      GFS_HIP(hipStreamSynchronize(s));  // h_keep is rewritten by the next round
    }
    K = kFmChunk;
  }
  for (int b = 0; b < B; b++)
    if (h->h_prob.p[b].lmeds && h->h_sel.p[b] >= 0) {  // inliers of the least-median model: within sigma of it
      double sigma = 2.5 * 1.4826 * (1 + 5. / (n_points[b] - 7)) * std::sqrt(min_median[b]);
      if (sigma < 0.001) sigma = 0.001;
      h->h_prob.p[b].t2 = (float)(sigma * sigma);
    }
  GFS_HIP(hipMemcpyAsync(h->d_prob.p, h->h_prob.p, B * sizeof(FmProblem), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_sel.p, h->h_sel.p, B * sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemsetAsync(h->d_cnt.p, 0, B * sizeof(int), s));
  GFS_LAUNCH("k_fmat_mask", k_fmat_mask, dim3(gfs::div_up(nmax, kFmThreads), B), dim3(kFmThreads), 0, s, (const FmProblem*)h->d_prob.p,
             (const int*)h->d_sel.p, (const double*)h->d_best.p, d1, d2, S, d_mask, h->d_cnt.p);
  GFS_HIP(hipMemcpyAsync(h->h_cnt.p, h->d_cnt.p, B * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_best.p, h->d_best.p, (size_t)B * 9 * sizeof(double), hipMemcpyDeviceToHost, s));
  return GFS_OK;
}

// F [B][9] and n_inliers [B] from the state fmat_run left (after the stream was synchronised)
void fmat_results(gfs_fmat* h, int B, const std::vector<int>& max_good, double* F, int32_t* n_inliers) {
  for (int b = 0; b < B; b++) {
    const FmProblem& P = h->h_prob.p[b];
    int c = h->h_sel.p[b] == -2 ? P.n : max_good[b];
    bool have = max_good[b] > 0;
    if (P.lmeds && h->h_sel.p[b] >= 0) {  // LMedS counts from the mask; "result = count >= modelPoints" decides whether F is returned
      c = h->h_cnt.p[b];
      have = c >= 7;
    }
    n_inliers[b] = c;
    if (F) {
      if (have && h->h_sel.p[b] >= 0)
        memcpy(F + 9 * (size_t)b, h->h_best.p + (size_t)b * 9, 9 * sizeof(double));
      else
        memset(F + 9 * (size_t)b, 0, 9 * sizeof(double));
    }
  }
}
}  // namespace

extern "C" {

int gfs_find_fundamental_ransac(gfs_fmat* h, int B, const int32_t* n_points, const float* const* pts1, const float* const* pts2,
                                double threshold, double confidence, int max_iters, uint8_t* const* mask, double* F,
                                int32_t* n_inliers) {
  GFS_REQUIRE(h && n_points && pts1 && pts2 && mask && n_inliers && B > 0, GFS_ERR_INVALID_ARG,
              "gfs_find_fundamental_ransac: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_find_fundamental_ransac: batch %d exceeds capacity %d", B, h->max_batch);
  if (threshold <= 0) threshold = 3;
  if (confidence < kDblEps || confidence > 1 - kDblEps) confidence = 0.99;
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int S = h->max_points;
  for (int b = 0; b < B; b++) {
    const int n = n_points[b];
    GFS_REQUIRE(n <= S, GFS_ERR_CAPACITY, "gfs_find_fundamental_ransac: problem %d has %d points, capacity %d", b, n, S);
    GFS_REQUIRE(n >= 8, GFS_ERR_UNSUPPORTED, "gfs_find_fundamental_ransac: problem %d has %d points (at least 8 needed)", b, n);
    GFS_REQUIRE(pts1[b] && pts2[b] && mask[b], GFS_ERR_INVALID_ARG, "gfs_find_fundamental_ransac: problem %d has NULL arrays", b);
    memcpy(h->h_p1.p + (size_t)b * S, pts1[b], (size_t)n * sizeof(float2));
    memcpy(h->h_p2.p + (size_t)b * S, pts2[b], (size_t)n * sizeof(float2));
    h->h_prob.p[b] = FmProblem{n, 1, 0xffffffffffffffffull, (float)(threshold * threshold), n < 15 ? 1 : 0, n < 15 ? 1000 : 10000};
    h->h_sel.p[b] = -1;
  }
  hipStream_t s = h->stream;
  const size_t NP = (size_t)B * S;
  GFS_HIP(hipMemcpyAsync(h->d_p1.p, h->h_p1.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_p2.p, h->h_p2.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
  std::vector<int> max_good;
  const int rc = fmat_run(h, B, S, (const float2*)h->d_p1.p, (const float2*)h->d_p2.p, confidence, max_iters, h->d_mask.p, max_good);
  if (rc) return rc;
  GFS_HIP(hipMemcpyAsync(h->h_mask.p, h->d_mask.p, NP, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int b = 0; b < B; b++) memcpy(mask[b], h->h_mask.p + (size_t)b * S, (size_t)n_points[b]);
  fmat_results(h, B, max_good, F, n_inliers);
  return GFS_OK;
}

int gfs_find_fundamental_ransac_device(gfs_fmat* h, int B, int stride, const void* dev_n, const void* dev_pts1, const void* dev_pts2,
                                       double threshold, double confidence, int max_iters, void* dev_mask, double* F,
                                       int32_t* n_inliers) {
  GFS_REQUIRE(h && dev_n && dev_pts1 && dev_pts2 && dev_mask && n_inliers && B > 0 && stride > 0, GFS_ERR_INVALID_ARG,
              "gfs_find_fundamental_ransac_device: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_find_fundamental_ransac_device: batch %d exceeds capacity %d", B, h->max_batch);
  if (threshold <= 0) threshold = 3;
  if (confidence < kDblEps || confidence > 1 - kDblEps) confidence = 0.99;
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  GFS_HIP(hipMemcpyAsync(h->h_cnt.p, dev_n, B * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int b = 0; b < B; b++) {
    const int n = h->h_cnt.p[b];
    GFS_REQUIRE(n >= 0 && n <= stride, GFS_ERR_INVALID_ARG, "gfs_find_fundamental_ransac_device: problem %d has %d points, stride %d", b, n, stride);
    // SearchByProjectionWithOF runs the check only for more than 8 points (src/ORBmatcher.cc:2397, 2461): fewer pass through
    const int run = n > 8 ? 1 : 0;
    h->h_prob.p[b] = FmProblem{n, run, 0xffffffffffffffffull, (float)(threshold * threshold), n < 15 ? 1 : 0, n < 15 ? 1000 : 10000};
    h->h_sel.p[b] = run ? -1 : -2;
  }
  std::vector<int> max_good;
  const int rc = fmat_run(h, B, stride, (const float2*)dev_pts1, (const float2*)dev_pts2, confidence, max_iters, (uint8_t*)dev_mask, max_good);
  if (rc) return rc;
  GFS_HIP(hipStreamSynchronize(s));
  fmat_results(h, B, max_good, F, n_inliers);
  return GFS_OK;
}

}  // extern "C"
