// ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (reference
// src/ORBmatcher.cc:1853-2063) on MI355X, single-camera frames (Nleft == -1), a batch of frame pairs per launch.
//
// One 256-thread workgroup per frame pair:
//   0. Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:734-761, 1073-1084): the 64 x 48 grid of the current frame as
//      a CSR table in LDS (cells keep ascending key-point order, like the reference's push_back);
//   A. in parallel over the last frame's map points: projection with Sophus::SE3f float arithmetic
//      (Thirdparty/Sophus/sophus/so3.hpp:358-367), Frame::GetFeaturesInArea (src/Frame.cc:1007-1071) and the Hamming
//      distances (ORBmatcher::DescriptorDistance :2536-2550) of the window's candidates, kept in visiting order;
//   B. the reference loop is order dependent (a key-point taken by an earlier map point with observations is skipped by
//      the later ones, :1914-1915), so the assignment itself runs in map-point order on one wave: the lanes check one
//      candidate each, a wave-wide arg-min (first minimum wins, as `dist < bestDist` does) picks the match;
//   C. rotation histogram + ComputeThreeMaxima (:2500-2532) and the final NULL-ing of inconsistent matches.
// Mode 1 is ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th, ...) (:43-206): the projections come
// with the map points (Frame::isInFrustum), the window depends on the viewing angle, and phase B keeps best and second best
// (two arg-min rounds) for the ratio test; no rotation histogram.
// Results are bit-exact integer work; the float decisions (image bounds, window membership, level ranges, histogram bins)
// use the reference's float expressions (the library is built with -ffp-contract=off).
#include <memory>
#include <type_traits>
#include <mutex>

#include "gfs_common.hpp"

namespace {

constexpr int kGridCols = 64, kGridRows = 48, kCells = kGridCols * kGridRows;
constexpr int kHisto = 30, kThHigh = 100;
constexpr int kSbpThreads = 1024;  // a thread per map point while their candidates are enumerated (dependent gathers: parallelism hides them)
constexpr int kSbpMaxCur = 4096;   // key-points of the current frame (LDS tables)
constexpr int kSbpMaxLast = 8192;  // map points of the last frame
constexpr int kCand = 64;          // candidates kept per map point (more: the assignment pass re-enumerates them)
constexpr int kChunk = 32;         // map points per step of the assignment pass (their candidate lists are prefetched into LDS)

struct SbpPair {
  int n_last, n_cur, n_levels, mono, check_orientation;
  int mode;  // 0 = frame to frame (:1853-2063), 1 = map points with Frame::isInFrustum projections (:43-206)
  float nn_ratio;
  float Tcw_q[4], Tcw_t[3], Tlw_q[4], Tlw_t[3];
  float fx, fy, cx, cy, bf, b, min_x, max_x, min_y, max_y, grid_w_inv, grid_h_inv, th;
  float scale[16];
};

struct SbpView {
  const float* last_xw;
  const uint8_t* last_desc;
  const int* last_octave;
  const float* last_angle;
  const uint8_t* last_has_obs;
  const gfs_keypoint* cur_kp;
  const float* cur_ur;
  const uint8_t* cur_desc;
  const uint8_t* cur_has_obs;
};

// SO3f * p (so3.hpp:358-367): uv = q.vec x p; uv += uv; p + w * uv + q.vec x uv
__device__ __forceinline__ void so3_act(const float* q, const float* p, float* o) {
  float uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int k = 0; k < 3; k++) uv[k] += uv[k];
  const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int k = 0; k < 3; k++) o[k] = (p[k] + q[3] * uv[k]) + c[k];
}

__device__ __forceinline__ int hamming256(const uint8_t* a, const uint8_t* b) {
  const uint4* pa = reinterpret_cast<const uint4*>(a);
  const uint4* pb = reinterpret_cast<const uint4*>(b);
  const uint4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
  return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
         __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

struct Proj {
  float u, v, radius, ur, ur_gate;  // window centre / half size, predicted right-image x and its tolerance
  int minLevel, maxLevel, x0, x1, y0, y1;
  bool ok;
};

// projection of map point l + the cell range of its search window (everything before the candidate loop of :1878-1911)
__device__ __forceinline__ Proj sbp_project(const SbpPair& P, const SbpView& V, int l, bool bForward, bool bBackward) {
  Proj R;
  R.ok = false;
  if (P.mode == 1) {  // the projection was done by Frame::isInFrustum: (mTrackProjX, mTrackProjY, mTrackProjXR), level, viewCos
    R.u = V.last_xw[3 * l];
    R.v = V.last_xw[3 * l + 1];
    R.ur = V.last_xw[3 * l + 2];
    const int level = V.last_octave[l];
    float r = (double)V.last_angle[l] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:250-255)
    if ((double)P.th != 1.0) r *= P.th;
    R.radius = r * P.scale[level];
    R.ur_gate = R.radius;
    R.minLevel = level - 1;
    R.maxLevel = level;
  } else {
    float x3Dc[3];
    so3_act(P.Tcw_q, V.last_xw + 3 * l, x3Dc);
    for (int k = 0; k < 3; k++) x3Dc[k] += P.Tcw_t[k];
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (invzc < 0) return R;
    R.u = P.fx * x3Dc[0] / x3Dc[2] + P.cx;
    R.v = P.fy * x3Dc[1] / x3Dc[2] + P.cy;
    if (R.u < P.min_x || R.u > P.max_x) return R;
    if (R.v < P.min_y || R.v > P.max_y) return R;
    const int oct = V.last_octave[l];
    R.radius = P.th * P.scale[oct];
    R.ur = R.u - P.bf * invzc;
    R.ur_gate = R.radius;
    if (bForward) {
      R.minLevel = oct;
      R.maxLevel = -1;
    } else if (bBackward) {
      R.minLevel = 0;
      R.maxLevel = oct;
    } else {
      R.minLevel = oct - 1;
      R.maxLevel = oct + 1;
    }
  }
  R.x0 = max(0, (int)floorf((R.u - P.min_x - R.radius) * P.grid_w_inv));
  if (R.x0 >= kGridCols) return R;
  R.x1 = min(kGridCols - 1, (int)ceilf((R.u - P.min_x + R.radius) * P.grid_w_inv));
  if (R.x1 < 0) return R;
  R.y0 = max(0, (int)floorf((R.v - P.min_y - R.radius) * P.grid_h_inv));
  if (R.y0 >= kGridRows) return R;
  R.y1 = min(kGridRows - 1, (int)ceilf((R.v - P.min_y + R.radius) * P.grid_h_inv));
  if (R.y1 < 0) return R;
  R.ok = true;
  return R;
}

// The current frame's key-points as the candidate loop needs them, in LDS (filled with the grid): position, octave, mvuRight, and
// whether a map point with observations sits there on entry.
struct SbpCur {
  const float* x;
  const float* y;
  const uint8_t* oct;
  const float* ur;
  const uint8_t* has_obs;
};

// GetFeaturesInArea + the static filters of the candidate loop, in the reference's visiting order (ix, iy, cell order).
// f(i2, dist) is called for every candidate that survives them; returns whether vIndices2 was non-empty.
// A cell is c = ix * kGridRows + iy and the items are stored by cell, so the cells iy = y0 .. y1 of one grid column are ONE
// contiguous run of s_items: a window is x1 - x0 + 1 runs (<= 13 for the largest radius), not (x1 - x0 + 1) (y1 - y0 + 1) cells of
// which two thirds are empty.  The items are walked four at a time: their fields come from LDS, the descriptors of the survivors
// are fetched together -- one round trip to memory per four items, for the survivors only.
template <class F>
__device__ __forceinline__ bool sbp_candidates(const SbpPair& P, const SbpView& V, const SbpCur& C, const Proj& R, int l,
                                               const unsigned short* s_start, const unsigned short* s_items, F&& f) {
  const bool bCheckLevels = (R.minLevel > 0) || (R.maxLevel >= 0);
  bool any = false;
  const uint4* dl = reinterpret_cast<const uint4*>(V.last_desc + 32 * (size_t)l);
  const uint4 a0 = dl[0], a1 = dl[1];
  int ix = R.x0;
  int k = s_start[ix * kGridRows + R.y0], kend = s_start[ix * kGridRows + R.y1 + 1];
  bool more = true;
  auto next_item = [&]() {  // the next key-point index of the window in visiting order, or -1
    while (more && k >= kend) {
      if (++ix > R.x1) {
        more = false;
        break;
      }
      k = s_start[ix * kGridRows + R.y0];
      kend = s_start[ix * kGridRows + R.y1 + 1];
    }
    return more ? (int)s_items[k++] : -1;
  };
  while (more) {
    int i2[4];
    bool pass[4];
    uint4 b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      i2[u] = next_item();
      pass[u] = false;
      if (i2[u] < 0) continue;
      const int j = i2[u];
      if (bCheckLevels) {
        const int oct = C.oct[j];
        if (oct < R.minLevel) continue;
        if (R.maxLevel >= 0 && oct > R.maxLevel) continue;
      }
      const float distx = C.x[j] - R.u, disty = C.y[j] - R.v;
      if (!(fabsf(distx) < R.radius && fabsf(disty) < R.radius)) continue;
      any = true;
      if (C.has_obs[j]) continue;  // a map point with observations was there on entry: never replaced
      const float ur2 = C.ur[j];
      if (ur2 > 0) {
        const float er = fabsf(R.ur - ur2);
        if (er > R.ur_gate) continue;
      }
      pass[u] = true;
      const uint4* dc = reinterpret_cast<const uint4*>(V.cur_desc + 32 * (size_t)j);
      b0[u] = dc[0];
      b1[u] = dc[1];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (pass[u])
        f(i2[u], __popc(a0.x ^ b0[u].x) + __popc(a0.y ^ b0[u].y) + __popc(a0.z ^ b0[u].z) + __popc(a0.w ^ b0[u].w) + __popc(a1.x ^ b1[u].x) +
                     __popc(a1.y ^ b1[u].y) + __popc(a1.z ^ b1[u].z) + __popc(a1.w ^ b1[u].w));
  }
  return any;
}

// minimum of an unsigned value over the wave's 64 lanes, in every lane: quad permutes and row rotations (DPP) inside the rows of 16,
// the four row results through scalar registers
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
  auto dpp = [](unsigned x, auto ctrl) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, decltype(ctrl)::value, 0xf, 0xf, true); };
  v = min(v, dpp(v, std::integral_constant<int, 0xB1>{}));   // lane ^ 1
  v = min(v, dpp(v, std::integral_constant<int, 0x4E>{}));   // lane ^ 2
  v = min(v, dpp(v, std::integral_constant<int, 0x124>{}));  // row_ror:4
  v = min(v, dpp(v, std::integral_constant<int, 0x128>{}));  // row_ror:8
  const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                 r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(r0, r1), min(r2, r3));
}

#ifdef GFS_SBP_TIMING
#define SBP_T(k) { const long long _n = clock64(); if (threadIdx.x == 0) sbp_t[k] = _n; }
#else
#define SBP_T(k)
#endif
__global__ __launch_bounds__(kSbpThreads) void k_sbp(const SbpPair* __restrict__ pairs, const float* __restrict__ last_xw,
                                                     const uint8_t* __restrict__ last_desc, const int* __restrict__ last_octave,
                                                     const float* __restrict__ last_angle, const uint8_t* __restrict__ last_has_obs,
                                                     const gfs_keypoint* __restrict__ cur_kp, const float* __restrict__ cur_ur,
                                                     const uint8_t* __restrict__ cur_desc, const uint8_t* __restrict__ cur_has_obs,
                                                     int SL, int SC, unsigned* __restrict__ cand, int* __restrict__ cand_cnt,
                                                     int* __restrict__ lsel, int* __restrict__ cur_match, int* __restrict__ nmatches) {
  __shared__ unsigned short s_cell[kSbpMaxCur];
  __shared__ unsigned short s_start[kCells + 1];
  __shared__ unsigned short s_items[kSbpMaxCur];
  __shared__ short s_state[kSbpMaxCur];  // -1 untouched, -2 reset to NULL, >= 0 map point of last entry l
  // 16 KB used twice: the cell counters of the grid build, then the two candidate buffers of the assignment pass
  __shared__ unsigned s_pool[2 * kChunk * kCand];
  int* s_cnt = reinterpret_cast<int*>(s_pool);
  static_assert(2 * kChunk * kCand >= kCells, "the pool must hold the cell counters");
  __shared__ int s_hold[kSbpMaxCur];  // last taker of a key-point (assignment pass)
  __shared__ uint8_t s_hasobs[kSbpMaxLast];  // Observations() > 0 of the last frame's map points
  __shared__ uint8_t s_oct[kSbpMaxCur];      // octave of the current key-points (the ratio test of the map-point overload)
  __shared__ int s_scan[16];
  __shared__ float s_kx[kSbpMaxCur], s_ky[kSbpMaxCur], s_ur[kSbpMaxCur];  // key-point position, mvuRight
  __shared__ uint8_t s_ho[kSbpMaxCur];                                    // a map point with observations on entry
  __shared__ int s_hist[kHisto];
  __shared__ int s_ctl[4];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
#ifdef GFS_SBP_TIMING
  __shared__ long long sbp_t[8];
#endif
  SBP_T(0)
  // the pair's header lives in LDS: its scale table is indexed by the octave, and a private copy indexed at run time is a copy in
  // scratch memory -- every field access a round trip to the L1
  __shared__ SbpPair s_P;
  if (tid == 0) s_P = pairs[f];
  __syncthreads();
  const SbpPair& P = s_P;
  const SbpView V{last_xw + (size_t)f * SL * 3, last_desc + (size_t)f * SL * 32, last_octave + (size_t)f * SL,
                  last_angle + (size_t)f * SL, last_has_obs + (size_t)f * SL, cur_kp + (size_t)f * SC,
                  cur_ur + (size_t)f * SC, cur_desc + (size_t)f * SC * 32, cur_has_obs + (size_t)f * SC};
  unsigned* cnd = cand + (size_t)f * SL * kCand;
  int* ccnt = cand_cnt + (size_t)f * SL;
  int* sel = lsel + (size_t)f * SL;
  const int N = P.n_cur, NL = P.n_last;
  // ---- 0. grid (Frame::AssignFeaturesToGrid: the key-points of a cell in index order) + the key-point fields the windows look at
  for (int c = tid; c < kCells; c += kSbpThreads) s_cnt[c] = 0;
  if (tid < kHisto) s_hist[tid] = 0;
  __syncthreads();
  for (int i = tid; i < N; i += kSbpThreads) {
    const gfs_keypoint kp = V.cur_kp[i];
    const int px = (int)roundf((kp.x - P.min_x) * P.grid_w_inv), py = (int)roundf((kp.y - P.min_y) * P.grid_h_inv);
    const bool in = px >= 0 && px < kGridCols && py >= 0 && py < kGridRows;
    s_cell[i] = in ? (unsigned short)(px * kGridRows + py) : (unsigned short)0xffff;
    s_state[i] = -1;
    s_oct[i] = (uint8_t)kp.octave;
    s_kx[i] = kp.x;
    s_ky[i] = kp.y;
    s_ur[i] = V.cur_ur[i];
    s_ho[i] = V.cur_has_obs[i];
    if (in) atomicAdd(&s_cnt[px * kGridRows + py], 1);
  }
  __syncthreads();
  {
    constexpr int per = kCells / kSbpThreads;  // 3 (3072 cells over 1024 threads)
    int cnt[per], local = 0;
#pragma unroll
    for (int k = 0; k < per; k++) {
      cnt[k] = s_cnt[tid * per + k];
      local += cnt[k];
    }
    int incl = local;  // exclusive scan over the threads: shuffles inside the wave, the sixteen wave totals through LDS
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
      const int v = __shfl_up(incl, ofs, 64);
      if (lane >= ofs) incl += v;
    }
    if (lane == 63) s_scan[tid >> 6] = incl;
    __syncthreads();
    int run = incl - local;
    for (int w = 0; w < (tid >> 6); w++) run += s_scan[w];
#pragma unroll
    for (int k = 0; k < per; k++) {
      s_start[tid * per + k] = (unsigned short)run;
      s_cnt[tid * per + k] = 0;  // now the fill counter of the cell
      run += cnt[k];
    }
    if (tid == kSbpThreads - 1) s_start[kCells] = (unsigned short)run;
  }
  __syncthreads();
  for (int i = tid; i < N; i += kSbpThreads) {  // into the cell in arrival order ...
    const unsigned short c = s_cell[i];
    if (c == 0xffff) continue;
    s_items[s_start[c] + atomicAdd(&s_cnt[c], 1)] = (unsigned short)i;
  }
  __syncthreads();
  for (int c = tid; c < kCells; c += kSbpThreads) {  // ... then every cell in index order (a handful of items: insertion sort)
    const int b = s_start[c], e = s_start[c + 1];
    for (int a = b + 1; a < e; a++) {
      const unsigned short v = s_items[a];
      int q = a;
      while (q > b && s_items[q - 1] > v) {
        s_items[q] = s_items[q - 1];
        q--;
      }
      s_items[q] = v;
    }
  }
  const SbpCur Cur{s_kx, s_ky, s_oct, s_ur, s_ho};
  // twc = Tcw.inverse().translation(); tlc = Tlw * twc  (se3.hpp:208-211; the SO3 constructor re-normalises the conjugate)
  float qi[4] = {-P.Tcw_q[0], -P.Tcw_q[1], -P.Tcw_q[2], P.Tcw_q[3]};
  {
    const float nrm = sqrtf((qi[0] * qi[0] + qi[2] * qi[2]) + (qi[1] * qi[1] + qi[3] * qi[3]));
    for (int k = 0; k < 4; k++) qi[k] /= nrm;
  }
  const float mt[3] = {P.Tcw_t[0] * -1.0f, P.Tcw_t[1] * -1.0f, P.Tcw_t[2] * -1.0f};
  float twc[3], tlc[3];
  so3_act(qi, mt, twc);
  so3_act(P.Tlw_q, twc, tlc);
  for (int k = 0; k < 3; k++) tlc[k] += P.Tlw_t[k];
  const bool bForward = tlc[2] > P.b && !P.mono, bBackward = -tlc[2] > P.b && !P.mono;
  __syncthreads();
  SBP_T(1)
  // ---- A. candidates of every map point (parallel)
  for (int l = tid; l < NL; l += kSbpThreads) {
    s_hasobs[l] = V.last_has_obs[l];
    const Proj R = sbp_project(P, V, l, bForward, bBackward);
    int n = 0;
    bool any = false;
    if (R.ok)
      any = sbp_candidates(P, V, Cur, R, l, s_start, s_items, [&](int i2, int dist) {
        if (n < kCand) cnd[(size_t)l * kCand + n] = (unsigned)i2 | ((unsigned)dist << 16);
        n++;
      });
    ccnt[l] = any ? n : -1;  // -1: vIndices2.empty() -> continue
  }
  __syncthreads();
  __syncthreads();
  SBP_T(2)
  // ---- B. assignment.  The reference walks the map points in index order; each takes its nearest candidate (first minimum in
  //         visiting order) among the key-points that do not hold a map point with observations yet (:1914-1915), i.e. that no
  //         EARLIER map point with Observations() > 0 has taken.  That is a serial dictatorship, and with one common priority order
  //         (the index) its outcome is the fixed point of deferred acceptance: every map point proposes to its best key-point
  //         that is not held by a lower-index blocker, every key-point keeps the lowest-index blocking proposer, the displaced
  //         ones propose again.  A key-point's holder index only ever decreases, so a map point's options only shrink and a
  //         handful of fully parallel rounds (the longest displacement chain) replace the walk over ~1000 map points on one wave.
  //         s_block[i2] = lowest index of a proposer with Observations() > 0 (a map point without observations takes a
  //         key-point without blocking it: later ones overwrite it, and every take counts -- :1929-1933).
  int* s_block = reinterpret_cast<int*>(s_pool);  // [kSbpMaxCur] (the cell counters are no longer needed)
  static_assert(2 * kChunk * kCand >= kSbpMaxCur, "the pool must hold the blocker table");
  for (int i = tid; i < N; i += kSbpThreads) {
    s_block[i] = 0x7fffffff;
    s_hold[i] = -1;
  }
  // best available candidate of map point l (everything the sequential code decides for it, given who blocks what): -1 = none
  auto decide = [&](int l) {
    const int n = ccnt[l];
    if (n < 0) return -1;
    unsigned key = 0xffffffffu, key2 = 0xffffffffu;  // dist << 16 | visiting order: the first minimum wins; the second best (mode 1)
    int idx = -1, idx2 = -1;
    auto offer = [&](int i2, int dist, int order) {
      if (s_block[i2] < l) return;  // mvpMapPoints[i2] && Observations() > 0 -> skip (:1914-1915)
      const unsigned kk = ((unsigned)dist << 16) | (unsigned)min(order, 0xffff);
      if (kk < key) {
        key2 = key;
        idx2 = idx;
        key = kk;
        idx = i2;
      } else if (kk < key2) {
        key2 = kk;
        idx2 = i2;
      }
    };
    if (n <= kCand) {
      for (int j = 0; j < n; j += 4) {  // (a row of the list is 256 bytes: four entries a load)
        const uint4 e4 = *reinterpret_cast<const uint4*>(cnd + (size_t)l * kCand + j);
        const unsigned e[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (j + u < n) offer((int)(e[u] & 0xffffu), (int)(e[u] >> 16), j + u);
      }
    } else {  // more candidates than the list holds: enumerate them again (rare)
      const Proj R = sbp_project(P, V, l, bForward, bBackward);
      int order = 0;
      sbp_candidates(P, V, Cur, R, l, s_start, s_items, [&](int i2, int dist) { offer(i2, dist, order++); });
    }
    const int bestDist = idx < 0 ? 256 : (int)(key >> 16);
    if (bestDist > kThHigh) return -1;
    if (P.mode == 1) {  // best / second-best ratio test when both sit on one pyramid level (:96-113)
      const int bestDist2 = idx2 < 0 ? 256 : (int)(key2 >> 16);
      const int bestLevel = s_oct[idx], bestLevel2 = idx2 < 0 ? -1 : (int)s_oct[idx2];
      if (bestLevel == bestLevel2 && (float)bestDist > P.nn_ratio * (float)bestDist2) return -1;
    }
    return idx;
  };
  __syncthreads();
  if (P.mode == 1) {
    // The ratio test makes a map point's choice depend on its SECOND best too, so a proposal can be withdrawn when a lower-index
    // blocker takes the second best away (or hands it back): the blocker table is not monotone and must not remember withdrawn
    // proposals.  Jacobi rounds: every map point decides from the blocker table of the previous round (read-only in this phase),
    // then the table is rebuilt from the current proposals.  sel[l] depends only on the proposals of map points below l, so after
    // round t the first t map points hold the sequential walk's choice for good: the fixed point is the reference's result, reached
    // in (longest dependency chain) rounds, and no thread reads the table while another writes it.
    for (int round = 0;; round++) {
      if (tid == 0) s_ctl[0] = 0;
      __syncthreads();
      bool changed = false;
      for (int l = tid; l < NL; l += kSbpThreads) {
        const int cur = round == 0 ? -3 : sel[l];
        const int now = decide(l);
        if (now != cur) {
          sel[l] = now;
          changed = true;
        }
      }
      if (changed) s_ctl[0] = 1;
      __syncthreads();
      const bool again = s_ctl[0] != 0;
      if (!again) break;  // (uniform: every thread read the same flag after the barrier)
      for (int i = tid; i < N; i += kSbpThreads) s_block[i] = 0x7fffffff;
      __syncthreads();
      for (int l = tid; l < NL; l += kSbpThreads) {
        const int now = sel[l];
        if (now >= 0 && s_hasobs[l]) atomicMin(&s_block[now], l);
      }
      __syncthreads();
    }
  } else {
    for (int round = 0;; round++) {
      if (tid == 0) s_ctl[0] = 0;
      __syncthreads();
      bool changed = false;
      for (int l = tid; l < NL; l += kSbpThreads) {
        const int cur = round == 0 ? -3 : sel[l];
        // a proposal stands until a lower-index blocker takes its key-point, "nothing" stands for good (options only shrink: a
        // key-point's blocker index only ever decreases, so the unique fixed point does not depend on the order of the updates)
        if (round > 0 && (cur < 0 || s_block[cur] >= l)) continue;
        const int now = decide(l);
        if (now != cur) {
          sel[l] = now;
          changed = true;
        }
        if (now >= 0 && s_hasobs[l] && atomicMin(&s_block[now], l) > l) changed = true;
      }
      if (changed) s_ctl[0] = 1;
      __syncthreads();
      const bool again = s_ctl[0] != 0;
      __syncthreads();
      if (!again) break;
    }
  }
  SBP_T(3)
  // takes: every map point with a proposal (an overwritten take counts like the reference counts it); the key-point ends up with the
  // LAST taker in index order = its blocker if it has one (nobody above a blocker can take it), else the highest-index taker
  int nm = 0;
  for (int l = tid; l < NL; l += kSbpThreads) {
    const int i2 = sel[l];
    if (i2 < 0) continue;
    nm++;
    atomicMax(&s_hold[i2], l);
    if (P.mode == 1) sel[l] = -1;  // (the rotation histogram below belongs to the frame-to-frame overload)
  }
  {
    for (int ofs = 32; ofs > 0; ofs >>= 1) nm += __shfl_down(nm, ofs, 64);
    if (tid == 0) s_ctl[0] = 0;
    __syncthreads();
    if (lane == 0 && nm) atomicAdd(&s_ctl[0], nm);
    __syncthreads();
  }
  for (int i = tid; i < N; i += kSbpThreads)
    if (s_hold[i] >= 0) s_state[i] = (short)s_hold[i];
  __syncthreads();
  if (P.mode == 0) {  // rotHist[bin].push_back(bestIdx2) of every assignment (:1935-1944), all map points at once
    for (int l = tid; l < NL; l += kSbpThreads) {
      const int best = sel[l];
      if (best < 0) continue;
      int bin = -1;
      if (P.check_orientation) {
        float rot = V.last_angle[l] - V.cur_kp[best].angle;
        if (rot < 0.0) rot += 360.0f;
        bin = (int)roundf(rot * (1.0f / kHisto));
        if (bin == kHisto) bin = 0;
        atomicAdd(&s_hist[bin], 1);
      }
      sel[l] = best | (bin << 16);
    }
  }
  __syncthreads();
  // ---- C. rotation consistency
  if (P.mode == 0 && P.check_orientation) {
    if (tid == 0) {  // ComputeThreeMaxima
      int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
      for (int i = 0; i < kHisto; i++) {
        const int s = s_hist[i];
        if (s > max1) {
          max3 = max2;
          max2 = max1;
          max1 = s;
          ind3 = ind2;
          ind2 = ind1;
          ind1 = i;
        } else if (s > max2) {
          max3 = max2;
          max2 = s;
          ind3 = ind2;
          ind2 = i;
        } else if (s > max3) {
          max3 = s;
          ind3 = i;
        }
      }
      if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
      } else if (max3 < 0.1f * (float)max1) {
        ind3 = -1;
      }
      int nm = s_ctl[0];
      for (int i = 0; i < kHisto; i++)
        if (i != ind1 && i != ind2 && i != ind3) nm -= s_hist[i];
      s_ctl[0] = nm;
      s_ctl[1] = ind1;
      s_ctl[2] = ind2;
      s_ctl[3] = ind3;
    }
    __syncthreads();
    const int ind1 = s_ctl[1], ind2 = s_ctl[2], ind3 = s_ctl[3];
    for (int l = tid; l < NL; l += kSbpThreads) {
      const int e = sel[l];
      if (e < 0) continue;
      const int bin = e >> 16, i2 = e & 0xffff;
      if (bin != ind1 && bin != ind2 && bin != ind3) s_state[i2] = -2;  // CurrentFrame.mvpMapPoints[i2] = NULL
    }
    __syncthreads();
  }
  SBP_T(4)
#ifdef GFS_SBP_TIMING
  __syncthreads();
  if (tid == 0 && f == 0) printf("SBPT grid=%lld A=%lld B=%lld C=%lld N=%d NL=%d\n", sbp_t[1] - sbp_t[0], sbp_t[2] - sbp_t[1], sbp_t[3] - sbp_t[2], sbp_t[4] - sbp_t[3], N, NL);
#endif
  for (int i = tid; i < N; i += kSbpThreads) cur_match[(size_t)f * SC + i] = s_state[i];
  if (tid == 0) nmatches[f] = s_ctl[0];
}

}  // namespace

struct gfs_sbp {
  int device, max_last, max_cur, max_batch;
  hipStream_t stream;
  std::mutex mu;
  // One pinned arena that mirrors one device block for the inputs, one for the results: a call is ONE copy in, the kernel, ONE copy
  // out (ten + two copies before -- ~10 us of host time and a copy-engine round trip each, twice the kernel's time for one frame).
  // The per-frame arrays are strided by the CALL's largest counts (rounded up to 64), not by the handle's capacity.
  gfs::DevBuf<uint8_t> d_in, d_res;
  gfs::PinBuf<uint8_t> h_in, h_res;
  gfs::DevBuf<int> d_cand_cnt, d_lsel;
  gfs::DevBuf<unsigned> d_cand;
  struct Layout {
    int SL, SC;
    size_t o_xw, o_desc, o_oct, o_ang, o_lobs, o_kp, o_ur, o_cdesc, o_cobs, in_bytes, r_nm, res_bytes;
  };
  Layout layout(size_t B, int SL, int SC) const {
    auto up = [](size_t v) { return gfs::align_up(v, 256); };
    const size_t L = (size_t)SL * B, Cn = (size_t)SC * B;
    Layout Y;
    Y.SL = SL;
    Y.SC = SC;
    Y.o_xw = up(B * sizeof(SbpPair));
    Y.o_desc = Y.o_xw + up(L * 12);
    Y.o_oct = Y.o_desc + up(L * 32);
    Y.o_ang = Y.o_oct + up(L * 4);
    Y.o_lobs = Y.o_ang + up(L * 4);
    Y.o_kp = Y.o_lobs + up(L);
    Y.o_ur = Y.o_kp + up(Cn * sizeof(gfs_keypoint));
    Y.o_cdesc = Y.o_ur + up(Cn * 4);
    Y.o_cobs = Y.o_cdesc + up(Cn * 32);
    Y.in_bytes = Y.o_cobs + up(Cn);
    Y.r_nm = up(Cn * 4);
    Y.res_bytes = Y.r_nm + up(B * 4);
    return Y;
  }
  // the staging views of a call (host side of the arena)
  struct Stage {
    SbpPair* pairs;
    float *last_xw, *last_angle, *cur_ur;
    uint8_t *last_desc, *last_has_obs, *cur_desc, *cur_has_obs;
    int* last_octave;
    gfs_keypoint* cur_kp;
    const int *cur_match, *nmatches;
  };
  Stage stage(const Layout& Y) const {
    uint8_t* b = h_in.p;
    return Stage{reinterpret_cast<SbpPair*>(b), reinterpret_cast<float*>(b + Y.o_xw), reinterpret_cast<float*>(b + Y.o_ang),
                 reinterpret_cast<float*>(b + Y.o_ur), b + Y.o_desc, b + Y.o_lobs, b + Y.o_cdesc, b + Y.o_cobs,
                 reinterpret_cast<int*>(b + Y.o_oct), reinterpret_cast<gfs_keypoint*>(b + Y.o_kp),
                 reinterpret_cast<const int*>(h_res.p), reinterpret_cast<const int*>(h_res.p + Y.r_nm)};
  }
};

// uploads the staged batch, runs k_sbp, downloads cur_match / nmatches into the pinned result block
static int sbp_run(gfs_sbp* h, int B, const gfs_sbp::Layout& Y) {
  hipStream_t s = h->stream;
  uint8_t* d = h->d_in.p;
  GFS_HIP(hipMemcpyAsync(d, h->h_in.p, Y.in_bytes, hipMemcpyHostToDevice, s));
  GFS_LAUNCH("k_sbp", k_sbp, dim3(B), dim3(kSbpThreads), 0, s, reinterpret_cast<const SbpPair*>(d), reinterpret_cast<const float*>(d + Y.o_xw),
             (const uint8_t*)(d + Y.o_desc), reinterpret_cast<const int*>(d + Y.o_oct), reinterpret_cast<const float*>(d + Y.o_ang),
             (const uint8_t*)(d + Y.o_lobs), reinterpret_cast<const gfs_keypoint*>(d + Y.o_kp), reinterpret_cast<const float*>(d + Y.o_ur),
             (const uint8_t*)(d + Y.o_cdesc), (const uint8_t*)(d + Y.o_cobs), Y.SL, Y.SC, h->d_cand.p, h->d_cand_cnt.p, h->d_lsel.p,
             reinterpret_cast<int*>(h->d_res.p), reinterpret_cast<int*>(h->d_res.p + Y.r_nm));
  GFS_HIP(hipMemcpyAsync(h->h_res.p, h->d_res.p, Y.res_bytes, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

extern "C" {

int gfs_sbp_create(int device, int max_last, int max_cur, int max_batch, gfs_sbp** out) {
  GFS_REQUIRE(out && max_last > 0 && max_cur > 0 && max_batch > 0, GFS_ERR_INVALID_ARG, "gfs_sbp_create: invalid argument");
  GFS_REQUIRE(max_cur <= kSbpMaxCur && max_last <= kSbpMaxLast, GFS_ERR_UNSUPPORTED,
              "gfs_sbp_create: at most %d current key-points and %d map points per frame", kSbpMaxCur, kSbpMaxLast);
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_sbp> h(new gfs_sbp);
  h->device = device;
  h->max_last = max_last;
  h->max_cur = max_cur;
  h->max_batch = max_batch;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const int SLmax = (int)gfs::align_up((size_t)max_last, 64), SCmax = (int)gfs::align_up((size_t)max_cur, 64);
  const size_t L = (size_t)SLmax * max_batch, B = max_batch;
  const gfs_sbp::Layout Y = h->layout(B, SLmax, SCmax);
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_in.alloc(Y.in_bytes));
  A(h->h_in.alloc(Y.in_bytes));
  A(h->d_res.alloc(Y.res_bytes));
  A(h->h_res.alloc(Y.res_bytes));
  A(h->d_cand.alloc(L * kCand));
  A(h->d_cand_cnt.alloc(L));
  A(h->d_lsel.alloc(L));
#undef A
  if (rc) {
    (void)hipStreamDestroy(h->stream);
    return rc;
  }
  *out = h.release();
  return GFS_OK;
}

void gfs_sbp_destroy(gfs_sbp* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_search_by_projection(gfs_sbp* h, const gfs_sbp_problem* problems, int B, int32_t* const* cur_match, int32_t* nmatches) {
  GFS_REQUIRE(h && problems && cur_match && nmatches && B > 0, GFS_ERR_INVALID_ARG, "gfs_search_by_projection: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_search_by_projection: batch %d exceeds capacity %d", B, h->max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int capL = h->max_last, capC = h->max_cur;
  int SL = 64, SC = 64;  // strides of the per-frame arrays in this call
  for (int f = 0; f < B; f++) {
    SL = std::max(SL, (int)gfs::align_up((size_t)std::max(problems[f].n_last, 0), 64));
    SC = std::max(SC, (int)gfs::align_up((size_t)std::max(problems[f].n_cur, 0), 64));
  }
  SL = std::min(SL, (int)gfs::align_up((size_t)capL, 64));
  SC = std::min(SC, (int)gfs::align_up((size_t)capC, 64));
  const gfs_sbp::Layout Y = h->layout((size_t)B, SL, SC);
  const gfs_sbp::Stage G = h->stage(Y);
  for (int f = 0; f < B; f++) {
    const gfs_sbp_problem& p = problems[f];
    GFS_REQUIRE(p.n_last >= 0 && p.n_last <= capL && p.n_cur >= 0 && p.n_cur <= capC, GFS_ERR_CAPACITY,
                "gfs_search_by_projection: pair %d has %d map points / %d key-points (capacity %d / %d)", f, p.n_last, p.n_cur, capL, capC);
    GFS_REQUIRE(p.n_levels > 0 && p.n_levels <= 16 && p.scale_factors, GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection: pair %d needs 1..16 scale factors", f);
    GFS_REQUIRE(p.n_cur == 0 || cur_match[f], GFS_ERR_INVALID_ARG, "gfs_search_by_projection: cur_match[%d] is NULL", f);
    GFS_REQUIRE(p.n_last == 0 || (p.last_xw && p.last_desc && p.last_octave && p.last_angle && p.last_mp_has_obs), GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection: pair %d has NULL last-frame arrays", f);
    GFS_REQUIRE(p.n_cur == 0 || (p.cur_kps_un && p.cur_u_right && p.cur_desc && p.cur_has_mp_obs), GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection: pair %d has NULL current-frame arrays", f);
    for (int l = 0; l < p.n_last; l++)
      GFS_REQUIRE(p.last_octave[l] >= 0 && p.last_octave[l] < p.n_levels, GFS_ERR_INVALID_ARG,
                  "gfs_search_by_projection: pair %d map point %d has octave %d outside [0, %d)", f, l, p.last_octave[l], p.n_levels);
    SbpPair& S = G.pairs[f];
    S.n_last = p.n_last;
    S.n_cur = p.n_cur;
    S.n_levels = p.n_levels;
    S.mono = p.mono;
    S.check_orientation = p.check_orientation;
    S.mode = 0;
    S.nn_ratio = 0.f;
    for (int k = 0; k < 4; k++) {
      S.Tcw_q[k] = p.Tcw_q[k];
      S.Tlw_q[k] = p.Tlw_q[k];
    }
    for (int k = 0; k < 3; k++) {
      S.Tcw_t[k] = p.Tcw_t[k];
      S.Tlw_t[k] = p.Tlw_t[k];
    }
    S.fx = p.fx;
    S.fy = p.fy;
    S.cx = p.cx;
    S.cy = p.cy;
    S.bf = p.bf;
    S.b = p.b;
    S.min_x = p.min_x;
    S.max_x = p.max_x;
    S.min_y = p.min_y;
    S.max_y = p.max_y;
    S.grid_w_inv = p.grid_w_inv;
    S.grid_h_inv = p.grid_h_inv;
    S.th = p.th;
    for (int k = 0; k < 16; k++) S.scale[k] = k < p.n_levels ? p.scale_factors[k] : 0.f;
    if (p.n_last > 0) {
      memcpy(G.last_xw + (size_t)f * SL * 3, p.last_xw, (size_t)p.n_last * 12);
      memcpy(G.last_desc + (size_t)f * SL * 32, p.last_desc, (size_t)p.n_last * 32);
      memcpy(G.last_octave + (size_t)f * SL, p.last_octave, (size_t)p.n_last * 4);
      memcpy(G.last_angle + (size_t)f * SL, p.last_angle, (size_t)p.n_last * 4);
      memcpy(G.last_has_obs + (size_t)f * SL, p.last_mp_has_obs, (size_t)p.n_last);
    }
    if (p.n_cur > 0) {
      memcpy(G.cur_kp + (size_t)f * SC, p.cur_kps_un, (size_t)p.n_cur * sizeof(gfs_keypoint));
      memcpy(G.cur_ur + (size_t)f * SC, p.cur_u_right, (size_t)p.n_cur * 4);
      memcpy(G.cur_desc + (size_t)f * SC * 32, p.cur_desc, (size_t)p.n_cur * 32);
      memcpy(G.cur_has_obs + (size_t)f * SC, p.cur_has_mp_obs, (size_t)p.n_cur);
    }
  }
  const int rc = sbp_run(h, B, Y);
  if (rc != GFS_OK) return rc;
  for (int f = 0; f < B; f++) {
    if (problems[f].n_cur > 0) memcpy(cur_match[f], G.cur_match + (size_t)f * SC, (size_t)problems[f].n_cur * 4);
    nmatches[f] = G.nmatches[f];
  }
  return GFS_OK;
}

int gfs_search_by_projection_map(gfs_sbp* h, const gfs_sbp_map_problem* problems, int B, int32_t* const* cur_match, int32_t* nmatches) {
  GFS_REQUIRE(h && problems && cur_match && nmatches && B > 0, GFS_ERR_INVALID_ARG, "gfs_search_by_projection_map: invalid argument");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_search_by_projection_map: batch %d exceeds capacity %d", B, h->max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int capL = h->max_last, capC = h->max_cur;
  int SL = 64, SC = 64;  // strides of the per-frame arrays in this call
  for (int f = 0; f < B; f++) {
    SL = std::max(SL, (int)gfs::align_up((size_t)std::max(problems[f].n_mp, 0), 64));
    SC = std::max(SC, (int)gfs::align_up((size_t)std::max(problems[f].n_cur, 0), 64));
  }
  SL = std::min(SL, (int)gfs::align_up((size_t)capL, 64));
  SC = std::min(SC, (int)gfs::align_up((size_t)capC, 64));
  const gfs_sbp::Layout Y = h->layout((size_t)B, SL, SC);
  const gfs_sbp::Stage G = h->stage(Y);
  for (int f = 0; f < B; f++) {
    const gfs_sbp_map_problem& p = problems[f];
    GFS_REQUIRE(p.n_mp >= 0 && p.n_mp <= capL && p.n_cur >= 0 && p.n_cur <= capC, GFS_ERR_CAPACITY,
                "gfs_search_by_projection_map: frame %d has %d map points / %d key-points (capacity %d / %d)", f, p.n_mp, p.n_cur, capL, capC);
    GFS_REQUIRE(p.n_levels > 0 && p.n_levels <= 16 && p.scale_factors, GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection_map: frame %d needs 1..16 scale factors", f);
    GFS_REQUIRE(p.n_cur == 0 || cur_match[f], GFS_ERR_INVALID_ARG, "gfs_search_by_projection_map: cur_match[%d] is NULL", f);
    GFS_REQUIRE(p.n_mp == 0 || (p.mp_proj && p.mp_desc && p.mp_level && p.mp_view_cos && p.mp_has_obs), GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection_map: frame %d has NULL map-point arrays", f);
    GFS_REQUIRE(p.n_cur == 0 || (p.cur_kps_un && p.cur_u_right && p.cur_desc && p.cur_has_mp_obs), GFS_ERR_INVALID_ARG,
                "gfs_search_by_projection_map: frame %d has NULL key-point arrays", f);
    for (int l = 0; l < p.n_mp; l++)
      GFS_REQUIRE(p.mp_level[l] >= 0 && p.mp_level[l] < p.n_levels, GFS_ERR_INVALID_ARG,
                  "gfs_search_by_projection_map: frame %d map point %d has level %d outside [0, %d)", f, l, p.mp_level[l], p.n_levels);
    SbpPair& S = G.pairs[f];
    memset(&S, 0, sizeof(S));
    S.n_last = p.n_mp;
    S.n_cur = p.n_cur;
    S.n_levels = p.n_levels;
    S.mode = 1;
    S.nn_ratio = p.nn_ratio;
    S.min_x = p.min_x;
    S.min_y = p.min_y;
    S.grid_w_inv = p.grid_w_inv;
    S.grid_h_inv = p.grid_h_inv;
    S.th = p.th;
    for (int k = 0; k < 16; k++) S.scale[k] = k < p.n_levels ? p.scale_factors[k] : 0.f;
    if (p.n_mp > 0) {
      memcpy(G.last_xw + (size_t)f * SL * 3, p.mp_proj, (size_t)p.n_mp * 12);
      memcpy(G.last_desc + (size_t)f * SL * 32, p.mp_desc, (size_t)p.n_mp * 32);
      memcpy(G.last_octave + (size_t)f * SL, p.mp_level, (size_t)p.n_mp * 4);
      memcpy(G.last_angle + (size_t)f * SL, p.mp_view_cos, (size_t)p.n_mp * 4);
      memcpy(G.last_has_obs + (size_t)f * SL, p.mp_has_obs, (size_t)p.n_mp);
    }
    if (p.n_cur > 0) {
      memcpy(G.cur_kp + (size_t)f * SC, p.cur_kps_un, (size_t)p.n_cur * sizeof(gfs_keypoint));
      memcpy(G.cur_ur + (size_t)f * SC, p.cur_u_right, (size_t)p.n_cur * 4);
      memcpy(G.cur_desc + (size_t)f * SC * 32, p.cur_desc, (size_t)p.n_cur * 32);
      memcpy(G.cur_has_obs + (size_t)f * SC, p.cur_has_mp_obs, (size_t)p.n_cur);
    }
  }
  const int rc = sbp_run(h, B, Y);
  if (rc != GFS_OK) return rc;
  for (int f = 0; f < B; f++) {
    if (problems[f].n_cur > 0) memcpy(cur_match[f], G.cur_match + (size_t)f * SC, (size_t)problems[f].n_cur * 4);
    nmatches[f] = G.nmatches[f];
  }
  return GFS_OK;
}

}  // extern "C"
