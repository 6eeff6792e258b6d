// ORB extractor for gfx950 (MI355X): INTER_AREA pyramid -> per-cell FAST-9/16 + score + NMS -> ordered
// candidate lists -> quadtree distribution -> intensity-centroid angle -> fixed-point 7x7 blur ->
// steered BRIEF-256.  Replaces ORB_SLAM3::ORBextractor::operator() (reference src/ORBextractor.cc:1145-1225).
//
// Bit-exactness rules: every float operation the reference performs is issued as an
// explicitly rounded single operation (__fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn: never contracted into FMA),
// integer stages are order-independent, and cosf/sinf are the restated glibc algorithm
// (oracle/check_sincosf.c proves it bit-identical to libm on [0, 6.5]).
//
// HBM layout per batch of B frames (all planes row-major u8, pitch = cols rounded up to 64):
//   level 0            : caller's dense [B][rows][cols] buffer (or the handle's staging copy)
//   pyr  [B][pyr_bytes]: levels 1..L-1 back to back            blur [B][blur_bytes]: blurred levels 0..L-1
//   slab [B][slab_entries] u32 + cell_cnt [B][n_cells]         : per-cell FAST survivors (x|y<<12|score<<24)
//   cand [B][cand_cap] u32 + cand_off [B][L+1]                 : the same, packed in (level, cell, raster) order
//   kpin [B][kp_cap] -> kps [B][kp_cap] gfs_keypoint + desc [B][kp_cap][32]
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

#include "gfs_common.hpp"
#include "orb_host.hpp"
#include "std_sort_replica.hpp"
#include "wave_std_sort.hpp"

using gfs::BlurTileDev;
using gfs::CellDev;
using gfs::LevelDev;

namespace {

struct KpIn {  // one keypoint after the quadtree, level coordinates
  float x, y;
  int level;
  float response;
  int slot;  // output slot (monoIndex / stereoIndex ordering of src/ORBextractor.cc:1209-1219)
};

struct Lvl0 {  // where level 0 lives for this call
  const uint8_t* base;
  size_t frame_stride;
  int pitch;
};

__device__ __forceinline__ const uint8_t* level_ptr(const LevelDev& L, int level, int b, Lvl0 l0, const uint8_t* pyr,
                                                    size_t pyr_frame, int* pitch) {
  if (level == 0) {
    *pitch = l0.pitch;
    return l0.base + (size_t)b * l0.frame_stride;
  }
  *pitch = L.pitch;
  return pyr + (size_t)b * pyr_frame + L.plane_off;
}

// ------------------------------------------------------------------------------------------------
// k_pyr_area: level l from level l-1, cv::resize(INTER_AREA) arithmetic (OpenCV resize.cpp
// ResizeArea_Invoker<uchar,float>): float32, products and sums in table order, no FMA,
// saturate_cast<uchar>(cvRound(sum)).  One thread per destination pixel; <= 4x4 source taps.
// Reference call: src/ORBextractor.cc:1240-1241.
// ------------------------------------------------------------------------------------------------
// One launch for the whole chain: workgroup (k, b) produces, level after level, a horizontal strip of every level of
// frame b (OrbGeometry::strip_rows: its share of the level plus the halo rows its own higher levels read), so a level
// only ever reads rows the same workgroup produced.  Neighbouring strips recompute a few identical halo rows instead of
// synchronising.
//
// k_pyr_area_lds: the strips live in LDS (two ping-pong buffers, rows of kPyrLdsPitch(cols) bytes: level 0 strip staged with
// word loads, every level is computed LDS -> LDS and streamed out to HBM once).  A WAVE walks down a chunk of <= 128 columns,
// lane <-> two neighbouring columns (their taps -- start, count, weights from the host-built tables -- in registers), the
// rows one after the other:
//  * the horizontal sum of a source row is the same number for every destination row that reads it (OpenCV recomputes `buf`
//    per (dy, sy) entry of its y table from the same bytes), and at scale factors < 2 consecutive destination rows share their
//    boundary source row: kept from one row to the next, ~1.2 horizontal sums per pixel instead of the 3 a thread per pixel
//    spends;
//  * the two columns of a lane go through v_pk_mul_f32 / v_pk_add_f32 (IEEE per component, no FMA: the same bits), and out
//    as one 16-bit store;
//  * a row's taps are wave-uniform: read from an LDS copy of the y tables (all levels of the strip, staged once beside the
//    level-0 rows) into scalar registers, the tap count is a scalar branch; word-aligned LDS rows make a lane's byte shift a
//    constant of the run, so a source row costs one address add per column;
//  * the (chunk, row) units of a level are cut into sixteen contiguous runs, chunk-major, one per wave.
// What it replaced (round 4, clock64 per level under -DGFS_PYR_TIMING): thread <-> pixel with 3 x 4 taps each, ~57 VALU
// instructions a pixel, VALU-bound at one workgroup a CU, plus ~7 k cycles a level of table loads and set-up in front of the
// first pixel: 1.17 ms per 512 VGA frames.
//
// k_pyr_area_hbm: same strips through HBM, thread <-> pixel (a level re-reads the rows its own workgroup wrote:
// workgroup-scope fence + barrier; an agent-scope fence would write back the XCD's L2 at every level, measured 4x slower) for
// images whose strips do not fit the LDS.
constexpr int kPyrThreads = 1024, kPyrWaves = kPyrThreads / 64;
constexpr int kPyrTabRows = gfs::OrbGeometry::kPyrTabRows, kPyrProgRows = gfs::OrbGeometry::kPyrProgRows;
typedef float pyr_f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kPyrThreads) void k_pyr_area_lds(const LevelDev* __restrict__ levels, int nlevels, Lvl0 l0,
                                                              uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                              const int* __restrict__ strip_rows, unsigned lds_a,
                                                              unsigned lds_x, int xt_total,
                                                              const int* __restrict__ xt_start, const int* __restrict__ xt_n,
                                                              const float* __restrict__ xt_alpha, const int* __restrict__ yt_start,
                                                              const int* __restrict__ yt_n, const float* __restrict__ yt_alpha) {
  extern __shared__ __align__(16) uint8_t smem[];
  // The y tables of the strip as a program over SOURCE rows (all levels, one entry per source row a level reads): the weight
  // it carries into the destination row that is open there, the weight into the row that OPENS there when it is the boundary
  // row of two (else 0), whether the open row ends there, whether a row opens there.  Built once from the tables in memory;
  // s_first: the source row a destination row starts on (where a wave's run begins).
  __shared__ float4 s_prog[kPyrProgRows + 1];
  __shared__ int s_first[kPyrTabRows];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int* rng = strip_rows + 2 * k * nlevels;
#ifdef GFS_PYR_TIMING
  __shared__ long long pyr_t[12];
  { const long long _n = clock64(); if (tid == 0) pyr_t[0] = _n; }
#endif
  // Staging: every load is asked for before the first one is waited for (four in flight per thread and table; one round trip
  // per loop iteration made this part 17 k cycles of a workgroup's 87 k).
  for (int i = tid; i < kPyrProgRows + 1; i += kPyrThreads) s_prog[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (1) the y-table row of this thread (one thread a destination row of the strip, all levels), kept in registers for now
  bool my_row = false;
  int my_first = 0, my_ny = 0, my_prev_last = -1, my_slot = 0, my_prog = 0;
  float4 my_ay = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    int off = 0, pb = 0;  // first destination row / first program entry of the level
    for (int level = 1; level < nlevels; level++) {
      const int r0 = rng[2 * level], n = rng[2 * level + 1] - r0, yo = levels[level].ytab_off;
      const int srow0 = rng[2 * (level - 1)];
      const int i = tid - off;
      if (i >= 0 && i < n) {
        const int d = r0 + i;
        my_row = true;
        my_first = yt_start[yo + d];
        my_ny = yt_n[yo + d];
        my_ay = *reinterpret_cast<const float4*>(yt_alpha + 4 * (size_t)(yo + d));
        if (d > 0) my_prev_last = yt_start[yo + d - 1] + yt_n[yo + d - 1] - 1;
        my_slot = off + i;
        my_prog = pb - srow0;
      }
      off += n;
      pb += rng[2 * (level - 1) + 1] - srow0;
    }
  }
  // (2) the column taps of every level >= 1 (lds_x = their offset behind the two strip buffers; 0 = they stay in memory: a
  // column somewhere has a fourth tap, or they do not fit): a run's taps are then two LDS reads instead of a round trip to the
  // L2 in front of its first row (~1 500 cycles, twice for a wave whose run crosses a chunk boundary while the others wait)
  float4* s_xt = reinterpret_cast<float4*>(smem + lds_x);
  if (lds_x)
    for (int i0 = tid; i0 < xt_total; i0 += 4 * kPyrThreads) {
      float4 a[4];
      int st[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = i0 + q * kPyrThreads;
        if (i < xt_total) {
          a[q] = *reinterpret_cast<const float4*>(xt_alpha + 4 * (size_t)i);
          st[q] = xt_start[i];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int i = i0 + q * kPyrThreads;
        if (i < xt_total) s_xt[i] = make_float4(a[q].x, a[q].y, a[q].z, __int_as_float(st[q]));
      }
    }
  {  // (3) the level-0 rows level 1 reads
    const LevelDev S0 = levels[0];
    int gp;
    const uint8_t* src = level_ptr(S0, 0, b, l0, pyr, pyr_frame, &gp);
    const int s0 = rng[0], s1 = rng[1], cols = S0.cols, sp = gfs::OrbGeometry::kPyrLdsPitch(cols);
    if (((cols | gp) & 15) == 0 && ((uintptr_t)src & 15) == 0) {
      const int qpr = cols >> 4, total = (s1 - s0) * qpr;
      for (int i0 = tid; i0 < total; i0 += 4 * kPyrThreads) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = i0 + q * kPyrThreads;
          if (i < total) {
            const int y = i / qpr, x = i - y * qpr;
            v[q] = *reinterpret_cast<const uint4*>(src + (size_t)(s0 + y) * gp + 16 * x);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = i0 + q * kPyrThreads;
          if (i < total) reinterpret_cast<uint4*>(smem)[i] = v[q];
        }
      }
    } else if (((cols | gp) & 3) == 0 && ((uintptr_t)src & 3) == 0) {
      const int wpr = cols >> 2, total = (s1 - s0) * wpr;
      for (int i0 = tid; i0 < total; i0 += 4 * kPyrThreads) {
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = i0 + q * kPyrThreads;
          if (i < total) {
            const int y = i / wpr, x = i - y * wpr;
            v[q] = *reinterpret_cast<const uint32_t*>(src + (size_t)(s0 + y) * gp + 4 * x);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int i = i0 + q * kPyrThreads;
          if (i < total) reinterpret_cast<uint32_t*>(smem)[i] = v[q];
        }
      }
    } else {
      const int total = (s1 - s0) * cols;
      for (int i = tid; i < total; i += kPyrThreads) {
        const int y = i / cols, x = i - y * cols;
        smem[y * sp + x] = src[(size_t)(s0 + y) * gp + x];
      }
    }
  }
  __syncthreads();  // (the program is zeroed)
  if (my_row) {     // (4) the thread's row into the program
    const float ays[4] = {my_ay.x, my_ay.y, my_ay.z, my_ay.w};
    s_first[my_slot] = my_first;
    float* pr = reinterpret_cast<float*>(s_prog + my_prog + my_first);
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (j < my_ny) {
        if (j == 0 && my_first == my_prev_last) {  // the boundary row of d - 1 and d
          pr[1] = ays[0];
          pr[3] = __int_as_float(1);
        } else {
          pr[4 * j] = ays[j];
        }
        if (j == my_ny - 1) pr[4 * j + 2] = __int_as_float(1);
      }
  }
  __syncthreads();
#ifdef GFS_PYR_TIMING
  { const long long _n = clock64(); if (tid == 0) pyr_t[8] = _n; }
#endif
  int tab0 = 0, pb = 0;  // first destination row / first program entry of the level
  for (int level = 1; level < nlevels; level++) {
    const LevelDev L = levels[level];
    const int scols = levels[level - 1].cols;
    const int r0 = rng[2 * level], nrow = rng[2 * level + 1] - r0, srow0 = rng[2 * (level - 1)], nsrc = rng[2 * (level - 1) + 1] - srow0;
    // source rows [srow0, ...) of level - 1 sit in the buffer of that level's parity, the rows produced here go to the other one
    const int sp = gfs::OrbGeometry::kPyrLdsPitch(scols), kp = gfs::OrbGeometry::kPyrLdsPitch(L.cols);
    const unsigned src_off = (level - 1) & 1 ? lds_a : 0u, keep_off = level & 1 ? lds_a : 0u;
    const bool has_keep = level + 1 < nlevels;
    uint8_t* dst = pyr + (size_t)b * pyr_frame + L.plane_off;
    const bool dst_even = (((uintptr_t)dst | (unsigned)L.pitch) & 1) == 0;
    const int nchunk = (L.cols + 127) >> 7, cw = (((L.cols + nchunk - 1) / nchunk) + 1) & ~1;  // columns of a chunk: even, <= 128
    const int units = nchunk * nrow;
    const int u1 = (int)((long long)(wave + 1) * units / kPyrWaves);
    int u = (int)((long long)wave * units / kPyrWaves);
    while (u < u1) {
      const int chunk = u / nrow, row_a = u - chunk * nrow, row_b = min(nrow, row_a + (u1 - u));
      const int dx0 = chunk * cw + 2 * lane;
      const bool on0 = 2 * lane < cw && dx0 < L.cols, on1 = on0 && dx0 + 1 < L.cols;
      const int xa = L.xtab_off + min(dx0, L.cols - 1), xb = L.xtab_off + min(dx0 + 1, L.cols - 1);
      int sxa, sxb;
      float4 axa, axb;
      bool x4 = false;  // a fourth tap only exists for scale factors above 2
      if (lds_x) {
        axa = s_xt[xa];
        axb = s_xt[xb];
        sxa = __float_as_int(axa.w);
        sxb = __float_as_int(axb.w);
      } else {
        sxa = xt_start[xa];
        sxb = xt_start[xb];
        x4 = __ballot(xt_n[xa] > 3 || xt_n[xb] > 3) != 0ull;
        axa = *reinterpret_cast<const float4*>(xt_alpha + 4 * (size_t)xa);
        axb = *reinterpret_cast<const float4*>(xt_alpha + 4 * (size_t)xb);
      }
      const pyr_f2 a0 = {axa.x, axb.x}, a1 = {axa.y, axb.y}, a2 = {axa.z, axb.z}, a3 = {axa.w, axb.w};
      // the aligned word a column's first byte lies in (+ the next one) and the shift that brings its four bytes down
      const uint32_t* wa = reinterpret_cast<const uint32_t*>(smem + src_off + (sxa & ~3));
      const uint32_t* wb = reinterpret_cast<const uint32_t*>(smem + src_off + (sxb & ~3));
      const int sha = (sxa & 3) * 8, shb = (sxb & 3) * 8, spw = sp >> 2;
      uint8_t* drow = dst + (size_t)(r0 + row_a) * L.pitch + dx0;
      uint8_t* krow = smem + keep_off + (size_t)row_a * kp + dx0;
      auto walk = [&](auto X4) {
        constexpr bool kX4 = decltype(X4)::value;
        // The run is a stream of SOURCE rows: every one is read once (its words and its program entry fetched an iteration
        // ahead), its horizontal sums go with the entry's weight into the open destination row (0 + p = p, x + 0 = x: all terms
        // are >= +0, so a sum that starts at +0 and source rows with weight 0 leave OpenCV's bits alone); when the row ends
        // there it is rounded and stored and the next one starts with the boundary weight (or +0).
        auto words = [&](int si, uint32_t(&w)[4]) {  // si: source row relative to the strip
          const int o = si * spw;
          w[0] = wa[o];
          w[1] = wa[o + 1];
          w[2] = wb[o];
          w[3] = wb[o + 1];
        };
        // buf[dx] of a source row for the lane's two columns: b0 a0 + b1 a1 + ... in table order (weights beyond a column's
        // taps are 0 in the tables and the bytes under them finite)
        auto hsum = [&](const uint32_t(&w)[4]) {
          const uint32_t va = __builtin_amdgcn_alignbit(w[1], w[0], sha), vb = __builtin_amdgcn_alignbit(w[3], w[2], shb);
          const pyr_f2 c0 = {(float)(va & 0xffu), (float)(vb & 0xffu)}, c1 = {(float)((va >> 8) & 0xffu), (float)((vb >> 8) & 0xffu)},
                       c2 = {(float)((va >> 16) & 0xffu), (float)((vb >> 16) & 0xffu)};
          pyr_f2 buf = c0 * a0;
          buf = buf + c1 * a1;
          buf = buf + c2 * a2;
          if constexpr (kX4) {
            const pyr_f2 c3 = {(float)(va >> 24), (float)(vb >> 24)};
            buf = buf + c3 * a3;
          }
          return buf;
        };
        auto emit = [&](pyr_f2 sum) {
          // saturate_cast<uchar>(cvRound(sum)): v_cvt_pk_u8_f32 rounds half to even and clamps to 0..255 (checked against
          // __float2int_rn + clamp over 16.7 M values with every tie and its neighbours: tools/probes/cvt_pk_u8_probe.hip)
          const unsigned v = __builtin_amdgcn_cvt_pk_u8_f32(sum.y, 1u, __builtin_amdgcn_cvt_pk_u8_f32(sum.x, 0u, 0u));
          if (on0) {
            if (on1 && dst_even) {
              *reinterpret_cast<uint16_t*>(drow) = (uint16_t)v;
            } else {
              drow[0] = (uint8_t)v;
              if (on1) drow[1] = (uint8_t)(v >> 8);
            }
            if (has_keep) *reinterpret_cast<uint16_t*>(krow) = (uint16_t)v;  // (kp is a multiple of 4, dx0 even: aligned; an odd
                                                                             //  level's last column writes one pad byte)
          }
          drow += L.pitch;
          krow += kp;
        };
        int si = __builtin_amdgcn_readfirstlane(s_first[tab0 + row_a]) - srow0, left = row_b - row_a;
        uint32_t w[4];
        words(si, w);
        float4 P = s_prog[pb + si];
        pyr_f2 sum;
        {  // the first source row: the run's first row opens there (behind the row before it) or is the open one
          const pyr_f2 h = hsum(w);
          words(min(si + 1, nsrc - 1), w);
          const float4 Pn = s_prog[pb + si + 1];
          const bool opens = __builtin_amdgcn_readfirstlane(__float_as_int(P.w)) != 0;
          sum = h * (opens ? P.y : P.x);
          if (!opens && __builtin_amdgcn_readfirstlane(__float_as_int(P.z)) != 0) {  // (a row of one tap)
            emit(sum);
            sum = h * P.y;
            left--;
          }
          P = Pn;
          si++;
        }
        while (left > 0) {
          const pyr_f2 h = hsum(w);
          words(min(si + 1, nsrc - 1), w);
          const float4 Pn = s_prog[pb + si + 1];
          sum = sum + h * P.x;
          if (__builtin_amdgcn_readfirstlane(__float_as_int(P.z)) != 0) {
            emit(sum);
            sum = h * P.y;
            left--;
          }
          P = Pn;
          si++;
        }
      };
      if (x4) walk(std::true_type{});
      else walk(std::false_type{});
      u += row_b - row_a;
    }
    pb += nsrc;
    tab0 += nrow;
    __syncthreads();
#ifdef GFS_PYR_TIMING
    { const long long _n = clock64(); if (tid == 0) pyr_t[level] = _n; }
#endif
  }
#ifdef GFS_PYR_TIMING
  if (tid == 0 && b == 7 && (k == 0 || k == 3))
    printf("PYRT k=%d stage=%lld l1=%lld l2=%lld l3=%lld l4=%lld l5=%lld l6=%lld l7=%lld total=%lld\n", k, pyr_t[8] - pyr_t[0],
           pyr_t[1] - pyr_t[8], pyr_t[2] - pyr_t[1], pyr_t[3] - pyr_t[2], pyr_t[4] - pyr_t[3], pyr_t[5] - pyr_t[4], pyr_t[6] - pyr_t[5],
           pyr_t[7] - pyr_t[6], pyr_t[7] - pyr_t[0]);
#endif
}

__global__ __launch_bounds__(kPyrThreads) void k_pyr_area_hbm(const LevelDev* __restrict__ levels, int nlevels, Lvl0 l0,
                                                              uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                              const int* __restrict__ strip_rows,
                                                              const int* __restrict__ xt_start, const int* __restrict__ xt_n,
                                                              const float* __restrict__ xt_alpha, const int* __restrict__ yt_start,
                                                              const int* __restrict__ yt_n, const float* __restrict__ yt_alpha) {
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int* rng = strip_rows + 2 * k * nlevels;
  for (int level = 1; level < nlevels; level++) {
    const LevelDev L = levels[level];
    const LevelDev S = levels[level - 1];
    const int r0 = rng[2 * level], r1 = rng[2 * level + 1];
    int sp;
    const uint8_t* src = level_ptr(S, level - 1, b, l0, pyr, pyr_frame, &sp);
    uint8_t* dst = pyr + (size_t)b * pyr_frame + L.plane_off;
    // thread <-> (column, row phase): the column's taps (start, count, four weights) stay in registers down the strip
    // (ncolt columns x rows_par row phases <= 1024 threads, a thread walks q = ceil(cols / ncolt) columns; q is chosen for the
    // most pixels per sweep of the workgroup)
    int ncolt = min(L.cols, kPyrThreads), rows_par = kPyrThreads / ncolt;
    {
      int best = rows_par * L.cols;  // q = 1
      for (int q = 2; q <= 8; q++) {
        const int c = (L.cols + q - 1) / q, r = kPyrThreads / c;
        if (c >= 64 && r * L.cols > best * q) {  // r * cols / q > best / 1
          best = r * L.cols / q;
          ncolt = c;
          rows_par = r;
        }
      }
    }
    const int cxi = tid % ncolt, ry = tid / ncolt;
    if (ry < rows_par) {
      for (int dx = cxi; dx < L.cols; dx += ncolt) {
        const int xi = L.xtab_off + dx;
        const int sx0 = xt_start[xi], nx = xt_n[xi];
        const float4 ax = *reinterpret_cast<const float4*>(xt_alpha + 4 * (size_t)xi);
        for (int dy = r0 + ry; dy < r1; dy += rows_par) {
          const int yi = L.ytab_off + dy;
          const int sy0 = yt_start[yi], ny = yt_n[yi];
          const float4 ay = *reinterpret_cast<const float4*>(yt_alpha + 4 * (size_t)yi);
          const float ays[4] = {ay.x, ay.y, ay.z, ay.w};
          float sum = 0.f;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j < ny) {
              const uint8_t* row = src + (size_t)(sy0 + j) * sp + sx0;
              float buf = __fadd_rn(0.f, __fmul_rn((float)row[0], ax.x));
              if (nx > 1) buf = __fadd_rn(buf, __fmul_rn((float)row[1], ax.y));
              if (nx > 2) buf = __fadd_rn(buf, __fmul_rn((float)row[2], ax.z));
              if (nx > 3) buf = __fadd_rn(buf, __fmul_rn((float)row[3], ax.w));
              const float t = __fmul_rn(ays[j], buf);
              sum = (j == 0) ? t : __fadd_rn(sum, t);
            }
          }
          const int r = __float2int_rn(sum);  // cvRound: round-half-even
          dst[(size_t)dy * L.pitch + dx] = (uint8_t)min(max(r, 0), 255);
        }
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// k_fast_cells: one WAVE per FAST cell (reference src/ORBextractor.cc:788-842 calling cv::FAST on the
// cell sub-image).  The (w x h) cell tile is staged in LDS; for every interior pixel the arc score
//      S = max over the 16 arcs of 9 contiguous ring pixels of min(v - ring)  (and of min(ring - v))
// is evaluated once; a pixel is a FAST corner at threshold t iff S > t and OpenCV's cornerScore is S - 1.
// 3x3 strict-max NMS runs inside the cell with zeros outside (OpenCV FAST_t<16> row buffers), first with
// iniThFAST and — only if the cell produced nothing — with minThFAST.  Survivors are emitted in raster order.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned has9(unsigned m) {  // any run of >= 9 set bits in the cyclic 16-bit mask?
  const unsigned m32 = m | (m << 16);
  unsigned r = m32 & (m32 >> 1);
  r &= r >> 2;
  r &= r >> 4;
  r &= m32 >> 8;
  return r & 0xffffu;
}

// One polarity only.  The antipodal-pair pre-test says which polarity a pixel can be a corner of (a corner with 9 darker ring pixels
// passes the DARK pair test), and a pixel cannot be a corner of both: two 9-arcs of a 16-ring overlap.  For the same reason the
// score of a dark corner is its dark arc score alone: every 9-arc holds a pixel of the dark arc, where ring - v < -t < 0, so the
// bright term of max(A, -Bm) stays below -t.  pol = 0: darker (ring < v - t), 1: brighter (ring > v + t).
#define GFS_RING16_(F)                                                                                                     \
  F(rp3[0]) F(rp3[1]) F(rp2[2]) F(rp1[3]) F(c[3]) F(rm1[3]) F(rm2[2]) F(rm3[1]) F(rm3[0]) F(rm3[-1]) F(rm2[-2]) F(rm1[-3]) \
  F(c[-3]) F(rp1[-3]) F(rp2[-2]) F(rp3[-1])
__device__ __forceinline__ bool arc_is_corner_pol(const uint8_t* __restrict__ c, int p, int th, int pol) {
  const int v = c[0];
  const int sgn = pol ? -1 : 1, off = pol ? v + th : th - v;  // sign bit of sgn * ring + off: ring < v - th resp. ring > v + th
  const uint8_t *rm3 = c - 3 * p, *rm2 = c - 2 * p, *rm1 = c - p, *rp1 = c + p, *rp2 = c + 2 * p, *rp3 = c + 3 * p;
  unsigned m = 0;
#define GFS_RING_(px) m = __builtin_amdgcn_alignbit(m, (unsigned)((int)(px) * sgn + off), 31);
  GFS_RING16_(GFS_RING_)
#undef GFS_RING_
  return has9(m & 0xffffu) != 0;
}
// S = max over the 16 arcs of 9 of min(d), d = v - ring (dark) resp. ring - v (bright)
__device__ __forceinline__ int arc_score_pol(const uint8_t* __restrict__ c, int p, int pol) {
  const int v = c[0];
  const int sgn = pol ? 1 : -1, off = pol ? -v : v;
  const uint8_t *rm3 = c - 3 * p, *rm2 = c - 2 * p, *rm1 = c - p, *rp1 = c + p, *rp2 = c + 2 * p, *rp3 = c + 3 * p;
  int d[16];
  {
    int k = 0;
#define GFS_RING_(px) d[k++] = (int)(px) * sgn + off;
    GFS_RING16_(GFS_RING_)
#undef GFS_RING_
  }
  int lo[16], lo4[16];
#pragma unroll
  for (int k = 0; k < 16; k++) lo[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
  for (int k = 0; k < 16; k++) lo4[k] = min(lo[k], lo[(k + 2) & 15]);
  int A = -256;
#pragma unroll
  for (int k = 0; k < 16; k++) A = max(A, min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]));
  return A;
}
#undef GFS_RING16_

// k_fast_cells, round 5: ONE WAVE per cell, kFcWaves independent cells per workgroup, no workgroup barrier and no LDS atomic
// anywhere.  Round 4's counters (profiles/r04h_pmc_sq.json: 902 M VALU wave-instructions per launch of 512 VGA frames = 3 055 per
// cell, x 4 cycles / 1 024 SIMDs = 1.47 of the 1.585 ms) say the kernel is bound by VALU issue, not by latency: four waves per cell
// each paid the loop and address overhead of every phase, a byte at a time.  Now
//   phase 0  the antipodal-pair pre-test (9 contiguous ring pixels contain one pixel of every antipodal pair, so a corner has
//            top|bottom AND left|right darker than v - t, or brighter than v + t) runs on FOUR pixels a lane from aligned LDS
//            words: the five words (rows y-3, y, y+3; left / right windows by v_alignbyte) are split into even / odd bytes and
//            compared as packed 16-bit pairs (v_pk_min/max_u16, v_pk_add/sub_i16): ~13 instructions a pixel slot, was ~30 + 5 LDS
//            byte reads.  A lane keeps the pass bits of its 4-pixel groups (one column group, every rstep-th row) in one register;
//   compact  one wave scan (DPP) of the per-lane counts, then every lane writes its own pixels' offsets;
//   phase 1  the exact 9-arc mask test on dense lanes over that list (two entries a lane in flight), corners compacted in place
//            by wave ballot;
//   phase 2  exact arc scores of the corners, phase 3 the 3x3 NMS over the corner list, phase 4 rank = raster order, emit.
// The workgroup -> (frame, cell) map is XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs, so id % 8 picks the
// XCD and all cells of a frame (or of a contiguous part of a frame when there are fewer than 8 frames) run on ONE XCD, neighbouring
// cells back to back: the 6-pixel overlap of neighbouring tiles is served by that XCD's L2 instead of being fetched from HBM by
// several XCDs (round 4: FETCH_SIZE 1.99 x the pixels).
#ifndef GFS_FC_WAVES
#define GFS_FC_WAVES 4
#endif
constexpr int kFcWaves = GFS_FC_WAVES;
#ifdef GFS_FAST_TIMING
#define FC_T_INIT long long fc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fc_last = clock64();
#define FC_T(k) { const long long _n = clock64(); fc_acc[k] += _n - fc_last; fc_last = _n; }
#define FC_T_END(nq_, nc_, nk_) if (lane == 0 && (blockIdx.x % 257) == 0 && wave == 0) printf("FASTT lvl=%d stage=%lld p0=%lld compact=%lld arc=%lld score=%lld nms=%lld emit=%lld nq=%d nc=%d nk=%d\n", (int)C.level, fc_acc[0], fc_acc[1], fc_acc[2], fc_acc[3], fc_acc[4], fc_acc[5], fc_acc[6], nq_, nc_, nk_);
#else
#define FC_T_INIT
#define FC_T(k)
#define FC_T_END(a, b, c)
#endif

typedef unsigned short fc_us2 __attribute__((ext_vector_type(2)));
typedef short fc_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fc_min(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(fc_us2, a), __builtin_bit_cast(fc_us2, b)));
}
__device__ __forceinline__ unsigned fc_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(fc_us2, a), __builtin_bit_cast(fc_us2, b)));
}
__device__ __forceinline__ unsigned fc_add(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (fc_s2)(__builtin_bit_cast(fc_s2, a) + __builtin_bit_cast(fc_s2, b)));
}
__device__ __forceinline__ unsigned fc_sub(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, (fc_s2)(__builtin_bit_cast(fc_s2, a) - __builtin_bit_cast(fc_s2, b)));
}
// sign bits (15, 31) of *dark / *bright set where the pixel of the pair passes the antipodal-pair pre-test of that polarity at
// threshold t2 = t | t << 16
__device__ __forceinline__ void fc_pair_test(unsigned T, unsigned Bt, unsigned L, unsigned R, unsigned v, unsigned t2, unsigned* dark,
                                             unsigned* bright) {
  const unsigned md = fc_max(fc_min(T, Bt), fc_min(L, R));  // darker: max(min(T,B), min(L,R)) < v - t
  const unsigned mb = fc_min(fc_max(T, Bt), fc_max(L, R));  // brighter: min(max(T,B), max(L,R)) > v + t
  *dark = fc_sub(fc_add(md, t2), v);
  *bright = fc_sub(fc_add(v, t2), mb);
}
// the four sign bits of an even (pixels 0, 2) and an odd (pixels 1, 3) pair as a nibble
__device__ __forceinline__ unsigned fc_nibble(unsigned fe, unsigned fo) {
  const unsigned tt = ((fe >> 15) & 0x00010001u) | ((fo >> 14) & 0x00020002u);  // bits 0, 1 (pixels 0, 1) and 16, 17 (pixels 2, 3)
  return (tt | (tt >> 14)) & 0xfu;
}
// a / n for 0 <= a < 4096, 1 <= n <= 256, by one reciprocal: (a + 0.5) / n is at least 0.5 / n away from an integer, far more than the
// error of v_rcp_f32 and the product (the integer division sequence is ~20 instructions)
__device__ __forceinline__ int fc_div(int a, int n) { return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)n)); }
// wave-wide inclusive scan of a small non-negative count (row_shr 1, 2, 4, 8 inside the 16-lane rows, then the row totals)
__device__ __forceinline__ int fc_wave_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);  // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1 and 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2 and 3
  return x;
}
__device__ __forceinline__ int fc_mbcnt(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// everything a lane wrote to LDS is visible to the other lanes of ITS wave behind this point (one wave runs in lock step and the
// LDS serves its instructions in order: only the compiler has to be kept from moving accesses across)
__device__ __forceinline__ void fc_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct FastMap {  // workgroup id -> (frame, cell): see fast_map()
  int units, parts, cells_per_part, grid;
};
inline FastMap fast_map(int B, int n_cells) {
  FastMap M;
  M.parts = 1;
  while (B * M.parts < 8 && M.parts < 8) M.parts *= 2;  // fewer than 8 frames: a frame's cells are cut into contiguous parts
  M.cells_per_part = (n_cells + M.parts - 1) / M.parts;
  M.units = B * M.parts;
  const int units_per_xcd = (M.units + 7) / 8;
  const int wgs_per_xcd = (units_per_xcd * M.cells_per_part + kFcWaves - 1) / kFcWaves;
  M.grid = 8 * wgs_per_xcd;
  return M;
}

__global__ __launch_bounds__(64 * kFcWaves) void k_fast_cells(const CellDev* __restrict__ cells, Lvl0 l0,
                                                             const uint8_t* __restrict__ pyr, size_t pyr_frame, int ini_th,
                                                             int min_th, int n_cells, FastMap M, int lds_wave, size_t slab_frame,
                                                             uint32_t* __restrict__ slab, int* __restrict__ cell_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int b, cell_id;
  {
    const int id = blockIdx.x, xcd = id & 7;
    const int slot = (id >> 3) * kFcWaves + wave;
    const int k = slot / M.cells_per_part, c = slot - k * M.cells_per_part;
    const int unit = k * 8 + xcd;
    if (unit >= M.units) return;
    b = unit / M.parts;
    cell_id = (unit - b * M.parts) * M.cells_per_part + c;
    if (cell_id >= n_cells) return;
  }
  const CellDev C = cells[cell_id];
  FC_T_INIT
  const int w = C.w, h = C.h;
  const int dw = w - 6, dh = h - 6;
  if (dw <= 0 || dh <= 0) {
    if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = 0;
    return;
  }
  const int sp = C.level == 0 ? l0.pitch : C.pitch;
  const uint8_t* src = C.level == 0 ? l0.base + (size_t)b * l0.frame_stride : pyr + (size_t)b * pyr_frame + C.plane_off;
  // Rows of the tile are whole words.  With a word-aligned plane (every pyramid level; a caller's level 0 whose pitch and base are
  // multiples of 4) a row is staged by aligned 4-byte loads from the word boundary in front of the cell, and pixel x of the cell sits
  // at byte al + x of its row; otherwise the tile is staged byte by byte with al = 0.
  const bool words = (sp & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
  src += (size_t)C.y0 * sp + C.x0;
  const int al = words ? (int)(reinterpret_cast<uintptr_t>(src) & 3) : 0;
  const int wp = (al + w + 3) & ~3, wpr = wp >> 2;
  uint8_t* tile = smem + (size_t)wave * lds_wave;             // [h][wp] (+ one word in front: the left window of word 0 is masked, not unread)
  uint8_t* sc = tile + (size_t)wp * h + 16;                   // [h][wp] arc score - 1 (0 = not a corner)
  unsigned short* list = reinterpret_cast<unsigned short*>(sc + (size_t)wp * h + 16);  // [dw * dh] offsets: pre-test list, corners, survivors
  // ---- staging: lanes as (row phase, word of the row); all loads of up to four passes in flight
  {
    const int rp = 64 / wpr;  // rows per pass (wpr <= 64: the host refuses wider cells)
    const int ry = fc_div(lane, wpr), x = lane - ry * wpr;
    const bool on = ry < rp;
    uint32_t* tw = reinterpret_cast<uint32_t*>(tile);
    uint32_t* sw = reinterpret_cast<uint32_t*>(sc);
    if (words) {
      const uint8_t* gp = src - al + 4 * x + (size_t)__mul24(ry, sp);  // row ry, then rp rows further per load
      const size_t gstep = (size_t)__mul24(rp, sp);
      int li = __mul24(ry, wpr) + x;
      const int lstep = __mul24(rp, wpr);
      constexpr int kInFlight = 12;  // every load of a typical cell (<= 60 rows of <= 12 words) before the first store: ONE round trip
      for (int y0 = ry; y0 < h; y0 += kInFlight * rp) {
        uint32_t v[kInFlight];
#pragma unroll
        for (int q = 0; q < kInFlight; q++) {
          if (on && y0 + q * rp < h) v[q] = *reinterpret_cast<const uint32_t*>(gp);
          gp += gstep;
        }
#pragma unroll
        for (int q = 0; q < kInFlight; q++) {
          if (on && y0 + q * rp < h) {
            tw[li] = v[q];
            sw[li] = 0u;
          }
          li += lstep;
        }
      }
    } else {
      for (int y = ry; on && y < h; y += rp) sw[y * wpr + x] = 0u;
      for (int i = lane; i < w * h; i += 64) {
        const int y = fc_div(i, w), xx = i - y * w;
        tile[y * wp + xx] = src[(size_t)y * sp + xx];
      }
    }
  }
  fc_wave_sync();
  FC_T(0)
  // ---- lane geometry of phase 0: column group gl (4 pixels = one word), rows 3 + rl + it * rstep
  const int ua = al + 3, ub = al + w - 4;            // first / last detection byte of a row
  const int gw0 = ua >> 2, ng = (ub >> 2) - gw0 + 1;  // word columns that hold detection pixels
  const int rstep = 64 / ng;
  int rl = fc_div(lane, ng);
  const int gl = lane - rl * ng;
  const bool lane_on = rl < rstep;
  if (!lane_on) rl = 0;  // (idle lanes walk the rows of row phase 0, masked: every LDS address stays where the active lanes read)
  const int u0 = 4 * (gw0 + gl);
  unsigned vmask = 0;  // which of the word's four pixels are detection pixels
#pragma unroll
  for (int j = 0; j < 4; j++) vmask |= (u0 + j >= ua && u0 + j <= ub) ? (1u << j) : 0u;
  if (!lane_on) vmask = 0;
  // (th_lo <= th_hi: the fallback pass runs only for a cell whose first pass kept nothing, and a threshold above the first one
  //  cannot find a corner the first one missed -- the reference's second cv::FAST call then returns nothing as well, src/ORBextractor.cc:825-827.
  //  Clamping keeps that result and the NMS's assumption that stale scores of pass 0 lie below pass 1's threshold.)
  const int th_hi = min(max(ini_th, 0), 255), th_lo = min(min(max(min_th, 0), 255), th_hi);
  const float inv_wp = 1.0f / (float)wp;
  int nk = 0;
  [[maybe_unused]] int fc_nq = 0, fc_nc = 0;
  // FAST with iniThFAST and, only if the cell produced nothing, again with minThFAST (src/ORBextractor.cc:815-827)
  for (int pass = 0; pass < 2; pass++) {
    const int th = pass == 0 ? th_hi : th_lo;
    const int T = max(th, 1);
    const unsigned t2 = (unsigned)th * 0x00010001u;
    // ---- phase 0 + compaction, 8 row steps (32 pass bits a lane) at a time.  A list entry = offset in the tile | polarity to
    //      test << 15 (0 dark, 1 bright) | 'if that is no corner, try the other polarity too' << 14.  A pixel that passed both (top darker, bottom brighter, left darker, right
    //      brighter) gets one entry per polarity, so that phase 1 tests every entry once -- unless the list (one slot per detection
    //      pixel, and every later chunk of rows must still fit one entry a pixel) would not hold them: that chunk then writes ONE
    //      entry a pixel carrying both bits, and phase 1 tries the second polarity where the first is no corner (`merged`;
    //      adversarial images only, the tests have three)
    int nq = 0;
    bool merged = false;
    for (int row0 = 0; row0 < dh; row0 += 8 * rstep) {
      const int nit = min(8, (dh - row0 + rstep - 1) / rstep);
      unsigned flags = 0, flagsb = 0;  // pass bits of the dark / bright pre-test, a nibble a row step
      // (rows past the cell, reached by the lanes of the last row phases in the last step, are read from what lies behind the tile
      // inside this wave's LDS -- at most rstep + 3 rows, OrbGeometry::fast_lds_wave covers them -- and masked)
      const uint32_t* rw = reinterpret_cast<const uint32_t*>(tile) + __mul24(3 + row0 + rl, wpr) + (gw0 + gl);
      int y = 3 + row0 + rl;
      for (int it = 0; it < nit; it++, y += rstep, rw += rstep * wpr) {
        const unsigned Tw = rw[-3 * wpr], Bw = rw[3 * wpr], Cw = rw[0], Cp = rw[-1], Cn = rw[1];
        const unsigned Lw = __builtin_amdgcn_alignbyte(Cw, Cp, 1);  // bytes u0 - 3 .. u0
        const unsigned Rw = __builtin_amdgcn_alignbyte(Cn, Cw, 3);  // bytes u0 + 3 .. u0 + 6
        const unsigned kE = 0x00ff00ffu;
        unsigned de, be, dod, bod;
        fc_pair_test(Tw & kE, Bw & kE, Lw & kE, Rw & kE, Cw & kE, t2, &de, &be);                                          // pixels 0, 2
        fc_pair_test((Tw >> 8) & kE, (Bw >> 8) & kE, (Lw >> 8) & kE, (Rw >> 8) & kE, (Cw >> 8) & kE, t2, &dod, &bod);  // pixels 1, 3
        const unsigned vm = y > h - 4 ? 0u : vmask;
        flags |= (fc_nibble(de, dod) & vm) << (4 * it);
        flagsb |= (fc_nibble(be, bod) & vm) << (4 * it);
      }
      FC_T(1)
      const unsigned both = flags | flagsb;
      int cnt = __builtin_popcount(flags) + __builtin_popcount(flagsb);
      int incl = fc_wave_scan(cnt);
      const int later_px = dw * max(dh - row0 - 8 * rstep, 0);  // detection pixels of the chunks still to come
      const bool merged_here = nq + __builtin_amdgcn_readlane(incl, 63) > dw * dh - later_px;  // (uniform)
      if (merged_here) {
        merged = true;
        cnt = __builtin_popcount(both);
        incl = fc_wave_scan(cnt);
      }
      int pos = nq + incl - cnt;
      nq += __builtin_amdgcn_readlane(incl, 63);
      const int obase = (3 + row0 + rl) * wp + u0, ostep = rstep * wp;
      if (merged_here) {
        unsigned rest = both;
        while (rest) {
          const int j = __builtin_ctz(rest);
          rest &= rest - 1;
          const unsigned dk = (flags >> j) & 1u, br = (flagsb >> j) & 1u;  // dark only: 0, bright only: 0x8000, both: 0x4000 (dark first)
          list[pos++] = (unsigned short)((obase + (j >> 2) * ostep + (j & 3)) | ((dk & br) << 14) | ((br & ~dk & 1u) << 15));
        }
      } else {
        while (flags) {
          const int j = __builtin_ctz(flags);
          flags &= flags - 1;
          list[pos++] = (unsigned short)(obase + (j >> 2) * ostep + (j & 3));
        }
        while (flagsb) {
          const int j = __builtin_ctz(flagsb);
          flagsb &= flagsb - 1;
          list[pos++] = (unsigned short)((obase + (j >> 2) * ostep + (j & 3)) | 0x8000);
        }
      }
      FC_T(2)
    }
    fc_wave_sync();
    // ---- phase 1: the 9-arc test over the list, corners compacted in place (a wave reads its 128 entries before it writes any)
    int nc = 0;
    for (int k0 = 0; k0 < nq; k0 += 128) {
      const int ka = k0 + lane, kb = k0 + 64 + lane;
      const bool ona = ka < nq, onb = kb < nq;
      int ea = ona ? list[ka] : 3 * wp + ua, eb = onb ? list[kb] : 3 * wp + ua;
      const int oa = ea & 0x3fff, ob = eb & 0x3fff;
      bool ca = arc_is_corner_pol(tile + oa, wp, th, ea >> 15) && ona;
      bool cb = arc_is_corner_pol(tile + ob, wp, th, eb >> 15) && onb;
      if (merged) {  // (an entry may ask for both polarities: a pixel that passed both pre-tests and is no dark corner is tried as a bright one)
        const bool seca = ona && !ca && (ea & 0x4000), secb = onb && !cb && (eb & 0x4000);
        if (__any(seca || secb)) {
          const bool c2a = arc_is_corner_pol(tile + oa, wp, th, 1), c2b = arc_is_corner_pol(tile + ob, wp, th, 1);
          ca = ca || (seca && c2a);
          cb = cb || (secb && c2b);
          if (seca && c2a) ea |= 0x8000;
          if (secb && c2b) eb |= 0x8000;
        }
        ea &= 0xbfff;  // a corner: offset | its polarity << 15
        eb &= 0xbfff;
      }
      const unsigned long long ma = __ballot(ca), mb = __ballot(cb);
      if (ca) list[nc + fc_mbcnt(ma)] = (unsigned short)ea;
      nc += (int)__popcll(ma);
      if (cb) list[nc + fc_mbcnt(mb)] = (unsigned short)eb;
      nc += (int)__popcll(mb);
    }
    fc_wave_sync();
    fc_nq = nq;
    fc_nc = nc;
    FC_T(3)
    // ---- phase 2: exact scores of the corners
    for (int k = lane; k < nc; k += 64) {
      const int e = list[k], o = e & 0x3fff;
      sc[o] = (uint8_t)(arc_score_pol(tile + o, wp, e >> 15) - 1);  // corner at th => S > th >= 0
    }
    fc_wave_sync();
    FC_T(4)
    // ---- phase 3: 3x3 strict-maximum suppression (scores below the threshold count as 0, the cell border is outside the
    //      detection area: nonmaxSuppression of cv::FAST on the cell image); survivors compacted in place
    nk = 0;
    for (int k0 = 0; k0 < nc; k0 += 64) {
      const int k = k0 + lane;
      bool keep = false;
      int o = 0;
      if (k < nc) {
        o = list[k] & 0x3fff;
        const uint8_t* q = sc + o;
        const int s = q[0];
        // a neighbour below the threshold counts as 0, and s >= T is above every such neighbour: "s > max of the eight" says the same
        const int m8 = max(max(max((int)q[-1], (int)q[1]), max((int)q[-wp - 1], (int)q[-wp])),
                           max(max((int)q[-wp + 1], (int)q[wp - 1]), max((int)q[wp], (int)q[wp + 1])));
        keep = s >= T && s > m8;
      }
      const unsigned long long m = __ballot(keep);
      if (keep) list[nk + fc_mbcnt(m)] = (unsigned short)o;
      nk += (int)__popcll(m);
    }
    fc_wave_sync();
    FC_T(5)
    if (nk > 0) break;
  }
  // ---- phase 4: raster order = ascending offset.  The survivors mark their offsets in a bitmap (the tile is no longer needed: the
  //      bitmap and the words' prefix counts take its place); a survivor's rank = set bits below its own.
  uint32_t* out = slab + (size_t)b * slab_frame + C.slab_off;
  const int offx = C.x0 - 16 - al, offy = C.y0 - 16;  // + j*wCell, + i*hCell (src/ORBextractor.cc:847-848)
  if (nk > 0) {
    const int nbw = (wp * h + 31) >> 5;  // bitmap words
    uint32_t* bm = reinterpret_cast<uint32_t*>(tile);
    unsigned short* pre = reinterpret_cast<unsigned short*>(bm + nbw);  // (wp * h / 8 + wp * h / 16 bytes <= wp * h)
    for (int i = lane; i < nbw; i += 64) bm[i] = 0u;
    fc_wave_sync();
    for (int k = lane; k < nk; k += 64) {
      const int o = list[k];
      atomicOr(&bm[o >> 5], 1u << (o & 31));
    }
    fc_wave_sync();
    int run = 0;
    for (int i0 = 0; i0 < nbw; i0 += 64) {
      const int i = i0 + lane;
      const int c = i < nbw ? __builtin_popcount(bm[i]) : 0;
      const int incl = fc_wave_scan(c);
      if (i < nbw) pre[i] = (unsigned short)(run + incl - c);
      run += __builtin_amdgcn_readlane(incl, 63);
    }
    fc_wave_sync();
    for (int k = lane; k < nk; k += 64) {
      const int o = list[k];
      const int r = pre[o >> 5] + __builtin_popcount(bm[o >> 5] & ((1u << (o & 31)) - 1u));
      const int y = (int)(((float)o + 0.5f) * inv_wp), x = o - y * wp;
      out[r] = (uint32_t)(x + offx) | ((uint32_t)(y + offy) << 12) | ((uint32_t)sc[o] << 24);
    }
  }
  if (lane == 0) cell_cnt[(size_t)b * n_cells + cell_id] = nk;
  FC_T(6)
  FC_T_END(fc_nq, fc_nc, nk)
}

// ------------------------------------------------------------------------------------------------
// k_cand_pack: one workgroup per frame; concatenates the per-cell slabs into the dense, deterministic
// (level, cell row-major, raster) order the reference feeds to DistributeOctTree.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cand_pack(const LevelDev* __restrict__ levels,
                                                   const CellDev* __restrict__ cells, int nlevels, int n_cells,
                                                   size_t slab_frame, const uint32_t* __restrict__ slab,
                                                   const int* __restrict__ cell_cnt, size_t cand_frame,
                                                   uint32_t* __restrict__ cand, int* __restrict__ cand_off) {
  __shared__ int s_scan[256];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* cnt = cell_cnt + (size_t)b * n_cells;
  const uint32_t* sl = slab + (size_t)b * slab_frame;
  uint32_t* out = cand + (size_t)b * cand_frame;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int l = 0; l < nlevels; l++) {
    const LevelDev L = levels[l];
    if (tid == 0) cand_off[(size_t)b * (nlevels + 1) + l] = s_base;
    for (int c0 = 0; c0 < L.n_cells; c0 += 256) {
      const int c = c0 + tid;
      const int my = c < L.n_cells ? cnt[L.cell_base + c] : 0;
      s_scan[tid] = my;
      __syncthreads();
      for (int ofs = 1; ofs < 256; ofs <<= 1) {
        const int v = tid >= ofs ? s_scan[tid - ofs] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
      }
      const int base = s_base + s_scan[tid] - my;
      if (my > 0) {
        const uint32_t* s = sl + cells[L.cell_base + c].slab_off;
        for (int i = 0; i < my; i++) out[base + i] = s[i];
      }
      __syncthreads();
      if (tid == 255) s_base += s_scan[255];
      __syncthreads();
    }
  }
  if (tid == 0) cand_off[(size_t)b * (nlevels + 1) + nlevels] = s_base;
}

// ------------------------------------------------------------------------------------------------
// k_octree: ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:567-768) on the device, one workgroup
// per (frame, level).  The std::list is represented by an array in list order; a node is a box plus a contiguous
// range of a key permutation (children always partition their parent's range in place, stably).
//   * "Loop A" (every multi-key node is split in one pass, :621-683) is one block-wide sweep: every key computes its
//     quadrant, a block scan of packed per-quadrant counters gives its stable rank, per-node scans give the new list
//     positions (children of later nodes come first, children n4..n1, single-key nodes keep their relative order
//     behind them: the push_front / erase semantics of the reference).
//   * "Loop B" (split the largest nodes first until the quota is reached, :689-744) sorts the (size, UL.x) pairs with
//     a replica of libstdc++'s std::sort on one lane (tie order matters), evaluates all candidate splits in parallel,
//     finds the break index with a scan and applies the prefix.
//   * finally the first maximum-response key of every node is emitted in list order (:751-765).
// Keys stay in global memory (a ping-pong pair of key arrays in list order: every thread walks a contiguous piece, served by
// L1 / L2), the node arrays live in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kOctThreads = 256;
constexpr int kOctMaxNodes = 768;
inline size_t oct_lds_bytes(int node_cap) { return (size_t)node_cap * (2 * 16 + 2 * 16 + 8 + 2 * 8 + 8 + 1) + 16; }

struct ONode {
  short x0, x1, y0, y1;
  int kb, ke;
};
struct OVs {  // vSizeAndPointerToNode entry
  int size;
  short x0, node;
};
struct U128 {
  unsigned long long lo, hi;  // four 32-bit per-quadrant counters
};
__device__ __forceinline__ U128 u128_add(U128 a, U128 b) { return U128{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ unsigned u128_field(U128 a, int q) {
  const unsigned long long w = q < 2 ? a.lo : a.hi;
  return (unsigned)((q & 1) ? (w >> 32) : (w & 0xffffffffull));
}
__device__ __forceinline__ U128 u128_one(int q) {
  U128 r{0, 0};
  const unsigned long long v = (q & 1) ? (1ull << 32) : 1ull;
  if (q < 2)
    r.lo = v;
  else
    r.hi = v;
  return r;
}

// exclusive block scan (256 threads) of one U128 per thread; *total = sum over the block
__device__ U128 oct_scan_u128(U128 v, U128* s_w /*[4]*/, U128* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  U128 incl = v;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const unsigned long long lo = __shfl_up(incl.lo, ofs, 64), hi = __shfl_up(incl.hi, ofs, 64);
    if (lane >= ofs) {
      incl.lo += lo;
      incl.hi += hi;
    }
  }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  U128 base{0, 0}, tot{0, 0};
  for (int w = 0; w < 4; w++) {
    if (w < wave) base = u128_add(base, s_w[w]);
    tot = u128_add(tot, s_w[w]);
  }
  *total = tot;
  return U128{base.lo + incl.lo - v.lo, base.hi + incl.hi - v.hi};
}
// exclusive block scan of one packed u64 (three 20-bit fields) per thread
__device__ unsigned long long oct_scan_u64(unsigned long long v, unsigned long long* s_w /*[4]*/, unsigned long long* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const unsigned long long t = __shfl_up(incl, ofs, 64);
    if (lane >= ofs) incl += t;
  }
  __syncthreads();
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int w = 0; w < 4; w++) {
    if (w < wave) base += s_w[w];
    tot += s_w[w];
  }
  *total = tot;
  return base + incl - v;
}

__device__ __forceinline__ int oct_quadrant(const ONode& nd, uint32_t key) {  // ExtractorNode::DivideNode :502-550
  const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
  const int x = (int)(key & 0xfff), y = (int)((key >> 12) & 0xfff);
  return (x < nd.x0 + halfX ? 0 : 1) + (y < nd.y0 + halfY ? 0 : 2);
}
__device__ __forceinline__ ONode oct_child(const ONode& nd, int q) {
  const int halfX = (int)ceilf((float)(nd.x1 - nd.x0) / 2), halfY = (int)ceilf((float)(nd.y1 - nd.y0) / 2);
  const int xm = nd.x0 + halfX, ym = nd.y0 + halfY;
  ONode c;
  c.x0 = (short)((q & 1) ? xm : nd.x0);
  c.x1 = (short)((q & 1) ? nd.x1 : xm);
  c.y0 = (short)((q & 2) ? ym : nd.y0);
  c.y1 = (short)((q & 2) ? nd.y1 : ym);
  c.kb = c.ke = 0;
  return c;
}

// k_octree's dynamic LDS (declared here for oct_sort_wave)
extern __shared__ __align__(16) unsigned char oct_lds[];

// (out of line: the sort's own register arrays are then not part of k_octree's live state around the call.  Its work arrays are
// named by their offset in oct_lds, not by pointers: pointer arguments would arrive as GENERIC ones and every access of the sort
// would be a FLAT instruction)
__device__ __attribute__((noinline)) void oct_sort_wave(unsigned lds_off, int n) {
  unsigned* K = reinterpret_cast<unsigned*>(oct_lds + lds_off);
  unsigned short* Pm = reinterpret_cast<unsigned short*>(K + n);
  vqs::wave_std_sort<unsigned>(K, Pm, Pm + n, Pm + 2 * n, Pm + 3 * n, Pm + 4 * n, n);
}

#ifdef GFS_OCT_TIMING
#define OCT_T_INIT long long ot_acc[6] = {0, 0, 0, 0, 0, 0}, ot_last = clock64(); int ot_cnt[2] = {0, 0};
#define OCT_T(k) { const long long _n = clock64(); ot_acc[k] += _n - ot_last; ot_last = _n; }
#define OCT_CNT(k) ot_cnt[k]++;
#define OCT_T_END if (threadIdx.x == 0 && blockIdx.x < 16) printf("OCTT blk=%d n=%d nn=%d gather=%lld init=%lld sweeps=%lld(%d) sort=%lld(%d) loopB=%lld emit=%lld\n", (int)blockIdx.x, n, nn, ot_acc[0], ot_acc[1], ot_acc[2], ot_cnt[0], ot_acc[3], ot_cnt[1], ot_acc[4], ot_acc[5]);
#else
#define OCT_T_INIT
#define OCT_T(k)
#define OCT_CNT(k)
#define OCT_T_END
#endif
__global__ __launch_bounds__(kOctThreads) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_octree(const LevelDev* __restrict__ levels, int nlevels,
                                                        const uint32_t* __restrict__ cand, const int* __restrict__ cand_off,
                                                        size_t cand_frame, uint32_t* __restrict__ perm0,
                                                        uint32_t* __restrict__ perm1, unsigned short* __restrict__ seg0,
                                                        unsigned short* __restrict__ seg1, const int* __restrict__ kept_off,
                                                        int kp_cap, uint32_t* __restrict__ kept, int* __restrict__ kept_cnt,
                                                        int node_cap, const CellDev* __restrict__ cells,
                                                        const int* __restrict__ cell_cnt, int n_cells,
                                                        const uint32_t* __restrict__ slab) {
  // node tables in dynamic LDS, node_cap entries each (the largest level's quota + slack, oct_lds_bytes): the kernel is a chain
  // of short dependent phases, so it is the number of workgroups a CU holds at once that sets its speed (97 B a node: 6
  // workgroups a CU for 1000 features; the fixed 768-entry tables allowed 2)
  // (the two node buffers and the two list buffers are addressed as base + index * node_cap: a run-time pick between two
  // pointers kept in an array would turn every access into a FLAT one — 64-bit addresses, no LDS offsets)
  ONode* const s_nodes0 = (ONode*)oct_lds;
  auto s_nodes = [&](int w) { return s_nodes0 + (size_t)w * node_cap; };
  U128* s_start = (U128*)(s_nodes0 + 2 * (size_t)node_cap);
  U128* s_end = s_start + node_cap;
  unsigned long long* s_npre = (unsigned long long*)(s_end + node_cap);  // per node: exclusive prefix of (children | multi << 20 | single << 40)
  OVs* const s_vs0 = (OVs*)(s_npre + node_cap);
  auto s_vs = [&](int w) { return s_vs0 + (size_t)w * node_cap; };
  unsigned short(*s_newidx)[4] = (unsigned short(*)[4])(s_vs0 + 2 * (size_t)node_cap);
  unsigned char* s_proc = (unsigned char*)(s_newidx + node_cap);
  __shared__ U128 s_w128[4];
  __shared__ gfs::SortFrame s_stack[32];  // 2 lg(768) + 1 = 19 pending parts at most
  __shared__ unsigned long long s_w64[4];
  __shared__ int s_ctl[8];
  const int tid = threadIdx.x;
  OCT_T_INIT
  const int l = blockIdx.x % nlevels, b = blockIdx.x / nlevels;
  const LevelDev L = levels[l];
  // cells != NULL: the level's candidates are taken straight from the per-cell slabs of k_fast_cells, in the order the reference
  // feeds them to DistributeOctTree (cells row-major, raster order inside a cell) — what k_cand_pack would lay out; the working
  // arrays of the level then live at the level's slab offset.  cells == NULL: a dense list (cand, cand_off) from the caller.
  __shared__ int s_gscan[kOctThreads];
  __shared__ unsigned s_coff[kOctThreads];
  __shared__ int s_gbase;
  size_t region = 0;
  int n = 0;
  if (cells) {
    region = cells[L.cell_base].slab_off;
  } else {
    const int* off = cand_off + (size_t)b * (nlevels + 1);
    n = off[l + 1] - off[l];
    region = (size_t)off[l];
  }
  // (a pick between two pointers, not an array of them: the loads stay GLOBAL instructions)
  uint32_t* const perm_a = perm0 + (size_t)b * cand_frame + region;
  uint32_t* const perm_b = perm1 + (size_t)b * cand_frame + region;
  unsigned short* const seg_a = seg0 + (size_t)b * cand_frame + region;
  unsigned short* const seg_b = seg1 + (size_t)b * cand_frame + region;
  auto perm = [&](int w) { return w ? perm_b : perm_a; };
  auto seg = [&](int w) { return w ? seg_b : seg_a; };
  if (cells) {
    const int* cnt = cell_cnt + (size_t)b * n_cells + L.cell_base;
    const uint32_t* sl = slab + (size_t)b * cand_frame;
    if (tid == 0) s_gbase = 0;
    __syncthreads();
    for (int c0 = 0; c0 < L.n_cells; c0 += kOctThreads) {
      const int ci = c0 + tid;
      const int my = ci < L.n_cells ? cnt[ci] : 0;
      {  // inclusive prefix of the cells' counts (wave shuffles + one exchange between the four waves)
        unsigned long long tot64;
        const unsigned long long excl = oct_scan_u64((unsigned long long)my, s_w64, &tot64);
        s_gscan[tid] = (int)excl + my;
      }
      // the chunk's candidates as one flat range [0, tot): element e belongs to the first cell whose inclusive prefix exceeds e
      // (a search over the scanned counts in LDS), so every thread copies the same number of elements, eight loads in flight,
      // and neighbouring threads write neighbouring slots.  (A thread per cell copied its own cell's candidates one batch after
      // the other: as many round trips as the fullest cell needs, 35 - 80 k of a workgroup's ~230 k cycles, clock64 in round 4.)
      s_coff[tid] = ci < L.n_cells ? cells[L.cell_base + ci].slab_off : 0u;
      __syncthreads();
      const int tot = s_gscan[kOctThreads - 1];
      for (int e0 = tid; e0 < tot; e0 += 8 * kOctThreads) {
        uint32_t t[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int e = e0 + u * kOctThreads;
          t[u] = 0u;
          if (e < tot) {
            int lo = 0, hi = kOctThreads - 1;  // first cell with s_gscan[cell] > e
#pragma unroll
            for (int st = 0; st < 8; st++) {
              const int mid = (lo + hi) >> 1;
              if (s_gscan[mid] > e) hi = mid;
              else lo = mid + 1;
            }
            const int excl = lo > 0 ? s_gscan[lo - 1] : 0;
            t[u] = sl[s_coff[lo] + (unsigned)(e - excl)];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int e = e0 + u * kOctThreads;
          if (e < tot) {
            perm(0)[s_gbase + e] = t[u];
            seg(0)[s_gbase + e] = 0;
          }
        }
      }
      __syncthreads();
      if (tid == kOctThreads - 1) s_gbase += s_gscan[kOctThreads - 1];
      __syncthreads();
    }
    n = s_gbase;
  }
  OCT_T(0)
  const uint32_t* c = cells ? nullptr : cand + (size_t)b * cand_frame + region;
  if (n <= 0) {
    if (tid == 0) kept_cnt[(size_t)b * nlevels + l] = 0;
    return;
  }
  const int N = L.quota;
  const int width = L.max_bx - 16, height = L.max_by - 16;
  int nIni = (int)roundf((float)width / (float)height);  // :573
  if (nIni == 0) nIni = 1;
  const float hX = (float)width / (float)nIni;
  const int chunk = (n + kOctThreads - 1) / kOctThreads;
  const int p0 = min(tid * chunk, n), p1 = min(p0 + chunk, n);
  if (c)
    for (int p = p0; p < p1; p++) {
      perm(0)[p] = c[p];  // the permutation arrays carry the keys themselves: no gather through an index in the sweeps
      seg(0)[p] = 0;
    }
  if (tid == 0) {
    ONode root;
    root.x0 = 0;
    root.x1 = (short)width;
    root.y0 = 0;
    root.y1 = (short)height;
    root.kb = 0;
    root.ke = n;
    s_nodes(0)[0] = root;
  }
  __syncthreads();
  int cur = 0, pc = 0, nn = 1;  // node-buffer index, permutation-buffer index, number of nodes in the list
  int nvs = 0, vcur = 0;

  // One split sweep over the whole list.  init = true: the virtual root is cut into the nIni initial nodes
  // (push_back order, :586-604); otherwise every node with > 1 keys is divided (push_front order).
  // walks keys[kb, ke) in order with eight loads in flight (f must not write what it is still to read)
  auto for_keys = [&](const uint32_t* keys, int kb, int ke, auto&& f) {
    for (int p = kb; p < ke; p += 8) {
      uint32_t t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = p + u < ke ? keys[p + u] : 0u;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (p + u < ke) f(p + u, t[u]);
    }
  };
  // (kOctChunk keys a thread or fewer, i.e. n <= 4096: the thread's keys and node indices are loaded ONCE per sweep, all loads in
  // flight together, and the three passes over them run from registers — a global round trip per key and pass made a sweep a
  // chain of ~30 of them, ~2 us each)
  constexpr int kOctChunk = 16;
  const bool cached = chunk <= kOctChunk;
  auto sweep = [&](bool init) {
    const ONode* nodes = s_nodes(cur);
    ONode* nxt = s_nodes(cur ^ 1);
    const uint32_t* pa = perm(pc);
    const unsigned short* sa = seg(pc);
    uint32_t kreg[kOctChunk];
    int nreg[kOctChunk];
    unsigned qbits = 0, abits = 0;  // per cached key: its quadrant (2 bits), and whether its node is being divided
    if (cached) {  // (unconditional: one base address, immediate offsets; the arrays are padded by a chunk, what lies beyond p1 is not used)
      const uint32_t* pa0 = pa + p0;
      const unsigned short* sa0 = sa + p0;
#pragma unroll
      for (int j = 0; j < kOctChunk; j++) {
        kreg[j] = pa0[j];
        nreg[j] = (int)sa0[j];
      }
    }
    // 1. per-key quadrant counters, exclusive block scan
    U128 local{0, 0};
    if (cached) {
#pragma unroll
      for (int j = 0; j < kOctChunk; j++) {
        if (p0 + j < p1) {
          const ONode nd = nodes[nreg[j]];
          if (init || nd.ke - nd.kb > 1) {
            const int q = init ? min((int)((float)(kreg[j] & 0xfff) / hX), nIni - 1) : oct_quadrant(nd, kreg[j]);
            local = u128_add(local, u128_one(q));
            qbits |= (unsigned)q << (2 * j);
            abits |= 1u << j;
          }
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // four keys' node reads in flight, not sixteen (registers)
      }
    } else {
      for (int p = p0; p < p1; p++) {
        const ONode nd = nodes[sa[p]];
        if (!init && nd.ke - nd.kb <= 1) continue;
        const uint32_t key = pa[p];
        const int q = init ? min((int)((float)(key & 0xfff) / hX), nIni - 1) : oct_quadrant(nd, key);
        local = u128_add(local, u128_one(q));
      }
    }
    U128 total;
    const U128 base = oct_scan_u128(local, s_w128, &total);
    U128 run = base;
    if (cached) {
#pragma unroll
      for (int j = 0; j < kOctChunk; j++) {
        const int p = p0 + j;
        if (p < p1) {
          const int ni = nreg[j];
          const int kb = nodes[ni].kb, ke = nodes[ni].ke;
          if (p == kb) s_start[ni] = run;
          if ((abits >> j) & 1u) run = u128_add(run, u128_one((int)((qbits >> (2 * j)) & 3u)));
          if (p + 1 == ke) s_end[ni] = run;
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int p = p0; p < p1; p++) {
        const int ni = sa[p];
        const ONode nd = nodes[ni];
        if (p == nd.kb) s_start[ni] = run;
        if (init || nd.ke - nd.kb > 1) {
          const uint32_t key = pa[p];
          const int q = init ? min((int)((float)(key & 0xfff) / hX), nIni - 1) : oct_quadrant(nd, key);
          run = u128_add(run, u128_one(q));
        }
        if (p + 1 == nd.ke) s_end[ni] = run;
      }
    }
    __syncthreads();
    // 2. per-node child counts -> packed (children, multi, single) and its exclusive scan over the list
    const int nchunk = (nn + kOctThreads - 1) / kOctThreads;
    const int i0 = min(tid * nchunk, nn), i1 = min(i0 + nchunk, nn);
    unsigned long long lsum = 0;
    for (int i = i0; i < i1; i++) {
      const ONode nd = nodes[i];
      unsigned long long v;
      if (init || nd.ke - nd.kb > 1) {
        int cc = 0, mm = 0;
        for (int q = 0; q < 4; q++) {
          const unsigned cq = u128_field(s_end[i], q) - u128_field(s_start[i], q);
          cc += cq > 0;
          mm += cq > 1;
        }
        v = (unsigned long long)cc | ((unsigned long long)mm << 20);
      } else {
        v = 1ull << 40;
      }
      s_npre[i] = v;
      lsum += v;
    }
    unsigned long long ntot;
    unsigned long long nbase = oct_scan_u64(lsum, s_w64, &ntot);
    const int T = (int)(ntot & 0xfffff), Mtot = (int)((ntot >> 20) & 0xfffff), Z = (int)(ntot >> 40);
    // 3. new nodes, new index table, vSizeAndPointerToNode in creation order
    for (int i = i0; i < i1; i++) {
      const unsigned long long v = s_npre[i];
      const int A = (int)(nbase & 0xfffff), Mx = (int)((nbase >> 20) & 0xfffff), Zx = (int)(nbase >> 40);
      nbase += v;
      const ONode nd = nodes[i];
      if (init || nd.ke - nd.kb > 1) {
        const int ci = (int)(v & 0xfffff);
        unsigned cnt[4];
        for (int q = 0; q < 4; q++) cnt[q] = u128_field(s_end[i], q) - u128_field(s_start[i], q);
        int kb = nd.kb, before = 0, mrank = 0;
        for (int q = 0; q < 4; q++) {
          if (cnt[q] == 0) {
            s_newidx[i][q] = 0;
            continue;
          }
          // init: push_back order (ascending); else children of later nodes first, n4..n1 inside a node
          const int pos = init ? (A + before) : (T - A - ci + (ci - 1 - before));
          ONode ch;
          if (init) {
            ch.x0 = (short)(int)(hX * (float)q);
            ch.x1 = (short)(int)(hX * (float)(q + 1));
            ch.y0 = 0;
            ch.y1 = (short)height;
          } else {
            ch = oct_child(nd, q);
          }
          ch.kb = kb;
          ch.ke = kb + (int)cnt[q];
          nxt[pos] = ch;
          s_newidx[i][q] = (unsigned short)pos;
          if (cnt[q] > 1) {
            s_vs(vcur)[Mx + mrank] = OVs{(int)cnt[q], ch.x0, (short)pos};
            mrank++;
          }
          kb += (int)cnt[q];
          before++;
        }
      } else {
        nxt[T + Zx] = nd;
        s_newidx[i][0] = (unsigned short)(T + Zx);
      }
    }
    __syncthreads();
    // 4. stable scatter of the keys into their child ranges
    uint32_t* pb = perm(pc ^ 1);
    unsigned short* sb = seg(pc ^ 1);
    run = base;
    if (cached) {
#pragma unroll
      for (int j = 0; j < kOctChunk; j++) {
        const int p = p0 + j;
        if (p < p1) {
          const int ni = nreg[j];
          if ((abits >> j) & 1u) {
            const int q = (int)((qbits >> (2 * j)) & 3u);
            const U128 st = s_start[ni], en = s_end[ni];
            unsigned ofs = 0;
            for (int qq = 0; qq < q; qq++) ofs += u128_field(en, qq) - u128_field(st, qq);
            const int dst = nodes[ni].kb + (int)ofs + (int)(u128_field(run, q) - u128_field(st, q));
            pb[dst] = kreg[j];
            sb[dst] = s_newidx[ni][q];
            run = u128_add(run, u128_one(q));
          } else {
            pb[p] = kreg[j];
            sb[p] = s_newidx[ni][0];
          }
        }
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      for (int p = p0; p < p1; p++) {
        const int ni = sa[p];
        const ONode nd = nodes[ni];
        if (init || nd.ke - nd.kb > 1) {
          const uint32_t key = pa[p];
          const int q = init ? min((int)((float)(key & 0xfff) / hX), nIni - 1) : oct_quadrant(nd, key);
          unsigned ofs = 0;
          for (int qq = 0; qq < q; qq++) ofs += u128_field(s_end[ni], qq) - u128_field(s_start[ni], qq);
          const int dst = nd.kb + (int)ofs + (int)(u128_field(run, q) - u128_field(s_start[ni], q));
          pb[dst] = pa[p];
          sb[dst] = s_newidx[ni][q];
          run = u128_add(run, u128_one(q));
        } else {
          pb[p] = pa[p];
          sb[p] = s_newidx[ni][0];
        }
      }
    }
    __syncthreads();
    cur ^= 1;
    pc ^= 1;
    nn = T + Z;
    nvs = Mtot;
  };

  sweep(true);
  OCT_T(1)
  bool finish = false;
  while (!finish) {
    const int prev = nn;
    vcur ^= 1;  // vSizeAndPointerToNode.clear(): entries are rebuilt by the sweep into the other buffer
    sweep(false);
    OCT_T(2) OCT_CNT(0)
    if (nn >= N || nn == prev) {
      finish = true;
    } else if (nn + nvs * 3 > N) {
      while (!finish) {  // :690-744
        const int prev2 = nn;
        OVs* vs = s_vs(vcur);
        const int V = nvs;
        ONode* nodes = s_nodes(cur);
        // sort(vSizeAndPointerToNode) :697-698 with compareNodes :552-565 (size, then UL.x: many entries are equivalent, so the
        // result is libstdc++'s own sequence of moves).  One wave replays std::sort on the packed keys (size << 12 | x), the
        // entries follow their keys; its LDS scratch is the sweeps' s_start / s_end, idle here.
        if (V <= 1024 && (size_t)V * 12 + 256 <= (size_t)node_cap * 32) {
          unsigned* sk = (unsigned*)s_start;
          unsigned short* spm = (unsigned short*)(sk + V);
          for (int i = tid; i < V; i += kOctThreads) {
            sk[i] = ((unsigned)vs[i].size << 12) | (unsigned)vs[i].x0;  // x0 in [0, 4096), size <= n < 2^20
            spm[i] = (unsigned short)i;
          }
          __syncthreads();
          if (tid < 64) oct_sort_wave((unsigned)((unsigned char*)s_start - oct_lds), V);  // K, Pm, l0, l1, cl, st laid out as above
          __syncthreads();
          OVs mine[4];  // V <= 1024 = 4 x 256
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (tid + u * kOctThreads < V) mine[u] = vs[spm[tid + u * kOctThreads]];
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (tid + u * kOctThreads < V) vs[tid + u * kOctThreads] = mine[u];
        } else if (tid == 0) {
          gfs::replica_std_sort_on(vs, vs + V, [](const OVs& a, const OVs& e) {
            if (a.size < e.size) return true;
            if (a.size > e.size) return false;
            return a.x0 < e.x0;
          }, s_stack, 32);
        }
        for (int i = tid; i < nn; i += kOctThreads) s_proc[i] = 0;
        __syncthreads();
        OCT_T(3) OCT_CNT(1)
        // candidate splits of every entry; processing order k = 0.. is j = V-1-k (from the back)
        const int vchunk = (V + kOctThreads - 1) / kOctThreads;
        const int k0 = min(tid * vchunk, V), k1 = min(k0 + vchunk, V);
        unsigned long long lsum = 0;
        // (a thread with one entry whose node has at most kOctChunk keys — the rule — loads them once, all loads in flight, and
        // both the evaluation here and the partition below run from registers)
        uint32_t nkey[kOctChunk];
        bool have = false;
        for (int k = k0; k < k1; k++) {
          const int j = V - 1 - k;
          const ONode nd = nodes[vs[j].node];
          unsigned long long pk = 0;  // four 16-bit quadrant counters (an indexed private array would live in scratch memory)
          if (k1 - k0 == 1 && nd.ke - nd.kb <= kOctChunk) {
            have = true;
#pragma unroll
            for (int u = 0; u < kOctChunk; u++) nkey[u] = nd.kb + u < nd.ke ? perm(pc)[nd.kb + u] : 0u;
#pragma unroll
            for (int u = 0; u < kOctChunk; u++)
              if (nd.kb + u < nd.ke) pk += 1ull << (16 * oct_quadrant(nd, nkey[u]));
          } else {
            for_keys(perm(pc), nd.kb, nd.ke, [&](int, uint32_t key) { pk += 1ull << (16 * oct_quadrant(nd, key)); });
          }
          int cc = 0, mm = 0;
          for (int q = 0; q < 4; q++) {
            const int cq = (int)((pk >> (16 * q)) & 0xffff);
            cc += cq > 0;
            mm += cq > 1;
          }
          s_npre[k] = (unsigned long long)cc | ((unsigned long long)mm << 20);
          lsum += s_npre[k];
        }
        unsigned long long vtot;
        unsigned long long vbase = oct_scan_u64(lsum, s_w64, &vtot);
        // break index t: the reference stops right after the split that makes lNodes.size() >= N
        if (tid == 0) s_ctl[0] = V;
        __syncthreads();
        {
          unsigned long long run = vbase;
          for (int k = k0; k < k1; k++) {
            run += s_npre[k];
            const int size_after = nn + (int)(run & 0xfffff) - (k + 1);
            if (size_after >= N) atomicMin(&s_ctl[0], k + 1);
          }
        }
        __syncthreads();
        const int t = s_ctl[0];
        // totals over the processed prefix
        unsigned long long lsum2 = 0;
        for (int k = k0; k < k1; k++)
          if (k < t) lsum2 += s_npre[k];
        unsigned long long ptot;
        unsigned long long pbase = oct_scan_u64(lsum2, s_w64, &ptot);
        const int T = (int)(ptot & 0xfffff), Mtot = (int)((ptot >> 20) & 0xfffff);
        ONode* nxt = s_nodes(cur ^ 1);
        OVs* vnew = s_vs(vcur ^ 1);
        for (int k = k0; k < k1; k++) {
          if (k >= t) break;
          const int j = V - 1 - k;
          const int ni = vs[j].node;
          const ONode nd = nodes[ni];
          s_proc[ni] = 1;
          const unsigned long long v = s_npre[k];
          const int A = (int)(pbase & 0xfffff), Mx = (int)((pbase >> 20) & 0xfffff);
          pbase += v;
          const int ci = (int)(v & 0xfffff);
          // stable 4-way partition of this node's keys (through the other permutation buffer, then back)
          uint32_t* pa = perm(pc);  // partitioned through the other buffer and copied back: pc does not flip here
          uint32_t* pb = perm(pc ^ 1);
          unsigned long long pk = 0;  // four 16-bit quadrant counters, then the four running positions
          if (have) {
#pragma unroll
            for (int u = 0; u < kOctChunk; u++)
              if (nd.kb + u < nd.ke) pk += 1ull << (16 * oct_quadrant(nd, nkey[u]));
          } else {
            for_keys(pa, nd.kb, nd.ke, [&](int, uint32_t key) { pk += 1ull << (16 * oct_quadrant(nd, key)); });
          }
          const int cnt[4] = {(int)(pk & 0xffff), (int)((pk >> 16) & 0xffff), (int)((pk >> 32) & 0xffff), (int)(pk >> 48)};
          unsigned long long pos = (unsigned long long)cnt[0] << 16 | (unsigned long long)(cnt[0] + cnt[1]) << 32 |
                                   (unsigned long long)(cnt[0] + cnt[1] + cnt[2]) << 48;  // offsets from nd.kb
          if (have) {  // all keys are in registers: partitioned in place
#pragma unroll
            for (int u = 0; u < kOctChunk; u++)
              if (nd.kb + u < nd.ke) {
                const int sh = 16 * oct_quadrant(nd, nkey[u]);
                pa[nd.kb + (int)((pos >> sh) & 0xffff)] = nkey[u];
                pos += 1ull << sh;
              }
          } else {
            for_keys(pa, nd.kb, nd.ke, [&](int, uint32_t key) {
              const int sh = 16 * oct_quadrant(nd, key);
              pb[nd.kb + (int)((pos >> sh) & 0xffff)] = key;
              pos += 1ull << sh;
            });
            for_keys(pb, nd.kb, nd.ke, [&](int p, uint32_t key) { pa[p] = key; });
          }
          int kb = nd.kb, before = 0, mrank = 0;
          for (int q = 0; q < 4; q++) {
            if (cnt[q] == 0) continue;
            const int posn = T - A - ci + (ci - 1 - before);
            ONode ch = oct_child(nd, q);
            ch.kb = kb;
            ch.ke = kb + cnt[q];
            nxt[posn] = ch;
            if (cnt[q] > 1) {
              vnew[Mx + mrank] = OVs{cnt[q], ch.x0, (short)posn};
              mrank++;
            }
            kb += cnt[q];
            before++;
          }
        }
        __syncthreads();
        // surviving nodes keep their relative order behind the new children
        const int nchunk = (nn + kOctThreads - 1) / kOctThreads;
        const int i0 = min(tid * nchunk, nn), i1 = min(i0 + nchunk, nn);
        unsigned long long ls = 0;
        for (int i = i0; i < i1; i++) ls += s_proc[i] ? 0 : 1;
        unsigned long long stot;
        unsigned long long sbase = oct_scan_u64(ls, s_w64, &stot);
        for (int i = i0; i < i1; i++) {
          if (s_proc[i]) continue;
          nxt[T + (int)sbase] = nodes[i];
          sbase++;
        }
        __syncthreads();
        cur ^= 1;
        vcur ^= 1;
        nn = T + (int)stot;
        nvs = Mtot;
        OCT_T(4)
        if (nn >= N || nn == prev2) finish = true;
      }
    }
  }
  // retain the best point of each node, first maximum wins (:751-765)
  const ONode* nodes = s_nodes(cur);
  uint32_t* out = kept + (size_t)b * kp_cap + kept_off[l];
  for (int i = tid; i < nn; i += kOctThreads) {
    const ONode nd = nodes[i];
    uint32_t best = perm(pc)[nd.kb];
    for (int pb_ = nd.kb + 1; pb_ < nd.ke; pb_ += 8) {  // eight loads in flight
      uint32_t t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = pb_ + u < nd.ke ? perm(pc)[pb_ + u] : 0u;
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (pb_ + u < nd.ke && (t[u] >> 24) > (best >> 24)) best = t[u];
    }
    out[i] = best;
  }
  OCT_T(5)
  OCT_T_END
  if (tid == 0) kept_cnt[(size_t)b * nlevels + l] = nn;
}

// k_kp_finalize: per frame, level-major concatenation of the kept keypoints and the monoIndex / stereoIndex slot
// assignment of operator() (reference src/ORBextractor.cc:1176-1221).
__global__ __launch_bounds__(256) void k_kp_finalize(const LevelDev* __restrict__ levels, int nlevels,
                                                     const uint32_t* __restrict__ kept, const int* __restrict__ kept_cnt,
                                                     const int* __restrict__ kept_off, int kp_cap, int lap0, int lap1,
                                                     KpIn* __restrict__ kpin, int* __restrict__ kp_count,
                                                     int* __restrict__ mono_out) {
  __shared__ unsigned long long s_w64[4];
  __shared__ int s_base[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const uint32_t* kp = kept + (size_t)b * kp_cap;
  const int* cnt = kept_cnt + (size_t)b * nlevels;
  int total = 0;
  for (int l = 0; l < nlevels; l++) total += cnt[l];
  KpIn* out = kpin + (size_t)b * kp_cap;
  if (tid == 0) {
    s_base[0] = 0;
    s_base[1] = 0;
  }
  __syncthreads();
  int rec = 0;  // running record index (level-major, list order)
  for (int l = 0; l < nlevels; l++) {
    const float scale = levels[l].scale;
    const int nl_ = cnt[l];
    for (int i0 = 0; i0 < nl_; i0 += 256) {
      const int i = i0 + tid;
      const bool valid = i < nl_;
      KpIn k;
      bool stereo = false;
      if (valid) {
        const uint32_t v = kp[kept_off[l] + i];
        k.x = (float)((int)(v & 0xfff) + 16);
        k.y = (float)((int)((v >> 12) & 0xfff) + 16);
        k.level = l;
        k.response = (float)(v >> 24);
        const float sx = l ? __fmul_rn(k.x, scale) : k.x;
        stereo = sx >= (float)lap0 && sx <= (float)lap1;
      }
      unsigned long long tot;
      const unsigned long long pre = oct_scan_u64(valid ? (stereo ? (1ull << 32) : 1ull) : 0ull, s_w64, &tot);
      if (valid) {
        const int mono_rank = s_base[0] + (int)(pre & 0xffffffffull), st_rank = s_base[1] + (int)(pre >> 32);
        k.slot = stereo ? (total - 1 - st_rank) : mono_rank;
        out[rec + i] = k;
      }
      __syncthreads();
      if (tid == 0) {
        s_base[0] += (int)(tot & 0xffffffffull);
        s_base[1] += (int)(tot >> 32);
      }
      __syncthreads();
    }
    rec += nl_;
  }
  if (tid == 0) {
    kp_count[b] = total;
    mono_out[b] = s_base[0];
  }
}

// ------------------------------------------------------------------------------------------------
// k_blur7: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) fixed-point path on the un-padded level
// (reference src/ORBextractor.cc:1188-1189; OpenCV smooth.simd.hpp): Q8.8 taps, u16 horizontal sums,
// u32 vertical sums, (v + 32768) >> 16.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * (n - 1) - p;
  return p;
}

// 64 x 32 output tile per workgroup; the (32+6) x 72-byte source tile (4-byte left pad keeps rows word aligned) is
// fetched with 32-bit loads, the horizontal pass produces 4 pixels per thread from three LDS words, the vertical pass
// 4 pixels per thread from seven 8-byte LDS reads and stores one 32-bit word.
constexpr int kBlurTW = 64, kBlurTH = 32;
__global__ __launch_bounds__(256) void k_blur7(const LevelDev* __restrict__ levels,
                                               const BlurTileDev* __restrict__ tiles, Lvl0 l0,
                                               const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                               uint8_t* __restrict__ blur, size_t blur_frame, int t0, int t1, int t2,
                                               int t3) {
  __shared__ __align__(16) uint32_t raw[kBlurTH + 6][18];        // 72 bytes per row: [x0-4, x0+68)
  __shared__ __align__(16) uint16_t hb[kBlurTH + 6][kBlurTW];
  const BlurTileDev T = tiles[blockIdx.x];  // (carries the level's geometry: no second dependent load of levels[T.level])
  const int b = blockIdx.y;
  struct {
    int rows, cols, pitch;
    unsigned blur_off;
  } L = {T.rows, T.cols, T.pitch, T.blur_off};
  const int sp = T.level == 0 ? l0.pitch : T.pitch;
  const uint8_t* src = T.level == 0 ? l0.base + (size_t)b * l0.frame_stride : pyr + (size_t)b * pyr_frame + T.plane_off;
  const int x0 = T.tx * kBlurTW, y0 = T.ty * kBlurTH;
  const int tid = threadIdx.x;
  const bool word_ok = ((sp & 3) == 0) && ((reinterpret_cast<size_t>(src) & 3) == 0);
  // all of a thread's words are asked for before the first one is stored (a load and its store per loop iteration made the
  // staging three round trips to memory one after the other: 6.5 k of a workgroup's 10 k cycles, clock64 in round 4)
  constexpr int kWords = (kBlurTH + 6) * 18, kIt = (kWords + 255) / 256;
  uint32_t v[kIt];
#pragma unroll
  for (int it = 0; it < kIt; it++) {
    const int i = tid + it * 256;
    v[it] = 0;
    if (i < kWords) {
      const int r = i / 18, w = i - r * 18;
      const int sy = reflect101(y0 + r - 3, L.rows);
      const int xb = x0 - 4 + 4 * w;  // first source column of this word
      if (word_ok && xb >= 0 && xb + 3 < L.cols) {
        v[it] = *reinterpret_cast<const uint32_t*>(src + (size_t)sy * sp + xb);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int sx = reflect101(xb + k, L.cols);
          const int sxc = min(max(sx, 0), L.cols - 1);  // columns far right of the image are never used
          v[it] |= (uint32_t)src[(size_t)sy * sp + sxc] << (8 * k);
        }
      }
    }
  }
#pragma unroll
  for (int it = 0; it < kIt; it++) {
    const int i = tid + it * 256;
    if (i < kWords) (&raw[0][0])[i] = v[it];
  }
  __syncthreads();
  for (int i = tid; i < (kBlurTH + 6) * (kBlurTW / 4); i += 256) {
    const int r = i >> 4, g = i & 15;  // outputs 4g..4g+3 need source bytes 4g+1 .. 4g+10 of the padded row
    const uint32_t w0 = raw[r][g], w1 = raw[r][g + 1], w2 = raw[r][g + 2];
    // Two outputs at a time in packed 16-bit lanes: (byte i, byte i + 1) of the 12 source bytes, zero-extended, is one v_perm; the
    // taps sum to 256 / 257, so t0 (a + b) + t1 (c + d) + t2 (e + f) + t3 g <= 65 535 never wraps (what OpenCV's saturating u16
    // arithmetic computes).  28 instructions for four outputs instead of 12 byte extractions + 32.
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto pair = [&](int i) -> us2 {  // (by[i], by[i + 1]), i <= 10 (compile-time after unrolling)
      const uint32_t hi = i + 1 <= 7 ? w1 : w2, lo = i + 1 <= 7 ? w0 : w1;
      const uint32_t j = i + 1 <= 7 ? i : i - 4;
      const uint32_t v = __builtin_amdgcn_perm(hi, lo, j | (0x0cu << 8) | ((j + 1) << 16) | (0x0cu << 24));
      return __builtin_bit_cast(us2, v);
    };
    const us2 T0 = {(unsigned short)t0, (unsigned short)t0}, T1 = {(unsigned short)t1, (unsigned short)t1},
              T2 = {(unsigned short)t2, (unsigned short)t2}, T3 = {(unsigned short)t3, (unsigned short)t3};
    uint32_t o2[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {  // outputs (0, 1) then (2, 3): taps at bytes 1 + 2 h2 + j
      const int b0 = 1 + 2 * h2;
      const us2 v = (pair(b0) + pair(b0 + 6)) * T0 + (pair(b0 + 1) + pair(b0 + 5)) * T1 + (pair(b0 + 2) + pair(b0 + 4)) * T2 + pair(b0 + 3) * T3;
      o2[h2] = __builtin_bit_cast(uint32_t, v);
    }
    *reinterpret_cast<uint2*>(&hb[r][4 * g]) = make_uint2(o2[0], o2[1]);
  }
  __syncthreads();
  uint8_t* dst = blur + (size_t)b * blur_frame + L.blur_off;
  for (int i = tid; i < kBlurTH * (kBlurTW / 4); i += 256) {
    const int r = i >> 4, g = i & 15;
    const int y = y0 + r, x = x0 + 4 * g;
    if (y >= L.rows || x >= L.cols) continue;
    unsigned acc[4] = {0, 0, 0, 0};
    const int taps[7] = {t0, t1, t2, t3, t2, t1, t0};
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const uint2 h = *reinterpret_cast<const uint2*>(&hb[r + j][4 * g]);
      typedef unsigned short us2 __attribute__((ext_vector_type(2)));
      const us2 hx = __builtin_bit_cast(us2, h.x), hy = __builtin_bit_cast(us2, h.y);  // (16 x 16 -> 32 bit multiply-adds on the halves)
      const unsigned short tj = (unsigned short)taps[j];
      acc[0] += (unsigned)hx.x * (unsigned)tj;
      acc[1] += (unsigned)hx.y * (unsigned)tj;
      acc[2] += (unsigned)hy.x * (unsigned)tj;
      acc[3] += (unsigned)hy.y * (unsigned)tj;
    }
    uint32_t pk = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) pk |= min((acc[k] + 32768u) >> 16, 255u) << (8 * k);
    uint8_t* d = dst + (size_t)y * L.pitch + x;
    if (x + 3 < L.cols) {
      *reinterpret_cast<uint32_t*>(d) = pk;  // pitch and blur_off are multiples of 64: aligned
    } else {
      for (int k = 0; k < 4 && x + k < L.cols; k++) d[k] = (uint8_t)(pk >> (8 * k));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_orient_brief: one wavefront per keypoint.
//   IC_Angle (src/ORBextractor.cc:71-95): int32 moments over the 749-pixel disc of the UNBLURRED level,
//     12 pixels per lane, exact wave reduction; cv::fastAtan2 polynomial in float32 (no FMA).
//   computeOrbDescriptor (:99-160): a = cosf(angle*pi/180), b = sinf(...), 512 steered samples of the
//     BLURRED level with std::round (half away from zero); 4 tests per lane, two lanes per output byte.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {  // OpenCV mathfuncs_core.simd.hpp atan_f32
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  const float eps = (float)2.2204460492503131e-16;
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = __fdiv_rn(ay, __fadd_rn(ax, eps));
    c2 = __fmul_rn(c, c);
    a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
  } else {
    c = __fdiv_rn(ax, __fadd_rn(ay, eps));
    c2 = __fmul_rn(c, c);
    a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
  }
  if (x < 0) a = __fsub_rn(180.f, a);
  if (y < 0) a = __fsub_rn(360.f, a);
  return a;
}

// glibc >= 2.28 sinf/cosf (sysdeps/ieee754/flt-32/s_sincosf.h), valid for |y| < 120; see oracle/check_sincosf.c.
__device__ __forceinline__ float sincos_poly(double x, double x2, int n, bool flip) {
  const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
               C4 = 0x1.99343027bf8c3p-16;
  const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
  if ((n & 1) == 0) {
    const double x3 = __dmul_rn(x, x2);
    const double s1 = __fma_rn(x2, S3, S2);
    const double x7 = __dmul_rn(x3, x2);
    const double s = __fma_rn(x3, S1, x);
    return (float)__fma_rn(x7, s1, s);
  }
  const double sg = flip ? -1.0 : 1.0;
  const double x4 = __dmul_rn(x2, x2);
  const double c2 = __fma_rn(x2, sg * C4, sg * C3);
  const double c1 = __fma_rn(x2, sg * C1, sg * C0);
  const double x6 = __dmul_rn(x4, x2);
  const double c = __fma_rn(x4, sg * C2, c1);
  return (float)__fma_rn(x6, c2, c);
}
__device__ __forceinline__ unsigned abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ff; }
__device__ __forceinline__ void glibc_sincosf(float y, float* sn, float* cs) {
  double x = (double)y;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    if (abstop12(y) < abstop12(0x1p-12f)) {
      *sn = y;
      *cs = 1.0f;
      return;
    }
    const double x2 = __dmul_rn(x, x);
    *sn = sincos_poly(x, x2, 0, false);
    *cs = sincos_poly(x, x2, 1, false);
    return;
  }
  const double r = __dmul_rn(x, 0x1.45F306DC9C883p+23);
  const int n = ((int)r + 0x800000) >> 24;
  x = __fma_rn(-(double)n, 0x1.921FB54442D18p0, x);
  const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
  const bool flip = (n & 2) != 0;
  const double x2 = __dmul_rn(x, x);
  const double xs = __dmul_rn(x, sgn);
  *sn = sincos_poly(xs, x2, n, flip);
  *cs = sincos_poly(xs, x2, n ^ 1, flip);
}

__device__ __forceinline__ int round_half_away(float x) {  // std::round(float) -> int
  float r = truncf(x);
  const float diff = __fsub_rn(x, r);  // exact
  if (fabsf(diff) >= 0.5f) r = __fadd_rn(r, copysignf(1.f, x));
  return (int)r;
}

constexpr int kObGroup = 16;                    // lanes per key point
constexpr int kObPerWg = 256 / kObGroup;        // key points per workgroup
// The kernel was bound by the L1's line rate (~3 400 line requests per wave of four key points = the measured 0.77 ms per 512
// frames; its ~1 540 VALU instructions per wave account for 0.3 ms): sixteen lanes read sixteen different rows of a patch in
// one instruction.  Now both patches are read row by row:
//  * IC_Angle: the 31 rows of the disc as 8 unaligned words each (u = -16 .. 15), eight lanes a row; a word's four pixels
//    meet their weights (u + 16 inside the disc, else 0; and 1 / 0) in v_dot4_u32_u8: m10 = sum(W . I) - 16 sum(M . I),
//    m01 = sum(v (M . I)).  Integer sums: any order gives the reference's numbers.
//  * the 37 x 37 window of the blurred level the 256 steered tests read (pattern coordinates <= 13 in magnitude: rotated and
//    rounded <= 18; checked when the extractor is created) goes to LDS as 37 rows of 10 aligned words; the tests read bytes
//    there.
constexpr int kObHalf = 18, kObRows = 2 * kObHalf + 1, kObRowWords = 10;
__global__ __launch_bounds__(256) void k_orient_brief(const LevelDev* __restrict__ levels, Lvl0 l0,
                                                      const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                      const uint8_t* __restrict__ blur, size_t blur_frame,
                                                      const KpIn* __restrict__ kpin, const int* __restrict__ kp_count,
                                                      int kp_cap, const int* __restrict__ umax,
                                                      const int8_t* __restrict__ pattern, gfs_keypoint* __restrict__ kps,
                                                      uint8_t* __restrict__ desc, int xcd_frames, int nchunks) {
  // (chunk of key points, frame) of this workgroup.  xcd_frames > 0 (a batch of at least eight frames, 1-D grid): all workgroups
  // of a frame run on ONE XCD -- MI355X hands workgroup i to XCD i % 8 -- so that the frame's two pyramids are read into one L2
  // instead of into all eight (round 6: FETCH_SIZE 2.4 x what the key points' windows cover before)
  int wg_chunk = blockIdx.x, wg_frame = blockIdx.y;
  if (xcd_frames > 0) {
    const int id = blockIdx.x, slot = id >> 3;
    wg_frame = (slot / nchunks) * 8 + (id & 7);
    wg_chunk = slot % nchunks;
    if (wg_frame >= xcd_frames) return;
  }
  __shared__ uint2 s_w[31 * 8];  // (W, M) of word j of disc row r
  __shared__ uint32_t s_patch[kObPerWg][kObRows * kObRowWords];
  if (threadIdx.x < 31 * 8) {
    const int r = threadIdx.x >> 3, j = threadIdx.x & 7, v = r - 15, um = umax[v < 0 ? -v : v];
    unsigned W = 0, M = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int u = -16 + 4 * j + q;
      if (u >= -um && u <= um) {
        W |= (unsigned)(u + 16) << (8 * q);
        M |= 1u << (8 * q);
      }
    }
    s_w[threadIdx.x] = make_uint2(W, M);
  }
  __syncthreads();
  // Sixteen lanes per key point, four key points per wave: the per-key-point scalar work (fastAtan2, glibc's sincosf) is done for four
  // of them at once instead of on 64 lanes for one; the pixel sums and the 256 tests are the same work either way.
  const int sl = threadIdx.x & (kObGroup - 1);
  const int b = wg_frame, n_kp = kp_count[b];
  const int k_raw = wg_chunk * kObPerWg + (threadIdx.x / kObGroup);
  if ((int)(wg_chunk * kObPerWg + (threadIdx.x >> 6) * (64 / kObGroup)) >= n_kp) return;  // the wave's first key point: wave-uniform
  const bool live = k_raw < n_kp;
  const int k = live ? k_raw : n_kp - 1;  // a group past the end repeats the last key point and stores nothing
  const KpIn in = kpin[(size_t)b * kp_cap + k];
  const LevelDev L = levels[in.level];
  int sp;
  const uint8_t* src = level_ptr(L, in.level, b, l0, pyr, pyr_frame, &sp);
  const int cx = __float2int_rn(in.x), cy = __float2int_rn(in.y);  // cvRound(pt)
  // descriptor window on the blurred level: asked for first, so that its rows travel while the angle is computed
  const uint8_t* bl = blur + (size_t)b * blur_frame + L.blur_off;
  const int bx = round_half_away(in.x), by = round_half_away(in.y);
  const uint8_t* bc = bl + (size_t)by * L.pitch + bx;
  const uint8_t* wcorner = bc - (size_t)kObHalf * L.pitch - kObHalf;
  const int ax = (int)(reinterpret_cast<uintptr_t>(wcorner) & 3);  // (all rows alike: the planes' pitches are multiples of 64)
  uint32_t* mp = s_patch[threadIdx.x / kObGroup];
  {
    const uint8_t* wbase = wcorner - ax;
    constexpr int kIter = (kObRows * kObRowWords + kObGroup - 1) / kObGroup;
    uint32_t wv[kIter];
#pragma unroll
    for (int it = 0; it < kIter; it++) {
      const int idx = min(sl + it * kObGroup, kObRows * kObRowWords - 1);
      const int row = idx / kObRowWords, j = idx - row * kObRowWords;
      wv[it] = *reinterpret_cast<const uint32_t*>(wbase + (size_t)row * L.pitch + 4 * j);
    }
#pragma unroll
    for (int it = 0; it < kIter; it++) mp[min(sl + it * kObGroup, kObRows * kObRowWords - 1)] = wv[it];
  }
  int m10, m01;
  {
    const uint8_t* row0 = src + (size_t)(cy - 15) * sp + (cx - 16);
    const int j = sl & 7;
    unsigned sW = 0;
    int sM = 0;
    m01 = 0;
#pragma unroll
    for (int it = 0; it < 16; it++) {
      const int r = 2 * it + (sl >> 3);
      if (r < 31) {
        uint32_t w;
        __builtin_memcpy(&w, row0 + (size_t)r * sp + 4 * j, 4);
        const uint2 wm = s_w[r * 8 + j];
        const int dm = (int)__builtin_amdgcn_udot4(w, wm.y, 0u, false);
        sW = __builtin_amdgcn_udot4(w, wm.x, sW, false);
        sM += dm;
        m01 += (r - 15) * dm;
      }
    }
    m10 = (int)sW - 16 * sM;
  }
#pragma unroll
  for (int ofs = kObGroup / 2; ofs > 0; ofs >>= 1) {
    m10 += __shfl_xor(m10, ofs, kObGroup);
    m01 += __shfl_xor(m01, ofs, kObGroup);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float a, bb;
  glibc_sincosf(__fmul_rn(angle, factorPI), &bb, &a);  // a = cos, b = sin
  // descriptor: lane sl takes the tests 16 sl .. 16 sl + 15 = descriptor bytes 2 sl, 2 sl + 1
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the window was written by this key point's own sixteen lanes
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const uint8_t* win = reinterpret_cast<const uint8_t*>(mp) + kObHalf * (4 * kObRowWords) + kObHalf + ax;  // (0, 0) of the window
  unsigned bits = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int4 pw = *reinterpret_cast<const int4*>(pattern + 64 * sl + 16 * q);  // tests 16 sl + 4 q .. + 3
    const int words[4] = {pw.x, pw.y, pw.z, pw.w};
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const int wv = words[t];
      const float x0 = (float)(int8_t)(wv & 0xff), y0 = (float)(int8_t)((wv >> 8) & 0xff);
      const float x1 = (float)(int8_t)((wv >> 16) & 0xff), y1 = (float)(int8_t)((wv >> 24) & 0xff);
      const int r0 = round_half_away(__fadd_rn(__fmul_rn(x0, bb), __fmul_rn(y0, a)));
      const int c0 = round_half_away(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, bb)));
      const int r1 = round_half_away(__fadd_rn(__fmul_rn(x1, bb), __fmul_rn(y1, a)));
      const int c1 = round_half_away(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, bb)));
      const int v0 = win[r0 * (4 * kObRowWords) + c0], v1 = win[r1 * (4 * kObRowWords) + c1];
      bits |= (unsigned)(v0 < v1) << (4 * q + t);
    }
  }
  if (!live) return;
  const size_t o = (size_t)b * kp_cap + in.slot;
  *reinterpret_cast<unsigned short*>(desc + o * 32 + 2 * sl) = (unsigned short)bits;  // (a descriptor is 32 bytes: aligned)
  if (sl == 0) {
    gfs_keypoint kp;
    // keypoint->pt *= scale for level != 0 (src/ORBextractor.cc:1204-1207)
    kp.x = in.level ? __fmul_rn(in.x, L.scale) : in.x;
    kp.y = in.level ? __fmul_rn(in.y, L.scale) : in.y;
    kp.size = L.patch_size;
    kp.angle = angle;
    kp.response = in.response;
    kp.octave = in.level;
    kp.class_id = -1;
    kps[o] = kp;
  }
}

const int8_t kPattern[256 * 4] = {
#include "brief_pattern.inc"
};
const int kBlurTaps[2][4] = {{18, 34, 48, 56}, {18, 34, 49, 55}};

// persistent host worker pool for the quadtree tasks (threads are created once per handle, not per call)
class WorkerPool {
 public:
  explicit WorkerPool(int n) : n_(std::max(1, n)) {
    for (int t = 1; t < n_; t++) th_.emplace_back([this, t]() { loop(t); });
  }
  ~WorkerPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& x : th_) x.join();
  }
  int size() const { return n_; }
  // runs f(i, worker) for i in [0, n); the calling thread participates as worker 0
  void run(int n, const std::function<void(int, int)>& f) {
    if (n <= 0) return;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &f;
      total_ = n;
      next_.store(0);
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    work(0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this]() { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work(int t) {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= total_) break;
      (*fn_)(i, t);
    }
  }
  void loop(int t) {
    unsigned long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return gen_ != seen; });
        seen = gen_;
        if (quit_) return;
      }
      work(t);
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int, int)>* fn_ = nullptr;
  std::atomic<int> next_{0};
  int total_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
  bool quit_ = false;
};

}  // namespace

struct gfs_orb {
  gfs_orb_config cfg;
  gfs::OrbParams P;
  gfs::OrbGeometry G;
  int geom_rows = 0, geom_cols = 0;
  size_t cap_pyr = 0, cap_blur = 0, cap_slab = 0, cap_cells = 0;
  int cap_kp = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_copy = nullptr;
  std::mutex mu;
  int host_threads = 1;
  std::unique_ptr<WorkerPool> pool;
  // device
  gfs::DevBuf<uint8_t> d_stage, d_pyr, d_blur;
  gfs::DevBuf<LevelDev> d_levels;
  gfs::DevBuf<CellDev> d_cells;
  gfs::DevBuf<BlurTileDev> d_tiles;
  gfs::DevBuf<int> d_strip_rows, d_strip_rows_fine, d_xt_start, d_xt_n, d_yt_start, d_yt_n, d_cell_cnt, d_cand_off, d_kp_count, d_mono;
  gfs::DevBuf<float> d_xt_alpha, d_yt_alpha;
  gfs::DevBuf<uint32_t> d_slab, d_cand, d_perm0, d_perm1, d_kept;
  gfs::DevBuf<unsigned short> d_seg0, d_seg1;
  gfs::DevBuf<int> d_kept_cnt, d_kept_off;
  int oct_node_cap = kOctMaxNodes;
  bool cands_packed = false;  // d_cand / d_cand_off hold the last call's dense candidate list
  bool device_octree = true;   // DistributeOctTree on the GPU (k_octree); false = host quadtree (GFS_ORB_OCTREE=host)
  bool pyr_in_hbm = false;     // test knob GFS_ORB_PYR_HBM: take the large-image pyramid path on any image
  bool octree_supported = true;
  bool host_counts_valid = false, host_cands_valid = false;
  gfs::DevBuf<KpIn> d_kpin;
  gfs::DevBuf<gfs_keypoint> d_kps;
  gfs::DevBuf<uint8_t> d_desc;
  gfs::DevBuf<int8_t> d_pattern;
  gfs::DevBuf<int> d_umax;  // the IC_Angle disc: half-width of row |v|
  // pinned host
  gfs::PinBuf<int> h_cand_off, h_kp_count, h_mono;
  gfs::PinBuf<uint32_t> h_cand;
  gfs::PinBuf<KpIn> h_kpin;
  // last call
  int last_B = 0;
  Lvl0 last_l0{};
  std::vector<gfs::OctreeScratch> scratch;
  std::vector<std::vector<int>> kept;
};

namespace {

int ensure_geometry(gfs_orb* h, int rows, int cols) {
  if (h->geom_rows == rows && h->geom_cols == cols) return GFS_OK;
  GFS_REQUIRE(rows <= h->cfg.max_rows && cols <= h->cfg.max_cols, GFS_ERR_CAPACITY,
              "image %dx%d exceeds the handle's max %dx%d", cols, rows, h->cfg.max_cols, h->cfg.max_rows);
  gfs::OrbGeometry G;
  G.build(h->P, rows, cols);
  GFS_REQUIRE(G.supported, GFS_ERR_UNSUPPORTED, "ORB geometry for %dx%d unsupported: %s", cols, rows, G.why);
  GFS_REQUIRE(kFcWaves * G.fast_lds_wave <= 160 * 1024 && G.max_tile_w + 6 <= 256 && (G.max_tile_w + 6) * G.max_tile_h < 16384,
              GFS_ERR_UNSUPPORTED, "FAST cell tile of %dx%d too large for LDS", cols, rows);
  GFS_REQUIRE(G.pyr_bytes <= h->cap_pyr && G.blur_bytes <= h->cap_blur && G.slab_entries <= h->cap_slab &&
                  G.cells.size() <= h->cap_cells && G.kp_cap <= h->cap_kp,
              GFS_ERR_CAPACITY, "image %dx%d needs more workspace than the handle reserved", cols, rows);
  hipStream_t s = h->stream;
  // A batch launched on a caller-supplied stream (gfs_orb_extract_batch_device returns without waiting) may still be reading
  // the tables of the previous image size: wait for everything in flight on the device before they are rewritten (rare path).
  if (h->geom_rows != 0) GFS_HIP(hipDeviceSynchronize());
  GFS_HIP(hipMemcpyAsync(h->d_levels.p, G.levels.data(), G.levels.size() * sizeof(LevelDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_cells.p, G.cells.data(), G.cells.size() * sizeof(CellDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_tiles.p, G.blur_tiles.data(), G.blur_tiles.size() * sizeof(BlurTileDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_strip_rows.p, G.strip_rows.data(), G.strip_rows.size() * 4, hipMemcpyHostToDevice, s));
  if (G.pyr_strips_fine > 0)
    GFS_HIP(hipMemcpyAsync(h->d_strip_rows_fine.p, G.strip_rows_fine.data(), G.strip_rows_fine.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_start.p, G.xt_start.data(), G.xt_start.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_n.p, G.xt_n.data(), G.xt_n.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_alpha.p, G.xt_alpha.data(), G.xt_alpha.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_start.p, G.yt_start.data(), G.yt_start.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_n.p, G.yt_n.data(), G.yt_n.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_alpha.p, G.yt_alpha.data(), G.yt_alpha.size() * 4, hipMemcpyHostToDevice, s));
  std::vector<int> kept_off(G.levels.size());
  bool oct_ok = true;
  int oct_cap = 8;
  {
    int o = 0;
    for (size_t l = 0; l < G.levels.size(); l++) {
      kept_off[l] = o;
      o += G.levels[l].kp_cap;
      const LevelDev& L = G.levels[l];
      int nIni = (int)std::round((float)(L.max_bx - 16) / (float)(L.max_by - 16));
      if (nIni == 0) nIni = 1;
      if (nIni > 4 || L.quota + 4 * nIni + 8 > kOctMaxNodes) oct_ok = false;  // outside what k_octree holds in LDS
      oct_cap = std::max(oct_cap, (L.quota + 4 * nIni + 8 + 7) / 8 * 8);
    }
  }
  GFS_HIP(hipMemcpyAsync(h->d_kept_off.p, kept_off.data(), kept_off.size() * sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipStreamSynchronize(s));
  h->octree_supported = oct_ok;
  h->oct_node_cap = std::min(oct_cap, kOctMaxNodes);
  h->G = std::move(G);
  h->geom_rows = rows;
  h->geom_cols = cols;
  return GFS_OK;
}

// The whole pipeline for B frames whose level 0 is described by l0.
int run_batch(gfs_orb* h, Lvl0 l0, int B, int rows, int cols, int lap0, int lap1, hipStream_t s) {
  const gfs::OrbGeometry& G = h->G;
  const int nl = h->P.nlevels;
  const int n_cells = (int)G.cells.size();
  const size_t cap_pyr = h->cap_pyr, cap_blur = h->cap_blur, cap_slab = h->cap_slab;
  // 1. pyramid chain (level l depends on l-1): one launch, row strips with recomputed halos
  if (G.pyr_lds_a + G.pyr_lds_b > 0 && !h->pyr_in_hbm) {
    // a few frames: the finer cut, so that the launch has workgroups for the chip (a frame's strips are dependent chains of levels)
    const bool fine = G.pyr_strips_fine > 0 && B * G.pyr_strips < 128;
    const int S = fine ? G.pyr_strips_fine : G.pyr_strips;
    const size_t la = fine ? G.pyr_lds_a_fine : G.pyr_lds_a, lb = fine ? G.pyr_lds_b_fine : G.pyr_lds_b;
    const size_t lx = fine ? G.pyr_lds_x_fine : G.pyr_lds_x;
    GFS_LAUNCH("k_pyr_area_lds", k_pyr_area_lds, dim3(S, B), dim3(kPyrThreads), la + lb + lx, s, h->d_levels.p, nl, l0, h->d_pyr.p, cap_pyr,
               fine ? h->d_strip_rows_fine.p : h->d_strip_rows.p, (unsigned)la, lx ? (unsigned)(la + lb) : 0u, (int)G.xt_start.size(),
               h->d_xt_start.p, h->d_xt_n.p, h->d_xt_alpha.p,
               h->d_yt_start.p, h->d_yt_n.p, h->d_yt_alpha.p);
  } else {
    GFS_LAUNCH("k_pyr_area_hbm", k_pyr_area_hbm, dim3(G.pyr_strips, B), dim3(kPyrThreads), 0, s, h->d_levels.p, nl, l0, h->d_pyr.p,
               cap_pyr, h->d_strip_rows.p, h->d_xt_start.p, h->d_xt_n.p, h->d_xt_alpha.p, h->d_yt_start.p, h->d_yt_n.p,
               h->d_yt_alpha.p);
  }
  // 2. FAST cells of all levels, all frames in one launch
  // per wave (= per cell): tile + score map (rows padded to whole words) + one list of offsets (u16), see k_fast_cells
  const FastMap fm = fast_map(B, n_cells);
  GFS_LAUNCH("k_fast_cells", k_fast_cells, dim3(fm.grid), dim3(64 * kFcWaves), (size_t)kFcWaves * G.fast_lds_wave, s, h->d_cells.p, l0,
             h->d_pyr.p, cap_pyr, h->P.ini_th, h->P.min_th, n_cells, fm, (int)G.fast_lds_wave, cap_slab, h->d_slab.p, h->d_cell_cnt.p);
  const int* tp = kBlurTaps[h->P.blur_variant ? 1 : 0];
  if (h->device_octree && h->octree_supported) {
    // 3-6 (device): quadtree (it takes its level's candidates straight from the cell slabs), slot assignment, blur, orientation +
    // descriptors — no host round trip at all; the dense candidate list is only laid out when an introspection call asks for it
    GFS_LAUNCH("k_octree", k_octree, dim3(B * nl), dim3(kOctThreads), oct_lds_bytes(h->oct_node_cap), s, h->d_levels.p, nl, nullptr,
               nullptr, cap_slab, h->d_perm0.p, h->d_perm1.p, h->d_seg0.p, h->d_seg1.p, h->d_kept_off.p, h->cap_kp, h->d_kept.p,
               h->d_kept_cnt.p, h->oct_node_cap, h->d_cells.p, h->d_cell_cnt.p, n_cells, h->d_slab.p);
    h->cands_packed = false;
    GFS_LAUNCH("k_kp_finalize", k_kp_finalize, dim3(B), dim3(256), 0, s, h->d_levels.p, nl, h->d_kept.p, h->d_kept_cnt.p,
               h->d_kept_off.p, h->cap_kp, lap0, lap1, h->d_kpin.p, h->d_kp_count.p, h->d_mono.p);
    GFS_LAUNCH("k_blur7", k_blur7, dim3((unsigned)G.blur_tiles.size(), B), dim3(256), 0, s, h->d_levels.p, h->d_tiles.p, l0,
               h->d_pyr.p, cap_pyr, h->d_blur.p, cap_blur, tp[0], tp[1], tp[2], tp[3]);
    {
      static const bool ob_xcd = !(getenv("GFS_ORB_OB_XCD") && atoi(getenv("GFS_ORB_OB_XCD")) == 0);
      const int nch = gfs::div_up(h->cap_kp, kObPerWg);
      const bool by_xcd = ob_xcd && B >= 8;  // (fewer frames than XCDs: a frame's key points over the whole chip)
      GFS_LAUNCH("k_orient_brief", k_orient_brief, by_xcd ? dim3((unsigned)(gfs::div_up(B, 8) * 8 * nch)) : dim3(nch, B), dim3(256), 0, s,
                 h->d_levels.p, l0, h->d_pyr.p, cap_pyr, h->d_blur.p, cap_blur, h->d_kpin.p, h->d_kp_count.p, h->cap_kp, h->d_umax.p,
                 h->d_pattern.p, h->d_kps.p, h->d_desc.p, by_xcd ? B : 0, nch);
    }
    h->last_B = B;
    h->last_l0 = l0;
    h->host_counts_valid = false;
    h->host_cands_valid = false;
    return GFS_OK;
  }
  // host quadtree: the dense, ordered candidate list goes to the host
  GFS_LAUNCH("k_cand_pack", k_cand_pack, dim3(B), dim3(256), 0, s, h->d_levels.p, h->d_cells.p, nl, n_cells, cap_slab, h->d_slab.p,
             h->d_cell_cnt.p, cap_slab, h->d_cand.p, h->d_cand_off.p);
  GFS_HIP(hipMemcpyAsync(h->h_cand_off.p, h->d_cand_off.p, (size_t)B * (nl + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  int max_total = 0;
  for (int b = 0; b < B; b++) max_total = std::max(max_total, h->h_cand_off.p[(size_t)b * (nl + 1) + nl]);
  if (max_total > 0)
    GFS_HIP(hipMemcpy2DAsync(h->h_cand.p, cap_slab * 4, h->d_cand.p, cap_slab * 4, (size_t)max_total * 4, B,
                             hipMemcpyDeviceToHost, s));
  GFS_HIP(hipEventRecord(h->ev_copy, s));
  // 3. blur of every level runs on the GPU while the host distributes keypoints
  GFS_LAUNCH("k_blur7", k_blur7, dim3((unsigned)G.blur_tiles.size(), B), dim3(256), 0, s, h->d_levels.p, h->d_tiles.p, l0,
             h->d_pyr.p, cap_pyr, h->d_blur.p, cap_blur, tp[0], tp[1], tp[2], tp[3]);
  GFS_HIP(hipEventSynchronize(h->ev_copy));
  // 4. quadtree distribution on the host (round 1): B x nlevels independent problems
  const int ntask = B * nl;
  if ((int)h->kept.size() < ntask) h->kept.resize(ntask);
  if ((int)h->scratch.size() < h->host_threads) h->scratch.resize(h->host_threads);
  h->pool->run(ntask, [&](int t, int tid) {
    const int b = t / nl, l = t % nl;
    const int* off = h->h_cand_off.p + (size_t)b * (nl + 1);
    const LevelDev& L = G.levels[l];
    std::vector<int>& out = h->kept[t];
    out.clear();
    gfs::distribute_octree(h->h_cand.p + (size_t)b * cap_slab + off[l], off[l + 1] - off[l], 16, L.max_bx, 16, L.max_by,
                           L.quota, h->scratch[tid], out);
  });
  // 5. keypoint records + output slot order (src/ORBextractor.cc:1181-1221)
  int max_n = 0;
  for (int b = 0; b < B; b++) {
    int n = 0;
    for (int l = 0; l < nl; l++) n += (int)h->kept[(size_t)b * nl + l].size();
    GFS_REQUIRE(n <= h->cap_kp, GFS_ERR_CAPACITY, "internal: %d keypoints exceed capacity %d", n, h->cap_kp);
    h->h_kp_count.p[b] = n;
    max_n = std::max(max_n, n);
    int mono = 0, stereo = n - 1, i = 0;
    const int* off = h->h_cand_off.p + (size_t)b * (nl + 1);
    KpIn* rec = h->h_kpin.p + (size_t)b * h->cap_kp;
    for (int l = 0; l < nl; l++) {
      const uint32_t* c = h->h_cand.p + (size_t)b * cap_slab + off[l];
      const float scale = h->P.scale[l];
      for (int idx : h->kept[(size_t)b * nl + l]) {
        KpIn k;
        k.x = (float)(gfs::cand_x(c[idx]) + 16);
        k.y = (float)(gfs::cand_y(c[idx]) + 16);
        k.level = l;
        k.response = (float)gfs::cand_score(c[idx]);
        const float sx = l ? k.x * scale : k.x;
        if (sx >= lap0 && sx <= lap1)
          k.slot = stereo--;
        else
          k.slot = mono++;
        rec[i++] = k;
      }
    }
    h->h_mono.p[b] = mono;
  }
  GFS_HIP(hipMemcpyAsync(h->d_kp_count.p, h->h_kp_count.p, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_mono.p, h->h_mono.p, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
  if (max_n > 0) {
    GFS_HIP(hipMemcpy2DAsync(h->d_kpin.p, (size_t)h->cap_kp * sizeof(KpIn), h->h_kpin.p, (size_t)h->cap_kp * sizeof(KpIn),
                             (size_t)max_n * sizeof(KpIn), B, hipMemcpyHostToDevice, s));
    // 6. orientation + descriptors
    GFS_LAUNCH("k_orient_brief", k_orient_brief, dim3(gfs::div_up(max_n, kObPerWg), B), dim3(256), 0, s, h->d_levels.p, l0,
               h->d_pyr.p, cap_pyr, h->d_blur.p, cap_blur, h->d_kpin.p, h->d_kp_count.p, h->cap_kp, h->d_umax.p,
               h->d_pattern.p, h->d_kps.p, h->d_desc.p, 0, 0);
  }
  h->last_B = B;
  h->last_l0 = l0;
  h->host_counts_valid = true;
  h->host_cands_valid = true;
  h->cands_packed = true;
  return GFS_OK;
}

// bring the per-frame counts (and, for the introspection calls, the candidate lists) of the last call to the host
int sync_host_view(gfs_orb* h, bool want_candidates) {
  if (h->host_counts_valid && (!want_candidates || h->host_cands_valid)) return GFS_OK;
  const int B = h->last_B, nl = h->P.nlevels;
  GFS_HIP(hipDeviceSynchronize());
  GFS_HIP(hipMemcpy(h->h_kp_count.p, h->d_kp_count.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
  GFS_HIP(hipMemcpy(h->h_mono.p, h->d_mono.p, (size_t)B * sizeof(int), hipMemcpyDeviceToHost));
  h->host_counts_valid = true;
  if (want_candidates) {
    if (!h->cands_packed) {  // the device quadtree read the slabs directly: lay the dense list out now (the slabs still hold the call)
      const int n_cells = (int)h->G.cells.size();
      GFS_LAUNCH("k_cand_pack", k_cand_pack, dim3(B), dim3(256), 0, h->stream, h->d_levels.p, h->d_cells.p, nl, n_cells, h->cap_slab,
                 h->d_slab.p, h->d_cell_cnt.p, h->cap_slab, h->d_cand.p, h->d_cand_off.p);
      GFS_HIP(hipDeviceSynchronize());
      h->cands_packed = true;
    }
    GFS_HIP(hipMemcpy(h->h_cand_off.p, h->d_cand_off.p, (size_t)B * (nl + 1) * sizeof(int), hipMemcpyDeviceToHost));
    GFS_HIP(hipMemcpy(h->h_cand.p, h->d_cand.p, (size_t)B * h->cap_slab * sizeof(uint32_t), hipMemcpyDeviceToHost));
    h->host_cands_valid = true;
  }
  return GFS_OK;
}

}  // namespace

extern "C" {

void gfs_orb_default_config(gfs_orb_config* c) {
  if (!c) return;
  c->nfeatures = 1000;
  c->scale_factor = 1.2f;
  c->nlevels = 8;
  c->ini_th_fast = 20;
  c->min_th_fast = 7;
  c->max_rows = 480;
  c->max_cols = 640;
  c->max_batch = 1;
  c->device = 0;
  c->blur_taps_variant = 0;
}

int gfs_orb_create(const gfs_orb_config* cfg, gfs_orb** out) {
  GFS_REQUIRE(cfg && out, GFS_ERR_INVALID_ARG, "gfs_orb_create: NULL argument");
  GFS_REQUIRE(cfg->nfeatures > 0 && cfg->nlevels > 0 && cfg->nlevels <= 16 && cfg->scale_factor > 1.f &&
                  cfg->max_rows > 0 && cfg->max_cols > 0 && cfg->max_batch > 0,
              GFS_ERR_INVALID_ARG, "gfs_orb_create: invalid configuration");
  if (!gfs::device_ok(cfg->device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(cfg->device));
  std::unique_ptr<gfs_orb> h(new gfs_orb);
  h->cfg = *cfg;
  h->P.init(cfg->nfeatures, cfg->scale_factor, cfg->nlevels, cfg->ini_th_fast, cfg->min_th_fast, cfg->blur_taps_variant);
  gfs::OrbGeometry G;
  G.build(h->P, cfg->max_rows, cfg->max_cols);
  GFS_REQUIRE(G.supported, GFS_ERR_UNSUPPORTED, "ORB geometry for max size %dx%d unsupported: %s", cfg->max_cols,
              cfg->max_rows, G.why);
  GFS_REQUIRE(kFcWaves * G.fast_lds_wave <= 160 * 1024 && G.max_tile_w + 6 <= 256 && (G.max_tile_w + 6) * G.max_tile_h < 16384, GFS_ERR_UNSUPPORTED,
              "FAST cell tile too large for LDS");
  const size_t B = cfg->max_batch;
  h->cap_pyr = G.pyr_bytes + 4096;
  h->cap_blur = G.blur_bytes + 4096;
  h->cap_slab = G.slab_entries + 1024;
  h->cap_cells = G.cells.size() + 64;
  h->cap_kp = G.kp_cap + 8;
  const int nl = cfg->nlevels;
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  GFS_HIP(hipEventCreateWithFlags(&h->ev_copy, hipEventDisableTiming));
  int rc = 0;
  const size_t tab_x = (size_t)cfg->max_cols * nl + 64, tab_y = (size_t)cfg->max_rows * nl + 64;
#define A(x) if (!rc) rc = (x)
  A(h->d_stage.alloc(B * (size_t)cfg->max_rows * cfg->max_cols));
  A(h->d_pyr.alloc(B * h->cap_pyr));
  A(h->d_blur.alloc(B * h->cap_blur));
  A(h->d_levels.alloc(nl));
  A(h->d_cells.alloc(h->cap_cells));
  A(h->d_tiles.alloc(G.blur_tiles.size() + 64));
  A(h->d_strip_rows.alloc((size_t)2 * gfs::OrbGeometry::kPyrMaxStrips * 16));
  A(h->d_strip_rows_fine.alloc((size_t)2 * gfs::OrbGeometry::kPyrMaxStrips * 16));
  GFS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pyr_area_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)gfs::OrbGeometry::kPyrLdsBudget));
  GFS_HIP(hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oct_lds_bytes(kOctMaxNodes)));
  GFS_HIP(hipFuncSetAttribute((const void*)k_fast_cells, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  A(h->d_xt_start.alloc(tab_x));
  A(h->d_xt_n.alloc(tab_x));
  A(h->d_xt_alpha.alloc(tab_x * 4));
  A(h->d_yt_start.alloc(tab_y));
  A(h->d_yt_n.alloc(tab_y));
  A(h->d_yt_alpha.alloc(tab_y * 4));
  A(h->d_cell_cnt.alloc(B * h->cap_cells));
  A(h->d_cand_off.alloc(B * (nl + 1)));
  A(h->d_kp_count.alloc(B));
  A(h->d_mono.alloc(B));
  A(h->d_slab.alloc(B * h->cap_slab));
  A(h->d_cand.alloc(B * h->cap_slab));
  A(h->d_kpin.alloc(B * h->cap_kp));
  A(h->d_perm0.alloc(B * h->cap_slab + 64));  // + 64: k_octree's cached sweeps read a whole chunk unconditionally
  A(h->d_perm1.alloc(B * h->cap_slab + 64));  // + 64: k_octree's cached sweeps read a whole chunk unconditionally
  A(h->d_seg0.alloc(B * h->cap_slab + 64));  // + 64: k_octree's cached sweeps read a whole chunk unconditionally
  A(h->d_seg1.alloc(B * h->cap_slab + 64));  // + 64: k_octree's cached sweeps read a whole chunk unconditionally
  A(h->d_kept.alloc(B * h->cap_kp));
  A(h->d_kept_cnt.alloc(B * nl));
  A(h->d_kept_off.alloc(nl));
  A(h->d_kps.alloc(B * h->cap_kp));
  A(h->d_desc.alloc(B * h->cap_kp * 32));
  A(h->h_cand_off.alloc(B * (nl + 1)));
  A(h->h_kp_count.alloc(B));
  A(h->h_mono.alloc(B));
  A(h->h_cand.alloc(B * h->cap_slab));
  A(h->h_kpin.alloc(B * h->cap_kp));
  A(h->d_umax.alloc(16));
  A(h->d_pattern.alloc(1024));
#undef A
  if (rc) return rc;
  // k_orient_brief keeps the window the steered tests can reach in LDS: (x, y) rotated and rounded stays within hypot(x, y) + 0.5
  for (int i = 0; i < 512; i++) {
    const double x = (double)(int8_t)kPattern[2 * i], y = (double)(int8_t)kPattern[2 * i + 1];
    GFS_REQUIRE(std::sqrt(x * x + y * y) + 0.5 < (double)(kObHalf + 1), GFS_ERR_UNSUPPORTED,
                "a BRIEF test point lies outside the %d x %d window of k_orient_brief", kObRows, kObRows);
  }
  GFS_HIP(hipMemcpy(h->d_umax.p, h->P.umax, 16 * sizeof(int), hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(h->d_pattern.p, kPattern, 1024, hipMemcpyHostToDevice));
  GFS_HIP(hipMemset(h->d_kp_count.p, 0, B * sizeof(int)));
  h->host_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("GFS_ORB_HOST_THREADS")) h->host_threads = std::max(1, atoi(e));
  h->pool.reset(new WorkerPool(h->host_threads));
  if (const char* e = getenv("GFS_ORB_OCTREE")) h->device_octree = strcmp(e, "host") != 0;
  h->pyr_in_hbm = getenv("GFS_ORB_PYR_HBM") != nullptr;
  *out = h.release();
  return GFS_OK;
}

void gfs_orb_destroy(gfs_orb* h) {
  if (!h) return;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->ev_copy) (void)hipEventDestroy(h->ev_copy);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_orb_get_tables(const gfs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                       int32_t* feats, int32_t* umax16) {
  GFS_REQUIRE(h, GFS_ERR_INVALID_ARG, "gfs_orb_get_tables: NULL handle");
  for (int i = 0; i < h->P.nlevels; i++) {
    if (scale) scale[i] = h->P.scale[i];
    if (inv_scale) inv_scale[i] = h->P.inv_scale[i];
    if (sigma2) sigma2[i] = h->P.sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = h->P.inv_sigma2[i];
    if (feats) feats[i] = h->P.quota[i];
  }
  if (umax16)
    for (int i = 0; i < 16; i++) umax16[i] = h->P.umax[i];
  return GFS_OK;
}

int gfs_orb_max_keypoints(const gfs_orb* h) { return h ? h->cap_kp : GFS_ERR_INVALID_ARG; }

int gfs_orb_extract_batch_device(gfs_orb* h, const void* dev_imgs, int B, int rows, int cols, int lap0, int lap1,
                                 void* stream) {
  GFS_REQUIRE(h && dev_imgs && B > 0 && rows > 0 && cols > 0, GFS_ERR_INVALID_ARG,
              "gfs_orb_extract_batch_device: invalid argument");
  GFS_REQUIRE(B <= h->cfg.max_batch, GFS_ERR_CAPACITY, "batch %d exceeds handle max_batch %d", B, h->cfg.max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->cfg.device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc) return rc;
  Lvl0 l0{(const uint8_t*)dev_imgs, (size_t)rows * cols, cols};
  return run_batch(h, l0, B, rows, cols, lap0, lap1, stream ? (hipStream_t)stream : h->stream);
}

int gfs_orb_extract_batch(gfs_orb* h, const uint8_t* const* imgs, int B, int rows, int cols, int stride, int lap0,
                          int lap1, gfs_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index) {
  GFS_REQUIRE(h && imgs && B > 0 && rows > 0 && cols > 0 && stride >= cols && n, GFS_ERR_INVALID_ARG,
              "gfs_orb_extract_batch: invalid argument");
  GFS_REQUIRE(B <= h->cfg.max_batch, GFS_ERR_CAPACITY, "batch %d exceeds handle max_batch %d", B, h->cfg.max_batch);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->cfg.device));
  int rc = ensure_geometry(h, rows, cols);
  if (rc) return rc;
  hipStream_t s = h->stream;
  for (int b = 0; b < B; b++)
    GFS_HIP(hipMemcpy2DAsync(h->d_stage.p + (size_t)b * rows * cols, cols, imgs[b], stride, cols, rows,
                             hipMemcpyHostToDevice, s));
  Lvl0 l0{h->d_stage.p, (size_t)rows * cols, cols};
  rc = run_batch(h, l0, B, rows, cols, lap0, lap1, s);
  if (rc) return rc;
  GFS_HIP(hipStreamSynchronize(s));
  if ((rc = sync_host_view(h, false))) return rc;
  for (int b = 0; b < B; b++) {
    const int nb = h->h_kp_count.p[b];
    n[b] = nb;
    if (mono_index) mono_index[b] = h->h_mono.p[b];
    if (nb > cap) {
      gfs::set_error("gfs_orb_extract_batch: %d keypoints exceed caller capacity %d", nb, cap);
      return GFS_ERR_CAPACITY;
    }
    if (nb && kps)
      GFS_HIP(hipMemcpyAsync(kps + (size_t)b * cap, h->d_kps.p + (size_t)b * h->cap_kp, (size_t)nb * sizeof(gfs_keypoint),
                             hipMemcpyDeviceToHost, s));
    if (nb && desc)
      GFS_HIP(hipMemcpyAsync(desc + (size_t)b * cap * 32, h->d_desc.p + (size_t)b * h->cap_kp * 32, (size_t)nb * 32,
                             hipMemcpyDeviceToHost, s));
  }
  GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

int gfs_orb_extract(gfs_orb* h, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                    gfs_keypoint* kps, uint8_t* desc, int cap, int* n) {
  if (n) *n = 0;
  if (!img || rows <= 0 || cols <= 0) return -1;  // _image.empty() -> -1 (src/ORBextractor.cc:1150)
  int nn = 0, mono = 0;
  const uint8_t* imgs[1] = {img};
  int rc = gfs_orb_extract_batch(h, imgs, 1, rows, cols, stride, lap0, lap1, kps, desc, cap, &nn, &mono);
  if (n) *n = nn;
  if (rc) return rc - 100;
  return mono;
}

int gfs_orb_device_results(gfs_orb* h, void** dev_kps, void** dev_desc, void** dev_counts, void** dev_mono, int* cap) {
  GFS_REQUIRE(h, GFS_ERR_INVALID_ARG, "gfs_orb_device_results: NULL handle");
  if (dev_kps) *dev_kps = h->d_kps.p;
  if (dev_desc) *dev_desc = h->d_desc.p;
  if (dev_counts) *dev_counts = h->d_kp_count.p;
  if (dev_mono) *dev_mono = h->d_mono.p;
  if (cap) *cap = h->cap_kp;
  return GFS_OK;
}

int gfs_orb_fetch(gfs_orb* h, int b, gfs_keypoint* kps, uint8_t* desc, int cap, int* n, int* mono_index) {
  GFS_REQUIRE(h && b >= 0 && b < h->last_B, GFS_ERR_INVALID_ARG, "gfs_orb_fetch: invalid frame index");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->cfg.device));
  GFS_HIP(hipDeviceSynchronize());
  {
    const int rc0 = sync_host_view(h, false);
    if (rc0) return rc0;
  }
  const int nb = h->h_kp_count.p[b];
  if (n) *n = nb;
  if (mono_index) *mono_index = h->h_mono.p[b];
  GFS_REQUIRE(nb <= cap, GFS_ERR_CAPACITY, "gfs_orb_fetch: %d keypoints exceed caller capacity %d", nb, cap);
  if (nb && kps) GFS_HIP(hipMemcpy(kps, h->d_kps.p + (size_t)b * h->cap_kp, (size_t)nb * sizeof(gfs_keypoint), hipMemcpyDeviceToHost));
  if (nb && desc) GFS_HIP(hipMemcpy(desc, h->d_desc.p + (size_t)b * h->cap_kp * 32, (size_t)nb * 32, hipMemcpyDeviceToHost));
  return GFS_OK;
}

int gfs_orb_level_size(const gfs_orb* h, int level, int* rows, int* cols) {
  GFS_REQUIRE(h && level >= 0 && level < h->P.nlevels && h->geom_rows > 0, GFS_ERR_INVALID_ARG,
              "gfs_orb_level_size: invalid level or no image processed yet");
  *rows = h->G.levels[level].rows;
  *cols = h->G.levels[level].cols;
  return GFS_OK;
}

int gfs_orb_fetch_level(gfs_orb* h, int b, int level, int blurred, uint8_t* dst) {
  GFS_REQUIRE(h && dst && b >= 0 && b < h->last_B && level >= 0 && level < h->P.nlevels, GFS_ERR_INVALID_ARG,
              "gfs_orb_fetch_level: invalid argument");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->cfg.device));
  GFS_HIP(hipDeviceSynchronize());
  const LevelDev& L = h->G.levels[level];
  const uint8_t* src;
  size_t pitch;
  if (blurred) {
    src = h->d_blur.p + (size_t)b * h->cap_blur + L.blur_off;
    pitch = L.pitch;
  } else if (level == 0) {
    src = h->last_l0.base + (size_t)b * h->last_l0.frame_stride;
    pitch = h->last_l0.pitch;
  } else {
    src = h->d_pyr.p + (size_t)b * h->cap_pyr + L.plane_off;
    pitch = L.pitch;
  }
  GFS_HIP(hipMemcpy2D(dst, L.cols, src, pitch, L.cols, L.rows, hipMemcpyDeviceToHost));
  return GFS_OK;
}

int gfs_orb_fetch_candidates(gfs_orb* h, int b, int level, int32_t* x, int32_t* y, int32_t* score, int cap) {
  GFS_REQUIRE(h && b >= 0 && b < h->last_B && level >= 0 && level < h->P.nlevels, GFS_ERR_INVALID_ARG,
              "gfs_orb_fetch_candidates: invalid argument");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->cfg.device));
  {
    const int rc0 = sync_host_view(h, true);
    if (rc0) return rc0;
  }
  const int nl = h->P.nlevels;
  const int* off = h->h_cand_off.p + (size_t)b * (nl + 1);
  const int cnt = off[level + 1] - off[level];
  const uint32_t* c = h->h_cand.p + (size_t)b * h->cap_slab + off[level];
  for (int i = 0; i < cnt && i < cap; i++) {
    if (x) x[i] = gfs::cand_x(c[i]);
    if (y) y[i] = gfs::cand_y(c[i]);
    if (score) score[i] = gfs::cand_score(c[i]);
  }
  return cnt;
}

// GPU test hook: the device DistributeOctTree (k_octree) on caller candidates; out = kept candidates as packed values
// (x | y << 12 | score << 24) in list order.  Returns the number kept or a negative gfs_status.
int gfs_orb_octree_device(int device, const int32_t* x, const int32_t* y, const int32_t* score, int n, int min_x, int max_x,
                          int min_y, int max_y, int n_features, int32_t* out_x, int32_t* out_y, int32_t* out_score, int cap) {
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  GFS_REQUIRE(min_x == 16 && min_y == 16 && n >= 0, GFS_ERR_INVALID_ARG, "gfs_orb_octree_device: min border must be 16");
  LevelDev L{};
  L.max_bx = max_x;
  L.max_by = max_y;
  L.quota = n_features;
  int nIni = (int)std::round((float)(max_x - 16) / (float)(max_y - 16));
  if (nIni == 0) nIni = 1;
  GFS_REQUIRE(nIni <= 4 && n_features + 4 * nIni + 8 <= kOctMaxNodes, GFS_ERR_UNSUPPORTED, "outside k_octree limits");
  const int kcap = n_features + 3 + 4 * nIni + 8;
  std::vector<uint32_t> c(std::max(n, 1));
  for (int i = 0; i < n; i++) c[i] = (uint32_t)x[i] | ((uint32_t)y[i] << 12) | ((uint32_t)score[i] << 24);
  gfs::DevBuf<LevelDev> dL;
  gfs::DevBuf<uint32_t> dc, p0, p1, dk;
  gfs::DevBuf<unsigned short> s0, s1;
  gfs::DevBuf<int> doff, dko, dcnt;
  int rc = 0;
  if ((rc = dL.alloc(1)) || (rc = dc.alloc(c.size())) || (rc = p0.alloc(c.size() + 64)) || (rc = p1.alloc(c.size() + 64)) ||
      (rc = s0.alloc(c.size() + 64)) || (rc = s1.alloc(c.size() + 64)) || (rc = dk.alloc(kcap)) || (rc = doff.alloc(2)) ||
      (rc = dko.alloc(1)) || (rc = dcnt.alloc(1)))
    return rc;
  const int off[2] = {0, n}, ko = 0;
  GFS_HIP(hipMemcpy(dL.p, &L, sizeof(L), hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(dc.p, c.data(), c.size() * 4, hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(doff.p, off, sizeof(off), hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(dko.p, &ko, sizeof(ko), hipMemcpyHostToDevice));
  const int node_cap = (n_features + 4 * nIni + 8 + 7) / 8 * 8;
  GFS_HIP(hipFuncSetAttribute((const void*)k_octree, hipFuncAttributeMaxDynamicSharedMemorySize, (int)oct_lds_bytes(kOctMaxNodes)));
  hipLaunchKernelGGL(k_octree, dim3(1), dim3(kOctThreads), oct_lds_bytes(node_cap), 0, dL.p, 1, dc.p, doff.p, c.size(), p0.p, p1.p, s0.p, s1.p,
                     dko.p, kcap, dk.p, dcnt.p, node_cap, (const CellDev*)nullptr, (const int*)nullptr, 0, (const uint32_t*)nullptr);
  GFS_HIP(hipDeviceSynchronize());
  int cnt = 0;
  GFS_HIP(hipMemcpy(&cnt, dcnt.p, sizeof(int), hipMemcpyDeviceToHost));
  std::vector<uint32_t> k(std::max(cnt, 1));
  if (cnt) GFS_HIP(hipMemcpy(k.data(), dk.p, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < cnt && i < cap; i++) {
    out_x[i] = gfs::cand_x(k[i]);
    out_y[i] = gfs::cand_y(k[i]);
    out_score[i] = gfs::cand_score(k[i]);
  }
  return cnt;
}

// Host-logic hook for the CPU test-suite: the product's index-based DistributeOctTree on caller candidates.
int gfs_orb_octree_host(const int32_t* x, const int32_t* y, const int32_t* score, int n, int min_x, int max_x,
                        int min_y, int max_y, int n_features, int32_t* out_idx, int cap) {
  std::vector<uint32_t> c(n);
  for (int i = 0; i < n; i++) c[i] = (uint32_t)x[i] | ((uint32_t)y[i] << 12) | ((uint32_t)score[i] << 24);
  gfs::OctreeScratch S;
  std::vector<int> out;
  gfs::distribute_octree(c.data(), n, min_x, max_x, min_y, max_y, n_features, S, out);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) out_idx[i] = out[i];
  return (int)out.size();
}

}  // extern "C"
