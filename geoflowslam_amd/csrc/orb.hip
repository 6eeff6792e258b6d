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
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

#include "gfs_common.hpp"
#include "orb_host.hpp"

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
__global__ __launch_bounds__(256) void k_pyr_area(const LevelDev* __restrict__ levels, int level, Lvl0 l0,
                                                  uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                  const int* __restrict__ xt_start, const int* __restrict__ xt_n,
                                                  const float* __restrict__ xt_alpha, const int* __restrict__ yt_start,
                                                  const int* __restrict__ yt_n, const float* __restrict__ yt_alpha) {
  const LevelDev L = levels[level];
  const LevelDev S = levels[level - 1];
  const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
  if (dx >= L.cols || dy >= L.rows) return;
  int sp;
  const uint8_t* src = level_ptr(S, level - 1, b, l0, pyr, pyr_frame, &sp);
  const int xi = L.xtab_off + dx, yi = L.ytab_off + dy;
  const int sx0 = xt_start[xi], nx = xt_n[xi], sy0 = yt_start[yi], ny = yt_n[yi];
  const float4 ax = *reinterpret_cast<const float4*>(xt_alpha + 4 * (size_t)xi);
  const float4 ay = *reinterpret_cast<const float4*>(yt_alpha + 4 * (size_t)yi);
  const float axs[4] = {ax.x, ax.y, ax.z, ax.w}, ays[4] = {ay.x, ay.y, ay.z, ay.w};
  float sum = 0.f;
  for (int j = 0; j < ny; j++) {
    const uint8_t* row = src + (size_t)(sy0 + j) * sp + sx0;
    float buf = 0.f;
    for (int k = 0; k < nx; k++) buf = __fadd_rn(buf, __fmul_rn((float)row[k], axs[k]));
    const float t = __fmul_rn(ays[j], buf);
    sum = (j == 0) ? t : __fadd_rn(sum, t);
  }
  int r = __float2int_rn(sum);  // cvRound: round-half-even
  r = min(max(r, 0), 255);
  pyr[(size_t)b * pyr_frame + L.plane_off + (size_t)dy * L.pitch + dx] = (uint8_t)r;
}

// ------------------------------------------------------------------------------------------------
// k_fast_cells: one workgroup per FAST cell (reference src/ORBextractor.cc:788-842 calling cv::FAST on the
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

__device__ __forceinline__ int arc_score(const uint8_t* __restrict__ c, int p, int th_low) {
  const int v = c[0];
  int d[16];
  d[0] = v - c[3 * p];
  d[1] = v - c[3 * p + 1];
  d[2] = v - c[2 * p + 2];
  d[3] = v - c[p + 3];
  d[4] = v - c[3];
  d[5] = v - c[-p + 3];
  d[6] = v - c[-2 * p + 2];
  d[7] = v - c[-3 * p + 1];
  d[8] = v - c[-3 * p];
  d[9] = v - c[-3 * p - 1];
  d[10] = v - c[-2 * p - 2];
  d[11] = v - c[-p - 3];
  d[12] = v - c[-3];
  d[13] = v - c[p - 3];
  d[14] = v - c[2 * p - 2];
  d[15] = v - c[3 * p - 1];
  unsigned md = 0, mb = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    md |= (unsigned)(d[k] > th_low) << k;
    mb |= (unsigned)(d[k] < -th_low) << k;
  }
  if ((has9(md) | has9(mb)) == 0) return 0;
  // sliding min / max over 9 cyclic neighbours by doubling
  int lo[16], hi[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo[k] = min(d[k], d[(k + 1) & 15]);
    hi[k] = max(d[k], d[(k + 1) & 15]);
  }
  int lo4[16], hi4[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    lo4[k] = min(lo[k], lo[(k + 2) & 15]);
    hi4[k] = max(hi[k], hi[(k + 2) & 15]);
  }
  int A = -256, Bm = 256;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const int l9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
    const int h9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
    A = max(A, l9);
    Bm = min(Bm, h9);
  }
  return max(A, -Bm);
}

__global__ __launch_bounds__(256) void k_fast_cells(const LevelDev* __restrict__ levels,
                                                    const CellDev* __restrict__ cells, Lvl0 l0,
                                                    const uint8_t* __restrict__ pyr, size_t pyr_frame, int ini_th,
                                                    int min_th, int n_cells, size_t slab_frame,
                                                    uint32_t* __restrict__ slab, int* __restrict__ cell_cnt) {
  extern __shared__ __align__(16) uint8_t smem[];
  __shared__ int s_count;
  __shared__ int s_scan[256];
  const int cell_id = blockIdx.x, b = blockIdx.y;
  const CellDev C = cells[cell_id];
  const LevelDev L = levels[C.level];
  const int w = C.w, h = C.h;
  const int tid = threadIdx.x;
  const int dw = w - 6, dh = h - 6;
  if (dw <= 0 || dh <= 0) {
    if (tid == 0) cell_cnt[(size_t)b * n_cells + cell_id] = 0;
    return;
  }
  uint8_t* tile = smem;             // [h][w]
  uint8_t* sc = smem + (size_t)w * h;  // [h][w] arc score - 1 (0 = not a corner at the low threshold)
  int sp;
  const uint8_t* src = level_ptr(L, C.level, b, l0, pyr, pyr_frame, &sp);
  src += (size_t)C.y0 * sp + C.x0;
  for (int i = tid; i < w * h; i += 256) {
    const int y = i / w, x = i - y * w;
    tile[i] = src[(size_t)y * sp + x];
    sc[i] = 0;
  }
  if (tid == 0) s_count = 0;
  __syncthreads();
  const int th_hi = min(max(ini_th, 0), 255), th_lo = min(max(min_th, 0), 255);
  const int th_low = min(th_hi, th_lo);
  for (int i = tid; i < dw * dh; i += 256) {
    const int yy = i / dw, xx = i - yy * dw;
    const int o = (yy + 3) * w + xx + 3;
    const int S = arc_score(tile + o, w, th_low);
    sc[o] = (uint8_t)(S > th_low ? S - 1 : 0);
  }
  __syncthreads();
  // each thread owns a contiguous raster chunk of the detection area so the emission order is raster
  const int total = dw * dh;
  const int chunk = (total + 255) / 256;
  const int i0 = min(tid * chunk, total), i1 = min(i0 + chunk, total);
  int T = max(th_hi, 1);
  int mine = 0;
  for (int pass = 0; pass < 2; pass++) {
    mine = 0;
    for (int i = i0; i < i1; i++) {
      const int yy = i / dw, xx = i - yy * dw;
      const int o = (yy + 3) * w + xx + 3;
      const int s = sc[o];
      if (s < T) continue;
#define NB(off) ((int)sc[o + (off)] >= T ? (int)sc[o + (off)] : 0)
      const bool keep = s > NB(-1) && s > NB(1) && s > NB(-w - 1) && s > NB(-w) && s > NB(-w + 1) && s > NB(w - 1) &&
                        s > NB(w) && s > NB(w + 1);
#undef NB
      mine += keep ? 1 : 0;
    }
    if (mine) atomicAdd(&s_count, mine);
    __syncthreads();
    const int cnt = s_count;
    __syncthreads();
    if (cnt > 0 || pass == 1) break;
    T = max(th_lo, 1);  // vKeysCell.empty() -> retry with minThFAST (src/ORBextractor.cc:825-827)
  }
  // exclusive scan of per-thread counts
  s_scan[tid] = mine;
  __syncthreads();
  for (int ofs = 1; ofs < 256; ofs <<= 1) {
    const int v = tid >= ofs ? s_scan[tid - ofs] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int pos = s_scan[tid] - mine;
  uint32_t* out = slab + (size_t)b * slab_frame + C.slab_off;
  const int offx = C.x0 - 16, offy = C.y0 - 16;  // + j*wCell, + i*hCell (src/ORBextractor.cc:847-848)
  for (int i = i0; i < i1 && mine > 0; i++) {
    const int yy = i / dw, xx = i - yy * dw;
    const int o = (yy + 3) * w + xx + 3;
    const int s = sc[o];
    if (s < T) continue;
#define NB(off) ((int)sc[o + (off)] >= T ? (int)sc[o + (off)] : 0)
    const bool keep = s > NB(-1) && s > NB(1) && s > NB(-w - 1) && s > NB(-w) && s > NB(-w + 1) && s > NB(w - 1) &&
                      s > NB(w) && s > NB(w + 1);
#undef NB
    if (keep) out[pos++] = (uint32_t)(xx + 3 + offx) | ((uint32_t)(yy + 3 + offy) << 12) | ((uint32_t)s << 24);
  }
  if (tid == 255) cell_cnt[(size_t)b * n_cells + cell_id] = s_scan[255];
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
// k_blur7: cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) fixed-point path on the un-padded level
// (reference src/ORBextractor.cc:1188-1189; OpenCV smooth.simd.hpp): Q8.8 taps, u16 horizontal sums,
// u32 vertical sums, (v + 32768) >> 16.  64x16 output tile per workgroup, raw + horizontal pass in LDS.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
  if (p < 0) p = -p;
  if (p >= n) p = 2 * (n - 1) - p;
  return p;
}

__global__ __launch_bounds__(256) void k_blur7(const LevelDev* __restrict__ levels,
                                               const BlurTileDev* __restrict__ tiles, Lvl0 l0,
                                               const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                               uint8_t* __restrict__ blur, size_t blur_frame, int t0, int t1, int t2,
                                               int t3) {
  __shared__ uint8_t raw[22][72];
  __shared__ uint16_t hb[22][64];
  const BlurTileDev T = tiles[blockIdx.x];
  const int b = blockIdx.y;
  const LevelDev L = levels[T.level];
  int sp;
  const uint8_t* src = level_ptr(L, T.level, b, l0, pyr, pyr_frame, &sp);
  const int x0 = T.tx * 64, y0 = T.ty * 16;
  const int tid = threadIdx.x;
  for (int i = tid; i < 22 * 70; i += 256) {
    const int r = i / 70, c = i - r * 70;
    const int sy = reflect101(y0 + r - 3, L.rows), sx = reflect101(x0 + c - 3, L.cols);
    raw[r][c] = src[(size_t)sy * sp + sx];
  }
  __syncthreads();
  for (int i = tid; i < 22 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    const unsigned v = t0 * (raw[r][c] + raw[r][c + 6]) + t1 * (raw[r][c + 1] + raw[r][c + 5]) +
                       t2 * (raw[r][c + 2] + raw[r][c + 4]) + t3 * raw[r][c + 3];
    hb[r][c] = (uint16_t)min(v, 0xffffu);
  }
  __syncthreads();
  uint8_t* dst = blur + (size_t)b * blur_frame + L.blur_off;
  for (int i = tid; i < 16 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    const int y = y0 + r, x = x0 + c;
    if (y >= L.rows || x >= L.cols) continue;
    const unsigned v = t0 * ((unsigned)hb[r][c] + hb[r + 6][c]) + t1 * ((unsigned)hb[r + 1][c] + hb[r + 5][c]) +
                       t2 * ((unsigned)hb[r + 2][c] + hb[r + 4][c]) + t3 * (unsigned)hb[r + 3][c];
    const unsigned o = (v + 32768u) >> 16;
    dst[(size_t)y * L.pitch + x] = (uint8_t)min(o, 255u);
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

__global__ __launch_bounds__(256) void k_orient_brief(const LevelDev* __restrict__ levels, Lvl0 l0,
                                                      const uint8_t* __restrict__ pyr, size_t pyr_frame,
                                                      const uint8_t* __restrict__ blur, size_t blur_frame,
                                                      const KpIn* __restrict__ kpin, const int* __restrict__ kp_count,
                                                      int kp_cap, const int8_t* __restrict__ ic_du,
                                                      const int8_t* __restrict__ ic_dv, int ic_n,
                                                      const int8_t* __restrict__ pattern, gfs_keypoint* __restrict__ kps,
                                                      uint8_t* __restrict__ desc) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (k >= kp_count[b]) return;
  const KpIn in = kpin[(size_t)b * kp_cap + k];
  const LevelDev L = levels[in.level];
  int sp;
  const uint8_t* src = level_ptr(L, in.level, b, l0, pyr, pyr_frame, &sp);
  const int cx = __float2int_rn(in.x), cy = __float2int_rn(in.y);  // cvRound(pt)
  const uint8_t* center = src + (size_t)cy * sp + cx;
  int m10 = 0, m01 = 0;
  for (int i = lane; i < ic_n; i += 64) {
    const int du = ic_du[i], dv = ic_dv[i];
    const int val = center[dv * sp + du];
    m10 += du * val;
    m01 += dv * val;
  }
#pragma unroll
  for (int ofs = 32; ofs > 0; ofs >>= 1) {
    m10 += __shfl_xor(m10, ofs, 64);
    m01 += __shfl_xor(m01, ofs, 64);
  }
  const float angle = fast_atan2_deg((float)m01, (float)m10);
  const float factorPI = (float)(3.14159265358979323846 / 180.f);
  float a, bb;
  glibc_sincosf(__fmul_rn(angle, factorPI), &bb, &a);  // a = cos, b = sin
  // descriptor on the blurred level
  const uint8_t* bl = blur + (size_t)b * blur_frame + L.blur_off;
  const int bx = round_half_away(in.x), by = round_half_away(in.y);
  const uint8_t* bc = bl + (size_t)by * L.pitch + bx;
  const int4 pw = *reinterpret_cast<const int4*>(pattern + 16 * lane);  // tests 4*lane .. 4*lane+3
  const int words[4] = {pw.x, pw.y, pw.z, pw.w};
  unsigned nib = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const int wv = words[t];
    const float x0 = (float)(int8_t)(wv & 0xff), y0 = (float)(int8_t)((wv >> 8) & 0xff);
    const float x1 = (float)(int8_t)((wv >> 16) & 0xff), y1 = (float)(int8_t)((wv >> 24) & 0xff);
    const int r0 = round_half_away(__fadd_rn(__fmul_rn(x0, bb), __fmul_rn(y0, a)));
    const int c0 = round_half_away(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, bb)));
    const int r1 = round_half_away(__fadd_rn(__fmul_rn(x1, bb), __fmul_rn(y1, a)));
    const int c1 = round_half_away(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, bb)));
    const int v0 = bc[r0 * L.pitch + c0], v1 = bc[r1 * L.pitch + c1];
    nib |= (unsigned)(v0 < v1) << t;
  }
  const unsigned other = __shfl_xor(nib, 1, 64);
  const size_t o = (size_t)b * kp_cap + in.slot;
  if ((lane & 1) == 0) desc[o * 32 + (lane >> 1)] = (uint8_t)(nib | (other << 4));
  if (lane == 0) {
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
  gfs::DevBuf<int> d_xt_start, d_xt_n, d_yt_start, d_yt_n, d_cell_cnt, d_cand_off, d_kp_count, d_mono;
  gfs::DevBuf<float> d_xt_alpha, d_yt_alpha;
  gfs::DevBuf<uint32_t> d_slab, d_cand;
  gfs::DevBuf<KpIn> d_kpin;
  gfs::DevBuf<gfs_keypoint> d_kps;
  gfs::DevBuf<uint8_t> d_desc;
  gfs::DevBuf<int8_t> d_ic_du, d_ic_dv, d_pattern;
  int ic_n = 0;
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
  GFS_REQUIRE(G.pyr_bytes <= h->cap_pyr && G.blur_bytes <= h->cap_blur && G.slab_entries <= h->cap_slab &&
                  G.cells.size() <= h->cap_cells && G.kp_cap <= h->cap_kp,
              GFS_ERR_CAPACITY, "image %dx%d needs more workspace than the handle reserved", cols, rows);
  hipStream_t s = h->stream;
  GFS_HIP(hipMemcpyAsync(h->d_levels.p, G.levels.data(), G.levels.size() * sizeof(LevelDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_cells.p, G.cells.data(), G.cells.size() * sizeof(CellDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_tiles.p, G.blur_tiles.data(), G.blur_tiles.size() * sizeof(BlurTileDev), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_start.p, G.xt_start.data(), G.xt_start.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_n.p, G.xt_n.data(), G.xt_n.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_xt_alpha.p, G.xt_alpha.data(), G.xt_alpha.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_start.p, G.yt_start.data(), G.yt_start.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_n.p, G.yt_n.data(), G.yt_n.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_yt_alpha.p, G.yt_alpha.data(), G.yt_alpha.size() * 4, hipMemcpyHostToDevice, s));
  GFS_HIP(hipStreamSynchronize(s));
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
  // 1. pyramid chain (level l depends on l-1)
  for (int l = 1; l < nl; l++) {
    const LevelDev& L = G.levels[l];
    dim3 grid(gfs::div_up(L.cols, 64), gfs::div_up(L.rows, 4), B);
    GFS_LAUNCH("k_pyr_area", k_pyr_area, grid, dim3(64, 4), 0, s, h->d_levels.p, l, l0, h->d_pyr.p, cap_pyr,
               h->d_xt_start.p, h->d_xt_n.p, h->d_xt_alpha.p, h->d_yt_start.p, h->d_yt_n.p, h->d_yt_alpha.p);
  }
  // 2. FAST cells of all levels, all frames in one launch
  const size_t lds = 2 * (size_t)G.max_tile_w * G.max_tile_h;
  GFS_LAUNCH("k_fast_cells", k_fast_cells, dim3(n_cells, B), dim3(256), lds, s, h->d_levels.p, h->d_cells.p, l0,
             h->d_pyr.p, cap_pyr, h->P.ini_th, h->P.min_th, n_cells, cap_slab, h->d_slab.p, h->d_cell_cnt.p);
  GFS_LAUNCH("k_cand_pack", k_cand_pack, dim3(B), dim3(256), 0, s, h->d_levels.p, h->d_cells.p, nl, n_cells, cap_slab,
             h->d_slab.p, h->d_cell_cnt.p, cap_slab, h->d_cand.p, h->d_cand_off.p);
  GFS_HIP(hipMemcpyAsync(h->h_cand_off.p, h->d_cand_off.p, (size_t)B * (nl + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  int max_total = 0;
  for (int b = 0; b < B; b++) max_total = std::max(max_total, h->h_cand_off.p[(size_t)b * (nl + 1) + nl]);
  if (max_total > 0)
    GFS_HIP(hipMemcpy2DAsync(h->h_cand.p, cap_slab * 4, h->d_cand.p, cap_slab * 4, (size_t)max_total * 4, B,
                             hipMemcpyDeviceToHost, s));
  GFS_HIP(hipEventRecord(h->ev_copy, s));
  // 3. blur of every level runs on the GPU while the host distributes keypoints
  const int* tp = kBlurTaps[h->P.blur_variant ? 1 : 0];
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
    GFS_LAUNCH("k_orient_brief", k_orient_brief, dim3(gfs::div_up(max_n, 4), B), dim3(256), 0, s, h->d_levels.p, l0,
               h->d_pyr.p, cap_pyr, h->d_blur.p, cap_blur, h->d_kpin.p, h->d_kp_count.p, h->cap_kp, h->d_ic_du.p,
               h->d_ic_dv.p, h->ic_n, h->d_pattern.p, h->d_kps.p, h->d_desc.p);
  }
  h->last_B = B;
  h->last_l0 = l0;
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
  GFS_REQUIRE(2 * (size_t)G.max_tile_w * G.max_tile_h <= 60000, GFS_ERR_UNSUPPORTED, "FAST cell tile too large for LDS");
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
  A(h->d_kps.alloc(B * h->cap_kp));
  A(h->d_desc.alloc(B * h->cap_kp * 32));
  A(h->h_cand_off.alloc(B * (nl + 1)));
  A(h->h_kp_count.alloc(B));
  A(h->h_mono.alloc(B));
  A(h->h_cand.alloc(B * h->cap_slab));
  A(h->h_kpin.alloc(B * h->cap_kp));
  std::vector<int8_t> du, dv;
  gfs::ic_angle_offsets(h->P.umax, du, dv);
  h->ic_n = (int)du.size();
  A(h->d_ic_du.alloc(du.size()));
  A(h->d_ic_dv.alloc(dv.size()));
  A(h->d_pattern.alloc(1024));
#undef A
  if (rc) return rc;
  GFS_HIP(hipMemcpy(h->d_ic_du.p, du.data(), du.size(), hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(h->d_ic_dv.p, dv.data(), dv.size(), hipMemcpyHostToDevice));
  GFS_HIP(hipMemcpy(h->d_pattern.p, kPattern, 1024, hipMemcpyHostToDevice));
  GFS_HIP(hipMemset(h->d_kp_count.p, 0, B * sizeof(int)));
  h->host_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("GFS_ORB_HOST_THREADS")) h->host_threads = std::max(1, atoi(e));
  h->pool.reset(new WorkerPool(h->host_threads));
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
