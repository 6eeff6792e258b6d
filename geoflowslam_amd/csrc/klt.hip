// Optical-flow front end on MI355X: cv::buildOpticalFlowPyramid (reference call sites src/Frame.cc:373, 505, 1415),
// cv::calcOpticalFlowPyrLK on such pyramids and ORBmatcher::fbKltTracking (src/ORBmatcher.cc:2186-2297, identical in
// src/Tracking.cc:3262-3366) — SURVEY.md §8(f) rank 4.  OpenCV's algorithm (video/src/lkpyramid.cpp, imgproc/src/pyramids.cpp,
// 4.5.4 semantics) is restated in oracle/klt_oracle.cpp; this file computes the same values bit for bit.
//
// Pyramid: one buffer per batch, a frame = the concatenation of its padded levels (bytes) + the same layout of short2
// derivatives.  5 launches per batch (level 0 + border, one pyrDown per level, one Scharr pass over all levels); every thread
// produces one padded pixel, borders included, so there is no separate border pass and no intra-launch dependency.
//
// Tracker: ONE WAVE PER POINT, no workgroup barriers.  The win x win window is cut into runs of 4 pixels in a row; lane l
// owns runs l, l + 64, ...  Per Newton step the wave first copies the (win + 1)^2 bytes under the window into LDS with a few
// coalesced unaligned dword loads (the texture addresser, not HBM, is what scattered per-run loads saturate), then every lane
// blends its runs from LDS: v_perm_b32 pairs neighbouring bytes into 16-bit halves and v_dot2_i32_i16 applies two of the four
// 14-bit bilinear weights at once.  The template patch (I << 5, Ix, Iy as packed int16, 24 B per run) is private to the owning
// lane and parked in LDS ([round][lane] layout, conflict-free b64 accesses); residual and mismatch vector again use packed
// subtract + v_dot2.  All sums (A11, A12, A22, b1, b2, L1 residual) are INTEGER sums: int32 per lane, then split into 16-bit
// halves, reduced over the wave by recursive halving on DPP (two ds_bpermute in total) and rebuilt exactly in double before
// the single rounding to float — independent of the summation order, which is what makes the result reproducible and equal to
// the oracle.  The scalar 2x2 algebra runs redundantly in all lanes; window origins are moved to scalar registers
// (readfirstlane) so the address arithmetic is SALU work.  fbKltTracking's forward pass, gates, backward pass and
// forward-backward distance run back to back in the same wave.  Measured (MI355X, 128 pairs x 1000 points, window 35):
// ~10 k VALU instructions per point, VALU ~85 % busy — the kernel is integer-ALU bound, HBM traffic is negligible.
// The round loops are unrolled at compile time for the window sizes of the reference's configurations (KLT_DISPATCH).
#include <memory>
#include <mutex>

#include "gfs_common.hpp"

namespace {

using gfs::DevBuf;
using gfs::PinBuf;

constexpr int kKltWavesPerBlock = 4;
constexpr int kKltThreads = 64 * kKltWavesPerBlock;
constexpr int kKltSlack = 256;  // bytes after the last level: masked tail pixels of a run may be loaded from there

struct KltGeom {
  int n_levels, win, runs, ntasks, rounds, inv_runs, nd, win_dwords, stage_lpr, stage_shift, stage_rows;
  int width, height;
  int lw[GFS_KLT_MAX_LEVELS], lh[GFS_KLT_MAX_LEVELS];
  long long off[GFS_KLT_MAX_LEVELS + 1];
  long long frame_stride;  // pixels between consecutive frames of a pyramid batch (images: bytes, derivatives: short2)
};

struct KltParams {
  int max_level, max_iter, flags;
  double eps2, min_eig_thr;
};

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pyramid kernels
// ---------------------------------------------------------------------------------------------------------------------
// Level 0 with its reflect-101 border: thread = 4 consecutive padded pixels (one aligned dword store).  A group that lies
// inside the image columns is one unaligned dword load; groups that touch the border take the per-byte path.
__global__ void __launch_bounds__(256) k_klt_level0(KltGeom G, const uint8_t* __restrict__ src, int stride, long long src_frame,
                                                     uint8_t* __restrict__ pyr) {
  const int pw = G.lw[0] + 2 * G.win, ph = G.lh[0] + 2 * G.win, total = pw * ph;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const int y = i / pw, x = i - y * pw;
  const uint8_t* sf = src + (long long)blockIdx.y * src_frame;
  uint8_t* dst = pyr + (long long)blockIdx.y * G.frame_stride + i;
  if (x >= G.win && x + 3 < G.win + G.lw[0] && i + 3 < total) {
    unsigned v;
    __builtin_memcpy(&v, sf + (long long)reflect101(y - G.win, G.lh[0]) * stride + (x - G.win), 4);
    *(unsigned*)dst = v;  // frame_stride and i are multiples of 4
    return;
  }
  for (int k = 0; k < 4 && i + k < total; k++) {
    const int yy = (i + k) / pw, xx = (i + k) - yy * pw;
    dst[k] = sf[(long long)reflect101(yy - G.win, G.lh[0]) * stride + reflect101(xx - G.win, G.lw[0])];
  }
}

// Level l >= 1: pyrDown of level l - 1 (5x5 binomial, (s + 128) >> 8) for every padded pixel; thread = 2 consecutive padded
// pixels.  The taps reach at most two pixels outside level l - 1, where its own reflect-101 border holds exactly what
// pyrDown's border rule would fetch.  Two neighbours inside the level share 7 source columns: five unaligned 8-byte loads.
__device__ __forceinline__ int pyrdown_one(const uint8_t* p, int ppw) {
  int s = 0;
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const uint8_t* row = p + (long long)r * ppw;
    const int hs = row[0] + row[4] + 4 * (row[1] + row[3]) + 6 * row[2];
    s += (r == 0 || r == 4) ? hs : (r == 2 ? 6 * hs : 4 * hs);
  }
  return (s + 128) >> 8;
}
__global__ void __launch_bounds__(256) k_klt_pyrdown(KltGeom G, int level, uint8_t* __restrict__ pyr) {
  const int cw = G.lw[level], ch = G.lh[level], pw = cw + 2 * G.win, ph = ch + 2 * G.win, total = pw * ph;
  const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i >= total) return;
  const int y = i / pw, x = i - y * pw;
  const int ppw = G.lw[level - 1] + 2 * G.win;
  uint8_t* frame = pyr + (long long)blockIdx.y * G.frame_stride;
  const uint8_t* prev = frame + G.off[level - 1] + (long long)(G.win - 2) * ppw + (G.win - 2);  // tap (-2, -2) of source pixel (0, 0)
  uint8_t* dst = frame + G.off[level] + i;
  if (x >= G.win && x + 1 < G.win + cw && y >= G.win && y < G.win + ch) {
    const uint8_t* p = prev + (long long)(2 * (y - G.win)) * ppw + 2 * (x - G.win);
    int s0 = 0, s1 = 0;
#pragma unroll
    for (int r = 0; r < 5; r++) {
      unsigned long long v;
      __builtin_memcpy(&v, p + (long long)r * ppw, 8);
      const int b0 = v & 0xff, b1 = (v >> 8) & 0xff, b2 = (v >> 16) & 0xff, b3 = (v >> 24) & 0xff, b4 = (v >> 32) & 0xff,
                b5 = (v >> 40) & 0xff, b6 = (v >> 48) & 0xff;
      const int h0 = b0 + b4 + 4 * (b1 + b3) + 6 * b2, h1 = b2 + b6 + 4 * (b3 + b5) + 6 * b4;
      const int wr = (r == 0 || r == 4) ? 1 : (r == 2 ? 6 : 4);
      s0 += wr * h0;
      s1 += wr * h1;
    }
    dst[0] = (uint8_t)((s0 + 128) >> 8);
    dst[1] = (uint8_t)((s1 + 128) >> 8);
    return;
  }
  for (int k = 0; k < 2 && i + k < total; k++) {
    const int yy = (i + k) / pw, xx = (i + k) - yy * pw;
    const int ix = reflect101(xx - G.win, cw), iy = reflect101(yy - G.win, ch);
    dst[k] = (uint8_t)pyrdown_one(prev + (long long)(2 * iy) * ppw + 2 * ix, ppw);
  }
}

// calcSharrDeriv on every level at once: thread = 4 consecutive padded pixels of one level (zero outside the level itself).
// The 3x3 stencils read the padded image, whose reflect-101 border equals calcSharrDeriv's own border rule; in flat addressing
// four neighbouring pixels need bytes [-1, +4] of three rows: three unaligned 8-byte loads instead of 32 byte loads.
__global__ void __launch_bounds__(256) k_klt_scharr(KltGeom G, const uint8_t* __restrict__ pyr, short2* __restrict__ deriv) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long long total = G.off[G.n_levels];
  if (i >= total) return;
  const uint8_t* frame = pyr + (long long)blockIdx.y * G.frame_stride;
  short2* out = deriv + (long long)blockIdx.y * G.frame_stride + i;
  int level = 0;
  while (level + 1 < G.n_levels && i >= G.off[level + 1]) level++;
  const int cw = G.lw[level], ch = G.lh[level], pw = cw + 2 * G.win;
  const int rel = (int)(i - G.off[level]);
  int y = rel / pw, x = rel - y * pw;
  const bool one_level = i + 3 < G.off[level + 1];
  bool any = false;
  if (one_level) {
    int xx = x, yy = y;
    for (int k = 0; k < 4; k++) {
      any |= xx >= G.win && xx < G.win + cw && yy >= G.win && yy < G.win + ch;
      if (++xx == pw) xx = 0, yy++;
    }
  }
  short2 d[4] = {make_short2(0, 0), make_short2(0, 0), make_short2(0, 0), make_short2(0, 0)};
  if (one_level && any) {
    const uint8_t* p = frame + i;
    unsigned long long ra, rb, rc;  // bytes [-1, +6] of the rows above, at and below
    __builtin_memcpy(&ra, p - pw - 1, 8);
    __builtin_memcpy(&rb, p - 1, 8);
    __builtin_memcpy(&rc, p + pw - 1, 8);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (x >= G.win && x < G.win + cw && y >= G.win && y < G.win + ch) {
        const int a0 = (ra >> (8 * k)) & 0xff, a1 = (ra >> (8 * k + 8)) & 0xff, a2 = (ra >> (8 * k + 16)) & 0xff;
        const int b0 = (rb >> (8 * k)) & 0xff, b2 = (rb >> (8 * k + 16)) & 0xff;
        const int c0 = (rc >> (8 * k)) & 0xff, c1 = (rc >> (8 * k + 8)) & 0xff, c2 = (rc >> (8 * k + 16)) & 0xff;
        const int t0m = (a0 + c0) * 3 + b0 * 10, t0p = (a2 + c2) * 3 + b2 * 10;
        const int t1m = c0 - a0, t1c = c1 - a1, t1p = c2 - a2;
        d[k] = make_short2((short)(t0p - t0m), (short)((t1p + t1m) * 3 + t1c * 10));
      }
      if (++x == pw) x = 0, y++;
    }
  } else if (!one_level) {  // the group straddles two levels (at most one group per level): pixel by pixel
    for (int k = 0; k < 4 && i + k < total; k++) {
      const long long j = i + k;
      int l2 = level;
      while (l2 + 1 < G.n_levels && j >= G.off[l2 + 1]) l2++;
      const int w2 = G.lw[l2], h2 = G.lh[l2], pw2 = w2 + 2 * G.win, r2 = (int)(j - G.off[l2]);
      const int y2 = r2 / pw2, x2 = r2 - y2 * pw2;
      if (x2 >= G.win && x2 < G.win + w2 && y2 >= G.win && y2 < G.win + h2) {
        const uint8_t* p = frame + j;
        const int a0 = p[-pw2 - 1], a1 = p[-pw2], a2 = p[-pw2 + 1], b0 = p[-1], b2 = p[1], c0 = p[pw2 - 1], c1 = p[pw2], c2 = p[pw2 + 1];
        d[k] = make_short2((short)(((a2 + c2) * 3 + b2 * 10) - ((a0 + c0) * 3 + b0 * 10)),
                           (short)(((c2 - a2) + (c0 - a0)) * 3 + (c1 - a1) * 10));
      }
    }
  }
  if (i + 3 < total) {
    *(uint4*)out = make_uint4(__builtin_bit_cast(unsigned, d[0]), __builtin_bit_cast(unsigned, d[1]), __builtin_bit_cast(unsigned, d[2]),
                              __builtin_bit_cast(unsigned, d[3]));  // 16-byte aligned: frame_stride and i are multiples of 4 elements
  } else {
    for (int k = 0; k < 4 && i + k < total; k++) out[k] = d[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Tracker
// ---------------------------------------------------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned load_u32(const uint8_t* p) {  // unaligned dword load (gfx950 runs in unaligned-access mode)
  unsigned v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ __forceinline__ v2s as_v2s(unsigned v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ unsigned as_u32(v2s v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ int dot2(unsigned a, unsigned b, int c) {  // a.lo * b.lo + a.hi * b.hi + c on signed 16-bit halves
  return __builtin_amdgcn_sdot2(as_v2s(a), as_v2s(b), c, false);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int CTRL>
__device__ __forceinline__ int dpp(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128;

// Exact wave-wide sums of two per-lane int32 values (|v| < 2^31 per lane, 64 lanes -> up to 37 bits): each value is split
// into a signed high and an unsigned low 16-bit half, the four halves are reduced together by recursive halving (lane pairs,
// then quads, split the four between them), row rotations and two cross-row exchanges; every lane then rebuilds
// fma(hi, 65536, lo) on the two halves, which are exact in float: ONE rounding of the exact sum == (float)(int64 sum).
__device__ __forceinline__ void wave_sum2_exact(int a, int b, int lane, float& fa, float& fb) {
  const int ah = a >> 16, al = a & 0xffff, bh = b >> 16, bl = b & 0xffff;
  const bool o0 = lane & 1, o1 = lane & 2;
  const int H = (o0 ? bh : ah) + dpp<kDppXor1>(o0 ? ah : bh);  // even lanes: a, odd lanes: b
  const int L = (o0 ? bl : al) + dpp<kDppXor1>(o0 ? al : bl);
  int v = (o1 ? L : H) + dpp<kDppXor2>(o1 ? H : L);            // lane & 3: 0 a.hi, 1 b.hi, 2 a.lo, 3 b.lo (sum over the quad)
  v += dpp<kDppRor4>(v);
  v += dpp<kDppRor8>(v);                                        // sum over the row of 16
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  const int sah = dpp<0x00>(v), sbh = dpp<0x55>(v), sal = dpp<0xAA>(v), sbl = dpp<0xFF>(v);
  fa = __fmaf_rn((float)sah, 65536.f, (float)sal);  // both halves are exact in float (< 2^23): one rounding of the exact sum
  fb = __fmaf_rn((float)sbh, 65536.f, (float)sbl);
}
// Same for three values (the gradient matrix): lane & 3 selects a, b, c after the pair / quad steps, the fourth slot idles.
__device__ __forceinline__ void wave_sum3_exact(int a, int b, int c, int lane, float& fa, float& fb, float& fc) {
  const bool o0 = lane & 1, o1 = lane & 2;
  // pairs: even lanes keep (a, c), odd lanes keep (b, 0); quads: lanes 0, 1 keep the first, lanes 2, 3 the second of their two
  auto reduce = [&](int x, int y, int z) {
    const int P = (o0 ? y : x) + dpp<kDppXor1>(o0 ? x : y);
    const int Q = (o0 ? 0 : z) + dpp<kDppXor1>(o0 ? z : 0);
    int v = (o1 ? Q : P) + dpp<kDppXor2>(o1 ? P : Q);  // lane & 3: 0 x, 1 y, 2 z, 3 nothing
    v += dpp<kDppRor4>(v);
    v += dpp<kDppRor8>(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
  };
  const int hi = reduce(a >> 16, b >> 16, c >> 16), lo = reduce(a & 0xffff, b & 0xffff, c & 0xffff);
  fa = __fmaf_rn((float)dpp<0x00>(hi), 65536.f, (float)dpp<0x00>(lo));
  fb = __fmaf_rn((float)dpp<0x55>(hi), 65536.f, (float)dpp<0x55>(lo));
  fc = __fmaf_rn((float)dpp<0xAA>(hi), 65536.f, (float)dpp<0xAA>(lo));
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// The four 14-bit weights as two v_dot2 operands: Wl = (w00, w10) multiplies the vertical pair under a pixel's left tap
// column, Wr = (w01, w11) the pair under its right tap column.
__device__ __forceinline__ void bilinear_weights(float a, float b, unsigned& Wl, unsigned& Wr) {
  const int w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
  const int w01 = __float2int_rn(a * (1.f - b) * 16384.f);
  const int w10 = __float2int_rn((1.f - a) * b * 16384.f);
  const int w11 = 16384 - w00 - w01 - w10;
  Wl = (unsigned)uniform(w00 | (w10 << 16));  // all four lie in [0, 16384]: they fit the signed halves of v_dot2
  Wr = (unsigned)uniform(w01 | (w11 << 16));
}
// Copies the (win + 1) rows x 4 * nd bytes of the window whose top-left byte is `base` (wave-uniform) into the wave's LDS
// window buffer.  16 (nd <= 16) or 32 lanes walk down one column of dwords each, 4 or 2 rows per step: consecutive lanes
// fetch consecutive (unaligned) dwords of a row, so the texture addresser sees a few coalesced wave loads per window instead
// of 4 scattered ones per run and round, and the addresses advance by one add per step.
__device__ __forceinline__ void stage_window(const KltGeom& G, const uint8_t* base, int pitch, unsigned* sWin, int lane) {
  const int col = lane & (G.stage_lpr - 1), row0 = lane >> G.stage_shift;
  if (col < G.nd) {
    const uint8_t* p = base + row0 * pitch + col * 4;
    unsigned* q = sWin + row0 * G.nd + col;
    const int dp = pitch * G.stage_rows, dq = G.nd * G.stage_rows;
    for (int row = row0; row <= G.win; row += 4 * G.stage_rows, p += 4 * dp, q += 4 * dq) {  // four loads in flight
      unsigned v[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (row + k * G.stage_rows <= G.win) v[k] = load_u32(p + k * dp);
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (row + k * G.stage_rows <= G.win) q[k * dq] = v[k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Five bytes of two staged window rows -> four bilinear samples << 5 (CV_DESCALE(..., W_BITS1 - 5)), as two packed pairs.
// i0 = dword index of the run in the window buffer.  Column i of the two rows is paired into one dword (upper, lower) by
// v_perm_b32 (bytes 0-3 select from the second operand, 4-7 from the first, 0x0c = zero); a column serves as the right taps
// of pixel i - 1 and as the left taps of pixel i, so 5 permutes feed 8 v_dot2.
__device__ __forceinline__ void blend4(const unsigned* sWin, int i0, int nd, unsigned Wl, unsigned Wr, unsigned& v01, unsigned& v23) {
  const unsigned a = sWin[i0], a4 = sWin[i0 + 1], b = sWin[i0 + nd], b4 = sWin[i0 + nd + 1];
  const unsigned c0 = __builtin_amdgcn_perm(b, a, 0x0c040c00u), c1 = __builtin_amdgcn_perm(b, a, 0x0c050c01u),
                 c2 = __builtin_amdgcn_perm(b, a, 0x0c060c02u), c3 = __builtin_amdgcn_perm(b, a, 0x0c070c03u),
                 c4 = __builtin_amdgcn_perm(b4, a4, 0x0c040c00u);
  const int v0 = dot2(c0, Wl, dot2(c1, Wr, 256)) >> 9, v1 = dot2(c1, Wl, dot2(c2, Wr, 256)) >> 9;
  const int v2 = dot2(c2, Wl, dot2(c3, Wr, 256)) >> 9, v3 = dot2(c3, Wl, dot2(c4, Wr, 256)) >> 9;
  v01 = (unsigned)v0 | ((unsigned)v1 << 16);  // samples are in [0, 8160]
  v23 = (unsigned)v2 | ((unsigned)v3 << 16);
}

// LKTrackerInvoker::operator() over the levels max_level .. 0 for one point, executed by one wave.
// lds: this wave's 3 * rounds * 64 uint2 + the window buffer.  All arguments and results are wave-uniform.
template <int R>  // R > 0: the number of rounds (= G.rounds) is a compile-time constant and the round loops unroll; 0: generic
__device__ void klt_track_point(const KltGeom& G, const KltParams& P, const uint8_t* __restrict__ Ipyr,
                                const short2* __restrict__ dIpyr, const uint8_t* __restrict__ Jpyr, float2 prev, float2& next,
                                int& status, float& err, uint2* lds, int lane) {
  const int win = G.win;
  const float half = (win - 1) * 0.5f;
  uint2* sI = lds;
  const int rounds = R ? R : G.rounds;
  uint2* sIx = lds + rounds * 64;
  uint2* sIy = lds + 2 * rounds * 64;
  unsigned* sWin = (unsigned*)(lds + 3 * rounds * 64);
  for (int level = P.max_level; level >= 0; level--) {
    const int w = G.lw[level], h = G.lh[level], pitch = w + 2 * win;
    const long long org = G.off[level] + (long long)win * pitch + win;
    const float sc = 1.f / (float)(1 << level);
    float px = prev.x * sc, py = prev.y * sc;
    float nx, ny;
    if (level == P.max_level) {
      if (P.flags & GFS_KLT_USE_INITIAL_FLOW) {
        nx = next.x * sc;
        ny = next.y * sc;
      } else {
        nx = px;
        ny = py;
      }
    } else {
      nx = next.x * 2.f;
      ny = next.y * 2.f;
    }
    next = make_float2(nx, ny);
    px -= half;
    py -= half;
    const int ipx = uniform((int)floorf(px)), ipy = uniform((int)floorf(py));
    if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
      if (level == 0) {
        status = 0;
        err = 0.f;
      }
      continue;
    }
    unsigned W0, W1;
    bilinear_weights(px - ipx, py - ipy, W0, W1);

    // --- template patch + gradient matrix ---
    const uint8_t* Ib = Ipyr + org + (long long)ipy * pitch + ipx;
    const short2* dIb = dIpyr + org + (long long)ipy * pitch + ipx;
    int a11 = 0, a12 = 0, a22 = 0;  // per lane <= 64 pixels x 4080^2 < 2^31
    stage_window(G, Ib, pitch, sWin, lane);
#pragma unroll R > 0 ? R : 1
    for (int r = 0; r < rounds; r++) {
      const int t = r * 64 + lane;
      unsigned I01 = 0, I23 = 0, X01 = 0, X23 = 0, Y01 = 0, Y23 = 0;
      if (t < G.ntasks) {
        const int ty = (t * G.inv_runs) >> 16, x0 = (t - ty * G.runs) * 4;
        blend4(sWin, t + ty, G.nd, W0, W1, I01, I23);
        const unsigned* d0 = (const unsigned*)(dIb + (ty * pitch + x0));
        const unsigned* d1 = d0 + pitch;
        unsigned da[5], db[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
          da[i] = d0[i];
          db[i] = d1[i];
        }
        unsigned cx[5], cy[5];  // column i of the two rows: (upper, lower) x derivatives, (upper, lower) y derivatives
#pragma unroll
        for (int i = 0; i < 5; i++) {
          cx[i] = __builtin_amdgcn_perm(db[i], da[i], 0x05040100u);
          cy[i] = __builtin_amdgcn_perm(db[i], da[i], 0x07060302u);
        }
        int ix[4], iy[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          ix[i] = dot2(cx[i], W0, dot2(cx[i + 1], W1, 8192)) >> 14;
          iy[i] = dot2(cy[i], W0, dot2(cy[i + 1], W1, 8192)) >> 14;
          if (x0 + i >= win) ix[i] = iy[i] = 0;  // beyond the window: contributes nothing anywhere
        }
        X01 = (unsigned)(ix[0] & 0xffff) | ((unsigned)ix[1] << 16);
        X23 = (unsigned)(ix[2] & 0xffff) | ((unsigned)ix[3] << 16);
        Y01 = (unsigned)(iy[0] & 0xffff) | ((unsigned)iy[1] << 16);
        Y23 = (unsigned)(iy[2] & 0xffff) | ((unsigned)iy[3] << 16);
        a11 = dot2(X01, X01, dot2(X23, X23, a11));
        a12 = dot2(X01, Y01, dot2(X23, Y23, a12));
        a22 = dot2(Y01, Y01, dot2(Y23, Y23, a22));
      }
      sI[r * 64 + lane] = make_uint2(I01, I23);
      sIx[r * 64 + lane] = make_uint2(X01, X23);
      sIy[r * 64 + lane] = make_uint2(Y01, Y23);
    }
    const float FLT_SCALE = 1.f / (1 << 20);
    float A11, A12, A22;
    wave_sum3_exact(a11, a12, a22, lane, A11, A12, A22);
    A11 *= FLT_SCALE;
    A12 *= FLT_SCALE;
    A22 *= FLT_SCALE;
    float D = __fsub_rn(__fmul_rn(A11, A22), __fmul_rn(A12, A12));
    const float dif = A11 - A22;
    const float min_eig = __fsub_rn(A22 + A11, sqrtf(__fadd_rn(__fmul_rn(dif, dif), __fmul_rn(__fmul_rn(4.f, A12), A12)))) /
                          (float)(2 * win * win);
    if (P.flags & GFS_KLT_GET_MIN_EIGENVALS) err = min_eig;
    if ((double)min_eig < P.min_eig_thr || D < 1.1920928955078125e-07f) {
      if (level == 0) status = 0;
      continue;
    }
    D = 1.f / D;
    nx -= half;
    ny -= half;
    float pdx = 0.f, pdy = 0.f;
    const uint8_t* Jb = Jpyr + org;
    for (int j = 0; j < P.max_iter; j++) {
      const int inx = uniform((int)floorf(nx)), iny = uniform((int)floorf(ny));
      if (inx < -win || inx >= w || iny < -win || iny >= h) {
        if (level == 0) status = 0;
        break;
      }
      bilinear_weights(nx - inx, ny - iny, W0, W1);
      const uint8_t* Jp = Jb + (long long)iny * pitch + inx;
      int b1 = 0, b2 = 0;  // per lane <= 64 pixels x 8160 x 4080 < 2^31
      stage_window(G, Jp, pitch, sWin, lane);
  #pragma unroll R > 0 ? R : 1
    for (int r = 0; r < rounds; r++) {
        const int t = r * 64 + lane;
        if (t < G.ntasks) {
          unsigned J01, J23;
          blend4(sWin, t + ((t * G.inv_runs) >> 16), G.nd, W0, W1, J01, J23);
          const uint2 I = sI[r * 64 + lane], X = sIx[r * 64 + lane], Y = sIy[r * 64 + lane];
          const unsigned d01 = as_u32(as_v2s(J01) - as_v2s(I.x)), d23 = as_u32(as_v2s(J23) - as_v2s(I.y));
          b1 = dot2(d01, X.x, dot2(d23, X.y, b1));
          b2 = dot2(d01, Y.x, dot2(d23, Y.y, b2));
        }
      }
      float fb1, fb2;
      wave_sum2_exact(b1, b2, lane, fb1, fb2);
      fb1 *= FLT_SCALE;
      fb2 *= FLT_SCALE;
      const float dx = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb2), __fmul_rn(A22, fb1)), D);
      const float dy = __fmul_rn(__fsub_rn(__fmul_rn(A12, fb1), __fmul_rn(A11, fb2)), D);
      nx += dx;
      ny += dy;
      next = make_float2(nx + half, ny + half);
      if (__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy)) <= P.eps2) break;
      if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {  // (double)f < 0.01 <=> f <= 0.01f (0.01f < 0.01 < its successor)
        next.x = __fsub_rn(next.x, __fmul_rn(dx, 0.5f));
        next.y = __fsub_rn(next.y, __fmul_rn(dy, 0.5f));
        break;
      }
      pdx = dx;
      pdy = dy;
    }
    if (status && level == 0 && !(P.flags & GFS_KLT_GET_MIN_EIGENVALS)) {  // L1 residual of the final position
      const float ex = next.x - half, ey = next.y - half;
      const int iex = uniform((int)floorf(ex)), iey = uniform((int)floorf(ey));
      if (iex < -win || iex >= w || iey < -win || iey >= h) {
        status = 0;
        continue;
      }
      bilinear_weights(ex - iex, ey - iey, W0, W1);
      const uint8_t* Jp = Jb + (long long)iey * pitch + iex;
      long long e = 0;
      stage_window(G, Jp, pitch, sWin, lane);
  #pragma unroll R > 0 ? R : 1
    for (int r = 0; r < rounds; r++) {
        const int t = r * 64 + lane;
        if (t < G.ntasks) {
          const int ty = (t * G.inv_runs) >> 16, x0 = (t - ty * G.runs) * 4;
          unsigned J01, J23;
          blend4(sWin, t + ty, G.nd, W0, W1, J01, J23);
          const uint2 I = sI[r * 64 + lane];
          const v2s d01 = as_v2s(J01) - as_v2s(I.x), d23 = as_v2s(J23) - as_v2s(I.y);
          const int d[4] = {d01.x, d01.y, d23.x, d23.y};
#pragma unroll
          for (int i = 0; i < 4; i++)
            if (x0 + i < win) e += abs(d[i]);
        }
      }
      err = __fmul_rn((float)wave_sum_i64(e), 1.f) / (float)(32 * win * win);
    }
  }
}

// Workgroup -> (frame, group of 4 points).  Workgroups are handed to the 8 XCDs round-robin by their linear id and every XCD
// has its own L2: with a multiple of 8 frames in the batch, frame f is worked on by XCD f % 8 only, so each pyramid is
// fetched into one L2 instead of all eight.  gx = workgroups per frame.
__device__ __forceinline__ void klt_block_to_work(int gx, int n_frames, int& f, int& bx) {
  const int L = blockIdx.x;
  if ((n_frames & 7) == 0) {
    const int slot = L >> 3;
    f = (L & 7) + 8 * (slot / gx);
    bx = slot % gx;
  } else {
    f = L / gx;
    bx = L % gx;
  }
}

// calcOpticalFlowPyrLK for B pairs: wave = one point.  pts layouts: [B][pt_stride] float2.
template <int R>
__global__ void __launch_bounds__(kKltThreads) k_klt_track(KltGeom G, KltParams P, const uint8_t* __restrict__ prev_img,
                                                           const short2* __restrict__ prev_deriv,
                                                           const uint8_t* __restrict__ next_img, const int* __restrict__ n_pts,
                                                           int pt_stride, const float2* __restrict__ prev_pts,
                                                           float2* __restrict__ next_pts, uint8_t* __restrict__ status_out,
                                                           float* __restrict__ err_out, int gx, int n_frames) {
  extern __shared__ uint2 s_klt[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int f, bx;
  klt_block_to_work(gx, n_frames, f, bx);
  const int i = bx * kKltWavesPerBlock + wave;
  if (i >= n_pts[f]) return;
  const long long fo = (long long)f * G.frame_stride, po = (long long)f * pt_stride + i;
  float2 next = (P.flags & GFS_KLT_USE_INITIAL_FLOW) ? next_pts[po] : make_float2(0.f, 0.f);
  int status = 1;
  float err = 0.f;
  klt_track_point<R>(G, P, prev_img + fo, prev_deriv + fo, next_img + fo, prev_pts[po], next, status, err,
                  s_klt + wave * (3 * G.rounds * 64 + (G.win_dwords + 1) / 2), lane);
  if (lane == 0) {
    next_pts[po] = next;
    status_out[po] = (uint8_t)status;
    err_out[po] = err;
  }
}

// fbKltTracking for B pairs: forward (prev -> cur, nbpyrlvl levels), gates (status, err > ferr, inBorder), backward
// (cur -> prev, level 0, started from the original key point), forward-backward distance.  Wave = one point.
template <int R>
__global__ void __launch_bounds__(kKltThreads) k_klt_fb(KltGeom G, KltParams P, const uint8_t* __restrict__ prev_img,
                                                        const short2* __restrict__ prev_deriv, const uint8_t* __restrict__ cur_img,
                                                        const short2* __restrict__ cur_deriv, const int* __restrict__ n_pts,
                                                        int pt_stride, const float2* __restrict__ kps, float2* __restrict__ priors,
                                                        uint8_t* __restrict__ kpstatus, int* __restrict__ n_good, float ferr,
                                                        float fmax_fbklt_dist, int gx, int n_frames) {
  extern __shared__ uint2 s_klt[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int f, bx;
  klt_block_to_work(gx, n_frames, f, bx);
  const int i = bx * kKltWavesPerBlock + wave;
  if (i >= n_pts[f]) return;
  uint2* lds = s_klt + wave * (3 * G.rounds * 64 + (G.win_dwords + 1) / 2);
  const long long fo = (long long)f * G.frame_stride, po = (long long)f * pt_stride + i;
  const float2 kp = kps[po];
  float2 fwd = priors[po];
  int status = 1;
  float err = 0.f;
  klt_track_point<R>(G, P, prev_img + fo, prev_deriv + fo, cur_img + fo, kp, fwd, status, err, lds, lane);
  bool ok = status && !(err > ferr) && 1.f <= fwd.x && fwd.x < (float)G.width - 1.f && 1.f <= fwd.y && fwd.y < (float)G.height - 1.f;
  if (ok) {
    KltParams Pb = P;
    Pb.max_level = 0;
    float2 back = kp;
    int bstatus = 1;
    float berr = 0.f;
    klt_track_point<R>(G, Pb, cur_img + fo, cur_deriv + fo, prev_img + fo, fwd, back, bstatus, berr, lds, lane);
    if (!bstatus) {
      ok = false;
    } else {
      const float dx = kp.x - back.x, dy = kp.y - back.y;
      ok = !(__dsqrt_rn(__dadd_rn(__dmul_rn((double)dx, (double)dx), __dmul_rn((double)dy, (double)dy))) > (double)fmax_fbklt_dist);
    }
  }
  if (lane == 0) {
    priors[po] = fwd;
    kpstatus[po] = ok ? 1 : 0;
    if (ok) atomicAdd(n_good + f, 1);
  }
}

// The points fbKltTracking kept, in order (the un_cur_pts / un_forw_pts / index vectors of SearchByProjectionWithOF,
// src/ORBmatcher.cc:2386-2395): one workgroup per frame, ballot + prefix sums keep the order.
__global__ void __launch_bounds__(256) k_klt_compact(const int* __restrict__ n_pts, int pt_stride, const float2* __restrict__ kps,
                                                      const float2* __restrict__ priors, const uint8_t* __restrict__ kpstatus,
                                                      float2* __restrict__ out_a, float2* __restrict__ out_b, int* __restrict__ out_index,
                                                      int* __restrict__ out_n) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = n_pts[f];
  const size_t fo = (size_t)f * pt_stride;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + tid;
    const bool keep = i < n && kpstatus[fo + i] != 0;
    const unsigned long long ball = __ballot(keep);
    if (lane == 0) s_wave[wave] = __popcll(ball);
    __syncthreads();
    int ofs = s_base;
    for (int w = 0; w < wave; w++) ofs += s_wave[w];
    if (keep) {
      const int dst = ofs + __popcll(ball & ((1ull << lane) - 1));
      out_a[fo + dst] = kps[fo + i];
      out_b[fo + dst] = priors[fo + i];
      out_index[fo + dst] = i;
    }
    __syncthreads();
    if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    __syncthreads();
  }
  if (tid == 0) out_n[f] = s_base;
}

// vkpstatus.at(index.at(i)) = false for every track the F check rejected (src/ORBmatcher.cc:2401-2405)
__global__ void __launch_bounds__(256) k_klt_apply_mask(const int* __restrict__ m_pts, int pt_stride, const int* __restrict__ index,
                                                         const uint8_t* __restrict__ mask, uint8_t* __restrict__ kpstatus) {
  const int f = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m_pts[f]) return;
  const size_t fo = (size_t)f * pt_stride;
  if (!mask[fo + j]) kpstatus[fo + index[fo + j]] = 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
// Round counts of the windows the reference's configurations use (LKWindowSize 15 / 30 / 35 / 40 -> 1 / 4 / 5 / 7 rounds) get
// their own instantiation; every other window runs the generic loop.
#define KLT_DISPATCH(rounds, CALL) \
  switch (rounds) {                \
    case 1: CALL(1); break;        \
    case 4: CALL(4); break;        \
    case 5: CALL(5); break;        \
    case 7: CALL(7); break;        \
    default: CALL(0); break;       \
  }

struct gfs_klt {
  int device = 0, max_batch = 0, max_points = 0;
  KltGeom G{};
  hipStream_t stream = nullptr;
  std::mutex mu;
  size_t lds_bytes = 0;
  DevBuf<uint8_t> d_images;  // staging for host images
  PinBuf<uint8_t> h_images;
  DevBuf<float2> d_a, d_b;   // prev / next (kps / priors)
  DevBuf<uint8_t> d_status;
  DevBuf<float> d_err;
  DevBuf<int> d_n, d_good;
  PinBuf<float2> h_a, h_b;
  PinBuf<uint8_t> h_status;
  PinBuf<float> h_err;
  PinBuf<int> h_n, h_good;
};

struct gfs_klt_pyramid {
  gfs_klt* owner = nullptr;
  DevBuf<uint8_t> img;
  DevBuf<short2> deriv;
  int n_frames = 0;
};

namespace {
int klt_build(gfs_klt* h, gfs_klt_pyramid* pyr, const uint8_t* dev_images, int stride, int B, hipStream_t s) {
  const KltGeom& G = h->G;
  const int pw0 = G.lw[0] + 2 * G.win, ph0 = G.lh[0] + 2 * G.win;
  GFS_LAUNCH("k_klt_level0", k_klt_level0, dim3(gfs::div_up(gfs::div_up(pw0 * ph0, 4), 256), B), dim3(256), 0, s, G, dev_images, stride,
             (long long)stride * G.height, pyr->img.p);
  for (int l = 1; l < G.n_levels; l++) {
    const int n = (G.lw[l] + 2 * G.win) * (G.lh[l] + 2 * G.win);
    GFS_LAUNCH("k_klt_pyrdown", k_klt_pyrdown, dim3(gfs::div_up(gfs::div_up(n, 2), 256), B), dim3(256), 0, s, G, l, pyr->img.p);
  }
  GFS_LAUNCH("k_klt_scharr", k_klt_scharr, dim3((unsigned)((G.off[G.n_levels] + 1023) / 1024), B), dim3(256), 0, s, G,
             (const uint8_t*)pyr->img.p, pyr->deriv.p);
  pyr->n_frames = B;
  return GFS_OK;
}

KltParams fb_params(const gfs_klt* h, int nbpyrlvl) {
  KltParams P;
  P.max_level = nbpyrlvl < 0 ? 0 : (nbpyrlvl > h->G.n_levels - 1 ? h->G.n_levels - 1 : nbpyrlvl);  // calcOpticalFlowPyrLK clamps
  P.max_iter = 30;
  const double e = (double)0.01f;  // src/ORBmatcher.cc:2217-2221
  P.eps2 = e * e;
  P.flags = GFS_KLT_USE_INITIAL_FLOW | GFS_KLT_GET_MIN_EIGENVALS;  // :2227
  P.min_eig_thr = 1e-4;
  return P;
}
}  // namespace

extern "C" {

int gfs_klt_create(int device, int width, int height, int win, int max_level, int max_batch, int max_points, gfs_klt** out) {
  GFS_REQUIRE(out, GFS_ERR_INVALID_ARG, "gfs_klt_create: out is NULL");
  *out = nullptr;
  GFS_REQUIRE(width > 0 && height > 0 && win >= 3 && win <= 63 && max_level >= 0 && max_batch > 0 && max_points > 0,
              GFS_ERR_INVALID_ARG, "gfs_klt_create: invalid argument (window 3 .. 63)");
  GFS_REQUIRE(width <= 8192 && height <= 8192, GFS_ERR_CAPACITY, "gfs_klt_create: image larger than 8192 x 8192");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  auto h = std::make_unique<gfs_klt>();
  h->device = device;
  h->max_batch = max_batch;
  h->max_points = max_points;
  KltGeom& G = h->G;
  G.win = win;
  G.width = width;
  G.height = height;
  G.runs = (win + 3) / 4;
  G.ntasks = G.runs * win;
  G.rounds = (G.ntasks + 63) / 64;
  G.inv_runs = (65536 + G.runs - 1) / G.runs;
  G.nd = G.runs + 1;  // dwords staged per window row: bytes [0, 4 * runs + 4) cover the win + 1 bilinear taps
  G.stage_lpr = G.nd <= 16 ? 16 : 32;  // lanes per window row while staging
  G.stage_shift = G.nd <= 16 ? 4 : 5;
  G.stage_rows = 64 / G.stage_lpr;
  G.win_dwords = G.nd * (win + 1);
  int cw = width, ch = height;
  G.off[0] = 0;
  if (max_level > GFS_KLT_MAX_LEVELS - 1) max_level = GFS_KLT_MAX_LEVELS - 1;
  for (int l = 0; l <= max_level; l++) {  // buildOpticalFlowPyramid: the next level only while it is larger than the window
    G.lw[l] = cw;
    G.lh[l] = ch;
    G.off[l + 1] = G.off[l] + (long long)(cw + 2 * win) * (ch + 2 * win);
    G.n_levels = l + 1;
    cw = (cw + 1) / 2;
    ch = (ch + 1) / 2;
    if (cw <= win || ch <= win) break;
  }
  G.frame_stride = (long long)gfs::align_up((size_t)G.off[G.n_levels] + kKltSlack, 256);
  h->lds_bytes = (size_t)kKltWavesPerBlock * (3 * G.rounds * 64 + (G.win_dwords + 1) / 2) * sizeof(uint2);
  // (the ceiling is one global setting per kernel: always the hardware limit, so that trackers with different windows — on
  // different host threads or created one after the other — cannot undercut each other)
  GFS_REQUIRE(h->lds_bytes <= 160 * 1024 - 2048, GFS_ERR_UNSUPPORTED, "gfs_klt: window %d needs %zu bytes of LDS", win, h->lds_bytes);
#define KLT_ATTR(R)                                                                                                            \
  GFS_HIP(hipFuncSetAttribute((const void*)k_klt_track<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048)); \
  GFS_HIP(hipFuncSetAttribute((const void*)k_klt_fb<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048))
  KLT_DISPATCH(G.rounds, KLT_ATTR)
#undef KLT_ATTR
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  const size_t NP = (size_t)max_batch * max_points, NI = (size_t)max_batch * width * height;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_images.alloc(NI));
  A(h->h_images.alloc(NI));
  A(h->d_a.alloc(NP));
  A(h->d_b.alloc(NP));
  A(h->d_status.alloc(NP));
  A(h->d_err.alloc(NP));
  A(h->d_n.alloc(max_batch));
  A(h->d_good.alloc(max_batch));
  A(h->h_a.alloc(NP));
  A(h->h_b.alloc(NP));
  A(h->h_status.alloc(NP));
  A(h->h_err.alloc(NP));
  A(h->h_n.alloc(max_batch));
  A(h->h_good.alloc(max_batch));
#undef A
  if (rc) {
    (void)hipStreamDestroy(h->stream);
    return rc;
  }
  *out = h.release();
  return GFS_OK;
}

void gfs_klt_destroy(gfs_klt* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  delete h;
}

int gfs_klt_layout(const gfs_klt* h, int32_t* lw, int32_t* lh, int64_t* off) {
  GFS_REQUIRE(h, GFS_ERR_INVALID_ARG, "gfs_klt_layout: NULL handle");
  for (int l = 0; l < h->G.n_levels; l++) {
    if (lw) lw[l] = h->G.lw[l];
    if (lh) lh[l] = h->G.lh[l];
  }
  if (off)
    for (int l = 0; l <= h->G.n_levels; l++) off[l] = h->G.off[l];
  return h->G.n_levels;
}

int gfs_klt_pyramid_create(gfs_klt* h, gfs_klt_pyramid** out) {
  GFS_REQUIRE(h && out, GFS_ERR_INVALID_ARG, "gfs_klt_pyramid_create: invalid argument");
  *out = nullptr;
  GFS_HIP(hipSetDevice(h->device));
  auto p = std::make_unique<gfs_klt_pyramid>();
  p->owner = h;
  const size_t n = (size_t)h->G.frame_stride * h->max_batch;
  int rc = p->img.alloc(n);
  if (!rc) rc = p->deriv.alloc(n);
  if (rc) return rc;
  GFS_HIP(hipMemset(p->img.p, 0, n));
  GFS_HIP(hipMemset(p->deriv.p, 0, n * sizeof(short2)));
  *out = p.release();
  return GFS_OK;
}

void gfs_klt_pyramid_destroy(gfs_klt_pyramid* p) {
  if (!p) return;
  if (p->owner) {
    (void)hipSetDevice(p->owner->device);
    (void)hipStreamSynchronize(p->owner->stream);
  }
  delete p;
}

int gfs_klt_build_pyramid(gfs_klt* h, gfs_klt_pyramid* pyr, const uint8_t* const* images, int stride, int B) {
  GFS_REQUIRE(h && pyr && images && B > 0, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid: invalid argument");
  GFS_REQUIRE(pyr->owner == h, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid: the pyramid belongs to another tracker");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_klt_build_pyramid: batch %d exceeds capacity %d", B, h->max_batch);
  GFS_REQUIRE(stride >= h->G.width, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid: stride %d < width %d", stride, h->G.width);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int W = h->G.width, H = h->G.height;
  for (int f = 0; f < B; f++) {
    GFS_REQUIRE(images[f], GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid: image %d is NULL", f);
    for (int y = 0; y < H; y++) memcpy(h->h_images.p + ((size_t)f * H + y) * W, images[f] + (size_t)y * stride, W);
  }
  GFS_HIP(hipMemcpyAsync(h->d_images.p, h->h_images.p, (size_t)B * W * H, hipMemcpyHostToDevice, h->stream));
  const int rc = klt_build(h, pyr, h->d_images.p, W, B, h->stream);
  if (rc) return rc;
  GFS_HIP(hipStreamSynchronize(h->stream));
  return GFS_OK;
}

int gfs_klt_build_pyramid_device(gfs_klt* h, gfs_klt_pyramid* pyr, const void* dev_images, int stride, int B, void* stream) {
  GFS_REQUIRE(h && pyr && dev_images && B > 0, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid_device: invalid argument");
  GFS_REQUIRE(pyr->owner == h, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid_device: the pyramid belongs to another tracker");
  GFS_REQUIRE(B <= h->max_batch, GFS_ERR_CAPACITY, "gfs_klt_build_pyramid_device: batch %d exceeds capacity %d", B, h->max_batch);
  GFS_REQUIRE(stride >= h->G.width, GFS_ERR_INVALID_ARG, "gfs_klt_build_pyramid_device: stride %d < width %d", stride, h->G.width);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const int rc = klt_build(h, pyr, (const uint8_t*)dev_images, stride, B, s);
  if (rc) return rc;
  if (!stream) GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

int gfs_klt_pyramid_download(gfs_klt* h, const gfs_klt_pyramid* pyr, int f, uint8_t* img, int16_t* deriv) {
  GFS_REQUIRE(h && pyr && pyr->owner == h && (img || deriv), GFS_ERR_INVALID_ARG, "gfs_klt_pyramid_download: invalid argument");
  GFS_REQUIRE(f >= 0 && f < pyr->n_frames, GFS_ERR_INVALID_ARG, "gfs_klt_pyramid_download: frame %d of %d", f, pyr->n_frames);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  GFS_HIP(hipStreamSynchronize(h->stream));
  const size_t n = (size_t)h->G.off[h->G.n_levels], o = (size_t)f * h->G.frame_stride;
  if (img) GFS_HIP(hipMemcpy(img, pyr->img.p + o, n, hipMemcpyDeviceToHost));
  if (deriv) GFS_HIP(hipMemcpy(deriv, pyr->deriv.p + o, n * sizeof(short2), hipMemcpyDeviceToHost));
  return GFS_OK;
}

int gfs_klt_track(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* next, int B, const int32_t* n_points,
                  const float* const* prev_pts, float* const* next_pts, uint8_t* const* status, float* const* err, int max_level,
                  int max_iter, double eps, int flags, double min_eig_thr) {
  GFS_REQUIRE(h && prev && next && n_points && prev_pts && next_pts && status && err && B > 0, GFS_ERR_INVALID_ARG,
              "gfs_klt_track: invalid argument");
  GFS_REQUIRE(prev->owner == h && next->owner == h, GFS_ERR_INVALID_ARG, "gfs_klt_track: pyramid of another tracker");
  GFS_REQUIRE(B <= h->max_batch && B <= prev->n_frames && B <= next->n_frames, GFS_ERR_CAPACITY,
              "gfs_klt_track: batch %d exceeds the capacity %d or the frames held (%d / %d)", B, h->max_batch, prev->n_frames,
              next->n_frames);
  GFS_REQUIRE((flags & ~(GFS_KLT_USE_INITIAL_FLOW | GFS_KLT_GET_MIN_EIGENVALS)) == 0, GFS_ERR_INVALID_ARG,
              "gfs_klt_track: unknown flag bits 0x%x", flags);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int S = h->max_points;
  int nmax = 0;
  for (int f = 0; f < B; f++) {
    const int n = n_points[f];
    GFS_REQUIRE(n >= 0 && n <= S, GFS_ERR_CAPACITY, "gfs_klt_track: pair %d has %d points, capacity %d", f, n, S);
    GFS_REQUIRE(n == 0 || (prev_pts[f] && next_pts[f] && status[f] && err[f]), GFS_ERR_INVALID_ARG, "gfs_klt_track: pair %d has NULL arrays", f);
    h->h_n.p[f] = n;
    if (n) memcpy(h->h_a.p + (size_t)f * S, prev_pts[f], (size_t)n * sizeof(float2));
    if (n && (flags & GFS_KLT_USE_INITIAL_FLOW)) memcpy(h->h_b.p + (size_t)f * S, next_pts[f], (size_t)n * sizeof(float2));
    nmax = n > nmax ? n : nmax;
  }
  if (nmax == 0) return GFS_OK;
  KltParams P;
  P.max_level = max_level < 0 ? 0 : (max_level > h->G.n_levels - 1 ? h->G.n_levels - 1 : max_level);
  P.max_iter = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);
  const double e = eps < 0 ? 0 : (eps > 10 ? 10 : eps);
  P.eps2 = e * e;
  P.flags = flags;
  P.min_eig_thr = min_eig_thr;
  hipStream_t s = h->stream;
  const size_t NP = (size_t)B * S;
  GFS_HIP(hipMemcpyAsync(h->d_n.p, h->h_n.p, B * sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_a.p, h->h_a.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
  if (flags & GFS_KLT_USE_INITIAL_FLOW) GFS_HIP(hipMemcpyAsync(h->d_b.p, h->h_b.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
#define KLT_RUN(R) GFS_LAUNCH("k_klt_track", k_klt_track<R>, dim3(gfs::div_up(nmax, kKltWavesPerBlock) * B), dim3(kKltThreads), h->lds_bytes, s, h->G, P,  \
             (const uint8_t*)prev->img.p, (const short2*)prev->deriv.p, (const uint8_t*)next->img.p, (const int*)h->d_n.p, S,  \
             (const float2*)h->d_a.p, h->d_b.p, h->d_status.p, h->d_err.p, gfs::div_up(nmax, kKltWavesPerBlock), B)
  KLT_DISPATCH(h->G.rounds, KLT_RUN)
#undef KLT_RUN
  GFS_HIP(hipMemcpyAsync(h->h_b.p, h->d_b.p, NP * sizeof(float2), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_status.p, h->d_status.p, NP, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_err.p, h->d_err.p, NP * sizeof(float), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < B; f++) {
    const int n = n_points[f];
    if (!n) continue;
    memcpy(next_pts[f], h->h_b.p + (size_t)f * S, (size_t)n * sizeof(float2));
    memcpy(status[f], h->h_status.p + (size_t)f * S, (size_t)n);
    memcpy(err[f], h->h_err.p + (size_t)f * S, (size_t)n * sizeof(float));
  }
  return GFS_OK;
}

int gfs_klt_fb_track(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* cur, int B, const int32_t* n_points,
                     const float* const* kps, float* const* priors, uint8_t* const* kpstatus, int32_t* n_good, int nbpyrlvl,
                     float ferr, float fmax_fbklt_dist) {
  GFS_REQUIRE(h && prev && cur && n_points && kps && priors && kpstatus && n_good && B > 0, GFS_ERR_INVALID_ARG,
              "gfs_klt_fb_track: invalid argument");
  GFS_REQUIRE(prev->owner == h && cur->owner == h, GFS_ERR_INVALID_ARG, "gfs_klt_fb_track: pyramid of another tracker");
  GFS_REQUIRE(B <= h->max_batch && B <= prev->n_frames && B <= cur->n_frames, GFS_ERR_CAPACITY,
              "gfs_klt_fb_track: batch %d exceeds the capacity %d or the frames held (%d / %d)", B, h->max_batch, prev->n_frames,
              cur->n_frames);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  const int S = h->max_points;
  int nmax = 0;
  for (int f = 0; f < B; f++) {
    const int n = n_points[f];
    GFS_REQUIRE(n >= 0 && n <= S, GFS_ERR_CAPACITY, "gfs_klt_fb_track: pair %d has %d points, capacity %d", f, n, S);
    GFS_REQUIRE(n == 0 || (kps[f] && priors[f] && kpstatus[f]), GFS_ERR_INVALID_ARG, "gfs_klt_fb_track: pair %d has NULL arrays", f);
    h->h_n.p[f] = n;
    n_good[f] = 0;
    if (n) {
      memcpy(h->h_a.p + (size_t)f * S, kps[f], (size_t)n * sizeof(float2));
      memcpy(h->h_b.p + (size_t)f * S, priors[f], (size_t)n * sizeof(float2));
    }
    nmax = n > nmax ? n : nmax;
  }
  if (nmax == 0) return GFS_OK;
  const KltParams P = fb_params(h, nbpyrlvl);
  hipStream_t s = h->stream;
  const size_t NP = (size_t)B * S;
  GFS_HIP(hipMemcpyAsync(h->d_n.p, h->h_n.p, B * sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_a.p, h->h_a.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_b.p, h->h_b.p, NP * sizeof(float2), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemsetAsync(h->d_good.p, 0, B * sizeof(int), s));
#define KLT_RUN(R) GFS_LAUNCH("k_klt_fb", k_klt_fb<R>, dim3(gfs::div_up(nmax, kKltWavesPerBlock) * B), dim3(kKltThreads), h->lds_bytes, s, h->G, P,  \
             (const uint8_t*)prev->img.p, (const short2*)prev->deriv.p, (const uint8_t*)cur->img.p, (const short2*)cur->deriv.p,  \
             (const int*)h->d_n.p, S, (const float2*)h->d_a.p, h->d_b.p, h->d_status.p, h->d_good.p, ferr, fmax_fbklt_dist,  \
             gfs::div_up(nmax, kKltWavesPerBlock), B)
  KLT_DISPATCH(h->G.rounds, KLT_RUN)
#undef KLT_RUN
  GFS_HIP(hipMemcpyAsync(h->h_b.p, h->d_b.p, NP * sizeof(float2), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_status.p, h->d_status.p, NP, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipMemcpyAsync(h->h_good.p, h->d_good.p, B * sizeof(int), hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  for (int f = 0; f < B; f++) {
    const int n = n_points[f];
    n_good[f] = n ? h->h_good.p[f] : 0;
    if (!n) continue;
    memcpy(priors[f], h->h_b.p + (size_t)f * S, (size_t)n * sizeof(float2));
    memcpy(kpstatus[f], h->h_status.p + (size_t)f * S, (size_t)n);
  }
  return GFS_OK;
}

int gfs_klt_fb_track_device(gfs_klt* h, const gfs_klt_pyramid* prev, const gfs_klt_pyramid* cur, int B, int pt_stride,
                            const void* dev_n, const void* dev_kps, void* dev_priors, void* dev_kpstatus, void* dev_n_good,
                            int nbpyrlvl, float ferr, float fmax_fbklt_dist, void* stream) {
  GFS_REQUIRE(h && prev && cur && dev_n && dev_kps && dev_priors && dev_kpstatus && dev_n_good && B > 0 && pt_stride > 0,
              GFS_ERR_INVALID_ARG, "gfs_klt_fb_track_device: invalid argument");
  GFS_REQUIRE(prev->owner == h && cur->owner == h, GFS_ERR_INVALID_ARG, "gfs_klt_fb_track_device: pyramid of another tracker");
  GFS_REQUIRE(B <= h->max_batch && B <= prev->n_frames && B <= cur->n_frames, GFS_ERR_CAPACITY,
              "gfs_klt_fb_track_device: batch %d exceeds the capacity %d or the frames held (%d / %d)", B, h->max_batch,
              prev->n_frames, cur->n_frames);
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const KltParams P = fb_params(h, nbpyrlvl);
  GFS_HIP(hipMemsetAsync(dev_n_good, 0, B * sizeof(int), s));
#define KLT_RUN(R) GFS_LAUNCH("k_klt_fb", k_klt_fb<R>, dim3(gfs::div_up(pt_stride, kKltWavesPerBlock) * B), dim3(kKltThreads), h->lds_bytes, s, h->G, P,  \
             (const uint8_t*)prev->img.p, (const short2*)prev->deriv.p, (const uint8_t*)cur->img.p, (const short2*)cur->deriv.p,  \
             (const int*)dev_n, pt_stride, (const float2*)dev_kps, (float2*)dev_priors, (uint8_t*)dev_kpstatus, (int*)dev_n_good,  \
             ferr, fmax_fbklt_dist, gfs::div_up(pt_stride, kKltWavesPerBlock), B)
  KLT_DISPATCH(h->G.rounds, KLT_RUN)
#undef KLT_RUN
  if (!stream) GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

int gfs_klt_compact_tracks_device(gfs_klt* h, int B, int pt_stride, const void* dev_n, const void* dev_kps, const void* dev_priors,
                                  const void* dev_kpstatus, void* dev_out_a, void* dev_out_b, void* dev_out_index, void* dev_out_n,
                                  void* stream) {
  GFS_REQUIRE(h && dev_n && dev_kps && dev_priors && dev_kpstatus && dev_out_a && dev_out_b && dev_out_index && dev_out_n && B > 0 &&
                  pt_stride > 0, GFS_ERR_INVALID_ARG, "gfs_klt_compact_tracks_device: invalid argument");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  GFS_LAUNCH("k_klt_compact", k_klt_compact, dim3(B), dim3(256), 0, s, (const int*)dev_n, pt_stride, (const float2*)dev_kps,
             (const float2*)dev_priors, (const uint8_t*)dev_kpstatus, (float2*)dev_out_a, (float2*)dev_out_b, (int*)dev_out_index,
             (int*)dev_out_n);
  if (!stream) GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

int gfs_klt_apply_mask_device(gfs_klt* h, int B, int pt_stride, const void* dev_m, const void* dev_index, const void* dev_mask,
                              void* dev_kpstatus, void* stream) {
  GFS_REQUIRE(h && dev_m && dev_index && dev_mask && dev_kpstatus && B > 0 && pt_stride > 0, GFS_ERR_INVALID_ARG,
              "gfs_klt_apply_mask_device: invalid argument");
  std::lock_guard<std::mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  GFS_LAUNCH("k_klt_apply_mask", k_klt_apply_mask, dim3(gfs::div_up(pt_stride, 256), B), dim3(256), 0, s, (const int*)dev_m, pt_stride,
             (const int*)dev_index, (const uint8_t*)dev_mask, (uint8_t*)dev_kpstatus);
  if (!stream) GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

}  // extern "C"
