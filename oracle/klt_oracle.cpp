// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of the optical-flow front end of the reference ("GeoFlow"):
//   * cv::buildOpticalFlowPyramid(image, mImGray, Size(w, w), 3)            (call sites src/Frame.cc:373, 505, 1415),
//   * cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, ...) on such pyramids        (src/ORBmatcher.cc:2224, 2271; src/Tracking.cc:3298, 3342),
//   * ORBmatcher::fbKltTracking / Tracking::fbKltTracking — forward track, gates, backward track, forward-backward distance
//     (src/ORBmatcher.cc:2186-2297, src/Tracking.cc:3262-3366; ORBmatcher::inBorder src/ORBmatcher.cc:2552-2557).
// The two cv:: functions live in OpenCV (video/src/lkpyramid.cpp, imgproc/src/pyramids.cpp), which is NOT in the container: their
// algorithm is restated from the published OpenCV 4.x sources from memory (pin: 4.5.4, as SURVEY.md App. A recommends):
//   pyramid  : level 0 = the image, level l = pyrDown(level l-1) = 5x5 [1 4 6 4 1]^2 / 256 with (sum + 128) >> 8, BORDER_REFLECT_101 on
//              the level itself, size ((w+1)/2, (h+1)/2); a level is only built while both of its sides exceed the window; every level is
//              stored with a border of `win` pixels filled by BORDER_REFLECT_101; derivative images (short2: Scharr dx, dy; 3-10-3
//              smoothing, reflect-101 inside the level) stored with a zero border (BORDER_CONSTANT).
//   tracker  : LKTrackerInvoker — W_BITS = 14 fixed-point bilinear weights (cvRound), patch I << 5 and derivatives as int16, the 2x2
//              gradient matrix, minEig gate (1e-4), <= maxCount Newton steps with the `delta . delta <= eps^2` and the
//              "oscillation" early exits, status cleared only at level 0, err = minEig (OPTFLOW_LK_GET_MIN_EIGENVALS) or the
//              L1 patch residual / (32 * win^2).
//
// ONE DELIBERATE DIFFERENCE, applied identically by the HIP path: OpenCV accumulates A11/A12/A22 and b1/b2 in float, in an order
// that depends on how the library was built (scalar, SSE2, AVX2 with/without FMA all differ in the last bits).  All summands are
// integers, so this restatement sums them EXACTLY (int64) and rounds once to float — the value every OpenCV build approximates
// within its own rounding error.  It makes the result independent of summation order (so the GPU can be bit-exact against it)
// at the price of ulp-level differences to any particular OpenCV binary; see DESIGN.md §2 (deviations) and §8.
// So that the size of that difference is MEASURED and not asserted, gfso_klt_set_accumulation() switches the tracker to
//   1: OpenCV 4.5.4's own scalar loop (the generic C++ path of LKTrackerInvoker, lkpyramid.cpp: `acctype iA11 = 0 ...
//      iA11 += (itemtype)(ixval*ixval)` with acctype = itemtype = float, raster order; `ib1 += (itemtype)(diff*dIptr[0])`), or
//   2: a four-lane model of the universal-intrinsics builds (lane k of a float32x4 accumulator takes the pixels x = 4i + k of
//      every row, the row tail goes to the scalar accumulator, the lanes are combined as (l0 + l2) + (l1 + l3) at the end; for
//      the mismatch vector two neighbouring integer products are added exactly before the conversion, as v_dotprod does).  A
//      model of how vectorised binaries regroup the sum, not a transcription of one.
// tests/test_klt_accumulation.py tracks the same points in all three modes and bounds the differences.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "gfs_oracle.h"

namespace {
inline int reflect101(int p, int len) {  // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}
inline int cv_round(float v) { return (int)std::lrintf(v); }  // cvRound: round half to even (default FP environment)
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }  // CV_DESCALE

int g_accum_mode = 0;  // 0 exact integer sums (what the HIP path implements), 1 OpenCV's scalar float loop, 2 four-lane float model

// One of the tracker's five sums (A11, A12, A22, b1, b2).  add(v, x, vec_end): the integer summand of pixel x of a row whose
// vectorised part ends at vec_end (mode 2).
struct Acc {
  int64_t s = 0;
  float f = 0.f, lane[4] = {0.f, 0.f, 0.f, 0.f};
  inline void add(int v, int x, int vec_end) {
    if (g_accum_mode == 0) s += v;
    else if (g_accum_mode == 1 || x >= vec_end) f += (float)v;
    else lane[x & 3] += (float)v;
  }
  inline void add_pair(int v0, int v1, int x, int vec_end) {  // pixels x, x + 1 (x even) of the mismatch loop
    if (g_accum_mode == 2 && x + 1 < vec_end) lane[(x >> 1) & 3] += (float)(v0 + v1);
    else {
      add(v0, x, 0);
      add(v1, x + 1, 0);
    }
  }
  inline float total() const {
    if (g_accum_mode == 0) return (float)s;
    if (g_accum_mode == 1) return f;
    return f + ((lane[0] + lane[2]) + (lane[1] + lane[3]));
  }
};

struct Layout {
  int n = 0;
  int lw[GFSO_KLT_MAX_LEVELS], lh[GFSO_KLT_MAX_LEVELS];
  int64_t off[GFSO_KLT_MAX_LEVELS + 1];
};
Layout make_layout(int w, int h, int win, int max_level) {
  Layout L;
  int cw = w, ch = h;
  L.off[0] = 0;
  if (max_level > GFSO_KLT_MAX_LEVELS - 1) max_level = GFSO_KLT_MAX_LEVELS - 1;
  for (int level = 0; level <= max_level; level++) {
    L.lw[level] = cw;
    L.lh[level] = ch;
    L.off[level + 1] = L.off[level] + (int64_t)(cw + 2 * win) * (ch + 2 * win);
    L.n = level + 1;
    cw = (cw + 1) / 2;
    ch = (ch + 1) / 2;
    if (cw <= win || ch <= win) break;  // buildOpticalFlowPyramid: stop before a level that is not larger than the window
  }
  return L;
}

struct Criteria {
  int max_count;
  double eps2;
};
Criteria make_criteria(int max_iter, double eps) {  // SparsePyrLKOpticalFlowImpl::calc, COUNT + EPS given
  Criteria c;
  c.max_count = max_iter < 0 ? 0 : (max_iter > 100 ? 100 : max_iter);
  double e = eps < 0 ? 0 : (eps > 10 ? 10 : eps);
  c.eps2 = e * e;
  return c;
}

// LKTrackerInvoker::operator() for one point on one level.  pi/pd/pj: pointers to the INTERIOR origin of the padded level images.
struct LevelView {
  const uint8_t* I;
  const int16_t* dI;
  const uint8_t* J;
  int w, h, pitch;  // level size, padded pitch in pixels (I, J: bytes; dI: short2 elements)
};

void track_level(const LevelView& V, int win, int level, int max_level, int flags, const Criteria& crit, double min_eig_thr,
                 const float* prev_pt_full, float* next_pt /* in/out, this level's scale on exit */, uint8_t* status, float* err,
                 std::vector<int16_t>& buf) {
  const float half = (win - 1) * 0.5f;
  float px = prev_pt_full[0] * (float)(1. / (1 << level)), py = prev_pt_full[1] * (float)(1. / (1 << level));
  float nx, ny;
  if (level == max_level) {
    if (flags & GFSO_KLT_USE_INITIAL_FLOW) {
      nx = next_pt[0] * (float)(1. / (1 << level));
      ny = next_pt[1] * (float)(1. / (1 << level));
    } else {
      nx = px;
      ny = py;
    }
  } else {
    nx = next_pt[0] * 2.f;
    ny = next_pt[1] * 2.f;
  }
  next_pt[0] = nx;
  next_pt[1] = ny;

  px -= half;
  py -= half;
  const int ipx = cv_floor(px), ipy = cv_floor(py);
  if (ipx < -win || ipx >= V.w || ipy < -win || ipy >= V.h) {
    if (level == 0) {
      *status = 0;
      *err = 0;
    }
    return;
  }
  float a = px - ipx, b = py - ipy;
  const int W_BITS = 14, W_BITS1 = 14;
  const float FLT_SCALE = 1.f / (1 << 20);
  int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
  int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
  int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
  int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

  int16_t* Iw = buf.data();
  int16_t* dIw = buf.data() + (size_t)win * win;
  Acc sA11, sA12, sA22;
  const int vec4 = win & ~3, vec8 = win & ~7;
  for (int y = 0; y < win; y++) {
    const uint8_t* src = V.I + (int64_t)(y + ipy) * V.pitch + ipx;
    const int16_t* dsrc = V.dI + ((int64_t)(y + ipy) * V.pitch + ipx) * 2;
    const int dstep = V.pitch * 2;
    for (int x = 0; x < win; x++, dsrc += 2) {
      int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + V.pitch] * iw10 + src[x + V.pitch + 1] * iw11, W_BITS1 - 5);
      int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS1);
      int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11, W_BITS1);
      Iw[y * win + x] = (int16_t)ival;
      dIw[(y * win + x) * 2] = (int16_t)ixval;
      dIw[(y * win + x) * 2 + 1] = (int16_t)iyval;
      sA11.add(ixval * ixval, x, vec4);
      sA12.add(ixval * iyval, x, vec4);
      sA22.add(iyval * iyval, x, vec4);
    }
  }
  float A11 = sA11.total() * FLT_SCALE, A12 = sA12.total() * FLT_SCALE, A22 = sA22.total() * FLT_SCALE;
  float D = A11 * A22 - A12 * A12;
  float min_eig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
  if (flags & GFSO_KLT_GET_MIN_EIGENVALS) *err = min_eig;
  if (min_eig < min_eig_thr || D < 1.1920928955078125e-07f /* FLT_EPSILON */) {
    if (level == 0) *status = 0;
    return;
  }
  D = 1.f / D;
  nx -= half;
  ny -= half;
  float pdx = 0.f, pdy = 0.f;
  for (int j = 0; j < crit.max_count; j++) {
    const int inx = cv_floor(nx), iny = cv_floor(ny);
    if (inx < -win || inx >= V.w || iny < -win || iny >= V.h) {
      if (level == 0) *status = 0;
      break;
    }
    a = nx - inx;
    b = ny - iny;
    iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    Acc sb1, sb2;
    for (int y = 0; y < win; y++) {
      const uint8_t* Jp = V.J + (int64_t)(y + iny) * V.pitch + inx;
      int p1 = 0, p2 = 0;
      for (int x = 0; x < win; x++) {
        int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + V.pitch] * iw10 + Jp[x + V.pitch + 1] * iw11, W_BITS1 - 5) -
                   Iw[y * win + x];
        const int q1 = diff * dIw[(y * win + x) * 2], q2 = diff * dIw[(y * win + x) * 2 + 1];
        if (g_accum_mode != 2) {
          sb1.add(q1, x, 0);
          sb2.add(q2, x, 0);
        } else if (x >= vec8) {
          sb1.add(q1, x, 0);
          sb2.add(q2, x, 0);
        } else if (x & 1) {
          sb1.add_pair(p1, q1, x - 1, vec8);
          sb2.add_pair(p2, q2, x - 1, vec8);
        } else {
          p1 = q1;
          p2 = q2;
        }
      }
    }
    float b1 = sb1.total() * FLT_SCALE, b2 = sb2.total() * FLT_SCALE;
    float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
    nx += dx;
    ny += dy;
    next_pt[0] = nx + half;
    next_pt[1] = ny + half;
    if ((double)dx * dx + (double)dy * dy <= crit.eps2) break;
    if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
      next_pt[0] -= dx * 0.5f;
      next_pt[1] -= dy * 0.5f;
      break;
    }
    pdx = dx;
    pdy = dy;
  }
  if (*status && level == 0 && !(flags & GFSO_KLT_GET_MIN_EIGENVALS)) {
    float ex = next_pt[0] - half, ey = next_pt[1] - half;
    const int iex = cv_floor(ex), iey = cv_floor(ey);
    if (iex < -win || iex >= V.w || iey < -win || iey >= V.h) {
      *status = 0;
      return;
    }
    float aa = ex - iex, bb = ey - iey;
    iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
    iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
    iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
    iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    int64_t e = 0;  // sum |diff| < 2^24 for win <= 45, so the float accumulation of the original is exact as well
    for (int y = 0; y < win; y++) {
      const uint8_t* Jp = V.J + (int64_t)(y + iey) * V.pitch + iex;
      for (int x = 0; x < win; x++) {
        int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + V.pitch] * iw10 + Jp[x + V.pitch + 1] * iw11, W_BITS1 - 5) -
                   Iw[y * win + x];
        e += diff < 0 ? -diff : diff;
      }
    }
    *err = (float)e * 1.f / (32 * win * win);
  }
}
}  // namespace

extern "C" void gfso_klt_set_accumulation(int mode) { g_accum_mode = mode < 0 || mode > 2 ? 0 : mode; }

extern "C" int gfso_klt_layout(int w, int h, int win, int max_level, int32_t* lw, int32_t* lh, int64_t* off) {
  Layout L = make_layout(w, h, win, max_level);
  for (int l = 0; l < L.n; l++) {
    if (lw) lw[l] = L.lw[l];
    if (lh) lh[l] = L.lh[l];
  }
  if (off)
    for (int l = 0; l <= L.n; l++) off[l] = L.off[l];
  return L.n;
}

extern "C" int gfso_klt_build_pyramid(const uint8_t* img, int w, int h, int stride, int win, int max_level, uint8_t* pyr_img,
                                      int16_t* pyr_deriv) {
  const Layout L = make_layout(w, h, win, max_level);
  std::vector<uint8_t> cur((size_t)w * h), nxt;
  for (int y = 0; y < h; y++) std::memcpy(&cur[(size_t)y * w], img + (size_t)y * stride, w);
  for (int l = 0; l < L.n; l++) {
    const int cw = L.lw[l], ch = L.lh[l], pw = cw + 2 * win, ph = ch + 2 * win;
    uint8_t* P = pyr_img + L.off[l];
    for (int y = 0; y < ph; y++) {  // copyMakeBorder(BORDER_REFLECT_101 | BORDER_ISOLATED)
      const int sy = reflect101(y - win, ch);
      for (int x = 0; x < pw; x++) P[(size_t)y * pw + x] = cur[(size_t)sy * cw + reflect101(x - win, cw)];
    }
    int16_t* Dv = pyr_deriv + L.off[l] * 2;
    std::memset(Dv, 0, (size_t)pw * ph * 2 * sizeof(int16_t));  // BORDER_CONSTANT
    for (int y = 0; y < ch; y++) {                                // calcSharrDeriv
      const int y0 = reflect101(y - 1, ch), y2 = reflect101(y + 1, ch);
      for (int x = 0; x < cw; x++) {
        const int xm = reflect101(x - 1, cw), xp = reflect101(x + 1, cw);
        auto t0 = [&](int xx) { return (cur[(size_t)y0 * cw + xx] + cur[(size_t)y2 * cw + xx]) * 3 + cur[(size_t)y * cw + xx] * 10; };
        auto t1 = [&](int xx) { return cur[(size_t)y2 * cw + xx] - cur[(size_t)y0 * cw + xx]; };
        int16_t* d = Dv + ((size_t)(y + win) * pw + x + win) * 2;
        d[0] = (int16_t)(t0(xp) - t0(xm));
        d[1] = (int16_t)((t1(xp) + t1(xm)) * 3 + t1(x) * 10);
      }
    }
    if (l + 1 < L.n) {  // pyrDown, 8U: FixPtCast<uchar, 8>
      const int nw = L.lw[l + 1], nh = L.lh[l + 1];
      nxt.assign((size_t)nw * nh, 0);
      for (int y = 0; y < nh; y++)
        for (int x = 0; x < nw; x++) {
          static const int k[5] = {1, 4, 6, 4, 1};
          int s = 0;
          for (int r = 0; r < 5; r++) {
            const uint8_t* row = &cur[(size_t)reflect101(2 * y + r - 2, ch) * cw];
            int hs = 0;
            for (int c = 0; c < 5; c++) hs += k[c] * row[reflect101(2 * x + c - 2, cw)];
            s += k[r] * hs;
          }
          nxt[(size_t)y * nw + x] = (uint8_t)((s + 128) >> 8);
        }
      cur.swap(nxt);
    }
  }
  return L.n;
}

extern "C" int gfso_klt_track(const uint8_t* prev_img, const int16_t* prev_deriv, const uint8_t* next_img, int w, int h, int win,
                              int pyr_max_level, int max_level, int n, const float* prev_pts, float* next_pts, uint8_t* status,
                              float* err, int max_iter, double eps, int flags, double min_eig_thr) {
  const Layout L = make_layout(w, h, win, pyr_max_level);
  if (max_level > L.n - 1) max_level = L.n - 1;  // calcOpticalFlowPyrLK: maxLevel clamped to what the pyramids hold
  if (max_level < 0) return -1;
  const Criteria crit = make_criteria(max_iter, eps);
  std::vector<int16_t> buf((size_t)win * win * 3);
  for (int i = 0; i < n; i++) {
    status[i] = 1;
    err[i] = 0;
    if (!(flags & GFSO_KLT_USE_INITIAL_FLOW)) next_pts[2 * i] = next_pts[2 * i + 1] = 0;
  }
  for (int level = max_level; level >= 0; level--) {
    const int pw = L.lw[level] + 2 * win;
    LevelView V;
    V.w = L.lw[level];
    V.h = L.lh[level];
    V.pitch = pw;
    const int64_t org = L.off[level] + (int64_t)win * pw + win;
    V.I = prev_img + org;
    V.dI = prev_deriv + org * 2;
    V.J = next_img + org;
    for (int i = 0; i < n; i++)
      track_level(V, win, level, max_level, flags, crit, min_eig_thr, prev_pts + 2 * i, next_pts + 2 * i, status + i, err + i, buf);
  }
  return 0;
}

extern "C" int gfso_fb_klt_tracking(const uint8_t* prev_img, const int16_t* prev_deriv, const uint8_t* cur_img,
                                    const int16_t* cur_deriv, int w, int h, int win, int pyr_max_level, int nbpyrlvl, float ferr,
                                    float fmax_fbklt_dist, int n, const float* kps, float* priors, uint8_t* kpstatus) {
  if (n == 0) return 0;  // src/ORBmatcher.cc:2197-2199
  const int flags = GFSO_KLT_USE_INITIAL_FLOW | GFSO_KLT_GET_MIN_EIGENVALS;  // :2227
  std::vector<uint8_t> st(n);
  std::vector<float> er(n);
  gfso_klt_track(prev_img, prev_deriv, cur_img, w, h, win, pyr_max_level, nbpyrlvl, n, kps, priors, st.data(), er.data(), 30,
                 (double)0.01f, flags, 1e-4);  // :2217-2227
  std::vector<float> newk, backk;
  std::vector<int> idx;
  for (int i = 0; i < n; i++) {  // :2236-2257
    kpstatus[i] = 0;
    if (!st[i]) continue;
    if (er[i] > ferr) continue;
    const float x = priors[2 * i], y = priors[2 * i + 1];
    if (!(1.f <= x && x < w - 1.f && 1.f <= y && y < h - 1.f)) continue;  // inBorder :2552-2557
    newk.push_back(x);
    newk.push_back(y);
    backk.push_back(kps[2 * i]);
    backk.push_back(kps[2 * i + 1]);
    kpstatus[i] = 1;
    idx.push_back(i);
  }
  if (idx.empty()) return 0;  // :2259-2261
  const int m = (int)idx.size();
  st.assign(m, 0);
  er.assign(m, 0);
  gfso_klt_track(cur_img, cur_deriv, prev_img, w, h, win, pyr_max_level, 0, m, newk.data(), backk.data(), st.data(), er.data(), 30,
                 (double)0.01f, flags, 1e-4);  // :2271-2274
  int good = 0;
  for (int k = 0; k < m; k++) {  // :2276-2291
    const int i = idx[k];
    if (!st[k]) {
      kpstatus[i] = 0;
      continue;
    }
    const float dx = kps[2 * i] - backk[2 * k], dy = kps[2 * i + 1] - backk[2 * k + 1];
    if (std::sqrt((double)dx * dx + (double)dy * dy) > fmax_fbklt_dist) {  // cv::norm(Point2f) computes in double
      kpstatus[i] = 0;
      continue;
    }
    good++;
  }
  return good;
}
