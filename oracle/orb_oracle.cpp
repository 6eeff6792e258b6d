// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of ORB_SLAM3::ORBextractor as modified by GeoFlow-SLAM
// (reference: src/ORBextractor.cc, include/ORBextractor.h) and of the OpenCV primitives it calls.
// Every function cites the reference file:line (or the OpenCV 4.5.4 source file) it follows.
// Build: g++ -O3 -std=c++17 (x86-64 baseline: no FMA contraction, matching CMakeLists.txt:35-40).
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <list>
#include <utility>
#include <vector>

#include "gfs_oracle.h"

namespace {

// ---- OpenCV scalar helpers (core/fast_math.hpp): cvRound = round-half-even via SSE cvtsd2si ----
inline int cvRoundD(double v) { return (int)std::lrint(v); }
inline int cvRoundF(float v) { return (int)std::lrintf(v); }
inline int cvFloorD(double v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cvCeilD(double v) {
  int i = (int)v;
  return i + (i < v);
}
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

constexpr int PATCH_SIZE = 31;       // src/ORBextractor.cc:51
constexpr int HALF_PATCH_SIZE = 15;  // :52
constexpr int EDGE_THRESHOLD = 19;   // :53

const int kBriefPattern[256 * 4] = {
#include "brief_pattern.inc"
};

struct Plane {  // padded level image; interior() mirrors mvImagePyramid[level] (a ROI of `temp`)
  int rows = 0, cols = 0, stride = 0;
  std::vector<uint8_t> buf;
  uint8_t* interior() { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
  const uint8_t* interior() const { return buf.data() + EDGE_THRESHOLD * stride + EDGE_THRESHOLD; }
};

inline int reflect101(int p, int n) {  // cv::borderInterpolate(BORDER_REFLECT_101)
  if (n == 1) return 0;
  while (p < 0 || p >= n) {
    if (p < 0)
      p = -p;
    else
      p = 2 * (n - 1) - p;
  }
  return p;
}

// ------------------------------------------------------------------------------------------------
// cv::resize(..., INTER_AREA) for 8UC1, scale >= 1, non-integer scale.
// OpenCV 4.5.4 modules/imgproc/src/resize.cpp: computeResizeAreaTab + ResizeArea_Invoker<uchar,float>.
// Call site: src/ORBextractor.cc:1240-1241.
// ------------------------------------------------------------------------------------------------
struct DecimateAlpha {
  int si, di;
  float alpha;
};

int computeResizeAreaTab(int ssize, int dsize, double scale, std::vector<DecimateAlpha>& tab) {
  tab.clear();
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale;
    double fsx2 = fsx1 + scale;
    double cellWidth = std::min(scale, ssize - fsx1);
    int sx1 = cvCeilD(fsx1), sx2 = cvFloorD(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cellWidth)});
    for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, dx, float(1.0 / cellWidth)});
    if (fsx2 - sx2 > 1e-3)
      tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
  }
  return (int)tab.size();
}

void resize_area_u8(const uint8_t* src, int srows, int scols, int sstride, uint8_t* dst, int drows, int dcols,
                    int dstride) {
  // cv::resize: inv_scale = (double)dsize/ssize ; hal::resize: scale = 1./inv_scale
  double inv_scale_x = (double)dcols / scols, inv_scale_y = (double)drows / srows;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<DecimateAlpha> xtab, ytab;
  int xtab_size = computeResizeAreaTab(scols, dcols, scale_x, xtab);
  int ytab_size = computeResizeAreaTab(srows, drows, scale_y, ytab);
  std::vector<float> buf(dcols), sum(dcols, 0.f);
  int prev_dy = ytab[0].di;
  for (int j = 0; j < ytab_size; j++) {
    float beta = ytab[j].alpha;
    int dy = ytab[j].di, sy = ytab[j].si;
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dcols; dx++) buf[dx] = 0.f;
    for (int k = 0; k < xtab_size; k++) {
      int dxn = xtab[k].di;
      float alpha = xtab[k].alpha;
      buf[dxn] += S[xtab[k].si] * alpha;  // float mul then float add (no FMA on the x86-64 baseline)
    }
    if (dy != prev_dy) {
      uint8_t* D = dst + (size_t)prev_dy * dstride;
      for (int dx = 0; dx < dcols; dx++) {
        D[dx] = sat_u8(cvRoundF(sum[dx]));
        sum[dx] = beta * buf[dx];
      }
      prev_dy = dy;
    } else {
      for (int dx = 0; dx < dcols; dx++) sum[dx] += beta * buf[dx];
    }
  }
  uint8_t* D = dst + (size_t)prev_dy * dstride;
  for (int dx = 0; dx < dcols; dx++) D[dx] = sat_u8(cvRoundF(sum[dx]));
}

// ------------------------------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, nonmax) == FAST_t<16> + cornerScore<16>
// OpenCV 4.5.4 modules/features2d/src/fast.cpp, fast_score.cpp. Call sites src/ORBextractor.cc:809,826.
// ------------------------------------------------------------------------------------------------
struct FastKp {
  int x, y, score;
};

const int kRing16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int cornerScore16(const uint8_t* ptr, const int pixel[25], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

void fast9_16(const uint8_t* img, int rows, int cols, int stride, int threshold, bool nonmax,
              std::vector<FastKp>& out) {
  out.clear();
  const int K = 8, N = 16 + K + 1;
  int pixel[25];
  for (int k = 0; k < 16; k++) pixel[k] = kRing16[k][0] + kRing16[k][1] * stride;
  for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
  threshold = std::min(std::max(threshold, 0), 255);
  uint8_t threshold_tab[512];
  for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
  if (cols <= 0 || rows <= 0) return;
  std::vector<uint8_t> bufs(3 * (size_t)cols, 0);
  std::vector<int> cpbufs(3 * (size_t)(cols + 1), 0);
  uint8_t* buf[3] = {bufs.data(), bufs.data() + cols, bufs.data() + 2 * cols};
  int* cpbuf[3] = {cpbufs.data(), cpbufs.data() + cols + 1, cpbufs.data() + 2 * (cols + 1)};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3] + 1;
    std::memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (int k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      int j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                      score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
        out.push_back({j, i - 1, score});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// cv::fastAtan2(float y, float x): OpenCV 4.5.4 core/src/mathfuncs_core.simd.hpp atan_f32 (degrees).
// Call site: src/ORBextractor.cc:94.
// ------------------------------------------------------------------------------------------------
const float atan2_p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);

float fast_atan2(float y, float x) {
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + (float)2.2204460492503131e-16);
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ------------------------------------------------------------------------------------------------
// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on CV_8U: fixed-point path
// OpenCV 4.5.4 imgproc/src/smooth.dispatch.cpp (getGaussianKernelFixedPoint_ED) + smooth.simd.hpp
// (hlineSmoothONa_yzy_a / vlineSmoothONa_yzy_a, ufixedpoint16 Q8.8 -> ufixedpoint32 Q16.16,
// rounding (v + 32768) >> 16). Call site: src/ORBextractor.cc:1188-1189.
// ------------------------------------------------------------------------------------------------
const int kBlurTaps[2][7] = {{18, 34, 48, 56, 48, 34, 18}, {18, 34, 49, 55, 49, 34, 18}};

void gaussian_blur7(const uint8_t* src, int rows, int cols, int stride, uint8_t* dst, int dstride, int variant) {
  const int* k = kBlurTaps[variant ? 1 : 0];
  std::vector<uint16_t> tmp((size_t)rows * cols);
  for (int y = 0; y < rows; y++) {
    const uint8_t* S = src + (size_t)y * stride;
    for (int x = 0; x < cols; x++) {
      uint32_t acc = 0;
      for (int i = -3; i <= 3; i++) {
        uint32_t term = (uint32_t)k[i + 3] * S[reflect101(x + i, cols)];
        uint32_t t16 = term > 0xffff ? 0xffff : term;  // ufixedpoint16 * uint8 saturates
        acc += t16;
        if (acc > 0xffff) acc = 0xffff;  // ufixedpoint16 + saturates
      }
      tmp[(size_t)y * cols + x] = (uint16_t)acc;
    }
  }
  for (int y = 0; y < rows; y++) {
    for (int x = 0; x < cols; x++) {
      uint64_t acc = 0;
      for (int j = -3; j <= 3; j++) {
        acc += (uint64_t)k[j + 3] * tmp[(size_t)reflect101(y + j, rows) * cols + x];
        if (acc > 0xffffffffull) acc = 0xffffffffull;  // ufixedpoint32 + saturates
      }
      uint64_t r = (acc + 32768) >> 16;
      dst[(size_t)y * dstride + x] = (uint8_t)(r > 255 ? 255 : r);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ExtractorNode / DistributeOctTree  (src/ORBextractor.cc:502-768, include/ORBextractor.h:31-45)
// ------------------------------------------------------------------------------------------------
struct OKey {
  float x, y, response;
  int src;  // index into the input candidate array
};
struct P2i {
  int x, y;
};
struct ExtractorNode {
  std::vector<OKey> vKeys;
  P2i UL{0, 0}, UR{0, 0}, BL{0, 0}, BR{0, 0};
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) {  // :502-550
  const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
  n1.UL = UL;
  n1.UR = {UL.x + halfX, UL.y};
  n1.BL = {UL.x, UL.y + halfY};
  n1.BR = {UL.x + halfX, UL.y + halfY};
  n2.UL = n1.UR;
  n2.UR = UR;
  n2.BL = n1.BR;
  n2.BR = {UR.x, UL.y + halfY};
  n3.UL = n1.BL;
  n3.UR = n1.BR;
  n3.BL = BL;
  n3.BR = {n1.BR.x, BL.y};
  n4.UL = n3.UR;
  n4.UR = n2.BR;
  n4.BL = n3.BR;
  n4.BR = BR;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const OKey& kp = vKeys[i];
    if (kp.x < n1.UR.x) {
      if (kp.y < n1.BR.y)
        n1.vKeys.push_back(kp);
      else
        n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BR.y)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

bool compareNodes(std::pair<int, ExtractorNode*>& e1, std::pair<int, ExtractorNode*>& e2) {  // :552-565
  if (e1.first < e2.first) return true;
  if (e1.first > e2.first) return false;
  return e1.second->UL.x < e2.second->UL.x;
}

std::vector<OKey> DistributeOctTree(const std::vector<OKey>& vToDistributeKeys, int minX, int maxX, int minY,
                                    int maxY, int N) {  // :567-768
  std::vector<OKey> vResultKeys;
  if (vToDistributeKeys.empty()) return vResultKeys;
  int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni == 0) nIni = 1;
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<ExtractorNode> lNodes;
  std::vector<ExtractorNode*> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    ExtractorNode ni;
    ni.UL = {(int)(hX * static_cast<float>(i)), 0};
    ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = {ni.UL.x, maxY - minY};
    ni.BR = {ni.UR.x, maxY - minY};
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const OKey& kp = vToDistributeKeys[i];
    vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
  }
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) {
      lit->bNoMore = true;
      lit++;
    } else if (lit->vKeys.empty())
      lit = lNodes.erase(lit);
    else
      lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, ExtractorNode*>> vSizeAndPointerToNode;
  vSizeAndPointerToNode.reserve(lNodes.size() * 4);
  auto push_children = [&](ExtractorNode& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) {
        lit++;
        continue;
      } else {
        ExtractorNode n1, n2, n3, n4;
        lit->DivideNode(n1, n2, n3, n4);
        push_children(n1, &nToExpand);
        push_children(n2, &nToExpand);
        push_children(n3, &nToExpand);
        push_children(n4, &nToExpand);
        lit = lNodes.erase(lit);
        continue;
      }
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, ExtractorNode*>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), compareNodes);  // libstdc++ introsort decides tie order
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          ExtractorNode n1, n2, n3, n4;
          vPrev[j].second->DivideNode(n1, n2, n3, n4);
          push_children(n1, nullptr);
          push_children(n2, nullptr);
          push_children(n3, nullptr);
          push_children(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    std::vector<OKey>& vNodeKeys = it->vKeys;
    OKey* pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++) {
      if (vNodeKeys[k].response > maxResponse) {
        pKP = &vNodeKeys[k];
        maxResponse = vNodeKeys[k].response;
      }
    }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

// IC_Angle, src/ORBextractor.cc:71-95
float IC_Angle(const uint8_t* interior, int stride, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = interior + (std::ptrdiff_t)cvRoundF(pty) * stride + cvRoundF(ptx);
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

const float factorPI = (float)(3.14159265358979323846 / 180.f);  // :97

// computeOrbDescriptor, src/ORBextractor.cc:99-160 (img = blurred un-padded clone, step = cols)
void computeOrbDescriptor(float angle_deg, float ptx, float pty, const uint8_t* img, int step, uint8_t* desc) {
  float angle = angle_deg * factorPI;
  float a = (float)std::cos(angle), b = (float)std::sin(angle);  // float overloads -> cosf/sinf
  const uint8_t* center = img + (std::ptrdiff_t)(int)std::round(pty) * step + (int)std::round(ptx);
  const int* pattern = kBriefPattern;
  auto GET_VALUE = [&](int idx) -> int {
    float r1 = pattern[2 * idx] * b + pattern[2 * idx + 1] * a;
    float r2 = pattern[2 * idx] * a - pattern[2 * idx + 1] * b;
    int index = (int)std::round(r1) * step + (int)std::round(r2);
    return center[index];
  };
  for (int i = 0; i < 32; ++i, pattern += 32) {
    int val = 0;
    for (int k = 0; k < 8; k++) {
      int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

}  // namespace

struct gfso_orb {
  int nfeatures, nlevels, iniThFAST, minThFAST, blur_variant;
  int num_threads = 1;  // OpenMP over levels / keypoints like the reference's ENABLE_OMP build (src/ORBextractor.cc:775-777,1133-1137)
  double scaleFactor;  // include/ORBextractor.h:109 declares it double
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  // per-call state
  std::vector<Plane> pyr;
  std::vector<std::vector<uint8_t>> blurred;
  std::vector<std::vector<FastKp>> cands;
  std::vector<std::vector<gfso_keypoint>> level_kps;
};

extern "C" {

gfso_orb* gfso_orb_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int blur_variant) {
  // ORBextractor::ORBextractor, src/ORBextractor.cc:421-479
  gfso_orb* o = new gfso_orb;
  o->nfeatures = nfeatures;
  o->scaleFactor = scale_factor;
  o->nlevels = nlevels;
  o->iniThFAST = ini_th;
  o->minThFAST = min_th;
  o->blur_variant = blur_variant;
  o->mvScaleFactor.resize(nlevels);
  o->mvLevelSigma2.resize(nlevels);
  o->mvScaleFactor[0] = 1.0f;
  o->mvLevelSigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    o->mvScaleFactor[i] = (float)(o->mvScaleFactor[i - 1] * o->scaleFactor);
    o->mvLevelSigma2[i] = o->mvScaleFactor[i] * o->mvScaleFactor[i];
  }
  o->mvInvScaleFactor.resize(nlevels);
  o->mvInvLevelSigma2.resize(nlevels);
  for (int i = 0; i < nlevels; i++) {
    o->mvInvScaleFactor[i] = 1.0f / o->mvScaleFactor[i];
    o->mvInvLevelSigma2[i] = 1.0f / o->mvLevelSigma2[i];
  }
  o->mnFeaturesPerLevel.resize(nlevels);
  float factor = (float)(1.0f / o->scaleFactor);
  float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sumFeatures = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    o->mnFeaturesPerLevel[level] = cvRoundF(nDesiredFeaturesPerScale);
    sumFeatures += o->mnFeaturesPerLevel[level];
    nDesiredFeaturesPerScale *= factor;
  }
  o->mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
  o->umax.resize(HALF_PATCH_SIZE + 1);
  int v, v0, vmax = cvFloorD(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
  int vmin = cvCeilD(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) o->umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (o->umax[v0] == o->umax[v0 + 1]) ++v0;
    o->umax[v] = v0;
    ++v0;
  }
  return o;
}

void gfso_orb_destroy(gfso_orb* o) { delete o; }
void gfso_orb_set_threads(gfso_orb* o, int n) { o->num_threads = n < 1 ? 1 : n; }

void gfso_orb_get_tables(const gfso_orb* o, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                         int32_t* feats, int32_t* umax) {
  for (int i = 0; i < o->nlevels; i++) {
    if (scale) scale[i] = o->mvScaleFactor[i];
    if (inv_scale) inv_scale[i] = o->mvInvScaleFactor[i];
    if (sigma2) sigma2[i] = o->mvLevelSigma2[i];
    if (inv_sigma2) inv_sigma2[i] = o->mvInvLevelSigma2[i];
    if (feats) feats[i] = o->mnFeaturesPerLevel[i];
  }
  if (umax)
    for (int i = 0; i < 16; i++) umax[i] = o->umax[i];
}

static void make_border(Plane& p) {  // copyMakeBorder(..., BORDER_REFLECT_101 [+ISOLATED]) :1243-1248
  uint8_t* in = p.interior();
  for (int y = -EDGE_THRESHOLD; y < p.rows + EDGE_THRESHOLD; y++) {
    int sy = reflect101(y, p.rows);
    uint8_t* drow = in + (std::ptrdiff_t)y * p.stride;
    const uint8_t* srow = in + (std::ptrdiff_t)sy * p.stride;
    if (y < 0 || y >= p.rows) std::memcpy(drow, srow, p.cols);
    for (int x = 1; x <= EDGE_THRESHOLD; x++) {
      drow[-x] = srow[reflect101(-x, p.cols)];
      drow[p.cols - 1 + x] = srow[reflect101(p.cols - 1 + x, p.cols)];
    }
  }
}

int gfso_orb_extract(gfso_orb* o, const uint8_t* img, int rows, int cols, int stride, int lap0, int lap1,
                     gfso_keypoint* kps_out, uint8_t* desc_out, int cap, int* n_out) {
  if (n_out) *n_out = 0;
  if (!img || rows <= 0 || cols <= 0) return -1;  // :1150
  const int nlevels = o->nlevels;
  // ---- ComputePyramid, :1227-1251
  o->pyr.assign(nlevels, Plane());
  for (int level = 0; level < nlevels; ++level) {
    float scale = o->mvInvScaleFactor[level];
    int w = cvRoundF((float)cols * scale), h = cvRoundF((float)rows * scale);
    Plane& p = o->pyr[level];
    p.rows = h;
    p.cols = w;
    p.stride = w + 2 * EDGE_THRESHOLD;
    p.buf.assign((size_t)p.stride * (h + 2 * EDGE_THRESHOLD), 0);
    if (level != 0) {
      const Plane& q = o->pyr[level - 1];
      resize_area_u8(q.interior(), q.rows, q.cols, q.stride, p.interior(), h, w, p.stride);
    } else {
      for (int y = 0; y < h; y++) std::memcpy(p.interior() + (size_t)y * p.stride, img + (size_t)y * stride, w);
    }
    make_border(p);
  }
  // ---- ComputeKeyPointsOctTree, :770-975
  o->cands.assign(nlevels, {});
  o->level_kps.assign(nlevels, {});
  const float W = 35;
#pragma omp parallel for num_threads(o->num_threads) schedule(dynamic, 1) if (o->num_threads > 1)
  for (int level = 0; level < nlevels; ++level) {
    const Plane& P = o->pyr[level];
    const int minBorderX = EDGE_THRESHOLD - 3;
    const int minBorderY = minBorderX;
    const int maxBorderX = P.cols - EDGE_THRESHOLD + 3;
    const int maxBorderY = P.rows - EDGE_THRESHOLD + 3;
    std::vector<OKey> vToDistributeKeys;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W);
    const int nRows = (int)(height / W);
    const int wCell = (int)std::ceil(width / nCols);
    const int hCell = (int)std::ceil(height / nRows);
    std::vector<FastKp> vKeysCell;
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        // rowRange(iniY,maxY).colRange(iniX,maxX): float -> int conversion truncates
        const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
        const uint8_t* cell = P.interior() + (std::ptrdiff_t)y0 * P.stride + x0;
        fast9_16(cell, y1 - y0, x1 - x0, P.stride, o->iniThFAST, true, vKeysCell);
        if (vKeysCell.empty()) fast9_16(cell, y1 - y0, x1 - x0, P.stride, o->minThFAST, true, vKeysCell);
        for (const FastKp& k : vKeysCell) {
          FastKp c{k.x + j * wCell, k.y + i * hCell, k.score};
          o->cands[level].push_back(c);
          vToDistributeKeys.push_back({(float)c.x, (float)c.y, (float)c.score, (int)vToDistributeKeys.size()});
        }
      }
    }
    std::vector<OKey> kept = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                               o->mnFeaturesPerLevel[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * o->mvScaleFactor[level]);
    for (const OKey& k : kept) {
      gfso_keypoint kp;
      kp.x = k.x + minBorderX;
      kp.y = k.y + minBorderY;
      kp.size = (float)scaledPatchSize;
      kp.angle = -1.f;
      kp.response = k.response;
      kp.octave = level;
      kp.class_id = -1;
      o->level_kps[level].push_back(kp);
    }
  }
  // computeOrientation, :481-500 (on the un-blurred padded level)
  for (int level = 0; level < nlevels; ++level)
    for (gfso_keypoint& kp : o->level_kps[level])
      kp.angle = IC_Angle(o->pyr[level].interior(), o->pyr[level].stride, kp.x, kp.y, o->umax);

  // ---- operator(), :1159-1222
  int nkeypoints = 0;
  for (int level = 0; level < nlevels; ++level) nkeypoints += (int)o->level_kps[level].size();
  if (n_out) *n_out = nkeypoints;
  o->blurred.assign(nlevels, {});
  int monoIndex = 0, stereoIndex = nkeypoints - 1;
  const bool emit = kps_out && desc_out && nkeypoints <= cap;
  std::vector<std::vector<uint8_t>> level_desc(nlevels);
  // blur + descriptors per level (independent of the packing order below, so they may run in parallel for timing)
#pragma omp parallel for num_threads(o->num_threads) schedule(dynamic, 1) if (o->num_threads > 1)
  for (int level = 0; level < nlevels; ++level) {
    std::vector<gfso_keypoint>& keypoints = o->level_kps[level];
    if (keypoints.empty()) continue;
    const Plane& P = o->pyr[level];
    // workingMat = mvImagePyramid[level].clone(); GaussianBlur(...)  :1188-1189
    std::vector<uint8_t> clone((size_t)P.rows * P.cols);
    for (int y = 0; y < P.rows; y++) std::memcpy(&clone[(size_t)y * P.cols], P.interior() + (size_t)y * P.stride, P.cols);
    o->blurred[level].resize(clone.size());
    gaussian_blur7(clone.data(), P.rows, P.cols, P.cols, o->blurred[level].data(), P.cols, o->blur_variant);
    level_desc[level].resize(keypoints.size() * 32);
    for (size_t i = 0; i < keypoints.size(); i++)
      computeOrbDescriptor(keypoints[i].angle, keypoints[i].x, keypoints[i].y, o->blurred[level].data(), P.cols,
                           &level_desc[level][i * 32]);
  }
  for (int level = 0; level < nlevels; ++level) {
    std::vector<gfso_keypoint>& keypoints = o->level_kps[level];
    if (keypoints.empty()) continue;
    const std::vector<uint8_t>& desc = level_desc[level];
    float scale = o->mvScaleFactor[level];
    for (size_t i = 0; i < keypoints.size(); i++) {
      gfso_keypoint kp = keypoints[i];
      if (level != 0) {
        kp.x *= scale;
        kp.y *= scale;
      }
      if (kp.x >= lap0 && kp.x <= lap1) {
        if (emit) {
          kps_out[stereoIndex] = kp;
          std::memcpy(desc_out + (size_t)stereoIndex * 32, &desc[i * 32], 32);
        }
        stereoIndex--;
      } else {
        if (emit) {
          kps_out[monoIndex] = kp;
          std::memcpy(desc_out + (size_t)monoIndex * 32, &desc[i * 32], 32);
        }
        monoIndex++;
      }
    }
  }
  return monoIndex;
}

void gfso_orb_level_size(const gfso_orb* o, int level, int* rows, int* cols) {
  *rows = o->pyr[level].rows;
  *cols = o->pyr[level].cols;
}
void gfso_orb_get_level(const gfso_orb* o, int level, uint8_t* dst) {
  const Plane& P = o->pyr[level];
  for (int y = 0; y < P.rows; y++) std::memcpy(dst + (size_t)y * P.cols, P.interior() + (size_t)y * P.stride, P.cols);
}
void gfso_orb_get_blurred(const gfso_orb* o, int level, uint8_t* dst) {
  const Plane& P = o->pyr[level];
  if (o->blurred[level].empty())
    std::memset(dst, 0, (size_t)P.rows * P.cols);
  else
    std::memcpy(dst, o->blurred[level].data(), (size_t)P.rows * P.cols);
}
int gfso_orb_num_candidates(const gfso_orb* o, int level) { return (int)o->cands[level].size(); }
void gfso_orb_get_candidates(const gfso_orb* o, int level, int32_t* x, int32_t* y, int32_t* score) {
  for (size_t i = 0; i < o->cands[level].size(); i++) {
    x[i] = o->cands[level][i].x;
    y[i] = o->cands[level][i].y;
    score[i] = o->cands[level][i].score;
  }
}
int gfso_orb_num_level_keypoints(const gfso_orb* o, int level) { return (int)o->level_kps[level].size(); }
void gfso_orb_get_level_keypoints(const gfso_orb* o, int level, gfso_keypoint* kps) {
  std::copy(o->level_kps[level].begin(), o->level_kps[level].end(), kps);
}

void gfso_resize_area_u8(const uint8_t* src, int srows, int scols, int sstride, uint8_t* dst, int drows, int dcols,
                         int dstride) {
  resize_area_u8(src, srows, scols, sstride, dst, drows, dcols, dstride);
}
int gfso_fast9_16(const uint8_t* img, int rows, int cols, int stride, int threshold, int nonmax, int32_t* x,
                  int32_t* y, int32_t* score, int cap) {
  std::vector<FastKp> out;
  fast9_16(img, rows, cols, stride, threshold, nonmax != 0, out);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) {
    x[i] = out[i].x;
    y[i] = out[i].y;
    score[i] = out[i].score;
  }
  return (int)out.size();
}
float gfso_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void gfso_gaussian_blur7(const uint8_t* src, int rows, int cols, int stride, uint8_t* dst, int dstride, int variant) {
  gaussian_blur7(src, rows, cols, stride, dst, dstride, variant);
}
int gfso_distribute_octree(const float* x, const float* y, const float* response, int n, int min_x, int max_x,
                           int min_y, int max_y, int n_features, int32_t* out_idx, int cap) {
  std::vector<OKey> in(n);
  for (int i = 0; i < n; i++) in[i] = {x[i], y[i], response[i], i};
  std::vector<OKey> out = DistributeOctTree(in, min_x, max_x, min_y, max_y, n_features);
  for (size_t i = 0; i < out.size() && (int)i < cap; i++) out_idx[i] = out[i].src;
  return (int)out.size();
}

}  // extern "C"

// std::sort with the compareNodes ordering on caller data (reference for the product's sort replica test)
extern "C" int gfso_std_sort_pairs(int32_t* size_key, int32_t* x_key, int32_t* payload, int n) {
  struct E {
    int s, x, p;
  };
  std::vector<E> v(n);
  for (int i = 0; i < n; i++) v[i] = E{size_key[i], x_key[i], payload[i]};
  std::sort(v.begin(), v.end(), [](const E& a, const E& b) {
    if (a.s < b.s) return true;
    if (a.s > b.s) return false;
    return a.x < b.x;
  });
  for (int i = 0; i < n; i++) {
    size_key[i] = v[i].s;
    x_key[i] = v[i].x;
    payload[i] = v[i].p;
  }
  return n;
}
extern "C" int gfso_std_partial_sort_pairs(int32_t* size_key, int32_t* x_key, int32_t* payload, int n) {
  struct E {
    int s, x, p;
  };
  std::vector<E> v(n);
  for (int i = 0; i < n; i++) v[i] = E{size_key[i], x_key[i], payload[i]};
  std::partial_sort(v.begin(), v.end(), v.end(), [](const E& a, const E& b) {  // introsort's depth-limit fallback
    if (a.s < b.s) return true;
    if (a.s > b.s) return false;
    return a.x < b.x;
  });
  for (int i = 0; i < n; i++) {
    size_key[i] = v[i].s;
    x_key[i] = v[i].x;
    payload[i] = v[i].p;
  }
  return n;
}
