// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
// (reference src/ORBmatcher.cc:1853-2063) for the single-camera case (Nleft == -1), with Frame::AssignFeaturesToGrid /
// PosInGrid / GetFeaturesInArea (src/Frame.cc:734-761, 1073-1084, 1007-1071), ORBmatcher::DescriptorDistance (:2536-2550)
// and ComputeThreeMaxima (:2500-2532).  Float arithmetic follows Sophus::SE3f / Eigen expression order
// (Thirdparty/Sophus/sophus/so3.hpp:358-367, se3.hpp:208-211,321-324; Pinhole::project src/CameraModels/Pinhole.cpp:43-49).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "gfs_oracle.h"

namespace {

constexpr int kGridCols = 64, kGridRows = 48, kHisto = 30, kThHigh = 100;

// SO3f * p: uv = q.vec x p; uv += uv; p + w * uv + q.vec x uv
inline void so3_act(const float* q, const float* p, float* o) {
  float uv[3] = {q[1] * p[2] - q[2] * p[1], q[2] * p[0] - q[0] * p[2], q[0] * p[1] - q[1] * p[0]};
  for (int k = 0; k < 3; k++) uv[k] += uv[k];
  const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int k = 0; k < 3; k++) o[k] = (p[k] + q[3] * uv[k]) + c[k];
}
inline void se3_act(const float* q, const float* t, const float* p, float* o) {  // so3() * p + translation()
  so3_act(q, p, o);
  for (int k = 0; k < 3; k++) o[k] += t[k];
}
inline int popcnt256(const uint8_t* a, const uint8_t* b) {
  int d = 0;
  for (int i = 0; i < 8; i++) {
    uint32_t x, y;
    std::memcpy(&x, a + 4 * i, 4);
    std::memcpy(&y, b + 4 * i, 4);
    d += __builtin_popcount(x ^ y);
  }
  return d;
}

}  // namespace

extern "C" int gfso_search_by_projection(const gfso_sbp_problem* p, int32_t* cur_match) {
  const int N = p->n_cur;
  for (int i = 0; i < N; i++) cur_match[i] = -1;
  // Frame::AssignFeaturesToGrid (cells hold key-point indices in ascending order)
  std::vector<std::vector<int>> grid(kGridCols * kGridRows);
  for (int i = 0; i < N; i++) {
    const int px = (int)std::round((p->cur_xy[2 * i] - p->min_x) * p->grid_w_inv);
    const int py = (int)std::round((p->cur_xy[2 * i + 1] - p->min_y) * p->grid_h_inv);
    if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
    grid[px * kGridRows + py].push_back(i);
  }
  // twc = Tcw.inverse().translation(); tlc = Tlw * twc
  float qi[4] = {-p->Tcw_q[0], -p->Tcw_q[1], -p->Tcw_q[2], p->Tcw_q[3]};  // conjugate, re-normalised by the SO3 constructor
  {
    const float n = std::sqrt((qi[0] * qi[0] + qi[2] * qi[2]) + (qi[1] * qi[1] + qi[3] * qi[3]));
    for (int k = 0; k < 4; k++) qi[k] /= n;
  }
  const float mt[3] = {p->Tcw_t[0] * -1.0f, p->Tcw_t[1] * -1.0f, p->Tcw_t[2] * -1.0f};
  float twc[3], tlc[3];
  so3_act(qi, mt, twc);
  se3_act(p->Tlw_q, p->Tlw_t, twc, tlc);
  const bool bForward = tlc[2] > p->b && !p->mono;
  const bool bBackward = -tlc[2] > p->b && !p->mono;
  const float factor = 1.0f / kHisto;
  std::vector<int> rotHist[kHisto];
  int nmatches = 0;
  // cur_match[i2]: -1 = CurrentFrame.mvpMapPoints[i2] untouched so far, >= 0 = overwritten with the map point of last entry l
  for (int l = 0; l < p->n_last; l++) {
    float x3Dc[3];
    se3_act(p->Tcw_q, p->Tcw_t, p->last_xw + 3 * l, x3Dc);
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    if (invzc < 0) continue;
    const float u = p->fx * x3Dc[0] / x3Dc[2] + p->cx, v = p->fy * x3Dc[1] / x3Dc[2] + p->cy;
    if (u < p->min_x || u > p->max_x) continue;
    if (v < p->min_y || v > p->max_y) continue;
    const int oct = p->last_octave[l];
    const float radius = p->th * p->scale_factors[oct];
    int minLevel, maxLevel;
    if (bForward) {
      minLevel = oct;
      maxLevel = -1;
    } else if (bBackward) {
      minLevel = 0;
      maxLevel = oct;
    } else {
      minLevel = oct - 1;
      maxLevel = oct + 1;
    }
    // Frame::GetFeaturesInArea
    const int nMinCellX = std::max(0, (int)std::floor((u - p->min_x - radius) * p->grid_w_inv));
    if (nMinCellX >= kGridCols) continue;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((u - p->min_x + radius) * p->grid_w_inv));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)std::floor((v - p->min_y - radius) * p->grid_h_inv));
    if (nMinCellY >= kGridRows) continue;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((v - p->min_y + radius) * p->grid_h_inv));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestIdx2 = -1;
    bool any = false;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (int i2 : grid[ix * kGridRows + iy]) {
          if (bCheckLevels) {
            if (p->cur_octave[i2] < minLevel) continue;
            if (maxLevel >= 0 && p->cur_octave[i2] > maxLevel) continue;
          }
          const float distx = p->cur_xy[2 * i2] - u, disty = p->cur_xy[2 * i2 + 1] - v;
          if (!(std::fabs(distx) < radius && std::fabs(disty) < radius)) continue;
          any = true;  // vIndices2 is not empty
          // CurrentFrame.mvpMapPoints[i2] && Observations() > 0
          const bool blocked = p->cur_has_mp_obs[i2] ? (cur_match[i2] == -1 ? true : p->last_mp_has_obs[cur_match[i2]] != 0)
                                                     : (cur_match[i2] >= 0 && p->last_mp_has_obs[cur_match[i2]] != 0);
          if (blocked) continue;
          if (p->cur_u_right[i2] > 0) {
            const float ur = u - p->bf * invzc;
            const float er = std::fabs(ur - p->cur_u_right[i2]);
            if (er > radius) continue;
          }
          const int dist = popcnt256(p->last_desc + 32 * l, p->cur_desc + 32 * i2);
          if (dist < bestDist) {
            bestDist = dist;
            bestIdx2 = i2;
          }
        }
    if (!any) continue;
    if (bestDist <= kThHigh) {
      cur_match[bestIdx2] = l;
      nmatches++;
      if (p->check_orientation) {
        float rot = p->last_angle[l] - p->cur_angle[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == kHisto) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (p->check_orientation) {  // ComputeThreeMaxima
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < kHisto; i++) {
      const int s = (int)rotHist[i].size();
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
    for (int i = 0; i < kHisto; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int idx : rotHist[i]) {
          cur_match[idx] = -2;  // CurrentFrame.mvpMapPoints[idx] = NULL
          nmatches--;
        }
  }
  return nmatches;
}

// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
// (reference src/ORBmatcher.cc:43-206), single-camera frames: best / second-best Hamming match inside a window whose size
// depends on the viewing angle (RadiusByViewingCos :250-255), ratio test when both are on the same pyramid level.
extern "C" int gfso_search_by_projection_map(const gfso_sbp_map_problem* p, int32_t* cur_match) {
  const int N = p->n_cur;
  for (int i = 0; i < N; i++) cur_match[i] = -1;
  std::vector<std::vector<int>> grid(kGridCols * kGridRows);
  for (int i = 0; i < N; i++) {
    const int px = (int)std::round((p->cur_xy[2 * i] - p->min_x) * p->grid_w_inv);
    const int py = (int)std::round((p->cur_xy[2 * i + 1] - p->min_y) * p->grid_h_inv);
    if (px < 0 || px >= kGridCols || py < 0 || py >= kGridRows) continue;
    grid[px * kGridRows + py].push_back(i);
  }
  const bool bFactor = (double)p->th != 1.0;
  int nmatches = 0;
  for (int l = 0; l < p->n_mp; l++) {
    const int level = p->mp_level[l];
    float r = (double)p->mp_view_cos[l] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos
    if (bFactor) r *= p->th;
    const float x = p->mp_proj[3 * l], y = p->mp_proj[3 * l + 1], xr = p->mp_proj[3 * l + 2];
    const float radius = r * p->scale_factors[level];
    const int minLevel = level - 1, maxLevel = level;
    const int nMinCellX = std::max(0, (int)std::floor((x - p->min_x - radius) * p->grid_w_inv));
    if (nMinCellX >= kGridCols) continue;
    const int nMaxCellX = std::min(kGridCols - 1, (int)std::ceil((x - p->min_x + radius) * p->grid_w_inv));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = std::max(0, (int)std::floor((y - p->min_y - radius) * p->grid_h_inv));
    if (nMinCellY >= kGridRows) continue;
    const int nMaxCellY = std::min(kGridRows - 1, (int)std::ceil((y - p->min_y + radius) * p->grid_h_inv));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    bool any = false;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
        for (int idx : grid[ix * kGridRows + iy]) {
          if (bCheckLevels) {
            if (p->cur_octave[idx] < minLevel) continue;
            if (maxLevel >= 0 && p->cur_octave[idx] > maxLevel) continue;
          }
          const float distx = p->cur_xy[2 * idx] - x, disty = p->cur_xy[2 * idx + 1] - y;
          if (!(std::fabs(distx) < radius && std::fabs(disty) < radius)) continue;
          any = true;
          const bool blocked = p->cur_has_mp_obs[idx] ? (cur_match[idx] == -1 ? true : p->mp_has_obs[cur_match[idx]] != 0)
                                                      : (cur_match[idx] >= 0 && p->mp_has_obs[cur_match[idx]] != 0);
          if (blocked) continue;
          if (p->cur_u_right[idx] > 0) {
            const float er = std::fabs(xr - p->cur_u_right[idx]);
            if (er > r * p->scale_factors[level]) continue;
          }
          const int dist = popcnt256(p->mp_desc + 32 * l, p->cur_desc + 32 * idx);
          if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestLevel2 = bestLevel;
            bestLevel = p->cur_octave[idx];
            bestIdx = idx;
          } else if (dist < bestDist2) {
            bestLevel2 = p->cur_octave[idx];
            bestDist2 = dist;
          }
        }
    if (!any) continue;
    if (bestDist <= kThHigh) {
      if (bestLevel == bestLevel2 && bestDist > p->nn_ratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= p->nn_ratio * bestDist2) {
        cur_match[bestIdx] = l;
        nmatches++;
      }
    }
  }
  return nmatches;
}
