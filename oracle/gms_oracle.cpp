// TEST INFRASTRUCTURE — NOT PRODUCT CODE (see gfs_oracle.h header). PARITY UNPINNED.
//
// CPU restatement of gms_matcher::GetInlierMask(vbInliers, WithScale = false, WithRotation = false) (reference
// Thirdparty/GMS/include/gms_matcher.h, vendored in the reference tree): constructor :43-60 (NormalizePoints :121-133,
// ConvertMatches :136-143, 20 x 20 left grid, neighbour table :206-240), GetInlierMask :289-301 -> SetScale(0) :241-251 ->
// run(1) :423-455 with AssignMatchPairs :356-383, VerifyCellPairs :386-421 (rotation pattern 1 = identity) for the four
// shifted grids (GetGridIndexLeft :146-176, GetGridIndexRight :178-183).  Call sites: src/ORBmatcher.cc:761-762, 812-813, 893-894.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "gfs_oracle.h"

namespace {
constexpr int kGridW = 20, kGridH = 20, kCells = kGridW * kGridH;
constexpr int kThreshFactor = 6;

inline int grid_index_left(float px, float py, int type) {  // :146-176
  int x = 0, y = 0;
  if (type == 1) {
    x = (int)std::floor(px * kGridW);
    y = (int)std::floor(py * kGridH);
  } else if (type == 2) {
    x = (int)std::floor(px * kGridW + 0.5);
    y = (int)std::floor(py * kGridH);
  } else if (type == 3) {
    x = (int)std::floor(px * kGridW);
    y = (int)std::floor(py * kGridH + 0.5);
  } else {
    x = (int)std::floor(px * kGridW + 0.5);
    y = (int)std::floor(py * kGridH + 0.5);
  }
  if (x >= kGridW || y >= kGridH) return -1;
  return x + y * kGridW;
}
inline void nb9(int idx, int* out) {  // GetNB9 :190-211
  for (int k = 0; k < 9; k++) out[k] = -1;
  const int ix = idx % kGridW, iy = idx / kGridW;
  for (int yi = -1; yi <= 1; yi++)
    for (int xi = -1; xi <= 1; xi++) {
      const int xx = ix + xi, yy = iy + yi;
      if (xx < 0 || xx >= kGridW || yy < 0 || yy >= kGridH) continue;
      out[xi + 4 + yi * 3] = xx + yy * kGridW;
    }
}
}  // namespace

extern "C" int gfso_gms_inlier_mask(const float* kp1_xy, int n1, int width1, int height1, const float* kp2_xy, int n2, int width2,
                                    int height2, const int32_t* query_idx, const int32_t* train_idx, int n_matches,
                                    uint8_t* inlier) {
  (void)n1;
  (void)n2;
  std::vector<float> p1(2 * (size_t)n1), p2(2 * (size_t)n2);
  for (int i = 0; i < n1; i++) {  // NormalizePoints: float / int
    p1[2 * i] = kp1_xy[2 * i] / width1;
    p1[2 * i + 1] = kp1_xy[2 * i + 1] / height1;
  }
  for (int i = 0; i < n2; i++) {
    p2[2 * i] = kp2_xy[2 * i] / width2;
    p2[2 * i + 1] = kp2_xy[2 * i + 1] / height2;
  }
  for (int i = 0; i < n_matches; i++) inlier[i] = 0;
  std::vector<int> stats((size_t)kCells * kCells), cell_pairs(kCells), npts(kCells), pair_l(n_matches, 0), pair_r(n_matches, 0);
  for (int type = 1; type <= 4; type++) {
    std::fill(stats.begin(), stats.end(), 0);
    std::fill(cell_pairs.begin(), cell_pairs.end(), -1);
    std::fill(npts.begin(), npts.end(), 0);
    for (int i = 0; i < n_matches; i++) {  // AssignMatchPairs
      const float* lp = &p1[2 * (size_t)query_idx[i]];
      const float* rp = &p2[2 * (size_t)train_idx[i]];
      const int l = pair_l[i] = grid_index_left(lp[0], lp[1], type);
      int r;
      if (type == 1)
        r = pair_r[i] = (int)std::floor(rp[0] * kGridW) + (int)std::floor(rp[1] * kGridH) * kGridW;  // GetGridIndexRight: no range check
      else
        r = pair_r[i];
      if (l < 0 || r < 0) continue;
      if (l >= kCells || r >= kCells) continue;
      stats[(size_t)l * kCells + r]++;
      npts[l]++;
    }
    for (int i = 0; i < kCells; i++) {  // VerifyCellPairs(1)
      const int* row = &stats[(size_t)i * kCells];
      long rowsum = 0;
      for (int j = 0; j < kCells; j++) rowsum += row[j];
      if (rowsum == 0) {
        cell_pairs[i] = -1;
        continue;
      }
      int max_number = 0;
      for (int j = 0; j < kCells; j++)
        if (row[j] > max_number) {
          cell_pairs[i] = j;
          max_number = row[j];
        }
      int nl[9], nr[9];
      nb9(i, nl);
      nb9(cell_pairs[i], nr);
      int score = 0, numpair = 0;
      double thresh = 0;
      for (int j = 0; j < 9; j++) {
        const int ll = nl[j], rr = nr[j];  // mRotationPatterns[0] is the identity
        if (ll == -1 || rr == -1) continue;
        score += stats[(size_t)ll * kCells + rr];
        thresh += npts[ll];
        numpair++;
      }
      thresh = kThreshFactor * std::sqrt(thresh / numpair);
      if (score < thresh) cell_pairs[i] = -2;
    }
    for (int i = 0; i < n_matches; i++) {
      // the reference reads mCellPairs[-1] (out of bounds) when the left index is -1; that never equals a right index
      if (pair_l[i] < 0) continue;
      if (cell_pairs[pair_l[i]] == pair_r[i]) inlier[i] = 1;
    }
  }
  int n = 0;
  for (int i = 0; i < n_matches; i++) n += inlier[i];
  return n;
}
