// Host-side tables and quadtree of the ORB extractor (see orb_host.hpp for the reference lines followed).
#include "orb_host.hpp"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <utility>

namespace gfs {

namespace {
inline int cv_round_f(float v) { return (int)std::lrintf(v); }  // cvRound: round-half-even
inline int cv_round_d(double v) { return (int)std::lrint(v); }
inline int cv_floor_d(double v) {
  int i = (int)v;
  return i - (i > v);
}
inline int cv_ceil_d(double v) {
  int i = (int)v;
  return i + (i < v);
}
}  // namespace

// ORBextractor::ORBextractor, src/ORBextractor.cc:421-479.  `scaleFactor` is a double member initialised from
// the float argument (include/ORBextractor.h:109), hence the (double) products rounded back to float.
void OrbParams::init(int nf, float sf, int nl, int ini, int mn, int bv) {
  nfeatures = nf;
  scale_factor = sf;
  nlevels = nl;
  ini_th = ini;
  min_th = mn;
  blur_variant = bv;
  const double sfd = sf;
  scale.assign(nl, 1.f);
  sigma2.assign(nl, 1.f);
  for (int i = 1; i < nl; i++) {
    scale[i] = (float)(scale[i - 1] * sfd);
    sigma2[i] = scale[i] * scale[i];
  }
  inv_scale.resize(nl);
  inv_sigma2.resize(nl);
  for (int i = 0; i < nl; i++) {
    inv_scale[i] = 1.0f / scale[i];
    inv_sigma2[i] = 1.0f / sigma2[i];
  }
  quota.assign(nl, 0);
  float factor = (float)(1.0f / sfd);
  float desired = nf * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
  int sum = 0;
  for (int l = 0; l < nl - 1; l++) {
    quota[l] = cv_round_f(desired);
    sum += quota[l];
    desired *= factor;
  }
  quota[nl - 1] = std::max(nf - sum, 0);
  int v, v0, vmax = cv_floor_d(kHalfPatch * std::sqrt(2.f) / 2 + 1);
  int vmin = cv_ceil_d(kHalfPatch * std::sqrt(2.f) / 2);
  const double hp2 = kHalfPatch * kHalfPatch;
  for (v = 0; v < 16; v++) umax[v] = 0;
  for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
  for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

// OpenCV computeResizeAreaTab (imgproc/src/resize.cpp) regrouped per destination index.
static bool area_tab(int ssize, int dsize, std::vector<int>& start, std::vector<int>& cnt, std::vector<float>& alpha) {
  const double scale = 1. / ((double)dsize / ssize);
  for (int dx = 0; dx < dsize; dx++) {
    double fsx1 = dx * scale, fsx2 = fsx1 + scale;
    double cellWidth = std::min(scale, ssize - fsx1);
    int sx1 = cv_ceil_d(fsx1), sx2 = cv_floor_d(fsx2);
    sx2 = std::min(sx2, ssize - 1);
    sx1 = std::min(sx1, sx2);
    int s0 = -1, n = 0;
    float a[kMaxAreaTaps + 4];
    auto push = [&](int si, float al) {
      if (n == 0) s0 = si;
      if (n < kMaxAreaTaps + 4) a[n] = al;
      n++;
      return si == s0 + n - 1;
    };
    bool ok = true;
    if (sx1 - fsx1 > 1e-3) ok &= push(sx1 - 1, (float)((sx1 - fsx1) / cellWidth));
    for (int sx = sx1; sx < sx2; sx++) ok &= push(sx, float(1.0 / cellWidth));
    if (fsx2 - sx2 > 1e-3) ok &= push(sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth));
    if (!ok || n == 0 || n > kMaxAreaTaps) return false;
    start.push_back(s0);
    cnt.push_back(n);
    for (int k = 0; k < kMaxAreaTaps; k++) alpha.push_back(k < n ? a[k] : 0.f);
  }
  return true;
}

void OrbGeometry::build(const OrbParams& p, int rows_, int cols_) {
  rows = rows_;
  cols = cols_;
  levels.assign(p.nlevels, LevelDev{});
  cells.clear();
  blur_tiles.clear();
  xt_start.clear();
  yt_start.clear();
  xt_n.clear();
  yt_n.clear();
  xt_alpha.clear();
  yt_alpha.clear();
  supported = true;
  why = "";
  size_t pyr = 0, blur = 0, slab = 0;
  kp_cap = 0;
  max_tile_w = max_tile_h = 0;
  fast_lds_wave = 0;
  for (int l = 0; l < p.nlevels; l++) {
    LevelDev& L = levels[l];
    // ComputePyramid, src/ORBextractor.cc:1228-1231
    L.cols = cv_round_f((float)cols * p.inv_scale[l]);
    L.rows = cv_round_f((float)rows * p.inv_scale[l]);
    L.pitch = (int)((L.cols + 127) / 128 * 128);  // whole 128-byte lines a row (round 6: two 64-pixel blur tiles share a line exactly)
    L.plane_off = (unsigned)pyr;
    if (l > 0) pyr += (size_t)L.pitch * L.rows;
    L.blur_off = (unsigned)blur;
    blur += (size_t)L.pitch * L.rows;
    L.scale = p.scale[l];
    L.patch_size = (float)(int)(kPatchSize * p.scale[l]);
    L.quota = p.quota[l];
    if (l > 0) {
      const LevelDev& S = levels[l - 1];
      const double sx = 1. / ((double)L.cols / S.cols), sy = 1. / ((double)L.rows / S.rows);
      const bool int_x = std::abs(sx - (int)std::lrint(sx)) < DBL_EPSILON, int_y = std::abs(sy - (int)std::lrint(sy)) < DBL_EPSILON;
      if (sx < 1 || sy < 1 || (int_x && int_y)) {
        supported = false;
        why = "level ratio must be > 1 and non-integer (cv::resize INTER_AREA general path)";
        return;
      }
      L.xtab_off = (int)xt_start.size();
      L.ytab_off = (int)yt_start.size();
      if (!area_tab(S.cols, L.cols, xt_start, xt_n, xt_alpha) || !area_tab(S.rows, L.rows, yt_start, yt_n, yt_alpha)) {
        supported = false;
        why = "INTER_AREA footprint wider than 4 source pixels (scale factor too large)";
        return;
      }
    }
    // cell grid, src/ORBextractor.cc:778-806
    const int minBX = kEdgeThreshold - 3, minBY = minBX;
    L.max_bx = L.cols - kEdgeThreshold + 3;
    L.max_by = L.rows - kEdgeThreshold + 3;
    const float W = 35;
    const float width = (float)(L.max_bx - minBX), height = (float)(L.max_by - minBY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) {
      supported = false;
      why = "pyramid level smaller than one FAST cell (reference divides by zero here)";
      return;
    }
    L.n_cell_cols = nCols;
    L.n_cell_rows = nRows;
    L.w_cell = (int)std::ceil(width / nCols);
    L.h_cell = (int)std::ceil(height / nRows);
    L.cell_base = (int)cells.size();
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBY + i * L.h_cell);
      float maxY = iniY + L.h_cell + 6;
      if (iniY >= L.max_by - 3) continue;
      if (maxY > L.max_by) maxY = (float)L.max_by;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBX + j * L.w_cell);
        float maxX = iniX + L.w_cell + 6;
        if (iniX >= L.max_bx - 6) continue;
        if (maxX > L.max_bx) maxX = (float)L.max_bx;
        CellDev c{};
        c.level = (short)l;
        c.x0 = (short)(int)iniX;
        c.y0 = (short)(int)iniY;
        c.w = (short)((int)maxX - (int)iniX);
        c.h = (short)((int)maxY - (int)iniY);
        const int dw = std::max(c.w - 6, 0), dh = std::max(c.h - 6, 0);
        c.pitch = L.pitch;
        c.plane_off = L.plane_off;
        c.slab_off = (unsigned)slab;
        c.slab_cap = (unsigned)(((dw + 1) / 2) * ((dh + 1) / 2));  // 3x3 strict-max NMS keeps <= 1 per 2x2 block
        slab += c.slab_cap;
        max_tile_w = std::max(max_tile_w, (int)c.w);
        max_tile_h = std::max(max_tile_h, (int)c.h);
        if (dw > 0 && dh > 0) {
          // k_fast_cells: rows of (3 + w + 3) & ~3 bytes at most (word alignment of the cell's first pixel), two maps + 16 bytes each and
          // the offset list; phase 0 reads up to rstep + 3 (masked) rows past the tile, which must stay inside the wave's region
          const size_t wp = (size_t)((c.w + 6) & ~3);
          const size_t ng_min = (size_t)std::max(1, dw / 4), past = wp * (c.h + 64 / ng_min + 3) + 16;
          const size_t need = std::max(2 * (wp * c.h + 16) + 2 * (size_t)dw * dh + 16, past);
          fast_lds_wave = std::max(fast_lds_wave, (need + 15) & ~(size_t)15);
        }
        cells.push_back(c);
      }
    }
    L.n_cells = (int)cells.size() - L.cell_base;
    int nIni = (int)std::round(width / height);
    if (nIni == 0) nIni = 1;
    L.kp_cap = L.quota + 3 + 4 * nIni;
    kp_cap += L.kp_cap;
    for (int ty = 0; ty < (L.rows + 31) / 32; ty++)  // 64 x 32 tiles (k_blur7)
      for (int tx = 0; tx < (L.cols + 63) / 64; tx++) blur_tiles.push_back(BlurTileDev{(short)l, (short)tx, (short)ty, 0, L.rows, L.cols, L.pitch, L.plane_off, L.blur_off, 0u});
  }
  {
    // Order of k_blur7's workgroups (round 6).  MI355X hands workgroup i to XCD i % 8, and every XCD has its own L2.  The two 64-pixel
    // tiles over one 128-byte line of every row (tx = 2 k, 2 k + 1) and the two tile rows that share a 6-row halo (ty = 2 m, 2 m + 1)
    // form a block of four that goes to positions p, p + 8, p + 16, p + 24 of the list: ONE XCD's L2 serves all four -- in the natural
    // order every line crossed the fabric twice and the halo rows once more, FETCH_SIZE 2.4 x the pixels (profiles r05e).  A whole
    // frame's list shifts by the same amount from frame to frame, so the grouping holds for every frame of a batch.
    auto at = [&](int l, int tx, int ty) -> const BlurTileDev* {
      for (const BlurTileDev& t : blur_tiles)
        if (t.level == l && t.tx == tx && t.ty == ty) return &t;
      return nullptr;
    };
    std::vector<std::array<BlurTileDev, 4>> quads;
    std::vector<std::array<BlurTileDev, 2>> pairs;
    std::vector<BlurTileDev> single;
    for (const BlurTileDev& t : blur_tiles) {
      if ((t.tx & 1) || (t.ty & 1)) {
        // an odd tile belongs to the even tile left of / above it when that one exists (it always does), else it is on its own
        continue;
      }
      const BlurTileDev *r = at(t.level, t.tx + 1, t.ty), *d = at(t.level, t.tx, t.ty + 1), *rd = at(t.level, t.tx + 1, t.ty + 1);
      if (r && d && rd) {
        quads.push_back({t, *r, *d, *rd});
      } else if (r) {
        pairs.push_back({t, *r});
        if (d) single.push_back(*d);
      } else if (d) {
        pairs.push_back({t, *d});
      } else {
        single.push_back(t);
      }
    }
    std::vector<BlurTileDev> out;
    out.reserve(blur_tiles.size());
    size_t g = 0;
    for (; g + 8 <= quads.size(); g += 8)
      for (int k = 0; k < 4; k++)
        for (size_t i = 0; i < 8; i++) out.push_back(quads[g + i][k]);
    for (; g < quads.size(); g++) {  // (fewer than eight blocks left: as two pairs)
      pairs.push_back({quads[g][0], quads[g][1]});
      pairs.push_back({quads[g][2], quads[g][3]});
    }
    for (g = 0; g + 8 <= pairs.size(); g += 8)
      for (int k = 0; k < 2; k++)
        for (size_t i = 0; i < 8; i++) out.push_back(pairs[g + i][k]);
    for (; g < pairs.size(); g++) {
      out.push_back(pairs[g][0]);
      out.push_back(pairs[g][1]);
    }
    out.insert(out.end(), single.begin(), single.end());
    if (out.size() == blur_tiles.size()) blur_tiles.swap(out);  // (every tile exactly once; else keep the natural order)
  }
  if (cols >= 4096 || rows >= 4096) {
    supported = false;
    why = "images must be smaller than 4096 x 4096 (12-bit packed candidate coordinates)";
    return;
  }
  pyr_bytes = pyr;
  blur_bytes = blur;
  slab_entries = slab;
  cand_cap = slab;
  // Row strips of the fused pyramid kernel.  Strip k owns rows [k R_l / S, (k+1) R_l / S) of every level; to produce them
  // without any inter-workgroup dependency it also computes, at every lower level, the rows its higher levels read
  // (INTER_AREA footprints from the y tables).  Neighbouring strips recompute a few identical halo rows.  S is the smallest
  // candidate whose level-0 and level-1 strips (the two LDS ping-pong buffers) fit the CU's LDS.
  pyr_strips = pyr_strips_fine = 0;
  pyr_lds_a = pyr_lds_b = pyr_lds_a_fine = pyr_lds_b_fine = pyr_lds_x = pyr_lds_x_fine = 0;
  strip_rows.clear();
  strip_rows_fine.clear();
  const int nl = p.nlevels;
  if (nl >= 2) {
    struct Cut {
      int S = 0;
      std::vector<int> tab;
      size_t la = 0, lb = 0;
      int tab_rows = 0;   // most rows of levels >= 1 one strip produces
      int prog_rows = 0;  // most rows of levels < nl - 1 one strip reads
    };
    auto cut = [&](int S) {
      Cut c;
      c.S = S;
      c.tab.assign((size_t)2 * S * nl, 0);
      for (int k = 0; k < S; k++) {
        int a = 0, b = 0, rows = 0, srows = 0;
        for (int l = nl - 1; l >= 0; l--) {
          const int R = levels[l].rows;
          int oa = l >= 1 ? (int)((long long)k * R / S) : 0, ob = l >= 1 ? (int)((long long)(k + 1) * R / S) : 0;
          if (l < nl - 1 && b > a) {  // footprint of the rows held at level l + 1
            const LevelDev& U = levels[l + 1];
            const int fa = yt_start[U.ytab_off + a], fb = yt_start[U.ytab_off + b - 1] + yt_n[U.ytab_off + b - 1];
            if (ob > oa) {
              oa = std::min(oa, fa);
              ob = std::max(ob, fb);
            } else {
              oa = fa;
              ob = fb;
            }
          }
          a = oa;
          b = ob;
          c.tab[2 * ((size_t)k * nl + l)] = a;
          c.tab[2 * ((size_t)k * nl + l) + 1] = b;
          const size_t bytes = (size_t)(b - a) * kPyrLdsPitch(levels[l].cols);
          if (l % 2 == 0) c.la = std::max(c.la, bytes);
          else c.lb = std::max(c.lb, bytes);
          if (l >= 1) rows += b - a;
          if (l < nl - 1) srows += b - a;
        }
        c.tab_rows = std::max(c.tab_rows, rows);
        c.prog_rows = std::max(c.prog_rows, srows);
      }
      c.la = (c.la + 15) / 16 * 16;
      c.lb = (c.lb + 15) / 16 * 16;
      return c;
    };
    const int cand[] = {4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 48, 64};
    size_t budget = kPyrLdsBudget;
    if (const char* e = getenv("GFS_ORB_PYR_LDS_KB")) budget = std::min(kPyrLdsBudget, (size_t)std::max(8, atoi(e)) * 1024);  // tuning knob, never above what the kernel may allocate
    size_t xbytes = 16 * xt_start.size();
    for (int n : xt_n)
      if (n > 3) xbytes = 0;
    if (getenv("GFS_ORB_PYR_XTAB_HBM")) xbytes = 0;  // test knob: the x tables stay in memory
    std::vector<Cut> cuts;
    for (int S : cand) cuts.push_back(cut(S));
    // k_pyr_area_lds streams the source rows of a run once, in order: consecutive rows of a level may share their boundary
    // source row, no more (true of INTER_AREA tables at scale factors >= 1; checked, not assumed)
    bool streamable = true;
    for (int l = 1; l < nl && streamable; l++)
      for (int r = 0; r + 1 < levels[l].rows; r++) {
        const int i = levels[l].ytab_off + r, last = yt_start[i] + yt_n[i] - 1;
        if (yt_start[i + 1] < last || (yt_start[i + 1] == last && yt_n[i + 1] == 1)) {  // (two rows may not end on one source row)
          streamable = false;
          break;
        }
      }
    // the coarsest cut that fits without the x tables, and with them; the tables only go to LDS when that does not cost much
    // finer strips (every strip recomputes its halo at every level: a 720p frame fits as 16 strips without them, as 48 with
    // them, and ran 184 us against 105 us per 32 frames)
    auto first_fit = [&](size_t extra) {
      for (size_t i = 0; i < cuts.size() && streamable; i++)
        if (cuts[i].la + cuts[i].lb + extra <= budget && cuts[i].tab_rows <= kPyrTabRows && cuts[i].prog_rows <= kPyrProgRows)
          return (int)i;
      return -1;
    };
    const int pick0 = first_fit(0), pickx = xbytes ? first_fit(xbytes) : -1;
    int pick = pick0;
    if (pickx >= 0 && 2 * cuts[pickx].S <= 3 * cuts[pick0].S) {
      pick = pickx;
      pyr_lds_x = xbytes;
    }
    if (pick < 0) {  // the strips do not fit the LDS at all: the HBM path, finest cut
      strip_rows = cuts.back().tab;
      pyr_strips = cuts.back().S;
    } else {
      strip_rows = cuts[pick].tab;
      pyr_strips = cuts[pick].S;
      pyr_lds_a = cuts[pick].la;
      pyr_lds_b = cuts[pick].lb;
      for (size_t i = pick + 1; i < cuts.size(); i++)
        if (cuts[i].S >= 4 * pyr_strips || i + 1 == cuts.size()) {
          strip_rows_fine = cuts[i].tab;
          pyr_strips_fine = cuts[i].S;
          pyr_lds_a_fine = cuts[i].la;
          pyr_lds_b_fine = cuts[i].lb;
          pyr_lds_x_fine = cuts[i].la + cuts[i].lb + xbytes <= budget ? xbytes : 0;
          break;
        }
    }
  }
}

void ic_angle_offsets(const int umax[16], std::vector<int8_t>& du, std::vector<int8_t>& dv) {
  du.clear();
  dv.clear();
  for (int u = -kHalfPatch; u <= kHalfPatch; ++u) {
    du.push_back((int8_t)u);
    dv.push_back(0);
  }
  for (int v = 1; v <= kHalfPatch; ++v) {
    const int d = umax[v];
    for (int u = -d; u <= d; ++u) {
      du.push_back((int8_t)u);
      dv.push_back((int8_t)v);
      du.push_back((int8_t)u);
      dv.push_back((int8_t)-v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// DistributeOctTree — index-based equivalent of the std::list algorithm (src/ORBextractor.cc:567-768).
// ------------------------------------------------------------------------------------------------
void distribute_octree(const uint32_t* c, int n, int min_x, int max_x, int min_y, int max_y, int N,
                       OctreeScratch& S, std::vector<int>& out) {
  if (n <= 0) return;
  int nIni = (int)std::round(static_cast<float>(max_x - min_x) / (max_y - min_y));  // :573
  if (nIni == 0) nIni = 1;
  const float hX = static_cast<float>(max_x - min_x) / nIni;
  auto& nodes = S.nodes;
  auto& perm = S.perm;
  auto& tmp = S.tmp;
  nodes.clear();
  perm.resize(n);
  tmp.resize(n);
  int head = -1, tail = -1, size = 0;
  auto push_back = [&](int idx) {
    nodes[idx].prev = tail;
    nodes[idx].next = -1;
    if (tail >= 0)
      nodes[tail].next = idx;
    else
      head = idx;
    tail = idx;
    size++;
  };
  auto push_front = [&](int idx) {
    nodes[idx].prev = -1;
    nodes[idx].next = head;
    if (head >= 0)
      nodes[head].prev = idx;
    else
      tail = idx;
    head = idx;
    size++;
  };
  auto erase = [&](int idx) {
    const int p = nodes[idx].prev, q = nodes[idx].next;
    if (p >= 0)
      nodes[p].next = q;
    else
      head = q;
    if (q >= 0)
      nodes[q].prev = p;
    else
      tail = p;
    size--;
  };
  // root nodes (:586-604): stable bucket of the candidates by (size_t)(pt.x / hX)
  std::vector<int> rcount(nIni + 1, 0);
  for (int i = 0; i < n; i++) {
    int r = (int)((float)cand_x(c[i]) / hX);
    r = std::min(std::max(r, 0), nIni - 1);
    tmp[i] = r;
    rcount[r + 1]++;
  }
  for (int r = 0; r < nIni; r++) rcount[r + 1] += rcount[r];
  {
    std::vector<int> pos(rcount.begin(), rcount.end() - 1);
    for (int i = 0; i < n; i++) perm[pos[tmp[i]]++] = i;
  }
  for (int r = 0; r < nIni; r++) {
    OctreeScratch::Node nd;
    nd.x0 = (int)(hX * static_cast<float>(r));
    nd.x1 = (int)(hX * static_cast<float>(r + 1));
    nd.y0 = 0;
    nd.y1 = max_y - min_y;
    nd.kb = rcount[r];
    nd.ke = rcount[r + 1];
    nd.no_more = false;
    nd.prev = nd.next = -1;
    nodes.push_back(nd);
    push_back((int)nodes.size() - 1);
  }
  for (int it = head; it >= 0;) {  // :606-616
    const int nx = nodes[it].next, cnt = nodes[it].ke - nodes[it].kb;
    if (cnt == 1)
      nodes[it].no_more = true;
    else if (cnt == 0)
      erase(it);
    it = nx;
  }

  std::vector<std::pair<int, int>> vsize, vprev;  // (size, node index)
  // ExtractorNode::DivideNode (:502-550) + the four push_front blocks (:640-676 / :704-734)
  auto split = [&](int it, int* n_to_expand) {
    const OctreeScratch::Node P = nodes[it];
    const int halfX = (int)std::ceil(static_cast<float>(P.x1 - P.x0) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(P.y1 - P.y0) / 2);
    const int xm = P.x0 + halfX, ym = P.y0 + halfY;
    int cnt[4] = {0, 0, 0, 0};
    for (int k = P.kb; k < P.ke; k++) {
      const uint32_t v = c[perm[k]];
      const int q = (cand_x(v) < xm ? 0 : 1) + (cand_y(v) < ym ? 0 : 2);
      tmp[k] = q;
      cnt[q]++;
    }
    int base[4] = {P.kb, P.kb + cnt[0], P.kb + cnt[0] + cnt[1], P.kb + cnt[0] + cnt[1] + cnt[2]};
    {
      int pos[4] = {base[0], base[1], base[2], base[3]};
      // stable 4-way partition through a side buffer (reuse S.tmp upper half is not safe: use a local copy)
      static thread_local std::vector<int> side;
      side.assign(perm.begin() + P.kb, perm.begin() + P.ke);
      for (int k = P.kb; k < P.ke; k++) perm[pos[tmp[k]]++] = side[k - P.kb];
    }
    const int bx0[4] = {P.x0, xm, P.x0, xm}, bx1[4] = {xm, P.x1, xm, P.x1};
    const int by0[4] = {P.y0, P.y0, ym, ym}, by1[4] = {ym, ym, P.y1, P.y1};
    for (int q = 0; q < 4; q++) {
      if (cnt[q] == 0) continue;
      OctreeScratch::Node ch;
      ch.x0 = bx0[q];
      ch.x1 = bx1[q];
      ch.y0 = by0[q];
      ch.y1 = by1[q];
      ch.kb = base[q];
      ch.ke = base[q] + cnt[q];
      ch.no_more = cnt[q] == 1;
      ch.prev = ch.next = -1;
      nodes.push_back(ch);
      const int idx = (int)nodes.size() - 1;
      push_front(idx);
      if (cnt[q] > 1) {
        if (n_to_expand) (*n_to_expand)++;
        vsize.push_back(std::make_pair(cnt[q], idx));
      }
    }
  };
  auto cmp = [&](const std::pair<int, int>& a, const std::pair<int, int>& b) {  // compareNodes :552-565
    if (a.first < b.first) return true;
    if (a.first > b.first) return false;
    return nodes[a.second].x0 < nodes[b.second].x0;
  };

  bool finish = false;
  while (!finish) {
    int prev_size = size, n_to_expand = 0;
    vsize.clear();
    for (int it = head; it >= 0;) {
      if (nodes[it].no_more) {
        it = nodes[it].next;
        continue;
      }
      const int nx = nodes[it].next;
      split(it, &n_to_expand);
      erase(it);
      it = nx;
    }
    if (size >= N || size == prev_size) {
      finish = true;
    } else if (size + n_to_expand * 3 > N) {
      while (!finish) {
        prev_size = size;
        vprev = vsize;
        vsize.clear();
        std::sort(vprev.begin(), vprev.end(), cmp);
        for (int j = (int)vprev.size() - 1; j >= 0; j--) {
          split(vprev[j].second, nullptr);
          erase(vprev[j].second);
          if (size >= N) break;
        }
        if (size >= N || size == prev_size) finish = true;
      }
    }
  }
  for (int it = head; it >= 0; it = nodes[it].next) {  // :751-765
    int best = perm[nodes[it].kb], best_r = cand_score(c[best]);
    for (int k = nodes[it].kb + 1; k < nodes[it].ke; k++) {
      const int r = cand_score(c[perm[k]]);
      if (r > best_r) {
        best = perm[k];
        best_r = r;
      }
    }
    out.push_back(best);
  }
}

}  // namespace gfs

// ---- test hook: the launch order of k_blur7's tiles (OrbGeometry::build), host only ----
extern "C" int gfs_test_orb_blur_tiles(int rows, int cols, int nlevels, float scale_factor, int32_t* level_tx_ty, int cap) {
  gfs::OrbParams P;
  P.init(1000, scale_factor, nlevels, 20, 7, 1);
  gfs::OrbGeometry G;
  G.build(P, rows, cols);
  if (!G.supported) return -1;
  const int n = (int)G.blur_tiles.size();
  for (int i = 0; i < n && i < cap; i++) {
    level_tx_ty[3 * i] = G.blur_tiles[i].level;
    level_tx_ty[3 * i + 1] = G.blur_tiles[i].tx;
    level_tx_ty[3 * i + 2] = G.blur_tiles[i].ty;
  }
  return n;
}

// ---- test hook: the std::sort replica used by the device quadtree, run on the host against caller data ----
#include "std_sort_replica.hpp"
extern "C" int gfs_test_sort_replica(int32_t* size_key, int32_t* x_key, int32_t* payload, int n) {
  struct E {
    int s, x, p;
  };
  std::vector<E> v(n);
  for (int i = 0; i < n; i++) v[i] = E{size_key[i], x_key[i], payload[i]};
  gfs::replica_std_sort(v.data(), v.data() + n, [](const E& a, const E& b) {
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
extern "C" int gfs_test_heap_sort_replica(int32_t* size_key, int32_t* x_key, int32_t* payload, int n) {
  struct E {
    int s, x, p;
  };
  std::vector<E> v(n);
  for (int i = 0; i < n; i++) v[i] = E{size_key[i], x_key[i], payload[i]};
  gfs::replica_heap_sort(v.data(), v.data() + n, [](const E& a, const E& b) {
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
