// CPU check of the pyramid strip geometry k_pyr_area_lds relies on (csrc/orb_host.cpp, OrbGeometry::build), for a list of image
// sizes / scale factors / level counts: build with  g++ -std=c++17 -I geoflowslam_amd/csrc tests/host/orb_geometry_check.cpp
// geoflowslam_amd/csrc/orb_host.cpp  (tests/test_host_logic.py::test_pyramid_strip_geometry).  Prints one line per case; exit code
// = number of violated invariants.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orb_host.hpp"

using gfs::OrbGeometry;

static int check(int cols, int rows, float sf, int nl) {
  gfs::OrbParams p;
  p.init(1000, sf, nl, 20, 7, 0);
  OrbGeometry G;
  G.build(p, rows, cols);
  if (!G.supported) {
    printf("%dx%d sf=%.2f nl=%d unsupported: %s\n", cols, rows, sf, nl, G.why);
    return 0;
  }
  int bad = 0;
  auto fail = [&](const char* what, int a, int b) {
    printf("  VIOLATED %s (%d, %d)\n", what, a, b);
    bad++;
  };
  auto cut = [&](const std::vector<int>& tab, int S, size_t la, size_t lb, size_t lx, const char* name) {
    if (S <= 0) return;
    int max_tab = 0, max_prog = 0;
    for (int l = 1; l < nl; l++) {
      std::vector<int> cover(G.levels[l].rows, 0);
      for (int k = 0; k < S; k++) {
        const int a = tab[2 * (k * nl + l)], b = tab[2 * (k * nl + l) + 1];
        if (a < 0 || b > G.levels[l].rows || a > b) fail("row range inside the level", a, b);
        for (int r = a; r < b; r++) cover[r]++;
        // every row a strip produces reads source rows the same strip holds at level l - 1
        const int sa = tab[2 * (k * nl + l - 1)], sb = tab[2 * (k * nl + l - 1) + 1];
        for (int r = a; r < b; r++) {
          const int i = G.levels[l].ytab_off + r;
          if (G.yt_start[i] < sa || G.yt_start[i] + G.yt_n[i] > sb) fail("taps inside the strip's source rows", r, k);
        }
      }
      for (int r = 0; r < G.levels[l].rows; r++)
        if (cover[r] < 1) fail("every row of a level produced by some strip", l, r);
    }
    for (int k = 0; k < S; k++) {
      int rows_out = 0, rows_in = 0;
      size_t need_a = 0, need_b = 0;
      for (int l = 0; l < nl; l++) {
        const int n = tab[2 * (k * nl + l) + 1] - tab[2 * (k * nl + l)];
        if (l >= 1) rows_out += n;
        if (l < nl - 1) rows_in += n;
        const size_t bytes = (size_t)n * OrbGeometry::kPyrLdsPitch(G.levels[l].cols);
        if (l % 2 == 0) need_a = std::max(need_a, bytes);
        else need_b = std::max(need_b, bytes);
      }
      max_tab = std::max(max_tab, rows_out);
      max_prog = std::max(max_prog, rows_in);
      if (la + lb > 0 && (need_a > la || need_b > lb)) fail("strip fits its LDS buffer", k, (int)need_a);
    }
    if (la + lb > 0) {
      if (max_tab > OrbGeometry::kPyrTabRows) fail("destination rows of a strip <= kPyrTabRows", max_tab, S);
      if (max_prog > OrbGeometry::kPyrProgRows) fail("source rows of a strip <= kPyrProgRows", max_prog, S);
      if (la + lb + lx > OrbGeometry::kPyrLdsBudget && !getenv("GFS_ORB_PYR_LDS_KB")) fail("LDS budget", (int)(la + lb + lx), S);
      if (la % 16 || lb % 16) fail("buffers multiples of 16 bytes", (int)la, (int)lb);
    }
    printf("%dx%d sf=%.2f nl=%d %s: %d strips, LDS %zu + %zu + x %zu B, rows %d / %d\n", cols, rows, sf, nl, name, S, la, lb, lx, max_tab,
           max_prog);
  };
  cut(G.strip_rows, G.pyr_strips, G.pyr_lds_a, G.pyr_lds_b, G.pyr_lds_x, "cut");
  cut(G.strip_rows_fine, G.pyr_strips_fine, G.pyr_lds_a_fine, G.pyr_lds_b_fine, G.pyr_lds_x_fine, "fine cut");
  // consecutive rows of a level share at most their boundary source row, and two rows never end on one source row
  for (int l = 1; l < nl; l++)
    for (int r = 0; r + 1 < G.levels[l].rows; r++) {
      const int i = G.levels[l].ytab_off + r, last = G.yt_start[i] + G.yt_n[i] - 1;
      if (G.pyr_lds_a + G.pyr_lds_b > 0 && (G.yt_start[i + 1] < last || (G.yt_start[i + 1] == last && G.yt_n[i + 1] == 1)))
        fail("streamable y table", l, r);
    }
  if (G.pyr_lds_x)
    for (int n : G.xt_n)
      if (n > 3) fail("x tables in LDS only without a fourth tap", n, 0);
  // the descriptors carry their level's geometry
  for (const auto& c : G.cells)
    if (c.pitch != G.levels[c.level].pitch || c.plane_off != G.levels[c.level].plane_off) fail("cell descriptor", c.level, c.pitch);
  for (const auto& t : G.blur_tiles) {
    const auto& L = G.levels[t.level];
    if (t.rows != L.rows || t.cols != L.cols || t.pitch != L.pitch || t.plane_off != L.plane_off || t.blur_off != L.blur_off)
      fail("blur tile descriptor", t.level, t.pitch);
  }
  return bad;
}

int main() {
  int bad = 0;
  const int sizes[][2] = {{640, 480}, {1280, 720}, {1920, 1080}, {577, 411}, {322, 242}, {160, 120}, {801, 603}, {3000, 200}, {200, 3000}, {4000, 3000}};
  for (auto& s : sizes)
    for (float sf : {1.2f, 1.1f, 1.5f, 2.5f})
      for (int nl : {8, 4, 2}) bad += check(s[0], s[1], sf, nl);
  printf("violations: %d\n", bad);
  return bad > 255 ? 255 : bad;
}
