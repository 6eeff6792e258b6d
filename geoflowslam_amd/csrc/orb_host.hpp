// Host-side geometry / parameter tables of the ORB extractor and the (round-1) host quadtree distribution.
// Everything here is derived from the reference constructor and control flow:
//   src/ORBextractor.cc:421-479 (ctor tables), :770-806 (cell grid), :1227-1251 (level sizes),
//   :502-768 (DistributeOctTree), and OpenCV's computeResizeAreaTab (imgproc/src/resize.cpp).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace gfs {

constexpr int kPatchSize = 31;      // src/ORBextractor.cc:51
constexpr int kHalfPatch = 15;      // :52
constexpr int kEdgeThreshold = 19;  // :53
constexpr int kMaxAreaTaps = 4;     // INTER_AREA taps per destination pixel we support (level ratio < 3)

// Device-visible per-level geometry (POD, uploaded as an array).
struct LevelDev {
  int rows, cols, pitch;
  unsigned plane_off;  // byte offset of this level inside one frame's pyramid buffer (level 0 unused)
  unsigned blur_off;   // byte offset inside one frame's blurred-pyramid buffer
  int xtab_off, ytab_off;  // first entry of this level's INTER_AREA tables (level >= 1)
  int w_cell, h_cell, n_cell_cols, n_cell_rows, cell_base, n_cells;
  int max_bx, max_by;  // maxBorderX / maxBorderY (level coords)
  float scale;         // mvScaleFactor[level]
  float patch_size;    // (float)(int)(PATCH_SIZE * mvScaleFactor[level])
  int quota;           // mnFeaturesPerLevel[level]
  int kp_cap;          // upper bound of keypoints this level can emit
};

// One FAST cell (src/ORBextractor.cc:788-806): the sub-image rowRange(y0,y1).colRange(x0,x1).
struct CellDev {
  short level, x0, y0, w, h;
  short pad;
  unsigned slab_off;  // first candidate slot of this cell in one frame's slab buffer
  unsigned slab_cap;
  // the level's plane (copies of LevelDev::pitch / plane_off: k_fast_cells reaches its pixels behind ONE descriptor load)
  int pitch;
  unsigned plane_off;
  unsigned pad2;
};

struct BlurTileDev {
  short level, tx, ty, pad;
  // the level (copies of LevelDev::rows / cols / pitch / plane_off / blur_off: one descriptor load in k_blur7)
  int rows, cols, pitch;
  unsigned plane_off, blur_off;
  unsigned pad2;
};

struct OrbParams {
  int nfeatures, nlevels, ini_th, min_th, blur_variant;
  float scale_factor;
  std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
  std::vector<int> quota;
  int umax[16];
  void init(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int blur_variant);
};

struct OrbGeometry {
  int rows = 0, cols = 0;
  std::vector<LevelDev> levels;
  std::vector<CellDev> cells;
  std::vector<BlurTileDev> blur_tiles;
  // INTER_AREA tables, concatenated over levels >= 1: entry i covers source indices
  // [start[i], start[i] + n[i]) with weights alpha[4*i .. 4*i+n[i])
  std::vector<int> xt_start, yt_start;
  std::vector<int> xt_n, yt_n;
  std::vector<float> xt_alpha, yt_alpha;
  // fused pyramid kernel: rows [strip_rows[2*(k*nlevels+l)], strip_rows[2*(k*nlevels+l)+1]) of level l held by strip k: for
  // l >= 1 the rows it computes (its own share of the level plus the halo its higher levels read), for l = 0 the source rows
  // level 1 reads.  pyr_lds_a / pyr_lds_b: bytes of the two LDS ping-pong buffers (even / odd levels; rows of kPyrLdsPitch(cols)
  // bytes); 0 = does not fit.  pyr_lds_x: bytes of the x tables of all levels kept in LDS behind them (16 bytes a column: three
  // weights + the first source column; only when no column has a fourth tap), 0 = read from memory.  A cut only counts as
  // fitting when the rows a strip produces over all levels >= 1 also fit the kernel's table of row taps (kPyrTabRows).  The
  // cut is the coarsest one that fits with the x tables, else the coarsest that fits without.
  static constexpr int kPyrTabRows = 384, kPyrProgRows = 512;  // destination rows / source rows of a strip over all levels
  static constexpr size_t kPyrLdsBudget = 150 * 1024;
  static constexpr int kPyrLdsPitch(int cols) { return (cols + 3) & ~3; }
  size_t pyr_lds_x = 0, pyr_lds_x_fine = 0;
  std::vector<int> strip_rows;
  int pyr_strips = 0;
  size_t pyr_lds_a = 0, pyr_lds_b = 0;
  // the same for a finer cut (about four times the strips): what a launch of a few frames uses -- with one frame the eight
  // workgroups of the coarse cut each walk all levels of a tall strip one after the other on an otherwise idle chip.  Same rows, same
  // arithmetic; the halo rows are recomputed by more workgroups.  pyr_strips_fine = 0: none (the coarse cut is already the finest).
  std::vector<int> strip_rows_fine;
  int pyr_strips_fine = 0;
  size_t pyr_lds_a_fine = 0, pyr_lds_b_fine = 0;
  static constexpr int kPyrMaxStrips = 64;
  size_t pyr_bytes = 0, blur_bytes = 0;  // per frame
  size_t slab_entries = 0;               // per frame
  size_t cand_cap = 0;                   // dense candidate list capacity per frame (== slab_entries)
  int kp_cap = 0;                        // keypoint capacity per frame
  int max_tile_w = 0, max_tile_h = 0;    // FAST cell tile bounds
  size_t fast_lds_wave = 0;              // LDS bytes of one cell in k_fast_cells (tile + score map + offset list), the largest cell's
  bool supported = true;
  const char* why = "";
  void build(const OrbParams& p, int rows, int cols);
};

// IC_Angle sampling offsets (src/ORBextractor.cc:71-95): 749 (du, dv) pairs of the circular patch.
void ic_angle_offsets(const int umax[16], std::vector<int8_t>& du, std::vector<int8_t>& dv);

// DistributeOctTree (src/ORBextractor.cc:567-768) on integer candidates (x, y relative to (16,16), response).
// Array / index based: same node order, same libstdc++ std::sort tie behaviour, same first-max selection.
// Appends the kept candidate indices, in std::list order, to `out`.
struct OctreeScratch {
  std::vector<int> perm, tmp;
  struct Node {
    int x0, x1, y0, y1;
    int kb, ke;
    int prev, next;
    bool no_more;
  };
  std::vector<Node> nodes;
};
void distribute_octree(const uint32_t* packed, int n, int min_x, int max_x, int min_y, int max_y, int N,
                       OctreeScratch& scratch, std::vector<int>& out);

inline int cand_x(uint32_t c) { return (int)(c & 0xfff); }
inline int cand_y(uint32_t c) { return (int)((c >> 12) & 0xfff); }
inline int cand_score(uint32_t c) { return (int)(c >> 24); }

}  // namespace gfs
