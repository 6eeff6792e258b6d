// GICP registration for gfx950 (MI355X).  Replaces RegistrationGICP::RegisterPointClouds
// (reference src/RegistrationGICP.cc:5-20 -> small_gicp::align<float,4>, registration_helper.cpp:57-122):
//   voxel-grid downsampling (util/downsampling_omp.hpp:26-95)  -> k_voxel_keys, k_radix_sort, k_voxel_reduce
//   exact k-NN (ann/kdtree.hpp) + covariances (util/normal_estimation.hpp:66-92) -> cell grid + k_knn_cov
//   GICPFactor::linearize / error (factors/gicp_factor.hpp:35-89) + ParallelReductionOMP
//   (registration/reduction_omp.hpp:21-69)                   -> k_gicp_linearize (a trial's error and the linearisation at the
//                                                              trial pose in ONE pass since round 6; + fixed-order sums)
//   LevenbergMarquardtOptimizer::optimize (registration/optimizer.hpp:83-147) -> device-side state machine
//                                                              (k_gicp_step); the host only polls "all done"; a small batch runs its
//                                                              whole loop in one launch (k_gicp_lm_coop).
// Everything is double precision like the reference.  The KdTree is replaced by a uniform cell grid (cell edge =
// max correspondence distance): nearest-neighbour results are exact (certified ring by ring), hence structure independent (ties aside).
//
// HBM layout: a batch holds 2B clouds (cloud c = 2*pair + {0: target, 1: source}), every per-cloud array has a
// fixed stride of P entries.  Down-sampled points are stored sorted by (cell z, cell y, cell x) so that one cell
// and its two x-neighbours form a contiguous run: a 27-cell probe is 9 binary searches + 9 linear runs.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <memory>
#include <type_traits>

#include "gfs_common.hpp"
#include "glibc_math.hpp"
#include "voxel_qsort.hpp"
#include "wave_reduce.hpp"

namespace {

typedef unsigned long long u64;
constexpr u64 kInvalidKey = ~0ull;
constexpr int kCoordBits = 21;
constexpr int kCoordOffset = 1 << (kCoordBits - 1);
constexpr int kCoordMask = (1 << kCoordBits) - 1;
// The CELL sort key (k_voxel_reduce -> k_radix_sort -> k_cell_build) orders the points of a cell by x as well: its x field counts
// 1 / kFine of a cell (sixty-fourths since round 6).  Fields: x fine 25 bits | y 20 bits | z 19 bits (|cell coordinate| < 2^18: k_voxel_keys' max_vox sees to it).
// Sub-cell bits of a cell sort key's x field: the points of a row of cells are ordered by 1 / 2^bits of a cell along x, and the walk of
// the linearisation's 1-NN search (nn_sweep) stops one such slice beyond its bound.  Measured (round 6, k_gicp_linearize per 512-pair
// step, serial): 3 bits 3.37 ms, 4 (rounds 3 - 5) 3.25, 5: 3.18, 6: 3.14 -- 6 is what the 25-bit field holds beside 19 bits of cell.
#ifndef GFS_FINE_BITS
#define GFS_FINE_BITS 6
#endif
constexpr int kFineBits = GFS_FINE_BITS, kFine = 1 << kFineBits;
static_assert(kFineBits >= 1 && kFineBits <= 6, "the x field of a cell sort key: 19 bits of cell + the sub-cell bits in 25");
constexpr int kCkSy = 25, kCkSz = 45;  // bit positions of the y and z fields of a cell sort key
constexpr int kCkOffX = 1 << 24, kCkOffY = 1 << 19, kCkOffZ = 1 << 18;
constexpr int kRed = 29;  // 21 (upper H) + 6 (b) + 1 (e) + 1 (inlier count)
constexpr int kLinBlock = 256;

struct PairState {
  double T[12];     // R (col-major 9) | t (3): current estimate
  double newT[12];  // trial estimate
  double delta[6];
  double H[21], b[6], e;  // last linearisation (upper triangle, (i,j) i<=j in row-major order)
  double lambda;
  int outer, inner;
  // 0 = first linearisation (at T) next; 1 = a trial pose (newT) is pending: the next pass evaluates its error with the frozen
  // correspondences AND linearises at newT into the other correspondence buffer -- if the trial is accepted that IS the next
  // linearisation (round 6: one pass and one scalar step per LM trial instead of two each); 3 = a trial is pending whose acceptance
  // ends the loop (converged step, or the last iteration): error only; 2 = done
  int phase;
  int converged;
  int iterations;  // RegistrationResult::iterations (index of the last outer iteration)
  int inliers;
  int n_lin, n_err;
  int buf;  // which of the two (tgt_index, maha6) buffers holds the correspondences of the last accepted linearisation
  int pad_;
};

struct GicpParams {
  double inv_leaf, cell, inv_cell, max_dist_sq, rot_eps, trans_eps;
  int max_iterations, k_neighbors;
  unsigned* tile_stats;  // optional [8]: workgroups of k_gicp_linearize by outcome of the tile staging (0 = tiled), diagnostics
  int lin_tile;  // k_gicp_linearize stages its target tile in LDS (GFS_GICP_LIN_TILE=0 switches it off)
  int nn_rings;  // ceil(max_corr / cell): rings of cells a 1-NN probe may need to certify "nothing within max_corr"
  // Cloud slots of pair b are 2b and 2b + 1.  src_slot: which of the two holds the SOURCE cloud (1 in the plain entry points;
  // the streaming entry alternates so that the previous call's preprocessed source becomes this call's target in place).
  // only: -1 = preprocess both slots, else only the slot of that parity.
  int src_slot, only;
  int knn_exact;  // GFS_GICP_KNN_EXACT=1 (test knob): k_knn_cov certifies nothing, every query takes the exact deferred passes
};

__device__ __forceinline__ int fast_floor_d(double v) {  // util/fast_floor.hpp:12-15
  const int n = (int)v;
  return n - (v < (double)n ? 1 : 0);
}
__device__ __forceinline__ u64 pack_key(int cx, int cy, int cz) {
  return (u64)(cx & kCoordMask) | ((u64)(cy & kCoordMask) << kCoordBits) | ((u64)(cz & kCoordMask) << (2 * kCoordBits));
}

// ------------------------------------------------------------------------------------------------
// k_voxel_keys (util/downsampling_omp.hpp:39-55; points/point_cloud.hpp:26-31 widens float -> double, w := 1)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_voxel_keys(const float4* __restrict__ tgt, const float4* __restrict__ src,
                                                    const int* __restrict__ nt, const int* __restrict__ ns,
                                                    int stride_pts, int P, double inv_leaf, double max_vox, u64* __restrict__ keys,
                                                    unsigned* __restrict__ idx, int* __restrict__ counts, int only,
                                                    int* __restrict__ n_done4, int* __restrict__ n_heap, int* __restrict__ n_active4) {
  const int c = blockIdx.y, pair = c >> 1, which = c & 1;
  // the first kernel of a call also zeroes the call's counters (round 6: three memset launches in front of it before): the done /
  // left-over / gave-up counters, the per-cloud heap-range counts of the voxel sort, the active-pair counts of the LM rounds
  if (blockIdx.x == 0 && threadIdx.x < 4) {
    if (threadIdx.x == 0) n_heap[c] = 0;
    if (c == 0) {
      n_done4[threadIdx.x] = 0;
      n_active4[threadIdx.x] = 0;
    }
  }
  if (only >= 0 && which != only) return;
  const int n = min(which ? ns[pair] : nt[pair], P);
  const float4* in = (which ? src : tgt) + (size_t)pair * stride_pts;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) counts[c] = n;
  if (i >= n) return;
  const float4 p = in[i];
  const double x = (double)p.x * inv_leaf, y = (double)p.y * inv_leaf, z = (double)p.z * inv_leaf;
  u64 key = kInvalidKey;
  // max_vox: 1e6 voxels (inside the 21-bit fields), less when the cell is so small against the leaf that a coordinate the voxel
  // fields admit would leave the fields of the CELL sort key (kCkOff*; gicp_run) -- such points are dropped like out-of-range ones
  if (fabs(x) < max_vox && fabs(y) < max_vox && fabs(z) < max_vox) {  // also rejects NaN / inf
    const int cx = fast_floor_d(x) + kCoordOffset, cy = fast_floor_d(y) + kCoordOffset, cz = fast_floor_d(z) + kCoordOffset;
    if (cx >= 0 && cx <= kCoordMask && cy >= 0 && cy <= kCoordMask && cz >= 0 && cz <= kCoordMask) key = pack_key(cx, cy, cz);
  }
  keys[(size_t)c * P + i] = key;
  idx[(size_t)c * P + i] = (unsigned)i;
}

// ------------------------------------------------------------------------------------------------
// k_radix_sort: one 1024-thread workgroup sorts one cloud's (key, value) pairs.  LSD radix, 9-bit digits,
// stable (4096-element tiles scattered in order; every wave ranks its 4 x 64 consecutive elements by wave ballots and
// a running per-wave digit count), uniform digits skipped.  Ping-pongs between buffers 0 and 1; which[c] says where
// the sorted data ended up.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_radix_sort(u64* __restrict__ keys0, u64* __restrict__ keys1,
                                                     unsigned* __restrict__ val0, unsigned* __restrict__ val1,
                                                     const int* __restrict__ counts, int P, int* __restrict__ which,
                                                     int* __restrict__ kinfo, int only, int sy, int sz) {
  // sy, sz: bit positions of the second and third field of a key (21 / 42 for voxel keys, kCkSy / kCkSz for cell sort keys)
  constexpr int kDB = 9, kNB = 1 << kDB;  // digit bits, bins
  constexpr int kEl = 4;                  // elements per thread and tile
  __shared__ unsigned hist[kNB];
  __shared__ unsigned bin_base[kNB];
  __shared__ unsigned wave_hist[16][kNB];
  __shared__ int s_uniform;
  __shared__ unsigned s_wtot[8];
  __shared__ int s_mn[3], s_mx[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (only >= 0 && (c & 1) != only) return;
  const int n = counts[c];
  u64* ka = keys0 + (size_t)c * P;
  u64* kb = keys1 + (size_t)c * P;
  unsigned* va = val0 + (size_t)c * P;
  unsigned* vb = val1 + (size_t)c * P;
  int flip = 0;
  // Key compaction: the keys pack three coordinate fields (x | y << sy | z << sz) but one cloud spans only a
  // few hundred cells per axis.  Re-basing every field to the cloud's minimum and packing the fields tightly is order
  // preserving and cuts the 8 radix passes to ceil((bx + by + bz) / 8) (typically 4).  kinfo = {xmin, ymin, zmin, bx, by}
  // lets later kernels decode; equality and the invalid key (all ones) are preserved.
  if (tid < 3) {
    s_mn[tid] = 0x7fffffff;
    s_mx[tid] = -1;
  }
  __syncthreads();
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
    gfs::strided_batch<8>(ka, tid, 1024, n, [&](int, u64 k) {
      if (k == kInvalidKey) return;
      const int f[3] = {(int)(k & ((1ull << sy) - 1)), (int)((k >> sy) & ((1ull << (sz - sy)) - 1)), (int)(k >> sz)};
      for (int a = 0; a < 3; a++) {
        mn[a] = min(mn[a], f[a]);
        mx[a] = max(mx[a], f[a]);
      }
    });
    for (int a = 0; a < 3; a++) {  // one LDS atomic a wave
      mn[a] = gfs::wave_min_i32(mn[a]);
      mx[a] = gfs::wave_max_i32(mx[a]);
      if ((tid & 63) == 0 && mx[a] >= 0) {
        atomicMin(&s_mn[a], mn[a]);
        atomicMax(&s_mx[a], mx[a]);
      }
    }
  }
  __syncthreads();
  const bool any_valid = s_mx[0] >= 0;
  const int mnx = any_valid ? s_mn[0] : 0, mny = any_valid ? s_mn[1] : 0, mnz = any_valid ? s_mn[2] : 0;
  auto nbits = [](int range) {
    int b = 1;
    while ((1 << b) <= range) b++;
    return b;
  };
  const int bx = any_valid ? nbits(s_mx[0] - mnx) : 1, by = any_valid ? nbits(s_mx[1] - mny) : 1,
            bz = any_valid ? nbits(s_mx[2] - mnz) : 1;
  gfs::strided_batch<8>(ka, tid, 1024, n, [&](int i, u64 k) {
    if (k == kInvalidKey) return;
    const u64 x = (k & ((1ull << sy) - 1)) - mnx, y = ((k >> sy) & ((1ull << (sz - sy)) - 1)) - mny, z = (k >> sz) - mnz;
    ka[i] = x | (y << bx) | (z << (bx + by));
  });
  if (tid == 0) {
    int* ki = kinfo + 8 * c;
    ki[0] = mnx;
    ki[1] = mny;
    ki[2] = mnz;
    ki[3] = bx;
    ki[4] = by;
  }
  __syncthreads();
  // valid keys have bx + by + bz bits; invalid keys (all ones) must still sort last: one more pass over a digit above
  // the packed width places them (skipped as uniform when there is no invalid key)
  const int npass = (bx + by + bz + kDB - 1) / kDB;  // <= 7
  for (int pass = 0; pass <= npass && n > 0; pass++) {
    const int shift = kDB * pass;
    for (int k = tid; k < kNB; k += 1024) hist[k] = 0;
    if (tid == 0) s_uniform = 0;
    __syncthreads();
    gfs::strided_batch<8>(ka, tid, 1024, n, [&](int, u64 k) { atomicAdd(&hist[(unsigned)(k >> shift) & (kNB - 1)], 1u); });
    __syncthreads();
    if (tid < kNB && hist[tid] == (unsigned)n) s_uniform = 1;
    __syncthreads();
    if (s_uniform) continue;  // block-uniform
    {  // exclusive scan of the 512 bin counts: a shuffle scan per wave of 64 bins + the totals of the 8 waves (two barriers)
      const unsigned own = tid < kNB ? hist[tid] : 0u;
      unsigned incl = own;
#pragma unroll
      for (int ofs = 1; ofs < 64; ofs <<= 1) {
        const unsigned v = __shfl_up(incl, ofs, 64);
        if (lane >= ofs) incl += v;
      }
      if (tid < kNB && lane == 63) s_wtot[wave] = incl;
      __syncthreads();
      if (tid < kNB) {
        unsigned base = 0;
        for (int w = 0; w < wave; w++) base += s_wtot[w];
        bin_base[tid] = base + incl - own;
      }
      __syncthreads();
    }
    for (int t0 = 0; t0 < n; t0 += 1024 * kEl) {
      for (int k = tid; k < 16 * kNB; k += 1024) (&wave_hist[0][0])[k] = 0;
      __syncthreads();
      u64 key[kEl];
      unsigned val[kEl], digit[kEl], rank[kEl];
      bool valid[kEl];
#pragma unroll
      for (int e = 0; e < kEl; e++) {  // wave w owns elements [t0 + 256 w, t0 + 256 (w + 1)) in index order
        const int i = t0 + wave * (64 * kEl) + e * 64 + lane;
        valid[e] = i < n;
        key[e] = 0;
        val[e] = 0;
        if (valid[e]) {
          key[e] = ka[i];
          val[e] = va[i];
        }
        digit[e] = (unsigned)(key[e] >> shift) & (kNB - 1);
      }
#pragma unroll
      for (int e = 0; e < kEl; e++) {
        u64 mask = __ballot(valid[e]);
#pragma unroll
        for (int bit = 0; bit < kDB; bit++) {
          const bool b1 = (digit[e] >> bit) & 1u;
          const u64 bm = __ballot(b1);
          mask &= b1 ? bm : ~bm;
        }
        const unsigned lane_rank = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned before = valid[e] ? wave_hist[wave][digit[e]] : 0u;  // same digit, earlier 64-element groups of this wave
        rank[e] = before + lane_rank;
        if (valid[e] && lane_rank == 0) wave_hist[wave][digit[e]] = before + (unsigned)__popcll(mask);
      }
      __syncthreads();
      if (tid < kNB) {
        unsigned acc = bin_base[tid];
        for (int w = 0; w < 16; w++) {
          const unsigned t = wave_hist[w][tid];
          wave_hist[w][tid] = acc;
          acc += t;
        }
        bin_base[tid] = acc;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < kEl; e++) {
        if (valid[e]) {
          const unsigned pos = wave_hist[wave][digit[e]] + rank[e];
          kb[pos] = key[e];
          vb[pos] = val[e];
        }
      }
      __syncthreads();
    }
    u64* tk = ka;
    ka = kb;
    kb = tk;
    unsigned* tv = va;
    va = vb;
    vb = tv;
    flip ^= 1;
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) which[c] = flip;
}

// ------------------------------------------------------------------------------------------------
// k_cell_sort_lds: the CELL sort of a cloud of at most kCsE * 1024 means (a VGA depth image sampled with stride 4) with the
// cloud RESIDENT IN LDS for all passes: 4-byte compacted keys (the three fields re-based to the cloud's extent, like
// k_radix_sort) + 2-byte indices = 6 bytes an element, 120 KB, + the sixteen per-wave digit histograms (32 KB).  Same stable LSD
// radix (9-bit digits), same permutation as k_radix_sort; the four passes through HBM of that kernel (2.1 GB moved for 0.42 GB of
// keys per 1 024 clouds, r03f_pmc_traffic) become one read and one write.  A pass works in place: every thread first takes ITS
// elements into registers -- wave w owns the contiguous chunk [w * chunk, (w + 1) * chunk) and walks it in rows of 64 lanes, so
// the stable order is (wave, row, lane) --, the waves count their digits (wave ballots, one histogram row a wave), one scan over
// (digit, wave) turns the counts into start positions, and the second walk over the same registers ranks and scatters.  Three
// barriers a pass, none inside the walks.  Clouds it cannot take (keys wider than 32 bits, more than kCsE * 1024 means) are
// counted in *n_left: the caller comes back with k_radix_sort (gicp_run, like the voxel sort's optimistic launch).
// ------------------------------------------------------------------------------------------------
constexpr int kCsE = 20, kCsRows = kCsE * 1024 / (16 * 64);  // rows of 64 elements per wave chunk: 20
constexpr int kCsLdsBytes = kCsE * 1024 * 6 + 16 * 512 * 4;

__global__ __launch_bounds__(1024) void k_cell_sort_lds(u64* __restrict__ keys0, unsigned* __restrict__ val0,
                                                        int* __restrict__ counts, int P, int* __restrict__ which,
                                                        int* __restrict__ kinfo, int only, int* __restrict__ n_left) {
  constexpr int kDB = 9, kNB = 1 << kDB;
  extern __shared__ __align__(16) unsigned char cs_lds[];
  unsigned* s_key = reinterpret_cast<unsigned*>(cs_lds);                               // [kCsE * 1024]
  unsigned short* s_val = reinterpret_cast<unsigned short*>(s_key + kCsE * 1024);      // [kCsE * 1024]
  unsigned* s_hist = reinterpret_cast<unsigned*>(s_val + kCsE * 1024);                 // [16][kNB]
  __shared__ int s_mn[3], s_mx[3];
  __shared__ unsigned s_wtot[8];
  __shared__ int s_uniform;
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (only >= 0 && (c & 1) != only) return;
  const int n = __builtin_amdgcn_readfirstlane(counts[c]);
  u64* ka = keys0 + (size_t)c * P;
  unsigned* va = val0 + (size_t)c * P;
  if (tid < 3) {
    s_mn[tid] = 0x7fffffff;
    s_mx[tid] = -1;
  }
  __syncthreads();
  // wave w owns [w * chunk, (w + 1) * chunk), chunk a multiple of 64; its row r is elements w * chunk + 64 r + lane
  const int chunk = ((n + 16 * 64 - 1) / (16 * 64)) * 64, rows = chunk >> 6;
  const int base = wave * chunk;
  u64 kreg[kCsRows];
#pragma unroll
  for (int r = 0; r < kCsRows; r++) {
    const int i = base + 64 * r + lane;
    kreg[r] = (r < rows && i < n) ? ka[i] : kInvalidKey;
  }
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
#pragma unroll
    for (int r = 0; r < kCsRows; r++) {
      const u64 k = kreg[r];
      if (k == kInvalidKey) continue;
      const int f[3] = {(int)(k & ((1ull << kCkSy) - 1)), (int)((k >> kCkSy) & ((1ull << (kCkSz - kCkSy)) - 1)), (int)(k >> kCkSz)};
      for (int a = 0; a < 3; a++) {
        mn[a] = min(mn[a], f[a]);
        mx[a] = max(mx[a], f[a]);
      }
    }
    for (int a = 0; a < 3; a++) {  // one LDS atomic a wave
      mn[a] = gfs::wave_min_i32(mn[a]);
      mx[a] = gfs::wave_max_i32(mx[a]);
      if (lane == 0 && mx[a] >= 0) {
        atomicMin(&s_mn[a], mn[a]);
        atomicMax(&s_mx[a], mx[a]);
      }
    }
  }
  __syncthreads();
  const bool any_valid = s_mx[0] >= 0;
  const int mnx = any_valid ? s_mn[0] : 0, mny = any_valid ? s_mn[1] : 0, mnz = any_valid ? s_mn[2] : 0;
  auto nbits = [](int range) {
    int b = 1;
    while ((1 << b) <= range) b++;
    return b;
  };
  const int bx = any_valid ? nbits(s_mx[0] - mnx) : 1, by = any_valid ? nbits(s_mx[1] - mny) : 1,
            bz = any_valid ? nbits(s_mx[2] - mnz) : 1;
  if (bx + by + bz > 32 || n > kCsE * 1024) {  // uniform: left to k_radix_sort (the caller launches it when the count says so)
    // The kernels queued behind this one (cell build, grid fill, k-NN, the first LM rounds) run before the host sees *n_left: they
    // must not walk an UNSORTED key array with stale bit widths.  The cloud is emptied for them (an empty cloud is a regular input);
    // the caller's rerun (gicp_run, sort_all_kernels) rebuilds everything from the input points, this count included.
    if (tid == 0) {
      atomicAdd(n_left, 1);
      which[c] = 0;
      counts[c] = 0;
      int* ki = kinfo + 8 * c;
      ki[0] = ki[1] = ki[2] = 0;
      ki[3] = ki[4] = 1;
    }
    return;
  }
  unsigned key[kCsRows];
  unsigned short val[kCsRows];
#pragma unroll
  for (int r = 0; r < kCsRows; r++) {
    const u64 k = kreg[r];
    key[r] = 0xffffffffu;  // (a cell key is never the invalid key: every mean is a valid point)
    if (k != kInvalidKey)
      key[r] = (unsigned)((k & ((1ull << kCkSy) - 1)) - mnx) | ((unsigned)(((k >> kCkSy) & ((1ull << (kCkSz - kCkSy)) - 1)) - mny) << bx) |
               ((unsigned)((k >> kCkSz) - mnz) << (bx + by));
    val[r] = (unsigned short)(base + 64 * r + lane);  // the indices enter as the identity (k_voxel_reduce)
  }
  if (tid == 0) {
    int* ki = kinfo + 8 * c;
    ki[0] = mnx;
    ki[1] = mny;
    ki[2] = mnz;
    ki[3] = bx;
    ki[4] = by;
    which[c] = 0;  // the sorted keys go back to buffer 0
  }
  const int npass = (bx + by + bz + kDB - 1) / kDB;
  unsigned* myhist = s_hist + wave * kNB;
  const u64 ltm = (1ull << lane) - 1ull;
  for (int pass = 0; pass < npass && n > 0; pass++) {
    const int shift = kDB * pass;
    if (pass > 0) {  // (pass 0 works on the registers of the load)
#pragma unroll
      for (int r = 0; r < kCsRows; r++) {
        const int i = base + 64 * r + lane;
        if (r < rows && i < n) {
          key[r] = s_key[i];
          val[r] = s_val[i];
        }
      }
    }
    for (int k = lane; k < kNB; k += 64) myhist[k] = 0;
    if (tid == 0) s_uniform = 0;
    // (A) digits of the wave's chunk: per row the lanes with one digit find each other by ballots, their first lane counts them
    auto same_digit = [&](bool valid, unsigned digit) {  // the valid lanes of the row that hold this lane's digit
      u64 mask = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < kDB; bit++) {
        const bool b1 = (digit >> bit) & 1u;
        const u64 bm = __ballot(b1);
        mask &= b1 ? bm : ~bm;
      }
      return mask;
    };
#pragma unroll
    for (int r = 0; r < kCsRows; r++) {
      if (r >= rows) continue;
      const bool valid = base + 64 * r + lane < n;
      const unsigned digit = (key[r] >> shift) & (kNB - 1);
      const u64 mask = same_digit(valid, digit);
      if (valid && (mask & ltm) == 0) myhist[digit] += (unsigned)__popcll(mask);  // one lane per distinct digit of the row
    }
    __syncthreads();
    // (B) start position of (digit, wave): exclusive scan over the digits of the totals, then over the waves inside a digit
    unsigned own = 0;
    if (tid < kNB) {
      for (int w = 0; w < 16; w++) own += s_hist[w * kNB + tid];
      if (own == (unsigned)n) s_uniform = 1;
    }
    unsigned incl = own;
#pragma unroll
    for (int ofs = 1; ofs < 64; ofs <<= 1) {
      const unsigned v = __shfl_up(incl, ofs, 64);
      if (lane >= ofs) incl += v;
    }
    if (tid < kNB && lane == 63) s_wtot[wave] = incl;
    __syncthreads();
    if (s_uniform) {  // every key has this digit: nothing moves (block-uniform); the keys stay where they are
      if (pass == 0) {
#pragma unroll
        for (int r = 0; r < kCsRows; r++) {
          const int i = base + 64 * r + lane;
          if (r < rows && i < n) {
            s_key[i] = key[r];
            s_val[i] = val[r];
          }
        }
      }
      __syncthreads();
      continue;
    }
    if (tid < kNB) {
      unsigned acc = incl - own;
      for (int w = 0; w < wave; w++) acc += s_wtot[w];
      for (int w = 0; w < 16; w++) {
        const unsigned t = s_hist[w * kNB + tid];
        s_hist[w * kNB + tid] = acc;
        acc += t;
      }
    }
    __syncthreads();
    // (C) rank and scatter: the wave walks its chunk again, its histogram row is now the running position of every digit
#pragma unroll
    for (int r = 0; r < kCsRows; r++) {
      if (r >= rows) continue;
      const bool valid = base + 64 * r + lane < n;
      const unsigned digit = (key[r] >> shift) & (kNB - 1);
      const u64 mask = same_digit(valid, digit);
      const unsigned before = valid ? myhist[digit] : 0u;
      const unsigned lane_rank = (unsigned)__popcll(mask & ltm);
      if (valid) {
        const unsigned pos = before + lane_rank;
        s_key[pos] = key[r];
        s_val[pos] = val[r];
        if (lane_rank == 0) myhist[digit] = before + (unsigned)__popcll(mask);
      }
    }
    __syncthreads();
  }
  // back to HBM as 64-bit compacted keys + 32-bit indices (k_cell_build)
  for (int i = tid; i < n; i += 1024) {
    ka[i] = (u64)s_key[i];
    va[i] = (unsigned)s_val[i];
  }
}

// block-wide exclusive scan of one int per thread (1024 threads); returns exclusive prefix, total in *total
__device__ __forceinline__ int block_scan_1024(int v, int* s_wave /*16*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const int t = __shfl_up(incl, ofs, 64);
    if (lane >= ofs) incl += t;
  }
  __syncthreads();
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < 16; w++) {
    const int t = s_wave[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

// ------------------------------------------------------------------------------------------------
// k_voxel_reduce: per sorted run of equal voxel keys, cut additionally at every multiple of 1024 in the
// sorted array (the reference's block-wise reduction, util/downsampling_omp.hpp:63-90), emit the mean
// sum / sum.w.  Also emits the cell key of every mean.  One workgroup per cloud; output in sorted order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_voxel_reduce(const float4* __restrict__ tgt, const float4* __restrict__ src,
                                                       int stride_pts, const u64* __restrict__ keys0,
                                                       const u64* __restrict__ keys1, const unsigned* __restrict__ val0,
                                                       const unsigned* __restrict__ val1, const int* __restrict__ which,
                                                       const int* __restrict__ counts, int P, double inv_cell,
                                                       double4* __restrict__ tmp_pts, u64* __restrict__ cell_keys,
                                                       unsigned* __restrict__ cell_idx, int* __restrict__ m_counts, int only,
                                                       int tiles_per_part, int* __restrict__ bbox) {
  // gridDim.y workgroups a cloud (small batches: one workgroup a cloud leaves the chip idle): part p takes the 1024-element tiles
  // [p * tiles_per_part, (p + 1) * tiles_per_part) and first counts the means the tiles in front of it emit -- the same numbers,
  // the same output positions as a single workgroup walking the whole array
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  __shared__ float s_px[1024], s_py[1024], s_pz[1024];
  __shared__ unsigned char s_stop[1024];
  const int c = blockIdx.x, pair = c >> 1, tid = threadIdx.x, part = blockIdx.y;
  if (only >= 0 && (c & 1) != only) return;
  const int n = counts[c];
  if (part == 0 && tid < 6) bbox[6 * c + tid] = tid < 3 ? kCoordMask : 0;  // k_cell_build's workgroups reduce into it
  const u64* keys = (which[c] ? keys1 : keys0) + (size_t)c * P;
  const unsigned* idx = (which[c] ? val1 : val0) + (size_t)c * P;
  const float4* in = ((c & 1) ? src : tgt) + (size_t)pair * stride_pts;
  double4* out = tmp_pts + (size_t)c * P;
  u64* ck = cell_keys + (size_t)c * P;
  unsigned* ci = cell_idx + (size_t)c * P;
  const int e0 = min(part * tiles_per_part * 1024, n), e1 = part + 1 == (int)gridDim.y ? n : min((part + 1) * tiles_per_part * 1024, n);
  {
    int before = 0;  // means emitted by [0, e0): a run start = a valid key that differs from its predecessor or sits on a 1024-cut
    for (int i = tid; i < e0; i += 8 * 1024) {  // (thread 0 sits on the 1024-cuts: i & 1023 == tid; e0 is a multiple of 1024)
      u64 k[8], kp[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int iu = min(i + u * 1024, e0 - 1);
        k[u] = keys[iu];
        kp[u] = keys[max(iu - 1, 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        before += (i + u * 1024 < e0 && k[u] != kInvalidKey && (tid == 0 || kp[u] != k[u])) ? 1 : 0;
    }
    int total;
    (void)block_scan_1024(before, s_wave, &total);
    __syncthreads();
    if (tid == 0) s_carry = total;
  }
  __syncthreads();
  // A tile's points are fetched by all its threads at once (sorted index -> point: two round trips a tile) into LDS; the thread
  // at a run's start then adds its run up from there, in the order of the sorted array as before (the same doubles).  Walking the
  // run through memory -- index, point, next key: two dependent round trips per point, the wave waiting for its longest run --
  // made a tile ~17 k cycles.
  // ... and the loads run two tiles ahead of the sums: a tile's sorted indices are asked for two iterations early, its keys and
  // (through the indices, which have arrived by then) its points one iteration early
  auto load_keys = [&](int t0, u64& key, u64& prev) {
    const int i = t0 + tid;
    key = kInvalidKey;
    prev = kInvalidKey;
    if (i < e1) {
      key = keys[i];
      if (i > 0) prev = keys[i - 1];
    }
  };
  auto load_idx = [&](int t0) { return t0 + tid < e1 ? idx[t0 + tid] : 0u; };
  u64 key_n, prev_n;
  load_keys(e0, key_n, prev_n);
  unsigned ix_n = load_idx(e0), ix_nn = load_idx(e0 + 1024);
  float4 p_n = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e0 + tid < e1) p_n = in[ix_n];
  for (int t0 = e0; t0 < e1; t0 += 1024) {
    const int i = t0 + tid;
    const u64 key = key_n, prevk = prev_n;
    const float4 p = p_n;
    // the next tile's keys and points, the indices of the one after it
    load_keys(t0 + 1024, key_n, prev_n);
    ix_n = ix_nn;
    p_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t0 + 1024 + tid < e1) p_n = in[ix_n];
    ix_nn = load_idx(t0 + 2048);
    const bool valid = key != kInvalidKey;
    const bool start = valid && (i == 0 || (i & 1023) == 0 || prevk != key);
    if (valid) {
      s_px[tid] = p.x;
      s_py[tid] = p.y;
      s_pz[tid] = p.z;
    }
    s_stop[tid] = (start || !valid) ? 1 : 0;  // a run ends in front of the next start, of the first invalid key, or with the tile
    int total;
    const int pos = block_scan_1024(start ? 1 : 0, s_wave, &total) + s_carry;  // (its barriers publish the tile)
    if (start) {
      double sx = 0, sy = 0, sz = 0, sw = 0;
      int j = tid;
      do {
        sx += (double)s_px[j];
        sy += (double)s_py[j];
        sz += (double)s_pz[j];
        sw += 1.0;
        j++;
      } while (j < 1024 && !s_stop[j]);
      const double mx = sx / sw, my = sy / sw, mz = sz / sw;
      out[pos] = make_double4(mx, my, mz, sw / sw);
      {  // cell sort key: the cell (as every search computes it: floor(v * inv_cell)) and, below it, the slice (1 / kFine) of the cell along x
        const double ux = mx * inv_cell;
        const int cxm = fast_floor_d(ux), cym = fast_floor_d(my * inv_cell), czm = fast_floor_d(mz * inv_cell);
        const int sub = min(max((int)((ux - (double)cxm) * (double)kFine), 0), kFine - 1);
        const long long xf = (long long)cxm * kFine + sub + kCkOffX, yf = (long long)cym + kCkOffY, zf = (long long)czm + kCkOffZ;
        // (|coordinate| < 2^18 cells: k_voxel_keys dropped everything else, and a mean lies inside its points' hull)
        ck[pos] = (u64)xf | ((u64)yf << kCkSy) | ((u64)zf << kCkSz);
      }
      ci[pos] = (unsigned)pos;
    }
    __syncthreads();
    if (tid == 0) s_carry += total;
    __syncthreads();
  }
  if (tid == 0 && part + 1 == (int)gridDim.y) m_counts[c] = s_carry;
}

// ------------------------------------------------------------------------------------------------
// k_cell_build: gather the means into cell-sorted order and build the table of occupied cells
// (sorted unique cell keys + first point of each cell).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_cell_build(const double4* __restrict__ tmp_pts, const u64* __restrict__ ck0,
                                                     const u64* __restrict__ ck1, const unsigned* __restrict__ ci0,
                                                     const unsigned* __restrict__ ci1, const int* __restrict__ which,
                                                     const int* __restrict__ m_counts, int P, double4* __restrict__ pts,
                                                     u64* __restrict__ ucell, unsigned* __restrict__ ubegin,
                                                     int* __restrict__ n_ucell, int* __restrict__ bbox,
                                                     const int* __restrict__ kinfo, int only) {
  // gridDim.y workgroups a cloud: part p takes the sorted elements [a0, a1) and first counts the cells that start in front of a0
  // (bbox[6c ..] was reset by k_voxel_reduce; the parts reduce into it with atomics)
  __shared__ int s_wave[16];
  __shared__ int s_bb[6];
  const int c = blockIdx.x, tid = threadIdx.x, part = blockIdx.y, nparts = gridDim.y;
  if (only >= 0 && (c & 1) != only) return;
  if (tid < 3) s_bb[tid] = kCoordMask;
  else if (tid < 6) s_bb[tid] = 0;
  const int m = m_counts[c];
  const u64* ck = (which[c] ? ck1 : ck0) + (size_t)c * P;
  const unsigned* ci = (which[c] ? ci1 : ci0) + (size_t)c * P;
  const double4* tp = tmp_pts + (size_t)c * P;
  double4* out = pts + (size_t)c * P;
  u64* uc = ucell + (size_t)c * (P + 1);
  unsigned* ub = ubegin + (size_t)c * (P + 1);
  __syncthreads();
  const int span = (m + nparts - 1) / nparts, a0 = min(part * span, m), a1 = min(a0 + span, m);
  for (int i = a0 + tid; i < a1; i += 4 * 1024) {  // independent gathers, four in flight (a plain loop waits for every one of them)
    unsigned ix[4];
    double4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) ix[u] = ci[min(i + u * 1024, a1 - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = tp[ix[u]];
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (i + u * 1024 < a1) out[i + u * 1024] = v[u];
  }
  // every thread owns a contiguous chunk of the part's sorted keys: one block scan numbers the cell starts
  const int chunk = (a1 - a0 + 1023) / 1024;
  const int i0 = min(a0 + tid * chunk, a1), i1 = min(i0 + chunk, a1);
  // the sort compacted the keys (k_radix_sort); a CELL is the key without the kFineBits sub-cell bits of its x field
  const int* ki = kinfo + 8 * c;
  const int kbx = ki[3], kby = ki[4];
  auto cell_of = [&](u64 key, int& kx, int& ky, int& kz) {
    const int xf = (int)(key & ((1ull << kbx) - 1)) + ki[0] - kCkOffX;
    kx = (xf >> kFineBits) + kCoordOffset;  // arithmetic shift = floor
    ky = (int)((key >> kbx) & ((1ull << kby) - 1)) + ki[1] - kCkOffY + kCoordOffset;
    kz = (int)(key >> (kbx + kby)) + ki[2] - kCkOffZ + kCoordOffset;
  };
  auto new_cell = [&](int i) {
    if (i == 0) return true;
    int ax, ay, az, bx_, by_, bz_;
    cell_of(ck[i - 1], ax, ay, az);
    cell_of(ck[i], bx_, by_, bz_);
    return ax != bx_ || ay != by_ || az != bz_;
  };
  // (a chunk of at most kCbChunk keys — clouds of up to 20 480 points — is loaded once, all loads in flight, with the key in front
  // of it; the two walks below then run from registers.  Larger chunks take the loops over global memory.)
  constexpr int kCbChunk = 20;
  const bool cached = chunk <= kCbChunk;
  u64 kc[kCbChunk + 1];  // kc[j] = key i0 - 1 + j
  unsigned newbits = 0;  // bit j: key i0 + j starts a cell
  if (cached) {
#pragma unroll
    for (int j = 0; j <= kCbChunk; j++) kc[j] = ck[min(max(i0 - 1 + j, 0), max(m - 1, 0))];
  }
  int cnt = 0;
  if (cached) {
    int px, py, pz;
    cell_of(kc[0], px, py, pz);
#pragma unroll
    for (int j = 0; j < kCbChunk; j++) {
      if (i0 + j < i1) {
        int qx, qy, qz;
        cell_of(kc[j + 1], qx, qy, qz);
        const bool nw = i0 + j == 0 || px != qx || py != qy || pz != qz;
        newbits |= (nw ? 1u : 0u) << j;
        px = qx;
        py = qy;
        pz = qz;
      }
    }
    cnt = __popc(newbits);
  } else {
    for (int i = i0; i < i1; i++) cnt += new_cell(i) ? 1 : 0;
  }
  int carry = 0;  // cells that start in [0, a0)
  if (a0 > 0) {
    int before = 0;
    for (int i = tid; i < a0; i += 8 * 1024) {
      u64 k[8], kp[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int iu = min(i + u * 1024, a0 - 1);
        k[u] = ck[iu];
        kp[u] = ck[max(iu - 1, 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        int ax, ay, az, bx_, by_, bz_;
        cell_of(kp[u], ax, ay, az);
        cell_of(k[u], bx_, by_, bz_);
        before += (i + u * 1024 < a0 && (i + u * 1024 == 0 || ax != bx_ || ay != by_ || az != bz_)) ? 1 : 0;
      }
    }
    (void)block_scan_1024(before, s_wave, &carry);
    __syncthreads();
  }
  int total;
  int pos = carry + block_scan_1024(cnt, s_wave, &total);
  total += carry;
  {
    int mn[3] = {kCoordMask, kCoordMask, kCoordMask}, mx[3] = {0, 0, 0};
    auto emit = [&](int i, u64 key) {
      int kx, ky, kz;
      cell_of(key, kx, ky, kz);
      uc[pos] = pack_key(kx, ky, kz);
      ub[pos] = (unsigned)i;
      pos++;
      mn[0] = min(mn[0], kx);
      mn[1] = min(mn[1], ky);
      mn[2] = min(mn[2], kz);
      mx[0] = max(mx[0], kx);
      mx[1] = max(mx[1], ky);
      mx[2] = max(mx[2], kz);
    };
    if (cached) {
#pragma unroll
      for (int j = 0; j < kCbChunk; j++)
        if ((newbits >> j) & 1u) emit(i0 + j, kc[j + 1]);
    } else {
      for (int i = i0; i < i1; i++)
        if (new_cell(i)) emit(i, ck[i]);
    }
    for (int a = 0; a < 3; a++) {  // one LDS atomic a wave (threads without a cell hold the neutral values)
      mn[a] = gfs::wave_min_i32(mn[a]);
      mx[a] = gfs::wave_max_i32(mx[a]);
      if ((tid & 63) == 0) {
        atomicMin(&s_bb[a], mn[a]);
        atomicMax(&s_bb[3 + a], mx[a]);
      }
    }
  }
  __syncthreads();
  if (tid == 0 && part + 1 == nparts) {
    n_ucell[c] = total;
    ub[total] = (unsigned)m;
    uc[total] = kInvalidKey;
  }
  // occupied-cell bounding box (x0,y0,z0,x1,y1,z1), reduced over the parts (a part without a cell start holds the neutral values)
  if (tid < 3) atomicMin(&bbox[6 * c + tid], s_bb[tid]);
  else if (tid < 6) atomicMax(&bbox[6 * c + tid], s_bb[tid]);
}

// ------------------------------------------------------------------------------------------------
// cell-grid helpers
// ------------------------------------------------------------------------------------------------
// (out of line: the path of clouds whose bounding box exceeds the dense grid — dozens of inlined copies of this loop made the search
// kernels a third longer than what they execute)
__device__ __attribute__((noinline)) int lower_bound_u64(const u64* __restrict__ a, int n, u64 key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------------
// Dense cell grid over the occupied-cell bounding box (+1 cell margin): G[lin(x,y,z)] = index of the first point
// (in the cell-sorted array) whose cell key is >= that cell, G[ncell] = M.  Points are sorted by (z, y, x) and lin()
// is the same lexicographic order, so the points of row (y, z) with x in [xlo, xhi] are exactly
// [G[lin(xlo,y,z)], G[lin(xhi,y,z) + 1]) : a row lookup is two independent loads instead of two binary searches.
// ginfo = {gx0, gy0, gz0, nx, ny, nz, ok}.  Clouds whose box exceeds kGridCap cells keep the binary-search path.
// ------------------------------------------------------------------------------------------------
constexpr int kGridCap = 1 << 21;
constexpr int kGridFillParts = 64;  // workgroups (of 256 threads: no LDS, no barrier — small ones find a CU sooner when other lanes' kernels are in flight) per cloud in k_grid_fill

__global__ __launch_bounds__(256) void k_grid_fill(const u64* __restrict__ ucell, const unsigned* __restrict__ ubegin,
                                                    const int* __restrict__ n_ucell, const int* __restrict__ m_counts,
                                                    const int* __restrict__ bbox, int P, unsigned* __restrict__ grid,
                                                    int* __restrict__ ginfo, int* __restrict__ far2_count, int only) {
  const int c = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (only >= 0 && (c & 1) != only) return;
  const int nu = n_ucell[c], m = m_counts[c];
  const u64* uc = ucell + (size_t)c * (P + 1);
  const unsigned* ub = ubegin + (size_t)c * (P + 1);
  unsigned* G = grid + (size_t)c * (kGridCap + 1);
  const int gx0 = bbox[6 * c] - 1, gy0 = bbox[6 * c + 1] - 1, gz0 = bbox[6 * c + 2] - 1;
  const long long nx = (long long)bbox[6 * c + 3] - gx0 + 2, ny = (long long)bbox[6 * c + 4] - gy0 + 2,
                  nz = (long long)bbox[6 * c + 5] - gz0 + 2;
  const bool ok = nu > 0 && nx > 0 && ny > 0 && nz > 0 && nx * ny * nz < (long long)kGridCap;
  if (tid == 0 && part == 0) {
    int* gi = ginfo + 8 * c;
    gi[0] = gx0;
    gi[1] = gy0;
    gi[2] = gz0;
    gi[3] = (int)nx;
    gi[4] = (int)ny;
    gi[5] = (int)nz;
    gi[6] = ok ? 1 : 0;
    gi[7] = 0;  // k_knn_cov's deferred-query counter
    far2_count[c] = 0;
  }
  if (!ok) return;
  const int ncell = (int)(nx * ny * nz);
  auto lin = [&](u64 key) {
    const int x = (int)(key & kCoordMask), y = (int)((key >> kCoordBits) & kCoordMask), z = (int)(key >> (2 * kCoordBits));
    return ((z - gz0) * (int)ny + (y - gy0)) * (int)nx + (x - gx0);
  };
  // every occupied cell fills the gap back to the previous occupied cell (lanes write contiguous entries); the cells
  // are dealt to kGridFillParts x 4 waves
  const int nwaves = 4 * kGridFillParts;
  const int per_wave = (nu + nwaves - 1) / nwaves;
  const int u_begin = (part * 4 + wave) * per_wave, u_end = min(u_begin + per_wave, nu);
  // (the wave's cells are fetched sixty-four at a time, a lane each, and handed round by v_readlane: a load per cell in front of
  // its stores was a round trip to memory per cell, ~8 one after the other per wave)
  for (int base = u_begin; base < u_end; base += 64) {
    const int u = base + lane;
    int hi = -1, lo = 0;
    unsigned v = 0;
    if (u < u_end) {
      hi = lin(uc[u]);
      lo = u > 0 ? lin(uc[u - 1]) + 1 : 0;
      v = ub[u];
    }
    const int cnt = min(64, u_end - base);
    for (int j = 0; j < cnt; j++) {
      const int hj = __builtin_amdgcn_readlane(hi, j), lj = __builtin_amdgcn_readlane(lo, j);
      const unsigned vj = (unsigned)__builtin_amdgcn_readlane((int)v, j);
      for (int k = lj + lane; k <= hj; k += 64) G[k] = vj;
    }
  }
  const int last = lin(uc[nu - 1]);
  for (int k = last + 1 + part * 256 + tid; k <= ncell; k += 256 * kGridFillParts) G[k] = (unsigned)m;
}

// points of row (y, z) whose cell x lies in [xlo, xhi] -> [*j0, *j1)
__device__ __forceinline__ void row_range(const int* __restrict__ gi, const unsigned* __restrict__ G, const u64* __restrict__ uc,
                                          const unsigned* __restrict__ ub, int nu, int xlo, int xhi, int y, int z, int* j0,
                                          int* j1) {
  *j0 = 0;
  *j1 = 0;
  if (gi[6]) {
    const int yy = y - gi[1], zz = z - gi[2];
    if ((unsigned)yy >= (unsigned)gi[4] || (unsigned)zz >= (unsigned)gi[5]) return;
    const int a = max(xlo - gi[0], 0), b = min(xhi - gi[0], gi[3] - 1);
    if (a > b) return;
    const size_t base = ((size_t)zz * gi[4] + yy) * gi[3];
    *j0 = (int)G[base + a];
    *j1 = (int)G[base + b + 1];
  } else {
    if (y < 0 || z < 0 || y > kCoordMask || z > kCoordMask) return;
    const int u0 = lower_bound_u64(uc, nu, pack_key(max(xlo, 0), y, z));
    const int u1 = lower_bound_u64(uc, nu, pack_key(min(xhi, kCoordMask), y, z) + 1);
    *j0 = (int)ub[u0];
    *j1 = (int)ub[u1];
  }
}

// boundaries (point indices) of the three cells cx-1, cx, cx+1 of row (y, z): cell k is [e[k], e[k+1]); cells outside
// the grid collapse to empty ranges
__device__ __forceinline__ void row_cells3(const int* __restrict__ gi, const unsigned* __restrict__ G, const u64* __restrict__ uc,
                                           const unsigned* __restrict__ ub, int nu, int cx, int y, int z, int* e) {
  e[0] = e[1] = e[2] = e[3] = 0;
  if (gi[6]) {
    const int yy = y - gi[1], zz = z - gi[2];
    if ((unsigned)yy >= (unsigned)gi[4] || (unsigned)zz >= (unsigned)gi[5]) return;
    const size_t base = ((size_t)zz * gi[4] + yy) * gi[3];
    const int a = cx - 1 - gi[0];
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = (int)G[base + min(max(a + k, 0), gi[3])];
  } else {
    if (y < 0 || z < 0 || y > kCoordMask || z > kCoordMask) return;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int x = cx - 1 + k;
      const u64 key = x < 0 ? pack_key(0, y, z) : x > kCoordMask ? pack_key(kCoordMask, y, z) + 1 : pack_key(x, y, z);
      e[k] = (int)ub[lower_bound_u64(uc, nu, key)];
    }
  }
}

// XCD-aware block -> (pair, which, chunk) map.  MI355X dispatches workgroup b to XCD b % 8 and every XCD has a private
// 4 MiB L2: giving all chunks of one frame pair the same (b % 8) keeps that pair's ~1.5 MB of points / covariances /
// grid resident in ONE L2 instead of being spread over (and evicted from) all eight.  Affects speed only.
__device__ __forceinline__ bool xcd_pair_map(int nchunks, int npairs, int per_pair, int* pair, int* sub, int* chunk) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  const int span = per_pair * nchunks;
  *pair = (slot / span) * 8 + xcd;
  const int rem = slot % span;
  *sub = rem / nchunks;
  *chunk = rem % nchunks;
  return *pair < npairs;
}
inline int xcd_grid(int nchunks, int npairs, int per_pair) { return ((npairs + 7) / 8) * 8 * per_pair * nchunks; }

// ------------------------------------------------------------------------------------------------
// Eigen 3.4 SelfAdjointEigenSolver<Matrix3d>::computeDirect (closed form) — same formulas as the oracle.
// m: symmetric 3x3 as (xx, xy, xz, yy, yz, zz).  V column-major 3x3 (columns = eigenvectors, ascending).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double sqn3(const double* a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }

__device__ void extract_kernel(const double* s /*9 col-major*/, double* res, double* rep) {
  int i0 = 0;
  double best = fabs(s[0]);
  if (fabs(s[4]) > best) {
    best = fabs(s[4]);
    i0 = 1;
  }
  if (fabs(s[8]) > best) i0 = 2;
  const int j1 = (i0 + 1) % 3, j2 = (i0 + 2) % 3;
  rep[0] = s[3 * i0];
  rep[1] = s[3 * i0 + 1];
  rep[2] = s[3 * i0 + 2];
  double c0[3], c1[3];
  cross3(rep, s + 3 * j1, c0);
  cross3(rep, s + 3 * j2, c1);
  const double n0 = sqn3(c0), n1 = sqn3(c1);
  if (n0 > n1) {
    const double d = sqrt(n0);
    res[0] = c0[0] / d;
    res[1] = c0[1] / d;
    res[2] = c0[2] / d;
  } else {
    const double d = sqrt(n1);
    res[0] = c1[0] / d;
    res[1] = c1[1] / d;
    res[2] = c1[2] / d;
  }
}

__device__ void eig3_direct(const double* cov6, double* V) {
  const double shift = (cov6[0] + cov6[3] + cov6[5]) / 3.0;
  double s[9] = {cov6[0] - shift, cov6[1], cov6[2], cov6[1], cov6[3] - shift, cov6[4], cov6[2], cov6[4], cov6[5] - shift};
  double scale = 0;
  for (int i = 0; i < 9; i++) scale = fmax(scale, fabs(s[i]));
  if (scale > 0)
    for (int i = 0; i < 9; i++) s[i] /= scale;
  double ev[3];
  {
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
    const double c0 = s[0] * s[4] * s[8] + 2.0 * s[1] * s[2] * s[5] - s[0] * s[5] * s[5] - s[4] * s[2] * s[2] - s[8] * s[1] * s[1];
    const double c1 = s[0] * s[4] - s[1] * s[1] + s[0] * s[8] - s[2] * s[2] + s[4] * s[8] - s[5] * s[5];
    const double c2 = s[0] + s[4] + s[8];
    const double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = fmax(a_over_3, 0.0);
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = fmax(q, 0.0);
    const double rho = sqrt(a_over_3);
    const double theta = atan2(sqrt(q), half_b) * s_inv3;
    const double cos_theta = gfs_glibc::cos(theta), sin_theta = gfs_glibc::sin(theta);
    ev[0] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    ev[1] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    ev[2] = c2_over_3 + 2.0 * rho * cos_theta;
  }
  const double eps = 2.220446049250313e-16;
  if ((ev[2] - ev[0]) <= eps) {
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  double d0 = ev[2] - ev[1];
  const double d1 = ev[1] - ev[0];
  int k = 0, l = 2;
  if (d0 > d1) {  // Eigen: swap(k, l); d0 = d1
    k = 2;
    l = 0;
    d0 = d1;
  }
  double tmp[9], colk[3], coll[3];
  for (int i = 0; i < 9; i++) tmp[i] = s[i];
  tmp[0] -= ev[k];
  tmp[4] -= ev[k];
  tmp[8] -= ev[k];
  extract_kernel(tmp, colk, coll);
  if (d0 <= 2 * eps * d1) {
    const double dot = colk[0] * coll[0] + colk[1] * coll[1] + colk[2] * coll[2];
    for (int i = 0; i < 3; i++) coll[i] -= dot * coll[i];
    const double nn = sqrt(sqn3(coll));
    for (int i = 0; i < 3; i++) coll[i] /= nn;
  } else {
    for (int i = 0; i < 9; i++) tmp[i] = s[i];
    tmp[0] -= ev[l];
    tmp[4] -= ev[l];
    tmp[8] -= ev[l];
    double dummy[3];
    extract_kernel(tmp, coll, dummy);
  }
  for (int i = 0; i < 3; i++) {
    V[3 * k + i] = colk[i];
    V[3 * l + i] = coll[i];
  }
  double c1v[3];
  cross3(V + 6, V, c1v);
  const double nn = sqrt(sqn3(c1v));
  for (int i = 0; i < 3; i++) V[3 + i] = c1v[i] / nn;
}

// ------------------------------------------------------------------------------------------------
// k_knn_cov: one thread per down-sampled point: exact 10-NN in its own cloud (the query itself included,
// util/normal_estimation.hpp:69) through ring-expanding cell probes, 3x3 covariance, closed-form
// eigen-decomposition, cov := V diag(1e-3, 1, 1) V^T.  Stored as 6 doubles (xx, xy, xz, yy, yz, zz).
// ------------------------------------------------------------------------------------------------
// v_min_f64 / v_max_f64 as they are (fmin / fmax add a canonicalising v_max_f64 per operand; no NaN reaches these)
__device__ __forceinline__ double min_f64(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int K>
struct TopK {
  double d[K];
  int id[K];
  int found;
  __device__ void init() {
#pragma unroll
    for (int i = 0; i < K; i++) {
      d[i] = 1.79769313486231570e308;
      id[i] = -1;
    }
    found = 0;
  }
  // ann/knn_result.hpp:80-101 (sorted insertion, after equal distances) without a branch per slot: with
  // c[i] = (dist < d[i]) the list after the insertion is  d'[i] = c[i-1] ? d[i-1] : (c[i] ? dist : d[i])
  //                                                             = min(max(dist, d[i-1]), d[i])   (d ascending),
  // and the indices follow the same two selects.  Five instructions per slot, the same for every lane of the wave (the
  // slot-by-slot walk with an early exit costs four times that once any lane of the wave inserts, which is always).
  __device__ void push(int index, double dist) {
    if (dist >= d[K - 1]) return;
    bool ci = true;  // dist < d[K-1]
#pragma unroll
    for (int i = K - 1; i >= 1; i--) {
      const bool cp = dist < d[i - 1];
      id[i] = cp ? id[i - 1] : (ci ? index : id[i]);
      d[i] = min_f64(max_f64(dist, d[i - 1]), d[i]);
      ci = cp;
    }
    id[0] = ci ? index : id[0];
    d[0] = min_f64(dist, d[0]);
    found = min(found + 1, K);
  }
  __device__ double nth(int n) const {  // d[n] without dynamic register indexing (keeps the arrays out of scratch)
    double v = d[K - 1];
#pragma unroll
    for (int i = 0; i < K; i++)
      if (i == n) v = d[i];
    return v;
  }
};

// The main 10-NN pass keeps its candidates as KEYS: the squared distance (a non-negative double, whose bit pattern orders like the
// value) with the low 20 mantissa bits replaced by the point index (clouds hold at most 2^20 points).  A key is still a double, so
// a compare-exchange of two candidates is v_min_f64 + v_max_f64 -- two instructions that carry the index along -- instead of the
// five per slot of the sorted insertion (compare, two selects for the index, max, min): four candidates are sorted (5 exchanges)
// and merged into the ascending list of ELEVEN keys by a pruned bitonic merge (27 exchanges), 16 instructions a candidate against
// 50.  The kernel is bound by exactly these instructions (95 % VALU, profiles/r03f_pmc_sq.json).
// Exactness: a key is the distance to within 2^-32; keys are compared, so two candidates whose distances agree in all but the last
// 20 bits are ordered by index instead.  That can only matter at the boundary between the k-th and the (k+1)-th candidate -- the
// reason for the eleventh slot: if those two keys share their high bits the query is handed to the deferred pass (exact sorted
// insertion), otherwise every candidate outside the list is strictly farther than the k-th, as with exact comparisons: the SET of
// the k neighbours is exact.  Their ORDER inside the list is (distance to 2^-32, then index): two neighbours whose distances
// differ only below that come out in index order, and the covariance -- a sum over the list -- is then folded in another order
// than the exact passes (and the reference) fold it: a last-bit difference of the covariance (tests hold it to 1e-9, the pose to
// 1e-5), not a different neighbour set.  Bounds derived from a key use its upper end (low bits all ones).
struct TopKey11 {
  static constexpr double kNone = 1.79769313486231570e308;  // above every key (distances are far below 1e300)
  double k[11];
  __device__ void init() {
#pragma unroll
    for (int i = 0; i < 11; i++) k[i] = kNone;
  }
  static __device__ __forceinline__ double key(double dist, int index) {
    const u64 b = (u64)__double_as_longlong(dist);
    return __longlong_as_double((long long)((b & ~0xfffffull) | (u64)(unsigned)index));
  }
  static __device__ __forceinline__ void cx(double& a, double& b) {  // a <- min, b <- max
    const double lo = min_f64(a, b), hi = max_f64(a, b);
    a = lo;
    b = hi;
  }
  // merges four keys (kNone = no candidate) into the list
  __device__ __forceinline__ void push4(double c0, double c1, double c2, double c3) {
    cx(c0, c1);
    cx(c2, c3);
    cx(c0, c2);
    cx(c1, c3);
    cx(c1, c2);  // c0 <= c1 <= c2 <= c3
    // V[0..10] = k (ascending), V[11] = kNone, V[12..15] = c3, c2, c1, c0 (descending): bitonic; merge, keep V[0..10]
    double v11 = kNone, v12 = c3, v13 = c2, v14 = c1, v15 = c0;
    cx(k[0], k[8]);
    cx(k[1], k[9]);
    cx(k[2], k[10]);
    cx(k[4], v12);
    cx(k[5], v13);
    cx(k[6], v14);
    cx(k[7], v15);  // (k[3] against v11 = kNone: nothing moves)
    // lower half: the eight smallest, bitonic -> sorted
    cx(k[0], k[4]);
    cx(k[1], k[5]);
    cx(k[2], k[6]);
    cx(k[3], k[7]);
    cx(k[0], k[2]);
    cx(k[1], k[3]);
    cx(k[4], k[6]);
    cx(k[5], k[7]);
    cx(k[0], k[1]);
    cx(k[2], k[3]);
    cx(k[4], k[5]);
    cx(k[6], k[7]);
    // upper half: its three smallest, sorted, into k[8..10]
    cx(k[8], v12);
    cx(k[9], v13);
    cx(k[10], v14);
    v11 = min_f64(v11, v15);
    cx(k[8], k[10]);
    cx(k[9], v11);
    cx(k[8], k[9]);
    k[10] = min_f64(k[10], v11);
  }
  __device__ __forceinline__ double at(int n) const {
    // k[n] without dynamic register indexing -- and without a chain of selects, which the optimiser turns back into an indexed
    // load from a scratch copy of the array: the bits are blended with integer masks
    u64 bits = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) bits |= (u64)__double_as_longlong(k[i]) & (u64)(-(long long)(i == n));
    return __longlong_as_double((long long)bits);
  }
  // upper end of the distance a key stands for
  static __device__ __forceinline__ double upper(double key) { return __longlong_as_double(__double_as_longlong(key) | 0xfffffll); }
  static __device__ __forceinline__ int index(double key) { return (int)((unsigned)__double_as_longlong(key) & 0xfffffu); }
  static __device__ __forceinline__ bool same_high_bits(double a, double b) {
    return ((u64)__double_as_longlong(a) >> 20) == ((u64)__double_as_longlong(b) >> 20);
  }
};

// point j of a cloud through a 32-bit byte offset (clouds hold < 2^20 points of 32 bytes): the address is a scalar base + a zero-extended
// VGPR offset (global_load ... v, s[base]) instead of a sign extension and a 64-bit shift-add per load
__device__ __forceinline__ double4 ld_pt(const double4* __restrict__ base, int j) {
  return *reinterpret_cast<const double4*>(reinterpret_cast<const char*>(base) + (unsigned)((unsigned)j << 5));
}
// (no arrays of double4 below: they end up in scratch memory behind FLAT instructions)
__device__ __forceinline__ double knn_key(const double4& t, const double4& q, int j, bool on) {
  const double ddx = t.x - q.x, ddy = t.y - q.y, ddz = t.z - q.z;
  return on ? TopKey11::key(ddx * ddx + ddy * ddy + ddz * ddz, j) : TopKey11::kNone;
}
// the run [j0, j1) of candidate points into the key list, four loads in flight
__device__ __forceinline__ void knn_scan_run_keys(const double4* __restrict__ p, const double4& q, int j0, int j1, TopKey11& loc) {
  for (int j = j0; j < j1; j += 4) {
    const double4 t0 = ld_pt(p, j), t1 = ld_pt(p, min(j + 1, j1 - 1)), t2 = ld_pt(p, min(j + 2, j1 - 1)), t3 = ld_pt(p, min(j + 3, j1 - 1));
    loc.push4(knn_key(t0, q, j, true), knn_key(t1, q, j + 1, j + 1 < j1), knn_key(t2, q, j + 2, j + 2 < j1),
              knn_key(t3, q, j + 3, j + 3 < j1));
  }
}
#ifdef GFS_KNN_UTIL
// variant builds (tools/probes/knn_util_probe.py): lanes at work summed over the steps of a scan, and the steps (per wave)
__device__ __forceinline__ void knn_util_step(unsigned* util, int slot) {
  if (!util) return;
  const unsigned long long m = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
    atomicAdd(util + slot, (unsigned)__popcll(m));
    atomicAdd(util + slot + 1, 1u);
  }
}
#define GFS_KNN_UTIL_STEP(util, slot) knn_util_step(util, slot)
#else
#define GFS_KNN_UTIL_STEP(util, slot)
#endif
// The query's OWN row of cells [lo, hi) -- it holds the query itself, at index i -- walked from the query outwards, two candidates a
// side per step (round 6).  The row is ordered by slices of 1 / kFine of a cell along x (the cell sort key), so everything beyond
// the outermost point seen on a side has an x of at least that point's minus one slice: the side stops once that gap alone exceeds
// the k-th distance so far.  A depth-camera cloud puts 25 points into a cell its surface crosses, the k-th neighbour is ~4 cm away
// and the row 30 cm long: the walk takes a third of the row.  The keys are (distance | index), so the list does not depend on the
// order of the visits, and what is left out is farther than the k-th candidate at that time (as with the neighbouring rows below).
__device__ __forceinline__ void knn_walk_own_row(const double4* __restrict__ p, const double4& q, int i, int lo, int hi, double slack,
                                                 int want, TopKey11& loc, unsigned* util = nullptr) {
  int r = i, l = i - 1;
  bool ra = r < hi, la = l >= lo;
  const int kw_at = max(want - 1, 0);
  while (ra || la) {
    GFS_KNN_UTIL_STEP(util, 0);
    const int j0 = min(r, hi - 1), j1 = max(l, lo), j2 = min(r + 1, hi - 1), j3 = max(l - 1, lo);
    const bool on2 = ra && r + 1 < hi, on3 = la && l - 1 >= lo;
    const double4 t0 = ld_pt(p, j0), t1 = ld_pt(p, j1), t2 = ld_pt(p, j2), t3 = ld_pt(p, j3);
    loc.push4(knn_key(t0, q, j0, ra), knn_key(t1, q, j1, la), knn_key(t2, q, j2, on2), knn_key(t3, q, j3, on3));
    const double kw = loc.at(kw_at);
    const double B = kw < TopKey11::kNone ? TopKey11::upper(kw) : TopKey11::kNone;
    if (ra) {
      r += 2;
      const double g = t2.x - slack - q.x;  // (t2 = t0 when the row ends at r: nothing is ahead then anyway)
      ra = r < hi && !(g > 0.0 && g * g > B);
    }
    if (la) {
      l -= 2;
      const double g = q.x - t3.x - slack;
      la = l >= lo && !(g > 0.0 && g * g > B);
    }
  }
}
// the same over up to four runs (begin, length) concatenated into one lane-private sequence
__device__ __forceinline__ void knn_scan_runs4_keys(const double4* __restrict__ p, const double4& q, int b0, int l0, int b1, int l1,
                                                    int b2, int l2, int b3, int l3, TopKey11& loc, unsigned* util = nullptr) {
  const int c1 = l0, c2 = c1 + l1, c3 = c2 + l2, total = c3 + l3;
  const int o0 = b0, o1 = b1 - c1, o2 = b2 - c2, o3 = b3 - c3;
  for (int v = 0; v < total; v += 4) {
    GFS_KNN_UTIL_STEP(util, 2);
    int j[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int vv = min(v + u, total - 1);
      j[u] = vv + (vv < c1 ? o0 : vv < c2 ? o1 : vv < c3 ? o2 : o3);
    }
    const int j0 = j[0], j1 = j[1], j2 = j[2], j3 = j[3];
    const double4 t0 = ld_pt(p, j0), t1 = ld_pt(p, j1), t2 = ld_pt(p, j2), t3 = ld_pt(p, j3);
    loc.push4(knn_key(t0, q, j0, true), knn_key(t1, q, j1, v + 1 < total), knn_key(t2, q, j2, v + 2 < total),
              knn_key(t3, q, j3, v + 3 < total));
  }
}

// covariance of the k nearest neighbours, regularised: cov := V diag(1e-3, 1, 1) V^T  (util/normal_estimation.hpp:66-92)
__device__ __forceinline__ void knn_write_cov(const TopK<10>& best, int kk, const double4* __restrict__ p, double* __restrict__ out) {
  const int n = min(best.found, kk);
  if (n < 5) {  // NormalCovarianceSetter::set_invalid: cov = diag(1,1,1,0)
    out[0] = 1;
    out[1] = 0;
    out[2] = 0;
    out[3] = 1;
    out[4] = 0;
    out[5] = 1;
    return;
  }
  double sp[3] = {0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 10; k++) {
    if (k < n) {
      const double4 t = p[best.id[k]];
      sp[0] += t.x;
      sp[1] += t.y;
      sp[2] += t.z;
      sc[0] += t.x * t.x;
      sc[1] += t.x * t.y;
      sc[2] += t.x * t.z;
      sc[3] += t.y * t.y;
      sc[4] += t.y * t.z;
      sc[5] += t.z * t.z;
    }
  }
  const double dn = (double)n;
  const double mean[3] = {sp[0] / dn, sp[1] / dn, sp[2] / dn};
  // lower triangle of (sum_cross - mean * sum^T) / n  (Eigen's computeDirect reads the lower triangle)
  double cv[6];
  cv[0] = (sc[0] - mean[0] * sp[0]) / dn;
  cv[1] = (sc[1] - mean[1] * sp[0]) / dn;  // (1,0)
  cv[2] = (sc[2] - mean[2] * sp[0]) / dn;  // (2,0)
  cv[3] = (sc[3] - mean[1] * sp[1]) / dn;
  cv[4] = (sc[4] - mean[2] * sp[1]) / dn;  // (2,1)
  cv[5] = (sc[5] - mean[2] * sp[2]) / dn;
  double V[9];
  eig3_direct(cv, V);
  const double dv[3] = {1e-3, 1.0, 1.0};
  int o = 0;
  for (int r = 0; r < 3; r++)
    for (int cc = r; cc < 3; cc++) {
      double acc = 0;
      for (int k = 0; k < 3; k++) acc += (V[3 * k + r] * dv[k]) * V[3 * k + cc];
      out[o++] = acc;
    }
}

// scans the run [j0, j1) of candidate points, four loads in flight
// (bound: candidates farther than that cannot belong to the result -- k nearer ones are known to exist -- and skip the insertion)
__device__ __forceinline__ void knn_scan_run(const double4* __restrict__ p, const double4& q, int j0, int j1, TopK<10>& loc,
                                             double bound = 1.79769313486231570e308) {
  for (int j = j0; j < j1; j += 4) {  // four loads in flight
    const double4 t0 = p[j], t1 = p[min(j + 1, j1 - 1)], t2 = p[min(j + 2, j1 - 1)], t3 = p[min(j + 3, j1 - 1)];
    {
      const double ddx = t0.x - q.x, ddy = t0.y - q.y, ddz = t0.z - q.z, d = ddx * ddx + ddy * ddy + ddz * ddz;
      if (d <= bound) loc.push(j, d);
    }
    if (j + 1 < j1) {
      const double ddx = t1.x - q.x, ddy = t1.y - q.y, ddz = t1.z - q.z, d = ddx * ddx + ddy * ddy + ddz * ddz;
      if (d <= bound) loc.push(j + 1, d);
    }
    if (j + 2 < j1) {
      const double ddx = t2.x - q.x, ddy = t2.y - q.y, ddz = t2.z - q.z, d = ddx * ddx + ddy * ddy + ddz * ddz;
      if (d <= bound) loc.push(j + 2, d);
    }
    if (j + 3 < j1) {
      const double ddx = t3.x - q.x, ddy = t3.y - q.y, ddz = t3.z - q.z, d = ddx * ddx + ddy * ddy + ddz * ddz;
      if (d <= bound) loc.push(j + 3, d);
    }
  }
}

// the same over up to four runs (begin, length) concatenated into one lane-private sequence (every lane walks only ITS surviving cells, four loads in flight)
__device__ __forceinline__ void knn_scan_runs4(const double4* __restrict__ p, const double4& q, int b0, int l0, int b1, int l1, int b2,
                                               int l2, int b3, int l3, TopK<10>& loc) {
  const int c1 = l0, c2 = c1 + l1, c3 = c2 + l2, total = c3 + l3;
  const int o0 = b0, o1 = b1 - c1, o2 = b2 - c2, o3 = b3 - c3;
  for (int v = 0; v < total; v += 4) {
    int j[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int vv = min(v + u, total - 1);
      j[u] = vv + (vv < c1 ? o0 : vv < c2 ? o1 : vv < c3 ? o2 : o3);
    }
    const double4 t0 = p[j[0]], t1 = p[j[1]], t2 = p[j[2]], t3 = p[j[3]];
    {
      const double ddx = t0.x - q.x, ddy = t0.y - q.y, ddz = t0.z - q.z;
      loc.push(j[0], ddx * ddx + ddy * ddy + ddz * ddz);
    }
    if (v + 1 < total) {
      const double ddx = t1.x - q.x, ddy = t1.y - q.y, ddz = t1.z - q.z;
      loc.push(j[1], ddx * ddx + ddy * ddy + ddz * ddz);
    }
    if (v + 2 < total) {
      const double ddx = t2.x - q.x, ddy = t2.y - q.y, ddz = t2.z - q.z;
      loc.push(j[2], ddx * ddx + ddy * ddy + ddz * ddz);
    }
    if (v + 3 < total) {
      const double ddx = t3.x - q.x, ddy = t3.y - q.y, ddz = t3.z - q.z;
      loc.push(j[3], ddx * ddx + ddy * ddy + ddz * ddz);
    }
  }
}

__global__ __launch_bounds__(128) void k_knn_cov(const double4* __restrict__ pts, const u64* __restrict__ ucell,
                                                 const unsigned* __restrict__ ubegin, const int* __restrict__ n_ucell,
                                                 const int* __restrict__ m_counts, const int* __restrict__ bbox,
                                                 const unsigned* __restrict__ grid, const int* __restrict__ ginfo,
                                                 int* __restrict__ ginfo_rw, unsigned* __restrict__ hard_list,
                                                 double* __restrict__ hard_d, int nchunks, int npairs, int P, GicpParams prm,
                                                 double* __restrict__ cov6) {
  int pair, which, chunk;
  if (!xcd_pair_map(nchunks, npairs, 2, &pair, &which, &chunk)) return;
  if (prm.only >= 0 && which != prm.only) return;
  const int c = 2 * pair + which;
  const int m = m_counts[c];
  const int i = chunk * 128 + threadIdx.x;
  if (i >= m) return;
  const unsigned* G = grid + (size_t)c * (kGridCap + 1);
  const int* gi = ginfo + 8 * c;
  const double4* p = pts + (size_t)c * P;
  const u64* uc = ucell + (size_t)c * (P + 1);
  const unsigned* ub = ubegin + (size_t)c * (P + 1);
  const int nu = n_ucell[c];
  const double4 q = p[i];
  const int cx = fast_floor_d(q.x * prm.inv_cell) + kCoordOffset, cy = fast_floor_d(q.y * prm.inv_cell) + kCoordOffset,
            cz = fast_floor_d(q.z * prm.inv_cell) + kCoordOffset;
  TopKey11 best;
  const int kk = min(prm.k_neighbors, 10);
  const int want = min(kk, m);
  // The 27-cell cube, branch-and-bound per lane like the 1-NN search of k_gicp_linearize: the own row (3 cells) is scanned
  // first; once k candidates are known, a cell whose box is farther than the k-th distance so far cannot contribute and is
  // skipped (same neighbours, same order of visits, same bound handed to the deferred passes).  The four rows sharing a face
  // with the own row come next, then the four diagonal ones with the bound those left; each lane walks only ITS surviving
  // cells as one concatenated sequence, four loads in flight (a wave iterates max-over-lanes(candidates) / 4 times: ~65 instead
  // of ~105 candidates on a depth-camera cloud).
  bool certified = false;
  double kth_for_deferred = TopKey11::kNone;
  {
    best.init();
    const double cell2 = prm.cell * prm.cell;
    const double ux = q.x * prm.inv_cell - (double)(cx - kCoordOffset), uy = q.y * prm.inv_cell - (double)(cy - kCoordOffset),
                 uz = q.z * prm.inv_cell - (double)(cz - kCoordOffset);
    const double lx0 = fmax(ux - 1e-9, 0.0) * prm.cell, lx2 = fmax(1.0 - ux - 1e-9, 0.0) * prm.cell;
    const double ly[3] = {fmax(uy - 1e-9, 0.0) * prm.cell, 0.0, fmax(1.0 - uy - 1e-9, 0.0) * prm.cell};
    const double lz[3] = {fmax(uz - 1e-9, 0.0) * prm.cell, 0.0, fmax(1.0 - uz - 1e-9, 0.0) * prm.cell};
    {
      int j0, j1;
      row_range(gi, G, uc, ub, nu, cx - 1, cx + 1, cy, cz, &j0, &j1);
#ifdef GFS_KNN_OWN_ROW_SCAN
      knn_scan_run_keys(p, q, j0, j1, best);  // (rounds 3 - 5: the whole row)
#else
      if (i >= j0 && i < j1)  // (always: the query is a point of its own cell)
        knn_walk_own_row(p, q, i, j0, j1, prm.cell * (1.0 / kFine + 1e-9), want, best, prm.tile_stats);
      else
        knn_scan_run_keys(p, q, j0, j1, best);
#endif
    }
    // k-th distance so far (its upper end), or "none yet"
    auto kth_bound = [&]() {
      const double kw = best.at(max(want - 1, 0));
      return kw < TopKey11::kNone ? TopKey11::upper(kw) : TopKey11::kNone;
    };
    constexpr int rows[2][4] = {{1, 3, 5, 7}, {0, 2, 6, 8}};  // (dy + 1) + 3 (dz + 1): faces, then diagonals
#pragma unroll
    for (int round = 0; round < 2; round++) {
      const double B = kth_bound();
      int rb[4], rl[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t9 = rows[round][u], dy = t9 % 3, dz = t9 / 3;
        const double row2 = ly[dy] * ly[dy] + lz[dz] * lz[dz];
        rb[u] = 0;
        rl[u] = 0;
        if (row2 <= B) {
          int e[4];
          row_cells3(gi, G, uc, ub, nu, cx, cy + dy - 1, cz + dz - 1, e);
          const int b0 = row2 + lx0 * lx0 <= B ? e[0] : e[1], e0 = row2 + lx2 * lx2 <= B ? e[3] : e[2];
          rb[u] = b0;
          rl[u] = e0 - b0;
        }
      }
      knn_scan_runs4_keys(p, q, rb[0], rl[0], rb[1], rl[1], rb[2], rl[2], rb[3], rl[3], best, prm.tile_stats);
    }
#ifdef GFS_KNN_UTIL
    if (prm.tile_stats) {  // queries (lanes) and waves; the candidates of the own row as a whole
      GFS_KNN_UTIL_STEP(prm.tile_stats, 4);
      int j0, j1;
      row_range(gi, G, uc, ub, nu, cx - 1, cx + 1, cy, cz, &j0, &j1);
      atomicAdd(prm.tile_stats + 6, (unsigned)(j1 - j0));
    }
#endif
    // certified: nothing outside the 27-cell cube can be nearer than the k-th candidate.  The cube's faces are 1 + u and 2 - u
    // cells away from the query along each axis (u = its position inside its own cell): at least one cell, up to 1.5
    const double fx = fmin(ux + 1.0, 2.0 - ux), fy = fmin(uy + 1.0, 2.0 - uy), fz = fmin(uz + 1.0, 2.0 - uz);
    const double reach = fmax(fmin(fx, fmin(fy, fz)) - 1e-9, 1.0) * prm.cell;
    // ... and the key list must name the k nearest beyond doubt: the (k+1)-th key differs from the k-th above the index bits
    const double kw = best.at(max(want - 1, 0)), kn = best.at(want);
    certified = want > 0 && kw < TopKey11::kNone && TopKey11::upper(kw) <= fmax(reach * reach, cell2) &&
                !(kn < TopKey11::kNone && TopKey11::same_high_bits(kw, kn));
    certified = (certified || want == 0) && !prm.knn_exact;
    kth_for_deferred = kw < TopKey11::kNone ? TopKey11::upper(kw) : TopKey11::kNone;
  }
  // Isolated point (k-th neighbour beyond one cell, ~2 % of a depth-camera cloud): it needs a (much) bigger probe.  Done
  // here it would stall the other 63 lanes of its wave (and ~70 % of the waves hold such a lane), so it is deferred.
  if (!certified) {
    const int slot = atomicAdd(&ginfo_rw[8 * c + 7], 1);
    hard_list[(size_t)c * P + slot] = (unsigned)i;
    // k candidates already known: the true k nearest lie within this distance (bounds the follow-up probe)
    hard_d[(size_t)c * P + slot] = kth_for_deferred;
    return;
  }
  TopK<10> res;  // the neighbours in key order (distance, then index)
  res.found = 0;
#pragma unroll
  for (int k = 0; k < 10; k++) {
    res.id[k] = best.k[k] < TopKey11::kNone ? TopKey11::index(best.k[k]) : -1;
    res.found += best.k[k] < TopKey11::kNone ? 1 : 0;
  }
  knn_write_cov(res, kk, p, cov6 + ((size_t)c * P + i) * 6);
}

// k_knn_cov_far<LANES, R0, DEFER>: the queries whose k-th neighbour is farther than one cell (sparse regions, a few %
// of the points).  A group of LANES lanes per deferred query, 256 / LANES queries per workgroup and round.  Search,
// group-wide: the cells (small probes) or rows (big probes) of the ring-r cube are spread over the group's lanes (a
// lone thread would walk them as one long dependent chain; a whole wave per query leaves the chip latency-bound on the
// per-query fixed costs), every lane keeps the top-k of its share, and the k global winners are extracted by k
// group-wide arg-min rounds over the lanes' list heads (ties: lower point index).  The probe grows until the k-th
// distance is certified (<= r cells) or it covers the whole cloud.  The covariance / eigen step (scalar per query)
// then runs for the workgroup's queries side by side.
// Two passes: <16, 2, true> takes k_knn_cov's list (front of hard_list, counter ginfo[7]) with one r = 2 probe and
// defers the really isolated points (~1 % of the list, but thousands of candidates each) to <64, 4, false>
// (back of hard_list, counter far2_count) so that they do not hold up the other queries of their workgroup.
// (three waves a SIMD: left alone the kernel takes 205 VGPRs = two waves, and it waits on dependent cell lookups — 1.24 -> 0.99 ms
// per 1 024 clouds with the cap, a 124-byte spill included; four waves: no further gain)
template <int LANES, int R0, bool DEFER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_knn_cov_far(const double4* __restrict__ pts, const u64* __restrict__ ucell,
                                                     const unsigned* __restrict__ ubegin, const int* __restrict__ n_ucell,
                                                     const int* __restrict__ m_counts, const int* __restrict__ bbox,
                                                     const unsigned* __restrict__ grid, const int* __restrict__ ginfo,
                                                     unsigned* __restrict__ hard_list, double* __restrict__ hard_d,
                                                     int* __restrict__ far2_count, int nchunks, int npairs, int P, GicpParams prm,
                                                     double* __restrict__ cov6) {
  constexpr int kQueries = 256 / LANES;
  __shared__ int s_ids[kQueries][10];
  __shared__ int s_found[kQueries];
  __shared__ int s_query[kQueries];
  int pair, which, chunk;
  if (!xcd_pair_map(nchunks, npairs, 2, &pair, &which, &chunk)) return;
  if (prm.only >= 0 && which != prm.only) return;
  const int c = 2 * pair + which;
  const int m = m_counts[c];
  const unsigned* G = grid + (size_t)c * (kGridCap + 1);
  const int* gi = ginfo + 8 * c;
  const double4* p = pts + (size_t)c * P;
  const u64* uc = ucell + (size_t)c * (P + 1);
  const unsigned* ub = ubegin + (size_t)c * (P + 1);
  unsigned* list = hard_list + (size_t)c * P;
  double* list_d = hard_d + (size_t)c * P;
  const int nu = n_ucell[c];
  const int nhard = DEFER ? gi[7] : far2_count[c];
  const int kk = min(prm.k_neighbors, 10);
  const int want = min(kk, m);
  const int bx0 = bbox[6 * c], by0 = bbox[6 * c + 1], bz0 = bbox[6 * c + 2], bx1 = bbox[6 * c + 3], by1 = bbox[6 * c + 4],
            bz1 = bbox[6 * c + 5];
  const int gl = threadIdx.x % LANES, grp = threadIdx.x / LANES;
  for (int base = chunk * kQueries; base < nhard; base += nchunks * kQueries) {  // uniform per workgroup
    const int h = base + grp;
    if (gl == 0) s_query[grp] = -1;
    if (h < nhard) {
      const int i = (int)(DEFER ? list[h] : list[P - 1 - h]);
      // squared distance within which the k nearest are known to lie (k_knn_cov found k candidates), or "infinite"
      const double Dk = DEFER ? list_d[h] : list_d[P - 1 - h];
      const bool bounded = Dk < 1.0e300;
      const double4 q = p[i];
      const int cx = fast_floor_d(q.x * prm.inv_cell) + kCoordOffset, cy = fast_floor_d(q.y * prm.inv_cell) + kCoordOffset,
                cz = fast_floor_d(q.z * prm.inv_cell) + kCoordOffset;
      // lower bound of the distance from q to the cells d steps away along one axis (a hair conservative)
      const double ux = q.x * prm.inv_cell - (double)(cx - kCoordOffset), uy = q.y * prm.inv_cell - (double)(cy - kCoordOffset),
                   uz = q.z * prm.inv_cell - (double)(cz - kCoordOffset);
      auto axis_lb = [&](int d, double u) {
        const double v = d == 0 ? 0.0 : d > 0 ? (double)d - u : u - (double)(d + 1);
        return fmax(v - 1e-9, 0.0) * prm.cell;
      };
      TopK<10> best;  // merged result, identical in all lanes of the group
      bool done = false;
      double Dnext = Dk;  // bound handed to the next pass
      // bounded: one probe of ceil(sqrt(Dk) / cell) rings is final, and cells (rows) farther than sqrt(Dk) are skipped
      int r = bounded ? max((int)ceil(sqrt(Dk) * prm.inv_cell), 2) : R0;
      if (DEFER && r > 2) r = -1;  // beyond this pass: hand over at once
      for (; r > 0;) {
        TopK<10> loc;
        loc.init();
        // cells outside the occupied-cell bounding box are empty: clamp the probe to it
        const int x0 = max(cx - r, bx0), x1 = min(cx + r, bx1), y0 = max(cy - r, by0), y1 = min(cy + r, by1),
                  z0 = max(cz - r, bz0), z1 = min(cz + r, bz1);
        const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
        const int upr = nx * nrows <= 128 ? nx : 1;  // work units per row: single cells for the r = 2 probe, whole rows otherwise
        const int nunits = nrows * upr;
        for (int t = gl; t < nunits; t += 2 * LANES) {  // two units per trip: both lookups are in flight together
          const int ta = t, tb = min(t + LANES, nunits - 1);
          const int rowa = ta / upr, xa = ta - rowa * upr, rowb = tb / upr, xb = tb - rowb * upr;
          const int ya = y0 + rowa % ny, za = z0 + rowa / ny, yb = y0 + rowb % ny, zb = z0 + rowb / ny;
          bool usea = true, useb = t + LANES < nunits;
          if (bounded) {
            const double la = axis_lb(ya - cy, uy), lza = axis_lb(za - cz, uz), lxa = upr == 1 ? 0.0 : axis_lb(x0 + xa - cx, ux);
            const double lb = axis_lb(yb - cy, uy), lzb = axis_lb(zb - cz, uz), lxb = upr == 1 ? 0.0 : axis_lb(x0 + xb - cx, ux);
            usea = la * la + lza * lza + lxa * lxa <= Dk;
            useb = useb && lb * lb + lzb * lzb + lxb * lxb <= Dk;
          }
          int ja0 = 0, ja1 = 0, jb0 = 0, jb1 = 0;
          if (usea) row_range(gi, G, uc, ub, nu, upr == 1 ? x0 : x0 + xa, upr == 1 ? x1 : x0 + xa, ya, za, &ja0, &ja1);
          if (useb) row_range(gi, G, uc, ub, nu, upr == 1 ? x0 : x0 + xb, upr == 1 ? x1 : x0 + xb, yb, zb, &jb0, &jb1);
          knn_scan_run(p, q, ja0, ja1, loc, Dk);
          knn_scan_run(p, q, jb0, jb1, loc, Dk);
        }
        best.init();
#pragma unroll
        for (int k = 0; k < 10; k++) {
          const double hd = loc.d[0];
          const int hj = loc.id[0];
          double bd = hd;
          int bj = hj;
#pragma unroll
          for (int ofs = LANES / 2; ofs > 0; ofs >>= 1) {
            const double od = __shfl_xor(bd, ofs, LANES);
            const int oj = __shfl_xor(bj, ofs, LANES);
            if (od < bd || (od == bd && (unsigned)oj < (unsigned)bj)) {
              bd = od;
              bj = oj;
            }
          }
          if (bj >= 0) {
            best.d[k] = bd;
            best.id[k] = bj;
            best.found = k + 1;
          }
          if (hj == bj && bj >= 0) {  // the owning lane pops its head
#pragma unroll
            for (int e = 0; e < 9; e++) {
              loc.d[e] = loc.d[e + 1];
              loc.id[e] = loc.id[e + 1];
            }
            loc.d[9] = 1.79769313486231570e308;
            loc.id[9] = -1;
          }
        }
        const double reach = (double)r * prm.cell;
        const bool covers_all = cx - r <= bx0 && cx + r >= bx1 && cy - r <= by0 && cy + r >= by1 && cz - r <= bz0 && cz + r >= bz1;
        const double kth = best.nth(max(want - 1, 0));
        if ((best.found >= want && kth <= reach * reach) || covers_all || r > kCoordMask) {
          done = true;
          break;
        }
        if (best.found >= want) Dnext = fmin(Dnext, kth);
        if (DEFER) break;
        // k candidates are known: the true k nearest lie within sqrt(kth), so the next probe is the last one
        r = best.found >= want ? min(2 * r, (int)ceil(sqrt(kth) * prm.inv_cell)) : 2 * r;
      }
      if (gl == 0) {
        if (done) {
#pragma unroll
          for (int k = 0; k < 10; k++) s_ids[grp][k] = best.id[k];
          s_found[grp] = best.found;
          s_query[grp] = i;
        } else {
          const int slot = atomicAdd(&far2_count[c], 1);
          list[P - 1 - slot] = (unsigned)i;
          list_d[P - 1 - slot] = Dnext;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < kQueries && s_query[threadIdx.x] >= 0) {
      TopK<10> res;
      res.found = s_found[threadIdx.x];
#pragma unroll
      for (int k = 0; k < 10; k++) res.id[k] = s_ids[threadIdx.x][k];
      knn_write_cov(res, kk, p, cov6 + ((size_t)c * P + s_query[threadIdx.x]) * 6);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// k_knn_cov_far_wg: the really isolated queries (back of hard_list, counter far2_count: a handful per depth-camera cloud, each
// with a probe of hundreds of rows), one 256-thread WORKGROUP per query.  The probe's rows are looked up once, all lanes at
// once (one round trip to memory instead of a chain of them per lane), their candidate runs are numbered through and the threads
// stride over that one sequence -- balanced whatever the rows hold, coalesced inside a row.  A query that knows k candidates
// already (Dk: k_knn_cov / the r = 2 pass found k points within sqrt(Dk)) needs ONE probe of ceil(sqrt(Dk) / cell) rings and
// only candidates with d <= Dk can belong to the result, so the sorted insertion runs for those alone.  Per thread a top-k in
// visiting order = ascending point index (rows are walked in (z, y) order, points are sorted by (z, y, x)); the k winners come
// out of wave arg-min rounds and a 4-list merge, ties to the lower point index: the same set the lane-group version
// (k_knn_cov_far<64, 4, false>, the previous form of this pass) selects.
// ------------------------------------------------------------------------------------------------
constexpr int kFarWgRows = 1024;  // rows of one chunk of the probe (begin, prefix) in LDS

__global__ __launch_bounds__(256) void k_knn_cov_far_wg(const double4* __restrict__ pts, const u64* __restrict__ ucell,
                                                        const unsigned* __restrict__ ubegin, const int* __restrict__ n_ucell,
                                                        const int* __restrict__ m_counts, const int* __restrict__ bbox,
                                                        const unsigned* __restrict__ grid, const int* __restrict__ ginfo,
                                                        const unsigned* __restrict__ hard_list, const double* __restrict__ hard_d,
                                                        const int* __restrict__ far2_count, int nchunks, int npairs, int P,
                                                        GicpParams prm, double* __restrict__ cov6) {
  __shared__ int s_beg[kFarWgRows], s_pre[kFarWgRows + 1];
  __shared__ int s_wave[4];
  __shared__ double s_md[4][10];
  __shared__ int s_mi[4][10];
  int pair, which, chunk;
  if (!xcd_pair_map(nchunks, npairs, 2, &pair, &which, &chunk)) return;
  if (prm.only >= 0 && which != prm.only) return;
  const int c = 2 * pair + which;
  const int m = m_counts[c];
  const unsigned* G = grid + (size_t)c * (kGridCap + 1);
  const int* gi = ginfo + 8 * c;
  const double4* p = pts + (size_t)c * P;
  const u64* uc = ucell + (size_t)c * (P + 1);
  const unsigned* ub = ubegin + (size_t)c * (P + 1);
  const unsigned* list = hard_list + (size_t)c * P;
  const double* list_d = hard_d + (size_t)c * P;
  const int nu = n_ucell[c];
  const int nhard = far2_count[c];
  const int kk = min(prm.k_neighbors, 10);
  const int want = min(kk, m);
  const int bx0 = bbox[6 * c], by0 = bbox[6 * c + 1], bz0 = bbox[6 * c + 2], bx1 = bbox[6 * c + 3], by1 = bbox[6 * c + 4],
            bz1 = bbox[6 * c + 5];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int h = chunk; h < nhard; h += nchunks) {  // uniform per workgroup
    const int i = (int)list[P - 1 - h];
    double Dk = list_d[P - 1 - h];
    const double4 q = p[i];
    const int cx = fast_floor_d(q.x * prm.inv_cell) + kCoordOffset, cy = fast_floor_d(q.y * prm.inv_cell) + kCoordOffset,
              cz = fast_floor_d(q.z * prm.inv_cell) + kCoordOffset;
    const double uy = q.y * prm.inv_cell - (double)(cy - kCoordOffset), uz = q.z * prm.inv_cell - (double)(cz - kCoordOffset);
    auto axis_lb = [&](int d, double u) {  // lower bound of the distance to the cells d steps away along one axis (a hair conservative)
      const double v = d == 0 ? 0.0 : d > 0 ? (double)d - u : u - (double)(d + 1);
      return fmax(v - 1e-9, 0.0) * prm.cell;
    };
    TopK<10> best;  // the merged result (thread 0)
    int r = Dk < 1.0e300 ? max((int)ceil(sqrt(Dk) * prm.inv_cell), 2) : 4;  // (unbounded: r is not used, the scan takes everything)
    while (true) {
      const bool bounded = Dk < 1.0e300;
      TopK<10> loc;
      loc.init();
      const int x0 = max(cx - r, bx0), x1 = min(cx + r, bx1), y0 = max(cy - r, by0), y1 = min(cy + r, by1), z0 = max(cz - r, bz0),
                z1 = min(cz + r, bz1);
      const int ny = y1 - y0 + 1, nrows = !bounded ? 1 : x0 <= x1 && y0 <= y1 && z0 <= z1 ? ny * (z1 - z0 + 1) : 0;
      for (int row0 = 0; row0 < nrows; row0 += kFarWgRows) {
        const int nr = min(kFarWgRows, nrows - row0);
        // (a) the rows of the chunk: four a thread, their look-ups in flight together; rows (and x cells) beyond sqrt(Dk) are cut off
        int len[4], beg[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int t = tid * 4 + u;  // a thread's rows are consecutive: one exclusive scan numbers the candidates
          len[u] = 0;
          beg[u] = 0;
          if (t < nr && !bounded) {  // no k candidates known anywhere near: the whole cloud as one run (68 candidates a thread at VGA density)
            len[u] = m;
          } else if (t < nr) {
            const int row = row0 + t, y = y0 + row % ny, z = z0 + row / ny;
            int xa = x0, xb = x1;
            bool use = true;
            if (bounded) {
              const double ly = axis_lb(y - cy, uy), lz = axis_lb(z - cz, uz), rest = Dk - ly * ly - lz * lz;
              use = rest >= 0.0;
              if (use) {  // x cells farther than sqrt(rest): an integer bound one cell on the safe side
                const int rx = (int)ceil(sqrt(rest) * prm.inv_cell) + 1;
                xa = max(xa, cx - rx);
                xb = min(xb, cx + rx);
              }
            }
            if (use && xa <= xb) {
              int j0, j1;
              row_range(gi, G, uc, ub, nu, xa, xb, y, z, &j0, &j1);
              beg[u] = j0;
              len[u] = j1 - j0;
            }
          }
        }
        const int mine = len[0] + len[1] + len[2] + len[3];
        int incl = mine;
#pragma unroll
        for (int ofs = 1; ofs < 64; ofs <<= 1) {
          const int v = __shfl_up(incl, ofs, 64);
          if (lane >= ofs) incl += v;
        }
        __syncthreads();  // (the previous chunk's tables are no longer read)
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
          const int t = s_wave[w];
          if (w < wave) base += t;
          total += t;
        }
        int run = base + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int t = tid * 4 + u;
          if (t < nr) {
            s_beg[t] = beg[u];
            s_pre[t] = run;
          }
          run += len[u];
        }
        if (tid == 0) s_pre[nr] = total;
        __syncthreads();
        // (b) the candidates of the chunk, numbered through: thread t takes v = t, t + 256, ...
        int rr = 0, vend = s_pre[1];
        constexpr int kFly = 8;  // loads in flight a thread (one workgroup a query: nothing else hides their latency)
        for (int v = tid; v < total; v += kFly * 256) {
          int j[kFly];
          bool on[kFly];
#pragma unroll
          for (int u = 0; u < kFly; u++) {
            const int vv = v + u * 256;
            on[u] = vv < total;
            j[u] = 0;
            if (on[u]) {
              while (vv >= vend) {
                rr++;
                vend = s_pre[rr + 1];
              }
              j[u] = s_beg[rr] + (vv - s_pre[rr]);
            }
          }
          double4 t4[kFly];
#pragma unroll
          for (int u = 0; u < kFly; u++) t4[u] = p[j[u]];
#pragma unroll
          for (int u = 0; u < kFly; u++) {
            const double ddx = t4[u].x - q.x, ddy = t4[u].y - q.y, ddz = t4[u].z - q.z;
            const double d = ddx * ddx + ddy * ddy + ddz * ddz;
            if (on[u] && d <= Dk) loc.push(j[u], d);
          }
        }
      }
      // (c) k winners of the workgroup: per wave by arg-min rounds over the lanes' list heads, then thread 0 merges the four lists
#pragma unroll
      for (int k = 0; k < 10; k++) {
        const double hd = loc.d[0];
        const int hj = loc.id[0];
        double bd = hd;
        int bj = hj;
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
          const double od = __shfl_xor(bd, ofs, 64);
          const int oj = __shfl_xor(bj, ofs, 64);
          if (od < bd || (od == bd && (unsigned)oj < (unsigned)bj)) {
            bd = od;
            bj = oj;
          }
        }
        if (lane == 0) {
          s_md[wave][k] = bd;
          s_mi[wave][k] = bj;
        }
        if (hj == bj && bj >= 0) {  // the owning lane pops its head
#pragma unroll
          for (int e = 0; e < 9; e++) {
            loc.d[e] = loc.d[e + 1];
            loc.id[e] = loc.id[e + 1];
          }
          loc.d[9] = 1.79769313486231570e308;
          loc.id[9] = -1;
        }
      }
      __syncthreads();
      bool done = true;  // uniform: every thread evaluates the same merged list
      {
        best.init();
        int hp[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 10; k++) {
          double bd = 1.79769313486231570e308;
          int bj = -1, bw = -1;
#pragma unroll
          for (int w = 0; w < 4; w++) {
            // (hp[w] is a run-time index into LDS, not into registers)
            const double od = hp[w] < 10 ? s_md[w][hp[w]] : 1.79769313486231570e308;
            const int oj = hp[w] < 10 ? s_mi[w][hp[w]] : -1;
            if (oj >= 0 && (bj < 0 || od < bd || (od == bd && (unsigned)oj < (unsigned)bj))) {
              bd = od;
              bj = oj;
              bw = w;
            }
          }
          if (bj >= 0) {
            best.d[k] = bd;
            best.id[k] = bj;
            best.found = k + 1;
#pragma unroll
            for (int w = 0; w < 4; w++) hp[w] += w == bw ? 1 : 0;
          }
        }
        const double reach = (double)r * prm.cell;
        const bool covers_all = !bounded || (cx - r <= bx0 && cx + r >= bx1 && cy - r <= by0 && cy + r >= by1 && cz - r <= bz0 && cz + r >= bz1);
        const double kth = best.nth(max(want - 1, 0));
        done = (best.found >= want && kth <= reach * reach) || covers_all || r > kCoordMask;
        if (!done) {  // k candidates are known now: the next probe is the last one
          if (best.found >= want) Dk = fmin(Dk, kth);
          r = best.found >= want ? min(2 * r, (int)ceil(sqrt(kth) * prm.inv_cell)) : 2 * r;
        }
      }
      __syncthreads();  // (s_md / s_mi are rewritten by the next probe or query)
      if (done) break;
    }
    if (tid == 0) knn_write_cov(best, kk, p, cov6 + ((size_t)c * P + i) * 6);
  }
}

// ------------------------------------------------------------------------------------------------
// LM state machine
// ------------------------------------------------------------------------------------------------
// Eigen::Matrix3d::inverse() (Inverse.h, size-3 cofactor form); A, M column-major.
__device__ __forceinline__ void inv3(const double* A, double* M) {
#define a_(i, j) A[(i) + 3 * (j)]
#define cof_(i, j) (a_(((i) + 1) % 3, ((j) + 1) % 3) * a_(((i) + 2) % 3, ((j) + 2) % 3) - a_(((i) + 1) % 3, ((j) + 2) % 3) * a_(((i) + 2) % 3, ((j) + 1) % 3))
  const double c00 = cof_(0, 0), c10 = cof_(1, 0), c20 = cof_(2, 0);
  const double det = c00 * a_(0, 0) + c10 * a_(1, 0) + c20 * a_(2, 0);
  const double id = 1.0 / det;
  M[0] = c00 * id;
  M[3] = c10 * id;
  M[6] = c20 * id;
  M[1] = cof_(0, 1) * id;
  M[4] = cof_(1, 1) * id;
  M[7] = cof_(2, 1) * id;
  M[2] = cof_(0, 2) * id;
  M[5] = cof_(1, 2) * id;
  M[8] = cof_(2, 2) * id;
#undef a_
#undef cof_
}

// per-wave shuffle reduction, then the 4 waves of the block are folded in fixed order: deterministic sums
// per-WAVE partial sums (wave totals by recursive halving, wave_reduce.hpp), stored side by side per block: slot `wave` of the
// block's kLinWaves slots.  The four waves of a block are added up, in the fixed order 0 + w0 + w1 + w2 + w3, by the kernel that
// folds the partials (fold_round): the same numbers as a block-wide sum in LDS, but a wave that has finished its points
// leaves instead of waiting at two barriers for the slowest of its block (clock64 per section, -DGFS_LIN_TIMING in round 4: a
// wave of k_gicp_linearize lived 55 - 74 k cycles, 7 - 23 k of them in the reduction, nearly all of that waiting).
constexpr int kLinWaves = kLinBlock / 64;
// Words that cross WORKGROUPS inside one launch (k_gicp_lm_coop): 8-byte relaxed agent-scope accesses = `global_load / global_store
// ... sc1` on gfx950 -- write-through stores and L1-bypassing loads, so that no release / acquire fence (an L2 write-back, an L1
// invalidate) is needed around them; the publishing wave drains its stores (s_waitcnt vmcnt(0)) before the arrival is counted.
template <class T>
__device__ __forceinline__ T ld_ag(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_ag(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int N, bool kAgent = false>
__device__ __forceinline__ void wave_reduce_store(double (&vals)[N], double* __restrict__ dst_block, int wave) {
  const double tot = gfs_red::wave_sum_many<N>(vals);
  const int lane = threadIdx.x & 63;
  if (lane < N) {
    if constexpr (kAgent) st_ag(dst_block + wave * N + lane, tot);
    else dst_block[wave * N + lane] = tot;
  }
}
// ... a batch of N <= 16 values whose places in the block's kRed-vector are given by `map`
template <int N, bool kAgent = false>
__device__ __forceinline__ void wave_reduce_store_map(double (&vals)[N], const int (&map)[N], double* __restrict__ dst_block, int wave) {
  const double tot = gfs_red::wave_sum_many<N>(vals);
  const int lane = threadIdx.x & 63;
  if (lane < N) {
    int at = 0;
#pragma unroll
    for (int k = 0; k < N; k++) at = lane == k ? map[k] : at;
    if constexpr (kAgent) st_ag(dst_block + wave * kRed + at, tot);
    else dst_block[wave * kRed + at] = tot;
  }
}
// A 256-point chunk of a cloud is one workgroup of four waves, or -- the search kernels that need no LDS -- four workgroups of one
// wave (a CU takes a new workgroup only when all its waves fit: one-wave workgroups fill the slots that finished waves leave).
// `chunk` comes in as the launch block's index within the pair; returns the lane's point and its wave's slot of the chunk.
__device__ __forceinline__ int lin_point_of(int& chunk, int* wave_slot) {
  const int per = kLinBlock / (int)blockDim.x, part = chunk % per;
  chunk /= per;
  *wave_slot = part * ((int)blockDim.x >> 6) + ((int)threadIdx.x >> 6);
  return chunk * kLinBlock + part * (int)blockDim.x + (int)threadIdx.x;
}

// ---- the target cloud as the 1-NN search sees it: cell boundaries of a row, candidate points.  Two sources with the same
//      interface: the cloud in HBM (dense cell grid G), and a workgroup's TILE of it staged in LDS (LinTile below).
struct NnGlobal {
  const double4* tp;
  const int* gi;
  const unsigned* G;
  const u64* uc;
  const unsigned* ub;
  int nu;
#ifdef GFS_LIN_UTIL
  unsigned* util;
#endif
  // boundaries (global point indices) of cells cx-1, cx, cx+1 of row (y, z); returns the offset that turns a global point index
  // into this source's index (0 here)
  __device__ __forceinline__ int cells3(int cx, int y, int z, int* e) const {
    row_cells3(gi, G, uc, ub, nu, cx, y, z, e);
    return 0;
  }
  __device__ __forceinline__ void load(int idx, double& x, double& y, double& z) const {
    const double4 q = ld_pt(tp, idx);
    x = q.x;
    y = q.y;
    z = q.z;
  }
};

// A workgroup's tile of the target cloud.  The 256 source points of a workgroup are consecutive in cell order, so under the
// current pose they land in a compact set of target cells; the rows of cells (y, z) their 27-cell neighbourhoods touch, each cut
// to the x range those neighbourhoods need, are copied once with coalesced loads -- coordinates as three arrays of doubles, and
// the rows' cell boundaries out of the dense grid -- and every lane then walks ITS cells out of LDS instead of gathering 32-byte
// points through the vector L1 (~2 500 L1 accesses per wave before).  Workgroups whose tile does not fit (a block that straddles
// distant surfaces) use the cloud in HBM; both sources give the same correspondences.
#ifndef GFS_TILE_CAP
#define GFS_TILE_CAP 1664
#endif
#ifndef GFS_TILE_CELLS
#define GFS_TILE_CELLS 1280
#endif
constexpr int kTileCap = GFS_TILE_CAP;      // staged points
constexpr int kTileRows = 192;              // rows of cells in the box of a workgroup
constexpr int kTileCells = GFS_TILE_CELLS;  // staged cell boundaries
constexpr int kTileNz = 64;                 // non-empty rows of a tile
struct LinTile {
  double x[kTileCap], y[kTileCap], z[kTileCap];
  unsigned cells[kTileCells];
  int row_a[kTileRows], row_b[kTileRows];  // x range (min, max) of the lanes centred on the row
  int row_lo[kTileRows];                   // x of the row's first staged cell boundary (-> index into cells)
  int row_delta[kTileRows];                // LDS index = global point index + delta
  int row_coff[kTileRows];                 // offset of the row's boundaries in cells[], -1: nothing staged (empty row)
  // the non-empty rows in order (for the flat copy): end of their points / boundaries in the tile, and where they come from
  int nz_pend[kTileNz], nz_cend[kTileNz], nz_delta[kTileNz], nz_cbase[kTileNz], nz_base[kTileNz];
  int box[4];  // cymin, cymax, czmin, czmax
  unsigned long long wave_tot[4];
};
struct NnTile {
  const LinTile* t;
  int y0, z0, ny;  // the box's first row (cymin - 1, czmin - 1) and its row count along y
#ifdef GFS_LIN_UTIL
  unsigned* util;
#endif
  __device__ __forceinline__ int cells3(int cx, int y, int z, int* e) const {
    const int r = (y - y0) + ny * (z - z0);
    const int coff = t->row_coff[r];
    e[0] = e[1] = e[2] = e[3] = 0;
    if (coff < 0) return 0;
    const int k0 = coff + (cx - 1 - t->row_lo[r]);
#pragma unroll
    for (int k = 0; k < 4; k++) e[k] = (int)t->cells[k0 + k];
    return t->row_delta[r];
  }
  __device__ __forceinline__ void load(int idx, double& x, double& y, double& z) const {
    x = t->x[idx];
    y = t->y[idx];
    z = t->z[idx];
  }
};

// The candidates of one row of cells (y, z): the strip [lo, hi) of points (global indices; index in the source = global + d) is
// ordered by slices of 1 / kFine of a cell along x (the cell sort key, k_voxel_reduce).  Walk outwards from `start` in both directions, two
// candidates a side per step; a side stops when everything still ahead of it is farther than the bound: a point ahead has an x of
// at least (x of the outermost point seen on that side) - slack, slack = one such slice (+ rounding).  row2 = squared
// lower bound of the distance in y and z.  Any walk order gives the exact nearest neighbour; the order only breaks exact ties.
template <class Src>
__device__ __forceinline__ void nn_sweep(const Src& src, double tx, double ty, double tz, int lo, int hi, int d, int start, double row2,
                                         double slack, double max_dist_sq, double& B, double& best, int& bj) {
  int r = start, l = start - 1;
  bool ra = r < hi, la = l >= lo;
  while (ra || la) {
#ifdef GFS_LIN_UTIL
    if (src.util) {  // lanes at work in this step of the walk / steps of the wave (variant builds only)
      const unsigned long long m = __ballot(1);
      if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
        atomicAdd(src.util + 4, (unsigned)__popcll(m));
        atomicAdd(src.util + 5, 1u);
      }
    }
#endif
    // clamped into the strip: a clamped index is just another (or the same) candidate of the strip
    const int j[4] = {min(r, hi - 1), max(l, lo), min(r + 1, hi - 1), max(l - 1, lo)};
    double qx[4], dd[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double qy, qz;
      src.load(j[u] + d, qx[u], qy, qz);
      dd[u] = (qx[u] - tx) * (qx[u] - tx) + (qy - ty) * (qy - ty) + (qz - tz) * (qz - tz);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (dd[u] < best) {
        best = dd[u];
        bj = j[u];
      }
    B = fmin(max_dist_sq, best);
    if (ra) {
      r += 2;
      const double g = qx[2] - slack - tx;  // everything right of j[2] has x - tx >= g
      ra = r < hi && !(g > 0.0 && g * g + row2 > B);
    }
    if (la) {
      l -= 2;
      const double g = tx - qx[3] - slack;
      la = l >= lo && !(g > 0.0 && g * g + row2 > B);
    }
  }
}

// Exact 1-NN of (tx, ty, tz) inside the 27-cell cube around its cell (cell edge >= max correspondence distance), branch and bound
// per lane with the bound B = min(max_dist^2, best so far).  `best` / `bj` come in holding the correspondence of the previous
// linearisation re-evaluated under the new pose (or +inf / -1).  The own row of cells first (one sweep over its three cells from
// the query's x outwards), then those of the other eight rows that are still within the bound: every lane works through ITS rows,
// one per round.
template <class Src>
__device__ __forceinline__ void nn_search27(const Src& src, const GicpParams& prm, double tx, double ty, double tz, int cx, int cy, int cz,
                                            double& best, int& bj) {
  double B = fmin(prm.max_dist_sq, best);
  const double ux = tx * prm.inv_cell - (double)(cx - kCoordOffset), uy = ty * prm.inv_cell - (double)(cy - kCoordOffset),
               uz = tz * prm.inv_cell - (double)(cz - kCoordOffset);
  // lower bounds of the distance to the neighbouring rows (a hair conservative: 1e-9 cells)
  const double ly0 = fmax(uy - 1e-9, 0.0) * prm.cell, ly2 = fmax(1.0 - uy - 1e-9, 0.0) * prm.cell;
  const double lz0 = fmax(uz - 1e-9, 0.0) * prm.cell, lz2 = fmax(1.0 - uz - 1e-9, 0.0) * prm.cell;
  const double slack = prm.cell * (1.0 / kFine + 1e-9);
  auto row = [&](int y, int z, double row2) {
    int e[4];
    const int d = src.cells3(cx, y, z, e);
    if (e[3] > e[0]) {
      const int n1 = e[2] - e[1];  // the walk starts where the query's x would fall among the points of its own cell
      const int start = e[1] + min(max((int)(ux * (double)n1), 0), n1);
      nn_sweep(src, tx, ty, tz, e[0], e[3], d, start, row2, slack, prm.max_dist_sq, B, best, bj);
    }
  };
#ifdef GFS_LIN_UTIL
  if (src.util) {  // searches (lanes) and waves that search
    const unsigned long long m = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
      atomicAdd(src.util + 6, (unsigned)__popcll(m));
      atomicAdd(src.util + 7, 1u);
    }
  }
#endif
  row(cy, cz, 0.0);
  unsigned need = 0;
#pragma unroll
  for (int t8 = 0; t8 < 8; t8++) {
    const int t9 = t8 < 4 ? t8 : t8 + 1, dy = t9 % 3, dz = t9 / 3;
    const double a = dy == 0 ? ly0 : dy == 2 ? ly2 : 0.0, c = dz == 0 ? lz0 : dz == 2 ? lz2 : 0.0;
    if (a * a + c * c <= B) need |= 1u << t8;
  }
  while (__any(need != 0)) {
#ifdef GFS_LIN_UTIL
    if (src.util) {  // rounds of neighbouring rows per wave, and the lanes that have one
      const unsigned long long m = __ballot(need != 0);
      if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) {
        atomicAdd(src.util + 2, (unsigned)__popcll(m));
        atomicAdd(src.util + 3, 1u);
      }
    }
#endif
    if (need != 0) {
      const int t8 = __ffs(need) - 1;
      need &= need - 1;
      const int t9 = t8 < 4 ? t8 : t8 + 1, dy = t9 % 3, dz = t9 / 3;
      const double a = dy == 0 ? ly0 : dy == 2 ? ly2 : 0.0, c = dz == 0 ? lz0 : dz == 2 ? lz2 : 0.0;
      const double row2 = a * a + c * c;
      if (row2 <= B) row(cy + dy - 1, cz + dz - 1, row2);
    }
  }
}

// Stages the workgroup's tile (all threads call it; `inrange` = the lane holds a source point whose image has usable cell
// coordinates cx, cy, cz).  Returns 0 -- for the whole workgroup -- when the tile is complete, else why not: 1 no dense grid, 2 no lane
// with a usable image, 3 too many rows in the box, 4 too many points, 5 too many cell boundaries.
// Two dependent round trips to memory: the rows' first / last points out of the grid, then ONE flat copy of all points and
// boundaries (every thread takes slots t, t + 256, ... of the tile, finds the slot's row by bisection over the non-empty rows and
// has all its loads in flight before it stores).
__device__ __forceinline__ int lin_stage_tile(LinTile& T, bool inrange, int cx, int cy, int cz, const double4* __restrict__ tp,
                                              const int* __restrict__ gi, const unsigned* __restrict__ G) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (!gi[6]) return 1;  // no dense grid for this cloud (uniform)
  if (tid < 4) T.box[tid] = (tid & 1) ? INT_MIN : INT_MAX;
  if (tid < kTileRows) {
    T.row_a[tid] = INT_MAX;
    T.row_b[tid] = INT_MIN;
  }
  __syncthreads();
  {
    int v[4] = {inrange ? cy : INT_MAX, inrange ? -cy : INT_MAX, inrange ? cz : INT_MAX, inrange ? -cz : INT_MAX};
#pragma unroll
    for (int ofs = 32; ofs > 0; ofs >>= 1)
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = min(v[k], __shfl_xor(v[k], ofs, 64));
    if (lane == 0 && v[0] != INT_MAX) {
      atomicMin(&T.box[0], v[0]);
      atomicMax(&T.box[1], -v[1]);
      atomicMin(&T.box[2], v[2]);
      atomicMax(&T.box[3], -v[3]);
    }
  }
  __syncthreads();
  const int cymin = T.box[0], cymax = T.box[1], czmin = T.box[2], czmax = T.box[3];
  if (cymin > cymax) return 2;  // no lane with a usable image
  const long long nyl = (long long)cymax - cymin + 3, nzl = (long long)czmax - czmin + 3;
  if (nyl * nzl > kTileRows) return 3;
  const int ny = (int)nyl, nz = (int)nzl, nrows = ny * nz, y0 = cymin - 1, z0 = czmin - 1;
  if (inrange) {  // x range of the lanes centred on each row
    const int r = (cy - y0) + ny * (cz - z0);
    atomicMin(&T.row_a[r], cx);
    atomicMax(&T.row_b[r], cx);
  }
  __syncthreads();
  int lo = INT_MAX, hi = INT_MIN, j0 = 0, len = 0, wid = 0;
  size_t base = 0;
  if (tid < nrows) {  // a row serves the lanes centred on it and on its eight neighbours
    const int ry = tid % ny, rz = tid / ny;
#pragma unroll
    for (int dz = -1; dz <= 1; dz++)
#pragma unroll
      for (int dy = -1; dy <= 1; dy++) {
        const int yy = ry + dy, zz = rz + dz;
        if (yy >= 0 && yy < ny && zz >= 0 && zz < nz) {
          lo = min(lo, T.row_a[yy + ny * zz]);
          hi = max(hi, T.row_b[yy + ny * zz]);
        }
      }
    if (lo <= hi) {
      lo -= 1;
      hi += 1;
      const int gy = y0 + ry - gi[1], gz = z0 + rz - gi[2];
      if ((unsigned)gy < (unsigned)gi[4] && (unsigned)gz < (unsigned)gi[5]) {
        base = ((size_t)gz * gi[4] + gy) * gi[3];
        j0 = (int)G[base + min(max(lo - gi[0], 0), gi[3])];
        const int j1 = (int)G[base + min(max(hi + 1 - gi[0], 0), gi[3])];
        len = max(j1 - j0, 0);
        wid = len > 0 ? hi - lo + 2 : 0;  // the boundaries of cells lo .. hi + 1
      }
    }
  }
  // exclusive prefix sums of (points, boundaries, non-empty rows) over the rows: 24 + 24 + 16 bits, wave scan + four wave totals
  // (a row longer than the tile cannot pass: clamp its length so that the packed sum cannot overflow its field)
  const unsigned long long pk = (unsigned long long)min(len, kTileCap + 1) | ((unsigned long long)min(wid, kTileCells + 1) << 24) |
                                ((unsigned long long)(len > 0 ? 1 : 0) << 48);
  unsigned long long inc = pk;
#pragma unroll
  for (int ofs = 1; ofs < 64; ofs <<= 1) {
    const unsigned long long o = __shfl_up(inc, ofs, 64);
    if (lane >= ofs) inc += o;
  }
  __syncthreads();  // every thread has read its neighbours' row_a / row_b: they are reused below
  if (lane == 63) T.wave_tot[wave] = inc;
  __syncthreads();
  unsigned long long before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const unsigned long long tw = T.wave_tot[w];
    if (w < wave) before += tw;
    total += tw;
  }
  const int tot_pts = (int)(total & 0xffffffu), tot_cells = (int)((total >> 24) & 0xffffffu), nnz = (int)(total >> 48);
  if (tot_pts > kTileCap) return 4;  // uniform
  if (tot_cells > kTileCells) return 5;
  if (nnz > kTileNz) return 3;
  if (tid < nrows) {
    const unsigned long long exc = before + inc - pk;
    const int poff = (int)(exc & 0xffffffu), coff = (int)((exc >> 24) & 0xffffffu), k = (int)(exc >> 48);
    T.row_lo[tid] = lo;
    T.row_delta[tid] = poff - j0;
    T.row_coff[tid] = len > 0 ? coff : -1;
    if (len > 0) {  // the non-empty rows, in order: where their points and boundaries end in the tile, and where they come from
      T.nz_pend[k] = poff + len;
      T.nz_cend[k] = coff + wid;
      T.nz_delta[k] = poff - j0;
      T.nz_cbase[k] = lo - gi[0] - coff;  // boundary slot s holds G[base + clamp(s + cbase, 0, nx)] (the grid has < 2^21 cells)
      T.nz_base[k] = (int)base;
    }
  }
  __syncthreads();
  {
    constexpr int kU = (kTileCap + kLinBlock - 1) / kLinBlock;
    double4 q[kU];
    int slot[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      slot[u] = tid + u * kLinBlock;
      if (slot[u] < tot_pts) {
        int a = 0, b = nnz - 1;  // first non-empty row whose end lies beyond the slot
        while (a < b) {
          const int mid = (a + b) >> 1;
          if (T.nz_pend[mid] > slot[u]) b = mid;
          else a = mid + 1;
        }
        q[u] = tp[slot[u] - T.nz_delta[a]];
      }
    }
#pragma unroll
    for (int u = 0; u < kU; u++)
      if (slot[u] < tot_pts) {
        T.x[slot[u]] = q[u].x;
        T.y[slot[u]] = q[u].y;
        T.z[slot[u]] = q[u].z;
      }
    constexpr int kUc = (kTileCells + kLinBlock - 1) / kLinBlock;
    unsigned cv[kUc];
#pragma unroll
    for (int u = 0; u < kUc; u++) {
      const int s = tid + u * kLinBlock;
      if (s < tot_cells) {
        int a = 0, b = nnz - 1;
        while (a < b) {
          const int mid = (a + b) >> 1;
          if (T.nz_cend[mid] > s) b = mid;
          else a = mid + 1;
        }
        cv[u] = G[T.nz_base[a] + min(max(s + T.nz_cbase[a], 0), gi[3])];
      }
    }
#pragma unroll
    for (int u = 0; u < kUc; u++) {
      const int s = tid + u * kLinBlock;
      if (s < tot_cells) T.cells[s] = cv[u];
    }
  }
  __syncthreads();
  return 0;
}

// ---- the factor of one correspondence (factors/gicp_factor.hpp:52-73), round 5.
// With J = [R skew(p) | -R] = R [S | -I] (S = skew(p), twist order rotation, translation) the quadratic form is
//   J' M J = [S | -I]' N [S | -I],  N = R' M R,  J' M r = [S | -I]' u,  u = N w,  w = R' r,  r' M r = w' u
// so only N (6), Q = N S (9: each column two products, S has two non-zeros a column), the cross products Q(:,c) x p and u x p are
// formed -- ~100 fused multiply-adds behind the Mahalanobis matrix instead of the ~175 of the dense J, M J, J' (M J) -- and, what
// matters more, the state that has to stay in registers between the search and the reduction is N, Q, p, u, w (48 VGPRs) instead of
// J, M J and 29 accumulators: the kernel fits 80 VGPRs = 6 waves a SIMD (112 = 4 before), and the search, a chain of dependent
// look-ups, is paid in occupancy.  The wave reduction runs in two batches for the same reason (B first: it needs p, u, w; A is N
// and Q themselves).  M = (Ct + R Cs R')^-1 is formed exactly as before (k_gicp_error reads it).  Same sums of the same kind of
// terms: the results differ from round 4's in their last bits (the bar on the pose is 1e-5, batched = single stays bit for bit).
struct LinFactor {
  double N[6];  // xx xy xz yy yz zz
  double Q[9];  // column-major N S
  double px, py, pz, u[3], w[3];
  bool on;
};
// index of each batch value in the 29-vector (upper H row-major over (r, c >= r), then b, e, count)
__device__ constexpr int kLinMapA[15] = {3, 8, 12, 4, 9, 13, 5, 10, 14, 15, 16, 17, 18, 19, 20};  // H_rt (column-major of -Q'), H_tt
__device__ constexpr int kLinMapB[14] = {0, 1, 2, 6, 7, 11, 21, 22, 23, 24, 25, 26, 27, 28};       // H_rr, b, e, count
// (a0 b0 + a1 b1 + a2 b2) and c + (...) as chains of explicit fused multiply-adds.  The bar for GICP is 1e-5 on the pose (the sums
// are re-associated by the reduction tree anyway), so the factor uses FMAs -- written out, NOT left to `fp contract(fast)`: which
// products the compiler fuses may differ between two instantiations of the same source (k_gicp_linearize, coop_lin_chunk,
// k_gicp_lm), and the three must give the same bits (round 6: one pair in ~60 differed in the last bit of e).  The search stays
// un-contracted so that the correspondences are decided on the same distances as in the oracle.
__device__ __forceinline__ double fma3(double a0, double b0, double a1, double b1, double a2, double b2) {
  return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0));
}
__device__ __forceinline__ double fma3p(double c, double a0, double b0, double a1, double b1, double a2, double b2) {
  return __builtin_fma(a2, b2, __builtin_fma(a1, b1, __builtin_fma(a0, b0, c)));
}
__device__ __forceinline__ double fmsub2(double a0, double b0, double a1, double b1) {  // a0 b0 - a1 b1
  return __builtin_fma(a0, b0, -(a1 * b1));
}
__device__ __forceinline__ void lin_factor_prepare(int i, const double4 p, double tx, double ty, double tz, int ti, const double* __restrict__ T12,
                                                   int pair, int cs, int ct, int P, const double4* __restrict__ pts,
                                                   const double* __restrict__ cov6, double* __restrict__ maha6, LinFactor& F) {
  F.on = ti >= 0;
  if (!F.on) return;
  const double* R = T12;
  const double4* tp = pts + (size_t)ct * P;
  const double* Cs = cov6 + ((size_t)cs * P + i) * 6;
  const double* Ct = cov6 + ((size_t)ct * P + ti) * 6;
  const double cs9[9] = {Cs[0], Cs[1], Cs[2], Cs[1], Cs[3], Cs[4], Cs[2], Cs[4], Cs[5]};
  double RC[9];
  for (int cc = 0; cc < 3; cc++)
    for (int r = 0; r < 3; r++) RC[r + 3 * cc] = fma3(R[r], cs9[3 * cc], R[r + 3], cs9[3 * cc + 1], R[r + 6], cs9[3 * cc + 2]);
  const double ct9[9] = {Ct[0], Ct[1], Ct[2], Ct[1], Ct[3], Ct[4], Ct[2], Ct[4], Ct[5]};
  double A[9];
  for (int cc = 0; cc < 3; cc++)
    for (int r = 0; r < 3; r++) A[r + 3 * cc] = fma3p(ct9[r + 3 * cc], RC[r], R[cc], RC[r + 3], R[cc + 3], RC[r + 6], R[cc + 6]);
  double M[9];
  inv3(A, M);
  double* mo = maha6 + ((size_t)pair * P + i) * 6;
  mo[0] = M[0];
  mo[1] = M[3];
  mo[2] = M[6];
  mo[3] = M[4];
  mo[4] = M[7];
  mo[5] = M[8];
  const double4 q = tp[ti];
  const double res[3] = {q.x - tx, q.y - ty, q.z - tz};
  double MR[9];  // M R
  for (int cc = 0; cc < 3; cc++)
    for (int r = 0; r < 3; r++) MR[r + 3 * cc] = fma3(M[r], R[3 * cc], M[r + 3], R[3 * cc + 1], M[r + 6], R[3 * cc + 2]);
  // N = R' (M R), upper triangle
  auto rtm = [&](int a, int b) { return fma3(R[3 * a], MR[3 * b], R[3 * a + 1], MR[3 * b + 1], R[3 * a + 2], MR[3 * b + 2]); };
  F.N[0] = rtm(0, 0);
  F.N[1] = rtm(0, 1);
  F.N[2] = rtm(0, 2);
  F.N[3] = rtm(1, 1);
  F.N[4] = rtm(1, 2);
  F.N[5] = rtm(2, 2);
  const double n9[9] = {F.N[0], F.N[1], F.N[2], F.N[1], F.N[3], F.N[4], F.N[2], F.N[4], F.N[5]};  // (symmetric: rows = columns)
  for (int k = 0; k < 3; k++) F.w[k] = fma3(R[3 * k], res[0], R[3 * k + 1], res[1], R[3 * k + 2], res[2]);
  for (int r = 0; r < 3; r++) F.u[r] = fma3(n9[r], F.w[0], n9[r + 3], F.w[1], n9[r + 6], F.w[2]);
  F.px = p.x;
  F.py = p.y;
  F.pz = p.z;
  // Q = N S, S = skew(p): S e0 = (0, pz, -py), S e1 = (-pz, 0, px), S e2 = (py, -px, 0)
  for (int r = 0; r < 3; r++) {
    F.Q[r] = fmsub2(p.z, n9[r + 3], p.y, n9[r + 6]);
    F.Q[r + 3] = fmsub2(p.x, n9[r + 6], p.z, n9[r]);
    F.Q[r + 6] = fmsub2(p.y, n9[r], p.x, n9[r + 3]);
  }
}
// batch B: H_rr = S' N S = Q(:,c) x p (upper triangle), b = (u x p, -u), e = w' u / 2, inlier count
__device__ __forceinline__ void lin_factor_batch_b(const LinFactor& F, double (&v)[14]) {
#pragma unroll
  for (int k = 0; k < 14; k++) v[k] = 0;
  if (!F.on) return;
  // (a x p)_0 = a1 pz - a2 py, _1 = a2 px - a0 pz, _2 = a0 py - a1 px
  v[0] = fmsub2(F.Q[1], F.pz, F.Q[2], F.py);  // H(0,0): (Q(:,0) x p)_0
  v[1] = fmsub2(F.Q[4], F.pz, F.Q[5], F.py);  // H(0,1): (Q(:,1) x p)_0
  v[2] = fmsub2(F.Q[7], F.pz, F.Q[8], F.py);  // H(0,2)
  v[3] = fmsub2(F.Q[5], F.px, F.Q[3], F.pz);  // H(1,1): (Q(:,1) x p)_1
  v[4] = fmsub2(F.Q[8], F.px, F.Q[6], F.pz);  // H(1,2)
  v[5] = fmsub2(F.Q[6], F.py, F.Q[7], F.px);  // H(2,2): (Q(:,2) x p)_2
  v[6] = fmsub2(F.u[1], F.pz, F.u[2], F.py);
  v[7] = fmsub2(F.u[2], F.px, F.u[0], F.pz);
  v[8] = fmsub2(F.u[0], F.py, F.u[1], F.px);
  v[9] = -F.u[0];
  v[10] = -F.u[1];
  v[11] = -F.u[2];
  v[12] = 0.5 * fma3(F.w[0], F.u[0], F.w[1], F.u[1], F.w[2], F.u[2]);
  v[13] = 1.0;
}
// batch A: H_rt = -S' N = -Q' as its entries (r, 3 + c) = -Q(c, r), and H_tt = N
__device__ __forceinline__ void lin_factor_batch_a(const LinFactor& F, double (&v)[15]) {
#pragma unroll
  for (int k = 0; k < 15; k++) v[k] = 0;
  if (!F.on) return;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) v[3 * c + r] = -F.Q[c + 3 * r];  // value 3 c + r -> H(r, 3 + c), kLinMapA
#pragma unroll
  for (int k = 0; k < 6; k++) v[9 + k] = F.N[k];
}
// the 29-vector of a point (callers that reduce all of it at once)
__device__ __forceinline__ void lin_factor_accumulate(const LinFactor& F, double (&acc)[kRed]) {
  double a[15], b[14];
  lin_factor_batch_a(F, a);
  lin_factor_batch_b(F, b);
#pragma unroll
  for (int k = 0; k < 15; k++) acc[kLinMapA[k]] += a[k];
#pragma unroll
  for (int k = 0; k < 14; k++) acc[kLinMapB[k]] += b[k];
}

// GICPFactor::linearize (factors/gicp_factor.hpp:35-73) of source point i of a pair under the pose T12 (R col-major | t): exact
// 1-NN in the target cloud (through `src`), rejection beyond max_dist, the Mahalanobis matrix, and this point's terms ADDED to
// acc (H upper triangle 21, b 6, e, inlier count).  Records the correspondence and the matrix for the error evaluations that follow.
template <class Src>
__device__ __forceinline__ void gicp_lin_point(const Src& src, int i, const double4 p, double tx, double ty, double tz, int cx, int cy, int cz,
                                               bool in_range, const double* __restrict__ T12, bool has_prev, int prev_j, const double4 prev_q,
                                               int pair, int cs, int ct, int P,
                                               const double4* __restrict__ pts, const double* __restrict__ cov6,
                                               const u64* __restrict__ ucell, const unsigned* __restrict__ ubegin,
                                               const int* __restrict__ n_ucell, const unsigned* __restrict__ G,
                                               const int* __restrict__ gi, const GicpParams& prm, int* __restrict__ tgt_index,
                                               double* __restrict__ maha6, LinFactor& F) {
  {
    const double4* tp = pts + (size_t)ct * P;
    const u64* uc = ucell + (size_t)ct * (P + 1);
    const unsigned* ub = ubegin + (size_t)ct * (P + 1);
    const int nu = n_ucell[ct];
    double best = 1.79769313486231570e308;
    int bj = -1;
    if (in_range) {
      if (prm.nn_rings <= 1) {
        if (has_prev && prev_j >= 0) {  // the previous correspondence under the new pose: the first bound
          best = (prev_q.x - tx) * (prev_q.x - tx) + (prev_q.y - ty) * (prev_q.y - ty) + (prev_q.z - tz) * (prev_q.z - tz);
          bj = prev_j;
        }
        nn_search27(src, prm, tx, ty, tz, cx, cy, cz, best, bj);
      } else {
      // (cell edge < max correspondence distance: experiment knob GFS_GICP_CELL)
      // exact 1-NN by growing cubes of cells: after probing radius r every unvisited point is farther than r*cell,
      // so the search stops as soon as the best distance is certified (usually r = 1), or at nn_rings (>= max_corr).
      // r = 1 fast path: 9 row ranges fetched up-front, candidates loaded four at a time (loads kept in flight).
      {
        int j0s[9], j1s[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++)
          row_range(gi, G, uc, ub, nu, cx - 1, cx + 1, cy + (t9 % 3) - 1, cz + t9 / 3 - 1, &j0s[t9], &j1s[t9]);
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++) {
          const int jb = j0s[t9], je = j1s[t9];
          for (int j = jb; j < je; j += 4) {
            const int ja = min(j + 1, je - 1), jb2 = min(j + 2, je - 1), jc = min(j + 3, je - 1);
            const double4 q0 = tp[j], q1 = tp[ja], q2 = tp[jb2], q3 = tp[jc];
            const double d0 = (q0.x - tx) * (q0.x - tx) + (q0.y - ty) * (q0.y - ty) + (q0.z - tz) * (q0.z - tz);
            const double d1 = (q1.x - tx) * (q1.x - tx) + (q1.y - ty) * (q1.y - ty) + (q1.z - tz) * (q1.z - tz);
            const double d2 = (q2.x - tx) * (q2.x - tx) + (q2.y - ty) * (q2.y - ty) + (q2.z - tz) * (q2.z - tz);
            const double d3 = (q3.x - tx) * (q3.x - tx) + (q3.y - ty) * (q3.y - ty) + (q3.z - tz) * (q3.z - tz);
            // clamped duplicates of the last candidate cannot win the strict '<'
            if (d0 < best) { best = d0; bj = j; }
            if (d1 < best) { best = d1; bj = ja; }
            if (d2 < best) { best = d2; bj = jb2; }
            if (d3 < best) { best = d3; bj = jc; }
          }
        }
      }
      const bool done1 = (bj >= 0 && best <= prm.cell * prm.cell) || prm.nn_rings <= 1;
      for (int r = 2; !done1 && r <= prm.nn_rings; r++) {
        best = 1.79769313486231570e308;
        bj = -1;
        for (int dz = -r; dz <= r; dz++)
          for (int dy = -r; dy <= r; dy++) {
            int j0, j1;
            row_range(gi, G, uc, ub, nu, cx - r, cx + r, cy + dy, cz + dz, &j0, &j1);
            for (int j = j0; j < j1; j++) {
              const double4 q = tp[j];
              const double dx = q.x - tx, dy2 = q.y - ty, dz2 = q.z - tz;
              const double d = dx * dx + dy2 * dy2 + dz2 * dz2;
              if (d < best) {
                best = d;
                bj = j;
              }
            }
          }
        const double reach = (double)r * prm.cell;
        if (bj >= 0 && best <= reach * reach) break;
      }
      }
    }
    const int ti = (bj >= 0 && !(best > prm.max_dist_sq)) ? bj : -1;  // DistanceRejector: reject iff sq_dist > max_dist_sq
    lin_factor_prepare(i, p, tx, ty, tz, ti, T12, pair, cs, ct, P, pts, cov6, maha6, F);
    tgt_index[(size_t)pair * P + i] = ti;
  }
}


// a source point, its image under the pose and the image's cell
__device__ __forceinline__ void lin_image(const double4& p, const double* __restrict__ T12, const GicpParams& prm, double& tx, double& ty,
                                          double& tz, int& cx, int& cy, int& cz, bool& in_range) {
  const double* R = T12;
  const double* t = T12 + 9;
  tx = R[0] * p.x + R[3] * p.y + R[6] * p.z + t[0];
  ty = R[1] * p.x + R[4] * p.y + R[7] * p.z + t[1];
  tz = R[2] * p.x + R[5] * p.y + R[8] * p.z + t[2];
  cx = fast_floor_d(tx * prm.inv_cell) + kCoordOffset;
  cy = fast_floor_d(ty * prm.inv_cell) + kCoordOffset;
  cz = fast_floor_d(tz * prm.inv_cell) + kCoordOffset;
  in_range = fabs(tx) < 2.0e4 && fabs(ty) < 2.0e4 && fabs(tz) < 2.0e4;
}

// GICPFactor::linearize (factors/gicp_factor.hpp:35-73) for every source point of every pair whose state
// machine is in the "linearise" phase; per-block partial sums of H (upper), b, e and the inlier count.
#ifdef GFS_LIN_WAVES
#define GFS_LIN_OCC __attribute__((amdgpu_waves_per_eu(GFS_LIN_WAVES, 8)))
#else
// 5 waves a SIMD is what the pass takes by itself (91 - 94 VGPRs); said explicitly because the scalar step it calls in its last
// workgroup (pair_step_last, not inlined) is otherwise scheduled without an occupancy target, takes 180 and drags the kernel to 2
#define GFS_LIN_OCC __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif
struct CoopSync {
  unsigned arrive;  // k_gicp_linearize: workgroups of the pair that have finished this pass (the last one resets it);
                    // k_gicp_lm_coop: arrivals at the pair's barriers (monotonic within the launch); k_gicp_init zeroes it
  unsigned gen;     // k_gicp_lm_coop: barriers completed
};
// (defined behind the scalar steps below) the pair's fold + scalar step, run by the workgroup of a pass that finishes last
__device__ void pair_step_last(PairState* S, const double* part, const double* epart, int nblk_pair, int phase, double rot_eps,
                               double trans_eps, int max_iterations, int pair, int* n_done, int* act_next, int* n_act_next, double* s_part,
                               double* s_sum);

// GICPFactor::error of a pending trial for source point i: 0.5 r' M r with the frozen correspondence q and matrix M (factors/gicp_factor.hpp:76-86);
// (tx, ty, tz) = the point under the trial pose as lin_image computes it (the same expression, un-contracted)
__device__ __forceinline__ double gicp_trial_error(const double4 q, double tx, double ty, double tz, const double* __restrict__ M) {
  const double r0 = q.x - tx, r1 = q.y - ty, r2 = q.z - tz;
  const double m0 = M[0] * r0 + M[1] * r1 + M[2] * r2, m1 = M[1] * r0 + M[3] * r1 + M[4] * r2, m2 = M[2] * r0 + M[4] * r1 + M[5] * r2;
  return 0.5 * (r0 * m0 + r1 * m1 + r2 * m2);
}

// One pass over the source points of every pair that is not done (PairState::phase):
//   phase 0: GICPFactor::linearize at T;
//   phase 1: the error of the pending trial newT with the frozen correspondences AND the linearisation at newT, written to the pair's
//            OTHER correspondence buffer -- both need the point under newT and the previous correspondence (the search's first bound is
//            the error's residual), so the error costs six more loads;
//   phase 3: the trial's error only.
// Per-block partial sums of H (upper), b, e, the inlier count (partial) and of the trial's error (epartial).
// kFuse (256-thread workgroups): the pair's scalar step (k_gicp_step) is taken by the workgroup of the pair that finishes LAST, inside
// this launch -- a round of the LM loop is ONE launch.  The partial sums then cross workgroups within the launch: they are stored
// write-through (st_ag), every wave drains its stores before the workgroup's arrival is counted (one agent-scope atomic a
// workgroup), and the last arriver folds them with L1-bypassing loads -- no release / acquire fence (round 5 had measured the
// fence form of this pattern: 2.97 -> 23.8 ms a step).  Nobody waits for anybody: no residency requirement.
template <bool kTiled, bool kFuse>
__global__ __launch_bounds__(kLinBlock) GFS_LIN_OCC void k_gicp_linearize(PairState* __restrict__ st,
                                                              const double4* __restrict__ pts,
                                                              const double* __restrict__ cov6, const u64* __restrict__ ucell,
                                                              const unsigned* __restrict__ ubegin,
                                                              const int* __restrict__ n_ucell, const int* __restrict__ m_counts,
                                                              const unsigned* __restrict__ grid, const int* __restrict__ ginfo,
                                                              int nchunks, int npairs, int P, GicpParams prm,
                                                              int* __restrict__ tgt_index, double* __restrict__ maha6, size_t buf_stride,
                                                              double* __restrict__ partial, double* __restrict__ epartial, int nblk,
                                                              const int* __restrict__ act, const int* __restrict__ n_act,
                                                              CoopSync* __restrict__ sync, int* __restrict__ n_done,
                                                              int* __restrict__ act_next, int* __restrict__ n_act_next,
                                                              int* __restrict__ n_act_clear) {
  __shared__ typename std::conditional<kTiled, LinTile, int>::type tile;  // the untiled instance keeps its LDS (and its occupancy)
  if constexpr (kFuse) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_act_clear = 0;  // the counter the NEXT round's last arrivers fill
  }
  int pair, sub, chunk;
  if (act) {  // a late round: only the pairs on the list (the grid covers an upper bound of their number)
    const int slot = blockIdx.x / nchunks;
    if (slot >= *n_act) return;
    pair = act[slot];
    chunk = blockIdx.x - slot * nchunks;
  } else if (!xcd_pair_map(nchunks, npairs, 1, &pair, &sub, &chunk)) {
    return;
  }
  // Everything this workgroup can ask for before it knows anything is asked for first -- the pair's state, the cloud sizes, the
  // grid header, the lane's source point and its previous correspondence -- so that the dependent round trips to memory that
  // remain are: state -> (tile rows) -> (tile points) -> search in LDS -> target covariance.
  const int cs = 2 * pair + prm.src_slot, ct = 2 * pair + 1 - prm.src_slot;
  const PairState* Sp = st + pair;
  const int phase = Sp->phase, n_lin = Sp->n_lin, cur = Sp->buf;
  const double* Tp = phase == 0 ? Sp->T : Sp->newT;
  double T12[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T12[k] = Tp[k];
  const int ms = m_counts[cs];
  const unsigned* G = grid + (size_t)ct * (kGridCap + 1);
  const int* gi = ginfo + 8 * ct;
  int wave_slot;
  const int i = lin_point_of(chunk, &wave_slot);  // < P: the grid covers at most the clouds' capacity
  const double4 p = pts[(size_t)cs * P + i];
  const double4* tp = pts + (size_t)ct * P;
  const int* ti_cur = tgt_index + (size_t)cur * buf_stride;
  const double* maha_cur = maha6 + (size_t)cur * buf_stride * 6;
  const int prev_j = ti_cur[(size_t)pair * P + i];
  const int nblk_pair = (ms + kLinBlock - 1) / kLinBlock;
  // (kFuse: a pair without source points still takes its scalar steps -- chunk 0's workgroup stays for that)
  if (phase == 2 || chunk >= (kFuse ? max(nblk_pair, 1) : nblk_pair)) return;
  const bool has_prev = n_lin > 0;
  double4 prev_q = make_double4(0, 0, 0, 0);
  if (has_prev && i < ms && prev_j >= 0) prev_q = tp[prev_j];
  LinFactor F;
  F.on = false;
  double tx = 0, ty = 0, tz = 0;
  int cx = 0, cy = 0, cz = 0;
  bool in_range = false;
  if (i < ms) lin_image(p, T12, prm, tx, ty, tz, cx, cy, cz, in_range);
  if (phase != 0) {  // the pending trial's error (the separate k_gicp_error pass of rounds 1 - 5)
    double e[1] = {0.0};
    if (i < ms && prev_j >= 0) e[0] = gicp_trial_error(prev_q, tx, ty, tz, maha_cur + ((size_t)pair * P + i) * 6);
    wave_reduce_store<1, kFuse>(e, epartial + ((size_t)pair * nblk + chunk) * kLinWaves, wave_slot);
  }
  if (phase != 3) {
    const int wr = phase == 0 ? cur : cur ^ 1;
    int* ti_wr = tgt_index + (size_t)wr * buf_stride;
    double* maha_wr = maha6 + (size_t)wr * buf_stride * 6;
    bool tiled = false;
    if constexpr (kTiled) {
      const int why = prm.nn_rings <= 1 ? lin_stage_tile(tile, i < ms && in_range, cx, cy, cz, tp, gi, G) : 6;
      tiled = why == 0;
      if (prm.tile_stats && threadIdx.x == 0) atomicAdd(prm.tile_stats + why, 1u);
    }
    if (i < ms) {
      bool done = false;
      if constexpr (kTiled) {
        if (tiled) {
          const NnTile src{&tile, tile.box[0] - 1, tile.box[2] - 1, tile.box[1] - tile.box[0] + 3};
          gicp_lin_point(src, i, p, tx, ty, tz, cx, cy, cz, in_range, T12, has_prev, prev_j, prev_q, pair, cs, ct, P, pts, cov6, ucell, ubegin,
                         n_ucell, G, gi, prm, ti_wr, maha_wr, F);
          done = true;
        }
      }
      if (!done) {
#ifdef GFS_LIN_UTIL
        const NnGlobal src{tp, gi, G, ucell + (size_t)ct * (P + 1), ubegin + (size_t)ct * (P + 1), n_ucell[ct], prm.tile_stats};
#else
        const NnGlobal src{tp, gi, G, ucell + (size_t)ct * (P + 1), ubegin + (size_t)ct * (P + 1), n_ucell[ct]};
#endif
        gicp_lin_point(src, i, p, tx, ty, tz, cx, cy, cz, in_range, T12, has_prev, prev_j, prev_q, pair, cs, ct, P, pts, cov6, ucell, ubegin, n_ucell,
                       G, gi, prm, ti_wr, maha_wr, F);
      }
    }
    double* dst = partial + ((size_t)pair * nblk + chunk) * kLinWaves * kRed;
    {
      double vb[14];
      lin_factor_batch_b(F, vb);
      wave_reduce_store_map<14, kFuse>(vb, kLinMapB, dst, wave_slot);
    }
    {
      double va[15];
      lin_factor_batch_a(F, va);
      wave_reduce_store_map<15, kFuse>(va, kLinMapA, dst, wave_slot);
    }
  }
  if constexpr (kFuse) {
    __shared__ double s_part[8 * 32], s_sum[32];
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave: its partial sums have left before the arrival is counted
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned old = __hip_atomic_fetch_add(&sync[pair].arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old + 1 == (unsigned)max(nblk_pair, 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
      pair_step_last(st + pair, partial + (size_t)pair * nblk * kLinWaves * kRed, epartial + (size_t)pair * nblk * kLinWaves, nblk_pair, phase,
                     prm.rot_eps, prm.trans_eps, prm.max_iterations, pair, n_done, act_next, n_act_next, s_part, s_sum);
      if (threadIdx.x == 0) st_ag(&sync[pair].arrive, 0u);  // (everybody of this pair has arrived; the next pass is another launch)
    }
  }
}

// (H + lambda I) delta = -b, then new_T = T * se3_exp(delta) (registration/optimizer.hpp:109-112, util/lie.hpp:54-103).
// H + lambda I is symmetric positive definite (lambda > 0), so an un-pivoted LDL^T is used: every loop has
// compile-time bounds and everything stays in registers (Eigen's LDLT pivots; the solutions agree to rounding).
__device__ void solve_and_propose(const double* H21, const double* b6, double lambda, const double* T12, double* delta6,
                                  double* newT12) {
  double A[6][6];
  {
    int o = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++) {
        A[r][c] = H21[o];
        A[c][r] = H21[o];
        o++;
      }
  }
#pragma unroll
  for (int k = 0; k < 6; k++) A[k][k] += lambda;
  double d[6], x[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    double dj = A[j][j];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < j) dj -= A[j][k] * A[j][k] * d[k];
    d[j] = dj;
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (i > j) {
        double v = A[i][j];
#pragma unroll
        for (int k = 0; k < 6; k++)
          if (k < j) v -= A[i][k] * A[j][k] * d[k];
        A[i][j] = dj != 0 ? v / dj : 0.0;
      }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double v = -b6[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < i) v -= A[i][k] * x[k];
    x[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) x[i] = d[i] != 0 ? x[i] / d[i] : 0.0;
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double v = x[i];
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k > i) v -= A[k][i] * x[k];
    x[i] = v;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) delta6[i] = x[i];
  // se3_exp
  const double w0 = x[0], w1 = x[1], w2 = x[2];
  const double theta_sq = w0 * w0 + w1 * w1 + w2 * w2;
  const double theta = sqrt(theta_sq);
  double imag, real;
  if (theta_sq < 1e-10) {
    const double tq = theta_sq * theta_sq;
    imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * tq;
    real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * tq;
  } else {
    const double ht = 0.5 * theta;
    imag = gfs_glibc::sin(ht) / theta;  // the reference's libm calls, with glibc's bits (glibc_math.hpp)
    real = gfs_glibc::cos(ht);
  }
  const double qw = real, qx = imag * w0, qy = imag * w1, qz = imag * w2;
  double E[12];
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    E[0] = 1 - (tyy + tzz);
    E[3] = txy - twz;
    E[6] = txz + twy;
    E[1] = txy + twz;
    E[4] = 1 - (txx + tzz);
    E[7] = tyz - twx;
    E[2] = txz - twy;
    E[5] = tyz + twx;
    E[8] = 1 - (txx + tyy);
  }
  const double v0 = x[3], v1 = x[4], v2 = x[5];
  if (theta < 1e-10) {
#pragma unroll
    for (int i = 0; i < 3; i++) E[9 + i] = E[i] * v0 + E[i + 3] * v1 + E[i + 6] * v2;
  } else {
    const double O[9] = {0, w2, -w1, -w2, 0, w0, w1, -w0, 0};
    double O2[9];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) O2[r + 3 * c] = O[r] * O[3 * c] + O[r + 3] * O[3 * c + 1] + O[r + 6] * O[3 * c + 2];
    const double k1 = (1.0 - gfs_glibc::cos(theta)) / theta_sq, k2 = (theta - gfs_glibc::sin(theta)) / (theta_sq * theta);
    double V[9];
#pragma unroll
    for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + k1 * O[i] + k2 * O2[i];
#pragma unroll
    for (int i = 0; i < 3; i++) E[9 + i] = V[i] * v0 + V[i + 3] * v1 + V[i + 6] * v2;
  }
  // newT = T * E
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < 3; i++) newT12[i + 3 * c] = T12[i] * E[3 * c] + T12[i + 3] * E[3 * c + 1] + T12[i + 6] * E[3 * c + 2];
#pragma unroll
  for (int i = 0; i < 3; i++) newT12[9 + i] = T12[i] * E[9] + T12[i + 3] * E[10] + T12[i + 6] * E[11] + T12[9 + i];
}

// The two scalar steps of LevenbergMarquardtOptimizer::optimize (registration/optimizer.hpp:83-147) on a pair's state, shared by the
// launch-per-step kernel (k_gicp_step: S in global memory) and the cooperative kernel (k_gicp_lm_coop: S is a copy
// in LDS) -- one body, so that the two forms cannot drift apart by a bit.
// (split so that a caller has ONE call site of the damped solve: its 6x6 LDL^T holds ~120 registers)
// the damped solve of the state's last linearisation: delta and the trial pose
__device__ __forceinline__ void lm_solve(PairState& S) {
  double H[21], b[6], T[12], delta[6], newT[12];
#pragma unroll
  for (int k = 0; k < 21; k++) H[k] = S.H[k];
#pragma unroll
  for (int k = 0; k < 6; k++) b[k] = S.b[k];
#pragma unroll
  for (int k = 0; k < 12; k++) T[k] = S.T[k];
  solve_and_propose(H, b, S.lambda, T, delta, newT);
#pragma unroll
  for (int k = 0; k < 6; k++) S.delta[k] = delta[k];
#pragma unroll
  for (int k = 0; k < 12; k++) S.newT[k] = newT[k];
}
// after a linearisation: sums = the folded 29-vector (H upper 21 | b 6 | e | inlier count); lm_solve follows
__device__ __forceinline__ void lm_after_linearize(PairState& S, const double* sums) {
#pragma unroll
  for (int k = 0; k < 21; k++) S.H[k] = sums[k];
#pragma unroll
  for (int k = 0; k < 6; k++) S.b[k] = sums[21 + k];
  S.e = sums[27];
  S.inliers = (int)(sums[28] + 0.5);
  S.n_lin++;
  S.inner = 0;
  S.phase = 1;
}
// after an error pass over the trial pose: accept / reject (optimizer.hpp:115-141).  *done: the pair has finished (phase 2);
// returns true when the trial was rejected and another damped solve (lm_solve, with the raised lambda) follows.
__device__ __forceinline__ bool lm_after_error(PairState& S, double new_e, const GicpParams& prm, bool* done) {
  S.n_err++;
  *done = false;
  if (new_e <= S.e) {
    const double dr = sqrt(S.delta[0] * S.delta[0] + S.delta[1] * S.delta[1] + S.delta[2] * S.delta[2]);
    const double dt = sqrt(S.delta[3] * S.delta[3] + S.delta[4] * S.delta[4] + S.delta[5] * S.delta[5]);
    S.converged = (dr <= prm.rot_eps && dt <= prm.trans_eps) ? 1 : 0;
    for (int k = 0; k < 12; k++) S.T[k] = S.newT[k];
    S.lambda /= 10.0;
    S.iterations = S.outer;
    S.outer++;
    if (S.converged || S.outer >= prm.max_iterations) {
      S.phase = 2;
      *done = true;
    } else {
      S.phase = 0;
    }
    return false;
  }
  S.lambda *= 10.0;
  S.inner++;
  if (S.inner >= 10) {  // max_inner_iterations: !success -> break
    S.iterations = S.outer;
    S.phase = 2;
    *done = true;
    return false;
  }
  return true;
}

// would accepting the pending trial end the loop?  (optimizer.hpp:119-121 decides with the same delta; 138: the iteration cap)
__device__ __forceinline__ bool lm_trial_is_last(const PairState& S, const GicpParams& prm) {
  const double dr = sqrt(S.delta[0] * S.delta[0] + S.delta[1] * S.delta[1] + S.delta[2] * S.delta[2]);
  const double dt = sqrt(S.delta[3] * S.delta[3] + S.delta[4] * S.delta[4] + S.delta[5] * S.delta[5]);
  return (dr <= prm.rot_eps && dt <= prm.trans_eps) || S.outer + 1 >= prm.max_iterations;
}
// One scalar step of a pair after a pass: sums[0 .. 28] = the folded linearisation (valid in phases 0 and 1), sums[29] = the folded
// error of the trial (phases 1 and 3).  The sequence of states is LevenbergMarquardtOptimizer::optimize's: the linearisation at an
// accepted pose is the one the reference computes next -- here it was computed in the same pass as the trial's error.
__device__ __forceinline__ void lm_round_step(PairState& S, const double* sums, const GicpParams& prm, bool* done) {
  *done = false;
  bool solve = true;
  if (S.phase == 0) {
    lm_after_linearize(S, sums);
  } else {
    const int was = S.phase;
    solve = lm_after_error(S, sums[29], prm, done);  // true: rejected, another damped solve of the same linearisation
    if (!solve && !*done) {  // accepted, the loop goes on
      if (was == 1) {  // the speculative linearisation at the accepted pose is the next one: its correspondences become the frozen ones
        S.buf ^= 1;
        lm_after_linearize(S, sums);
        solve = true;
      }  // (was == 3 cannot get here -- the same delta decided both; if it ever did, phase 0 linearises at T in the next pass)
    }
  }
  if (solve) {
    lm_solve(S);
    S.phase = lm_trial_is_last(S, prm) ? 3 : 1;
  }
}

// fixed-order sum of `n` blocks' partial vectors (kLinWaves wave vectors of `stride` doubles each, wave_reduce_store): the waves of
// a block first, then 8 strided sub-sums per component over the blocks, then 8 -> 1.  Components 0 .. 28 come from `part` (the
// linearisation, if `lin`), component 29 from `epart` (the trial's error, if `err`): one pass for both, the order of the
// additions of every component is that of a fold of its own.
template <bool kAgent = false>
__device__ __forceinline__ void fold_round(const double* __restrict__ part, const double* __restrict__ epart, bool lin, bool err, int n,
                                           double* s_part /*[8][32]*/, double* s_out /*[32]*/) {
  const int comp = threadIdx.x & 31, sub = threadIdx.x >> 5;  // 256 threads = 32 components x 8 sub-sums
  double a = 0;
  const bool mine = (comp < kRed && lin) || (comp == kRed && err);
  const double* src = comp < kRed ? part + comp : epart;
  const int stride = comp < kRed ? kRed : 1;
  if (mine)
    for (int k = sub; k < n; k += 8) {
      double blk = 0;  // the block's sum: its waves in order
      if constexpr (kAgent) {  // (written by other workgroups of this launch: L1-bypassing loads, all four in flight)
        double v[kLinWaves];
#pragma unroll
        for (int w = 0; w < kLinWaves; w++) v[w] = ld_ag(src + ((size_t)k * kLinWaves + w) * stride);
#pragma unroll
        for (int w = 0; w < kLinWaves; w++) blk += v[w];
      } else {
#pragma unroll
        for (int w = 0; w < kLinWaves; w++) blk += src[((size_t)k * kLinWaves + w) * stride];
      }
      a += blk;
    }
  s_part[sub * 32 + comp] = a;
  __syncthreads();
  if (threadIdx.x < 32) {
    double v = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) v += s_part[q * 32 + threadIdx.x];
    s_out[threadIdx.x] = v;
  }
  __syncthreads();
}

// after a pass: fold the pair's partial sums (fixed order) and take the scalar step; the pairs that are not done after it go on
// the list of the next round (the order is whatever the atomics give -- pairs are independent)
__global__ __launch_bounds__(256) void k_gicp_step(PairState* __restrict__ st, const double* __restrict__ partial,
                                                   const double* __restrict__ epartial, const int* __restrict__ m_counts, int nblk_max,
                                                   GicpParams prm, int* __restrict__ n_done, int* __restrict__ act_next,
                                                   int* __restrict__ n_act_next, int* __restrict__ n_act_clear) {
  __shared__ double s_part[8 * 32], s_sum[32];
  const int pair = blockIdx.x;
  if (pair == 0 && threadIdx.x == 0) *n_act_clear = 0;  // the counter the NEXT round's step fills
  const int phase = st[pair].phase;
  if (phase == 2) return;
  const int ms = m_counts[2 * pair + prm.src_slot];
  const int nblk = (ms + kLinBlock - 1) / kLinBlock;
  fold_round(partial + (size_t)pair * nblk_max * kLinWaves * kRed, epartial + (size_t)pair * nblk_max * kLinWaves, phase != 3, phase != 0, nblk,
             s_part, s_sum);
  if (threadIdx.x != 0) return;
  bool done;
  lm_round_step(st[pair], s_sum, prm, &done);
  if (done) atomicAdd(n_done, 1);
  else act_next[atomicAdd(n_act_next, 1)] = pair;
}

// the same fold + step for the workgroup of a fused pass (k_gicp_linearize<.., true>) that arrives last: the partial sums come from
// other workgroups of the running launch (L1-bypassing loads), the state is this pair's alone until the next launch
__device__ __attribute__((noinline)) void pair_step_last(PairState* S, const double* part, const double* epart, int nblk_pair, int phase,
                                                         double rot_eps, double trans_eps, int max_iterations, int pair, int* n_done,
                                                         int* act_next, int* n_act_next, double* s_part, double* s_sum) {
  fold_round<true>(part, epart, phase != 3, phase != 0, nblk_pair, s_part, s_sum);
  if (threadIdx.x != 0) return;
  GicpParams q{};
  q.rot_eps = rot_eps;
  q.trans_eps = trans_eps;
  q.max_iterations = max_iterations;
  bool done;
  lm_round_step(*S, s_sum, q, &done);
  if (done) atomicAdd(n_done, 1);
  else act_next[atomicAdd(n_act_next, 1)] = pair;
}

// ------------------------------------------------------------------------------------------------
// k_gicp_lm_coop: the Levenberg-Marquardt loop of a FEW pairs without the host and without a launch per step -- G workgroups a
// pair walk the pair's 256-point chunks (workgroup g takes chunks g, g + G, ...), write the SAME per-wave partial sums into the
// SAME slots as k_gicp_linearize, meet at a per-pair barrier, and the workgroup that arrives last folds them with
// the SAME fold_round and takes the SAME scalar step (lm_round_step) as k_gicp_step: the results are those of the
// launch-per-step form bit for bit, whatever G is and whichever workgroup folds.
//   Used (gicp_run) where four launches + a host poll per step are most of the time: a single pair (one live stream: ~5 steps of
// 75 workgroups each), and the tail of a batch whose last few pairs iterate on an otherwise idle chip.
//   What crosses workgroups -- the partial sums, the pair's state, the barrier words -- moves through 8-byte agent-scope accesses
// (ld_ag / st_ag: write-through stores, L1-bypassing loads; every storing wave drains before its workgroup's arrival is counted,
// MI355X guide: inter-workgroup communication).  tgt_index / maha6 of a point are written and re-read by the same thread (a chunk
// belongs to one workgroup for the whole launch): plain accesses.
//   The grid must be co-resident (waiting workgroups spin): gicp_run sizes it from the occupancy query, with a process-wide
// budget.  Spins are bounded: a workgroup that waits too long sets the error word and leaves, gicp_run then repeats the call
// with launches.
// ------------------------------------------------------------------------------------------------
// a wave-uniform double (read out of LDS) moved to scalar registers: what a scalar load of the launch-per-step kernels' state gives
__device__ __forceinline__ double uniform_d(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
constexpr int kStateWords = (int)(sizeof(PairState) / 8);
static_assert(sizeof(PairState) % 8 == 0 && kStateWords <= 64, "PairState is moved as 8-byte words by one wave");
constexpr unsigned kCoopSpinLimit = 4u << 20;  // polls of ~1 us: seconds, far beyond any legitimate wait

// What the two per-chunk bodies need, uniform per launch.  It sits in LDS and the bodies are separate (noinline) functions that
// read it back into scalar registers: inlined into the kernel's loops the linearisation took 200 - 240 VGPRs instead of the 94 it
// takes in k_gicp_linearize (1 - 2 workgroups a CU: no budget for a tail of several pairs); as a function of its own it keeps its
// registers, and the kernel around it is small.
struct CoopCtx {
  const double4* pts;
  const double* cov6;
  const u64* ucell;
  const unsigned* ubegin;
  const int* n_ucell;
  const unsigned* grid;
  const int* ginfo;
  int* tgt_index;
  double* maha6;
  double* partial;
  double* epartial;
  size_t buf_stride;
  int P, nblk;
  GicpParams prm;
};
typedef __attribute__((address_space(3))) const unsigned* lds_words;
// a struct of wave-uniform values out of LDS into scalar registers, word by word
template <class T>
__device__ __forceinline__ void uniform_load(T& dst, const void* lds_generic) {
  static_assert(sizeof(T) % 4 == 0, "whole words");
  const lds_words w = (lds_words)lds_generic;
  unsigned tmp[sizeof(T) / 4];
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; k++) tmp[k] = (unsigned)__builtin_amdgcn_readfirstlane((int)w[k]);
  __builtin_memcpy(&dst, tmp, sizeof(T));
}

// One chunk (256 source points, this workgroup's threads) of a pair's pass under the state in LDS: the untiled body of
// k_gicp_linearize -- the pending trial's error and / or the linearisation, by phase -- with the partial sums stored write-through
__device__ __attribute__((noinline)) void coop_pass_chunk(const void* ctx_lds, const void* state_lds, int pair_v, int chunk_v, int ms_v) {
  CoopCtx C;
  uniform_load(C, ctx_lds);
  struct Head {
    double T[12], newT[12];
  } Hd;
  uniform_load(Hd, state_lds);  // PairState: T[12] | newT[12] | ...
  const PairState* Sl = reinterpret_cast<const PairState*>(state_lds);
  const int pair = __builtin_amdgcn_readfirstlane(pair_v), chunk = __builtin_amdgcn_readfirstlane(chunk_v),
            ms = __builtin_amdgcn_readfirstlane(ms_v), phase = __builtin_amdgcn_readfirstlane(Sl->phase),
            cur = __builtin_amdgcn_readfirstlane(Sl->buf);
  const bool has_prev = __builtin_amdgcn_readfirstlane(Sl->n_lin) > 0;
  // (element by element: `const double* T12 = phase == 0 ? Hd.T : Hd.newT` -- a run-time choice between two register-resident
  //  arrays -- gave the pose of the wrong branch in this non-inlined function with ROCm 7.2's compiler; found by the parity tests)
  double T12[12];
#pragma unroll
  for (int k = 0; k < 12; k++) T12[k] = phase == 0 ? Hd.T[k] : Hd.newT[k];
  const int tid = threadIdx.x, P = C.P, wave_slot = tid >> 6;
  const int cs = 2 * pair + C.prm.src_slot, ct = 2 * pair + 1 - C.prm.src_slot;
  const unsigned* G0 = C.grid + (size_t)ct * (kGridCap + 1);
  const int* gi = C.ginfo + 8 * ct;
  const double4* tp = C.pts + (size_t)ct * P;
  const int* ti_cur = C.tgt_index + (size_t)cur * C.buf_stride;
  const double* maha_cur = C.maha6 + (size_t)cur * C.buf_stride * 6;
  const int i = chunk * kLinBlock + tid;
  double4 p = make_double4(0, 0, 0, 0), prev_q = make_double4(0, 0, 0, 0);
  int prev_j = -1;
  double tx = 0, ty = 0, tz = 0;
  int cx = 0, cy = 0, cz = 0;
  bool in_range = false;
  if (i < ms) {
    p = C.pts[(size_t)cs * P + i];
    if (has_prev) prev_j = ti_cur[(size_t)pair * P + i];
    if (prev_j >= 0) prev_q = tp[prev_j];
    lin_image(p, T12, C.prm, tx, ty, tz, cx, cy, cz, in_range);
  }
  if (phase != 0) {
    double e[1] = {0.0};
    if (i < ms && prev_j >= 0) e[0] = gicp_trial_error(prev_q, tx, ty, tz, maha_cur + ((size_t)pair * P + i) * 6);
    wave_reduce_store<1, true>(e, C.epartial + ((size_t)pair * C.nblk + chunk) * kLinWaves, wave_slot);
    if (phase == 3) return;
  }
  const int wr = phase == 0 ? cur : cur ^ 1;
  LinFactor F;
  F.on = false;
  if (i < ms) {
#ifdef GFS_LIN_UTIL
    const NnGlobal src{tp, gi, G0, C.ucell + (size_t)ct * (P + 1), C.ubegin + (size_t)ct * (P + 1), C.n_ucell[ct], nullptr};
#else
    const NnGlobal src{tp, gi, G0, C.ucell + (size_t)ct * (P + 1), C.ubegin + (size_t)ct * (P + 1), C.n_ucell[ct]};
#endif
    gicp_lin_point(src, i, p, tx, ty, tz, cx, cy, cz, in_range, T12, has_prev, prev_j, prev_q, pair, cs, ct, P, C.pts, C.cov6, C.ucell, C.ubegin,
                   C.n_ucell, G0, gi, C.prm, C.tgt_index + (size_t)wr * C.buf_stride, C.maha6 + (size_t)wr * C.buf_stride * 6, F);
  }
  double* dst = C.partial + ((size_t)pair * C.nblk + chunk) * kLinWaves * kRed;
  {
    double vb[14];
    lin_factor_batch_b(F, vb);
    wave_reduce_store_map<14, true>(vb, kLinMapB, dst, wave_slot);
  }
  {
    double va[15];
    lin_factor_batch_a(F, va);
    wave_reduce_store_map<15, true>(va, kLinMapA, dst, wave_slot);
  }
}
static_assert(offsetof(PairState, T) == 0 && offsetof(PairState, newT) == 12 * sizeof(double), "coop_pass_chunk reads the head of PairState");

__global__ __launch_bounds__(kLinBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_gicp_lm_coop(
    PairState* __restrict__ st, const double4* __restrict__ pts, const double* __restrict__ cov6, const u64* __restrict__ ucell,
    const unsigned* __restrict__ ubegin, const int* __restrict__ n_ucell, const int* __restrict__ m_counts,
    const unsigned* __restrict__ grid, const int* __restrict__ ginfo, int nchunks, int P, GicpParams prm, int* __restrict__ tgt_index,
    double* __restrict__ maha6, size_t buf_stride, double* __restrict__ partial, double* __restrict__ epartial, int nblk,
    const int* __restrict__ act, const int* __restrict__ n_act, int n_fixed, CoopSync* __restrict__ sync, int* __restrict__ n_done,
    int* __restrict__ n_err) {
  __shared__ double s_part[8 * 32], s_sum[32];
  __shared__ u64 s_state[64];
  __shared__ CoopCtx s_ctx;
  __shared__ int s_last;
  const int tid = threadIdx.x;
  // pairs of this launch: the list the previous round's k_gicp_step wrote (act), or pairs 0 .. n_fixed - 1; every workgroup
  // derives the same G from the same count
  const int n = act ? *n_act : n_fixed;
  if (n <= 0) return;
  const int G = max(1, min(nchunks, (int)gridDim.x / n));
  const int slot = blockIdx.x / G, g = blockIdx.x - slot * G;
  if (slot >= n) return;
  const int pair = act ? act[slot] : slot;
  const int ms = m_counts[2 * pair + prm.src_slot];
  const int nblk_pair = (ms + kLinBlock - 1) / kLinBlock;
  const int Gp = max(1, min(G, nblk_pair));  // workgroups that have a chunk (at least one: the scalar steps must run)
  if (g >= Gp) return;
  if (tid == 0) s_ctx = CoopCtx{pts, cov6, ucell, ubegin, n_ucell, grid, ginfo, tgt_index, maha6, partial, epartial, buf_stride, P, nblk, prm};
  u64* Sw = reinterpret_cast<u64*>(st + pair);
  PairState& S = *reinterpret_cast<PairState*>(s_state);
  CoopSync* sy = sync + pair;
  unsigned episode = 0;
  while (true) {
    if (tid < kStateWords) s_state[tid] = ld_ag(Sw + tid);
    __syncthreads();
    const int phase = __builtin_amdgcn_readfirstlane(S.phase);
    if (phase == 2) break;
    for (int chunk = g; chunk < nblk_pair; chunk += Gp) coop_pass_chunk(&s_ctx, s_state, pair, chunk, ms);
    // ---- the pair's barrier: every storing wave drains, the workgroup's arrival is counted once, the last one takes the step
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(&sy->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = old == (episode + 1) * (unsigned)Gp - 1 ? 1 : 0;
    }
    __syncthreads();
    if (s_last) {
      bool done = false;
      fold_round<true>(partial + (size_t)pair * nblk * kLinWaves * kRed, epartial + (size_t)pair * nblk * kLinWaves, phase != 3, phase != 0,
                       nblk_pair, s_part, s_sum);
      if (tid == 0) lm_round_step(S, s_sum, prm, &done);
      __syncthreads();
      if (tid < kStateWords) st_ag(Sw + tid, s_state[tid]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the state words are stored by wave 0, which also stores the flag)
      if (tid == 0) {
        if (done) atomicAdd(n_done, 1);
        st_ag(&sy->gen, episode + 1);
      }
    } else if (tid == 0) {
      unsigned spins = 0;
      while (ld_ag(&sy->gen) < episode + 1) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kCoopSpinLimit) {  // never in a sized launch; leave a mark instead of a hung GPU
          atomicAdd(n_err, 1);
          s_last = -1;
          break;
        }
      }
    }
    __syncthreads();
    if (s_last < 0) return;  // (the other workgroups of the pair run into the same limit; gicp_run repeats the call with launches)
    episode++;
  }
}

// ------------------------------------------------------------------------------------------------
// k_gicp_lm: LevenbergMarquardtOptimizer::optimize (registration/optimizer.hpp:83-147) of ONE pair per workgroup, start to
// finish: linearise -> damped solve -> error of the trial -> accept / reject, until converged or out of iterations.  The
// pairs of a batch converge after different numbers of iterations (3 ... 20): with one launch per step of the state machine
// (k_gicp_linearize / solve / error / decide, the default) the late rounds run for a handful of pairs on an otherwise idle GPU
// and every round pays four launches and a host poll; here a finished pair simply frees its compute unit.  Selected with
// GFS_GICP_LM=persistent; results are identical to rounding (the sums are folded in a different fixed order).
// 512 threads walk the source points with a stride, every thread adds its points into its own 29 sums, which are then
// folded in a fixed order (wave halving tree, 8 waves in order): deterministic and independent of the batch.
// ------------------------------------------------------------------------------------------------
constexpr int kLmBlock = 512;

__global__ __launch_bounds__(kLmBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_gicp_lm(
    PairState* __restrict__ st, const double4* __restrict__ pts, const double* __restrict__ cov6, const u64* __restrict__ ucell,
    const unsigned* __restrict__ ubegin, const int* __restrict__ n_ucell, const int* __restrict__ m_counts,
    const unsigned* __restrict__ grid, const int* __restrict__ ginfo, int npairs, int P, GicpParams prm, int* __restrict__ tgt_index,
    double* __restrict__ maha6) {
  __shared__ PairState S;
  __shared__ double s_red[(kLmBlock / 64) * 32], s_sum[32];
  const int pair = blockIdx.x, tid = threadIdx.x;
  if (pair >= npairs) return;
  if (tid == 0) S = st[pair];  // initialised by k_gicp_init
  __syncthreads();
  const int cs = 2 * pair + prm.src_slot, ct = 2 * pair + 1 - prm.src_slot;
  const int ms = m_counts[cs];
  const unsigned* G = grid + (size_t)ct * (kGridCap + 1);
  const int* gi = ginfo + 8 * ct;
  while (true) {
    const int phase = S.phase;  // uniform: read after a barrier
    if (phase == 2) break;
    if (phase == 0) {
      double acc[kRed];
#pragma unroll
      for (int k = 0; k < kRed; k++) acc[k] = 0;
      const bool has_prev = S.n_lin > 0;
      for (int i = tid; i < ms; i += kLmBlock) {
#ifdef GFS_LIN_UTIL
        const NnGlobal src{pts + (size_t)ct * P, gi, G, ucell + (size_t)ct * (P + 1), ubegin + (size_t)ct * (P + 1), n_ucell[ct], nullptr};
#else
        const NnGlobal src{pts + (size_t)ct * P, gi, G, ucell + (size_t)ct * (P + 1), ubegin + (size_t)ct * (P + 1), n_ucell[ct]};
#endif
        const double4 p = pts[(size_t)cs * P + i];
        double tx, ty, tz;
        int cx, cy, cz;
        bool in_range;
        lin_image(p, S.T, prm, tx, ty, tz, cx, cy, cz, in_range);
        const int prev_j = has_prev ? tgt_index[(size_t)pair * P + i] : -1;
        const double4 prev_q = prev_j >= 0 ? pts[(size_t)ct * P + prev_j] : make_double4(0, 0, 0, 0);
        LinFactor F;
        gicp_lin_point(src, i, p, tx, ty, tz, cx, cy, cz, in_range, S.T, has_prev, prev_j, prev_q, pair, cs, ct, P, pts, cov6, ucell, ubegin, n_ucell, G, gi,
                       prm, tgt_index, maha6, F);
        lin_factor_accumulate(F, acc);
      }
      const double r = gfs_red::block_sum_many<kRed, kLmBlock / 64>(acc, s_red);
      if (tid < kRed) s_sum[tid] = r;
      __syncthreads();
      if (tid == 0) {  // as k_gicp_solve
        double H[21], b[6], T[12], delta[6], newT[12];
#pragma unroll
        for (int k = 0; k < 21; k++) H[k] = s_sum[k];
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = s_sum[21 + k];
#pragma unroll
        for (int k = 0; k < 12; k++) T[k] = S.T[k];
        solve_and_propose(H, b, S.lambda, T, delta, newT);
#pragma unroll
        for (int k = 0; k < 21; k++) S.H[k] = H[k];
#pragma unroll
        for (int k = 0; k < 6; k++) S.b[k] = b[k];
#pragma unroll
        for (int k = 0; k < 6; k++) S.delta[k] = delta[k];
#pragma unroll
        for (int k = 0; k < 12; k++) S.newT[k] = newT[k];
        S.e = s_sum[27];
        S.inliers = (int)(s_sum[28] + 0.5);
        S.n_lin++;
        S.inner = 0;
        S.phase = 1;
      }
      __threadfence_block();  // tgt_index / maha6 of this linearisation are read by other threads below
      __syncthreads();
    } else {
      double e[1] = {0.0};
      const double* R = S.newT;
      const double* t = S.newT + 9;
      for (int i = tid; i < ms; i += kLmBlock) {  // GICPFactor::error with the frozen correspondences (as k_gicp_error)
        const int ti = tgt_index[(size_t)pair * P + i];
        if (ti >= 0) {
          const double4 p = pts[(size_t)cs * P + i];
          const double tx = R[0] * p.x + R[3] * p.y + R[6] * p.z + t[0];
          const double ty = R[1] * p.x + R[4] * p.y + R[7] * p.z + t[1];
          const double tz = R[2] * p.x + R[5] * p.y + R[8] * p.z + t[2];
          const double4 q = pts[(size_t)ct * P + ti];
          const double r0 = q.x - tx, r1 = q.y - ty, r2 = q.z - tz;
          const double* M = maha6 + ((size_t)pair * P + i) * 6;
          const double m0 = M[0] * r0 + M[1] * r1 + M[2] * r2, m1 = M[1] * r0 + M[3] * r1 + M[4] * r2,
                       m2 = M[2] * r0 + M[4] * r1 + M[5] * r2;
          e[0] += 0.5 * (r0 * m0 + r1 * m1 + r2 * m2);
        }
      }
      const double r = gfs_red::block_sum_many<1, kLmBlock / 64>(e, s_red);
      if (tid == 0) {  // as k_gicp_decide (registration/optimizer.hpp:115-141)
        const double new_e = r;
        S.n_err++;
        if (new_e <= S.e) {
          const double dr = sqrt(S.delta[0] * S.delta[0] + S.delta[1] * S.delta[1] + S.delta[2] * S.delta[2]);
          const double dt = sqrt(S.delta[3] * S.delta[3] + S.delta[4] * S.delta[4] + S.delta[5] * S.delta[5]);
          S.converged = (dr <= prm.rot_eps && dt <= prm.trans_eps) ? 1 : 0;
          for (int k = 0; k < 12; k++) S.T[k] = S.newT[k];
          S.lambda /= 10.0;
          S.iterations = S.outer;
          S.outer++;
          S.phase = (S.converged || S.outer >= prm.max_iterations) ? 2 : 0;
        } else {
          S.lambda *= 10.0;
          S.inner++;
          if (S.inner >= 10) {  // max_inner_iterations: !success -> break
            S.iterations = S.outer;
            S.phase = 2;
          } else {
            double H[21], b[6], T[12], delta[6], newT[12];
#pragma unroll
            for (int k = 0; k < 21; k++) H[k] = S.H[k];
#pragma unroll
            for (int k = 0; k < 6; k++) b[k] = S.b[k];
#pragma unroll
            for (int k = 0; k < 12; k++) T[k] = S.T[k];
            solve_and_propose(H, b, S.lambda, T, delta, newT);
#pragma unroll
            for (int k = 0; k < 6; k++) S.delta[k] = delta[k];
#pragma unroll
            for (int k = 0; k < 12; k++) S.newT[k] = newT[k];
          }
        }
      }
      __syncthreads();
    }
  }
  if (tid == 0) st[pair] = S;
}

__global__ void k_gicp_init(PairState* __restrict__ st, const double* __restrict__ init_T, int B, int max_iterations,
                            int* __restrict__ n_done, CoopSync* __restrict__ sync) {
  const int pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= B) return;
  sync[pair].arrive = 0;
  sync[pair].gen = 0;
  PairState& S = st[pair];
  const double* T = init_T + 16 * (size_t)pair;
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) S.T[r + 3 * c] = T[r + 4 * c];
  for (int r = 0; r < 3; r++) S.T[9 + r] = T[12 + r];
  for (int k = 0; k < 12; k++) S.newT[k] = S.T[k];
  for (int k = 0; k < 6; k++) S.delta[k] = 0;
  for (int k = 0; k < 21; k++) S.H[k] = 0;
  for (int k = 0; k < 6; k++) S.b[k] = 0;
  S.e = 0;
  S.lambda = 1e-3;  // init_lambda
  S.outer = S.inner = 0;
  S.converged = 0;
  S.iterations = 0;
  S.inliers = 0;
  S.n_lin = S.n_err = 0;
  S.buf = S.pad_ = 0;
  S.phase = max_iterations > 0 ? 0 : 2;
  if (max_iterations <= 0) atomicAdd(n_done, 1);
}

// the call's results into pinned host memory (the device's view of it): the pairs' final states, the down-sampled sizes, the counters'
// last look -- one launch in the place of three device-to-host copies (each a blit kernel of its own on the stream)
__global__ __launch_bounds__(256) void k_gicp_publish(const PairState* __restrict__ st, const int* __restrict__ m_counts,
                                                      const int* __restrict__ n_done, int B, u64* __restrict__ host_state,
                                                      int* __restrict__ host_m, int* __restrict__ host_last) {
  const int t = blockIdx.x * 256 + threadIdx.x, nw = B * kStateWords;
  const u64* src = reinterpret_cast<const u64*>(st);
  for (int k = t; k < nw; k += gridDim.x * 256) host_state[k] = src[k];
  for (int k = t; k < 2 * B; k += gridDim.x * 256) host_m[k] = m_counts[k];
  if (t < 3) host_last[t] = n_done[t];
}

}  // namespace

struct gfs_gicp {
  int device, P, Bmax, nblk;
  hipStream_t stream;
  std::recursive_mutex mu;  // the host-pointer entries hold it across staging upload + run
  gfs::DevBuf<float4> d_in_t, d_in_s;  // staging for the host-pointer entry
  gfs::DevBuf<unsigned> d_hard;  // per cloud: indices of the points k_knn_cov deferred to k_knn_cov_far
  gfs::DevBuf<double> d_hard_d;  // ... and the squared distance bounding their k nearest (same slots)
  gfs::DevBuf<int> d_far2;  // per cloud: number of points deferred a second time (stored from the back of d_hard)
  gfs::DevBuf<int> d_nt, d_ns, d_counts, d_which, d_m, d_which2, d_nucell, d_tgt_index, d_ndone, d_bbox, d_ginfo, d_kinfo1, d_kinfo2;
  gfs::DevBuf<u64> d_keys0, d_keys1, d_ck0, d_ck1, d_ucell;
  gfs::DevBuf<unsigned> d_val0, d_val1, d_ci0, d_ci1, d_ubegin, d_grid;
  gfs::DevBuf<unsigned> d_leaf;  // per cloud: (begin, end) of the < 1024-element ranges of the voxel sort (voxel_qsort.hpp)
  gfs::DevBuf<int> d_nleaf, d_nheap;
  gfs::DevBuf<unsigned> d_heap;  // per cloud: (begin, end) of the ranges that hit std::sort's depth limit (heap-sort fallback)
  int heap_cap = 0;
  // one launch per step of the LM state machine (default).  GFS_GICP_LM=persistent: the whole loop of a pair in one workgroup
  // (k_gicp_lm) — no launches / host polls inside the loop, but only one workgroup of parallelism per pair: measured 4x slower at
  // 128 pairs per batch (25 vs 6 ms per 512 pairs), it pays only for batches of many thousand small pairs.
  gfs::DevBuf<int> d_active, d_nactive;  // [3][B] pairs still iterating (written by k_gicp_step for the next round), [3] their number (round r: r % 3)
  bool lm_rounds = true;
  // k_gicp_lm_coop (the loop of a few pairs in one launch): GFS_GICP_COOP=0 switches it off; coop_cap = workgroups of it the device
  // holds at once (occupancy query, one fewer per CU than the API says: the hardware may admit fewer); a launch takes its
  // workgroups from a process-wide budget of half of that (CoopBudget) and gives them back when the call has been waited for
  bool coop = true, coop_failed = false;
  bool fuse_step = false;  // GFS_GICP_FUSE_STEP=1 (read at creation): see the round loop of gicp_run
  int coop_cap = 0, coop_reserved = 0, coop_launches = 0, coop_last_wgs = 0;
  // GFS_GICP_COOP_TAIL = f > 0: the TAIL of a larger batch also goes to the kernel once (pairs left) x (chunks a pair) <= f x budget.
  // Off by default -- measured (round 6, 64-pair block in 2 lanes): 2.07 ms a step without, 2.09 / 2.11 / 2.17 / 2.29 ms with f = 1 / 2 /
  // 4 / 8: on a chip that other lanes keep busy a barrier episode (drain, arrival, fold by the last workgroup, publish, poll) costs
  // what the two launch boundaries it replaces cost, and the resident workgroups are in the other lanes' way.
  double coop_tail = 0.0;
  gfs::DevBuf<CoopSync> d_sync;
  bool tile_stats_on = false;  // GFS_GICP_TILE_STATS=1
  gfs::DevBuf<unsigned> d_tile_stats;  // [8] outcome counters of k_gicp_linearize's tile staging (gfs_gicp_tile_stats)
  bool vqs_lds = true;  // GFS_GICP_VQS_LDS=0: without k_voxel_qsort_top_lds
  bool sort_all_kernels = false;  // set for the second run of a call whose first run found a cloud the LDS sort kernel could not take
  int sort_all_hold = 0;  // ... and the next calls launch every sort kernel at once instead of finding out again (a stream of wide scenes
                          // would pay every call twice); counted down, then the optimistic launch is tried again
  bool stable_voxel_order = false;  // GFS_GICP_VOXEL_ORDER=stable: the round-1 stable radix order instead of the reference's
  gfs::DevBuf<double4> d_tmp, d_pts;
  gfs::DevBuf<double> d_cov6, d_maha6, d_partial, d_epartial;
  gfs::DevBuf<PairState> d_state;
  gfs::PinBuf<PairState> h_state;
  gfs::PinBuf<int> h_ndone, h_m;
  static constexpr int kAheadMax = 8;
  hipEvent_t ev_round[kAheadMax + 1] = {};  // completion of the LM rounds in flight
  int ahead = 1;  // LM rounds queued beyond the one whose done counter the host waits for (GFS_GICP_AHEAD, measured in round 6: see gicp_run)
  gfs::PinBuf<double> h_initT;
  const double* hd_initT = nullptr;  // h_initT as the device sees it (k_gicp_init reads the initial poses over the bus)
  PairState* hd_state = nullptr;     // h_state, h_m, the call's last look at the counters (h_ndone's tail) as the device sees them:
  int *hd_m = nullptr, *hd_last = nullptr;  // k_gicp_publish writes the call's results there (one launch instead of three copies)
  int last_B = 0;
  // streaming (gfs_gicp_align_next*): slot parity that holds the preprocessed SOURCE clouds of the last call, and what they
  // were preprocessed with
  int last_src_slot = 1, last_stride = 0;
  double last_leaf = 0, last_cell = 0;
  int last_k = 0;
  gfs::DevBuf<int> d_zero;  // [Bmax] zeros: the point counts of the slot that is not re-read in a streaming call
};

// Workgroups of k_gicp_lm_coop in flight in this process, per device: the kernel's workgroups wait for each other, so all of them
// -- of every handle's launch -- must be resident together.  A launch reserves before it is enqueued and releases after its call's
// final synchronisation; a launch that cannot get enough falls back to the launch-per-step rounds.
struct CoopBudget {
  static std::atomic<int>& in_use(int device) {
    static std::atomic<int> a[64];
    return a[device & 63];
  }
  static int reserve(int device, int want, int least, int total) {
    std::atomic<int>& u = in_use(device);
    int cur = u.load();
    while (true) {
      const int get = std::min(want, total - cur);
      if (get < least) return 0;
      if (u.compare_exchange_weak(cur, cur + get)) return get;
    }
  }
  static void release(int device, int n) {
    if (n > 0) in_use(device).fetch_sub(n);
  }
};

// The n >= 1024 levels of the voxel sort: the LDS-resident kernel for clouds of at most kVqsLdsE * 1024 points whose keys compact to
// 31 bits, the register-cached kernel for larger ones of that key width, the general kernel for the rest (each flags what it leaves).
// n_left != nullptr (only when every cloud fits the LDS kernel by size): the other two kernels are NOT launched and the LDS kernel
// counts the clouds it left in *n_left -- the caller has to look at it and come back with n_left = nullptr if it is not zero.  (A
// launch of the general kernel that finds nothing to do is not free: its 1024-thread workgroups need a CU to themselves, and with
// other lanes' kernels in flight it waits ~0.3 ms for one.)
// kVqsLdsE: keys, point indices and the rendezvous array in LDS (8 bytes an element); kVqsLdsGvE: the keys only (4 bytes), the other
// two in a global scratch block (d_keys1 is free while this kernel sorts) -- what a handle sized for 720p clouds gets.
constexpr int kVqsLdsE = 19, kVqsLdsGvE = 37;
static hipError_t voxel_qsort_top(gfs_gicp* h, int C2, hipStream_t s, int only, int* n_left) {
  const int P = h->P;
  int flagged_only = 0;
  const bool lds_covers_sizes = h->vqs_lds && P <= 1024 * kVqsLdsGvE;
  if (!lds_covers_sizes) n_left = nullptr;
  if (h->vqs_lds) {
    const int _pid = ::gfs::profile_on() ? ::gfs::profile_begin("k_voxel_qsort_top_lds", s) : -1;
    if (P <= 1024 * kVqsLdsE || P > 1024 * kVqsLdsGvE)
      hipLaunchKernelGGL((vqs::k_voxel_qsort_top_lds<kVqsLdsE, false>), dim3(C2), dim3(1024), kVqsLdsE * 1024 * 8, s, h->d_keys0.p,
                         h->d_val0.p, h->d_keys1.p, h->d_counts.p, P, h->d_which.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, only, n_left);
    else
      hipLaunchKernelGGL((vqs::k_voxel_qsort_top_lds<kVqsLdsGvE, true>), dim3(C2), dim3(1024), kVqsLdsGvE * 1024 * 4, s, h->d_keys0.p,
                         h->d_val0.p, h->d_keys1.p, h->d_counts.p, P, h->d_which.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, only, n_left);
    if (_pid >= 0) ::gfs::profile_end(_pid, s);
    flagged_only = 1;
    if (n_left) return hipGetLastError();
  }
#define VQS_TOP_REG(E)                                                                                                         \
  do {                                                                                                                         \
    const int _pid = ::gfs::profile_on() ? ::gfs::profile_begin("k_voxel_qsort_top_reg", s) : -1;                              \
    hipLaunchKernelGGL(vqs::k_voxel_qsort_top_reg<E>, dim3(C2), dim3(1024), 0, s, h->d_keys0.p, h->d_val0.p, h->d_keys1.p,     \
                       h->d_val1.p, h->d_counts.p, P, h->d_which.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, only,            \
                       flagged_only);                                                                                          \
    if (_pid >= 0) ::gfs::profile_end(_pid, s);                                                                                \
  } while (0)
  // (when P fits an LDS kernel only key width can leave a cloud over: the general kernel)
  if (P <= 1024 * 20) {
    if (!lds_covers_sizes) VQS_TOP_REG(20);
    flagged_only = 1;
  } else if (P <= 1024 * 40) {
    if (!lds_covers_sizes) VQS_TOP_REG(40);
    flagged_only = 1;
  }
#undef VQS_TOP_REG
  {
    const int _pid = ::gfs::profile_on() ? ::gfs::profile_begin("k_voxel_qsort_top", s) : -1;
    hipLaunchKernelGGL(vqs::k_voxel_qsort_top, dim3(C2), dim3(1024), 0, s, h->d_keys0.p, h->d_val0.p, h->d_keys1.p, h->d_val1.p,
                       h->d_counts.p, P, h->d_which.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, only, flagged_only);
    if (_pid >= 0) ::gfs::profile_end(_pid, s);
  }
  return hipGetLastError();
}

// std::sort of the < 1024-element ranges (both key widths), then the heap-sort fallbacks they deferred
static int voxel_qsort_leaves(gfs_gicp* h, int C2, int leaf_parts, hipStream_t s, int only, bool narrow_only,
                              const float4* in_even = nullptr, const float4* in_odd = nullptr, int stride_pts = 0) {
  const int P = h->P;
  GFS_LAUNCH("k_voxel_qsort_leaf", vqs::k_voxel_qsort_leaf<unsigned>, dim3(leaf_parts, C2), dim3(256), 0, s, h->d_keys0.p,
             h->d_val0.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, P, only, h->d_heap.p, h->d_nheap.p, h->heap_cap);
  // (narrow_only: the caller knows, or will find out and come back, that no cloud has keys wider than 31 bits)
  if (!narrow_only)
    GFS_LAUNCH("k_voxel_qsort_leaf64", vqs::k_voxel_qsort_leaf<u64>, dim3(leaf_parts, C2), dim3(256), 0, s, h->d_keys0.p,
               h->d_val0.p, h->d_kinfo1.p, h->d_leaf.p, h->d_nleaf.p, P, only, h->d_heap.p, h->d_nheap.p, h->heap_cap);
  const int heap_parts = std::max(1, std::min(16, P / 4096));
  GFS_LAUNCH("k_voxel_qsort_heap", vqs::k_voxel_qsort_heap<unsigned>, dim3(heap_parts, C2), dim3(256), 0, s, h->d_keys0.p,
             h->d_val0.p, h->d_kinfo1.p, h->d_heap.p, h->d_nheap.p, h->heap_cap, P, only, in_even, in_odd, stride_pts);
  if (!narrow_only)
    GFS_LAUNCH("k_voxel_qsort_heap64", vqs::k_voxel_qsort_heap<u64>, dim3(heap_parts, C2), dim3(256), 0, s, h->d_keys0.p,
               h->d_val0.p, h->d_kinfo1.p, h->d_heap.p, h->d_nheap.p, h->heap_cap, P, only, in_even, in_odd, stride_pts);
  return GFS_OK;
}

extern "C" {

void gfs_gicp_default_config(gfs_gicp_config* c) {
  if (!c) return;
  c->num_threads = 4;                    // src/RegistrationGICP.cc:10
  c->downsampling_resolution = 0.02;     // :11
  c->max_correspondence_distance = 0.1;  // :12-13
  c->rotation_eps = 0.1 * 3.14159265358979323846 / 180.0;
  c->translation_eps = 1e-3;
  c->max_iterations = 20;
  c->num_neighbors = 10;  // registration_helper.cpp:60-61
}

int gfs_gicp_create(int device, int max_points, int max_batch, gfs_gicp** out) {
  GFS_REQUIRE(out && max_points > 0 && max_batch > 0, GFS_ERR_INVALID_ARG, "gfs_gicp_create: invalid argument");
  GFS_REQUIRE(max_points <= (1 << 20), GFS_ERR_CAPACITY, "gfs_gicp_create: at most 2^20 points per cloud");
  if (!gfs::device_ok(device)) return GFS_ERR_NO_DEVICE;
  GFS_HIP(hipSetDevice(device));
  std::unique_ptr<gfs_gicp> h(new gfs_gicp);
  h->device = device;
  h->P = (int)gfs::align_up((size_t)max_points, 1024);
  h->Bmax = max_batch;
  h->nblk = gfs::div_up(h->P, kLinBlock);
  GFS_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  if (const char* e = getenv("GFS_GICP_VOXEL_ORDER")) h->stable_voxel_order = strcmp(e, "stable") == 0;
  if (const char* e = getenv("GFS_GICP_VQS_LDS")) h->vqs_lds = atoi(e) != 0;
  GFS_HIP(hipFuncSetAttribute((const void*)vqs::k_voxel_qsort_top_lds<kVqsLdsE, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kVqsLdsE * 1024 * 8));
  GFS_HIP(hipFuncSetAttribute((const void*)vqs::k_voxel_qsort_top_lds<kVqsLdsGvE, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kVqsLdsGvE * 1024 * 4));
  GFS_HIP(hipFuncSetAttribute((const void*)k_cell_sort_lds, hipFuncAttributeMaxDynamicSharedMemorySize, kCsLdsBytes));
  if (const char* e = getenv("GFS_GICP_LM")) h->lm_rounds = strcmp(e, "persistent") != 0;
  if (const char* e = getenv("GFS_GICP_TILE_STATS")) h->tile_stats_on = atoi(e) != 0;
  if (const char* e = getenv("GFS_GICP_COOP")) h->coop = atoi(e) != 0;
  if (const char* e = getenv("GFS_GICP_FUSE_STEP")) h->fuse_step = atoi(e) != 0;
  if (const char* e = getenv("GFS_GICP_COOP_TAIL")) h->coop_tail = atof(e);
  {
    int per_cu = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_gicp_lm_coop, kLinBlock, 0) == hipSuccess &&
        hipGetDeviceProperties(&prop, device) == hipSuccess)
      cus = prop.multiProcessorCount;
    h->coop_cap = std::max(0, std::min(per_cu, 8) - 1) * cus;
    if (const char* e = getenv("GFS_GICP_COOP_WGS")) h->coop_cap = std::min(h->coop_cap, 2 * std::max(0, atoi(e)));  // (budget = cap / 2)
    (void)hipGetLastError();
  }
  const size_t P = h->P, B = max_batch, C2 = 2 * B;
  int rc = 0;
#define A(x) if (!rc) rc = (x)
  A(h->d_in_t.alloc(P));
  A(h->d_in_s.alloc(P));
  A(h->d_nt.alloc(B));
  A(h->d_ns.alloc(B));
  A(h->d_counts.alloc(C2));
  A(h->d_which.alloc(C2));
  A(h->d_which2.alloc(C2));
  A(h->d_m.alloc(C2));
  A(h->d_nucell.alloc(C2));
  A(h->d_bbox.alloc(C2 * 6));
  A(h->d_ginfo.alloc(C2 * 8));
  A(h->d_hard.alloc((size_t)C2 * P));
  A(h->d_hard_d.alloc((size_t)C2 * P));
  A(h->d_far2.alloc(C2));
  A(h->d_kinfo1.alloc(C2 * 8));
  A(h->d_kinfo2.alloc(C2 * 8));
  A(h->d_grid.alloc(C2 * ((size_t)kGridCap + 1)));
  A(h->d_ndone.alloc(4));  // [0] pairs done, [1] clouds the LDS sort kernel left to the kernels that were not launched, [2] k_gicp_lm_coop's timeouts
  A(h->d_sync.alloc(B));
  A(h->d_active.alloc(3 * (size_t)B));
  A(h->d_nactive.alloc(4));
  A(h->d_tile_stats.alloc(8));
  A(h->d_keys0.alloc(C2 * P));
  A(h->d_keys1.alloc(C2 * P));
  A(h->d_val0.alloc(C2 * P));
  A(h->d_val1.alloc(C2 * P));
  A(h->d_leaf.alloc(C2 * P));
  A(h->d_nleaf.alloc(C2));
  A(h->d_nheap.alloc(C2));
  h->heap_cap = (int)(P / 8 + 2);  // ranges are disjoint and longer than 16 elements
  A(h->d_heap.alloc(C2 * (size_t)h->heap_cap));
  A(h->d_ck0.alloc(C2 * P));
  A(h->d_ck1.alloc(C2 * P));
  A(h->d_ci0.alloc(C2 * P));
  A(h->d_ci1.alloc(C2 * P));
  A(h->d_ucell.alloc(C2 * (P + 1)));
  A(h->d_ubegin.alloc(C2 * (P + 1)));
  A(h->d_tmp.alloc(C2 * P));
  A(h->d_pts.alloc(C2 * P));
  A(h->d_cov6.alloc(C2 * P * 6));
  A(h->d_maha6.alloc(2 * B * P * 6));  // two buffers a pair: the frozen correspondences and those of the speculative linearisation
  A(h->d_tgt_index.alloc(2 * B * P));
  A(h->d_partial.alloc(B * h->nblk * kLinWaves * kRed));
  A(h->d_epartial.alloc(B * h->nblk * kLinWaves));
  A(h->d_state.alloc(B));
  A(h->h_state.alloc(B));
  A(h->h_ndone.alloc(2 * (gfs_gicp::kAheadMax + 1) + 4));  // [2 slot ..] the rounds in flight, the last three: the call's last look
  for (int k = 0; k <= gfs_gicp::kAheadMax; k++) GFS_HIP(hipEventCreateWithFlags(&h->ev_round[k], hipEventDisableTiming));
  if (const char* e = getenv("GFS_GICP_AHEAD")) h->ahead = std::max(1, std::min(gfs_gicp::kAheadMax, atoi(e)));
  GFS_HIP(hipHostGetDevicePointer((void**)&h->hd_last, h->h_ndone.p + 2 * (gfs_gicp::kAheadMax + 1), 0));
  A(h->h_m.alloc(C2));
  A(h->h_initT.alloc(B * 16));
  GFS_HIP(hipHostGetDevicePointer((void**)&h->hd_initT, h->h_initT.p, 0));
  GFS_HIP(hipHostGetDevicePointer((void**)&h->hd_state, h->h_state.p, 0));
  GFS_HIP(hipHostGetDevicePointer((void**)&h->hd_m, h->h_m.p, 0));
  A(h->d_zero.alloc(B));
#undef A
  if (!rc) (void)hipMemset(h->d_tile_stats.p, 0, 8 * sizeof(unsigned));
  if (!rc) GFS_HIP(hipMemset(h->d_zero.p, 0, B * sizeof(int)));
  if (rc) return rc;
  *out = h.release();
  return GFS_OK;
}

void gfs_gicp_destroy(gfs_gicp* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipStreamDestroy(h->stream);
  for (int k = 0; k <= gfs_gicp::kAheadMax; k++)
    if (h->ev_round[k]) (void)hipEventDestroy(h->ev_round[k]);
  delete h;
}

// streaming = false: preprocess both clouds of every pair (slots 2b = target, 2b + 1 = source).
// streaming = true : dev_target / dev_nt are ignored; the slot that holds the previous call's preprocessed source clouds
//                    becomes the target as it is, the new source clouds go through preprocessing into the other slot.
static int gicp_run(gfs_gicp* h, const void* dev_target, const void* dev_nt, const void* dev_source, const void* dev_ns, int B,
                    int stride_pts, const double* init_T, const gfs_gicp_config* cfg, gfs_gicp_result* out, void* stream,
                    bool streaming) {
  GFS_REQUIRE(h && (streaming || (dev_target && dev_nt)) && dev_source && dev_ns && B > 0 && stride_pts > 0 && cfg && out,
              GFS_ERR_INVALID_ARG, "gfs_gicp_align: invalid argument");
  GFS_REQUIRE(B <= h->Bmax, GFS_ERR_CAPACITY, "batch %d exceeds handle max_batch %d", B, h->Bmax);
  GFS_REQUIRE(cfg->downsampling_resolution > 0 && cfg->max_correspondence_distance > 0 && cfg->num_neighbors >= 1 &&
                  cfg->num_neighbors <= 10,
              GFS_ERR_UNSUPPORTED, "gfs_gicp: need resolution > 0, max_corr > 0, 1 <= num_neighbors <= 10");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = stream ? (hipStream_t)stream : h->stream;
  const int P = h->P, C2 = 2 * B;
  // (the round loop below looks at the count the LDS sort kernel leaves; the persistent LM kernel has no such look)
  const bool optimistic_sort = h->lm_rounds && !h->sort_all_kernels && h->sort_all_hold == 0;
  if (h->sort_all_hold > 0 && !h->sort_all_kernels) h->sort_all_hold--;
  GicpParams prm;
  prm.inv_leaf = 1.0 / cfg->downsampling_resolution;
  // cell edge = max correspondence distance: one ring of cells certifies every 1-NN probe and (for voxel-sized
  // spacing) every 10-NN probe.  Measured on MI355X (profiles/r01a vs a 0.04 m cell): the per-row lookups, not
  // the candidate scans, dominate, so fewer / larger cells win until the probes are LDS-tiled.
  prm.cell = cfg->max_correspondence_distance;
  if (const char* e = getenv("GFS_GICP_CELL")) prm.cell = atof(e);  // experiment knob (results are exact for any cell size)
  prm.inv_cell = 1.0 / prm.cell;
  prm.nn_rings = (int)std::ceil(cfg->max_correspondence_distance / prm.cell - 1e-12);
  if (prm.nn_rings < 1) prm.nn_rings = 1;
  prm.lin_tile = 0;  // measured (profiles/README.md, round 3): with the x-ordered sweep the search in HBM at 4 waves per SIMD beats the staged tile at 3
  prm.tile_stats = h->tile_stats_on ? h->d_tile_stats.p : nullptr;  // one atomic per workgroup on one address: only on request
  if (const char* e = getenv("GFS_GICP_LIN_TILE")) prm.lin_tile = atoi(e) != 0;  // 0: every workgroup searches the cloud in HBM
  prm.max_dist_sq = cfg->max_correspondence_distance * cfg->max_correspondence_distance;
  prm.rot_eps = cfg->rotation_eps;
  prm.trans_eps = cfg->translation_eps;
  prm.max_iterations = cfg->max_iterations;
  prm.k_neighbors = cfg->num_neighbors;
  prm.src_slot = 1;
  prm.only = -1;
  prm.knn_exact = getenv("GFS_GICP_KNN_EXACT") && atoi(getenv("GFS_GICP_KNN_EXACT")) != 0;
  if (streaming) {
    // (the stride is that of THIS call's source buffer: the target slot is not read from any input buffer again)
    GFS_REQUIRE(h->last_B == B && h->last_leaf == cfg->downsampling_resolution && h->last_cell == prm.cell &&
                    h->last_k == cfg->num_neighbors,
                GFS_ERR_INVALID_ARG,
                "gfs_gicp_align_next: needs a previous call on this handle with the same batch size and preprocessing "
                "parameters (its source clouds are this call's targets)");
    prm.src_slot = 1 - h->last_src_slot;
    prm.only = prm.src_slot;
  }
  // k_voxel_keys / k_voxel_reduce read the input of even slots through their `tgt` and of odd slots through their `src` argument
  const float4* in_even = (const float4*)(streaming ? (prm.src_slot == 0 ? dev_source : nullptr) : dev_target);
  const float4* in_odd = (const float4*)(streaming ? (prm.src_slot == 1 ? dev_source : nullptr) : dev_source);
  const int* n_even = (const int*)(streaming ? (prm.src_slot == 0 ? dev_ns : (const void*)h->d_zero.p) : dev_nt);
  const int* n_odd = (const int*)(streaming ? (prm.src_slot == 1 ? dev_ns : (const void*)h->d_zero.p) : dev_ns);
  for (int b = 0; b < B; b++)
    for (int k = 0; k < 16; k++) h->h_initT.p[16 * b + k] = init_T ? init_T[16 * b + k] : (k % 5 == 0 ? 1.0 : 0.0);
  // (no upload: k_gicp_init reads the poses out of the pinned buffer -- 128 bytes a pair -- and k_voxel_keys zeroes the counters)
  const int npts = std::min(stride_pts, P);
  // The cell sort key gives the cell z 19 bits, y 20 and x (in 1 / kFine of a cell) 25: every coordinate of a voxel mean fits while
  // |coordinate| < 2^18 cells.  The voxel fields admit 10^6 leaves, which is less whenever cell >= 4 leaves (the reference's
  // 0.1 / 0.02 m); for a smaller ratio the admitted range shrinks to what the cell key can hold (>= 13 km at 0.05 m).
  const double max_vox = std::min(1.0e6, (double)((1 << 18) - 2) * prm.cell * prm.inv_leaf);
  // ---- preprocess_points x 2B (registration_helper.cpp:22-34)
  GFS_LAUNCH("k_voxel_keys", k_voxel_keys, dim3(gfs::div_up(npts, 256), C2), dim3(256), 0, s, in_even, in_odd, n_even, n_odd,
             stride_pts, P, prm.inv_leaf, max_vox, h->d_keys0.p, h->d_val0.p, h->d_counts.p, prm.only, h->d_ndone.p, h->d_nheap.p,
             h->d_nactive.p);
  if (h->stable_voxel_order) {
    GFS_LAUNCH("k_radix_sort", k_radix_sort, dim3(C2), dim3(1024), 0, s, h->d_keys0.p, h->d_keys1.p, h->d_val0.p, h->d_val1.p,
               h->d_counts.p, P, h->d_which.p, h->d_kinfo1.p, prm.only, kCoordBits, 2 * kCoordBits);
  } else {
    // the reference's (unstable) quick_sort_omp permutation, reproduced exactly: util/sort_omp.hpp:58-85
    const int leaf_parts = std::max(1, std::min(64, P / 1024));  // x 4 waves: a wave per leaf for nearly every cloud (the longest leaf sets the time)
    int rc_leaf = 0;
    GFS_HIP(voxel_qsort_top(h, C2, s, prm.only, optimistic_sort ? h->d_ndone.p + 1 : nullptr));
    // (GFS_GICP_VOXEL_TIES=exact: the reference's permutation even where it cannot change a voxel mean)
    static const bool exact_ties = getenv("GFS_GICP_VOXEL_TIES") && strcmp(getenv("GFS_GICP_VOXEL_TIES"), "exact") == 0;
    const bool narrow_only = optimistic_sort && h->vqs_lds && P <= 1024 * kVqsLdsGvE;  // (what voxel_qsort_top counts in d_ndone[1])
    rc_leaf = exact_ties ? voxel_qsort_leaves(h, C2, leaf_parts, s, prm.only, narrow_only)
                         : voxel_qsort_leaves(h, C2, leaf_parts, s, prm.only, narrow_only, in_even, in_odd, stride_pts);
    if (rc_leaf) return rc_leaf;
  }
  // small batches: several workgroups a cloud for the one-workgroup-per-cloud kernels (256 CUs; the clouds of a streaming call are C2 / 2)
  const int tiles = P / 1024, wg_parts = std::max(1, std::min(std::min(8, tiles), 384 / std::max(1, prm.only >= 0 ? C2 / 2 : C2)));
  const int tiles_per_part = gfs::div_up(tiles, wg_parts);
  GFS_LAUNCH("k_voxel_reduce", k_voxel_reduce, dim3(C2, wg_parts), dim3(1024), 0, s, in_even, in_odd, stride_pts, h->d_keys0.p,
             h->d_keys1.p, h->d_val0.p, h->d_val1.p, h->d_which.p, h->d_counts.p, P, prm.inv_cell, h->d_tmp.p, h->d_ck0.p, h->d_ci0.p,
             h->d_m.p, prm.only, tiles_per_part, h->d_bbox.p);
  // the cell sort: resident in LDS when the handle's clouds fit (it counts the clouds it cannot take in d_ndone[1], which the LM
  // loop's first poll looks at -- the call is then run again with k_radix_sort, like the voxel sort's optimistic launch)
  if (optimistic_sort && h->vqs_lds && P <= 1024 * kCsE) {
    GFS_LAUNCH("k_cell_sort_lds", k_cell_sort_lds, dim3(C2), dim3(1024), kCsLdsBytes, s, h->d_ck0.p, h->d_ci0.p, h->d_m.p, P,
               h->d_which2.p, h->d_kinfo2.p, prm.only, h->d_ndone.p + 1);
  } else {
    GFS_LAUNCH("k_radix_sort", k_radix_sort, dim3(C2), dim3(1024), 0, s, h->d_ck0.p, h->d_ck1.p, h->d_ci0.p, h->d_ci1.p,
               h->d_m.p, P, h->d_which2.p, h->d_kinfo2.p, prm.only, kCkSy, kCkSz);
  }
  GFS_LAUNCH("k_cell_build", k_cell_build, dim3(C2, wg_parts), dim3(1024), 0, s, h->d_tmp.p, h->d_ck0.p, h->d_ck1.p, h->d_ci0.p,
             h->d_ci1.p, h->d_which2.p, h->d_m.p, P, h->d_pts.p, h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_bbox.p, h->d_kinfo2.p,
             prm.only);
  GFS_LAUNCH("k_grid_fill", k_grid_fill, dim3(kGridFillParts, C2), dim3(256), 0, s, h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p,
             h->d_bbox.p, P, h->d_grid.p, h->d_ginfo.p, h->d_far2.p, prm.only);
  const int knn_chunks = gfs::div_up(npts, 128);
  GFS_LAUNCH("k_knn_cov", k_knn_cov, dim3(xcd_grid(knn_chunks, B, 2)), dim3(128), 0, s, h->d_pts.p, h->d_ucell.p,
             h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_bbox.p, h->d_grid.p, h->d_ginfo.p, h->d_ginfo.p, h->d_hard.p,
             h->d_hard_d.p, knn_chunks, B, P, prm, h->d_cov6.p);
  const int far_chunks = std::max(1, std::min(32, knn_chunks));  // grid-stride over the deferred lists
  GFS_LAUNCH("k_knn_cov_far", (k_knn_cov_far<16, 2, true>), dim3(xcd_grid(far_chunks, B, 2)), dim3(256), 0, s, h->d_pts.p,
             h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_bbox.p, h->d_grid.p, h->d_ginfo.p, h->d_hard.p,
             h->d_hard_d.p, h->d_far2.p, far_chunks, B, P, prm, h->d_cov6.p);
  const int far2_chunks = std::max(1, std::min(32, knn_chunks));  // a handful of queries per cloud (a few dozen in the worst scenes), a workgroup each
  static const bool far2_groups = getenv("GFS_GICP_FAR2") && strcmp(getenv("GFS_GICP_FAR2"), "groups") == 0;  // the lane-group form
  if (far2_groups)
    GFS_LAUNCH("k_knn_cov_far_groups", (k_knn_cov_far<64, 4, false>), dim3(xcd_grid(4, B, 2)), dim3(256), 0, s, h->d_pts.p,
               h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_bbox.p, h->d_grid.p, h->d_ginfo.p, h->d_hard.p,
               h->d_hard_d.p, h->d_far2.p, 4, B, P, prm, h->d_cov6.p);
  else
    GFS_LAUNCH("k_knn_cov_far_wg", k_knn_cov_far_wg, dim3(xcd_grid(far2_chunks, B, 2)), dim3(256), 0, s, h->d_pts.p, h->d_ucell.p,
               h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_bbox.p, h->d_grid.p, h->d_ginfo.p, h->d_hard.p, h->d_hard_d.p,
               h->d_far2.p, far2_chunks, B, P, prm, h->d_cov6.p);
  // ---- LevenbergMarquardtOptimizer::optimize: device state machine, host polls the done counter
  GFS_LAUNCH("k_gicp_init", k_gicp_init, dim3(gfs::div_up(B, 64)), dim3(64), 0, s, h->d_state.p, h->hd_initT, B,
             prm.max_iterations, h->d_ndone.p, h->d_sync.p);
  if (!h->lm_rounds) {
    // the whole Levenberg-Marquardt loop of a pair in one workgroup (k_gicp_lm): one launch, no host poll
    GFS_LAUNCH("k_gicp_lm", k_gicp_lm, dim3(B), dim3(kLmBlock), 0, s, h->d_state.p, h->d_pts.p, h->d_cov6.p, h->d_ucell.p,
               h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_grid.p, h->d_ginfo.p, B, P, prm, h->d_tgt_index.p, h->d_maha6.p);
  } else {
    const int nblk_run = gfs::div_up(npts, kLinBlock);
    const int max_rounds = std::max(1, prm.max_iterations) * 11 + 2;
    const size_t buf_stride = (size_t)h->Bmax * P;  // elements between a pair's two (tgt_index, maha6 / 6) buffers
    int known_done = 0;  // pairs known to be done: the count polled one round behind
    // k_gicp_lm_coop takes the loop of `n_ub` pairs (an upper bound of those still iterating; the exact list, if any, is on the
    // device) off the host: the rest of the loop is ONE launch.  Its workgroups wait for each other, so they come out of the
    // process-wide budget; with too few left the rounds below carry on.
    static const bool lin_wg_default = !getenv("GFS_GICP_LIN_WG") || atoi(getenv("GFS_GICP_LIN_WG")) / 64 * 64 >= 256;
    const bool coop_ok = h->coop && !h->coop_failed && !prm.lin_tile && lin_wg_default && h->coop_cap > 0 && prm.max_iterations > 0;
    const int coop_budget = h->coop_cap / 2;
    auto launch_coop = [&](int n_ub, const int* act_list, const int* n_act_list) -> bool {
      static const int max_wg = getenv("GFS_GICP_COOP_MAXWG") ? std::max(1, atoi(getenv("GFS_GICP_COOP_MAXWG"))) : (1 << 30);  // debugging: fewer workgroups a pair
      const int want = std::min(std::min(n_ub * nblk_run, coop_budget), std::max(n_ub, max_wg));
      const int got = CoopBudget::reserve(h->device, want, std::max(n_ub, want / 2), coop_budget);
      if (got <= 0) return false;
      h->coop_reserved += got;
      h->coop_launches++;
      h->coop_last_wgs = got;
      GFS_LAUNCH("k_gicp_lm_coop", k_gicp_lm_coop, dim3(got), dim3(kLinBlock), 0, s, h->d_state.p, h->d_pts.p, h->d_cov6.p, h->d_ucell.p,
                 h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_grid.p, h->d_ginfo.p, nblk_run, P, prm, h->d_tgt_index.p, h->d_maha6.p,
                 buf_stride, h->d_partial.p, h->d_epartial.p, h->nblk, act_list, n_act_list, n_ub, h->d_sync.p, h->d_ndone.p, h->d_ndone.p + 2);
      return true;
    };
    // a batch small enough for a workgroup per chunk of every pair (one live stream: B = 1) runs its whole loop there
    bool in_coop = coop_ok && (long long)B * nblk_run <= coop_budget && launch_coop(B, nullptr, nullptr);
    for (int round = 0; round < max_rounds && !in_coop; round++) {
      // Late rounds run for a handful of pairs: their grids are cut to the pairs still iterating.  The host knows an upper bound
      // (the done counter it polled for round - 2); the list itself is written on the device by the previous round's
      // k_gicp_decide.  While most pairs are active the full grid with its XCD-aware pair -> block map is kept.
      const int ub = B - known_done;
      // ... and once the pairs left would each get at least 1 / coop_tail of a workgroup per chunk, the tail goes to the cooperative kernel
      if (coop_ok && round >= 2 && ub < B && (double)ub * nblk_run <= h->coop_tail * coop_budget &&
          launch_coop(ub, h->d_active.p + (size_t)(round % 3) * B, h->d_nactive.p + (round % 3))) {
        in_coop = true;
        break;
      }
      const bool listed = round >= 2 && 4 * ub <= B;
      const int* act = listed ? h->d_active.p + (size_t)(round % 3) * B : nullptr;
      const int* n_act = h->d_nactive.p + (round % 3);
      int* act_next = h->d_active.p + (size_t)((round + 1) % 3) * B;
      int* n_act_next = h->d_nactive.p + ((round + 1) % 3);
      int* n_act_clear = h->d_nactive.p + ((round + 2) % 3);
      const dim3 grid_pts(listed ? std::max(ub, 1) * nblk_run : xcd_grid(nblk_run, B, 1));
      // GFS_GICP_LIN_WG=64 / 128: the kernels without LDS as one- or two-wave workgroups, four or two to a chunk (lin_point_of).
      // Measured in round 4: k_gicp_linearize alone 150 -> 137 us per launch of 512 pairs with one-wave workgroups (finished waves
      // make room at once), the overlapped bench unchanged to slightly lower (37.2 against 37.3 k frames/s, the 64-pair block 28.8
      // against 29.0 k: the other lanes' kernels already fill those gaps, and four times the workgroups go through the dispatcher)
      static const int lin_wg = getenv("GFS_GICP_LIN_WG") ? std::max(64, std::min(256, atoi(getenv("GFS_GICP_LIN_WG")) / 64 * 64)) : 256;
      const int per = kLinBlock / (lin_wg == 128 ? 128 : lin_wg == 256 ? 256 : 64), wg = kLinBlock / per;
      const dim3 grid_w(listed ? std::max(ub, 1) * nblk_run * per : xcd_grid(nblk_run * per, B, 1));
      // one pass per round: the pending trials' errors + the linearisations, and -- in the pair's last workgroup -- its scalar step
      // (GFS_GICP_LIN_WG = 64 / 128, workgroups smaller than the fold needs: the step as a launch of its own, k_gicp_step)
      // GFS_GICP_FUSE_STEP=1 (256-thread workgroups): the step inside the pass, taken by the pair's last workgroup -- a round is ONE
      // launch.  Measured (round 6): SLOWER -- a lane of 32 pairs 8 x (45 + 11) -> 8 x 60 us of kernels a call, the 64-pair block 2.03 ->
      // 2.07 ms, the headline 42.8 -> 41.3 k frames/s: the last workgroup's fold through L1-bypassing loads and its one-lane solve
      // lengthen every pass by more than the launch boundary they save.  Kept behind the knob with its parity test.
      const bool fuse = h->fuse_step && wg == kLinBlock;
      if (prm.lin_tile) {
        GFS_LAUNCH("k_gicp_linearize", (k_gicp_linearize<true, false>), grid_pts, dim3(kLinBlock), 0, s, h->d_state.p,
                   h->d_pts.p, h->d_cov6.p, h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_grid.p, h->d_ginfo.p,
                   nblk_run, B, P, prm, h->d_tgt_index.p, h->d_maha6.p, buf_stride, h->d_partial.p, h->d_epartial.p, h->nblk, act, n_act,
                   nullptr, nullptr, nullptr, nullptr, nullptr);
      } else if (fuse) {
        GFS_LAUNCH("k_gicp_linearize", (k_gicp_linearize<false, true>), grid_w, dim3(wg), 0, s, h->d_state.p,
                   h->d_pts.p, h->d_cov6.p, h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_grid.p, h->d_ginfo.p,
                   nblk_run * per, B, P, prm, h->d_tgt_index.p, h->d_maha6.p, buf_stride, h->d_partial.p, h->d_epartial.p, h->nblk, act, n_act,
                   h->d_sync.p, h->d_ndone.p, act_next, n_act_next, n_act_clear);
      } else {
        GFS_LAUNCH("k_gicp_linearize", (k_gicp_linearize<false, false>), grid_w, dim3(wg), 0, s, h->d_state.p,
                   h->d_pts.p, h->d_cov6.p, h->d_ucell.p, h->d_ubegin.p, h->d_nucell.p, h->d_m.p, h->d_grid.p, h->d_ginfo.p,
                   nblk_run * per, B, P, prm, h->d_tgt_index.p, h->d_maha6.p, buf_stride, h->d_partial.p, h->d_epartial.p, h->nblk, act, n_act,
                   nullptr, nullptr, nullptr, nullptr, nullptr);
      }
      if (!fuse || prm.lin_tile)
        GFS_LAUNCH("k_gicp_step", k_gicp_step, dim3(B), dim3(256), 0, s, h->d_state.p, h->d_partial.p, h->d_epartial.p, h->d_m.p, h->nblk, prm,
                   h->d_ndone.p, act_next, n_act_next, n_act_clear);
      // One round (`ahead`, GFS_GICP_AHEAD) is queued beyond the one being polled: the GPU does not idle on the host's round trip; the
      // speculative round behind convergence finds every pair in phase 2 and its blocks exit at once.  The state machine lives on the
      // device, so how far ahead the host runs changes no result.  Round 6 measured deeper queues, because a kernel trace of the
      // 64-pair block under rocprofv3 shows the GICP stream idle for ~50 us behind every round's counter copy (tools/probes/
      // chain_timeline.py): ahead = 1 / 2 / 3 / 4 / 6 -> 2.000 / 2.014 / 2.019 / 2.041 / 2.054 ms per 64 pairs (2 lanes), 11.92 / 11.93 /
      // 11.98 / 12.05 / 12.09 ms per 512 -- the gaps are the profiler's, without it the host keeps up, and every extra round in flight
      // is two empty launches and a copy at the end of the call.
      const int nslot = h->ahead + 1, slot = round % nslot;
      GFS_HIP(hipMemcpyAsync(h->h_ndone.p + 2 * slot, h->d_ndone.p, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
      GFS_HIP(hipEventRecord(h->ev_round[slot], s));
      if (round >= h->ahead) {
        const int pslot = (round - h->ahead) % nslot;
        GFS_HIP(hipEventSynchronize(h->ev_round[pslot]));
        if (optimistic_sort && h->h_ndone.p[2 * pslot + 1] > 0) {
          // a cloud needed the voxel-sort kernels that were not launched (keys wider than 31 bits): everything again, with them
          GFS_HIP(hipStreamSynchronize(s));
          h->sort_all_kernels = true;
          const int rc_again = gicp_run(h, dev_target, dev_nt, dev_source, dev_ns, B, stride_pts, init_T, cfg, out, stream, streaming);
          h->sort_all_kernels = false;
          h->sort_all_hold = 64;
          return rc_again;
        }
        known_done = h->h_ndone.p[2 * pslot];
        if (known_done >= B) break;
      }
    }
  }
  int* const last_look = h->h_ndone.p + 2 * (gfs_gicp::kAheadMax + 1);  // {pairs done, clouds the LDS sorts left, workgroups that gave up waiting}
  last_look[0] = last_look[1] = last_look[2] = 0;
  GFS_LAUNCH("k_gicp_publish", k_gicp_publish, dim3(std::max(1, std::min(16, gfs::div_up(B * kStateWords, 256)))), dim3(256), 0, s, h->d_state.p,
             h->d_m.p, h->d_ndone.p, B, reinterpret_cast<u64*>(h->hd_state), h->hd_m, h->hd_last);
  const hipError_t rc_sync = hipStreamSynchronize(s);
  CoopBudget::release(h->device, h->coop_reserved);  // (whatever happened: the kernel is not running any more)
  h->coop_reserved = 0;
  GFS_HIP(rc_sync);
  if (last_look[2] > 0 && !h->coop_failed) {
    // a workgroup of k_gicp_lm_coop gave up waiting (its grid was not resident together: another process on this device?): never
    // return such a result -- the call again with a launch per step, and this handle stays with that
    h->coop_failed = true;
    return gicp_run(h, dev_target, dev_nt, dev_source, dev_ns, B, stride_pts, init_T, cfg, out, stream, streaming);
  }
  if (optimistic_sort && last_look[1] > 0) {  // (the polls inside the loop normally catch this after round 0; never return such a result)
    h->sort_all_kernels = true;
    const int rc_again = gicp_run(h, dev_target, dev_nt, dev_source, dev_ns, B, stride_pts, init_T, cfg, out, stream, streaming);
    h->sort_all_kernels = false;
    h->sort_all_hold = 64;
    return rc_again;
  }
  for (int b = 0; b < B; b++) {
    const PairState& S = h->h_state.p[b];
    gfs_gicp_result& r = out[b];
    memset(&r, 0, sizeof(r));
    for (int c = 0; c < 3; c++)
      for (int rr = 0; rr < 3; rr++) r.T_target_source[rr + 4 * c] = S.T[rr + 3 * c];
    for (int rr = 0; rr < 3; rr++) r.T_target_source[12 + rr] = S.T[9 + rr];
    r.T_target_source[15] = 1.0;
    r.converged = S.converged;
    r.iterations = (uint64_t)S.iterations;
    r.num_inliers = (uint64_t)S.inliers;
    int o = 0;
    for (int rr = 0; rr < 6; rr++)
      for (int c = rr; c < 6; c++) {
        r.H[rr + 6 * c] = S.H[o];
        r.H[c + 6 * rr] = S.H[o];
        o++;
      }
    for (int k = 0; k < 6; k++) r.b[k] = S.b[k];
    r.error = S.e;
    r.n_target_downsampled = h->h_m.p[2 * b + 1 - prm.src_slot];
    r.n_source_downsampled = h->h_m.p[2 * b + prm.src_slot];
    r.n_linearize = S.n_lin;
    r.n_error_evals = S.n_err;
  }
  h->last_B = B;
  h->last_src_slot = prm.src_slot;
  h->last_stride = stride_pts;
  h->last_leaf = cfg->downsampling_resolution;
  h->last_cell = prm.cell;
  h->last_k = cfg->num_neighbors;
  return GFS_OK;
}

int gfs_gicp_align_batch_device(gfs_gicp* h, const void* dev_target, const void* dev_nt, const void* dev_source,
                                const void* dev_ns, int B, int stride_pts, const double* init_T,
                                const gfs_gicp_config* cfg, gfs_gicp_result* out, void* stream) {
  return gicp_run(h, dev_target, dev_nt, dev_source, dev_ns, B, stride_pts, init_T, cfg, out, stream, false);
}

int gfs_gicp_align_next_batch_device(gfs_gicp* h, const void* dev_source, const void* dev_ns, int B, int stride_pts,
                                     const double* init_T, const gfs_gicp_config* cfg, gfs_gicp_result* out, void* stream) {
  return gicp_run(h, nullptr, nullptr, dev_source, dev_ns, B, stride_pts, init_T, cfg, out, stream, true);
}

int gfs_gicp_align_next(gfs_gicp* h, const float* source_xyzw, int ns, const double init_T_target_source[16],
                        const gfs_gicp_config* cfg, gfs_gicp_result* out) {
  GFS_REQUIRE(h && source_xyzw && ns >= 0 && cfg && out, GFS_ERR_INVALID_ARG, "gfs_gicp_align_next: invalid argument");
  GFS_REQUIRE(ns <= h->P, GFS_ERR_CAPACITY, "gfs_gicp_align_next: %d points exceed capacity %d", ns, h->P);
  std::lock_guard<std::recursive_mutex> lk(h->mu);  // staging buffers are shared: upload + run are one critical section
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (ns) GFS_HIP(hipMemcpyAsync(h->d_in_s.p, source_xyzw, (size_t)ns * 16, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_ns.p, &ns, sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipStreamSynchronize(s));
  return gicp_run(h, nullptr, nullptr, h->d_in_s.p, h->d_ns.p, 1, h->P, init_T_target_source, cfg, out, nullptr, true);
}

int gfs_gicp_align(gfs_gicp* h, const float* target_xyzw, int nt, const float* source_xyzw, int ns,
                   const double init_T_target_source[16], const gfs_gicp_config* cfg, gfs_gicp_result* out) {
  GFS_REQUIRE(h && target_xyzw && source_xyzw && nt >= 0 && ns >= 0 && cfg && out, GFS_ERR_INVALID_ARG,
              "gfs_gicp_align: invalid argument");
  GFS_REQUIRE(nt <= h->P && ns <= h->P, GFS_ERR_CAPACITY, "gfs_gicp_align: %d / %d points exceed capacity %d", nt, ns, h->P);
  std::lock_guard<std::recursive_mutex> lk(h->mu);  // staging buffers are shared: upload + run are one critical section
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  if (nt) GFS_HIP(hipMemcpyAsync(h->d_in_t.p, target_xyzw, (size_t)nt * 16, hipMemcpyHostToDevice, s));
  if (ns) GFS_HIP(hipMemcpyAsync(h->d_in_s.p, source_xyzw, (size_t)ns * 16, hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_nt.p, &nt, sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipMemcpyAsync(h->d_ns.p, &ns, sizeof(int), hipMemcpyHostToDevice, s));
  GFS_HIP(hipStreamSynchronize(s));
  return gfs_gicp_align_batch_device(h, h->d_in_t.p, h->d_nt.p, h->d_in_s.p, h->d_ns.p, 1, h->P, init_T_target_source, cfg,
                                     out, nullptr);
}

// Test hook (tests/test_gpu_gicp.py): runs the voxel sort of the preprocessing (voxel_qsort.hpp) on n caller-supplied 64-bit
// keys laid out like voxel keys (3 x 21 bits, or all ones = invalid) and returns the permutation.
// GPU test hook: vqs::wave_std_sort (csrc/wave_std_sort.hpp) on caller keys, n <= 1024; perm_out[i] = original index of the element
// that std::sort leaves at position i.
__global__ __launch_bounds__(64) void k_test_wave_std_sort(const unsigned* __restrict__ keys, int n, unsigned short* __restrict__ perm) {
  __shared__ unsigned K[1024];
  __shared__ unsigned short Pm[1024], l0[1024], l1[1024], cl[1024], st[3 * 40];
  for (int i = threadIdx.x; i < n; i += 64) {
    K[i] = keys[i];
    Pm[i] = (unsigned short)i;
  }
  VQS_WAVE_SYNC();
  vqs::wave_std_sort<unsigned>(K, Pm, l0, l1, cl, st, n);
  for (int i = threadIdx.x; i < n; i += 64) perm[i] = Pm[i];
}

int gfs_test_wave_std_sort(int device, const unsigned* keys, int n, unsigned short* perm_out) {
  GFS_REQUIRE(keys && perm_out && n >= 0 && n <= 1024, GFS_ERR_INVALID_ARG, "gfs_test_wave_std_sort: invalid argument");
  GFS_HIP(hipSetDevice(device));
  gfs::DevBuf<unsigned> dk;
  gfs::DevBuf<unsigned short> dp;
  int rc;
  if ((rc = dk.alloc((size_t)std::max(n, 1))) || (rc = dp.alloc((size_t)std::max(n, 1)))) return rc;
  if (n) GFS_HIP(hipMemcpy(dk.p, keys, (size_t)n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_test_wave_std_sort, dim3(1), dim3(64), 0, 0, dk.p, n, dp.p);
  GFS_HIP(hipGetLastError());
  GFS_HIP(hipDeviceSynchronize());
  if (n) GFS_HIP(hipMemcpy(perm_out, dp.p, (size_t)n * 2, hipMemcpyDeviceToHost));
  return GFS_OK;
}

int gfs_test_voxel_sort(gfs_gicp* h, const unsigned long long* keys, int n, unsigned* perm_out) {
  GFS_REQUIRE(h && keys && perm_out && n >= 0 && n <= h->P, GFS_ERR_INVALID_ARG, "gfs_test_voxel_sort: invalid argument");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const int P = h->P;
  std::vector<unsigned> iota((size_t)std::max(n, 1));
  for (int i = 0; i < n; i++) iota[i] = (unsigned)i;
  int counts[2] = {n, 0};
  GFS_HIP(hipMemcpyAsync(h->d_counts.p, counts, sizeof(counts), hipMemcpyHostToDevice, s));
  if (n) GFS_HIP(hipMemcpyAsync(h->d_keys0.p, keys, (size_t)n * 8, hipMemcpyHostToDevice, s));
  if (n) GFS_HIP(hipMemcpyAsync(h->d_val0.p, iota.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  const int leaf_parts = std::max(1, std::min(64, P / 1024));  // x 4 waves: a wave per leaf for nearly every cloud (the longest leaf sets the time)
  GFS_HIP(hipMemsetAsync(h->d_nheap.p, 0, 2 * sizeof(int), s));
  GFS_HIP(voxel_qsort_top(h, 2, s, -1, nullptr));
  {
    const int rc_leaf = voxel_qsort_leaves(h, 2, leaf_parts, s, -1, false);
    if (rc_leaf) return rc_leaf;
  }
  if (n) GFS_HIP(hipMemcpyAsync(perm_out, h->d_val0.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  GFS_HIP(hipStreamSynchronize(s));
  return GFS_OK;
}

int gfs_gicp_tile_stats(gfs_gicp* h, unsigned long long* out8, int reset) {
  GFS_REQUIRE(h && out8, GFS_ERR_INVALID_ARG, "gfs_gicp_tile_stats: invalid argument");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  unsigned v[8];
  GFS_HIP(hipMemcpy(v, h->d_tile_stats.p, sizeof(v), hipMemcpyDeviceToHost));
  for (int k = 0; k < 8; k++) out8[k] = v[k];
  if (reset) GFS_HIP(hipMemset(h->d_tile_stats.p, 0, sizeof(v)));
  return GFS_OK;
}

// Diagnostics: how many queries of cloud (b, which) of the last call went to the deferred k-NN passes, and the far ones' bounds.
// out = {points, deferred to the r = 2 pass, deferred to the isolated-point pass}; dk (may be null): the latter's bounds (cap entries)
int gfs_gicp_coop_stats(gfs_gicp* h, int out[4]) {
  GFS_REQUIRE(h && out, GFS_ERR_INVALID_ARG, "gfs_gicp_coop_stats: invalid argument");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  out[0] = h->coop_launches;
  out[1] = h->coop_last_wgs;
  out[2] = h->coop_failed ? 1 : 0;
  out[3] = h->coop && h->lm_rounds ? h->coop_cap / 2 : 0;
  return GFS_OK;
}

int gfs_gicp_knn_stats(gfs_gicp* h, int b, int which, int out[3], double* dk, int cap) {
  GFS_REQUIRE(h && b >= 0 && b < h->last_B && (which == 0 || which == 1) && out, GFS_ERR_INVALID_ARG, "gfs_gicp_knn_stats: invalid argument");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  GFS_HIP(hipDeviceSynchronize());
  const int c = 2 * b + (which ? h->last_src_slot : 1 - h->last_src_slot);
  int gi[8];
  GFS_HIP(hipMemcpy(out, h->d_m.p + c, sizeof(int), hipMemcpyDeviceToHost));
  GFS_HIP(hipMemcpy(gi, h->d_ginfo.p + 8 * c, sizeof(gi), hipMemcpyDeviceToHost));
  out[1] = gi[7];
  GFS_HIP(hipMemcpy(out + 2, h->d_far2.p + c, sizeof(int), hipMemcpyDeviceToHost));
  const int n = std::min(out[2], cap);
  if (dk && n > 0) {
    std::vector<double> t((size_t)n);
    GFS_HIP(hipMemcpy(t.data(), h->d_hard_d.p + (size_t)c * h->P + (h->P - n), (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) dk[k] = t[(size_t)(n - 1 - k)];
  }
  return GFS_OK;
}

int gfs_gicp_fetch_preprocessed(gfs_gicp* h, int b, int which, double* pts, double* covs, int cap, int* m) {
  GFS_REQUIRE(h && b >= 0 && b < h->last_B && (which == 0 || which == 1) && m, GFS_ERR_INVALID_ARG,
              "gfs_gicp_fetch_preprocessed: invalid argument");
  std::lock_guard<std::recursive_mutex> lk(h->mu);
  GFS_HIP(hipSetDevice(h->device));
  GFS_HIP(hipDeviceSynchronize());
  const int c = 2 * b + (which ? h->last_src_slot : 1 - h->last_src_slot);
  int mm = 0;
  GFS_HIP(hipMemcpy(&mm, h->d_m.p + c, sizeof(int), hipMemcpyDeviceToHost));
  *m = mm;
  GFS_REQUIRE(mm <= cap, GFS_ERR_CAPACITY, "gfs_gicp_fetch_preprocessed: %d points exceed capacity %d", mm, cap);
  if (pts && mm) GFS_HIP(hipMemcpy(pts, h->d_pts.p + (size_t)c * h->P, (size_t)mm * 32, hipMemcpyDeviceToHost));
  if (covs && mm) {
    std::vector<double> c6((size_t)mm * 6);
    GFS_HIP(hipMemcpy(c6.data(), h->d_cov6.p + (size_t)c * h->P * 6, (size_t)mm * 48, hipMemcpyDeviceToHost));
    for (int i = 0; i < mm; i++) {
      const double* s6 = &c6[(size_t)i * 6];
      double* o = covs + (size_t)i * 9;
      o[0] = s6[0];
      o[1] = s6[1];
      o[2] = s6[2];
      o[3] = s6[1];
      o[4] = s6[3];
      o[5] = s6[4];
      o[6] = s6[2];
      o[7] = s6[4];
      o[8] = s6[5];
    }
  }
  return GFS_OK;
}

}  // extern "C"
