// Device replica of small_gicp's quick_sort_omp (reference Thirdparty/small_gicp/include/small_gicp/util/sort_omp.hpp:58-85)
// as it is used by voxelgrid_sampling_omp (util/downsampling_omp.hpp:57) on (voxel key, point index) pairs with a
// comparator that looks at the key only.  That sort is NOT stable, and which points of a voxel end up on either side
// of a 1024-element block boundary of the sorted array (downsampling_omp.hpp:63-90) follows from the exact permutation
// it produces.  The permutation is a deterministic function of the key sequence (the OpenMP tasks work on disjoint
// ranges), so it is reproduced here step for step:
//
//   quick_sort_omp_impl, n >= 1024:  pivot = median of three medians of three (9 samples at n/8 strides),
//       middle1 = std::partition(first, last, key < pivot), middle2 = std::partition(middle1, last, !(pivot < key)),
//       recurse on [first, middle1) and [middle2, last)                                   -> k_voxel_qsort_top
//   n < 1024: std::sort = libstdc++ introsort (median-of-3 to first, unguarded Hoare partition, depth limit
//       2 lg n with heap-sort fallback, final insertion sort)                             -> k_voxel_qsort_leaf
//
// Parallel form (validated against libstdc++ on the CPU before it was written for the GPU):
//   * std::partition (bidirectional version, bits/stl_algo.h __partition): with m = number of elements satisfying the
//     predicate, the k-th non-satisfying element among the first m positions (from the left) is swapped with the k-th
//     satisfying element among the remaining positions (from the right); nothing else moves.  Ranks come from prefix
//     counts, the exchange goes through a side buffer -> one workgroup partitions ALL active ranges of a cloud per sweep.
//   * __unguarded_partition(first + 1, last, pivot = *first): left stoppers L (key >= pivot, ascending positions), right
//     stoppers R (key <= pivot, descending positions); pairs k with L_k < R_k are swapped; cut = min(L_K, R_{K-1}).
//   * __final_insertion_sort is a stable insertion sort of a sequence of chunks (<= 16 elements, the leaves of the
//     introsort recursion) that are already in order among themselves: it equals a stable sort of every chunk, i.e.
//     final position = chunk start + rank of (key, position) inside the chunk.
// tests/test_gpu_gicp.py compares the permutation with the oracle's (which calls libstdc++'s std::partition / std::sort).
#pragma once

#include "wave_std_sort.hpp"

namespace vqs {

typedef unsigned long long u64;
constexpr int kMaxSeg = 1024;      // active ranges (each >= 1024 elements) of one cloud: clouds of up to 2^20 points
constexpr int kLeafThreshold = 1024;  // sort_omp.hpp:61
constexpr int kIntroThreshold = 16;   // libstdc++ _S_threshold

__device__ __forceinline__ u64 lanemask_lt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// block-wide exclusive scan, 1024 threads (same helper as gicp.hip's, local copy to keep this header self-contained)
__device__ __forceinline__ int scan_1024(int v, int* s_wave /*16*/, int* total) {
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

__device__ __forceinline__ u64 med3(u64 a, u64 b, u64 c) {  // sort_omp.hpp:66-68 on the keys
  return a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b));
}

// One sweep over the elements of all active ranges ("virtual" index space = the ranges concatenated): wave w owns the
// virtual indices [w * 64 * E, (w + 1) * 64 * E), row r of it is one coalesced 64-element access.  Ranges are >= 1024
// long, so a row touches at most two of them.  f(valid, s, v) is called by all lanes of the wave for every row.
template <class F>
__device__ __forceinline__ void sweep(int A, int E, int nseg, const int* seg_v, F&& f) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int v0 = wave * 64 * E;
  if (v0 >= A) return;
  int lo = 0, hi = nseg;  // last s with seg_v[s] <= v0
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_v[mid] <= v0)
      lo = mid;
    else
      hi = mid;
  }
  int s_w = lo;
  for (int r = 0; r < E; r++) {
    const int row0 = v0 + r * 64;
    if (row0 >= A) break;
    const int v = row0 + lane;
    const bool valid = v < A;
    const int nxt = seg_v[s_w + 1];
    const int s = (valid && v >= nxt) ? s_w + 1 : s_w;
    f(valid, s, v);
    if (row0 + 64 >= nxt && s_w + 1 < nseg) s_w++;
  }
}

// ------------------------------------------------------------------------------------------------
// k_voxel_qsort_top: one 1024-thread workgroup per cloud.  Compacts the 3 x 21-bit voxel keys to the cloud's extent
// (order preserving, like k_radix_sort), then runs the n >= 1024 levels of quick_sort_omp_impl for all ranges of a level
// at once; ranges that drop below 1024 elements are appended to the cloud's leaf list for k_voxel_qsort_leaf.
// keys / vals are permuted in place; side_k / side_v hold the elements in flight of a partition sweep.
// kinfo[8c ..] = {xmin, ymin, zmin, bx, by, total bits}.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_voxel_qsort_top(u64* keys_all, unsigned* vals_all, u64* side_k_all, unsigned* side_v_all,
                                                          const int* __restrict__ counts, int P, int* __restrict__ which,
                                                          int* __restrict__ kinfo, unsigned* __restrict__ leaf_all,
                                                          int* __restrict__ nleaf, int only, int only_flagged) {
  constexpr u64 kInvalid = ~0ull;
  constexpr int kCB = 21, kCM = (1 << kCB) - 1;
  __shared__ int seg_b[kMaxSeg], seg_e[kMaxSeg], seg_v[kMaxSeg + 1], seg_base[kMaxSeg + 1], seg_off[kMaxSeg], seg_m1[kMaxSeg];
  __shared__ u64 seg_pv[kMaxSeg];
  __shared__ int s_wave[16];
  __shared__ int s_nseg, s_nleaf, s_A;
  __shared__ int s_mn[3], s_mx[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar
  if (only >= 0 && (c & 1) != only) return;
  if (only_flagged && kinfo[8 * c + 6] == 0) return;  // k_voxel_qsort_top_reg took this cloud
  const int n = counts[c];
  u64* ka = keys_all + (size_t)c * P;
  unsigned* va = vals_all + (size_t)c * P;
  u64* sk = side_k_all + (size_t)c * P;
  unsigned* sv = side_v_all + (size_t)c * P;
  unsigned* leaf = leaf_all + (size_t)c * P;
  // ---- key compaction (see k_radix_sort)
  if (tid < 3) {
    s_mn[tid] = 0x7fffffff;
    s_mx[tid] = -1;
  }
  __syncthreads();
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
    for (int i = tid; i < n; i += 1024) {
      const u64 k = ka[i];
      if (k == kInvalid) continue;
      const int f[3] = {(int)(k & kCM), (int)((k >> kCB) & kCM), (int)(k >> (2 * kCB))};
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
  for (int i = tid; i < n; i += 1024) {
    const u64 k = ka[i];
    if (k == kInvalid) continue;
    const u64 x = (k & kCM) - mnx, y = ((k >> kCB) & kCM) - mny, z = (k >> (2 * kCB)) - mnz;
    ka[i] = x | (y << bx) | (z << (bx + by));
  }
  if (tid == 0) {
    int* ki = kinfo + 8 * c;
    ki[0] = mnx;
    ki[1] = mny;
    ki[2] = mnz;
    ki[3] = bx;
    ki[4] = by;
    ki[5] = bx + by + bz;
    which[c] = 0;  // sorted in place
    s_nleaf = 0;
    s_nseg = 0;
    if (n >= kLeafThreshold) {
      seg_b[0] = 0;
      seg_e[0] = n;
      s_nseg = 1;
    } else if (n >= 2) {
      leaf[0] = 0u;
      leaf[1] = (unsigned)n;
      s_nleaf = 1;
    }
  }
  __syncthreads();
  // ---- levels of the 3-way quicksort
  while (true) {
    const int nseg = s_nseg;  // uniform (read after a barrier)
    if (nseg == 0) break;
    int my_b = 0, my_e = 0;
    if (tid < nseg) {
      my_b = seg_b[tid];
      my_e = seg_e[tid];
      const int len = my_e - my_b, off = len / 8;  // sort_omp.hpp:70-75
      const u64* f = ka + my_b;
      const u64 m1 = med3(f[0], f[off], f[off * 2]), m2 = med3(f[off * 3], f[off * 4], f[off * 5]),
                m3 = med3(f[off * 6], f[off * 7], f[len - 1]);
      seg_pv[tid] = med3(m1, m2, m3);
      seg_off[tid] = 0;
    }
    {
      int tot;
      const int ex = scan_1024(tid < nseg ? my_e - my_b : 0, s_wave, &tot);
      if (tid < nseg) seg_v[tid] = ex;
      if (tid == 0) {
        seg_v[nseg] = tot;
        s_A = tot;
      }
    }
    __syncthreads();
    const int A = s_A, E = (A + 1023) / 1024;
    for (int pass = 0; pass < 2; pass++) {
      // pass 0: std::partition(first, last, key < pivot); pass 1: std::partition(middle1, last, !(pivot < key))
      auto flag_of = [&](bool valid, int s, int v, u64* key_out, int* i_out) {
        bool flag = false;
        if (valid) {
          const int li = v - seg_v[s], i = seg_b[s] + li;
          const u64 key = ka[i], pv = seg_pv[s];
          *key_out = key;
          *i_out = i;
          flag = pass == 0 ? key < pv : (li >= seg_off[s] && !(pv < key));
        }
        return flag;
      };
      // (1) flagged elements per wave
      {
        int cnt = 0;
        sweep(A, E, nseg, seg_v, [&](bool valid, int s, int v) {
          u64 key;
          int i;
          const bool flag = flag_of(valid, s, v, &key, &i);
          cnt += __popcll(__ballot(flag));
        });
        if (lane == 0) s_wave[wave] = cnt;
      }
      __syncthreads();
      // (2) prefix count at the first element of every range
      int wave_base = 0, total = 0;
      for (int w = 0; w < 16; w++) {
        const int t = s_wave[w];
        if (w < wave) wave_base += t;
        total += t;
      }
      {
        int run = wave_base;
        sweep(A, E, nseg, seg_v, [&](bool valid, int s, int v) {
          u64 key;
          int i;
          const bool flag = flag_of(valid, s, v, &key, &i);
          const u64 b = __ballot(flag);
          if (valid && v == seg_v[s]) seg_base[s] = run + __popcll(b & lanemask_lt());
          run += __popcll(b);
        });
        if (tid == 0) seg_base[nseg] = total;
      }
      __syncthreads();
      // (3) the elements that std::partition swaps park themselves in the side buffer, (4) and fetch their partner
      for (int step = 0; step < 2; step++) {
        int run = wave_base;
        sweep(A, E, nseg, seg_v, [&](bool valid, int s, int v) {
          u64 key = 0;
          int i = 0;
          const bool flag = flag_of(valid, s, v, &key, &i);
          const u64 b = __ballot(flag);
          const int pre = run + __popcll(b & lanemask_lt());
          run += __popcll(b);
          if (!valid) return;
          const int vs = seg_v[s], off = seg_off[s], len = seg_e[s] - seg_b[s];
          const int lr = v - vs - off;  // index inside the partitioned region
          if (lr < 0) return;
          const int m = seg_base[s + 1] - seg_base[s], r = pre - seg_base[s];
          int mine = -1, theirs = -1;
          if (lr < m && !flag) {  // k-th misplaced element of the front part, from the left
            const int k = lr - r;
            mine = vs + off + k;
            theirs = vs + len - 1 - k;
          } else if (lr >= m && flag) {  // k-th misplaced element of the back part, from the right
            const int k = m - r - 1;
            mine = vs + len - 1 - k;
            theirs = vs + off + k;
          }
          if (mine < 0) return;
          if (step == 0) {
            sk[mine] = key;
            sv[mine] = va[i];
          } else {
            ka[i] = sk[theirs];
            va[i] = sv[theirs];
          }
        });
        __syncthreads();
      }
      if (tid < nseg) {
        const int m = seg_base[tid + 1] - seg_base[tid];
        if (pass == 0) {
          seg_m1[tid] = my_b + m;
          seg_off[tid] = m;
        } else {
          seg_off[tid] += m;  // middle2 - first
        }
      }
      __syncthreads();
    }
    // ---- children: [first, middle1) and [middle2, last)
    int child_b[2] = {0, 0}, child_e[2] = {0, 0}, nact = 0;
    if (tid < nseg) {
      child_b[0] = my_b;
      child_e[0] = seg_m1[tid];
      child_b[1] = my_b + seg_off[tid];
      child_e[1] = my_e;
      for (int k = 0; k < 2; k++) nact += (child_e[k] - child_b[k] >= kLeafThreshold) ? 1 : 0;
    }
    int tot_act;
    int pos = scan_1024(nact, s_wave, &tot_act);
    __syncthreads();  // everybody has read the old tables
    if (tid < nseg) {
      for (int k = 0; k < 2; k++) {
        const int len = child_e[k] - child_b[k];
        if (len >= kLeafThreshold) {
          seg_b[pos] = child_b[k];
          seg_e[pos] = child_e[k];
          pos++;
        } else if (len >= 2) {
          const int slot = atomicAdd(&s_nleaf, 1);
          leaf[2 * slot] = (unsigned)child_b[k];
          leaf[2 * slot + 1] = (unsigned)child_e[k];
        }
      }
    }
    if (tid == 0) s_nseg = tot_act;
    __syncthreads();
  }
  if (tid == 0) nleaf[c] = s_nleaf;
}

// ------------------------------------------------------------------------------------------------
// k_voxel_qsort_top_reg<EMAX>: the same levels for the common case — keys that compact to <= 31 bits and clouds of at most
// 1024 * EMAX points — with every thread caching its E <= EMAX elements (wave w owns positions [64 E w, 64 E (w + 1)), row r
// = one coalesced access) in registers for the whole partition sweep: per level one load, the four steps of std::partition
// (flag counts per wave -> prefix at the range starts -> misplaced elements parked in the side buffer -> partners fetched),
// one store.  The second std::partition of a level (elements equal to the pivot to the front of [middle1, last)) moves at
// most as many elements as there are points in the pivot's voxel: the lanes that end up holding such a key report their
// positions, and one thread per range replays the few swaps directly in memory (a full register sweep only if a range has
// more than kEq of them).  Clouds that do not qualify are flagged (kinfo[8c + 6] = 1) for k_voxel_qsort_top.
// ------------------------------------------------------------------------------------------------
template <int EMAX>
__global__ __launch_bounds__(1024) void k_voxel_qsort_top_reg(u64* keys_all, unsigned* vals_all, u64* side_all, unsigned* kscr_all,
                                                              const int* __restrict__ counts, int P, int* __restrict__ which,
                                                              int* __restrict__ kinfo, unsigned* __restrict__ leaf_all,
                                                              int* __restrict__ nleaf, int only, int only_flagged) {
  constexpr u64 kInvalid = ~0ull;
  constexpr int kCB = 21, kCM = (1 << kCB) - 1;
  constexpr int kSeg = EMAX + 1, kEq = 32;
  __shared__ int seg_b[kSeg], seg_e[kSeg], seg_base[kSeg + 1], seg_m1[kSeg], seg_m2[kSeg], eq_cnt[kSeg], eq_pos[kSeg][kEq];
  __shared__ int seg_local[kSeg], seg_wave[kSeg];  // flagged count before a range's first element inside its wave's chunk; that wave
  __shared__ unsigned seg_pv[kSeg];
  __shared__ int s_wave[16], s_wbase[17];
  __shared__ unsigned short side_pos[1024 * EMAX];  // rendezvous of a partition sweep: slot (rank from either end) -> position
  __shared__ int s_nseg, s_nleaf, s_over;
  __shared__ int s_mn[3], s_mx[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar
  if (only >= 0 && (c & 1) != only) return;
  if (only_flagged && kinfo[8 * c + 6] == 0) return;  // k_voxel_qsort_top_lds took this cloud
  const int n = __builtin_amdgcn_readfirstlane(counts[c]);
  u64* ka = keys_all + (size_t)c * P;
  unsigned* va = vals_all + (size_t)c * P;
  // Two copies of (key, point index): a partition sweep reads one and writes the other (elements are re-read in every step of a
  // sweep, coalesced and from L2, rather than held in registers: it keeps the kernel at 2 workgroups per compute unit)
  unsigned* kscr = kscr_all + (size_t)c * P;
  // (picked by a select, not kept in an array of pointers: the accesses stay GLOBAL instructions instead of FLAT ones)
  unsigned* const kbuf0 = kscr;
  unsigned* const kbuf1 = (unsigned*)(side_all + (size_t)c * P);
  unsigned* const vbuf0 = va;
  unsigned* const vbuf1 = (unsigned*)(side_all + (size_t)c * P) + P;
  auto kbuf = [&](int w) { return w ? kbuf1 : kbuf0; };
  auto vbuf = [&](int w) { return w ? vbuf1 : vbuf0; };
  int cur = 0;
  unsigned* leaf = leaf_all + (size_t)c * P;
  if (tid < 3) {
    s_mn[tid] = 0x7fffffff;
    s_mx[tid] = -1;
  }
  __syncthreads();
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
    for (int i = tid; i < n; i += 1024) {
      const u64 k = ka[i];
      if (k == kInvalid) continue;
      const int f[3] = {(int)(k & kCM), (int)((k >> kCB) & kCM), (int)(k >> (2 * kCB))};
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
  if (bx + by + bz > 31 || n > 1024 * EMAX) {  // uniform: left to the general kernel
    if (tid == 0) kinfo[8 * c + 6] = 1;
    return;
  }
  for (int i = tid; i < n; i += 1024) {
    const u64 k = ka[i];
    unsigned kk = 0xffffffffu;
    if (k != kInvalid) kk = (unsigned)((k & kCM) - mnx) | ((unsigned)(((k >> kCB) & kCM) - mny) << bx) | ((unsigned)((k >> (2 * kCB)) - mnz) << (bx + by));
    kscr[i] = kk;
  }
  if (tid == 0) {
    int* ki = kinfo + 8 * c;
    ki[0] = mnx;
    ki[1] = mny;
    ki[2] = mnz;
    ki[3] = bx;
    ki[4] = by;
    ki[5] = bx + by + bz;
    ki[6] = 0;
    which[c] = 0;
    s_nleaf = 0;
    s_nseg = 0;
    if (n >= kLeafThreshold) {
      seg_b[0] = 0;
      seg_e[0] = n;
      s_nseg = 1;
    } else if (n >= 2) {
      leaf[0] = 0u;
      leaf[1] = (unsigned)n;
      s_nleaf = 1;
    }
  }
  __syncthreads();
  const int E = (n + 1023) / 1024;
  const u64 ltm = lanemask_lt();
  while (true) {
    const int nseg = __builtin_amdgcn_readfirstlane(s_nseg);
    if (nseg == 0) break;
    if (tid < nseg) {
      const int b = seg_b[tid], len = seg_e[tid] - b, off = len / 8;  // sort_omp.hpp:70-75
      const unsigned* f = kbuf(cur) + b;
      auto m3 = [](unsigned x, unsigned y, unsigned z) { return x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y)); };
      const unsigned m1 = m3(f[0], f[off], f[off * 2]), m2 = m3(f[off * 3], f[off * 4], f[off * 5]),
                     mm = m3(f[off * 6], f[off * 7], f[len - 1]);
      seg_pv[tid] = m3(m1, m2, mm);
      eq_cnt[tid] = 0;
    }
    if (tid == 0) s_over = 0;
    __syncthreads();
    struct SegP {
      int b, e, first, base, m;
      unsigned pv;
    };
    // Stores every cached element where std::partition leaves it: the k-th misplaced element of the front part (from the
    // left) and the k-th misplaced element of the back part (from the right) announce their positions in side_pos and take
    // each other's place; everything else stays.  (All elements are in registers, so the scatter is in place.)
    auto partition_sweep = [&](int mode) {
      const unsigned* kin = kbuf(cur);
      const unsigned* vin = vbuf(cur);
      unsigned* kout = kbuf(cur ^ 1);
      unsigned* vout = vbuf(cur ^ 1);
      auto load_seg = [&](int sidx, bool with_counts) {
        SegP q;
        q.b = q.e = q.first = 0x7fffffff;
        q.base = q.m = 0;
        q.pv = 0;
        if (sidx < nseg) {  // wave-uniform: kept in scalar registers
          q.b = __builtin_amdgcn_readfirstlane(seg_b[sidx]);
          q.e = __builtin_amdgcn_readfirstlane(seg_e[sidx]);
          q.first = mode == 0 ? q.b : __builtin_amdgcn_readfirstlane(seg_m1[sidx]);
          q.pv = (unsigned)__builtin_amdgcn_readfirstlane((int)seg_pv[sidx]);
          if (with_counts) {
            q.base = __builtin_amdgcn_readfirstlane(seg_base[sidx]);
            q.m = __builtin_amdgcn_readfirstlane(seg_base[sidx + 1]) - q.base;
          }
        }
        return q;
      };
#define VQS_ROW_BEGIN(with_counts)                                       \
  const int row0 = (wave * E + r) * 64, i = row0 + lane;                 \
  while (row0 >= A.e && sA < nseg) {                                     \
    A = B;                                                               \
    sA++;                                                                \
    B = load_seg(sA + 1, with_counts);                                   \
  }                                                                      \
  const bool inB = i >= B.b, in = inB || (i >= A.b && i < A.e);          \
  const unsigned pv = inB ? B.pv : A.pv;                                 \
  const int first = inB ? B.first : A.first;                             \
  const unsigned key_r = key_nx;                                         \
  {                                                                      \
    const int i_nx = i + 64;  /* next row of this wave, fetched while this one is processed */ \
    key_nx = (r + 1 < E && i_nx < n) ? kin[i_nx] : 0u;                   \
  }                                                                      \
  const bool flag = in && (mode == 0 ? key_r < pv : (i >= first && !(pv < key_r)));
      {  // (A) flagged elements per wave, and before every range start inside its wave
        int cnt = 0, sA = 0;
        SegP A = load_seg(0, false), B = load_seg(1, false);
        unsigned key_nx = (wave * E * 64 + lane) < n ? kin[wave * E * 64 + lane] : 0u;
#pragma unroll 1
        for (int r = 0; r < E; r++) {
            VQS_ROW_BEGIN(false)
            const u64 bl = __ballot(flag);
            if (in && i == (inB ? B.b : A.b)) {
              const int sidx = inB ? sA + 1 : sA;
              seg_local[sidx] = cnt + __popcll(bl & ltm);
              seg_wave[sidx] = wave;
            }
            cnt += __popcll(bl);
            __builtin_amdgcn_sched_barrier(0);  // rows one after the other: interleaving them only costs registers
          }
        if (lane == 0) s_wave[wave] = cnt;
      }
      __syncthreads();
      if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; w++) {
          s_wbase[w] = acc;
          acc += s_wave[w];
        }
        s_wbase[16] = acc;
      }
      __syncthreads();
      if (tid < nseg) seg_base[tid] = s_wbase[seg_wave[tid]] + seg_local[tid];
      if (tid == 0) seg_base[nseg] = s_wbase[16];
      const int wave_base = __builtin_amdgcn_readfirstlane(s_wbase[wave]);
      __syncthreads();
      for (int step = 0; step < 2; step++) {  // (B) announce, (C) take the partner's place
        int run = wave_base, sA = 0;
        SegP A = load_seg(0, true), B = load_seg(1, true);
        unsigned key_nx = (wave * E * 64 + lane) < n ? kin[wave * E * 64 + lane] : 0u;
#pragma unroll 1
        for (int r = 0; r < E; r++) {
            VQS_ROW_BEGIN(true)
            const u64 bl = __ballot(flag);
            const int pre = run + __popcll(bl & ltm);
            run += __popcll(bl);
            int mine = -1, theirs = -1;
            if (in) {
              const int lr = i - first, last = inB ? B.e : A.e;
              if (lr >= 0) {
                const int m = inB ? B.m : A.m, rk = pre - (inB ? B.base : A.base);
                if (lr < m && !flag) {  // k-th misplaced element of the front part, from the left
                  const int k = lr - rk;
                  mine = first + k;
                  theirs = last - 1 - k;
                } else if (lr >= m && flag) {  // k-th misplaced element of the back part, from the right
                  const int k = m - rk - 1;
                  mine = last - 1 - k;
                  theirs = first + k;
                }
              }
            }
            if (step == 0) {
              if (mine >= 0) side_pos[mine] = (unsigned short)i;
            } else if (i < n) {  // every element goes to the other buffer: to its partner's place, or where it is
              const int dest = mine >= 0 ? (int)side_pos[theirs] : i;
              kout[dest] = key_r;
              vout[dest] = vin[i];
              // after the first partition: where the keys equal to the pivot sit (all of them in [middle1, last) now)
              if (in && mode == 0 && key_r == pv) {
                const int sidx = inB ? sA + 1 : sA;
                const int slot = atomicAdd(&eq_cnt[sidx], 1);
                if (slot < kEq) eq_pos[sidx][slot] = dest;
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        __syncthreads();
      }
#undef VQS_ROW_BEGIN
      cur ^= 1;
    };
    partition_sweep(0);
    if (tid < nseg) {  // second partition by replaying its few swaps
      const int m = eq_cnt[tid], m1 = seg_b[tid] + (seg_base[tid + 1] - seg_base[tid]);
      seg_m1[tid] = m1;
      seg_m2[tid] = m1 + m;
      if (m > kEq) {
        s_over = 1;
      } else {
        int* q = eq_pos[tid];  // sorted in place (ascending positions)
#pragma unroll 1
        for (int a = 1; a < m; a++) {
          const int v = q[a];
          int bpos = a;
          while (bpos > 0 && q[bpos - 1] > v) {
            q[bpos] = q[bpos - 1];
            bpos--;
          }
          q[bpos] = v;
        }
        // k-th position of [m1, m1 + m) not holding a pivot key (from the left) <-> k-th pivot key beyond (from the right)
        int a = 0, t = m - 1;
#pragma unroll 1
        for (int j = m1; j < m1 + m; j++) {
          if (a < m && q[a] == j) {
            a++;
            continue;
          }
          const int jt = q[t--];  // >= m1 + m by counting
          unsigned* kc = kbuf(cur);
          unsigned* vc = vbuf(cur);
          const unsigned kj = kc[j], vj = vc[j];
          kc[j] = kc[jt];
          vc[j] = vc[jt];
          kc[jt] = kj;
          vc[jt] = vj;
        }
      }
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_over)) {  // a voxel with more than kEq points: the general sweep
      partition_sweep(1);
      if (tid < nseg) seg_m2[tid] = seg_m1[tid] + (seg_base[tid + 1] - seg_base[tid]);
      __syncthreads();
    }
    // children [first, middle1) and [middle2, last)
    int child_b[2] = {0, 0}, child_e[2] = {0, 0}, nact = 0;
    if (tid < nseg) {
      child_b[0] = seg_b[tid];
      child_e[0] = seg_m1[tid];
      child_b[1] = seg_m2[tid];
      child_e[1] = seg_e[tid];
      for (int k = 0; k < 2; k++) nact += (child_e[k] - child_b[k] >= kLeafThreshold) ? 1 : 0;
    }
    int tot_act;
    int pos = scan_1024(nact, s_wave, &tot_act);
    __syncthreads();
    if (tid < nseg) {
      for (int k = 0; k < 2; k++) {
        const int len = child_e[k] - child_b[k];
        if (len >= kLeafThreshold) {
          seg_b[pos] = child_b[k];
          seg_e[pos] = child_e[k];
          pos++;
        } else if (len >= 2) {
          const int slot = atomicAdd(&s_nleaf, 1);
          leaf[2 * slot] = (unsigned)child_b[k];
          leaf[2 * slot + 1] = (unsigned)child_e[k];
        }
      }
    }
    if (tid == 0) s_nseg = tot_act;
    __syncthreads();
  }
  // the keys go back as 64-bit compacted keys (k_voxel_qsort_leaf, k_voxel_reduce)
  {
    const unsigned* kc = kbuf(cur);
    const unsigned* vc = vbuf(cur);
    for (int i = tid; i < n; i += 1024) {
      const unsigned k = kc[i];
      const unsigned v = vc[i];
      ka[i] = k == 0xffffffffu ? kInvalid : (u64)k;
      if (cur) va[i] = v;
    }
  }
  if (tid == 0) nleaf[c] = s_nleaf;
}

// ------------------------------------------------------------------------------------------------
// k_voxel_qsort_top_lds<EMAX>: k_voxel_qsort_top_reg with the cloud RESIDENT IN LDS for all levels (clouds of at most 1024 * EMAX
// points whose keys compact to <= 31 bits; EMAX = 19 covers a VGA depth image sampled with stride 4): 4-byte keys, 2-byte point
// indices (they are < n) and the 2-byte rendezvous array = 8 bytes an element = 152 KB of the 160.  The partition of a level works
// IN PLACE: std::partition only swaps pairs -- the k-th misplaced element of the front part (from the left) with the k-th misplaced
// element of the back part (from the right) -- so after the back elements have announced their positions, the thread that owns
// the FRONT element of a pair exchanges the two; no element is touched by two threads, nothing has to be copied aside, and the
// 2.4 GB per 1024 clouds that the global ping-pong buffers of k_voxel_qsort_top_reg move through HBM stay on the CU.
// Clouds that do not qualify are flagged kinfo[8c + 6] = 2 for k_voxel_qsort_top_reg (which flags 1 for k_voxel_qsort_top) and
// counted in *n_left.
// ------------------------------------------------------------------------------------------------
#ifdef GFS_VQS_TIMING
#define VQS_T_INIT long long vt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, vt_last = clock64(); int vt_lv = 0;
#define VQS_T(k) { if (k == 1) vt_lv++; const long long _n = clock64(); vt_acc[k] += _n - vt_last; vt_last = _n; }
#define VQS_T_END if (threadIdx.x == 0 && blockIdx.x < 4) printf("VQST c=%d n=%d lv=%d load=%lld pivot=%lld A=%lld scan=%lld B=%lld C=%lld replay=%lld child=%lld store=%lld Aset=%lld Aloop=%lld Bset=%lld Bloop=%lld\n", (int)blockIdx.x, n, vt_lv, vt_acc[0], vt_acc[1], vt_acc[2], vt_acc[3], vt_acc[4], vt_acc[5], vt_acc[6], vt_acc[7], vt_acc[8], vt_acc[9], vt_acc[10], vt_acc[11], vt_acc[12]);
#else
#define VQS_T_INIT
#define VQS_T(k)
#define VQS_T_END
#endif
// kGV (clouds of up to 1024 * EMAX = 37 888 points: a 1280x720 depth image sampled with stride 5): only the KEYS stay in LDS (4 bytes
// an element = 152 KB), the 2-byte point indices and the 2-byte rendezvous array live in a per-cloud scratch block in global memory
// (side_all: 4 bytes an element, L2-resident).  The phases that walk the whole array -- (A) counting and (B) announcing -- read keys
// only, i.e. LDS only; (B)'s scattered 2-byte stores and the K pair exchanges of (C) (index loads batched four pairs deep) are what
// touches global memory: per level about n / 2 two-byte stores and n / 4 pairs, against three full passes of 8-byte loads and stores for
// the ping-pong kernel (k_voxel_qsort_top_reg) these clouds used to fall back to.
template <int EMAX, bool kGV = false>
__global__ __launch_bounds__(1024) void k_voxel_qsort_top_lds(u64* keys_all, unsigned* vals_all, u64* side_all,
                                                              const int* __restrict__ counts, int P, int* __restrict__ which,
                                                              int* __restrict__ kinfo, unsigned* __restrict__ leaf_all,
                                                              int* __restrict__ nleaf, int only, int* __restrict__ n_left) {
  constexpr u64 kInvalid = ~0ull;
  constexpr int kCB = 21, kCM = (1 << kCB) - 1;
  constexpr int kSeg = EMAX + 1, kEq = 32;
  __shared__ int seg_b[kSeg], seg_e[kSeg], seg_base[kSeg + 1], seg_m1[kSeg], seg_m2[kSeg], eq_cnt[kSeg], eq_pos[kSeg][kEq];
  __shared__ int seg_local[kSeg], seg_wave[kSeg];  // flagged count before a range's first element inside its wave's chunk; that wave
  __shared__ unsigned seg_pv[kSeg];
  __shared__ int seg_K[kSeg];  // pairs a range's partition exchanges
  __shared__ unsigned short eq_j[kGV ? kSeg : 1][kEq];  // kGV: the front positions of the second partition's swaps
  __shared__ int s_wave[16], s_wbase[17];
  extern __shared__ __align__(16) unsigned char vq_lds[];
  VQS_T_INIT
  unsigned* s_key = reinterpret_cast<unsigned*>(vq_lds);                                  // [1024 * EMAX] compacted keys
  __shared__ int s_nseg, s_nleaf, s_over;
  __shared__ int s_mn[3], s_mx[3];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar
  if (only >= 0 && (c & 1) != only) return;
  // [1024 * EMAX] point indices (< n) and the rendezvous of a partition sweep: slot (rank from the front) -> position of the back element
  unsigned short* s_val = kGV ? reinterpret_cast<unsigned short*>(side_all + (size_t)c * P) : reinterpret_cast<unsigned short*>(s_key + 1024 * EMAX);
  unsigned short* side_pos = kGV ? s_val + P : s_val + 1024 * EMAX;
  const int n = __builtin_amdgcn_readfirstlane(counts[c]);
  u64* ka = keys_all + (size_t)c * P;
  unsigned* va = vals_all + (size_t)c * P;
  unsigned* leaf = leaf_all + (size_t)c * P;
  if (tid < 3) {
    s_mn[tid] = 0x7fffffff;
    s_mx[tid] = -1;
  }
  __syncthreads();
  // the thread's keys: all loads of a chunk in flight at once (one workgroup a CU: nothing else hides their latency); up to 19 rows
  // are ONE chunk that stays in registers for the compaction below, more rows are read twice, 19 at a time
  constexpr int kRows = EMAX <= 19 ? EMAX : 19, kChunks = (EMAX + kRows - 1) / kRows;
  u64 kreg[kRows];
  {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
#pragma unroll
    for (int ch = 0; ch < kChunks; ch++) {
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const int i = (ch * kRows + r) * 1024 + tid;
        kreg[r] = (ch * kRows + r < EMAX && i < n) ? ka[i] : kInvalid;
      }
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const u64 k = kreg[r];
        if (k == kInvalid) continue;
        const int f[3] = {(int)(k & kCM), (int)((k >> kCB) & kCM), (int)(k >> (2 * kCB))};
        for (int a = 0; a < 3; a++) {
          mn[a] = min(mn[a], f[a]);
          mx[a] = max(mx[a], f[a]);
        }
      }
    }
    for (int a = 0; a < 3; a++) {  // one LDS atomic a wave, not one a thread
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
  if (bx + by + bz > 31 || n > 1024 * EMAX) {  // uniform: left to the kernels that work through HBM
    if (tid == 0) {
      kinfo[8 * c + 6] = 2;
      if (n_left) {  // the host launches those kernels only when a cloud asked for them (gicp_run); until then the cloud stays
        atomicAdd(n_left, 1);  // unsorted and the leaf kernels must find nothing to do in it
        nleaf[c] = 0;
        which[c] = 0;
      }
    }
    return;
  }
#pragma unroll
  for (int ch = 0; ch < kChunks; ch++) {
    if (kChunks > 1) {
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const int i = (ch * kRows + r) * 1024 + tid;
        kreg[r] = (ch * kRows + r < EMAX && i < n) ? ka[i] : kInvalid;
      }
    }
#pragma unroll
    for (int r = 0; r < kRows; r++) {
      const int i = (ch * kRows + r) * 1024 + tid;
      const u64 k = kreg[r];
      unsigned kk = 0xffffffffu;
      if (k != kInvalid) kk = (unsigned)((k & kCM) - mnx) | ((unsigned)(((k >> kCB) & kCM) - mny) << bx) | ((unsigned)((k >> (2 * kCB)) - mnz) << (bx + by));
      if (ch * kRows + r < EMAX && i < n) {
        s_key[i] = kk;
        s_val[i] = (unsigned short)i;  // the point indices enter as the identity (k_voxel_keys, gfs_test_voxel_sort)
      }
    }
  }
  if (tid == 0) {
    int* ki = kinfo + 8 * c;
    ki[0] = mnx;
    ki[1] = mny;
    ki[2] = mnz;
    ki[3] = bx;
    ki[4] = by;
    ki[5] = bx + by + bz;
    ki[6] = 0;
    which[c] = 0;
    s_nleaf = 0;
    s_nseg = 0;
    if (n >= kLeafThreshold) {
      seg_b[0] = 0;
      seg_e[0] = n;
      s_nseg = 1;
    } else if (n >= 2) {
      leaf[0] = 0u;
      leaf[1] = (unsigned)n;
      s_nleaf = 1;
    }
  }
  __syncthreads();  VQS_T(0)
  const int E4 = (n + 4095) / 4096, chunk = E4 * 256;  // rows of 256 elements a wave; its part of the array
  const u64 ltm = lanemask_lt();
  while (true) {
    const int nseg = __builtin_amdgcn_readfirstlane(s_nseg);
    if (nseg == 0) break;
    if (tid < nseg) {
      const int b = seg_b[tid], len = seg_e[tid] - b, off = len / 8;  // sort_omp.hpp:70-75
      const unsigned* f = s_key + b;
      auto m3 = [](unsigned x, unsigned y, unsigned z) { return x < y ? (y < z ? y : (x < z ? z : x)) : (x < z ? x : (y < z ? z : y)); };
      const unsigned m1 = m3(f[0], f[off], f[off * 2]), m2 = m3(f[off * 3], f[off * 4], f[off * 5]),
                     mm = m3(f[off * 6], f[off * 7], f[len - 1]);
      seg_pv[tid] = m3(m1, m2, mm);
      eq_cnt[tid] = 0;
    }
    if (tid == 0) s_over = 0;
    __syncthreads();  VQS_T(1)
    struct SegP {
      int b, e, first, base, m;
      unsigned pv;
    };
    // std::partition only swaps pairs: the k-th misplaced element of the front part (from the left) with the k-th misplaced
    // element of the back part (from the right).  (A) counts, (B) both announce their positions -- the back one in slot
    // first + k, the front one in slot last - 1 - k -- and (C) exchanges the pairs, K of them a range.
    // (A) and (B) walk the array in rows of 256 elements (four consecutive ones a lane, one 16-byte LDS read): wave w owns
    // [w * chunk, (w + 1) * chunk).  A row lies inside one range almost always; then everything about the range is scalar.
    auto partition_sweep = [&](int mode) {
      auto load_seg = [&](int sidx, bool with_counts) {
        SegP q;
        q.b = q.e = q.first = 0x7fffffff;
        q.base = q.m = 0;
        q.pv = 0;
        if (sidx < nseg) {  // wave-uniform: kept in scalar registers
          q.b = __builtin_amdgcn_readfirstlane(seg_b[sidx]);
          q.e = __builtin_amdgcn_readfirstlane(seg_e[sidx]);
          q.first = mode == 0 ? q.b : __builtin_amdgcn_readfirstlane(seg_m1[sidx]);
          q.pv = (unsigned)__builtin_amdgcn_readfirstlane((int)seg_pv[sidx]);
          if (with_counts) {
            q.base = __builtin_amdgcn_readfirstlane(seg_base[sidx]);
            q.m = __builtin_amdgcn_readfirstlane(seg_base[sidx + 1]) - q.base;
          }
        }
        return q;
      };
#define VQS_ROW_BEGIN(with_counts)                                                   \
  const int row0 = wave * chunk + r * 256;                                           \
  if (row0 >= n) break;                                                              \
  while (row0 >= A.e && sA < nseg) {                                                 \
    A = B;                                                                           \
    sA++;                                                                            \
    B = load_seg(sA + 1, with_counts);                                               \
  }                                                                                  \
  const uint4 k4 = k4_nx;                                                            \
  if (r + 1 < E4) k4_nx = *reinterpret_cast<const uint4*>(s_key + row0 + 256 + 4 * lane); /* fetched while this row is processed */ \
  const unsigned kk[4] = {k4.x, k4.y, k4.z, k4.w};                                   \
  const bool whole = mode == 0 && row0 >= A.b && row0 + 256 <= A.e;  /* scalar */
      {  // (A) flagged elements per wave, and before every range start inside its wave
        int cnt = 0, sA = 0;
        uint4 k4_nx = *reinterpret_cast<const uint4*>(s_key + wave * chunk + 4 * lane);
        SegP A = load_seg(0, false), B = load_seg(1, false);
        VQS_T(9)
#pragma unroll 1
        for (int r = 0; r < E4; r++) {
          VQS_ROW_BEGIN(false)
          u64 mk[4];
          if (whole) {
#pragma unroll
            for (int j = 0; j < 4; j++) mk[j] = __ballot(kk[j] < A.pv);
            if (row0 == A.b && lane == 0) {
              seg_local[sA] = cnt;
              seg_wave[sA] = wave;
            }
          } else {
            bool f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int i = row0 + 4 * lane + j;
              const bool inB = i >= B.b, in = inB || (i >= A.b && i < A.e);
              const unsigned pv = inB ? B.pv : A.pv;
              const int first = inB ? B.first : A.first;
              f[j] = in && (mode == 0 ? kk[j] < pv : (i >= first && !(pv < kk[j])));
              mk[j] = __ballot(f[j]);
            }
            int before = cnt + __popcll(mk[0] & ltm) + __popcll(mk[1] & ltm) + __popcll(mk[2] & ltm) + __popcll(mk[3] & ltm);
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int i = row0 + 4 * lane + j;
              if (i == A.b || i == B.b) {
                const int sidx = i == B.b ? sA + 1 : sA;
                seg_local[sidx] = before;
                seg_wave[sidx] = wave;
              }
              before += f[j] ? 1 : 0;
            }
          }
          cnt += __popcll(mk[0]) + __popcll(mk[1]) + __popcll(mk[2]) + __popcll(mk[3]);
        }
        if (lane == 0) s_wave[wave] = cnt;
        VQS_T(10)
      }
      __syncthreads();  VQS_T(2)
      if (tid == 0) {
        int acc = 0;
        for (int w = 0; w < 16; w++) {
          s_wbase[w] = acc;
          acc += s_wave[w];
        }
        s_wbase[16] = acc;
      }
      __syncthreads();
      if (tid < nseg) {
        seg_base[tid] = s_wbase[seg_wave[tid]] + seg_local[tid];
        seg_K[tid] = 0;
      }
      if (tid == 0) seg_base[nseg] = s_wbase[16];
      const int wave_base = __builtin_amdgcn_readfirstlane(s_wbase[wave]);
      __syncthreads();  VQS_T(3)
      {  // (B) announce
        int run = wave_base, sA = 0;
        uint4 k4_nx = *reinterpret_cast<const uint4*>(s_key + wave * chunk + 4 * lane);
        SegP A = load_seg(0, true), B = load_seg(1, true);
        auto row_body = [&](auto whole_c, int row0, const unsigned* kk) {
          constexpr bool kWhole = decltype(whole_c)::value;
          bool f[4], in[4], inB[4];
          u64 mk[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int i = row0 + 4 * lane + j;
            inB[j] = kWhole ? false : i >= B.b;
            in[j] = kWhole ? true : (inB[j] || (i >= A.b && i < A.e));
            const unsigned pv = inB[j] ? B.pv : A.pv;
            const int first = inB[j] ? B.first : A.first;
            f[j] = in[j] && (mode == 0 ? kk[j] < pv : (i >= first && !(pv < kk[j])));
            mk[j] = __ballot(f[j]);
          }
          int pre = run + __popcll(mk[0] & ltm) + __popcll(mk[1] & ltm) + __popcll(mk[2] & ltm) + __popcll(mk[3] & ltm);
          bool eq[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int i = row0 + 4 * lane + j;
            const int first = inB[j] ? B.first : A.first, last = inB[j] ? B.e : A.e;
            const int m = inB[j] ? B.m : A.m, rk = pre - (inB[j] ? B.base : A.base);  // flagged elements of the range before this one
            const int lr = i - first;
            // the k-th misplaced element of the front part (from the left: not flagged, k = lr - rk) announces itself in slot
            // last - 1 - k, the k-th misplaced element of the back part (from the right: flagged, k = m - rk - 1) in slot first + k
            const bool moves = in[j] && lr >= 0 && (f[j] ? lr >= m : lr < m);
            const int slot = f[j] ? first + (m - rk - 1) : last - 1 - (lr - rk);
            if (moves) side_pos[slot] = (unsigned short)i;
            // after the first partition: where the keys equal to the pivot sit (all of them in [middle1, last) then); the ones
            // that move are recorded by (C)
            eq[j] = mode == 0 && in[j] && !f[j] && lr >= m && kk[j] == (inB[j] ? B.pv : A.pv);
            pre += f[j] ? 1 : 0;
          }
          if (__ballot(eq[0] || eq[1] || eq[2] || eq[3])) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (eq[j]) {
                const int sidx = inB[j] ? sA + 1 : sA;
                const int slot = atomicAdd(&eq_cnt[sidx], 1);
                if (slot < kEq) eq_pos[sidx][slot] = row0 + 4 * lane + j;
              }
            }
          }
          {  // K = the front part's elements that are not flagged: known to the thread of the front part's last element
            const int xa = A.first + A.m - 1, xb = B.first + B.m - 1;  // scalar
            if ((A.m > 0 && xa >= row0 && xa < row0 + 256) || (!kWhole && B.m > 0 && xb >= row0 && xb < row0 + 256)) {
              int pre2 = run + __popcll(mk[0] & ltm) + __popcll(mk[1] & ltm) + __popcll(mk[2] & ltm) + __popcll(mk[3] & ltm);
#pragma unroll
              for (int j = 0; j < 4; j++) {
                const int i = row0 + 4 * lane + j;
                if (in[j]) {
                  const int first = inB[j] ? B.first : A.first, m = inB[j] ? B.m : A.m, rk = pre2 - (inB[j] ? B.base : A.base);
                  if (i - first == m - 1) seg_K[inB[j] ? sA + 1 : sA] = m - rk - (f[j] ? 1 : 0);
                }
                pre2 += f[j] ? 1 : 0;
              }
            }
          }
          run += __popcll(mk[0]) + __popcll(mk[1]) + __popcll(mk[2]) + __popcll(mk[3]);
        };
        VQS_T(11)
#pragma unroll 1
        for (int r = 0; r < E4; r++) {
          VQS_ROW_BEGIN(true)
          if (whole)
            row_body(std::true_type{}, row0, kk);
          else
            row_body(std::false_type{}, row0, kk);
        }
        VQS_T(12)
      }
      __syncthreads();  VQS_T(4)
      if constexpr (kGV) {  // (C) with the point indices and the rendezvous array in global memory: four pairs a trip, their loads in flight together
        __threadfence_block();  // (B)'s rendezvous stores, made by other waves of this workgroup
        int sg = 0, off = 0, Ks = nseg > 0 ? seg_K[0] : 0, tbase_of_batch = 0;
        bool more = true;
        while (more) {
          int jb_slot[4], jf_slot[4], sgv[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            sgv[u] = -1;
            jb_slot[u] = jf_slot[u] = 0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (!more) continue;
            // (trip u of this batch: pair number tcur; the batches of a thread are 4096 pairs apart)
            const int tcur = tid + 1024 * u + tbase_of_batch;
            while (sg < nseg && tcur >= off + Ks) {
              off += Ks;
              sg++;
              Ks = sg < nseg ? seg_K[sg] : 0;
            }
            if (sg >= nseg) {
              more = false;
              continue;
            }
            const int k = tcur - off, first = mode == 0 ? seg_b[sg] : seg_m1[sg], last = seg_e[sg];
            sgv[u] = sg;
            jb_slot[u] = first + k;
            jf_slot[u] = last - 1 - k;
          }
          int jb[4], jf[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            jb[u] = sgv[u] >= 0 ? (int)side_pos[jb_slot[u]] : 0;
            jf[u] = sgv[u] >= 0 ? (int)side_pos[jf_slot[u]] : 0;
          }
          unsigned short vf[4], vb[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            vf[u] = sgv[u] >= 0 ? s_val[jf[u]] : (unsigned short)0;
            vb[u] = sgv[u] >= 0 ? s_val[jb[u]] : (unsigned short)0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (sgv[u] < 0) continue;
            const unsigned kf = s_key[jf[u]], kb = s_key[jb[u]];
            s_key[jf[u]] = kb;
            s_key[jb[u]] = kf;
            s_val[jf[u]] = vb[u];
            s_val[jb[u]] = vf[u];
            if (mode == 0 && kf == seg_pv[sgv[u]]) {
              const int slot = atomicAdd(&eq_cnt[sgv[u]], 1);
              if (slot < kEq) eq_pos[sgv[u]][slot] = jb[u];
            }
          }
          tbase_of_batch += 4096;
        }
        __threadfence_block();
      } else {  // (C) exchange the pairs, all ranges' pairs numbered through: thread t takes pairs t, t + 1024, ...
        int sg = 0, off = 0, Ks = nseg > 0 ? seg_K[0] : 0;
        for (int t = tid;; t += 1024) {
          while (sg < nseg && t >= off + Ks) {
            off += Ks;
            sg++;
            Ks = sg < nseg ? seg_K[sg] : 0;
          }
          if (sg >= nseg) break;
          const int k = t - off, first = mode == 0 ? seg_b[sg] : seg_m1[sg], last = seg_e[sg];
          const int jb = side_pos[first + k], jf = side_pos[last - 1 - k];
          const unsigned kf = s_key[jf], kb = s_key[jb];
          const unsigned short vf = s_val[jf], vb = s_val[jb];
          s_key[jf] = kb;
          s_val[jf] = vb;
          s_key[jb] = kf;
          s_val[jb] = vf;
          if (mode == 0 && kf == seg_pv[sg]) {
            const int slot = atomicAdd(&eq_cnt[sg], 1);
            if (slot < kEq) eq_pos[sg][slot] = jb;
          }
        }
      }
      __syncthreads();  VQS_T(5)
#undef VQS_ROW_BEGIN
    };
    partition_sweep(0);
    if (tid < nseg) {  // second partition by replaying its few swaps
      const int m = eq_cnt[tid], m1 = seg_b[tid] + (seg_base[tid + 1] - seg_base[tid]);
      seg_m1[tid] = m1;
      seg_m2[tid] = m1 + m;
      if (m > kEq) {
        s_over = 1;
      } else {
        int* q = eq_pos[tid];  // sorted in place (ascending positions)
#pragma unroll 1
        for (int a = 1; a < m; a++) {
          const int v = q[a];
          int bpos = a;
          while (bpos > 0 && q[bpos - 1] > v) {
            q[bpos] = q[bpos - 1];
            bpos--;
          }
          q[bpos] = v;
        }
        // k-th position of [m1, m1 + m) not holding a pivot key (from the left) <-> k-th pivot key beyond (from the right)
        int a = 0, t = m - 1;
        [[maybe_unused]] int nsw = 0;
#pragma unroll 1
        for (int j = m1; j < m1 + m; j++) {
          if (a < m && q[a] == j) {
            a++;
            continue;
          }
          const int jt = q[t--];  // >= m1 + m by counting
          const unsigned kj = s_key[j];
          s_key[j] = s_key[jt];
          s_key[jt] = kj;
          if constexpr (kGV) {  // the point indices follow below, their loads batched (a global round trip per swap otherwise)
            eq_j[tid][nsw++] = (unsigned short)j;
          } else {
            const unsigned short vj = s_val[j];
            s_val[j] = s_val[jt];
            s_val[jt] = vj;
          }
        }
        if constexpr (kGV) {  // swap s of the loop above paired j = eq_j[s] with jt = q[m - 1 - s]; all positions are distinct
#pragma unroll 1
          for (int s0 = 0; s0 < nsw; s0 += 4) {
            unsigned short va_[4], vb_[4];
            int ja[4], jb_[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const bool on = s0 + u < nsw;
              ja[u] = on ? (int)eq_j[tid][s0 + u] : 0;
              jb_[u] = on ? q[m - 1 - (s0 + u)] : 0;
              va_[u] = on ? s_val[ja[u]] : (unsigned short)0;
              vb_[u] = on ? s_val[jb_[u]] : (unsigned short)0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (s0 + u < nsw) {
                s_val[ja[u]] = vb_[u];
                s_val[jb_[u]] = va_[u];
              }
          }
        }
      }
    }
    __syncthreads();  VQS_T(6)
    if (__builtin_amdgcn_readfirstlane(s_over)) {  // a voxel with more than kEq points: the general sweep
      partition_sweep(1);
      if (tid < nseg) seg_m2[tid] = seg_m1[tid] + (seg_base[tid + 1] - seg_base[tid]);
      __syncthreads();
    }
    // children [first, middle1) and [middle2, last)
    int child_b[2] = {0, 0}, child_e[2] = {0, 0}, nact = 0;
    if (tid < nseg) {
      child_b[0] = seg_b[tid];
      child_e[0] = seg_m1[tid];
      child_b[1] = seg_m2[tid];
      child_e[1] = seg_e[tid];
      for (int k = 0; k < 2; k++) nact += (child_e[k] - child_b[k] >= kLeafThreshold) ? 1 : 0;
    }
    int tot_act;
    int pos = scan_1024(nact, s_wave, &tot_act);
    __syncthreads();
    if (tid < nseg) {
      for (int k = 0; k < 2; k++) {
        const int len = child_e[k] - child_b[k];
        if (len >= kLeafThreshold) {
          seg_b[pos] = child_b[k];
          seg_e[pos] = child_e[k];
          pos++;
        } else if (len >= 2) {
          const int slot = atomicAdd(&s_nleaf, 1);
          leaf[2 * slot] = (unsigned)child_b[k];
          leaf[2 * slot + 1] = (unsigned)child_e[k];
        }
      }
    }
    if (tid == 0) s_nseg = tot_act;
    __syncthreads();  VQS_T(7)
  }
  // the keys go back as 64-bit compacted keys (k_voxel_qsort_leaf, k_voxel_reduce)
  {
    for (int i = tid; i < n; i += 1024) {
      const unsigned k = s_key[i];
      ka[i] = k == 0xffffffffu ? kInvalid : (u64)k;
      va[i] = (unsigned)s_val[i];
    }
  }
  VQS_T(8) VQS_T_END
  if (tid == 0) nleaf[c] = s_nleaf;
}

// ------------------------------------------------------------------------------------------------
// k_voxel_qsort_leaf<KT>: std::sort of every leaf range (< 1024 elements) by one wave, in LDS.  KT = unsigned when the
// cloud's compacted keys fit 31 bits (kinfo[5] <= 31; the invalid key becomes 0xffffffff), else u64; clouds of the
// other width are skipped (both instantiations are launched).
// ------------------------------------------------------------------------------------------------
// (l0 / l1, the stopper lists of a partition: only their first K + 1 <= 512 entries are ever read — pair k is swapped while
// L_k < R_k, and 512 disjoint pairs do not fit a range of < 1024 elements.  The stack of pending parts lives in a register, entry k
// in lane k.  With 4-byte keys that is 40 KB a workgroup: four workgroups a CU instead of three.)
template <typename KT>
struct LeafLds {
  KT K[4][1024];
  unsigned short Pm[4][1024], l0[4][512], l1[4][512], cl[4][1024];
};

template <typename KT>
__global__ __launch_bounds__(256) void k_voxel_qsort_leaf(u64* keys_all, unsigned* vals_all,
                                                          const int* __restrict__ kinfo, const unsigned* __restrict__ leaf_all,
                                                          const int* __restrict__ nleaf, int P, int only,
                                                          unsigned* __restrict__ heap_all, int* __restrict__ nheap, int heap_cap) {
  constexpr bool kNarrow = sizeof(KT) == 4;
  __shared__ LeafLds<KT> S;
  const int c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (only >= 0 && (c & 1) != only) return;
  if ((kinfo[8 * c + 5] <= 31) != kNarrow) return;
  const int nl = nleaf[c];
  u64* ka = keys_all + (size_t)c * P;
  unsigned* va = vals_all + (size_t)c * P;
  const unsigned* leaf = leaf_all + (size_t)c * P;
  unsigned* heap = heap_all + (size_t)c * heap_cap;
  KT* K = S.K[wave];
  unsigned short* Pm = S.Pm[wave];
  unsigned short* l0 = S.l0[wave];
  unsigned short* l1 = S.l1[wave];
  unsigned short* cl = S.cl[wave];
  // stack of pending [lo, hi) parts with their depth budget: entry k = lane k's stk (lo | hi << 10 | depth << 21)
  unsigned stk = 0;
  auto stk_pack = [](int lo, int hi, int depth) { return (unsigned)lo | ((unsigned)hi << 10) | ((unsigned)depth << 21); };
  const u64 lt = lanemask_lt();
  for (int l = blockIdx.x * 4 + wave; l < nl; l += gridDim.x * 4) {
    const int b = __builtin_amdgcn_readfirstlane((int)leaf[2 * l]), n = __builtin_amdgcn_readfirstlane((int)leaf[2 * l + 1]) - b;
    for (int p = lane; p < n; p += 64) {
      const u64 k = ka[b + p];
      K[p] = kNarrow ? (KT)(k == ~0ull ? 0xffffffffull : k) : (KT)k;
      Pm[p] = (unsigned short)p;
    }
    int lg = 0;
    for (int v = n; v > 1; v >>= 1) lg++;
    int sp = 0;
    if (lane == 0) stk = stk_pack(0, n, 2 * lg);
    sp = 1;
    VQS_WAVE_SYNC();
    while (sp > 0) {  // __introsort_loop, the recursion on [cut, last) through a stack
      sp--;
      const unsigned top = (unsigned)__builtin_amdgcn_readlane((int)stk, __builtin_amdgcn_readfirstlane(sp));
      int lo = (int)(top & 1023u), hi = (int)((top >> 10) & 2047u), depth = (int)(top >> 21);
      bool heaped = false;
      while (hi - lo > kIntroThreshold) {
        if constexpr (kNarrow) {
          if (hi - lo <= 64 && depth > 0) {
            // Ranges of <= 64 elements: one element per lane, the rest of their introsort recursion runs in registers (no
            // stack: both parts of a partition stay in the lanes of their range); further small ranges from the stack share
            // the wave as long as they fit.  Lane-space coordinates: a range occupies lanes [mylo, myhi).
            int mylo = 0, myhi = 0, mydepth = 0, myS = 0, mypos0 = 0, total = 0;
            bool valid = false;
            int blo = lo, bhi = hi, bdepth = depth;
            while (true) {
              const int len = bhi - blo;
              if (lane >= total && lane < total + len) {
                valid = true;
                mylo = total;
                myhi = total + len;
                mydepth = bdepth;
                myS = total;
                mypos0 = blo;
              }
              total += len;
              if (sp == 0) break;
              const unsigned nxt = (unsigned)__builtin_amdgcn_readlane((int)stk, __builtin_amdgcn_readfirstlane(sp - 1));
              const int tlo = (int)(nxt & 1023u), thi = (int)((nxt >> 10) & 2047u), td = (int)(nxt >> 21);
              if (thi - tlo > 64 - total || (td == 0 && thi - tlo > kIntroThreshold)) break;
              sp--;
              blo = tlo;
              bhi = thi;
              bdepth = td;
            }
            const int p = valid ? mypos0 + (lane - myS) : 0;
            unsigned k = valid ? (unsigned)K[p] : 0u, pm = valid ? (unsigned)Pm[p] : 0u;
            const u64 bit = 1ull << lane, ltm = bit - 1ull, gtm = ~(bit | ltm);
            bool bail = false;
            while (true) {
              const bool act = valid && (myhi - mylo > kIntroThreshold);
              if (__ballot(act) == 0ull) break;
              if (__ballot(act && mydepth == 0) != 0ull) {  // depth limit inside the batch: back to the stack (heap sort there)
                bail = true;
                break;
              }
              if (act) mydepth--;
              auto pull = [&](unsigned v, int from) { return (unsigned)__builtin_amdgcn_ds_bpermute(from << 2, (int)v); };
              // __move_median_to_first(first, first + 1, mid, last - 1)
              const int A = act ? mylo + 1 : lane, B = act ? mylo + (myhi - mylo) / 2 : lane, C = act ? myhi - 1 : lane;
              const unsigned ka_ = pull(k, A), kb_ = pull(k, B), kc_ = pull(k, C);
              int pick;
              if (ka_ < kb_) {
                if (kb_ < kc_)
                  pick = B;
                else if (ka_ < kc_)
                  pick = C;
                else
                  pick = A;
              } else if (ka_ < kc_)
                pick = A;
              else if (kb_ < kc_)
                pick = C;
              else
                pick = B;
              const unsigned pv = pick == A ? ka_ : pick == B ? kb_ : kc_;
              int src = lane;
              if (act) src = lane == mylo ? pick : lane == pick ? mylo : lane;
              k = pull(k, src);
              pm = pull(pm, src);
              // __unguarded_partition(first + 1, last, first): left stoppers (key >= pivot) ascending, right stoppers
              // (key <= pivot) descending, pair r is swapped while L_r < R_r
              const bool interior = act && lane > mylo;
              const bool geL = interior && !(k < pv), leR = interior && !(pv < k);
              const u64 hi_m = myhi >= 64 ? ~0ull : ((1ull << myhi) - 1ull), lo_m = (2ull << mylo) - 1ull;
              const u64 rmask = hi_m & ~lo_m;  // lanes (mylo, myhi)
              const u64 bL = __ballot(geL) & rmask, bR = __ballot(leR) & rmask;
              const int rankL = __popcll(bL & ltm), rankR = __popcll(bR & gtm), nL = __popcll(bL), nR = __popcll(bR);
              const int park = valid ? mylo : lane;  // a lane nobody reads (pair slots start at mylo + 1)
              const int Lp = __builtin_amdgcn_ds_permute((geL ? mylo + 1 + rankL : park) << 2, lane);
              const int Rp = __builtin_amdgcn_ds_permute((leR ? mylo + 1 + rankR : park) << 2, lane);
              const int r = lane - (mylo + 1);
              const bool cond = interior && r < min(nL, nR) && Lp < Rp;
              const int ksw = __popcll(__ballot(cond) & rmask);
              const int pR = (int)pull((unsigned)Rp, geL ? mylo + 1 + rankL : lane);
              const int pL = (int)pull((unsigned)Lp, leR ? mylo + 1 + rankR : lane);
              src = lane;
              if (geL && rankL < ksw)
                src = pR;
              else if (leR && rankR < ksw)
                src = pL;
              k = pull(k, src);
              pm = pull(pm, src);
              const int cL = (int)pull((unsigned)Lp, act && ksw < nL ? mylo + 1 + ksw : lane);
              const int cR = (int)pull((unsigned)Rp, act && ksw > 0 ? mylo + ksw : lane);
              if (act) {
                int cut = myhi;
                if (ksw > 0) cut = cR;
                if (ksw < nL) cut = min(cut, cL);
                if (lane < cut)
                  myhi = cut;
                else
                  mylo = cut;
              }
            }
            if (valid) {
              K[p] = (KT)k;
              Pm[p] = (unsigned short)pm;
              const int len = myhi - mylo;
              if (len <= kIntroThreshold) cl[p] = (unsigned short)((mypos0 + (mylo - myS)) | (len << 10));
            }
            if (bail) {
              const bool mgr = valid && lane == mylo && (myhi - mylo > kIntroThreshold);
              const u64 bm = __ballot(mgr);
              const int slot = sp + __popcll(bm & ltm);
              // the managers push their entries to the lanes [sp, sp + count) (everybody else to lane 63: the stack is < 40 deep)
              const int got = __builtin_amdgcn_ds_permute((mgr ? slot : 63) << 2,
                                                          (int)stk_pack(mypos0 + (mylo - myS), mypos0 + (myhi - myS), mydepth));
              if (lane >= sp && lane < sp + (int)__popcll(bm)) stk = (unsigned)got;
              sp += __popcll(bm);
            }
            VQS_WAVE_SYNC();
            heaped = true;  // chunks are marked
            break;
          }
        }
        if (depth == 0) {
          // depth limit: the range goes to k_voxel_qsort_heap (a heap sort is one long serial chain; it gets a wave of its own
          // instead of holding up this leaf and its workgroup).  It stays as it is here: chunks of one below = identity.
          if (lane == 0) {
            const int slot = atomicAdd(&nheap[c], 1);
            heap[2 * slot] = (unsigned)(b + lo);
            heap[2 * slot + 1] = (unsigned)(b + hi);
          }
          for (int p = lo + lane; p < hi; p += 64) cl[p] = (unsigned short)(p | (1 << 10));  // final: chunks of one
          VQS_WAVE_SYNC();
          heaped = true;
          break;
        }
        depth--;
        {  // __move_median_to_first(first, first + 1, mid, last - 1)
          const int mid = lo + (hi - lo) / 2, A = lo + 1, B = mid, C = hi - 1;
          const KT ka_ = K[A], kb_ = K[B], kc_ = K[C];
          int pick;
          if (ka_ < kb_) {
            if (kb_ < kc_)
              pick = B;
            else if (ka_ < kc_)
              pick = C;
            else
              pick = A;
          } else if (ka_ < kc_)
            pick = A;
          else if (kb_ < kc_)
            pick = C;
          else
            pick = B;
          VQS_WAVE_SYNC();
          if (lane == 0) {
            const KT tk = K[lo];
            K[lo] = K[pick];
            K[pick] = tk;
            const unsigned short tp = Pm[lo];
            Pm[lo] = Pm[pick];
            Pm[pick] = tp;
          }
          VQS_WAVE_SYNC();
        }
        // __unguarded_partition(first + 1, last, first)
        const KT pv = K[lo];
        const int cntn = hi - lo - 1;
        int cntL = 0, cntR = 0;
        for (int r0 = 0; r0 < cntn; r0 += 64) {
          const int t = r0 + lane;
          const bool valid = t < cntn;
          const int iL = lo + 1 + t, iR = hi - 1 - t;
          const bool geL = valid && !(K[valid ? iL : lo] < pv);
          const bool leR = valid && !(pv < K[valid ? iR : lo]);
          const u64 bL = __ballot(geL), bR = __ballot(leR);
          if (geL && cntL + (int)__popcll(bL & lt) < 512) l0[cntL + __popcll(bL & lt)] = (unsigned short)iL;
          if (leR && cntR + (int)__popcll(bR & lt) < 512) l1[cntR + __popcll(bR & lt)] = (unsigned short)iR;
          cntL += __popcll(bL);
          cntR += __popcll(bR);
        }
        VQS_WAVE_SYNC();
        const int nmin = min(cntL, cntR);
        int ksw = 0;
        for (int r0 = 0; r0 < nmin; r0 += 64) {
          const int k = r0 + lane;
          const bool ok = k < nmin && l0[k] < l1[k];
          const u64 bb = __ballot(ok);
          ksw += __popcll(bb);
          if (bb != ~0ull) break;
        }
        int cut = hi;
        if (ksw > 0) cut = l1[ksw - 1];
        if (ksw < cntL) cut = min(cut, (int)l0[ksw]);
        cut = __builtin_amdgcn_readfirstlane(cut);
        for (int r0 = 0; r0 < ksw; r0 += 64) {
          const int k = r0 + lane;
          if (k < ksw) {
            const int i = l0[k], j = l1[k];
            const KT tk = K[i];
            K[i] = K[j];
            K[j] = tk;
            const unsigned short tp = Pm[i];
            Pm[i] = Pm[j];
            Pm[j] = tp;
          }
        }
        VQS_WAVE_SYNC();
        if (lane == sp) stk = stk_pack(cut, hi, depth);
        sp++;
        VQS_WAVE_SYNC();
        hi = cut;
      }
      if (!heaped) {
        const int len = hi - lo;  // <= 16: one chunk of the final insertion sort
        if (lane < len) cl[lo + lane] = (unsigned short)(lo | (len << 10));
        VQS_WAVE_SYNC();
      }
    }
    // __final_insertion_sort = stable sort of every chunk; then the permutation is applied to the point indices
    unsigned myv[16];
    unsigned short dst[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int p = r * 64 + lane;
      myv[r] = 0;
      dst[r] = 0;
      if (p < n) {
        const int cinfo = cl[p], c0 = cinfo & 1023, len = cinfo >> 10;
        const KT kp = K[p];
        int rank = 0;
        for (int q = c0; q < c0 + len; q++) {
          const KT kq = K[q];
          rank += (kq < kp || (kq == kp && q < p)) ? 1 : 0;
        }
        dst[r] = (unsigned short)(c0 + rank);
        myv[r] = va[b + Pm[p]];
      }
    }
    __threadfence_block();
    VQS_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int p = r * 64 + lane;
      if (p < n) {
        const KT kp = K[p];
        ka[b + dst[r]] = kNarrow ? (kp == (KT)0xffffffffull ? ~0ull : (u64)kp) : (u64)kp;
        va[b + dst[r]] = myv[r];
      }
    }
    VQS_WAVE_SYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// k_voxel_qsort_heap<KT>: the ranges whose introsort recursion ran into libstdc++'s depth limit (2 lg n partitions deep):
// std::__partial_sort(first, last, last), i.e. heap sort, one wave per range (heap_sort_wave).  Such a range is final
// afterwards (the final insertion sort moves nothing in it).
// ------------------------------------------------------------------------------------------------
template <typename KT>
struct HeapLds {
  KT K[4][1024];
  unsigned short Pm[4][1024], Rk[4][1024];
};

template <typename KT>
__global__ __launch_bounds__(256) void k_voxel_qsort_heap(u64* keys_all, unsigned* vals_all, const int* __restrict__ kinfo,
                                                          const unsigned* __restrict__ heap_all, const int* __restrict__ nheap,
                                                          int heap_cap, int P, int only, const float4* __restrict__ pts_even,
                                                          const float4* __restrict__ pts_odd, int stride_pts) {
  // pts_even / pts_odd: the input clouds of the even / odd slots as k_voxel_reduce reads them (null: no shortcut, the exact
  // permutation in every case -- the test hook)
  constexpr bool kNarrow = sizeof(KT) == 4;
  __shared__ HeapLds<KT> S;
  const int c = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (only >= 0 && (c & 1) != only) return;
  if ((kinfo[8 * c + 5] <= 31) != kNarrow) return;
  const int nh = nheap[c];
  u64* ka = keys_all + (size_t)c * P;
  unsigned* va = vals_all + (size_t)c * P;
  const unsigned* heap = heap_all + (size_t)c * heap_cap;
  KT* K = S.K[wave];
  unsigned short* Pm = S.Pm[wave];
  for (int l = blockIdx.x * 4 + wave; l < nh; l += gridDim.x * 4) {
    const int b = __builtin_amdgcn_readfirstlane((int)heap[2 * l]), n = __builtin_amdgcn_readfirstlane((int)heap[2 * l + 1]) - b;
    for (int p = lane; p < n; p += 64) {
      const u64 k = ka[b + p];
      K[p] = kNarrow ? (KT)(k == ~0ull ? 0xffffffffull : k) : (KT)k;
      Pm[p] = (unsigned short)p;
    }
    VQS_WAVE_SYNC();
    // A heap sort is one long serial chain (~1 us per element here), and these ranges come from the far, sparsely sampled parts
    // of a depth image where nearly every voxel holds one point.  When all keys of the range are distinct the result of ANY
    // correct sort is the reference's: rank every element by counting smaller keys (all lanes in parallel), and replay
    // libstdc++'s heap sort only if two elements collide on a rank (equal keys: their order is the heap's).
    bool replay = false, partial = false;
    int keep = 1;
    unsigned short* Rk = S.Rk[wave];
    {
      KT mykey[16];
      int rank[16];
      const int rows = (n + 63) >> 6;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        mykey[r] = r * 64 + lane < n ? K[r * 64 + lane] : (KT)0;
        rank[r] = 0;
      }
      // (keys are read eight at a time — same addresses in every lane, LDS broadcasts — so that one LDS round trip feeds 128
      // comparisons; the tail of the last group is padded with the largest key, which is smaller than nothing)
      const int n8 = (n + 7) & ~7;
      for (int p = n + lane; p < n8; p += 64) K[p] = ~(KT)0;
      VQS_WAVE_SYNC();
      for (int j = 0; j < n8; j += 8) {
        KT kj[8];
#pragma unroll
        for (int u = 0; u < 8; u++) kj[u] = K[j + u];
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (r < rows) {
#pragma unroll
            for (int u = 0; u < 8; u++) rank[r] += kj[u] < mykey[r] ? 1 : 0;
          }
      }
      VQS_WAVE_SYNC();
      unsigned short* slot = Pm;  // Pm is the identity so far: reuse it as the rendezvous (rank -> who claims it)
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (r * 64 + lane < n) slot[rank[r]] = (unsigned short)(r * 64 + lane);
      VQS_WAVE_SYNC();
      bool clash = false;
#pragma unroll
      for (int r = 0; r < 16; r++)
        if (r * 64 + lane < n) clash |= slot[rank[r]] != (unsigned short)(r * 64 + lane);
      replay = __ballot(clash) != 0ull;
      VQS_WAVE_SYNC();
      if (!replay) {  // slot[] = Pm[] is already the sorting permutation; the keys follow it
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (r * 64 + lane < n) K[rank[r]] = mykey[r];
      }
      if (replay && (pts_even || pts_odd)) {
        // Equal keys exist (voxels with several points).  What their order decides downstream (k_voxel_reduce,
        // util/downsampling_omp.hpp:63-90) is (a) which points of a voxel fall on either side of a 1024-element block cut and (b) the
        // order in which a voxel's coordinates are added up in double precision.  If no tied voxel of this range straddles a cut,
        // and the sums are exact whatever the order -- the coordinates are floats: a sum of up to 64 of them is exact in a double
        // as long as their binary exponents span less than 23 -- every order gives the reference's means bit for bit, and the serial
        // replay of the heap (~0.5 ms on one wave, with the rest of the chip waiting) is not needed: ties are ordered by position.
        const float4* in = ((c & 1) ? pts_odd : pts_even);
        int eqb[16], le[16];
#pragma unroll
        for (int r = 0; r < 16; r++) eqb[r] = le[r] = 0;
        for (int j = 0; j < n8; j += 8) {
          KT kj[8];
#pragma unroll
          for (int u = 0; u < 8; u++) kj[u] = K[j + u];
#pragma unroll
          for (int r = 0; r < 16; r++)
            if (r < rows) {
#pragma unroll
              for (int u = 0; u < 8; u++) {
                le[r] += kj[u] <= mykey[r] ? 1 : 0;
                eqb[r] += (kj[u] == mykey[r] && j + u < r * 64 + lane) ? 1 : 0;
              }
            }
        }
        bool bad = in == nullptr;
        int below_bad = n;  // smallest rank among the voxels whose order does matter (they straddle a block cut)
        int emin = 0x7fffffff, emax = -0x7fffffff;
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (r * 64 + lane < n) {
            const int g = le[r] - rank[r];  // points of this element's voxel in the range (the padding keys are larger than any key)
            if (g > 1) {
              const int first = b + rank[r];
              const bool matters = g > 64 || (first >> 10) != ((first + g - 1) >> 10);
              bad = bad || matters;
              if (matters) below_bad = min(below_bad, rank[r]);
              if (in) {
                const float4 pt = in[(size_t)(c >> 1) * stride_pts + va[b + r * 64 + lane]];
                const float co[3] = {pt.x, pt.y, pt.z};
#pragma unroll
                for (int a = 0; a < 3; a++) {
                  const unsigned bits = __float_as_uint(co[a]) & 0x7fffffffu;
                  if (bits == 0) continue;                     // zeros add exactly
                  if (bits >= 0x7f800000u) bad = true;          // inf / nan: leave it to the replay
                  const int e = max((int)(bits >> 23), 1);      // (subnormals share the smallest exponent)
                  emin = min(emin, e);
                  emax = max(emax, e);
                }
              }
            }
          }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) {
          emin = min(emin, __shfl_xor(emin, ofs, 64));
          emax = max(emax, __shfl_xor(emax, ofs, 64));
          below_bad = min(below_bad, __shfl_xor(below_bad, ofs, 64));
        }
        // (one exponent window for all three coordinates and all tied voxels of the range: coarser than needed, and still passed
        // by every depth-camera cloud -- |x|, |y| >= half a pixel's footprint, z >= the sensor's near limit)
        const bool exact_sums = in != nullptr && (emax < emin || emax - emin < 23);
        const bool harmless = __ballot(bad) == 0ull && exact_sums;
        if (!harmless && exact_sums && below_bad < n) {
          // Some tied voxel does straddle a cut: its order is the heap's, and so is the order of everything popped before it (the
          // pops come out in descending key order).  But the replay can stop once that voxel has been popped: the ties left in the
          // heap are of the harmless kind and take the places rank + position among their equals.
          partial = true;
          keep = below_bad;
          VQS_WAVE_SYNC();
#pragma unroll
          for (int r = 0; r < 16; r++)
            if (r * 64 + lane < n) Rk[r * 64 + lane] = (unsigned short)(rank[r] + eqb[r]);
          for (int p = lane; p < n; p += 64) Pm[p] = (unsigned short)p;
          VQS_WAVE_SYNC();
        }
        if (harmless) {
          VQS_WAVE_SYNC();
#pragma unroll
          for (int r = 0; r < 16; r++)
            if (r * 64 + lane < n) {
              K[rank[r] + eqb[r]] = mykey[r];
              slot[rank[r] + eqb[r]] = (unsigned short)(r * 64 + lane);
            }
          replay = false;
          VQS_WAVE_SYNC();
        }
      }
      if (replay && !partial) {
        // Equal keys exist: their order is the heap's.  The pops come out in descending key order, so once every key >= the
        // smallest tied key has been popped, what is left in the heap are distinct keys whose places are their ranks: only the
        // first n - (number of keys below the smallest tied key) pops are replayed.
        int below = n;  // rank of the smallest tied key = number of keys below it
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const bool cl = r * 64 + lane < n && slot[rank[r]] != (unsigned short)(r * 64 + lane);
          below = min(below, cl ? rank[r] : n);
        }
#pragma unroll
        for (int ofs = 32; ofs > 0; ofs >>= 1) below = min(below, __shfl_xor(below, ofs, 64));
        keep = below;
        VQS_WAVE_SYNC();
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (r * 64 + lane < n) Rk[r * 64 + lane] = (unsigned short)rank[r];
        for (int p = lane; p < n; p += 64) Pm[p] = (unsigned short)p;
      }
      VQS_WAVE_SYNC();
    }
#ifdef GFS_VQS_HEAP_STATS
    if (lane == 0 && replay) printf("VQSH c=%d l=%d nh=%d b=%d n=%d replay=%d partial=%d keep=%d\n", c, l, nh, b, n, (int)replay, (int)partial, keep);
#endif
    if (replay) {
      heap_sort_wave<KT>(K, Pm, 0, n, keep);
      if (keep > 1) {  // the `keep` smallest keys (all distinct) are still in heap order in [0, keep): each to its rank
        KT kh[16];
        unsigned short ph[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int hpos = r * 64 + lane;
          kh[r] = hpos < keep ? K[hpos] : (KT)0;
          ph[r] = hpos < keep ? Pm[hpos] : (unsigned short)0;
        }
        VQS_WAVE_SYNC();
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (r * 64 + lane < keep) {
            const int dest = Rk[ph[r]];
            K[dest] = kh[r];
            Pm[dest] = ph[r];
          }
        VQS_WAVE_SYNC();
      }
    }
    unsigned myv[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int p = r * 64 + lane;
      myv[r] = p < n ? va[b + Pm[p]] : 0u;
    }
    __threadfence_block();
    VQS_WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int p = r * 64 + lane;
      if (p < n) {
        const KT kp = K[p];
        ka[b + p] = kNarrow ? (kp == (KT)0xffffffffull ? ~0ull : (u64)kp) : (u64)kp;
        va[b + p] = myv[r];
      }
    }
    VQS_WAVE_SYNC();
  }
}

}  // namespace vqs
