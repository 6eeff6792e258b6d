// One-wave replicas of libstdc++'s sorting pieces on (key, 16-bit payload) pairs in LDS, comparator = key only:
//   heap_sort_wave   std::__partial_sort(first, last, last) (= __make_heap + __sort_heap), the introsort depth-limit fallback
//   wave_std_sort    std::sort (GCC 11 bits/stl_algo.h: __introsort_loop + __final_insertion_sort) of up to 1024 elements
// Same comparisons and moves as the serial code, hence the same permutation where keys are equal (std::sort is not stable).  Used by
// the voxel sort of the GICP preprocessing (csrc/voxel_qsort.hpp: small_gicp's quick_sort_omp leaves, util/sort_omp.hpp:61) and by
// the quadtree of the ORB front end (csrc/orb.hip k_octree: the (size, x) list of ORBextractor.cc:697-698).
#pragma once

namespace vqs {

typedef unsigned long long u64;
constexpr int kWaveIntroThreshold = 16;  // libstdc++ _S_threshold

template <typename KT>
__device__ __forceinline__ void heap_adjust_soa(KT* K, unsigned short* Pm, int first, int hole, int len, KT vk,
                                                unsigned short vp) {  // libstdc++ __adjust_heap + __push_heap
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (K[first + child] < K[first + child - 1]) child--;
    K[first + hole] = K[first + child];
    Pm[first + hole] = Pm[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    K[first + hole] = K[first + child - 1];
    Pm[first + hole] = Pm[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && K[first + parent] < vk) {
    K[first + hole] = K[first + parent];
    Pm[first + hole] = Pm[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  K[first + hole] = vk;
  Pm[first + hole] = vp;
}

template <typename KT>
__device__ void heap_sort_soa(KT* K, unsigned short* Pm, int lo, int hi) {  // __partial_sort(first, last, last), one lane
  const int len = hi - lo;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      heap_adjust_soa<KT>(K, Pm, lo, parent, len, K[lo + parent], Pm[lo + parent]);
      if (parent == 0) break;
      parent--;
    }
  }
  int last = hi;
  while (last - lo > 1) {
    --last;
    const KT vk = K[last];
    const unsigned short vp = Pm[last];
    K[last] = K[lo];
    Pm[last] = Pm[lo];
    heap_adjust_soa<KT>(K, Pm, lo, 0, last - lo, vk, vp);
  }
}

#define VQS_WAVE_SYNC()                                  \
  do {                                                   \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                     \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

// Wave-cooperative __partial_sort(first, last, last) (= __make_heap + __sort_heap) on K / Pm [lo, hi) in LDS: the same
// comparisons and moves as libstdc++'s serial code, arranged so that a pop costs a few LDS round trips instead of ~3 per
// heap level.  __make_heap sifts the nodes in decreasing index order; nodes of one depth have disjoint subtrees, so a whole
// depth is done at once (one lane per node, each running the serial __adjust_heap).  A pop walks the hole from the root to a
// leaf along the larger children (bottom-up variant: no comparison with the value on the way down): the 63 lanes read the
// child pairs of the next SIX levels below the hole at once, the walk through them is scalar bit arithmetic on two ballots,
// and the lanes on the path write the chosen children up; __push_heap then climbs (rarely more than a level).
template <typename KT>
__device__ void heap_sort_wave(KT* K, unsigned short* Pm, int lo, int hi, int stop_len = 1) {
  const int lane = threadIdx.x & 63;
  const int len = hi - lo;
  KT* H = K + lo;
  unsigned short* Q = Pm + lo;
  if (len < 2) return;
  {
    const int last_parent = (len - 2) / 2;
    for (int d = 31 - __clz(last_parent + 1); d >= 0; d--) {
      const int first_node = (1 << d) - 1, end_node = min((2 << d) - 2, last_parent);
      for (int base = first_node; base <= end_node; base += 64) {
        const int node = base + lane;
        if (node <= end_node) heap_adjust_soa<KT>(K, Pm, lo, node, len, H[node], Q[node]);
      }
      VQS_WAVE_SYNC();
    }
  }
  const int t = lane + 1, dt = 31 - __clz(t), tofs = t - (1 << dt);  // local node t (1-based) of the 6-level subtree below the hole
  auto bcast = [](KT v, int from) {  // value of lane `from` (wave-uniform index)
    if constexpr (sizeof(KT) == 4) {
      return (KT)__builtin_amdgcn_readlane((int)v, from);
    } else {
      const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, from);
      const unsigned hi32 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), from);
      return (KT)(((u64)hi32 << 32) | lo32);
    }
  };
#ifdef VQS_NO_POPS
  return;
#endif
  // (stop_len > 1: only the pops down to a heap of stop_len elements are replayed — see k_voxel_qsort_heap)
  for (int L = len - 1; L >= max(1, stop_len); L--) {  // __pop_heap(first, first + L, first + L): value = H[L], H[L] = H[0], sift in [0, L)
    // (the popped value and the old root are not needed by the descent: no wait on these reads until the end of the pop)
    const KT vk = H[L], rk = H[0];
    const unsigned short vp = Q[L], rp = Q[0];
    const int half = (L - 1) / 2;  // nodes below `half` have two children
    int hole = 0;
    KT upk = 0;  // the key just moved into the parent of the hole: __push_heap's first comparison needs no LDS read
    unsigned short upp = 0;
    bool moved = false;
    while (hole < half) {
      const int g = ((hole + 1) << dt) + tofs - 1;
      const bool valid = lane < 63 && g < half;
      KT kl = 0, kr = 0;
      unsigned short pl = 0, pr = 0;
      if (valid) {
        kl = H[2 * g + 1];
        kr = H[2 * g + 2];
        pl = Q[2 * g + 1];
        pr = Q[2 * g + 2];
      }
      const bool left = valid && kr < kl;  // __adjust_heap: the second child unless it is smaller than the first
      // the walk through the six levels: scalar bit arithmetic on the two ballots, branch free (a finished walk idles)
      const u64 bv = __ballot(valid), bl = __ballot(left);
      int tt = 1, last_t = 1, alive = 1;
      u64 pathmask = 0;
#pragma unroll
      for (int lev = 0; lev < 6; lev++) {
        const int idx = tt - 1;
        const int ok = alive & (int)((bv >> idx) & 1ull);
        const int lf = (int)((bl >> idx) & 1ull);
        pathmask |= (u64)ok << idx;
        last_t = ok ? tt : last_t;
        tt = ok ? 2 * tt + 1 - lf : tt;
        alive = ok;
      }
      const KT ck = left ? kl : kr;
      const unsigned short cp = left ? pl : pr;
      if ((pathmask >> lane) & 1ull) {
        H[g] = ck;
        Q[g] = cp;
      }
      if (pathmask) {
        upk = bcast(ck, last_t - 1);
        upp = (unsigned short)__builtin_amdgcn_readlane((int)cp, last_t - 1);
        moved = true;
      }
      const int dtt = 31 - __clz(tt);
      hole = ((hole + 1) << dtt) + (tt - (1 << dtt)) - 1;
      VQS_WAVE_SYNC();
    }
    if ((L & 1) == 0 && hole == (L - 2) / 2) {  // a last node with a single child
      const int child = 2 * (hole + 1) - 1;
      const KT ck = H[child];
      const unsigned short cp = Q[child];
      VQS_WAVE_SYNC();
      if (lane == 0) {
        H[hole] = ck;
        Q[hole] = cp;
      }
      upk = ck;
      upp = cp;
      moved = true;
      hole = child;
      VQS_WAVE_SYNC();
    }
    if (lane == 0) {  // H[L] = H[0] of __pop_heap (node L is outside the sifted heap [0, L))
      H[L] = rk;
      Q[L] = rp;
    }
    bool first = moved;
    while (hole > 0) {  // __push_heap
      const int parent = (hole - 1) / 2;
      KT pk;
      unsigned short pp;
      if (first) {
        pk = upk;
        pp = upp;
        first = false;
      } else {
        pk = H[parent];
        pp = Q[parent];
      }
      if (!(pk < vk)) break;
      VQS_WAVE_SYNC();
      if (lane == 0) {
        H[hole] = pk;
        Q[hole] = pp;
      }
      hole = parent;
      VQS_WAVE_SYNC();
    }
    if (lane == 0) {
      H[hole] = vk;
      Q[hole] = vp;
    }
    VQS_WAVE_SYNC();
  }
}


// std::sort of K[0, n) (n <= 1024) with the payloads Pm[0, n) following their keys: one wave, everything in LDS.  l0 / l1 / cl: n
// 16-bit entries of scratch each, st: 3 * 40 entries (the pending [cut, last) parts of __introsort_loop).  On return K / Pm hold
// the sorted sequence.  Steps (as in k_voxel_qsort_leaf, which keeps its own copy with the <= 64-element register path and the
// deferral of heap sorts to another kernel):
//   __introsort_loop     while a part has more than 16 elements: depth limit -> heap sort; else median of three to the front,
//                        __unguarded_partition — left stoppers (key >= pivot, ascending) and right stoppers (key <= pivot,
//                        descending) are listed with ballots, pair k is swapped while L_k < R_k, cut = min(L_K, R_{K-1}) —, the
//                        right part goes on the stack, the left part is continued;
//   __final_insertion_sort = a stable sort of each of the <= 16-element parts the loop leaves: position = start + rank of
//                        (key, position) inside the part.
template <typename KT>
__device__ void wave_std_sort(KT* K, unsigned short* Pm, unsigned short* l0, unsigned short* l1, unsigned short* cl, unsigned short* st,
                              int n) {
  const int lane = threadIdx.x & 63;
  const u64 lt = (1ull << lane) - 1ull;
  if (n < 2) return;
  int lg = 0;
  for (int v = n; v > 1; v >>= 1) lg++;
  if (lane == 0) {
    st[0] = 0;
    st[1] = (unsigned short)n;
    st[2] = (unsigned short)(2 * lg);
  }
  int sp = 1;
  VQS_WAVE_SYNC();
  while (sp > 0) {
    sp--;
    int lo = __builtin_amdgcn_readfirstlane((int)st[3 * sp]), hi = __builtin_amdgcn_readfirstlane((int)st[3 * sp + 1]),
        depth = __builtin_amdgcn_readfirstlane((int)st[3 * sp + 2]);
    bool marked = false;
    while (hi - lo > kWaveIntroThreshold) {
      if (depth == 0) {  // std::__partial_sort(first, last, last); the range is final: parts of one element below
        heap_sort_wave<KT>(K, Pm, lo, hi);
        for (int p = lo + lane; p < hi; p += 64) cl[p] = (unsigned short)(p | (1 << 10));
        VQS_WAVE_SYNC();
        marked = true;
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
        if (geL) l0[cntL + __popcll(bL & lt)] = (unsigned short)iL;
        if (leR) l1[cntR + __popcll(bR & lt)] = (unsigned short)iR;
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
      if (lane == 0) {
        st[3 * sp] = (unsigned short)cut;
        st[3 * sp + 1] = (unsigned short)hi;
        st[3 * sp + 2] = (unsigned short)depth;
      }
      sp++;
      VQS_WAVE_SYNC();
      hi = cut;
    }
    if (!marked) {
      const int len = hi - lo;  // <= 16: one part of the final insertion sort
      if (lane < len) cl[lo + lane] = (unsigned short)(lo | (len << 10));
      VQS_WAVE_SYNC();
    }
  }
  // __final_insertion_sort.  (n > 16: __insertion_sort of the first 16 elements, __unguarded_insertion_sort of the rest — an
  // insertion sort is stable and after the loop above no element has to cross a part boundary, so it is the stable sort of
  // every part; n <= 16 is one part.)
  KT myk[16];
  unsigned short myp[16], dst[16];
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int p = r * 64 + lane;
    myk[r] = 0;
    myp[r] = 0;
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
      myk[r] = kp;
      myp[r] = Pm[p];
    }
  }
  VQS_WAVE_SYNC();
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int p = r * 64 + lane;
    if (p < n) {
      K[dst[r]] = myk[r];
      Pm[dst[r]] = myp[r];
    }
  }
  VQS_WAVE_SYNC();
}

}  // namespace vqs
