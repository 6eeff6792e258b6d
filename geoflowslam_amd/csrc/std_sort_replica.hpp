// Replica of libstdc++'s std::sort (GCC 11 bits/stl_algo.h: __introsort_loop + __final_insertion_sort, heap fallback)
// as one host/device function.  DistributeOctTree sorts (size, node) pairs with a comparator that leaves many elements
// equivalent (reference src/ORBextractor.cc:552-565, 697-698), so the reference's result depends on the exact sequence
// of comparisons and moves libstdc++ performs; the device quadtree must reproduce it.  `less(a, b)` compares two
// elements; elements are moved as whole values.  tests/test_host_logic.py checks this against std::sort.
#pragma once

#if defined(__HIPCC__)
#define GFS_HD __host__ __device__
#else
#define GFS_HD
#endif

namespace gfs {

template <typename T, typename Less>
GFS_HD inline void replica_adjust_heap(T* first, long holeIndex, long len, T value, Less less) {
  const long topIndex = holeIndex;
  long secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (less(first[secondChild], first[secondChild - 1])) secondChild--;
    first[holeIndex] = first[secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    first[holeIndex] = first[secondChild - 1];
    holeIndex = secondChild - 1;
  }
  long parent = (holeIndex - 1) / 2;  // __push_heap
  while (holeIndex > topIndex && less(first[parent], value)) {
    first[holeIndex] = first[parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  first[holeIndex] = value;
}

template <typename T, typename Less>
GFS_HD inline void replica_heap_sort(T* first, T* last, Less less) {  // __partial_sort(first, last, last)
  const long len = last - first;
  if (len >= 2) {  // __make_heap
    long parent = (len - 2) / 2;
    while (true) {
      T value = first[parent];
      replica_adjust_heap(first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  while (last - first > 1) {  // __sort_heap / __pop_heap
    --last;
    T value = *last;
    *last = *first;
    replica_adjust_heap(first, 0L, (long)(last - first), value, less);
  }
}

template <typename T, typename Less>
GFS_HD inline void replica_unguarded_linear_insert(T* last, Less less) {
  T val = *last;
  T* next = last - 1;
  while (less(val, *next)) {
    *last = *next;
    last = next;
    --next;
  }
  *last = val;
}

template <typename T, typename Less>
GFS_HD inline void replica_insertion_sort(T* first, T* last, Less less) {
  if (first == last) return;
  for (T* i = first + 1; i != last; ++i) {
    if (less(*i, *first)) {
      T val = *i;
      for (T* p = i; p != first; --p) *p = *(p - 1);  // move_backward(first, i, i + 1)
      *first = val;
    } else {
      replica_unguarded_linear_insert(i, less);
    }
  }
}

// the pending right-hand parts of __introsort_loop: at most 2 lg(n) + 1 at a time (every level of the depth limit leaves one)
struct SortFrame {
  int lo, hi, depth;
};

// stack: caller-provided storage for the recursion (on the device: LDS, a private array would live in scratch memory)
template <typename T, typename Less>
GFS_HD inline void replica_std_sort_on(T* first, T* last, Less less, SortFrame* stack, int stack_cap) {
  const long n = last - first;
  if (n <= 0) return;
  constexpr long kThreshold = 16;
  long lg = 0;
  for (long v = n; v > 1; v >>= 1) lg++;
  // __introsort_loop with an explicit stack for the recursion on the right part
  int sp = 0;
  stack[sp++] = SortFrame{0, (int)n, (int)(lg * 2)};
  while (sp > 0) {
    const SortFrame f = stack[--sp];
    T* lo = first + f.lo;
    T* hi = first + f.hi;
    long depth = f.depth;
    while (hi - lo > kThreshold) {
      if (depth == 0) {
        replica_heap_sort(lo, hi, less);
        hi = lo;  // return
        break;
      }
      --depth;
      // __unguarded_partition_pivot
      T* mid = lo + (hi - lo) / 2;
      {
        T* a = lo + 1;
        T* b = mid;
        T* c = hi - 1;
        T* pick;
        if (less(*a, *b)) {
          if (less(*b, *c))
            pick = b;
          else if (less(*a, *c))
            pick = c;
          else
            pick = a;
        } else if (less(*a, *c))
          pick = a;
        else if (less(*b, *c))
          pick = c;
        else
          pick = b;
        T tmp = *lo;
        *lo = *pick;
        *pick = tmp;
      }
      T* pf = lo + 1;
      T* pl = hi;
      while (true) {
        while (less(*pf, *lo)) ++pf;
        --pl;
        while (less(*lo, *pl)) --pl;
        if (!(pf < pl)) break;
        T tmp = *pf;
        *pf = *pl;
        *pl = tmp;
        ++pf;
      }
      T* cut = pf;
      // recurse on [cut, hi) first (the reference recursion runs to completion before the loop continues on
      // [lo, cut)); the two ranges are disjoint, so running the left loop first and the right part later from
      // the stack performs the same comparisons and moves inside each range.
      if (sp < stack_cap) stack[sp++] = SortFrame{(int)(cut - first), (int)(hi - first), (int)depth};
      hi = cut;
    }
  }
  // __final_insertion_sort
  if (n > kThreshold) {
    replica_insertion_sort(first, first + kThreshold, less);
    for (T* i = first + kThreshold; i != last; ++i) replica_unguarded_linear_insert(i, less);
  } else {
    replica_insertion_sort(first, last, less);
  }
}

template <typename T, typename Less>
GFS_HD inline void replica_std_sort(T* first, T* last, Less less) {
  SortFrame stack[96];
  replica_std_sort_on(first, last, less, stack, 96);
}

}  // namespace gfs
