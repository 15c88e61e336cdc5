// stdsort_emu.h — where libstdc++'s std::sort leaves elements with EQUAL keys, computed by one wavefront.
//
// The reference sorts every ring sector's index list with std::sort and a comparator on the curvature only
// (laserOdometry.cpp:185); std::sort is not stable, so which of two points with the same curvature is picked first depends on
// libstdc++'s introsort (bits/stl_algo.h: __introsort_loop with median-of-three pivots and an unguarded Hoare partition down
// to 16-element pieces, heap sort below the depth limit 2 floor(log2 n), then __final_insertion_sort).  The insertion sort
// is stable with respect to the arrangement it starts from (it only moves an element past strictly larger ones), so the final
// order of equal keys is their order in the arrangement the partition phase leaves behind.  That arrangement is all this
// header computes (alego_params.sort_mode = 2).
//
// The partition phase is emulated step by step, but each Hoare partition runs wavefront-parallel: with pivot p, the k-th swap
// exchanges the k-th position from the left holding a key >= p with the k-th position from the right holding a key <= p, for
// as long as the former lies left of the latter (both pointers only ever cross untouched elements before they meet) — two
// ballot compactions, one count, one parallel swap.  The cut is the left pointer's final position.  The heap sort of a piece
// that ran out of depth is sequential (one lane); it needs adversarial input to be reached.
//
// tests/test_gpu_parity.py::test_device_std_sort_arrangement pins the result against std::sort itself (the test side is compiled
// against this container's libstdc++).
#ifndef ALEGO_STDSORT_EMU_H_
#define ALEGO_STDSORT_EMU_H_
#include "dev_common.h"

template <int NMAX>
struct SortEmu {
  uint32_t ak[NMAX];   // key at arrangement position
  uint16_t ai[NMAX];   // element (index into the input sequence) at arrangement position
  uint16_t lp[NMAX];   // positions with key >= pivot, ascending
  uint16_t rp[NMAX];   // positions with key <= pivot, descending
  uint16_t pos[NMAX];  // result: arrangement position of element i
  int stack[48][3];    // pending (first, last, depth_limit) of __introsort_loop's recursion
};

template <int NMAX>
DEV_INLINE void emu_adjust_heap(SortEmu<NMAX>& E, int base, int hole, int len, uint32_t vk, uint16_t vi) {   // std::__adjust_heap + __push_heap
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (E.ak[base + child] < E.ak[base + child - 1]) --child;
    E.ak[base + hole] = E.ak[base + child]; E.ai[base + hole] = E.ai[base + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    E.ak[base + hole] = E.ak[base + child - 1]; E.ai[base + hole] = E.ai[base + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && E.ak[base + parent] < vk) {
    E.ak[base + hole] = E.ak[base + parent]; E.ai[base + hole] = E.ai[base + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  E.ak[base + hole] = vk; E.ai[base + hole] = vi;
}

template <int NMAX>
DEV_INLINE void emu_heap_sort(SortEmu<NMAX>& E, int first, int last) {   // std::__partial_sort(first, last, last): __make_heap, __sort_heap
  const int len = last - first;
  if (len >= 2) {
    for (int parent = (len - 2) / 2;; --parent) {
      emu_adjust_heap(E, first, parent, len, E.ak[first + parent], E.ai[first + parent]);
      if (parent == 0) break;
    }
  }
  while (last - first > 1) {
    --last;
    const uint32_t vk = E.ak[last];
    const uint16_t vi = E.ai[last];
    E.ak[last] = E.ak[first]; E.ai[last] = E.ai[first];
    emu_adjust_heap(E, first, 0, last - first, vk, vi);
  }
}

// One wavefront (a 64-thread workgroup; every lane calls).  E.ak[0..n) holds the keys in input order on entry; on exit
// E.pos[i] = position of input element i in the arrangement std::sort's partition phase produces (the final sorted order is the
// stable sort of that arrangement by key).
template <int NMAX>
DEV_INLINE void stdsort_arrangement(SortEmu<NMAX>& E, int n, int depth_limit = -1 /* < 0: std::sort's own 2 floor(log2 n) */) {
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int i = lane; i < n; i += 64) E.ai[i] = (uint16_t)i;
  if (lane == 0) { E.stack[0][0] = 0; E.stack[0][1] = n; E.stack[0][2] = depth_limit >= 0 ? depth_limit : (n > 0 ? 2 * (31 - __clz(n)) : 0); }
  __syncthreads();
  int sp = 1;
  while (sp > 0) {
    --sp;
    // (wavefront-uniform values: kept in scalar registers)
    int first = __builtin_amdgcn_readfirstlane(E.stack[sp][0]), last = __builtin_amdgcn_readfirstlane(E.stack[sp][1]);
    int depth = __builtin_amdgcn_readfirstlane(E.stack[sp][2]);
    __syncthreads();
    while (last - first > 16) {   // _S_threshold
      if (depth == 0) {
        if (lane == 0) emu_heap_sort(E, first, last);
        __syncthreads();
        break;
      }
      --depth;
      // __move_median_to_first(first, first + 1, mid, last - 1)
      const int mid = first + (last - first) / 2;
      const uint32_t ka = E.ak[first + 1], kb = E.ak[mid], kc = E.ak[last - 1];
      int ch;
      if (ka < kb) { if (kb < kc) ch = mid; else if (ka < kc) ch = last - 1; else ch = first + 1; }
      else if (ka < kc) ch = first + 1;
      else if (kb < kc) ch = last - 1;
      else ch = mid;
      __syncthreads();
      if (lane == 0) {
        const uint32_t tk = E.ak[first]; const uint16_t ti = E.ai[first];
        E.ak[first] = E.ak[ch]; E.ai[first] = E.ai[ch];
        E.ak[ch] = tk; E.ai[ch] = ti;
      }
      __syncthreads();
      // __unguarded_partition(first + 1, last, pivot = *first)
      const uint32_t p = E.ak[first];
      int nL = 0, nR = 0;
      for (int x0 = first + 1; x0 < last; x0 += 64) {
        const int x = x0 + lane;
        const bool f = x < last && E.ak[min(x, last - 1)] >= p;
        const unsigned long long m = __ballot(f);
        if (f) E.lp[nL + (int)__popcll(m & below)] = (uint16_t)x;
        nL += (int)__popcll(m);
      }
      for (int x0 = last - 1; x0 > first; x0 -= 64) {
        const int x = x0 - lane;
        const bool f = x > first && E.ak[max(x, first + 1)] <= p;
        const unsigned long long m = __ballot(f);
        if (f) E.rp[nR + (int)__popcll(m & below)] = (uint16_t)x;
        nR += (int)__popcll(m);
      }
      __syncthreads();
      const int nmin = min(nL, nR);
      int m = 0;   // swaps: pairs whose left position is still left of the right one (a prefix of the pairs)
      for (int k0 = 0; k0 < nmin; k0 += 64) {
        const int k = k0 + lane;
        m += (int)__popcll(__ballot(k < nmin && E.lp[min(k, nmin - 1)] < E.rp[min(k, nmin - 1)]));
      }
      for (int k = lane; k < m; k += 64) {
        const int a = E.lp[k], b = E.rp[k];
        const uint32_t tk = E.ak[a]; const uint16_t ti = E.ai[a];
        E.ak[a] = E.ak[b]; E.ai[a] = E.ai[b];
        E.ak[b] = tk; E.ai[b] = ti;
      }
      // where the left pointer stops next: on the next untouched key >= p if that lies before the last swapped right
      // position, else on that position (it now holds a key >= p)
      int cut;
      if (m < nL && (m == 0 || E.lp[m] < E.rp[m - 1])) cut = E.lp[m];
      else cut = m > 0 ? (int)E.rp[m - 1] : last;   // (m == 0 && nL == 0 cannot happen: the median left a key >= p in the range)
      cut = __builtin_amdgcn_readfirstlane(cut);
      __syncthreads();
      if (lane == 0) { E.stack[sp][0] = cut; E.stack[sp][1] = last; E.stack[sp][2] = depth; }   // __introsort_loop(cut, last, depth_limit)
      ++sp;
      last = cut;
      __syncthreads();
    }
  }
  __syncthreads();
  for (int i = lane; i < n; i += 64) E.pos[E.ai[i]] = (uint16_t)i;
  __syncthreads();
}

#endif
