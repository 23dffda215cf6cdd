// ctd_rank.h -- device helpers shared by the passes of the ranked fast argmax (ncc_fast.hip, argmax_rerank.hip).
#pragma once
#include "ctd_common.h"

namespace ctd {

// Margin inside which two scores count as tied for the exact re-scoring: the caller's eps plus the resolution of two
// keys of the in-kernel ranking (fixed point, 2^-21 = 4.8e-7 absolute each way, ncc_fast.hip).  The all-D kernel and the
// fix-up check use this one expression.
__device__ inline float rank_margin(float eps, float top) {
  (void)top;
  return eps + 1e-6f;
}

// Work list of the pixels the exact re-scoring has to settle.  Atomics on ONE address retire one every ~7 ns on this
// part -- two thousand list appends of the fix-up pass were a third of its time -- so the list comes in `parts`
// segments with a counter each (a cache line apart), keyed by the pixel's row: appends of different rows go to
// different counters.  A segment holds at most the pixels of the rows with its key (every pixel is listed at most once).
__device__ inline int worklist_key(const WorkList& wl, long pix) {
  return wl.parts == 1 ? 0 : (int)((pix / wl.row_width) & (wl.parts - 1));
}

// Appends `item` for the lanes with `take` set: one atomic per wavefront and distinct key (thousands of lanes bumping a
// counter one by one cost the merge kernel as much as its memory traffic).  Every lane of the wavefront that is still
// running must call it.
__device__ inline void worklist_push(bool take, int64_t item, const WorkList& wl) {
  unsigned long long m = __ballot(take);
  if (m == 0) return;                                          // wave-uniform
  const int lane = threadIdx.x & 63;
  const int key = take ? worklist_key(wl, item) : -1;
  while (m) {                                                  // wave-uniform: the distinct keys present
    const int leader = __ffsll((long long)m) - 1;
    const int k = __shfl(key, leader);
    const unsigned long long mk = __ballot(take && key == k);
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(wl.counters + k * kWorkListStride, (unsigned)__popcll(mk));
    base = __shfl(base, leader);
    if (take && key == k) wl.list[(long)k * wl.seg_cap + base + __popcll(mk & ((1ull << lane) - 1ull))] = item;
    m &= ~mk;
  }
}

// Two candidate items per lane that share their key by construction (two pixels of one image row): ONE counter update
// per wavefront (a lone wavefront per fix-up item pays a full global round trip for every atomic it waits on).
__device__ inline void worklist_push2_same_row(bool take0, long item0, bool take1, long item1, const WorkList& wl) {
  const unsigned long long m0 = __ballot(take0), m1 = __ballot(take1);
  if ((m0 | m1) == 0) return;                                  // wave-uniform
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)(m0 ? m0 : m1)) - 1;
  const int k = __shfl(worklist_key(wl, m0 ? item0 : item1), leader);
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(wl.counters + k * kWorkListStride, (unsigned)(__popcll(m0) + __popcll(m1)));
  base = __shfl(base, 0);
  const unsigned long long below = (1ull << lane) - 1ull;
  if (take0) wl.list[(long)k * wl.seg_cap + base + __popcll(m0 & below)] = item0;
  if (take1) wl.list[(long)k * wl.seg_cap + base + __popcll(m0) + __popcll(m1 & below)] = item1;
}

// Claims pixel `pix` for the work list: true for exactly one caller per pixel (flag byte set atomically).
__device__ inline bool worklist_claim(unsigned* __restrict__ flags, long pix) {
  const unsigned bit = 1u << (8 * (unsigned)(pix & 3));
  return (atomicOr(flags + (pix >> 2), bit) & bit) == 0;
}

}  // namespace ctd
