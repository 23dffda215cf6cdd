// ctd_rank.h -- device helpers shared by the passes of the ranked fast argmax (ncc_fast.hip, argmax_rerank.hip).
#pragma once
#include "ctd_common.h"

namespace ctd {

// Margin inside which two scores count as tied for the exact re-scoring: the caller's eps plus the truncation of two
// keys (5 mantissa bits each: index tag + tie flag).  The volume kernel, the merge and the fix-up check use this one
// expression.
__device__ inline float rank_margin(float eps, float top) { return eps + 8e-6f * fmaxf(1.f, fabsf(top)); }

// Appends `item` to a work list for the lanes with `take` set: one atomic per wavefront (thousands of lanes bumping
// one counter one by one cost the merge kernel as much as its memory traffic).  Every lane of the wavefront that is
// still running must call it.
__device__ inline void worklist_push(bool take, int64_t item, unsigned* __restrict__ counter, int64_t* __restrict__ list) {
  const unsigned long long m = __ballot(take);
  if (m == 0) return;                                          // wave-uniform
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
  unsigned base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned)__popcll(m));
  base = __shfl(base, leader);
  if (take) list[base + __popcll(m & ((1ull << lane) - 1ull))] = item;
}

// Same for N candidate items per lane with ONE counter update per wavefront (a lone wavefront per fix-up item pays a
// full global round trip for every atomic it waits on).
template <int N>
__device__ inline void worklist_push_n(const bool (&take)[N], const long (&item)[N], unsigned* __restrict__ counter,
                                       int64_t* __restrict__ list) {
  const int lane = threadIdx.x & 63;
  unsigned long long m[N];
  unsigned total = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    m[i] = __ballot(take[i]);
    total += (unsigned)__popcll(m[i]);
  }
  if (total == 0) return;                                      // wave-uniform
  unsigned base = 0;
  if (lane == 0) base = atomicAdd(counter, total);
  base = __shfl(base, 0);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (take[i]) list[base + __popcll(m[i] & ((1ull << lane) - 1ull))] = item[i];
    base += (unsigned)__popcll(m[i]);
  }
}

// Claims pixel `pix` for the work list: true for exactly one caller per pixel (flag byte set atomically).
__device__ inline bool worklist_claim(unsigned* __restrict__ flags, long pix) {
  const unsigned bit = 1u << (8 * (unsigned)(pix & 3));
  return (atomicOr(flags + (pix >> 2), bit) & bit) == 0;
}

}  // namespace ctd
