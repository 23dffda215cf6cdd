// ctd_tail.h -- device roles shared by the tail passes of the fast NCC path (ncc_fast.hip, argmax_rerank.hip).
#pragma once
#include "ctd_common.h"

namespace ctd {

typedef float tail_f32x4 __attribute__((ext_vector_type(4)));

// Order-preserving map f32 -> u32 (and back): a > b  <=>  f32_ordered(a) > f32_ordered(b), for every non-NaN pair.
__device__ inline unsigned f32_ordered(float x) {
  const unsigned b = __float_as_uint(x);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float f32_unordered(unsigned u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

// A listed window's exact score that clearly beats the pixel's ranked best is entered into the pixel's int64 INDEX
// WORD itself as (ordered score << 32 | ~d): plain indices have a zero high word, so any such patch key is larger, and
// a 64-bit atomic maximum keeps the best exact score with the lowest disparity among equals.  decode_role turns the
// word back into a plain index (and writes the score) once the fix-up kernel is complete.
__device__ inline unsigned long long patch_key(float val, int d) {
  return ((unsigned long long)f32_ordered(val) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)d);
}

// Second half of the run items: a workgroup per (frame, group of kRunPlanes disparity planes, share of the rows)
// copies, for every listed fully clamped pattern window (row h, from `run_rows`), the run values of the pixels
// w <= d - tail whose run has started (first disparity w + tail <= d) into the planes of its group -- (row, w) pairs
// are flattened over the threads so that every thread has independent loads in flight.
// bx = frame * ceil(D / kRunPlanes) + plane group, by / gy = this workgroup's share of the rows; s_rows: >= C * H ints of LDS.
constexpr int kRunPlanes = 8;

__device__ inline void runs_role(float* __restrict__ out, const float* __restrict__ run_vals,
                                 const unsigned* __restrict__ counters, const unsigned long long* __restrict__ run_rows,
                                 int per_frame, int C, int H, int W, int D, int bs, int bx, int by, int gy, int* s_rows) {
  __shared__ int s_n;
  const int tid = threadIdx.x;
  // a workgroup serves kRunPlanes consecutive disparity planes of one frame: the list scan and the loads of the run
  // values (the same for every plane, only the run gets longer) are paid once for all of them -- the pass is a chain
  // of dependent global round trips per workgroup, not bandwidth
  const int n_pg = (D + kRunPlanes - 1) / kRunPlanes;
  const int f = bx / n_pg, d0 = (bx - f * n_pg) * kRunPlanes;
  const int d1 = min(d0 + kRunPlanes, D) - 1;              // last plane of the group
  const int tail = bs - 1 - bs / 2;
  const int seg_max = min(d1 - tail + 1, W);               // plane d: pixels w in [0, d - tail]
  const unsigned n_r = counters[2];
  if (seg_max <= 0 || n_r == 0) return;                    // (workgroup-uniform)
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (unsigned j = tid; j < n_r; j += blockDim.x) {
    const unsigned long long e = run_rows[j];
    const int z = (int)(e >> 20), h = (int)(e & 0xFFFFF);
    // rows are dealt to the gy workgroups of a plane group by h (late planes carry ~D pixels per row)
    if ((!per_frame || z / C == f) && h % gy == by) s_rows[atomicAdd(&s_n, 1)] = h;
  }
  __syncthreads();
  const long HW = (long)H * W;
  // 32 lanes x 4 pixels span 128 pixels of a row, 8 rows per sweep of the workgroup: no index divisions, 16-byte
  // accesses wherever the quad lies inside the run and the row starts are 16-byte aligned
  const int wq = tid & 31, rs = tid >> 5;
  const bool vec_ok = (D % 4 == 0) && (W % 4 == 0) && (tail % 4 == 0);
  for (int w0 = 4 * wq; w0 < seg_max; w0 += 128) {
    const bool full_max = vec_ok && w0 + 3 < seg_max;
    for (int j0 = rs; j0 < s_n; j0 += 8 * 4) {
      tail_f32x4 v[4];
      int hh[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + 8 * u, s_n - 1);
        hh[u] = s_rows[j];
        const float* src = run_vals + ((long)f * H + hh[u]) * D + w0 + tail;
        if (full_max) {
          v[u] = *(const tail_f32x4*)src;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[u][i] = w0 + i < seg_max ? src[i] : __int_as_float(0x7fc00000);
        }
      }
      for (int d = d0; d <= d1; ++d) {
        const int seg = min(d - tail + 1, W);
        if (w0 >= seg) continue;
        const bool full = vec_ok && w0 + 3 < seg;
        float* plane = out + ((long)f * D + d) * HW;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (j0 + 8 * u >= s_n) continue;
          float* dst = plane + (long)hh[u] * W + w0;
          const bool all_set = v[u][0] == v[u][0] && v[u][1] == v[u][1] && v[u][2] == v[u][2] && v[u][3] == v[u][3];
          if (full && all_set) {
            *(tail_f32x4*)dst = v[u];
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (w0 + i < seg && v[u][i] == v[u][i]) dst[i] = v[u][i];
          }
        }
      }
    }
  }
}

// Patched index words -> plain indices and best scores.  One wavefront per (listed pattern window, frame), lane <->
// disparity: the pixels such a window can have patched are (f, h, x + d).  Pixels on the work list are left to the
// resolve pass (which rewrites idx and best itself).  `wave` / `n_waves`: this wavefront's number among the role's.
__device__ inline void decode_role(unsigned long long* __restrict__ idx, float* __restrict__ best,
                                   const unsigned char* __restrict__ flags, const unsigned* __restrict__ counters,
                                   const unsigned long long* __restrict__ list_a,
                                   const unsigned long long* __restrict__ list_b, int per_frame, int frames, int H, int W,
                                   int D, unsigned wave, unsigned n_waves) {
  const int lane = threadIdx.x & 63;
  // listed FRAME windows: every score of their pixel was a placeholder, so the pixel is normally on the work list
  // (all its keys tie); a pixel with a single score (D = 1, or w = 0 of a one-disparity run) is not -- decode it here
  const unsigned n_a = counters[3];               // (slot 0's value, copied by the ranked fix-up kernel; slot 0 itself is being cleared)
  for (unsigned j = wave * 64 + lane; j < n_a; j += n_waves * 64) {
    const unsigned long long e = list_a[j];
    const long pix = ((long)(e >> 40) * H + (long)((e >> 20) & 0xFFFFF)) * W + ((long)(e & 0xFFFFF) - 0x80000);
    const unsigned long long k = idx[pix];
    if ((k >> 32) != 0ull && flags[pix] == 0) {
      idx[pix] = (unsigned long long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
      best[pix] = f32_unordered((unsigned)(k >> 32));
    }
  }
  const unsigned n_b = counters[1];
  const unsigned per_b = per_frame ? 1u : (unsigned)frames;
  for (unsigned item = wave; item < n_b * per_b; item += n_waves) {
    const unsigned jb = item / per_b;
    const unsigned long long e = list_b[jb];
    const int z = (int)(e >> 40), h = (int)((e >> 20) & 0xFFFFF), col = (int)(e & 0xFFFFF) - 0x80000;
    const int f = per_frame ? z : (int)(item - jb * per_b);          // (ranked calls are single channel: z = frame)
    for (int d = lane; d < D; d += 64) {
      const int w = col + d;
      if (w < 0 || w >= W) continue;
      const long pix = ((long)f * H + h) * W + w;
      const unsigned long long k = idx[pix];
      if ((k >> 32) != 0ull && flags[pix] == 0) {
        idx[pix] = (unsigned long long)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        best[pix] = f32_unordered((unsigned)(k >> 32));
      }
    }
  }
}

}  // namespace ctd
