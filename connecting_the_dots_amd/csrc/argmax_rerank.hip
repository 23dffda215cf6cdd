// argmax_rerank.hip -- argmax over disparity of a FAST (tolerance-level) NCC volume with the
// indices of the REFERENCE-ORDER volume.
//
// No reference code exists for the argmax; the contract is
// torch.argmax(xcorrvol_cpu(in0, in1, D, bs), 0): first index wins ties.  The fast volume
// differs from the reference-order one by <= ~1e-6, which can flip near-ties, so every
// disparity whose fast score lies within `eps` of the pixel's best fast score is re-scored
// with the reference's own operation order (XCorrVolFunctor, torchext/ext/ext.h:120-191,
// two passes over the window, no FMA) and the best exact score wins, lowest d first.
// (Along the left border, where the reference volume is exactly constant in d, only the
// lowest disparity of the constant run is a candidate.)  Two kernels: a coalesced sweep that
// finalises every pixel whose runner-up is farther than eps from its best, and a resolve pass
// in which a whole wavefront (lane <-> disparity) re-scores each remaining pixel.
#include "ctd_internal.h"

#ifndef CTD_RESOLVE_ABLATE
#define CTD_RESOLVE_ABLATE 0   // timing experiments only
#endif

namespace ctd {

// reference-order NCC of one (h, w, d), single channel, straight from global memory
__device__ static float ncc_exact_point(const float* __restrict__ a, const float* __restrict__ b, int H, int W,
                                        int h, int w, int d, int bs) {
  const int half = bs / 2;
  const float bs2 = (float)(bs * bs);
  float mu0 = 0.f, mu1 = 0.f;
  for (int bh = 0; bh < bs; ++bh) {
    const int hh = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      int w0 = w + bw - half;
      const int w1 = clampi(w0 - d, 0, W - 1);
      w0 = clampi(w0, 0, W - 1);
      mu0 += a[(long)hh * W + w0] / bs2;
      mu1 += b[(long)hh * W + w1] / bs2;
    }
  }
  float s0 = 0.f, s1 = 0.f, dot = 0.f;
  for (int bh = 0; bh < bs; ++bh) {
    const int hh = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      int w0 = w + bw - half;
      const int w1 = clampi(w0 - d, 0, W - 1);
      w0 = clampi(w0, 0, W - 1);
      const float v0 = a[(long)hh * W + w0] - mu0;
      const float v1 = b[(long)hh * W + w1] - mu1;
      dot += v0 * v1;
      s0 += v0 * v0;
      s1 += v1 * v1;
    }
  }
  const float norm = (float)((double)sqrtf(s0 * s1) + 1e-8);
  float val = 0.f;
  val += dot / norm;
  return val;
}


// reference-order NCC of one disparity from windows staged in LDS: sA[bs][bs] is the frame window,
// sB[bs][bs + D - 1] the pattern rows from column w - half - (D-1) on (replicate border baked in), so tap
// (bh, bw) of disparity d sits at sB[bh][bw + (D-1) - d].  Same operation order as ncc_exact_point.
__device__ static float ncc_exact_point_lds(const float* sA, const float* sB, int bs, int span, int off) {
  const float bs2 = (float)(bs * bs);
  float mu0 = 0.f, mu1 = 0.f;
  for (int bh = 0; bh < bs; ++bh)
    for (int bw = 0; bw < bs; ++bw) {
      mu0 += sA[bh * bs + bw] / bs2;
      mu1 += sB[bh * span + bw + off] / bs2;
    }
  float s0 = 0.f, s1 = 0.f, dot = 0.f;
  for (int bh = 0; bh < bs; ++bh)
    for (int bw = 0; bw < bs; ++bw) {
      const float v0 = sA[bh * bs + bw] - mu0;
      const float v1 = sB[bh * span + bw + off] - mu1;
      dot += v0 * v1;
      s0 += v0 * v0;
      s1 += v1 * v1;
    }
  const float norm = (float)((double)sqrtf(s0 * s1) + 1e-8);
  float val = 0.f;
  val += dot / norm;
  return val;
}

// Pass 1: one coalesced sweep over d, 4 adjacent pixels per thread (16-byte loads), branch-free tracking of
// the best score / index and of the runner-up score.  Pixels whose runner-up is within eps of the best are
// marked (idx = -1 - argmax) for pass 2; all others are final.
template <bool VEC4>
__global__ __launch_bounds__(256) void argmax_scan_kernel(const float* __restrict__ vol, int64_t* __restrict__ idx,
                                                          float* __restrict__ best, int D, long HW, int W,
                                                          int bs, float eps, long total_threads) {
  constexpr int PX = VEC4 ? 4 : 1;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_threads) return;
  const long p0 = t * PX;                         // first pixel (flat over frames); HW % 4 == 0 when VEC4
  const long f = p0 / HW, q0 = p0 - f * HW;
  const float* v = vol + f * D * HW + q0;
  // Past d_clamped = w + (bs-1-bs/2) every pattern tap clamps to column 0 (ext.h:152-154): the scores of
  // that run are one window's score repeated, so only its first element takes part in the ranking.
  const int w0 = (int)(q0 % W);                   // VEC4 implies W % 4 == 0: the 4 pixels share a row
  float v0[PX], v1[PX];
  int i0[PX], dc[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) { v0[k] = -INFINITY; v1[k] = -INFINITY; i0[k] = 0; dc[k] = w0 + k + (bs - 1 - bs / 2); }
#pragma unroll 8
  for (int d = 0; d < D; ++d) {
    float x[PX];
    if constexpr (VEC4) {
      const float4 q = *(const float4*)(v + (long)d * HW);
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else {
      x[0] = v[(long)d * HW];
    }
#pragma unroll
    for (int k = 0; k < PX; ++k) {
      const float y = d > dc[k] ? -INFINITY : x[k];
      v1[k] = fmaxf(v1[k], fminf(v0[k], y));      // runner-up = second largest seen so far
      i0[k] = y > v0[k] ? d : i0[k];              // strict >: first index wins ties
      v0[k] = fmaxf(v0[k], y);
    }
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const bool hard = v1[k] >= v0[k] - eps;
    idx[p0 + k] = hard ? (int64_t)(-1 - i0[k]) : (int64_t)i0[k];
    if (best) best[p0 + k] = v0[k];
  }
}

// Pass 2: every wavefront visits chunks of 64 index slots.  A chunk with marked pixels is re-swept over d
// with the same coalesced access as pass 1 (lane <-> pixel; per-lane strided reads of the volume would touch
// one 4 KB page per lane and disparity) to collect each marked pixel's candidate set as a bit mask: scores
// within eps of the pixel's best, the run of disparities whose window is clamped to column 0 counted once
// (lowest d).  A pixel with a single candidate is final; otherwise the whole wave resolves it: the frame
// window and the reachable pattern rows are staged in LDS, lane <-> disparity re-scores its candidates in
// reference order, and a wave reduction picks the best exact score, lowest d first.
constexpr int kMaskWords = 8;            // disparities per candidate mask = 512

__global__ __launch_bounds__(256) void argmax_resolve_kernel(const float* __restrict__ vol,
                                                             const float* __restrict__ in0,
                                                             const float* __restrict__ in1, long in1_frame_stride,
                                                             int64_t* __restrict__ idx, float* __restrict__ best,
                                                             int D, int H, int W, int bs, float eps, long total) {
  extern __shared__ float lds_resolve[];
  const int lane = threadIdx.x & 63;
  const int half = bs / 2, span = bs + D - 1;
  float* sA = lds_resolve + (threadIdx.x >> 6) * (bs * bs + bs * span);   // per-wave staging area
  float* sB = sA + bs * bs;
  const long HW = (long)H * W;
  // grid-stride over 256-pixel chunks (a few thousand workgroups instead of one tiny workgroup per chunk)
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p - lane < total; p += (long)gridDim.x * blockDim.x) {
    const bool mine = p < total && idx[p] < 0;
    unsigned long long todo = __ballot(mine);
    if (!todo) continue;
    const long pc = p < total ? p : total - 1;             // idle tail lanes shadow the last pixel
    const long f = pc / HW, q = pc - f * HW;
    const int w = (int)(q % W);
    const float* v = vol + f * D * HW + q;
    const int d_clamped = w + (bs - 1 - bs / 2);
    const float m = mine ? v[(-1 - idx[pc]) * HW] : INFINITY;
    unsigned long long mask[kMaskWords];
#pragma unroll
    for (int k = 0; k < kMaskWords; ++k) mask[k] = 0ull;
    bool have_clamped = false;
    int n_cand = 0, only = 0;
#pragma unroll
    for (int wd = 0; wd < kMaskWords; ++wd) {
      if (wd * 64 < D) {                                   // wave-uniform
        unsigned long long bits = 0ull;
        const int d_hi = min(D, wd * 64 + 64);
        // 16 independent loads in flight per batch: a lone wave must not pay one HBM round trip per disparity
        for (int d0 = wd * 64; d0 < d_hi; d0 += 16) {
          float x[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) x[k] = v[(long)min(d0 + k, D - 1) * HW];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int d = d0 + k;
            bool cand = d < d_hi && x[k] >= m - eps;
            if (cand && d >= d_clamped) {
              cand = !have_clamped;
              have_clamped = true;
            }
            if (cand) { bits |= 1ull << (d & 63); ++n_cand; only = d; }
          }
        }
        mask[wd] = bits;
      }
    }
    if (mine && n_cand == 1) {                             // nothing to compare against
      idx[p] = only;
      if (best) best[p] = v[(long)only * HW];
    }
    todo = __ballot(mine && n_cand > 1);
    while (todo) {
      const int j = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const long pj = p - lane + j;                        // wave-uniform
      const long fj = pj / HW, qj = pj - fj * HW;
      const int hj = (int)(qj / W), wj = (int)(qj - (long)hj * W);
      const float* a = in0 + fj * HW;
      const float* b = in1 + fj * in1_frame_stride;
      for (int i = lane; i < bs * bs; i += 64) {
        const int bh = i / bs, bw = i - bh * bs;
        sA[i] = a[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj + bw - half, 0, W - 1)];
      }
      for (int i0 = lane; i0 < bs * span; i0 += 64 * 8) {          // 8 independent loads in flight per lane
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = min(i0 + 64 * k, bs * span - 1);
          const int bh = i / span, c = i - bh * span;
          t[k] = b[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj - half - (D - 1) + c, 0, W - 1)];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (i0 + 64 * k < bs * span) sB[i0 + 64 * k] = t[k];
      }
      float eb = -INFINITY;
      int ei = 0x7fffffff;
#pragma unroll
      for (int wd = 0; wd < kMaskWords; ++wd) {
        if (wd * 64 < D) {
          const unsigned lo = __shfl((unsigned)(mask[wd] & 0xffffffffull), j);
          const unsigned hi = __shfl((unsigned)(mask[wd] >> 32), j);
          const unsigned long long mj = ((unsigned long long)hi << 32) | lo;
          if ((mj >> lane) & 1ull) {
            const int d = wd * 64 + lane;
#if CTD_RESOLVE_ABLATE >= 1
            const float e = sA[lane % 7] + sB[d];
#else
            const float e = ncc_exact_point_lds(sA, sB, bs, span, (D - 1) - d);
#endif
            if (e > eb || (e == eb && d < ei)) { eb = e; ei = d; }
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float oe = __shfl_xor(eb, off);
        const int oi = __shfl_xor(ei, off);
        if (oe > eb || (oe == eb && oi < ei)) { eb = oe; ei = oi; }
      }
      if (lane == 0) {
        idx[pj] = ei;
        if (best) best[pj] = vol[fj * D * HW + (long)ei * HW + qj];
      }
    }
  }
}

int argmax_rerank_f32(const float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                      float* best, int frames, int D, int H, int W, int bs, float eps, hipStream_t stream) {
  if (D > kMaskWords * 64) return CTD_ERR_UNSUPPORTED;
  const long total = (long)frames * H * W;
  const long HW = (long)H * W;
  const bool vec4 = W % 4 == 0 && ((uintptr_t)vol % 16) == 0;
  if (vec4)
    hipLaunchKernelGGL(argmax_scan_kernel<true>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, vol,
                       idx, best, D, HW, W, bs, eps, total / 4);
  else
    hipLaunchKernelGGL(argmax_scan_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, vol, idx,
                       best, D, HW, W, bs, eps, total);
  CTD_LAUNCH_CHECK();
  const size_t lds = sizeof(float) * 4 * ((size_t)bs * bs + (size_t)bs * (bs + D - 1));
  if (lds > 64 * 1024) return CTD_ERR_UNSUPPORTED;
  const long chunks = (total + 255) / 256;
  hipLaunchKernelGGL(argmax_resolve_kernel, dim3((unsigned)(chunks < 2048 ? chunks : 2048)), dim3(256), lds, stream, vol, in0,
                     in1, in1_frame_stride, idx, best, D, H, W, bs, eps, total);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
