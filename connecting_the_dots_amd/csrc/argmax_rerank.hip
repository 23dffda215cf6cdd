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
// lowest disparity of the constant run is a candidate.)  One pass over the volume: a thread per pixel keeps the running maximum and a bit mask of
// the disparities that were within eps of the running maximum when they were visited (a
// superset of the final candidates); pixels with more than one mask bit re-read just those.
#include "ctd_internal.h"

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

constexpr int kMaskWords = 8;   // up to 512 disparities in the candidate mask

__global__ __launch_bounds__(256) void argmax_rerank_kernel(const float* __restrict__ vol,
                                                            const float* __restrict__ in0,
                                                            const float* __restrict__ in1, long in1_frame_stride,
                                                            int64_t* __restrict__ idx, float* __restrict__ best,
                                                            int D, int H, int W, int bs, float eps, long total) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  const long HW = (long)H * W;
  const long f = p / HW, q = p - f * HW;
  const int h = (int)(q / W), w = (int)(q - (long)h * W);
  const float* v = vol + f * D * HW + q;

  unsigned long long mask[kMaskWords];
#pragma unroll
  for (int k = 0; k < kMaskWords; ++k) mask[k] = 0ull;
  float m = v[0];
  int mi = 0;
  mask[0] = 1ull;
  for (int d0 = 1; d0 < D; d0 += 8) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = (d0 + k < D) ? v[(long)(d0 + k) * HW] : -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int d = d0 + k;
      if (t[k] > m) { m = t[k]; mi = d; }
      if (t[k] >= m - eps) {                       // within eps of the running maximum (superset)
        const int word = d >> 6;
#pragma unroll
        for (int wd = 0; wd < kMaskWords; ++wd)
          if (wd == word) mask[wd] |= 1ull << (d & 63);
      }
    }
  }
  // final candidates: mask bits whose fast score is within eps of the FINAL maximum.  Once
  // d >= w + (bs-1-bs/2) every pattern tap clamps to column 0 (ext.h:152-154), so all such d
  // have the same reference score by construction: only the lowest of them can win.
  const int d_clamped = w + (bs - 1 - bs / 2);
  bool have_clamped = false;
  int ncand = 0;
#pragma unroll
  for (int wd = 0; wd < kMaskWords; ++wd) {
    unsigned long long bits = mask[wd], keep = 0ull;
    while (bits) {
      const int b = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      const int d = wd * 64 + b;
      if (d < D && v[(long)d * HW] >= m - eps) {
        if (d >= d_clamped) {
          if (have_clamped) continue;
          have_clamped = true;
        }
        keep |= 1ull << b;
        ++ncand;
      }
    }
    mask[wd] = keep;
  }
  int out_i = mi;
  if (ncand > 1) {
    const float* a = in0 + f * HW;
    const float* b = in1 + f * in1_frame_stride;
    float eb = 0.f;
    bool first = true;
#pragma unroll
    for (int wd = 0; wd < kMaskWords; ++wd) {
      unsigned long long bits = mask[wd];
      while (bits) {
        const int bb = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        const int d = wd * 64 + bb;
        const float e = ncc_exact_point(a, b, H, W, h, w, d, bs);
        if (first || e > eb) { eb = e; out_i = d; first = false; }   // ascending d, strict >: first index wins
      }
    }
  }
  idx[p] = out_i;
  if (best) best[p] = v[(long)out_i * HW];
}

int argmax_rerank_f32(const float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                      float* best, int frames, int D, int H, int W, int bs, float eps, hipStream_t stream) {
  if (D > kMaskWords * 64) return CTD_ERR_UNSUPPORTED;
  const long total = (long)frames * H * W;
  hipLaunchKernelGGL(argmax_rerank_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, vol, in0, in1,
                     in1_frame_stride, idx, best, D, H, W, bs, eps, total);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
