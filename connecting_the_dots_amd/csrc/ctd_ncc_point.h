// ctd_ncc_point.h -- one NCC output in the reference's operation order from windows staged in LDS.
// XCorrVolFunctor<T>::operator()  /root/reference/torchext/ext/ext.h:120-191 (two passes over the window,
// means first, channels accumulated in order).  Used by the exact re-scoring of the ranked argmax
// (argmax_rerank.hip); bit-identical to the reference's FMA-free CPU build.
#ifndef CTD_NCC_POINT_H
#define CTD_NCC_POINT_H
#include "ctd_common.h"

namespace ctd {

__device__ inline float ncc_norm(float s0, float s1) {
  // T norm = sqrt(sigma0 * sigma1) + 1e-8;  sqrt in float, the add in double (ext.h:185)
  return (float)((double)sqrtf(s0 * s1) + 1e-8);
}
__device__ inline double ncc_norm(double s0, double s1) { return sqrt(s0 * s1) + 1e-8; }

// reference-order NCC of one disparity from windows staged in LDS: sA[bs][bs] is the frame window,
// sB[bs][bs + D - 1] the pattern rows from column w - half - (D-1) on (replicate border baked in), so tap
// (bh, bw) of disparity d sits at sB[bh][bw + (D-1) - d].  Same operation order as ncc_exact_point.
// `dot / norm` of one channel (ext.h:186 adds it to the running `val`)
__device__ inline float ncc_exact_term_lds(const float* sA, const float* sB, int bs, int span, int off) {
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
  return dot / ncc_norm(s0, s1);
}
__device__ inline float ncc_exact_point_lds(const float* sA, const float* sB, int bs, int span, int off) {
  float val = 0.f;
  val += ncc_exact_term_lds(sA, sB, bs, span, off);
  return val;
}

// compile-time block size: the tap loops are unrolled by rows so that the LDS reads of a row are in flight together
// sAq / sBq hold the staged values already divided by bs^2 (same correctly rounded quotient, computed once)
template <int BS>
__device__ inline float ncc_exact_point_lds_bs(const float* sA, const float* sB, const float* sAq, const float* sBq,
                                               int span, int off) {
  float mu0 = 0.f, mu1 = 0.f;
  for (int bh = 0; bh < BS; ++bh) {
    float a[BS], b[BS];
#pragma unroll
    for (int bw = 0; bw < BS; ++bw) { a[bw] = sAq[bh * BS + bw]; b[bw] = sBq[bh * span + bw + off]; }
#pragma unroll
    for (int bw = 0; bw < BS; ++bw) {
      mu0 += a[bw];
      mu1 += b[bw];
    }
  }
  float s0 = 0.f, s1 = 0.f, dot = 0.f;
  for (int bh = 0; bh < BS; ++bh) {
    float a[BS], b[BS];
#pragma unroll
    for (int bw = 0; bw < BS; ++bw) { a[bw] = sA[bh * BS + bw]; b[bw] = sB[bh * span + bw + off]; }
#pragma unroll
    for (int bw = 0; bw < BS; ++bw) {
      const float v0 = a[bw] - mu0;
      const float v1 = b[bw] - mu1;
      dot += v0 * v1;
      s0 += v0 * v0;
      s1 += v1 * v1;
    }
  }
  float val = 0.f;
  val += dot / ncc_norm(s0, s1);
  return val;
}

}  // namespace ctd
#endif
