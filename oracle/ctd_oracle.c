/*
 * ctd_oracle.c -- CPU restatement of the reference's disparity hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load the library built from
 * this file (oracle/libctd_oracle.so).  The product path
 * (connecting_the_dots_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * below bit-for-bit (native ops) or to the stated tolerance (ATen-composed ops)
 * against tests/golden/ (.npz files), which tests/golden/make_golden.py generated in
 * the build container by running the reference itself: the unmodified
 * torchext/ext/ext_cpu.cpp compiled by oracle/build_ref.py, and
 * model/networks.py imported through oracle/ref_python.py.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (no -march=native: FMA contraction would break bit-parity with the
 * reference CPU build).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define T float
#define SFX f32
#define SQRT sqrtf
#define FABS fabsf
#include "ctd_oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FABS

#define T double
#define SFX f64
#define SQRT sqrt
#define FABS fabs
#include "ctd_oracle_impl.h"
#undef T
#undef SFX
#undef SQRT
#undef FABS

/* ------------------------------------------------------------------------- *
 * CrossCheckFunctor::operator()   torchext/ext/ext.h:49-66   (host ext_cpu.cpp:39-56)
 * in0 int64 [n0] (indices into in1), in1 int64 [n1] (indices into in0) -> out u8 [n0]:
 * 1 where the match is mutual.  `int idx1 = in0[idx0]` truncates to int (ext.h:61).
 * The reference reads in1[idx1] without an upper bound check (undefined behaviour for
 * idx1 >= n1); restated as "not mutual".
 * ------------------------------------------------------------------------- */
int ctd_oracle_crosscheck(const int64_t* in0, const int64_t* in1, long nelem0, long nelem1, uint8_t* out) {
  for (long idx0 = 0; idx0 < nelem0; ++idx0) {
    int idx1 = (int)in0[idx0];
    out[idx0] = idx1 >= 0 && idx1 < nelem1 && in1[idx1] >= 0 && idx0 == in1[idx1];
  }
  return 0;
}

/* reflect index without repeating the edge (torch.nn.ReflectionPad2d) */
static inline int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return i;
}

/* ------------------------------------------------------------------------- *
 * LCN.tforward   model/networks.py:507-533   (f32, [N,1,H,W])
 *   boxs  = ones-conv over ReflectionPad2d(r)          :513-518, :524
 *   avgs  = boxs / (2r+1)^2                            :526
 *   stds  = sqrt(box(x^2)/(2r+1)^2 - avgs^2 + 1e-6)+eps:528-531
 *   out   = (x - avgs) / stds,  stds                   :533
 * The reference's box sums come out of ATen's conv2d whose summation order is
 * unspecified; this restatement sums separably (dx ascending inside, dy
 * ascending outside) in double and rounds once to float, which is within
 * half an ulp of the exact sum, then follows the reference's f32 elementwise
 * order.  Parity against the reference is therefore by tolerance (tests).
 * ------------------------------------------------------------------------- */
int ctd_oracle_lcn_f32(const float* x, float* y, float* stds, int n, int height,
                       int width, int radius, float eps) {
  const int ks = 2 * radius + 1;
  const float cnt = (float)(ks * ks);
  if (radius < 0 || radius >= height || radius >= width) return 1;
  double* r1 = (double*)malloc(sizeof(double) * (size_t)height * width);
  double* r2 = (double*)malloc(sizeof(double) * (size_t)height * width);
  if (!r1 || !r2) { free(r1); free(r2); return 2; }
  for (int b = 0; b < n; ++b) {
    const float* xb = x + (size_t)b * height * width;
    for (int h = 0; h < height; ++h)
      for (int w = 0; w < width; ++w) {
        double s1 = 0, s2 = 0;
        for (int dx = -radius; dx <= radius; ++dx) {
          float v = xb[(size_t)h * width + reflect_idx(w + dx, width)];
          float v2 = v * v;                     /* data**2 is an f32 tensor, :528 */
          s1 += (double)v;
          s2 += (double)v2;
        }
        r1[(size_t)h * width + w] = s1;
        r2[(size_t)h * width + w] = s2;
      }
    for (int h = 0; h < height; ++h)
      for (int w = 0; w < width; ++w) {
        double s1 = 0, s2 = 0;
        for (int dy = -radius; dy <= radius; ++dy) {
          int hh = reflect_idx(h + dy, height);
          s1 += r1[(size_t)hh * width + w];
          s2 += r2[(size_t)hh * width + w];
        }
        float boxs = (float)s1, boxs_2n = (float)s2;
        float avgs = boxs / cnt;
        float var = boxs_2n / cnt - avgs * avgs + 1e-6f;
        float sd = sqrtf(var) + eps;
        size_t o = ((size_t)b * height + h) * width + w;
        y[o] = (xb[(size_t)h * width + w] - avgs) / sd;
        stds[o] = sd;
      }
  }
  free(r1);
  free(r2);
  return 0;
}

/* ------------------------------------------------------------------------- *
 * data/lcn/lcn.pyx:16-58  (Cython data-generation variant; f32 image [H,W])
 * two-pass window mean / std over (2ks+1)^2, zero border of width ks,
 * out = (x-mean)/(std+eps); returns the RAW std image.
 * ------------------------------------------------------------------------- */
int ctd_oracle_lcn_datagen_f32(const float* img, float* out, float* out_std,
                               int height, int width, int ks, float eps) {
  memset(out, 0, sizeof(float) * (size_t)height * width);
  memset(out_std, 0, sizeof(float) * (size_t)height * width);
  const float num = (float)((ks * 2 + 1) * (ks * 2 + 1));   /* lcn.pyx:26 */
  for (int y = ks; y < height - ks; ++y)
    for (int x = ks; x < width - ks; ++x) {
      float mean = 0;
      for (int i = -ks; i <= ks; ++i)
        for (int j = -ks; j <= ks; ++j)
          mean += img[(size_t)(y + i) * width + x + j];
      mean = mean / num;
      float std = 0;
      for (int i = -ks; i <= ks; ++i)
        for (int j = -ks; j <= ks; ++j) {
          float t = img[(size_t)(y + i) * width + x + j] - mean;
          std += t * t;
        }
      std = sqrtf(std / num);
      out[(size_t)y * width + x] = (img[(size_t)y * width + x] - mean) / (std + eps);
      out_std[(size_t)y * width + x] = std;
    }
  return 0;
}

/* ------------------------------------------------------------------------- *
 * DispToDepth.tforward   model/networks.py:313-321
 *   disp = relu(disp) + 1e-12 ;  depth = bf / disp
 * "python_float / tensor" dispatches to Tensor.__rtruediv__ = reciprocal()*bf,
 * i.e. two roundings; restated the same way.
 * ------------------------------------------------------------------------- */
int ctd_oracle_disp_to_depth_f32(const float* disp, float* depth, long n, float bf) {
  for (long i = 0; i < n; ++i) {
    float d = disp[i] > 0.f ? disp[i] : 0.f;
    d = d + 1e-12f;
    depth[i] = (1.0f / d) * bf;
  }
  return 0;
}
