// lcn.hip -- fused local contrast normalisation.
//
// Replaces the op chain of LCN.tforward (model/networks.py:507-533): ReflectionPad2d,
// two all-ones (2r+1)^2 Conv2d passes (x and x^2) and six elementwise kernels, each a
// full HBM round trip in the reference.  Here one kernel reads x once (12 B/pixel of
// algorithmic traffic: 4 in, 8 out) and produces both outputs.
//
// Numerics: the box sums are accumulated separably in f64 (dx ascending inside a row,
// dy ascending across rows) and rounded once to f32, which is within half an ulp of the
// exact sum -- ATen's conv2d summation order is unspecified, so this is the closest
// defined target -- and the elementwise tail follows the reference's f32 operation order:
//     avgs = boxs / n;  stds = sqrt(boxs2 / n - avgs*avgs + 1e-6) + eps;  y = (x - avgs) / stds
// The CPU oracle (oracle/ctd_oracle.c: ctd_oracle_lcn_f32) uses the same order, so the
// two agree bit for bit.
#include "ctd_internal.h"

namespace ctd {

constexpr int kLcnTW = 32;
constexpr int kLcnTH = 32;
constexpr int kLcnRows = 8;   // block = 32 x 8 threads, each thread owns kLcnTH / kLcnRows CONSECUTIVE output rows
// (32 x 32 tiles: 42 staged rows per 32 outputs instead of 26 per 16 -- a fifth less of the f64 row pass -- and 28.6 KB of
// LDS = five workgroups per CU instead of four: 26.8 -> 25.6 us on 16 x 432 x 512, tools/ab_lcn.sh)
constexpr int kLcnHC = 8;     // output columns per item of the horizontal pass

__device__ inline int reflect_idx(int i, int n) {
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return i;
}

// R > 0: compile-time radius (tap loops unrolled, LDS reads pipelined); R == 0: run-time `radius_rt`.
// Both separable passes give a thread several adjacent outputs, so that the f32 -> f64 conversions, the squares and
// the LDS reads of the taps they share happen once: every output is still its own ascending sum from 0 (same bits).
template <int R>
__global__ __launch_bounds__(kLcnTW* kLcnRows) void lcn_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               float* __restrict__ stds, int H, int W, int radius_rt,
                                                               float eps) {
  const int radius = R > 0 ? R : radius_rt;
  extern __shared__ double lds_d[];
  const int TRr = kLcnTH + 2 * radius;       // staged rows
  const int TCc = kLcnTW + 2 * radius;       // staged columns
  double* rs1 = lds_d;                       // [TRr][kLcnTW] row sums of x
  double* rs2 = lds_d + TRr * kLcnTW;        // [TRr][kLcnTW] row sums of x^2
  float* tile = (float*)(lds_d + 2 * TRr * kLcnTW);   // [TRr][TCc] reflect-padded input

  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * kLcnTW + tx;
  const int w_lo = blockIdx.x * kLcnTW, h_lo = blockIdx.y * kLcnTH;
  const long base = (long)blockIdx.z * H * W;
  const float* xb = x + base;

  // batches of independent loads: one memory round trip per 8 elements of a thread instead of one each
  for (int i0 = tid; i0 < TRr * TCc; i0 += kLcnTW * kLcnRows * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + kLcnTW * kLcnRows * u, TRr * TCc - 1);
      const int r = i / TCc, c = i - r * TCc;
      // tiles hanging over the bottom / right edge stage clamped garbage that is never stored
      const int hh = reflect_idx(min(h_lo + r - radius, H - 1 + radius), H);
      const int ww = reflect_idx(min(w_lo + c - radius, W - 1 + radius), W);
      t[u] = xb[(long)hh * W + ww];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + kLcnTW * kLcnRows * u < TRr * TCc) tile[i0 + kLcnTW * kLcnRows * u] = t[u];
  }
  __syncthreads();

  if (R > 0) {
    // item = (staged row, chunk of kLcnHC output columns): kLcnHC + 2R taps converted once, kLcnHC sums each
    constexpr int NT = kLcnHC + 2 * (R > 0 ? R : 1);
    constexpr int CH = kLcnTW / kLcnHC;
    for (int it = tid; it < TRr * CH; it += kLcnTW * kLcnRows) {
      const int r = it / CH, c0 = (it - r * CH) * kLcnHC;
      const float* row = tile + r * TCc + c0;
      double v1[NT], v2[NT];
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const float v = row[k];
        const float q = v * v;                // data**2 is an f32 tensor (networks.py:528)
        v1[k] = (double)v;
        v2[k] = (double)q;
      }
#pragma unroll
      for (int o = 0; o < kLcnHC; ++o) {
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) {
          s1 += v1[o + k];
          s2 += v2[o + k];
        }
        rs1[r * kLcnTW + c0 + o] = s1;
        rs2[r * kLcnTW + c0 + o] = s2;
      }
    }
  } else {
    for (int r = ty; r < TRr; r += kLcnRows) {
      const float* row = tile + r * TCc + tx;
      double s1 = 0, s2 = 0;
      for (int k = 0; k <= 2 * radius; ++k) {
        float v = row[k];
        float v2 = v * v;
        s1 += (double)v;
        s2 += (double)v2;
      }
      rs1[r * kLcnTW + tx] = s1;
      rs2[r * kLcnTW + tx] = s2;
    }
  }
  __syncthreads();

  const int w = w_lo + tx;
  const float cnt = (float)((2 * radius + 1) * (2 * radius + 1));
  constexpr int RPT = kLcnTH / kLcnRows;     // consecutive output rows of a thread
  constexpr int NV = RPT + 2 * (R > 0 ? R : 0);
  double c1[R > 0 ? NV : 1], c2[R > 0 ? NV : 1];
  if (R > 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      c1[k] = rs1[(ty * RPT + k) * kLcnTW + tx];
      c2[k] = rs2[(ty * RPT + k) * kLcnTW + tx];
    }
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int r = ty * RPT + i;
    const int h = h_lo + r;
    double s1 = 0, s2 = 0;
    if (R > 0) {
#pragma unroll
      for (int k = 0; k < 2 * R + 1; ++k) {
        s1 += c1[i + k];
        s2 += c2[i + k];
      }
    } else {
      for (int k = 0; k <= 2 * radius; ++k) {
        s1 += rs1[(r + k) * kLcnTW + tx];
        s2 += rs2[(r + k) * kLcnTW + tx];
      }
    }
    if (w >= W || h >= H) continue;
    float boxs = (float)s1, boxs2 = (float)s2;
    float avgs = boxs / cnt;
    float var = boxs2 / cnt - avgs * avgs + 1e-6f;
    float sd = sqrtf(var) + eps;
    float xv = tile[(r + radius) * TCc + tx + radius];
    long o = base + (long)h * W + w;
    y[o] = (xv - avgs) / sd;
    stds[o] = sd;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Tolerance-level variant (`algo = 'fast'`, radius 1 .. 7): the same tile, the box sums in F32 as sliding windows -- the first
// output of a run is a fresh 11-term sum, each next one adds the entering and subtracts the leaving tap (3 additions per
// output and pass instead of 11 half-rate f64 ones), runs of 8 columns / 4 rows.  The contract for the LCN is a tolerance
// (ATen's conv2d summation order is unspecified): every output within 1e-5 |b| + 1e-6 of the reference's
// networks.LCN goldens and of the f64 kernel (tests/test_lcn_gpu.py); a sliding sum of at most 18 taps carries <= 14
// roundings of the largest partial sum, 1e-7 relative on these positive sums.  Half the LDS (seven workgroups per CU).
// ---------------------------------------------------------------------------------------------------------
template <int R, int TW, int TH>
__global__ __launch_bounds__(256) void lcn_fast_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       float* __restrict__ stds, int H, int W, float eps) {
  constexpr int ROWS = 256 / TW;                                   // thread rows of the block
  constexpr int TRr = TH + 2 * R, TCc = TW + 2 * R, NTAP = 2 * R + 1;
  // sliding sums from radius 4 on; below, a window is at most 7 taps a side -- a fresh sum costs no more, and the residue a
  // bright sample leaves in a running sum (one ulp of ITS square) is divided by only 9 .. 49 taps: 3.5e-6 of std on the
  // variance floor at radius 1 (tools/fuzz_lcn.py)
  constexpr bool SLIDE = R >= 4;
  __shared__ float rs1[TRr][TW], rs2[TRr][TW];                     // row sums of x and x^2
  __shared__ float tile[TRr][TCc];                                 // reflect-padded input
  const int tid = threadIdx.x;
  const int tx = tid % TW, ty = tid / TW;
  const int w_lo = blockIdx.x * TW, h_lo = blockIdx.y * TH;
  const long base = (long)blockIdx.z * H * W;
  const float* xb = x + base;
  for (int i0 = tid; i0 < TRr * TCc; i0 += 256 * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 256 * u, TRr * TCc - 1);
      const int r = i / TCc, c = i - r * TCc;
      const int hh = reflect_idx(min(h_lo + r - R, H - 1 + R), H);
      const int ww = reflect_idx(min(w_lo + c - R, W - 1 + R), W);
      t[u] = xb[(long)hh * W + ww];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 256 * u < TRr * TCc) (&tile[0][0])[i0 + 256 * u] = t[u];
  }
  __syncthreads();
  // Everything below works on x - c: (x - avg) / std and std do not change under a shift, and E[x^2] - avg^2 of the shifted
  // samples no longer cancels against the image's DC level (frames with an offset of 10 and a deviation of 3 lose a factor
  // 12 of the f32 sums' accuracy otherwise: the reference golden "n").  c = 0 when the staged tile reaches zero (the plain
  // sums: EXACT on a zero background with sparse bright samples), else the tile's MEAN (one reduction): close to every
  // window's own mean wherever the tile has one level, and a sparse bright sample moves it by 1 / 1924 of its height.
  // (Until round 5 c was the tile's centre SAMPLE: a bright dot there on a dark flat background made every window of the
  // tile cancel against 0.81 -- std off by up to 2.5e-3 relative.  Tried on the way: the tile's value closest to zero,
  // med3(0, min, max) -- frames with a DC level fall back to the plain sums' conditioning, the golden "n" again; the mean
  // alone -- a zero background is then -mean, not 0, and the windows without a sample sit on the 1e-6 floor: 4e-6 of std.)
  __shared__ float red[3][4];
  {
    float sm = 0.f, mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < TRr * TCc; i += 256) {
      const float v = (&tile[0][0])[i];
      sm += v;
      mn = fminf(mn, v);
      mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      sm += __shfl_xor(sm, o);
      mn = fminf(mn, __shfl_xor(mn, o));
      mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = sm; red[1][tid >> 6] = mn; red[2][tid >> 6] = mx; }
  }
  __syncthreads();
  const float t_min = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
  const float t_max = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
  const float t_mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.f / (float)(TRr * TCc));
  const float ctr = (t_min <= 0.f && t_max >= 0.f) ? 0.f : t_mean;
  // horizontal pass: item = (staged row, run of kLcnHC output columns)
  constexpr int CH = TW / kLcnHC, NT = kLcnHC + 2 * R;
  for (int it = tid; it < TRr * CH; it += 256) {
    const int r = it / CH, c0 = (it - r * CH) * kLcnHC;
    float v[NT], q[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      v[k] = tile[r][c0 + k] - ctr;
      q[k] = v[k] * v[k];
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NTAP; ++k) {
      s1 += v[k];
      s2 += q[k];
    }
    rs1[r][c0] = s1;
    rs2[r][c0] = s2;
#pragma unroll
    for (int o = 1; o < kLcnHC; ++o) {
      if constexpr (SLIDE) {
        s1 = (s1 + v[o + NTAP - 1]) - v[o - 1];
        s2 = (s2 + q[o + NTAP - 1]) - q[o - 1];
      } else {
        s1 = s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NTAP; ++k) {
          s1 += v[o + k];
          s2 += q[o + k];
        }
      }
      rs1[r][c0 + o] = s1;
      rs2[r][c0 + o] = s2;
    }
  }
  __syncthreads();
  // vertical pass: a thread owns TH / ROWS consecutive output rows of its column
  const int w = w_lo + tx;
  constexpr int RPT = TH / ROWS, NV = RPT + 2 * R;
  static_assert(TH % ROWS == 0 && 256 % TW == 0 && TW % kLcnHC == 0, "tile shape");
  constexpr float cnt = (float)(NTAP * NTAP);
  float c1[NV], c2[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    c1[k] = rs1[ty * RPT + k][tx];
    c2[k] = rs2[ty * RPT + k][tx];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NTAP; ++k) {
    s1 += c1[k];
    s2 += c2[k];
  }
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    if (i > 0) {
      if constexpr (SLIDE) {
        s1 = (s1 + c1[i + NTAP - 1]) - c1[i - 1];
        s2 = (s2 + c2[i + NTAP - 1]) - c2[i - 1];
      } else {
        s1 = s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NTAP; ++k) {
          s1 += c1[i + k];
          s2 += c2[i + k];
        }
      }
    }
    const int r = ty * RPT + i, h = h_lo + r;
    if (w >= W || h >= H) continue;
    const float avgs = s1 / cnt;
    const float var = s2 / cnt - avgs * avgs + 1e-6f;
    const float sd = sqrtf(var) + eps;
    const float xv = tile[r + R][tx + R] - ctr;
    const long o = base + (long)h * W + w;
    y[o] = (xv - avgs) / sd;
    stds[o] = sd;
  }
}

#ifndef CTD_LCN_FAST_TW
#define CTD_LCN_FAST_TW 64   // (A/B, tools/time_lcn_variants.py, 16 x 432 x 512: 64 x 16 16.5 us, 32 x 32 16.9-17.9, 32 x 16 17.1-17.4, 64 x 8 17.8-18.0, 64 x 32 18.3-19.0, 32 x 64 19.1-19.3, 128 x 16 20.6, 128 x 8 22.1)
#define CTD_LCN_FAST_TH 16
#endif

int lcn_fast_f32(const float* x, float* y, float* stds, int N, int H, int W, int radius, float eps, hipStream_t stream) {
  // radius 5 is the one the reference uses (exp_synph.py:41) and the one the tile shape was tuned for; 1 .. 7 share the kernel
  constexpr int TW = CTD_LCN_FAST_TW, TH = CTD_LCN_FAST_TH;
  dim3 grid(ceil_div(W, TW), ceil_div(H, TH), N), block(256);
  switch (radius) {
#define CTD_LCN_FAST_CASE(R) \
    case R: hipLaunchKernelGGL((lcn_fast_kernel<R, TW, TH>), grid, block, 0, stream, x, y, stds, H, W, eps); break;
    CTD_LCN_FAST_CASE(1) CTD_LCN_FAST_CASE(2) CTD_LCN_FAST_CASE(3) CTD_LCN_FAST_CASE(4)
    CTD_LCN_FAST_CASE(5) CTD_LCN_FAST_CASE(6) CTD_LCN_FAST_CASE(7)
#undef CTD_LCN_FAST_CASE
    default: return CTD_ERR_UNSUPPORTED;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int lcn_f32(const float* x, float* y, float* stds, int N, int H, int W, int radius, float eps, hipStream_t stream) {
  const int TRr = kLcnTH + 2 * radius, TCc = kLcnTW + 2 * radius;
  size_t lds = sizeof(double) * 2 * TRr * kLcnTW + sizeof(float) * (size_t)TRr * TCc;
  if (lds > 160 * 1024) return CTD_ERR_UNSUPPORTED;
  auto kern = radius == 5 ? lcn_kernel<5> : lcn_kernel<0>;      // 5: the only radius the reference uses (exp_synph.py:41)
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(ceil_div(W, kLcnTW), ceil_div(H, kLcnTH), N), block(kLcnTW, kLcnRows);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, x, y, stds, H, W, radius, eps);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Data-generation variant: data/lcn/lcn.pyx:16-58 (`lcn.normalize(img, kernel_size, epsilon)`, used on the
// ambient-gradient image in data/create_syn_data.py:181).  Two-pass window mean / standard deviation over
// (2ks+1)^2 taps accumulated in f32 in row-major tap order (bit-identical to the Cython loop), output
// (x - mean) / (std + eps) and the RAW std; a border of width ks stays zero in both outputs.
// ---------------------------------------------------------------------------------------------------------
constexpr int kLdTW = 64, kLdTH = 8;

__global__ __launch_bounds__(256) void lcn_datagen_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                          float* __restrict__ out_std, int H, int W, int ks, float eps) {
  extern __shared__ float lds_f[];
  const int TC = kLdTW + 2 * ks, TR = kLdTH + 2 * ks;
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kLdTW, y0 = blockIdx.y * kLdTH;
  const long base = (long)blockIdx.z * H * W;
  for (int i0 = threadIdx.x; i0 < TR * TC; i0 += 256 * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 256 * u, TR * TC - 1);
      const int r = i / TC, c = i - r * TC;
      t[u] = img[base + (long)clampi(y0 + r - ks, 0, H - 1) * W + clampi(x0 + c - ks, 0, W - 1)];   // clamped taps are
    }                                                                                                // only read by border pixels
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 256 * u < TR * TC) lds_f[i0 + 256 * u] = t[u];
  }
  __syncthreads();
  const float num = (float)((ks * 2 + 1) * (ks * 2 + 1));            // lcn.pyx:33
#pragma unroll
  for (int k = 0; k < kLdTH / 4; ++k) {
    const int ty = ty0 + 4 * k, x = x0 + tx, y = y0 + ty;
    if (x >= W || y >= H) continue;
    float o = 0.f, sd = 0.f;
    if (y >= ks && y < H - ks && x >= ks && x < W - ks) {
      const float* win = lds_f + ty * TC + tx;                       // top-left tap of the window
      float mean = 0.f;
      for (int i = 0; i <= 2 * ks; ++i)
        for (int j = 0; j <= 2 * ks; ++j) mean += win[i * TC + j];
      mean = mean / num;
      float acc = 0.f;
      for (int i = 0; i <= 2 * ks; ++i)
        for (int j = 0; j <= 2 * ks; ++j) {
          const float d = win[i * TC + j] - mean;
          acc = acc + d * d;
        }
      sd = sqrtf(acc / num);
      o = (win[ks * TC + ks] - mean) / (sd + eps);
    }
    out[base + (long)y * W + x] = o;
    out_std[base + (long)y * W + x] = sd;
  }
}

int lcn_datagen_f32(const float* img, float* out, float* out_std, int N, int H, int W, int ks, float eps,
                    hipStream_t stream) {
  const size_t lds = sizeof(float) * (size_t)(kLdTW + 2 * ks) * (kLdTH + 2 * ks);
  if (lds > 64 * 1024) return CTD_ERR_UNSUPPORTED;
  dim3 grid(ceil_div(W, kLdTW), ceil_div(H, kLdTH), N);
  hipLaunchKernelGGL(lcn_datagen_kernel, grid, dim3(256), lds, stream, img, out, out_std, H, W, ks, eps);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
