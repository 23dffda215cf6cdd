// photometric.hip -- block photometric loss, forward and backward.
//
// Replaces photometric_loss_forward_kernel / photometric_loss_backward_kernel
// (torchext/ext/ext_kernel.cu:54-112) and their functors PhotometricLossForward /
// PhotometricLossBackward (torchext/ext/ext.h:201-344): MSE, SAD, soft-census MSE,
// soft-census SAD over a bs x bs replicate-clamped block, summed over channels.
//
// Forward: one thread per output pixel, taps in the reference order (bidx outer, channel
// inner, each term divided by bs^2 before it is accumulated), built without FMA contraction:
// bit-identical to the reference CPU build.
//
// Backward: the reference scatters with atomicAdd into a zeroed buffer (ext.h:315,338-339),
// which is non-deterministic on a GPU.  Here every INPUT pixel gathers its contributions,
// visiting (output pixel ascending, tap ascending) -- exactly the order in which the
// reference's serial CPU loop adds them -- so the result is deterministic, needs no
// pre-zeroed buffer and no atomics, and is bit-identical to the reference CPU build.
#include "ctd_internal.h"

namespace ctd {

__device__ inline float t_sqrt(float x) { return sqrtf(x); }
__device__ inline double t_sqrt(double x) { return sqrt(x); }
__device__ inline float t_abs(float x) { return fabsf(x); }
__device__ inline double t_abs(double x) { return fabs(x); }

// h(x) = 0.5 * (1 + x / sqrt(x^2 + eps)); inner part in T, the 0.5 multiply in double (ext.h:249)
template <typename T>
__device__ inline T soft_step(T x, T eps) {
  return (T)(0.5 * (double)((T)1 + x / t_sqrt(x * x + eps)));
}

template <typename T, int TYPE>
__global__ __launch_bounds__(256) void photometric_fwd_kernel(const T* __restrict__ es, const T* __restrict__ ta,
                                                              T* __restrict__ out, int C, int H, int W, int bs,
                                                              T eps) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63);
  const int h = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int n = blockIdx.z;
  if (w >= W || h >= H) return;
  const int half = bs / 2;
  const T bs2 = (T)(bs * bs);
  const long HW = (long)H * W;
  const T* e = es + (long)n * C * HW;
  const T* t = ta + (long)n * C * HW;
  T loss = 0;
  for (int bh = 0; bh < bs; ++bh) {
    const int h0 = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      const int w0 = clampi(w + bw - half, 0, W - 1);
      for (int c = 0; c < C; ++c) {
        const long i = c * HW + (long)h0 * W + w0;
        if (TYPE == 0 || TYPE == 1) {
          const T diff = e[i] - t[i];
          if (TYPE == 0) loss += diff * diff / bs2;
          else loss += t_abs(diff) / bs2;
        } else {
          const long ic = c * HW + (long)h * W + w;
          const T des = e[i] - e[ic];
          const T dta = t[i] - t[ic];
          const T diff = soft_step(des, eps) - soft_step(dta, eps);
          if (TYPE == 2) loss += diff * diff / bs2;
          else loss += t_abs(diff) / bs2;
        }
      }
    }
  }
  out[(long)n * HW + (long)h * W + w] = loss;
}

// gradient contribution of tap (input pixel i) of the output pixel with centre ic -- ext.h:303-340
template <typename T, int TYPE>
__device__ inline T tap_grad(const T* __restrict__ e, const T* __restrict__ t, long i, long ic, T go, T bs2, T eps) {
  if (TYPE == 0 || TYPE == 1) {
    const T diff = e[i] - t[i];
    T grad;
    if (TYPE == 0) grad = (T)2 * diff;
    else grad = diff < 0 ? (T)-1 : (diff > 0 ? (T)1 : (T)0);
    return grad / bs2 * go;
  } else {
    const T des = e[i] - e[ic];
    const T dta = t[i] - t[ic];
    const T diff = soft_step(des, eps) - soft_step(dta, eps);
    T gl;
    if (TYPE == 2) gl = (T)2 * diff;
    else gl = diff < 0 ? (T)-1 : (diff > 0 ? (T)1 : (T)0);
    gl = gl / bs2;
    const T tmp = des * des + eps;
    const T gh = (T)(0.5 * (double)eps / (double)t_sqrt(tmp * tmp * tmp));
    return go * gl * gh;
  }
}

// range of block offsets b in [0, bs) with clamp(p + b - half, 0, n-1) == p0
__device__ inline void tap_range(int p, int p0, int n, int bs, int half, int& lo, int& hi) {
  const int b = p0 - p + half;          // the unclamped offset
  lo = b;
  hi = b;
  if (p0 == 0) lo = 0;                  // everything that falls off the low edge clamps here
  if (p0 == n - 1) hi = bs - 1;         // ... and off the high edge
  if (lo < 0) lo = 0;
  if (hi > bs - 1) hi = bs - 1;
}

template <typename T, int TYPE>
__global__ __launch_bounds__(256) void photometric_bwd_kernel(const T* __restrict__ es, const T* __restrict__ ta,
                                                              const T* __restrict__ grad_out,
                                                              T* __restrict__ grad_in, int C, int H, int W, int bs,
                                                              T eps) {
  const int w0 = blockIdx.x * 64 + (threadIdx.x & 63);
  const int h0 = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int nc = blockIdx.z;            // n * C + c
  if (w0 >= W || h0 >= H) return;
  const int n = nc / C;
  const int half = bs / 2;
  const T bs2 = (T)(bs * bs);
  const long HW = (long)H * W;
  const T* e = es + (long)nc * HW;
  const T* t = ta + (long)nc * HW;
  const T* go = grad_out + (long)n * HW;
  const long i0 = (long)h0 * W + w0;
  T acc = 0;
  // output pixels whose block can reach (h0, w0): |h - h0| <= bs (clamping only pulls taps inwards)
  const int h_lo = max(0, h0 - (bs - 1 - half)), h_hi = min(H - 1, h0 + half);
  const int w_lo = max(0, w0 - (bs - 1 - half)), w_hi = min(W - 1, w0 + half);
  for (int h = h_lo; h <= h_hi; ++h) {
    int bh_lo, bh_hi;
    tap_range(h, h0, H, bs, half, bh_lo, bh_hi);
    for (int w = w_lo; w <= w_hi; ++w) {
      int bw_lo, bw_hi;
      tap_range(w, w0, W, bs, half, bw_lo, bw_hi);
      const long ic = (long)h * W + w;
      const T g = go[ic];
      const bool self = (h == h0) && (w == w0);
      if (!self || TYPE < 2) {
        for (int bh = bh_lo; bh <= bh_hi; ++bh)
          for (int bw = bw_lo; bw <= bw_hi; ++bw) acc += tap_grad<T, TYPE>(e, t, i0, ic, g, bs2, eps);
      } else {
        // census, output pixel == this pixel: every tap also subtracts its gradient from the centre,
        // interleaved with the taps that land on the centre itself (ext.h:338-339)
        for (int bh = 0; bh < bs; ++bh) {
          const int hh = clampi(h + bh - half, 0, H - 1);
          for (int bw = 0; bw < bs; ++bw) {
            const int ww = clampi(w + bw - half, 0, W - 1);
            const long i = (long)hh * W + ww;
            const T gr = tap_grad<T, TYPE>(e, t, i, ic, g, bs2, eps);
            if (i == i0) acc += gr;
            acc += -gr;
          }
        }
      }
    }
  }
  grad_in[(long)nc * HW + i0] = acc;
}

template <typename T>
static int launch_fwd(const T* es, const T* ta, T* out, int B, int C, int H, int W, int bs, int type, float eps,
                      hipStream_t stream) {
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B), block(256);
  const T e = (T)eps;
  switch (type) {
    case 0: hipLaunchKernelGGL((photometric_fwd_kernel<T, 0>), grid, block, 0, stream, es, ta, out, C, H, W, bs, e); break;
    case 1: hipLaunchKernelGGL((photometric_fwd_kernel<T, 1>), grid, block, 0, stream, es, ta, out, C, H, W, bs, e); break;
    case 2: hipLaunchKernelGGL((photometric_fwd_kernel<T, 2>), grid, block, 0, stream, es, ta, out, C, H, W, bs, e); break;
    case 3: hipLaunchKernelGGL((photometric_fwd_kernel<T, 3>), grid, block, 0, stream, es, ta, out, C, H, W, bs, e); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <typename T>
static int launch_bwd(const T* es, const T* ta, const T* go, T* gi, int B, int C, int H, int W, int bs, int type,
                      float eps, hipStream_t stream) {
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B * C), block(256);
  const T e = (T)eps;
  switch (type) {
    case 0: hipLaunchKernelGGL((photometric_bwd_kernel<T, 0>), grid, block, 0, stream, es, ta, go, gi, C, H, W, bs, e); break;
    case 1: hipLaunchKernelGGL((photometric_bwd_kernel<T, 1>), grid, block, 0, stream, es, ta, go, gi, C, H, W, bs, e); break;
    case 2: hipLaunchKernelGGL((photometric_bwd_kernel<T, 2>), grid, block, 0, stream, es, ta, go, gi, C, H, W, bs, e); break;
    case 3: hipLaunchKernelGGL((photometric_bwd_kernel<T, 3>), grid, block, 0, stream, es, ta, go, gi, C, H, W, bs, e); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int photometric_fwd_f32(const float* es, const float* ta, float* out, int B, int C, int H, int W, int bs, int type,
                        float eps, hipStream_t s) { return launch_fwd<float>(es, ta, out, B, C, H, W, bs, type, eps, s); }
int photometric_fwd_f64(const double* es, const double* ta, double* out, int B, int C, int H, int W, int bs, int type,
                        float eps, hipStream_t s) { return launch_fwd<double>(es, ta, out, B, C, H, W, bs, type, eps, s); }
int photometric_bwd_f32(const float* es, const float* ta, const float* go, float* gi, int B, int C, int H, int W,
                        int bs, int type, float eps, hipStream_t s) {
  return launch_bwd<float>(es, ta, go, gi, B, C, H, W, bs, type, eps, s);
}
int photometric_bwd_f64(const double* es, const double* ta, const double* go, double* gi, int B, int C, int H, int W,
                        int bs, int type, float eps, hipStream_t s) {
  return launch_bwd<double>(es, ta, go, gi, B, C, H, W, bs, type, eps, s);
}

// SAD / soft-census cost volume by composition of the reference ops:
//   cost[d] = photometric_loss_forward(es = P_d, ta = I)[0,0],  P_d[h,x] = P[h, clamp(x - d)]
// evaluated in one kernel with the same arithmetic and tap order, so it is bit-identical to
// shifting the pattern on the host and calling the reference forward once per disparity.
template <int TYPE>
__global__ __launch_bounds__(256) void costvol_kernel(const float* __restrict__ im, const float* __restrict__ pat,
                                                      long pat_frame_stride, float* __restrict__ cost, int H, int W,
                                                      int D, int bs, float eps) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63);
  const int h = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int f = blockIdx.z / D, d = blockIdx.z - f * D;
  if (w >= W || h >= H) return;
  const int half = bs / 2;
  const float bs2 = (float)(bs * bs);
  const long HW = (long)H * W;
  const float* t = im + (long)f * HW;                       // ta = image
  const float* e = pat + (long)f * pat_frame_stride;        // es = shifted pattern
  const float ec = e[(long)h * W + clampi(w - d, 0, W - 1)];
  const float tc = t[(long)h * W + w];
  float loss = 0.f;
  for (int bh = 0; bh < bs; ++bh) {
    const int h0 = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      const int w0 = clampi(w + bw - half, 0, W - 1);
      const float ev = e[(long)h0 * W + clampi(w0 - d, 0, W - 1)];
      const float tv = t[(long)h0 * W + w0];
      if (TYPE == 0 || TYPE == 1) {
        const float diff = ev - tv;
        if (TYPE == 0) loss += diff * diff / bs2;
        else loss += fabsf(diff) / bs2;
      } else {
        const float diff = soft_step(ev - ec, eps) - soft_step(tv - tc, eps);
        if (TYPE == 2) loss += diff * diff / bs2;
        else loss += fabsf(diff) / bs2;
      }
    }
  }
  cost[((long)f * D + d) * HW + (long)h * W + w] = loss;
}

int costvol_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W, int D,
                int bs, int type, float eps, hipStream_t stream) {
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), frames * D), block(256);
  switch (type) {
    case 0: hipLaunchKernelGGL(costvol_kernel<0>, grid, block, 0, stream, im, pat, pat_frame_stride, cost, H, W, D, bs, eps); break;
    case 1: hipLaunchKernelGGL(costvol_kernel<1>, grid, block, 0, stream, im, pat, pat_frame_stride, cost, H, W, D, bs, eps); break;
    case 2: hipLaunchKernelGGL(costvol_kernel<2>, grid, block, 0, stream, im, pat, pat_frame_stride, cost, H, W, D, bs, eps); break;
    case 3: hipLaunchKernelGGL(costvol_kernel<3>, grid, block, 0, stream, im, pat, pat_frame_stride, cost, H, W, D, bs, eps); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
