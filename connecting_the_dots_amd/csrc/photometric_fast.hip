// photometric_fast.hip -- tolerance-level (|a-b| <= 1e-5|b| + 1e-6) block photometric loss, f32.
//
// Same functions as photometric.hip (PhotometricLossForward / PhotometricLossBackward,
// /root/reference/torchext/ext/ext.h:201-344) with the summation order left free and the correctly rounded
// sqrt / divide chains replaced by v_rsq_f32:
//
//   h(x)            = 0.5 * (1 + x * rsq(x^2 + eps))                        (ext.h:249-250)
//   dh/dx           = 0.5 * eps * rsq(x^2 + eps)^3                          (ext.h:332-333)
//
// Forward: one thread per output pixel over an LDS tile with the replicate border baked in.
//
// Backward without atomics and without the reference's scatter: for the census types the contribution of
// tap q of output pixel p is  g(p->q) = go[p] * K(p,q),  K(p,q) = gl(diff)/bs^2 * dh/dx(des),
// des = es[q] - es[p],  and it is added at q and subtracted at p (ext.h:338-339).  gl is odd in diff, diff
// is odd under p <-> q and dh/dx is even, so K(q,p) = -K(p,q) and the gradient of an input pixel q is
//
//   grad[q] = sum over p within bs/2 of q of  K(p,q) * ( m(p->q) * go[p] + m(q->p) * go[q] )
//
// -- one K per pixel pair, the cost of a forward pass.  m(a->b) is how many taps of a's clamped window land
// on b (1 in the interior, more for b on the image border, ext.h:296-297).  For MSE / SAD the gradient is
// gl(es[q]-ta[q])/bs^2 * sum_p m(p->q) go[p].
//
// sign() of the census-SAD gradient is discontinuous: where the fast diff is within 1e-6 of zero it is
// recomputed with the reference's own operation chain, so the sign (hence the gradient) agrees with the
// reference wherever the reference's sign is not itself decided by its last bit.
#include "ctd_internal.h"

namespace ctd {

constexpr int kPTW = 64, kPTH = 8;      // output tile per 256-thread workgroup (2 pixels per thread)

// reference-order soft step (ext.h:249), used only to settle the sign near zero
__device__ inline float soft_step_ref(float x, float eps) {
  return (float)(0.5 * (double)(1.f + x / sqrtf(x * x + eps)));
}

// number of offsets o in [-half, half] with clamp(a + o, 0, n-1) == b   (a inside the image)
__device__ inline int tap_mult(int a, int b, int n, int half) {
  if (a < 0 || a >= n) return 0;
  int lo = b - a, hi = b - a;                     // the unclamped offset
  if (b == 0) lo = -half;                         // everything that falls off the low edge clamps onto 0
  if (b == n - 1) hi = half;
  lo = max(lo, -half);
  hi = min(hi, half);
  return max(hi - lo + 1, 0);
}

template <int BS>
__device__ inline void stage_tile(float (*dst)[kPTW + BS - 1], const float* __restrict__ src, int H, int W, int x0,
                                  int y0) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1;
  for (int i = threadIdx.x; i < TW * TH; i += 256) {
    const int r = i / TW, c = i - r * TW;
    dst[r][c] = src[(long)clampi(y0 + r - HALF, 0, H - 1) * W + clampi(x0 + c - HALF, 0, W - 1)];
  }
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void photometric_fast_fwd_kernel(const float* __restrict__ es,
                                                                   const float* __restrict__ ta,
                                                                   float* __restrict__ out, int C, int H, int W,
                                                                   float eps) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1;
  __shared__ float sE[TH][TW], sT[TH][TW];
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kPTW, y0 = blockIdx.y * kPTH, n = blockIdx.z;
  const long HW = (long)H * W;
  float loss[2] = {0.f, 0.f};
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    stage_tile<BS>(sE, es + ((long)n * C + c) * HW, H, W, x0, y0);
    stage_tile<BS>(sT, ta + ((long)n * C + c) * HW, H, W, x0, y0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int ty = ty0 + 4 * k;
      const float ec = sE[ty + HALF][tx + HALF], tc = sT[ty + HALF][tx + HALF];
      float acc = 0.f;
#pragma unroll 1                                         // rows rolled: a fully unrolled window hoists ~2*BS^2 LDS
      for (int dy = 0; dy < BS; ++dy)                     // loads into registers and leaves one wave per SIMD
#pragma unroll
        for (int dx = 0; dx < BS; ++dx) {
          const float e = sE[ty + dy][tx + dx], t = sT[ty + dy][tx + dx];
          if (TYPE == 0) {
            const float d = e - t;
            acc = fmaf(d, d, acc);
          } else if (TYPE == 1) {
            acc += fabsf(e - t);
          } else {
            const float des = e - ec, dta = t - tc;
            const float r1 = __builtin_amdgcn_rsqf(fmaf(des, des, eps)), r2 = __builtin_amdgcn_rsqf(fmaf(dta, dta, eps));
            const float d2 = des * r1 - dta * r2;                  // 2 * (h(des) - h(dta))
            if (TYPE == 2) acc = fmaf(d2, d2, acc);
            else acc += fabsf(d2);
          }
        }
      const float scale = (TYPE == 2 ? 0.25f : (TYPE == 3 ? 0.5f : 1.f)) / (float)(BS * BS);
      loss[k] = fmaf(acc, scale, loss[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int x = x0 + tx, y = y0 + ty0 + 4 * k;
    if (x < W && y < H) out[(long)n * HW + (long)y * W + x] = loss[k];
  }
}

// gradient of the two pixels a thread owns in the staged tile; BORDER = general tap multiplicities
template <int TYPE, int BS, bool BORDER>
__device__ inline void bwd_tile(const float (*sE)[kPTW + BS - 1], const float (*sT)[kPTW + BS - 1],
                                const float (*sG)[kPTW + BS - 1], float* __restrict__ grad_plane, int H, int W, int x0,
                                int y0, float eps) {
  constexpr int HALF = BS / 2;
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ty = ty0 + 4 * k;
    const int qx = x0 + tx, qy = y0 + ty;
    if (qx >= W || qy >= H) continue;
    const float eq = sE[ty + HALF][tx + HALF], tq = sT[ty + HALF][tx + HALF], gq = sG[ty + HALF][tx + HALF];
    // multiplicities per axis: A = taps of p landing on q, B = taps of q landing on p
    float mxa[BS], mxb[BS];
    if (BORDER) {
#pragma unroll
      for (int d = 0; d < BS; ++d) {
        const int px = qx + d - HALF;
        mxa[d] = (float)tap_mult(px, qx, W, HALF);
        mxb[d] = (px >= 0 && px < W) ? (float)tap_mult(qx, px, W, HALF) : 0.f;
      }
    }
    float acc = 0.f;
    bool near_zero = false;
#pragma unroll 1                                           // rows rolled (register pressure), columns unrolled
    for (int dy = 0; dy < BS; ++dy) {
      float mya = 1.f, myb = 1.f;
      if (BORDER) {
        const int py = qy + dy - HALF;
        mya = (float)tap_mult(py, qy, H, HALF);
        myb = (py >= 0 && py < H) ? (float)tap_mult(qy, py, H, HALF) : 0.f;
      }
#pragma unroll
      for (int dx = 0; dx < BS; ++dx) {
        const float gp = sG[ty + dy][tx + dx];
        if (TYPE == 0 || TYPE == 1) {
          acc = BORDER ? fmaf(mxa[dx] * mya, gp, acc) : acc + gp;
        } else {
          const float ep = sE[ty + dy][tx + dx], tp = sT[ty + dy][tx + dx];
          const float des = eq - ep, dta = tq - tp;                // tap q seen from centre p
          const float r1 = __builtin_amdgcn_rsqf(fmaf(des, des, eps)), r2 = __builtin_amdgcn_rsqf(fmaf(dta, dta, eps));
          float d2 = des * r1 - dta * r2;                          // 2 * (h(des) - h(dta))
          const float dh = r1 * r1 * r1;                           // (des^2 + eps)^(-3/2)
          const float w1 = BORDER ? mxa[dx] * mya * gp : gp;       // q as a tap of centre p
          const float w2 = BORDER ? mxb[dx] * myb * gq : gq;       // p as a tap of centre q
          if (TYPE == 2) {
            acc = fmaf(d2 * dh, w1 + w2, acc);                     // gl = 2 * diff = d2, K(q,p) = -K(p,q)
          } else {
            const float s1 = d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f);   // sign(diff); K(q,p) = -K(p,q)
            acc = fmaf(dh * s1, w1 + w2, acc);
            if (dx != HALF || dy != HALF) near_zero = near_zero || fabsf(d2) < 2e-6f;
          }
        }
      }
    }
    if (TYPE == 3 && near_zero) {
      // Rare (a few pixels in 10^4): some diff is too close to zero for the fast sign to be trusted.  Find those
      // pairs again and replace both of their signs by the ones the reference's own arithmetic gives (near
      // zero h(-x) - h(-y) need not be the exact negative of h(x) - h(y)).  Rolled loops: this path stays small.
#pragma unroll 1
      for (int dy = 0; dy < BS; ++dy)
#pragma unroll 1
        for (int dx = 0; dx < BS; ++dx) {
          const float ep = sE[ty + dy][tx + dx], tp = sT[ty + dy][tx + dx];
          const float des = eq - ep, dta = tq - tp;
          const float r1 = __builtin_amdgcn_rsqf(fmaf(des, des, eps)), r2 = __builtin_amdgcn_rsqf(fmaf(dta, dta, eps));
          const float d2 = des * r1 - dta * r2;
          if (!(fabsf(d2) < 2e-6f) || (dx == HALF && dy == HALF)) continue;
          const float gp = sG[ty + dy][tx + dx];
          const int px = qx + dx - HALF, py = qy + dy - HALF;
          const float w1 = (float)(tap_mult(px, qx, W, HALF) * tap_mult(py, qy, H, HALF)) * gp;
          const float w2 = (px >= 0 && px < W && py >= 0 && py < H)
                               ? (float)(tap_mult(qx, px, W, HALF) * tap_mult(qy, py, H, HALF)) * gq : 0.f;
          const float sf = d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f);          // what the main loop used
          const float a = soft_step_ref(des, eps) - soft_step_ref(dta, eps);
          const float b = soft_step_ref(-des, eps) - soft_step_ref(-dta, eps);
          const float s1 = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
          const float s2 = b > 0.f ? 1.f : (b < 0.f ? -1.f : 0.f);
          acc = fmaf(r1 * r1 * r1, (s1 - sf) * w1 - (s2 + sf) * w2, acc);
        }
    }
    float g;
    if (TYPE == 0) g = 2.f * (eq - tq) * acc / (float)(BS * BS);
    else if (TYPE == 1) g = (eq < tq ? -acc : (eq > tq ? acc : 0.f)) / (float)(BS * BS);
    else g = acc * (0.5f * eps / (float)(BS * BS));
    grad_plane[(long)qy * W + qx] = g;
  }
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void photometric_fast_bwd_kernel(const float* __restrict__ es,
                                                                   const float* __restrict__ ta,
                                                                   const float* __restrict__ grad_out,
                                                                   float* __restrict__ grad_in, int C, int H, int W,
                                                                   float eps) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1;
  __shared__ float sE[TH][TW], sT[TH][TW], sG[TH][TW];
  const int x0 = blockIdx.x * kPTW, y0 = blockIdx.y * kPTH, n = blockIdx.z;
  const long HW = (long)H * W;
  // a tile whose pixels all lie at least 2*HALF from the image border only meets multiplicities of 1
  const bool interior = x0 >= 2 * HALF && y0 >= 2 * HALF && x0 + kPTW - 1 <= W - 1 - 2 * HALF &&
                        y0 + kPTH - 1 <= H - 1 - 2 * HALF;
  stage_tile<BS>(sG, grad_out + (long)n * HW, H, W, x0, y0);
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    stage_tile<BS>(sE, es + ((long)n * C + c) * HW, H, W, x0, y0);
    stage_tile<BS>(sT, ta + ((long)n * C + c) * HW, H, W, x0, y0);
    __syncthreads();
    float* plane = grad_in + ((long)n * C + c) * HW;
    if (interior) bwd_tile<TYPE, BS, false>(sE, sT, sG, plane, H, W, x0, y0, eps);
    else bwd_tile<TYPE, BS, true>(sE, sT, sG, plane, H, W, x0, y0, eps);
  }
}

template <int TYPE, int BS>
static int launch_fast(bool bwd, const float* es, const float* ta, const float* go, float* dst, int B, int C, int H,
                       int W, float eps, hipStream_t stream) {
  const dim3 grid(ceil_div(W, kPTW), ceil_div(H, kPTH), B);
  if (bwd)
    hipLaunchKernelGGL((photometric_fast_bwd_kernel<TYPE, BS>), grid, dim3(256), 0, stream, es, ta, go, dst, C, H, W, eps);
  else
    hipLaunchKernelGGL((photometric_fast_fwd_kernel<TYPE, BS>), grid, dim3(256), 0, stream, es, ta, dst, C, H, W, eps);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <int BS>
static int dispatch_type(bool bwd, int type, const float* es, const float* ta, const float* go, float* dst, int B, int C,
                         int H, int W, float eps, hipStream_t s) {
  switch (type) {
    case 0: return launch_fast<0, BS>(bwd, es, ta, go, dst, B, C, H, W, eps, s);
    case 1: return launch_fast<1, BS>(bwd, es, ta, go, dst, B, C, H, W, eps, s);
    case 2: return launch_fast<2, BS>(bwd, es, ta, go, dst, B, C, H, W, eps, s);
    case 3: return launch_fast<3, BS>(bwd, es, ta, go, dst, B, C, H, W, eps, s);
    default: return CTD_ERR_INVALID_ARG;
  }
}

static int dispatch_fast(bool bwd, const float* es, const float* ta, const float* go, float* dst, int B, int C, int H,
                         int W, int bs, int type, float eps, hipStream_t s) {
  switch (bs) {                                            // odd block sizes only (symmetric window)
    case 3: return dispatch_type<3>(bwd, type, es, ta, go, dst, B, C, H, W, eps, s);
    case 5: return dispatch_type<5>(bwd, type, es, ta, go, dst, B, C, H, W, eps, s);
    case 7: return dispatch_type<7>(bwd, type, es, ta, go, dst, B, C, H, W, eps, s);
    case 9: return dispatch_type<9>(bwd, type, es, ta, go, dst, B, C, H, W, eps, s);
    default: return CTD_ERR_UNSUPPORTED;
  }
}

int photometric_fwd_fast_f32(const float* es, const float* ta, float* out, int B, int C, int H, int W, int bs, int type,
                             float eps, hipStream_t s) {
  return dispatch_fast(false, es, ta, nullptr, out, B, C, H, W, bs, type, eps, s);
}
int photometric_bwd_fast_f32(const float* es, const float* ta, const float* go, float* gi, int B, int C, int H, int W,
                             int bs, int type, float eps, hipStream_t s) {
  return dispatch_fast(true, es, ta, go, gi, B, C, H, W, bs, type, eps, s);
}

}  // namespace ctd
