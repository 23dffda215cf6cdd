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
#include <type_traits>

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

// block loss of the pixel at tile position (tx, ty), one channel
template <int TYPE, int BS>
__device__ inline float fwd_pixel(const float (*sE)[kPTW + BS - 1], const float (*sT)[kPTW + BS - 1], int tx, int ty,
                                  float eps) {
  constexpr int HALF = BS / 2;
  const float ec = sE[ty + HALF][tx + HALF], tc = sT[ty + HALF][tx + HALF];
  float acc = 0.f;
#pragma unroll 1                                         // rows rolled: a fully unrolled window hoists ~2*BS^2 LDS
  for (int dy = 0; dy < BS; ++dy)                         // loads into registers and leaves one wave per SIMD
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
  return acc * ((TYPE == 2 ? 0.25f : (TYPE == 3 ? 0.5f : 1.f)) / (float)(BS * BS));
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void photometric_fast_fwd_kernel(const float* __restrict__ es,
                                                                   const float* __restrict__ ta,
                                                                   float* __restrict__ out, int C, int H, int W,
                                                                   float eps) {
  constexpr int TW = kPTW + BS - 1, TH = kPTH + BS - 1;
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
    for (int k = 0; k < 2; ++k) loss[k] += fwd_pixel<TYPE, BS>(sE, sT, tx, ty0 + 4 * k, eps);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int x = x0 + tx, y = y0 + ty0 + 4 * k;
    if (x < W && y < H) out[(long)n * HW + (long)y * W + x] = loss[k];
  }
}

// gradient of the two pixels a thread owns in the staged tile; BORDER = general tap multiplicities
template <int TYPE, int BS, bool BORDER, typename Sink>
__device__ inline void bwd_tile(const float (*sE)[kPTW + BS - 1], const float (*sT)[kPTW + BS - 1],
                                const float (*sG)[kPTW + BS - 1], Sink&& sink, int H, int W, int x0, int y0,
                                float eps) {
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
    sink(qx, qy, g);
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
    auto sink = [&](int qx, int qy, float g) { plane[(long)qy * W + qx] = g; };
    if (interior) bwd_tile<TYPE, BS, false>(sE, sT, sG, sink, H, W, x0, y0, eps);
    else bwd_tile<TYPE, BS, true>(sE, sT, sG, sink, H, W, x0, y0, eps);
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

// ------------------------------------------------------------------------------------------------------
// Fused pattern similarity loss: RectifiedPatternSimilarityLoss.tforward,
// /root/reference/model/networks.py:358-378, as one forward and one backward kernel.
//   u1 = u - disp; gx = 2*(u1/(W-1) - 0.5); gy = 2*(v/(H-1) - 0.5)                       (:362-369)
//   pattern_proj = grid_sample(pattern, (gx, gy), bilinear, border, align_corners=False)   (:371)
//   diff = photometric_loss(pattern_proj, im, 9, type, eps); val = sum(mask*diff)/sum(mask) (:376-377)
// The warped pattern is sampled straight into the LDS tile (halo included: the block loss reads
// replicate-clamped taps of pattern_proj, i.e. the sample at the clamped pixel); it is written once because
// the module returns it.  Backward recomputes the tile, runs the pair-symmetric block-loss backward and
// applies d pattern_proj / d disp = -(W/(W-1)) * d/dix of the bilinear interpolant (0 where ATen clips).
// ------------------------------------------------------------------------------------------------------
struct WarpSample {
  float value, d_ddisp;
};

// ATen grid_sampler_2d, bilinear / border / align_corners=false, for the grid networks.py builds
__device__ inline WarpSample warp_pattern(const float* __restrict__ pat, int H, int W, int x, int y, float disp) {
  const float u1 = (float)x - disp;
  const float gx = 2.f * (u1 / (float)(W - 1) - 0.5f), gy = 2.f * ((float)y / (float)(H - 1) - 0.5f);
  float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;   // unnormalize
  // clip_coordinates_set_grad: gradient 0 at and beyond the borders
  float gmul = 1.f;
  if (ix <= 0.f) { ix = 0.f; gmul = 0.f; }
  else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); gmul = 0.f; }
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  const float wx1 = ix - fx, wx0 = 1.f - wx1, wy1 = iy - fy, wy0 = 1.f - wy1;
  const bool xin = x1 <= W - 1, yin = y1 <= H - 1;      // x0, y0 are inside after clipping
  const float p00 = pat[(long)y0 * W + x0];
  const float p01 = xin ? pat[(long)y0 * W + x1] : 0.f;
  const float p10 = yin ? pat[(long)y1 * W + x0] : 0.f;
  const float p11 = (xin && yin) ? pat[(long)y1 * W + x1] : 0.f;
  WarpSample r;
  r.value = p00 * (wx0 * wy0) + p01 * (wx1 * wy0) + p10 * (wx0 * wy1) + p11 * (wx1 * wy1);
  const float dv_dix = (p01 - p00) * wy0 + (p11 - p10) * wy1;
  // d ix / d gx = W/2, d gx / d u1 = 2/(W-1), d u1 / d disp = -1
  r.d_ddisp = -gmul * dv_dix * ((float)W / (float)(W - 1));
  return r;
}

template <int BS>
__device__ inline void stage_warped(float (*dst)[kPTW + BS - 1], const float* __restrict__ pat,
                                    const float* __restrict__ disp, int H, int W, int x0, int y0) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1;
  for (int i = threadIdx.x; i < TW * TH; i += 256) {
    const int r = i / TW, c = i - r * TW;
    const int y = clampi(y0 + r - HALF, 0, H - 1), x = clampi(x0 + c - HALF, 0, W - 1);
    dst[r][c] = warp_pattern(pat, H, W, x, y, disp[(long)y * W + x]).value;
  }
}

// (sum mask*diff, sum mask) over the tile's pixels, fixed-order tree; the result is valid in thread 0
template <int TYPE, int BS>
__device__ inline float2 pattern_fwd_tile(const float* __restrict__ disp, const float* __restrict__ im,
                                          const float* __restrict__ mask, const float* __restrict__ pattern,
                                          float* __restrict__ proj_out, int H, int W, int x0, int y0, int n,
                                          float eps) {
  constexpr int TW = kPTW + BS - 1, TH = kPTH + BS - 1, HALF = BS / 2;
  __shared__ float sE[TH][TW], sT[TH][TW];
  __shared__ float2 red[256];
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
  const long HW = (long)H * W;
  stage_warped<BS>(sE, pattern, disp + (long)n * HW, H, W, x0, y0);
  stage_tile<BS>(sT, im + (long)n * HW, H, W, x0, y0);
  __syncthreads();
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ty = ty0 + 4 * k, x = x0 + tx, y = y0 + ty;
    if (x < W && y < H) {
      const float diff = fwd_pixel<TYPE, BS>(sE, sT, tx, ty, eps);
      const long o = (long)n * HW + (long)y * W + x;
      const float m = mask ? mask[o] : 1.f;
      acc.x = fmaf(m, diff, acc.x);
      acc.y += m;
      proj_out[o] = sE[ty + HALF][tx + HALF];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int stride = 128; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) {
      red[threadIdx.x].x += red[threadIdx.x + stride].x;
      red[threadIdx.x].y += red[threadIdx.x + stride].y;
    }
    __syncthreads();
  }
  return red[0];
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void pattern_loss_fwd_kernel(const float* __restrict__ disp,
                                                               const float* __restrict__ im,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ pattern,
                                                               float* __restrict__ proj_out,
                                                               float2* __restrict__ partials, int H, int W,
                                                               float eps) {
  const float2 r = pattern_fwd_tile<TYPE, BS>(disp, im, mask, pattern, proj_out, H, W, blockIdx.x * kPTW,
                                              blockIdx.y * kPTH, blockIdx.z, eps);
  if (threadIdx.x == 0) partials[((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = r;
}

// ---- several pyramid levels in one launch (the training loop calls the loss once per scale on
// shrinking images, exp_synph.py:107-111; the 60x80 levels are launch-bound).  The level table travels in the
// kernel arguments; a workgroup finds its level from its linear index.
constexpr int kMaxLevels = 8;
struct PatternLevelDev {
  const float *disp, *im, *mask, *pattern, *grad_proj;
  float *proj, *grad_disp;
  int B, H, W, tiles_x, tiles_y;
  unsigned block_begin;                      // first workgroup (== first partial) of this level
};
struct PatternLevelsDev {
  PatternLevelDev lv[kMaxLevels];
  int n;
};

__device__ inline int find_level(const PatternLevelsDev& t, unsigned block, int& bx, int& by, int& n) {
  int l = 0;
#pragma unroll
  for (int k = 1; k < kMaxLevels; ++k)
    if (k < t.n && block >= t.lv[k].block_begin) l = k;
  const unsigned local = block - t.lv[l].block_begin;
  bx = (int)(local % t.lv[l].tiles_x);
  by = (int)((local / t.lv[l].tiles_x) % t.lv[l].tiles_y);
  n = (int)(local / ((unsigned)t.lv[l].tiles_x * t.lv[l].tiles_y));
  return l;
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void pattern_loss_multi_fwd_kernel(PatternLevelsDev t, float2* __restrict__ partials,
                                                                     float eps) {
  int bx, by, n;
  const int l = find_level(t, blockIdx.x, bx, by, n);
  const PatternLevelDev& L = t.lv[l];
  const float2 r = pattern_fwd_tile<TYPE, BS>(L.disp, L.im, L.mask, L.pattern, L.proj, L.H, L.W, bx * kPTW, by * kPTH, n, eps);
  if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// one workgroup per level: terms[level][3]
__global__ __launch_bounds__(256) void pattern_loss_multi_finish_kernel(PatternLevelsDev t, unsigned total_blocks,
                                                                        const float2* __restrict__ partials,
                                                                        float* __restrict__ terms) {
  __shared__ double rx[256], ry[256];
  const int l = blockIdx.x;
  const unsigned lo = t.lv[l].block_begin, hi = l + 1 < t.n ? t.lv[l + 1].block_begin : total_blocks;
  double ax = 0, ay = 0;
  for (unsigned i = lo + threadIdx.x; i < hi; i += 256) { ax += (double)partials[i].x; ay += (double)partials[i].y; }
  rx[threadIdx.x] = ax;
  ry[threadIdx.x] = ay;
  __syncthreads();
  for (int stride = 128; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) { rx[threadIdx.x] += rx[threadIdx.x + stride]; ry[threadIdx.x] += ry[threadIdx.x + stride]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    terms[3 * l + 0] = (float)rx[0];
    terms[3 * l + 1] = (float)ry[0];
    terms[3 * l + 2] = (float)rx[0] / (float)ry[0];
  }
}

// terms[0] = numerator, terms[1] = denominator, terms[2] = numerator / denominator; one workgroup, fixed order
__global__ __launch_bounds__(256) void pattern_loss_finish_kernel(const float2* __restrict__ partials, long n,
                                                                  float* __restrict__ terms) {
  __shared__ double rx[256], ry[256];
  double ax = 0, ay = 0;
  for (long i = threadIdx.x; i < n; i += 256) { ax += (double)partials[i].x; ay += (double)partials[i].y; }
  rx[threadIdx.x] = ax;
  ry[threadIdx.x] = ay;
  __syncthreads();
  for (int stride = 128; stride > 0; stride >>= 1) {
    if (threadIdx.x < stride) { rx[threadIdx.x] += rx[threadIdx.x + stride]; ry[threadIdx.x] += ry[threadIdx.x + stride]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    terms[0] = (float)rx[0];
    terms[1] = (float)ry[0];
    terms[2] = (float)rx[0] / (float)ry[0];
  }
}

// grad_disp = d val / d disp for val = terms[0] / terms[1]:  go[p] = grad_val * mask[p] / terms[1]
// (+ optionally grad_proj, the gradient arriving at the returned pattern_proj)
template <int TYPE, int BS>
__device__ inline void pattern_bwd_tile(const float* __restrict__ disp, const float* __restrict__ im,
                                        const float* __restrict__ mask, const float* __restrict__ pattern,
                                        float scale, const float* __restrict__ grad_proj,
                                        float* __restrict__ grad_disp, int H, int W, int x0, int y0, int n, float eps) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1;
  __shared__ float sE[TH][TW], sT[TH][TW], sG[TH][TW];
  const long HW = (long)H * W;
  const float* dsp = disp + (long)n * HW;
  stage_warped<BS>(sE, pattern, dsp, H, W, x0, y0);
  stage_tile<BS>(sT, im + (long)n * HW, H, W, x0, y0);
  if (mask) {
    stage_tile<BS>(sG, mask + (long)n * HW, H, W, x0, y0);
    __syncthreads();
    for (int i = threadIdx.x; i < TW * TH; i += 256) (&sG[0][0])[i] *= scale;
  } else {
    for (int i = threadIdx.x; i < TW * TH; i += 256) (&sG[0][0])[i] = scale;
  }
  __syncthreads();
  const bool interior = x0 >= 2 * HALF && y0 >= 2 * HALF && x0 + kPTW - 1 <= W - 1 - 2 * HALF &&
                        y0 + kPTH - 1 <= H - 1 - 2 * HALF;
  auto sink = [&](int qx, int qy, float g) {
    const long o = (long)qy * W + qx;
    if (grad_proj) g += grad_proj[(long)n * HW + o];
    grad_disp[(long)n * HW + o] = g * warp_pattern(pattern, H, W, qx, qy, dsp[o]).d_ddisp;
  };
  if (interior) bwd_tile<TYPE, BS, false>(sE, sT, sG, sink, H, W, x0, y0, eps);
  else bwd_tile<TYPE, BS, true>(sE, sT, sG, sink, H, W, x0, y0, eps);
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void pattern_loss_bwd_kernel(const float* __restrict__ disp,
                                                               const float* __restrict__ im,
                                                               const float* __restrict__ mask,
                                                               const float* __restrict__ pattern,
                                                               const float* __restrict__ terms,
                                                               const float* __restrict__ grad_val,
                                                               const float* __restrict__ grad_proj,
                                                               float* __restrict__ grad_disp, int H, int W,
                                                               float eps) {
  pattern_bwd_tile<TYPE, BS>(disp, im, mask, pattern, grad_val[0] / terms[1], grad_proj, grad_disp, H, W,
                             blockIdx.x * kPTW, blockIdx.y * kPTH, blockIdx.z, eps);
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void pattern_loss_multi_bwd_kernel(PatternLevelsDev t, const float* __restrict__ terms,
                                                                     const float* __restrict__ grad_vals, float eps) {
  int bx, by, n;
  const int l = find_level(t, blockIdx.x, bx, by, n);
  const PatternLevelDev& L = t.lv[l];
  pattern_bwd_tile<TYPE, BS>(L.disp, L.im, L.mask, L.pattern, grad_vals[l] / terms[3 * l + 1], L.grad_proj, L.grad_disp,
                             L.H, L.W, bx * kPTW, by * kPTH, n, eps);
}

size_t pattern_loss_workspace_bytes(int B, int H, int W) {
  return sizeof(float2) * (size_t)B * ceil_div(H, kPTH) * ceil_div(W, kPTW);
}

template <int TYPE>
static int pattern_loss_launch(bool bwd, const float* disp, const float* im, const float* mask, const float* pattern,
                               float* proj, float* terms, const float* grad_val, const float* grad_proj,
                               float* grad_disp, int B, int H, int W, float eps, void* ws, hipStream_t stream) {
  const dim3 grid(ceil_div(W, kPTW), ceil_div(H, kPTH), B);
  if (!bwd) {
    float2* partials = (float2*)ws;
    hipLaunchKernelGGL((pattern_loss_fwd_kernel<TYPE, 9>), grid, dim3(256), 0, stream, disp, im, mask, pattern, proj, partials,
                       H, W, eps);
    CTD_LAUNCH_CHECK();
    hipLaunchKernelGGL(pattern_loss_finish_kernel, dim3(1), dim3(256), 0, stream, partials,
                       (long)grid.x * grid.y * grid.z, terms);
  } else {
    hipLaunchKernelGGL((pattern_loss_bwd_kernel<TYPE, 9>), grid, dim3(256), 0, stream, disp, im, mask, pattern, terms,
                       grad_val, grad_proj, grad_disp, H, W, eps);
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

static int pattern_loss_dispatch(bool bwd, int type, const float* disp, const float* im, const float* mask,
                                 const float* pattern, float* proj, float* terms, const float* grad_val,
                                 const float* grad_proj, float* grad_disp, int B, int H, int W, float eps, void* ws,
                                 hipStream_t s) {
  switch (type) {
    case 0: return pattern_loss_launch<0>(bwd, disp, im, mask, pattern, proj, terms, grad_val, grad_proj, grad_disp, B, H, W, eps, ws, s);
    case 1: return pattern_loss_launch<1>(bwd, disp, im, mask, pattern, proj, terms, grad_val, grad_proj, grad_disp, B, H, W, eps, ws, s);
    case 2: return pattern_loss_launch<2>(bwd, disp, im, mask, pattern, proj, terms, grad_val, grad_proj, grad_disp, B, H, W, eps, ws, s);
    case 3: return pattern_loss_launch<3>(bwd, disp, im, mask, pattern, proj, terms, grad_val, grad_proj, grad_disp, B, H, W, eps, ws, s);
    default: return CTD_ERR_INVALID_ARG;
  }
}

int pattern_loss_fwd_f32(const float* disp, const float* im, const float* mask, const float* pattern, float* proj,
                         float* terms, int B, int H, int W, int type, float eps, void* ws, size_t ws_bytes,
                         hipStream_t s) {
  if (!ws || ws_bytes < pattern_loss_workspace_bytes(B, H, W)) return CTD_ERR_WORKSPACE;
  return pattern_loss_dispatch(false, type, disp, im, mask, pattern, proj, terms, nullptr, nullptr, nullptr, B, H, W, eps,
                               ws, s);
}
int pattern_loss_bwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                         const float* terms, const float* grad_val, const float* grad_proj, float* grad_disp, int B,
                         int H, int W, int type, float eps, hipStream_t s) {
  return pattern_loss_dispatch(true, type, disp, im, mask, pattern, nullptr, const_cast<float*>(terms), grad_val, grad_proj,
                               grad_disp, B, H, W, eps, nullptr, s);
}

static int build_levels(int n_levels, const ctd_pattern_level* levels, PatternLevelsDev& t, unsigned& total) {
  if (n_levels < 1 || n_levels > kMaxLevels || !levels) return CTD_ERR_INVALID_ARG;
  total = 0;
  t.n = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    const ctd_pattern_level& s = levels[l];
    if (s.B <= 0 || s.H < 2 || s.W < 2 || !s.disp || !s.im || !s.pattern) return CTD_ERR_INVALID_ARG;
    PatternLevelDev& d = t.lv[l];
    d.disp = s.disp; d.im = s.im; d.mask = s.mask; d.pattern = s.pattern; d.grad_proj = s.grad_proj;
    d.proj = s.pattern_proj; d.grad_disp = s.grad_disp;
    d.B = s.B; d.H = s.H; d.W = s.W;
    d.tiles_x = ceil_div(s.W, kPTW);
    d.tiles_y = ceil_div(s.H, kPTH);
    d.block_begin = total;
    const double blocks = (double)d.tiles_x * d.tiles_y * s.B;
    if (total + blocks >= 2147483648.0) return CTD_ERR_INVALID_ARG;
    total += (unsigned)blocks;
  }
  return CTD_OK;
}

size_t pattern_loss_multi_workspace_bytes(int n_levels, const ctd_pattern_level* levels) {
  PatternLevelsDev t;
  unsigned total = 0;
  if (build_levels(n_levels, levels, t, total)) return 0;
  return sizeof(float2) * (size_t)total;
}

int pattern_loss_multi_fwd_f32(int n_levels, const ctd_pattern_level* levels, float* terms, int type, float eps, void* ws,
                               size_t ws_bytes, hipStream_t stream) {
  PatternLevelsDev t;
  unsigned total = 0;
  int st = build_levels(n_levels, levels, t, total);
  if (st) return st;
  for (int l = 0; l < n_levels; ++l)
    if (!levels[l].pattern_proj) return CTD_ERR_INVALID_ARG;
  if (!ws || ws_bytes < sizeof(float2) * (size_t)total) return CTD_ERR_WORKSPACE;
  float2* partials = (float2*)ws;
  switch (type) {
    case 0: hipLaunchKernelGGL((pattern_loss_multi_fwd_kernel<0, 9>), dim3(total), dim3(256), 0, stream, t, partials, eps); break;
    case 1: hipLaunchKernelGGL((pattern_loss_multi_fwd_kernel<1, 9>), dim3(total), dim3(256), 0, stream, t, partials, eps); break;
    case 2: hipLaunchKernelGGL((pattern_loss_multi_fwd_kernel<2, 9>), dim3(total), dim3(256), 0, stream, t, partials, eps); break;
    case 3: hipLaunchKernelGGL((pattern_loss_multi_fwd_kernel<3, 9>), dim3(total), dim3(256), 0, stream, t, partials, eps); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  hipLaunchKernelGGL(pattern_loss_multi_finish_kernel, dim3(n_levels), dim3(256), 0, stream, t, total, partials, terms);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int pattern_loss_multi_bwd_f32(int n_levels, const ctd_pattern_level* levels, const float* terms, const float* grad_vals,
                               int type, float eps, hipStream_t stream) {
  PatternLevelsDev t;
  unsigned total = 0;
  int st = build_levels(n_levels, levels, t, total);
  if (st) return st;
  for (int l = 0; l < n_levels; ++l)
    if (!levels[l].grad_disp) return CTD_ERR_INVALID_ARG;
  switch (type) {
    case 0: hipLaunchKernelGGL((pattern_loss_multi_bwd_kernel<0, 9>), dim3(total), dim3(256), 0, stream, t, terms, grad_vals, eps); break;
    case 1: hipLaunchKernelGGL((pattern_loss_multi_bwd_kernel<1, 9>), dim3(total), dim3(256), 0, stream, t, terms, grad_vals, eps); break;
    case 2: hipLaunchKernelGGL((pattern_loss_multi_bwd_kernel<2, 9>), dim3(total), dim3(256), 0, stream, t, terms, grad_vals, eps); break;
    case 3: hipLaunchKernelGGL((pattern_loss_multi_bwd_kernel<3, 9>), dim3(total), dim3(256), 0, stream, t, terms, grad_vals, eps); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// ------------------------------------------------------------------------------------------------------
// Tolerance-level SAD / census cost volume: cost[f][d] = photometric_loss(P_d, I) with
// P_d[h][x] = P[h][clamp(x - d)] and the block loss's own replicate-clamped taps, i.e. the pattern tap of
// output (h, x), offset (dy, dx) is P[clamp(h+dy)][clamp(clamp(x+dx) - d)].  64x8 output tiles, the image tile
// and the pattern span of kCvChunk disparities in LDS; a thread keeps 8 disparities x 2 pixels of accumulators
// so that the image-side soft step of a tap is computed once for 8 disparities.
// ------------------------------------------------------------------------------------------------------
constexpr int kCvChunk = 32, kCvD = 8;

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void costvol_fast_kernel(const float* __restrict__ im, const float* __restrict__ pat,
                                                           long pat_frame_stride, float* __restrict__ cost, int H, int W,
                                                           int D, int n_chunks, float eps) {
  constexpr int HALF = BS / 2, TW = kPTW + BS - 1, TH = kPTH + BS - 1, SW = TW + kCvChunk - 1;
  __shared__ float sT[TH][TW], sP[TH][SW];
  const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
  const int x0 = blockIdx.x * kPTW, y0 = blockIdx.y * kPTH;
  const int f = blockIdx.z / n_chunks, d0 = (blockIdx.z - f * n_chunks) * kCvChunk;
  const long HW = (long)H * W;
  stage_tile<BS>(sT, im + (long)f * HW, H, W, x0, y0);
  // span column s of the tile holds pattern column (x0 - HALF - (kCvChunk - 1)) + s - d0, clamped (the second clamp)
  const float* p = pat + (long)f * pat_frame_stride;
  const int span_col0 = x0 - HALF - (kCvChunk - 1) - d0;
  for (int i = threadIdx.x; i < TH * SW; i += 256) {
    const int r = i / SW, c = i - r * SW;
    sP[r][c] = p[(long)clampi(y0 + r - HALF, 0, H - 1) * W + clampi(span_col0 + c, 0, W - 1)];
  }
  __syncthreads();
  const int x = x0 + tx;
  // first clamp of the tap column, tile relative: tap dx of pixel x sits at image column clamp(x + dx - HALF)
  int cx[BS];
#pragma unroll
  for (int dx = 0; dx < BS; ++dx) cx[dx] = clampi(x + dx - HALF, 0, W - 1) - x0 + HALF + (kCvChunk - 1);
  for (int db = 0; db < kCvChunk; db += kCvD) {
    if (d0 + db >= D) break;
    float acc[2][kCvD], ec[2][kCvD], tc[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      tc[k] = sT[ty0 + 4 * k + HALF][tx + HALF];
#pragma unroll
      for (int q = 0; q < kCvD; ++q) {
        acc[k][q] = 0.f;
        ec[k][q] = sP[ty0 + 4 * k + HALF][cx[HALF] - (db + q)];    // centre of P_d: P[y][clamp(x - d)]
      }
    }
#pragma unroll 1
    for (int dy = 0; dy < BS; ++dy)
#pragma unroll
      for (int dx = 0; dx < BS; ++dx)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int ty = ty0 + 4 * k;
          const float t = sT[ty + dy][tx + dx];
          float tb = 0.f;
          if (TYPE >= 2) {
            const float dta = t - tc[k];
            tb = dta * __builtin_amdgcn_rsqf(fmaf(dta, dta, eps));
          }
          const float* row = &sP[ty + dy][cx[dx] - db];
#pragma unroll
          for (int q = 0; q < kCvD; ++q) {
            const float e = row[-q];
            if (TYPE == 0) {
              const float df = e - t;
              acc[k][q] = fmaf(df, df, acc[k][q]);
            } else if (TYPE == 1) {
              acc[k][q] += fabsf(e - t);
            } else {
              const float des = e - ec[k][q];
              const float d2 = des * __builtin_amdgcn_rsqf(fmaf(des, des, eps)) - tb;   // 2 * (h(des) - h(dta))
              if (TYPE == 2) acc[k][q] = fmaf(d2, d2, acc[k][q]);
              else acc[k][q] += fabsf(d2);
            }
          }
        }
    const float scale = (TYPE == 2 ? 0.25f : (TYPE == 3 ? 0.5f : 1.f)) / (float)(BS * BS);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int y = y0 + ty0 + 4 * k;
      if (x < W && y < H) {
#pragma unroll
        for (int q = 0; q < kCvD; ++q) {
          const int d = d0 + db + q;
          if (d < D) cost[((long)f * D + d) * HW + (long)y * W + x] = acc[k][q] * scale;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Census TRANSFORM cost volume (types 2 / 3): the soft census term of a tap, h(P_d[tap] - P_d[centre]), does not
// depend on d away from the right image border -- with x' = x - d it is the pattern's own census value
//     CP(y, x'; dy, dx) = h(P[clamp(y+dy)][clamp(x'+dx)] - P[y][clamp(x')])        (x' may be negative: both clamps apply)
// so it is evaluated once per pattern column and tap, not once per output and tap (the v_rsq_f32 leaves the disparity
// loop: 81 per PATTERN PIXEL instead of 81 per output, D = 128..256 times fewer), and likewise CI(y, x; dy, dx) once
// per image pixel.  Per tap the workgroup stages CI for its 64 x 2 pixels and CP for the 64 + 127 pattern columns its
// 128 disparities reach (2 evaluations per thread and tap, three taps per barrier), then every thread accumulates its
// 4 pixels x 16 disparities:
//   census_sad: the values are staged as 24-bit FIXED POINT, u = round((t + 1) * 2^23) with t = des * rsq(des^2 + eps) in
//     (-1, 1), and one v_sad_u32 per output and tap does |u_p - u_i| + acc (exact integer sum, 81 * 2^24 < 2^32; the
//     rounding of a staged value is 2^-24, that of an f32 t 3e-8: the same accuracy);
//   census_mse: staged as floats, one subtract and one fma per output and tap.
// Only the HALF right-most image columns differ (there the tap column is clamped to W-1 BEFORE the shift by d, so the
// term does depend on d): the workgroups of the last tile column recompute those outputs term by term afterwards.
// Reference: torchext/ext/ext.h:244-259 (per-tap soft census), composition rule of SURVEY 8a/A6.
// ------------------------------------------------------------------------------------------------------
constexpr int kCcW = 64, kCcR = 2, kCcD = 128, kCcDT = 16;   // pixel tile, disparities per workgroup / per thread
constexpr int kCcTaps = 3;                                    // taps staged per barrier

__device__ inline unsigned sad_u32(unsigned a, unsigned b, unsigned acc) {   // |a - b| + acc in one VALU instruction (no builtin)
  unsigned r;
  asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
  return r;
}

template <int TYPE, int BS>
__global__ __launch_bounds__(256) void costvol_census_kernel(const float* __restrict__ im, const float* __restrict__ pat,
                                                             long pat_frame_stride, float* __restrict__ cost, int H, int W,
                                                             int D, int n_chunks, float eps) {
  static_assert(TYPE == 2 || TYPE == 3, "census types only");
  constexpr int HALF = BS / 2, TH = kCcR + BS - 1, TW = kCcW + BS - 1;
  constexpr int CPW = kCcW + kCcD;                  // staged pattern census columns j = x' - xp0, j in [0, CPW)
  constexpr int SPW = CPW + BS - 1;                 // raw pattern span: column xp0 - HALF + s
  typedef typename std::conditional<TYPE == 3, unsigned, float>::type cen_t;
  __shared__ float sI[TH][TW];
  __shared__ float sP[TH][SPW];
  constexpr int TS = kCcTaps;                       // taps staged per barrier
  __shared__ __attribute__((aligned(16))) cen_t cI[2][TS][kCcR][kCcW];
  __shared__ __attribute__((aligned(16))) cen_t cP[2][TS][kCcR][CPW];
  const int t = threadIdx.x;
  const int x0 = blockIdx.x * kCcW, y0 = blockIdx.y * kCcR;
  const int f = blockIdx.z / n_chunks, d0 = (blockIdx.z - f * n_chunks) * kCcD;
  const long HW = (long)H * W;
  const float* ip = im + (long)f * HW;
  const float* pp = pat + (long)f * pat_frame_stride;
  const int xp0 = x0 - d0 - kCcD;                   // pattern column of census slot 0 (may be negative)
  // raw tiles, both clamps baked in: image columns clamp(x0 - HALF + c), pattern columns clamp(xp0 - HALF + s)
  for (int i0 = t; i0 < TH * TW; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = min(i0 + 256 * u, TH * TW - 1);
      const int r = i / TW, c = i - r * TW;
      v[u] = ip[(long)clampi(y0 + r - HALF, 0, H - 1) * W + clampi(x0 + c - HALF, 0, W - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + 256 * u < TH * TW) (&sI[0][0])[i0 + 256 * u] = v[u];
  }
  for (int i0 = t; i0 < TH * SPW; i0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + 256 * u, TH * SPW - 1);
      const int r = i / SPW, c = i - r * SPW;
      v[u] = pp[(long)clampi(y0 + r - HALF, 0, H - 1) * W + clampi(xp0 + c - HALF, 0, W - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + 256 * u < TH * SPW) (&sP[0][0])[i0 + 256 * u] = v[u];
  }
  // this thread's outputs: pixel quad q of row `row`, disparities d0 + 16 g + k
  const int q = t & 15, row = (t >> 4) & 1, g = t >> 5;
  const int jb = 4 * q - kCcDT * g + kCcD - kCcDT;  // first staged pattern column of its five quads (multiple of 4)
  // its two census evaluations per tap: element e = t and t + 256 of [image 2 x 64 | pattern 2 x CPW]
  constexpr int NI = kCcR * kCcW;
  static_assert(NI + kCcR * CPW == 512, "two staged census values per thread and tap");
  typename std::conditional<TYPE == 3, unsigned, float>::type acc[4][kCcDT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < kCcDT; ++k) acc[i][k] = 0;
  __syncthreads();
  // centre values of the two elements (tap independent) and their tile coordinates
  const bool e0_img = t < NI;                       // (NI = 128: the first two wavefronts; wave-uniform)
  const int e0r = e0_img ? t / kCcW : (t - NI) / CPW, e0c = e0_img ? t % kCcW : (t - NI) % CPW;
  const int e1 = t + 256 - NI, e1r = e1 / CPW, e1c = e1 % CPW;
  const float c0 = e0_img ? sI[e0r + HALF][e0c + HALF] : sP[e0r + HALF][e0c + HALF];
  const float c1 = sP[e1r + HALF][e1c + HALF];
  auto soft = [&](float des) -> cen_t {
    const float tt = des * __builtin_amdgcn_rsqf(fmaf(des, des, eps));      // 2 h(des) - 1, in (-1, 1)
    if constexpr (TYPE == 3) return (unsigned)fmaf(tt, 8388608.f, 8388608.5f);   // round((tt + 1) * 2^23)
    else return tt;
  };
  int buf = 0;
  // taps in groups of TS per barrier (81 = 27 x 3 for block 9; a last partial group stages and accumulates fewer):
  // one barrier per group -- a thread that writes buffer b for group n + 2 has passed the barrier of group n + 1, i.e.
  // everybody finished reading group n
#pragma unroll 1
  for (int tap0 = 0; tap0 < BS * BS; tap0 += TS) {
#pragma unroll
    for (int u = 0; u < TS; ++u) {
      const int tap = tap0 + u;
      if (tap < BS * BS) {                           // (uniform)
        const int dy = tap / BS, dx = tap - dy * BS;
        const float v0 = e0_img ? sI[e0r + dy][e0c + dx] : sP[e0r + dy][e0c + dx];
        const float v1 = sP[e1r + dy][e1c + dx];
        if (e0_img) cI[buf][u][e0r][e0c] = soft(v0 - c0);
        else cP[buf][u][e0r][e0c] = soft(v0 - c0);
        cP[buf][u][e1r][e1c] = soft(v1 - c1);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < TS; ++u) {
      if (tap0 + u >= BS * BS) break;
      typedef cen_t c4 __attribute__((ext_vector_type(4)));
      const c4 ci = *(const c4*)&cI[buf][u][row][4 * q];
      cen_t cp[20];
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const c4 w4 = *(const c4*)&cP[buf][u][row][jb + 4 * m];
        cp[4 * m] = w4[0]; cp[4 * m + 1] = w4[1]; cp[4 * m + 2] = w4[2]; cp[4 * m + 3] = w4[3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < kCcDT; ++k) {
          if constexpr (TYPE == 3) {
            acc[i][k] = sad_u32(cp[i - k + kCcDT], ci[i], acc[i][k]);                    // |u_p - u_i| + acc
          } else {
            const float d2 = cp[i - k + kCcDT] - ci[i];
            acc[i][k] = fmaf(d2, d2, acc[i][k]);
          }
        }
    }
    buf ^= 1;
  }
  // 2 (h_p - h_i) = t_p - t_i: census_sad 0.5 / bs^2 (and 2^-23 for the fixed point), census_mse 0.25 / bs^2
  const float scale = TYPE == 3 ? 0.5f / (float)(BS * BS) / 8388608.f : 0.25f / (float)(BS * BS);
  const int y = y0 + row, xq = x0 + 4 * q;
  const int x_last_plain = W - 1 - (BS - 1 - HALF);  // right of it the first clamp makes the term depend on d
  if (y < H) {
#pragma unroll
    for (int k = 0; k < kCcDT; ++k) {
      const int d = d0 + kCcDT * g + k;
      if (d >= D) break;
      float* o = cost + ((long)f * D + d) * HW + (long)y * W + xq;
      if (xq + 3 <= x_last_plain && (W & 3) == 0 && ((uintptr_t)cost & 15) == 0) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        *(f4*)o = f4{(float)acc[0][k] * scale, (float)acc[1][k] * scale, (float)acc[2][k] * scale, (float)acc[3][k] * scale};
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (xq + i <= x_last_plain) o[i] = (float)acc[i][k] * scale;
      }
    }
  }
  // the right-most columns, term by term (ext.h:244-259 order of clamps: tap column first, shift second): the
  // workgroup's threads share them, pixel fastest
  const int xb0 = max(x0, x_last_plain + 1), nb = min(x0 + kCcW, W) - xb0;
  if (nb > 0) {
    const int nd = min(kCcD, D - d0);
    for (int o = t; o < nb * kCcR * nd; o += 256) {
      const int px = o % nb, r = (o / nb) % kCcR, dd = o / (nb * kCcR);
      const int x = xb0 + px, yy = y0 + r, d = d0 + dd;
      if (yy >= H) continue;
      // (span slots are addressed by the UNCLAMPED column, x - d >= xp0 + 1: the staged values carry the clamp)
      const float ec = sP[r + HALF][x - d - xp0 + HALF];
      const float tc = sI[r + HALF][x - x0 + HALF];
      float a = 0.f;
      for (int dy = 0; dy < BS; ++dy)
#pragma unroll
        for (int dx = 0; dx < BS; ++dx) {
          const int cw = min(x + dx - HALF, W - 1);                          // first clamp (x + dx - HALF >= 0 here)
          const float e = sP[r + dy][cw - d - xp0 + HALF];
          const float des = e - ec, dta = sI[r + dy][x - x0 + dx] - tc;
          const float d2 = des * __builtin_amdgcn_rsqf(fmaf(des, des, eps)) - dta * __builtin_amdgcn_rsqf(fmaf(dta, dta, eps));
          a = TYPE == 2 ? fmaf(d2, d2, a) : a + fabsf(d2);
        }
      cost[((long)f * D + d) * HW + (long)yy * W + x] = a * ((TYPE == 2 ? 0.25f : 0.5f) / (float)(BS * BS));
    }
  }
}

template <int BS>
static int costvol_census_type(int type, const float* im, const float* pat, long pat_frame_stride, float* cost, int frames,
                               int H, int W, int D, float eps, hipStream_t stream) {
  const int n_chunks = ceil_div(D, kCcD);
  const dim3 grid(ceil_div(W, kCcW), ceil_div(H, kCcR), frames * n_chunks);
  if (grid.y > 65535 || (long)frames * n_chunks > 65535) return CTD_ERR_INVALID_ARG;
  if (type == 2)
    hipLaunchKernelGGL((costvol_census_kernel<2, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps);
  else
    hipLaunchKernelGGL((costvol_census_kernel<3, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <int BS>
static int costvol_fast_type(int type, const float* im, const float* pat, long pat_frame_stride, float* cost, int frames,
                             int H, int W, int D, float eps, hipStream_t stream) {
  const int n_chunks = ceil_div(D, kCvChunk);
  const dim3 grid(ceil_div(W, kPTW), ceil_div(H, kPTH), frames * n_chunks);
  switch (type) {
    case 0: hipLaunchKernelGGL((costvol_fast_kernel<0, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps); break;
    case 1: hipLaunchKernelGGL((costvol_fast_kernel<1, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps); break;
    case 2: hipLaunchKernelGGL((costvol_fast_kernel<2, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps); break;
    case 3: hipLaunchKernelGGL((costvol_fast_kernel<3, BS>), grid, dim3(256), 0, stream, im, pat, pat_frame_stride, cost, H, W, D, n_chunks, eps); break;
    default: return CTD_ERR_INVALID_ARG;
  }
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int costvol_fast_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W,
                     int D, int bs, int type, float eps, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  // SAD / MSE, block 9: the sum is separable (a replicate-border box filter of |P[r][c - d] - I[r][c]|) -- the all-D
  // pipeline of ncc_fast.hip, one subtract per output instead of 81; needs the caller's workspace for the padded planes
  if (workspace && costvol_sep_supported(H, W, D, bs, type) && ((uintptr_t)cost) % 16 == 0 &&
      workspace_bytes >= costvol_sep_workspace_bytes(frames, H, W, D, pat_frame_stride != 0))
    return costvol_sep_f32(im, pat, pat_frame_stride, cost, frames, H, W, D, type, workspace, workspace_bytes, stream);
  if (type >= 2) {                                   // census types: the census-transform kernel
    switch (bs) {
      case 3: return costvol_census_type<3>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
      case 5: return costvol_census_type<5>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
      case 7: return costvol_census_type<7>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
      case 9: return costvol_census_type<9>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
      default: return CTD_ERR_UNSUPPORTED;
    }
  }
  if ((long)frames * ceil_div(D, kCvChunk) > 65535) return CTD_ERR_INVALID_ARG;
  switch (bs) {
    case 3: return costvol_fast_type<3>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
    case 5: return costvol_fast_type<5>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
    case 7: return costvol_fast_type<7>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
    case 9: return costvol_fast_type<9>(type, im, pat, pat_frame_stride, cost, frames, H, W, D, eps, stream);
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
