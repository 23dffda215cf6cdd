// losses.hip -- fused disparity-to-depth, edge-aware disparity loss and two-view geometric loss.
//
// Each replaces a chain of stock PyTorch ops of the reference (model/networks.py), forward and
// backward, with one or two kernels and deterministic two-stage reductions:
//   * DispToDepth.tforward              networks.py:313-321   (3 elementwise kernels + autograd)
//   * SobelFilter + DisparityLoss       networks.py:380-412, 537-565 (pad, 2 convs, ~10 elementwise, mean)
//   * ProjectionDepthSimilarityLoss.fwd networks.py:416-498   (2 bmm, normalise, grid_sample, abs, clamp, mean)
// The reference's arithmetic here is ATen's (conv / bmm / grid_sample summation orders are unspecified),
// so parity is by tolerance against vectors captured from the reference modules (tests/golden/losses.npz).
#include "ctd_internal.h"

namespace ctd {

// ------------------------------------------------------------------------------------------------
// block reduction helpers: wave shuffle -> LDS -> one partial per workgroup; partials are summed in a
// fixed order by a single-workgroup kernel (no float atomics: bitwise reproducible)
// ------------------------------------------------------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__device__ inline void block_partial(float v, float* __restrict__ partials) {
  __shared__ float s[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) s[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += s[k];
    partials[blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z)] = t;
  }
}

constexpr int kGeoGroup = 4;    // tiles per f64 unit sum of the geometric loss (finish_mean_kernel's `group` for it)

// `group`: partials are first summed (f64, index order) in groups of that many consecutive ones, thread t then adds groups
// t, t + 256, ... -- 1 for most losses; the geometric loss uses 4, the order its one-launch kernel sums in (a workgroup
// of four tiles publishes one f64 sum there)
__global__ __launch_bounds__(256) void finish_mean_kernel(const float* __restrict__ partials, long n, double count,
                                                          float* __restrict__ out, int accumulate, int group = 1) {
  __shared__ double s[256];
  double t = 0;
  if (group <= 1) {
    for (long i = threadIdx.x; i < n; i += 256) t += (double)partials[i];
  } else {
    for (long u = threadIdx.x; u * group < n; u += 256) {
      double q = 0;
      for (int k = 0; k < group && u * group + k < n; ++k) q += (double)partials[u * group + k];
      t += q;
    }
  }
  s[threadIdx.x] = t;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) s[threadIdx.x] += s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = (float)(s[0] / count);
    out[0] = accumulate ? out[0] + v : v;
  }
}

// ------------------------------------------------------------------------------------------------
// DispToDepth: depth = (1 / (relu(disp) + 1e-12)) * bf   ("bf / tensor" is reciprocal()*bf in torch)
// ------------------------------------------------------------------------------------------------
__global__ void d2d_fwd_kernel(const float* __restrict__ disp, float* __restrict__ depth, long n, float bf) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = fmaxf(disp[i], 0.f) + 1e-12f;
    depth[i] = (1.0f / x) * bf;
  }
}
__global__ void d2d_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ go, float* __restrict__ gi,
                               long n, float bf) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = disp[i];
    const float r = 1.0f / (fmaxf(d, 0.f) + 1e-12f);
    const float g = (go[i] * bf) * (-(r * r));
    gi[i] = d > 0.f ? g : 0.f;
  }
}

// same with the disparity given as the int64 argmax index plus a constant (disparity 0 would be depth 1e12)
__global__ void d2d_idx_kernel(const int64_t* __restrict__ idx, float* __restrict__ depth, long n, float bf, float offset) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = fmaxf((float)idx[i] + offset, 0.f) + 1e-12f;
    depth[i] = (1.0f / x) * bf;
  }
}
int idx_to_depth_f32(const int64_t* idx, float* depth, long n, float bf, float offset, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(d2d_idx_kernel, dim3(blocks), dim3(256), 0, s, idx, depth, n, bf, offset);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int disp_to_depth_fwd_f32(const float* disp, float* depth, long n, float bf, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(d2d_fwd_kernel, dim3(blocks), dim3(256), 0, s, disp, depth, n, bf);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}
int disp_to_depth_bwd_f32(const float* disp, const float* go, float* gi, long n, float bf, hipStream_t s) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(d2d_bwd_kernel, dim3(blocks), dim3(256), 0, s, disp, go, gi, n, bf);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// ------------------------------------------------------------------------------------------------
// Sobel 5x5 (networks.py:543-548, /240, ky = kx^T) on the replicate-padded disparity + DisparityLoss
// ------------------------------------------------------------------------------------------------
__constant__ float kSobel[5][5] = {{-5.f / 240, -4.f / 240, 0.f, 4.f / 240, 5.f / 240},
                                   {-8.f / 240, -10.f / 240, 0.f, 10.f / 240, 8.f / 240},
                                   {-10.f / 240, -20.f / 240, 0.f, 20.f / 240, 10.f / 240},
                                   {-8.f / 240, -10.f / 240, 0.f, 10.f / 240, 8.f / 240},
                                   {-5.f / 240, -4.f / 240, 0.f, 4.f / 240, 5.f / 240}};
constexpr float kB0 = 0.0503428816795f, kB1 = 1.07274045944f;   // networks.py:389-390

__device__ inline void sobel_at(const float* __restrict__ x, int H, int W, int h, int w, float& gx, float& gy) {
  gx = 0.f;
  gy = 0.f;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int hh = clampi(h + i - 2, 0, H - 1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float v = x[(long)hh * W + clampi(w + j - 2, 0, W - 1)];
      gx = fmaf(kSobel[i][j], v, gx);
      gy = fmaf(kSobel[j][i], v, gy);
    }
  }
}

// per-pixel loss term and its derivatives w.r.t. the gradient magnitude g and the edge probability e
__device__ inline float disparity_term(float g, bool has_edge, float e, float& dLdg, float& dLde) {
  if (has_edge) {
    const float A = expf(-g / kB0), Bq = expf(-g / kB1);
    const float pdf = (1.f - e) / kB0 * A + e / kB1 * Bq;
    const bool pass = pdf >= 1e-4f;                      // clamp(min=1e-4): gradient where pdf >= min
    const float pc = pass ? pdf : 1e-4f;
    const float dLdp = pass ? -1.f / pdf : 0.f;
    dLdg = dLdp * (-(1.f - e) / (kB0 * kB0) * A - e / (kB1 * kB1) * Bq);
    dLde = dLdp * (-A / kB0 + Bq / kB1);
    return -logf(pc);
  }
  dLde = 0.f;
  dLdg = (g >= 0.f && g <= 1.f) ? 1.f : 0.f;             // clamp(g, 0, 1)
  return fminf(fmaxf(g, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void disparity_loss_fwd_kernel(const float* __restrict__ disp,
                                                                 const float* __restrict__ edge,
                                                                 float* __restrict__ partials, int H, int W) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6);
  const long plane = (long)blockIdx.z * H * W;
  float term = 0.f;
  if (w < W && h < H) {
    float gx, gy, dg, de;
    sobel_at(disp + plane, H, W, h, w, gx, gy);
    const float g = sqrtf(gx * gx + gy * gy + 1e-8f);
    term = disparity_term(g, edge != nullptr, edge ? edge[plane + (long)h * W + w] : 0.f, dg, de);
  }
  block_partial(term, partials);
}

// backward stage 1: per output pixel, d loss / d gx and / d gy (scaled by grad_out / N), d loss / d edge
__global__ __launch_bounds__(256) void disparity_loss_bwd1_kernel(const float* __restrict__ disp,
                                                                  const float* __restrict__ edge,
                                                                  const float* __restrict__ grad_loss, float inv_n,
                                                                  float* __restrict__ ggx, float* __restrict__ ggy,
                                                                  float* __restrict__ grad_edge, int H, int W) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (w >= W || h >= H) return;
  const long plane = (long)blockIdx.z * H * W, o = plane + (long)h * W + w;
  float gx, gy, dg, de;
  sobel_at(disp + plane, H, W, h, w, gx, gy);
  const float g = sqrtf(gx * gx + gy * gy + 1e-8f);
  disparity_term(g, edge != nullptr, edge ? edge[o] : 0.f, dg, de);
  const float sc = grad_loss[0] * inv_n;
  ggx[o] = sc * dg * gx / g;
  ggy[o] = sc * dg * gy / g;
  if (grad_edge) grad_edge[o] = sc * de;
}

// backward stage 2: gather through the two transposed 5x5 filters and the replicate padding
__global__ __launch_bounds__(256) void disparity_loss_bwd2_kernel(const float* __restrict__ ggx,
                                                                  const float* __restrict__ ggy,
                                                                  float* __restrict__ grad_disp, int H, int W) {
  const int w0 = blockIdx.x * 64 + (threadIdx.x & 63), h0 = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (w0 >= W || h0 >= H) return;
  const long plane = (long)blockIdx.z * H * W;
  float acc = 0.f;
  for (int h = max(0, h0 - 2); h <= min(H - 1, h0 + 2); ++h) {
    // filter rows i of output row h that read (clamped) input row h0
    int i_lo = h0 - h + 2, i_hi = i_lo;
    if (h0 == 0) i_lo = 0;
    if (h0 == H - 1) i_hi = 4;
    i_lo = max(i_lo, 0);
    i_hi = min(i_hi, 4);
    for (int w = max(0, w0 - 2); w <= min(W - 1, w0 + 2); ++w) {
      int j_lo = w0 - w + 2, j_hi = j_lo;
      if (w0 == 0) j_lo = 0;
      if (w0 == W - 1) j_hi = 4;
      j_lo = max(j_lo, 0);
      j_hi = min(j_hi, 4);
      const float a = ggx[plane + (long)h * W + w], b = ggy[plane + (long)h * W + w];
      for (int i = i_lo; i <= i_hi; ++i)
        for (int j = j_lo; j <= j_hi; ++j) acc += a * kSobel[i][j] + b * kSobel[j][i];
    }
  }
  grad_disp[plane + (long)h0 * W + w0] = acc;
}

static long loss_grid_blocks(int B, int H, int W) { return (long)ceil_div(W, 64) * ceil_div(H, 4) * B; }

size_t disparity_loss_workspace_bytes(int B, int H, int W) {
  size_t partials = align_up(sizeof(float) * (size_t)loss_grid_blocks(B, H, W), 256);
  size_t planes = align_up(sizeof(float) * (size_t)B * H * W, 256);
  return partials + 2 * planes;                       // forward uses the partials, backward the two planes
}

int disparity_loss_fwd_f32(const float* disp, const float* edge, float* loss, int B, int H, int W, void* ws,
                           size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < disparity_loss_workspace_bytes(B, H, W)) return CTD_ERR_WORKSPACE;
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B), block(256);
  float* partials = (float*)ws;
  hipLaunchKernelGGL(disparity_loss_fwd_kernel, grid, block, 0, s, disp, edge, partials, H, W);
  CTD_LAUNCH_CHECK();
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(256), 0, s, partials, loss_grid_blocks(B, H, W),
                     (double)B * H * W, loss, 0);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int disparity_loss_bwd_f32(const float* disp, const float* edge, const float* grad_loss, float* grad_disp,
                           float* grad_edge, int B, int H, int W, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < disparity_loss_workspace_bytes(B, H, W)) return CTD_ERR_WORKSPACE;
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B), block(256);
  size_t partials = align_up(sizeof(float) * (size_t)loss_grid_blocks(B, H, W), 256);
  size_t planes = align_up(sizeof(float) * (size_t)B * H * W, 256);
  float* ggx = (float*)((char*)ws + partials);
  float* ggy = (float*)((char*)ws + partials + planes);
  hipLaunchKernelGGL(disparity_loss_bwd1_kernel, grid, block, 0, s, disp, edge, grad_loss,
                     (float)(1.0 / ((double)B * H * W)), ggx, ggy, grad_edge, H, W);
  CTD_LAUNCH_CHECK();
  hipLaunchKernelGGL(disparity_loss_bwd2_kernel, grid, block, 0, s, ggx, ggy, grad_disp, H, W);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// ------------------------------------------------------------------------------------------------
// Geometric loss, one direction (ProjectionDepthSimilarityLoss.fwd, networks.py:483-498):
//   xyz = depth0 * ray; xyz = (xyz - t0) @ R0; xyz = xyz @ R1^T + t1; uvd = xyz @ K^T
//   uv = uvd[:2] / (relu(d) + 1e-12); normalise with (W-1), (H-1); grid_sample(depth1, bilinear, border,
//   align_corners=False); mean(clamp(|d - sample|, 0, clamp))
// ------------------------------------------------------------------------------------------------
struct Pose {
  float R0[9], t0[3], R1[9], t1[3], K[9];
};

__device__ inline Pose load_pose(const float* K, const float* R0, const float* t0, const float* R1, const float* t1,
                                 int b) {
  Pose p;
#pragma unroll
  for (int i = 0; i < 9; ++i) { p.R0[i] = R0[b * 9 + i]; p.R1[i] = R1[b * 9 + i]; p.K[i] = K[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { p.t0[i] = t0[b * 3 + i]; p.t1[i] = t1[b * 3 + i]; }
  return p;
}

struct GeoPoint {
  float ray[3], s[3], uvd[3], den, ix, iy;     // intermediate values kept for the backward
  float wx1, wy1, sample, diff;
  int x0, y0;
  bool clip_x, clip_y;
};

__device__ inline GeoPoint geo_forward(const Pose& P, const float* __restrict__ ray3, float depth0,
                                       const float* __restrict__ depth1, int H, int W) {
  GeoPoint g;
  float p[3], q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { g.ray[i] = ray3[i]; p[i] = depth0 * g.ray[i] - P.t0[i]; }
#pragma unroll
  for (int j = 0; j < 3; ++j) q[j] = p[0] * P.R0[0 * 3 + j] + p[1] * P.R0[1 * 3 + j] + p[2] * P.R0[2 * 3 + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) g.s[j] = q[0] * P.R1[j * 3 + 0] + q[1] * P.R1[j * 3 + 1] + q[2] * P.R1[j * 3 + 2] + P.t1[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) g.uvd[j] = g.s[0] * P.K[j * 3 + 0] + g.s[1] * P.K[j * 3 + 1] + g.s[2] * P.K[j * 3 + 2];
  g.den = fmaxf(g.uvd[2], 0.f) + 1e-12f;
  const float u = g.uvd[0] / g.den, v = g.uvd[1] / g.den;
  const float un = 2.f * (u / (float)(W - 1) - 0.5f), vn = 2.f * (v / (float)(H - 1) - 0.5f);
  float ix = ((un + 1.f) * (float)W - 1.f) * 0.5f, iy = ((vn + 1.f) * (float)H - 1.f) * 0.5f;   // align_corners=False
  g.clip_x = !(ix >= 0.f && ix <= (float)(W - 1));      // border padding clips the coordinate, gradient 0 there
  g.clip_y = !(iy >= 0.f && iy <= (float)(H - 1));
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  g.ix = ix;
  g.iy = iy;
  const float fx = floorf(ix), fy = floorf(iy);
  g.x0 = (int)fx;
  g.y0 = (int)fy;
  g.wx1 = ix - fx;
  g.wy1 = iy - fy;
  const int x1 = g.x0 + 1, y1 = g.y0 + 1;
  const float nw = depth1[(long)g.y0 * W + g.x0];
  const float ne = x1 < W ? depth1[(long)g.y0 * W + x1] : 0.f;
  const float sw = y1 < H ? depth1[(long)y1 * W + g.x0] : 0.f;
  const float se = (x1 < W && y1 < H) ? depth1[(long)y1 * W + x1] : 0.f;
  g.sample = nw * (1.f - g.wx1) * (1.f - g.wy1) + ne * g.wx1 * (1.f - g.wy1) + sw * (1.f - g.wx1) * g.wy1 +
             se * g.wx1 * g.wy1;
  g.diff = fabsf(g.uvd[2] - g.sample);
  return g;
}

__global__ __launch_bounds__(256) void geometric_fwd_kernel(const float* __restrict__ depth0,
                                                            const float* __restrict__ depth1,
                                                            const float* __restrict__ ray, const float* __restrict__ K,
                                                            const float* __restrict__ R0, const float* __restrict__ t0,
                                                            const float* __restrict__ R1, const float* __restrict__ t1,
                                                            float* __restrict__ partials, int H, int W, float clamp) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  const long plane = (long)b * H * W;
  float term = 0.f;
  if (w < W && h < H) {
    const Pose P = load_pose(K, R0, t0, R1, t1, b);
    const long q = (long)h * W + w;
    const GeoPoint g = geo_forward(P, ray + q * 3, depth0[plane + q], depth1 + plane, H, W);
    term = clamp > 0.f ? fminf(g.diff, clamp) : g.diff;
  }
  block_partial(term, partials);
}

// Both directions of the symmetric loss in ONE launch and the final means by the LAST workgroup to finish.
// Tiles of 64 x 4 pixels as in the one-direction kernel (tile z < B: depth0 -> view 1, else depth1 -> view 0 with the
// poses swapped; x fastest, then y, then z, as one call numbers them), a wavefront per tile.  A work unit = kGeoGroup
// consecutive tiles of ONE direction = the four wavefronts of a workgroup; it publishes the f64 sum of its tiles' f32
// partials (tile order), then the workgroup draws a ticket; the holder of the last ticket adds the units of each
// direction -- thread t units t, t + 256, ..., then the tree of finish_mean_kernel, which sums the one-direction
// kernel's partials in the same groups (its `group` argument): the result does not depend on which workgroup finishes
// last and equals the two-call path bit for bit.  The tickets are left at zero.
// Visibility without fences (a device-scope release fence writes back the whole L2 of the XCD: with one per workgroup
// the launch took 260 us): the unit sums go out as device-scope atomic stores (written through to where every XCD sees
// them), awaited (vmcnt) before the ticket is drawn; the last workgroup reads them behind an acquire-only fence (as
// device-scope atomic loads they went out one round trip at a time).  Tickets in two levels -- kGeoSymGroups group
// words, then one global word -- because 13 824 device-scope increments of ONE word took 176 us (~12 ns apiece,
// serialised).  Units of four tiles instead of single tiles: a quarter of the values for the last workgroup to read.
// As two launches + two one-workgroup reductions the pair cost 2 x 12.3 + 2 x 8.6 us of the config-3 step.
//
// Memory-model note.  This is NOT a C++ release / acquire pair: the ticket increments are relaxed.  It is the hand-off
// form MI355X_MICROARCH.md lists as valid on gfx950 ("Valid forms": write-through `sc1` payload stores by ONE lane of
// the storing workgroup -> that lane's `s_waitcnt vmcnt(0)` -> its agent-scope atomic add; the workgroup whose add came
// last learns it from the returned value, passes it through a workgroup barrier, and reads behind an agent-scope
// acquire): the payload has reached the memory side before the add is issued because the same lane waited for it, and
// the inline asm wait is invisible to the compiler pass that drops redundant waits.  That is a property of gfx950's
// sc1 stores, not of the language, hence the guard below: any other target must take the two-call path
// (ctd_geometric_fwd_f32 twice), which needs no cross-workgroup visibility inside a launch.  A release on the
// increments instead (buffer_wbl2 per workgroup) was measured at 260 us for this launch against 34.
// The host side clears the ticket words whenever this call returns an error (geometric_sym_fwd_f32 below and the
// Python wrapper): a dirty word would leave every later launch on them without a last workgroup.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "geometric_sym_fwd_kernel's hand-off relies on gfx950 sc1 write-through stores: build for gfx950 only"
#endif
constexpr int kGeoSymBlocks = 4096, kGeoSymGroups = 64;      // ticket words: [0] global, [1 .. kGeoSymGroups] groups

__global__ __launch_bounds__(256) void geometric_sym_fwd_kernel(const float* __restrict__ depth0,
                                                                const float* __restrict__ depth1,
                                                                const float* __restrict__ ray, const float* __restrict__ K,
                                                                const float* __restrict__ R0, const float* __restrict__ t0,
                                                                const float* __restrict__ R1, const float* __restrict__ t1,
                                                                double* __restrict__ unit_sums, unsigned* __restrict__ ticket,
                                                                float* __restrict__ loss, int B, int H, int W, float clamp,
                                                                double count) {
  static_assert(kGeoGroup == 4, "a unit is the four wavefronts of a workgroup");
  __shared__ bool s_last;
  __shared__ float s_tile[4];
  __shared__ double s_sum[256];
  const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4;
  const long n_dir = (long)tiles_x * tiles_y * B;                 // tiles per direction
  const long units_dir = (n_dir + kGeoGroup - 1) / kGeoGroup, n_units = 2 * units_dir;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
    const bool rev = unit >= units_dir;
    const long t_dir = (unit - (rev ? units_dir : 0)) * kGeoGroup + wave;     // this wavefront's tile within the direction
    float t = 0.f;
    if (t_dir < n_dir) {                                          // (wave-uniform)
      // lane <-> column, the tile's four rows in the lane's registers (four independent chains of dependent loads in
      // flight): row sums by shuffle, then row 0 + 1 + 2 + 3 -- the order block_partial adds its four wavefronts in
      const int tx = (int)(t_dir % tiles_x), ty = (int)((t_dir / tiles_x) % tiles_y), b = (int)(t_dir / ((long)tiles_x * tiles_y));
      const int w = tx * 64 + lane;
      const long plane = (long)b * H * W;
      const Pose P = rev ? load_pose(K, R1, t1, R0, t0, b) : load_pose(K, R0, t0, R1, t1, b);
      const float* da = rev ? depth1 : depth0;
      const float* db = rev ? depth0 : depth1;
      float term[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int h = ty * 4 + k;
        term[k] = 0.f;
        if (w < W && h < H) {
          const long q = (long)h * W + w;
          const GeoPoint g = geo_forward(P, ray + q * 3, da[plane + q], db + plane, H, W);
          term[k] = clamp > 0.f ? fminf(g.diff, clamp) : g.diff;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) t += wave_sum(term[k]);
    }
    if (lane == 0) s_tile[wave] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
      double q = 0;
      for (int k = 0; k < kGeoGroup; ++k)
        if ((unit - (rev ? units_dir : 0)) * kGeoGroup + k < n_dir) q += (double)s_tile[k];
      __hip_atomic_store(unit_sums + unit, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the unit sums have been written where all XCDs see them
    const unsigned group = blockIdx.x % kGeoSymGroups;
    const unsigned in_group = (gridDim.x - group + kGeoSymGroups - 1) / kGeoSymGroups;
    bool last = __hip_atomic_fetch_add(ticket + 1 + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1u;
    if (last) {
      const unsigned n_groups = gridDim.x < (unsigned)kGeoSymGroups ? gridDim.x : (unsigned)kGeoSymGroups;
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_groups - 1u;
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;                                            // (workgroup-uniform)
  // acquire side: invalidate what this CU / XCD may hold of the sums' lines (no write-back involved), then plain loads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  float v[2];
  for (int dir = 0; dir < 2; ++dir) {
    double t = 0;
    const double* pd = unit_sums + dir * units_dir;
    long i = threadIdx.x;
    for (; i + 7 * 256 < units_dir; i += 8 * 256) {                // eight independent loads in flight, added in index order
      double x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = pd[i + k * 256];
#pragma unroll
      for (int k = 0; k < 8; ++k) t += x[k];
    }
    for (; i < units_dir; i += 256) t += pd[i];
    s_sum[threadIdx.x] = t;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) s_sum[threadIdx.x] += s_sum[threadIdx.x + st];
      __syncthreads();
    }
    v[dir] = (float)(s_sum[0] / count);
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = v[0] + v[1];
  if (threadIdx.x <= kGeoSymGroups) __hip_atomic_store(ticket + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// backward: grad_depth0 written (every element), grad_depth1 ACCUMULATED with float atomics into a buffer the
// caller zeroed (the bilinear scatter of ATen's grid_sample backward does the same)
__global__ __launch_bounds__(256) void geometric_bwd_kernel(const float* __restrict__ depth0,
                                                            const float* __restrict__ depth1,
                                                            const float* __restrict__ ray, const float* __restrict__ K,
                                                            const float* __restrict__ R0, const float* __restrict__ t0,
                                                            const float* __restrict__ R1, const float* __restrict__ t1,
                                                            const float* __restrict__ grad_loss, float inv_n,
                                                            float* __restrict__ grad_depth0,
                                                            float* __restrict__ grad_depth1, int H, int W, float clamp,
                                                            int accumulate0) {
  const int w = blockIdx.x * 64 + (threadIdx.x & 63), h = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (w >= W || h >= H) return;
  const int b = blockIdx.z;
  const long plane = (long)b * H * W, q = (long)h * W + w;
  const Pose P = load_pose(K, R0, t0, R1, t1, b);
  const GeoPoint g = geo_forward(P, ray + q * 3, depth0[plane + q], depth1 + plane, H, W);
  float gd = grad_loss[0] * inv_n;
  if (clamp > 0.f && !(g.diff >= 0.f && g.diff <= clamp)) gd = 0.f;     // clamp(diff, 0, c) passes inside [0, c]
  const float e = g.uvd[2] - g.sample;
  const float sgn = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
  const float g_d_direct = gd * sgn, g_sample = -gd * sgn;
  // scatter to depth1 and gradient w.r.t. the sampling position
  const int x1 = g.x0 + 1, y1 = g.y0 + 1;
  const float wx0 = 1.f - g.wx1, wy0 = 1.f - g.wy1;
  float* g1 = grad_depth1 + plane;
  const float* d1 = depth1 + plane;
  const float nw = d1[(long)g.y0 * W + g.x0];
  const float ne = x1 < W ? d1[(long)g.y0 * W + x1] : 0.f;
  const float sw = y1 < H ? d1[(long)y1 * W + g.x0] : 0.f;
  const float se = (x1 < W && y1 < H) ? d1[(long)y1 * W + x1] : 0.f;
  if (g_sample != 0.f) {
    atomicAdd(g1 + (long)g.y0 * W + g.x0, g_sample * wx0 * wy0);
    if (x1 < W) atomicAdd(g1 + (long)g.y0 * W + x1, g_sample * g.wx1 * wy0);
    if (y1 < H) atomicAdd(g1 + (long)y1 * W + g.x0, g_sample * wx0 * g.wy1);
    if (x1 < W && y1 < H) atomicAdd(g1 + (long)y1 * W + x1, g_sample * g.wx1 * g.wy1);
  }
  float g_ix = g_sample * (-nw * wy0 + ne * wy0 - sw * g.wy1 + se * g.wy1);
  float g_iy = g_sample * (-nw * wx0 - ne * g.wx1 + sw * wx0 + se * g.wx1);
  if (g.clip_x) g_ix = 0.f;
  if (g.clip_y) g_iy = 0.f;
  const float g_u = g_ix * ((float)W * 0.5f) * (2.f / (float)(W - 1));
  const float g_v = g_iy * ((float)H * 0.5f) * (2.f / (float)(H - 1));
  float g_uvd[3];
  g_uvd[0] = g_u / g.den;
  g_uvd[1] = g_v / g.den;
  const float g_den = -(g_u * g.uvd[0] + g_v * g.uvd[1]) / (g.den * g.den);
  g_uvd[2] = g_d_direct + (g.uvd[2] > 0.f ? g_den : 0.f);
  float g_s[3], g_q[3], g_p[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) g_s[i] = g_uvd[0] * P.K[0 * 3 + i] + g_uvd[1] * P.K[1 * 3 + i] + g_uvd[2] * P.K[2 * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) g_q[i] = g_s[0] * P.R1[0 * 3 + i] + g_s[1] * P.R1[1 * 3 + i] + g_s[2] * P.R1[2 * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i) g_p[i] = g_q[0] * P.R0[i * 3 + 0] + g_q[1] * P.R0[i * 3 + 1] + g_q[2] * P.R0[i * 3 + 2];
  const float gdep = g_p[0] * g.ray[0] + g_p[1] * g.ray[1] + g_p[2] * g.ray[2];
  grad_depth0[plane + q] = accumulate0 ? grad_depth0[plane + q] + gdep : gdep;
}

size_t geometric_workspace_bytes(int B, int H, int W) {
  return align_up(sizeof(float) * (size_t)loss_grid_blocks(B, H, W), 256);
}

int geometric_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                      const float* t0, const float* R1, const float* t1, float* loss, int accumulate, int B, int H,
                      int W, float clamp, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!ws || ws_bytes < geometric_workspace_bytes(B, H, W)) return CTD_ERR_WORKSPACE;
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B), block(256);
  hipLaunchKernelGGL(geometric_fwd_kernel, grid, block, 0, s, depth0, depth1, ray, K, R0, t0, R1, t1, (float*)ws, H, W,
                     clamp);
  CTD_LAUNCH_CHECK();
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(256), 0, s, (const float*)ws, loss_grid_blocks(B, H, W),
                     (double)B * H * W, loss, accumulate, kGeoGroup);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

int geometric_sym_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                          const float* t0, const float* R1, const float* t1, float* loss, int B, int H, int W, float clamp,
                          void* ws, size_t ws_bytes, unsigned* ticket, hipStream_t s) {
  // (one f64 per unit of four tiles: 2 x ceil(n_dir / 4) x 8 bytes <= the 2 x n_dir x 4 bytes of the tiles' f32 partials + 64)
  if (!ws || ws_bytes < geometric_workspace_bytes(2 * B, H, W) || ((uintptr_t)ws & 7) || !ticket) return CTD_ERR_WORKSPACE;
  const long n_dir = (long)ceil_div(W, 64) * ceil_div(H, 4) * B;
  const long n_units = 2 * ((n_dir + kGeoGroup - 1) / kGeoGroup);
  dim3 grid((unsigned)(n_units < kGeoSymBlocks ? n_units : kGeoSymBlocks)), block(256);
  hipLaunchKernelGGL(geometric_sym_fwd_kernel, grid, block, 0, s, depth0, depth1, ray, K, R0, t0, R1, t1, (double*)ws, ticket,
                     loss, B, H, W, clamp, (double)B * H * W);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    // whatever state the words are in, the next call must find them at zero (stream-ordered behind anything queued)
    (void)hipMemsetAsync(ticket, 0, sizeof(unsigned) * (kGeoSymGroups + 1), s);
    return CTD_ERR_HIP + (int)e;
  }
  return CTD_OK;
}

int geometric_bwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                      const float* t0, const float* R1, const float* t1, const float* grad_loss, float* grad_depth0,
                      int accumulate0, float* grad_depth1, int B, int H, int W, float clamp, hipStream_t s) {
  dim3 grid(ceil_div(W, 64), ceil_div(H, 4), B), block(256);
  hipLaunchKernelGGL(geometric_bwd_kernel, grid, block, 0, s, depth0, depth1, ray, K, R0, t0, R1, t1, grad_loss,
                     (float)(1.0 / ((double)B * H * W)), grad_depth0, grad_depth1, H, W, clamp, accumulate0);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
