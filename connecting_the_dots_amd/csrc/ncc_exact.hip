// ncc_exact.hip -- zero-mean NCC block-matching volume in REFERENCE OPERATION ORDER.
//
// Replaces xcorrvol_kernel / XCorrVolFunctor (torchext/ext/ext_kernel.cu:40-50,
// torchext/ext/ext.h:120-191).  The reference runs one thread per (d,h,w) output and
// re-reads 2*C*bs^2 taps from global memory twice per output.  Here:
//
//   * the per-window means / variances are hoisted: (mu0, sigma0) depend only on the
//     pixel, (mu1, sigma1) only on (h, x = w - d) because the reference shifts the
//     column BEFORE clamping it (ext.h:152-154).  A small pre-pass computes them with
//     the reference's own two-pass tap order, so the bits are identical;
//   * one thread owns one pixel, keeps its bs^2 centred taps v0 = in0 - mu0 in VGPRs
//     and walks the disparities ND at a time; pattern rows are staged in LDS with the
//     replicate border baked in, each LDS word feeds ND*bs multiply-adds;
//   * dot += v0*v1 stays a separate multiply and add (built with -ffp-contract=off):
//     the parity anchor is the reference CPU build, which has no FMA.
//
// This kernel is VALU-bound by construction (3 non-fusable f32 ops per tap per output,
// >= 243 ops per output for bs = 9); the HBM-roofline kernel is ncc_fast.hip.
#include "ctd_internal.h"
#include "ctd_ncc_point.h"

namespace ctd {

constexpr int kTW = 64;   // pixels per workgroup along w (one wavefront wide)
constexpr int kTH = 4;    // pixel rows per workgroup
constexpr int kND = 8;    // disparities register-blocked per thread

// ---------------------------------------------------------------------------------
// window statistics, reference order (ext.h:145-183): pass 1 mean with "x / T(bs^2)"
// per tap, pass 2 sum of squared deviations.  One thread per (f, c, h, xi);
// x = xi + x_start is the UNCLAMPED window centre column.
// ---------------------------------------------------------------------------------
template <typename T>
__global__ void window_stats_kernel(const T* __restrict__ in, long frame_stride, T* __restrict__ stats,
                                    int C, int H, int W, int x_start, int W_out, int bs, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int xi = (int)(i % W_out);
  long r = i / W_out;
  int h = (int)(r % H);
  r /= H;
  int c = (int)(r % C);
  int f = (int)(r / C);
  const T* img = in + f * frame_stride + (long)c * H * W;
  const int half = bs / 2;
  const int x = xi + x_start;
  const T bs2 = (T)(bs * bs);
  T mu = 0;
  for (int bh = 0; bh < bs; ++bh) {
    int hh = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      int ww = clampi(x + bw - half, 0, W - 1);
      mu += img[(long)hh * W + ww] / bs2;
    }
  }
  T sigma = 0;
  for (int bh = 0; bh < bs; ++bh) {
    int hh = clampi(h + bh - half, 0, H - 1);
    for (int bw = 0; bw < bs; ++bw) {
      int ww = clampi(x + bw - half, 0, W - 1);
      T v = img[(long)hh * W + ww] - mu;
      sigma += v * v;
    }
  }
  stats[2 * i] = mu;
  stats[2 * i + 1] = sigma;
}

// ---------------------------------------------------------------------------------
// main kernel, f32, compile-time block size.
//   grid  (ceil(W/kTW), ceil(H/kTH), frames), block (kTW, kTH)
//   LDS   tile0 [(kTH+BS-1)][kTW+BS-1]           frame window, replicate border
//         tile1 [(kTH+BS-1)][kTW+BS-1+Dpad-1]    pattern window for all disparities
// ---------------------------------------------------------------------------------
template <int BS, bool WRITE_VOL, bool ARGMAX, bool MULTI_C>
__global__ __launch_bounds__(kTW* kTH, 3) void ncc_exact_kernel(
    const float* __restrict__ in0, const float* __restrict__ in1, long in1_frame_stride,
    const float2* __restrict__ stats0, const float2* __restrict__ stats1, float* __restrict__ out,
    int64_t* __restrict__ idx_out, float* __restrict__ best_out, int C, int H, int W, int D) {
  constexpr int HALF = BS / 2;
  constexpr int TR = kTH + BS - 1;             // tile rows
  constexpr int TW0 = kTW + BS - 1;
  const int Dpad = (D + kND - 1) / kND * kND;
  const int TW1 = kTW + BS - 1 + Dpad - 1;
  const int W1 = W + Dpad - 1;                 // stats1 row: x = w - d in [-(Dpad-1), W-1]

  extern __shared__ float smem[];
  float* tile0 = smem;
  float* tile1 = smem + TR * TW0;

  const int tx = threadIdx.x, ty = threadIdx.y;
  const int tid = ty * kTW + tx;
  const int w_lo = blockIdx.x * kTW, h_lo = blockIdx.y * kTH;
  const int f = blockIdx.z;
  const int w = w_lo + tx, h = h_lo + ty;
  const bool active = (w < W) && (h < H);
  const long HW = (long)H * W;
  const float* img0 = in0 + (long)f * C * HW;
  const float* img1 = in1 + (long)f * in1_frame_stride;
  const float2* st0 = stats0 + (long)f * C * HW;
  const float2* st1 = stats1 + (in1_frame_stride ? (long)f * C * H * W1 : 0);
  float* vol = out + (long)f * D * HW;
  const int x0_tile1 = w_lo - HALF - (Dpad - 1);

  float bestv = 0.f;
  int besti = 0;

  for (int c = 0; c < C; ++c) {
    __syncthreads();
    for (int i = tid; i < TR * TW0; i += kTW * kTH) {
      int r = i / TW0, col = i - r * TW0;
      int hh = clampi(h_lo + r - HALF, 0, H - 1);
      int ww = clampi(w_lo + col - HALF, 0, W - 1);
      tile0[i] = img0[(long)c * HW + (long)hh * W + ww];
    }
    for (int i = tid; i < TR * TW1; i += kTW * kTH) {
      int r = i / TW1, col = i - r * TW1;
      int hh = clampi(h_lo + r - HALF, 0, H - 1);
      int ww = clampi(x0_tile1 + col, 0, W - 1);
      tile1[i] = img1[(long)c * HW + (long)hh * W + ww];
    }
    __syncthreads();
    if (!active) continue;

    const float2 s0 = st0[(long)c * HW + (long)h * W + w];
    float v0[BS * BS];
#pragma unroll
    for (int bh = 0; bh < BS; ++bh)
#pragma unroll
      for (int bw = 0; bw < BS; ++bw) v0[bh * BS + bw] = tile0[(ty + bh) * TW0 + tx + bw] - s0.x;

    const float2* st1x = st1 + ((long)c * H + h) * W1 + (Dpad - 1) + w;   // st1x[-d] <-> x = w - d
#pragma unroll 1
    for (int d0 = 0; d0 < Dpad; d0 += kND) {
      float acc[kND], mu1[kND];
#pragma unroll
      for (int j = 0; j < kND; ++j) {
        mu1[j] = st1x[-d0 - j].x;
        acc[j] = 0.f;
      }
      // The window of group d0+kND overlaps this group's by kND words; hide that from GVN's
      // load-PRE, which would otherwise carry 9 x kND loaded words across the back edge in
      // VGPRs (and spill the v0 taps to scratch to make room).
      int col0 = Dpad - kND - d0;
      asm volatile("" : "+s"(col0));
      const float* rowp = tile1 + ty * TW1 + tx + col0;
#pragma unroll
      for (int bh = 0; bh < BS; ++bh) {
        float br[kND + BS - 1];
#pragma unroll
        for (int k = 0; k < kND + BS - 1; ++k) br[k] = rowp[bh * TW1 + k];
#pragma unroll
        for (int j = 0; j < kND; ++j)
#pragma unroll
          for (int bw = 0; bw < BS; ++bw) {
            float v1 = br[bw + kND - 1 - j] - mu1[j];
            float p = v0[bh * BS + bw] * v1;   // separate multiply ...
            acc[j] = acc[j] + p;               // ... and add: no FMA (ext.h:179 on the CPU build)
          }
        // keep the scheduler from hoisting the next rows' LDS reads over this row's
        // 216 VALU ops (it otherwise runs the kernel out of VGPRs and spills)
        __builtin_amdgcn_sched_barrier(0);
      }
      // pin the sums here: LLVM otherwise sinks each acc[j] chain into its "d < D" block
      // below, which keeps all 9 x (kND+BS-1) window words live and spills.
#pragma unroll
      for (int j = 0; j < kND; ++j) asm volatile("" : "+v"(acc[j]));
#pragma unroll
      for (int j = 0; j < kND; ++j) {
        const int d = d0 + j;
        float q = acc[j] / ncc_norm(s0.y, st1x[-d].y);
        long o = (long)(d < D ? d : D - 1) * HW + (long)h * W + w;
        float val = 0.f;
        if (MULTI_C) val = (c == 0) ? 0.f : vol[o];   // channels accumulate in order c = 0..C-1
        val += q;                                     // "T val = 0; val += dot / norm" (ext.h:142,186)
        if (d < D) {
          if (WRITE_VOL) vol[o] = val;
          if (ARGMAX) {                               // only instantiated for C == 1
            if (d == 0 || val > bestv) { bestv = val; besti = d; }
          }
        }
      }
    }
  }
  if (ARGMAX && active) {
    long p = (long)f * HW + (long)h * W + w;
    idx_out[p] = besti;
    if (best_out) best_out[p] = bestv;
  }
}

// ---------------------------------------------------------------------------------
// generic LDS-tiled kernel: any block size (even ones too: the window is rows h - bs/2 .. h - bs/2 + bs - 1,
// ext.h:148-150), f32 or f64 -- what the reference's AT_DISPATCH_FLOATING_TYPES covers (ext_kernel.cu:45) beyond the
// compile-time f32 kernel above.  Same decomposition: window means / sums of squared deviations hoisted into
// window_stats_kernel (reference tap order, so the bits are the reference's), a 64 x 4 pixel tile, the frame tile and
// the pattern rows for ALL disparities staged in LDS with the replicate border baked in; a thread then walks the
// disparities of its pixel with the taps coming from LDS (2 bs^2 LDS reads per output where the reference's kernel
// makes 4 bs^2 global loads).  dot accumulates in the reference's order, multiply and add unfused.
//   grid (ceil(W/64), ceil(H/4), frames x disparity chunks), block (64, 4); LDS (4 + bs - 1) x ((64 + bs - 1) +
//   (64 + bs - 1 + Dc - 1)) T for a chunk of Dc disparities (all of D when that fits 64 KB)
// ---------------------------------------------------------------------------------
template <typename T>
struct Stat2 { T mu, sigma; };

template <typename T>
__global__ __launch_bounds__(kTW* kTH) void ncc_tiled_generic_kernel(
    const T* __restrict__ in0, const T* __restrict__ in1, long in1_frame_stride, const Stat2<T>* __restrict__ stats0,
    const Stat2<T>* __restrict__ stats1, T* __restrict__ out, int C, int H, int W, int D, int bs, int d_chunk,
    int n_chunks) {
  extern __shared__ double smem_generic[];
  T* tile0 = (T*)smem_generic;
  const int half = bs / 2;
  // blockIdx.z = frame * n_chunks + chunk: this workgroup serves the disparities [d_lo, d_hi) (the pattern rows of a
  // chunk, not of all D, are what has to fit LDS)
  const int f = blockIdx.z / n_chunks, d_lo = (blockIdx.z - f * n_chunks) * d_chunk, d_hi = min(d_lo + d_chunk, D);
  const int TR = kTH + bs - 1, TW0 = kTW + bs - 1, TW1 = kTW + bs - 1 + (d_hi - d_lo) - 1;
  T* tile1 = tile0 + TR * TW0;
  const int W1 = W + D - 1;                    // stats1 row: x = w - d in [-(D-1), W-1]
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kTW + tx;
  const int w_lo = blockIdx.x * kTW, h_lo = blockIdx.y * kTH;
  const int w = w_lo + tx, h = h_lo + ty;
  const bool active = (w < W) && (h < H);
  const long HW = (long)H * W;
  const T* img0 = in0 + (long)f * C * HW;
  const T* img1 = in1 + (long)f * in1_frame_stride;
  const Stat2<T>* st0 = stats0 + (long)f * C * HW;
  const Stat2<T>* st1 = stats1 + (in1_frame_stride ? (long)f * C * H * W1 : 0);
  T* vol = out + (long)f * D * HW;
  const int x0_tile1 = w_lo - half - (d_hi - 1);
  for (int c = 0; c < C; ++c) {
    __syncthreads();
    for (int i = tid; i < TR * TW0; i += kTW * kTH) {
      const int r = i / TW0, col = i - r * TW0;
      tile0[i] = img0[(long)c * HW + (long)clampi(h_lo + r - half, 0, H - 1) * W + clampi(w_lo + col - half, 0, W - 1)];
    }
    for (int i = tid; i < TR * TW1; i += kTW * kTH) {
      const int r = i / TW1, col = i - r * TW1;
      tile1[i] = img1[(long)c * HW + (long)clampi(h_lo + r - half, 0, H - 1) * W + clampi(x0_tile1 + col, 0, W - 1)];
    }
    __syncthreads();
    if (!active) continue;
    const Stat2<T> s0 = st0[(long)c * HW + (long)h * W + w];
    const Stat2<T>* st1x = st1 + ((long)c * H + h) * W1 + (D - 1) + w;       // st1x[-d] <-> x = w - d
    for (int d = d_lo; d < d_hi; ++d) {
      const Stat2<T> s1 = st1x[-d];
      const T* r0 = tile0 + ty * TW0 + tx;
      const T* r1 = tile1 + ty * TW1 + tx + (d_hi - 1) - d;
      T dot = 0;
      for (int bh = 0; bh < bs; ++bh)
        for (int bw = 0; bw < bs; ++bw) {
          const T v0 = r0[bh * TW0 + bw] - s0.mu;
          const T v1 = r1[bh * TW1 + bw] - s1.mu;
          const T p = v0 * v1;                    // separate multiply and add (-ffp-contract=off): ext.h:179, no FMA
          dot = dot + p;
        }
      const long o = (long)d * HW + (long)h * W + w;
      T val = c == 0 ? (T)0 : vol[o];             // channels accumulate in order (ext.h:142,186)
      val += dot / ncc_norm(s0.sigma, s1.sigma);
      vol[o] = val;
    }
  }
}

template <typename T>
size_t generic_workspace_bytes(int frames, int C, int H, int W, int D, bool per_frame_pattern) {
  const size_t n0 = align_up((size_t)frames * C * H * W * sizeof(Stat2<T>), 256);
  const size_t n1 = align_up((size_t)(per_frame_pattern ? frames : 1) * C * H * (W + D - 1) * sizeof(Stat2<T>), 256);
  return n0 + n1;
}

template <typename T>
static int launch_generic(const T* in0, const T* in1, long in1_frame_stride, T* out, int frames, int C, int H, int W, int D,
                          int bs, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const bool per_frame = in1_frame_stride != 0;
  if (workspace == nullptr || workspace_bytes < generic_workspace_bytes<T>(frames, C, H, W, D, per_frame))
    return CTD_ERR_WORKSPACE;
  // disparities per workgroup: as many as keep the two tiles within 64 KB of LDS (all of them for the usual shapes)
  const size_t row = (size_t)(kTH + bs - 1) * sizeof(T);
  const long fit = (long)(64 * 1024 / row) - 2 * (kTW + bs - 1) + 1;
  if (fit < 1) return CTD_ERR_UNSUPPORTED;                       // (block sizes in the hundreds)
  const int d_chunk = fit < D ? (int)fit : D;
  const int n_chunks = ceil_div(D, d_chunk);
  if ((long)frames * n_chunks > 65535) return CTD_ERR_UNSUPPORTED;
  const size_t lds = row * ((kTW + bs - 1) + (kTW + bs - 1 + d_chunk - 1));
  Stat2<T>* stats0 = (Stat2<T>*)workspace;
  Stat2<T>* stats1 = (Stat2<T>*)((char*)workspace + align_up((size_t)frames * C * H * W * sizeof(Stat2<T>), 256));
  const int W1 = W + D - 1;
  long total = (long)frames * C * H * W;
  hipLaunchKernelGGL(window_stats_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in0,
                     (long)C * H * W, (T*)stats0, C, H, W, 0, W, bs, total);
  CTD_LAUNCH_CHECK();
  total = (long)(per_frame ? frames : 1) * C * H * W1;
  hipLaunchKernelGGL(window_stats_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in1,
                     in1_frame_stride, (T*)stats1, C, H, W, -(D - 1), W1, bs, total);
  CTD_LAUNCH_CHECK();
  auto kern = ncc_tiled_generic_kernel<T>;
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid(ceil_div(W, kTW), ceil_div(H, kTH), frames * n_chunks), block(kTW, kTH);
  timing_begin(stream);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, in0, in1, in1_frame_stride, stats0, stats1, out, C, H, W, D, bs, d_chunk,
                     n_chunks);
  timing_end(stream, W);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// argmax over d of a materialised volume; first index wins (strict >), one thread per pixel.
__global__ void argmax_disp_kernel(const float* __restrict__ vol, int64_t* __restrict__ idx,
                                   float* __restrict__ best, int D, long HW, long total) {
  long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= total) return;
  long f = p / HW, q = p - f * HW;
  const float* v = vol + f * D * HW + q;
  float m = v[0];
  int mi = 0;
  int d = 1;
  for (; d + 8 <= D; d += 8) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = v[(long)(d + k) * HW];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (t[k] > m) { m = t[k]; mi = d + k; }
  }
  for (; d < D; ++d) {
    float t = v[(long)d * HW];
    if (t > m) { m = t; mi = d; }
  }
  idx[p] = mi;
  if (best) best[p] = m;
}

struct ExactWorkspace {
  float2* stats0;
  float2* stats1;
  size_t bytes;
};

static ExactWorkspace exact_workspace(void* base, int frames, int C, int H, int W, int D, bool per_frame_pattern) {
  const int W1 = W + (D + kND - 1) / kND * kND - 1;
  size_t n0 = (size_t)frames * C * H * W * sizeof(float2);
  size_t n1 = (size_t)(per_frame_pattern ? frames : 1) * C * H * W1 * sizeof(float2);
  ExactWorkspace ws;
  ws.stats0 = (float2*)base;
  ws.stats1 = (float2*)((char*)base + align_up(n0, 256));
  ws.bytes = align_up(n0, 256) + align_up(n1, 256);
  return ws;
}

template <int BS, bool WRITE_VOL, bool ARGMAX, bool MULTI_C>
static int launch_exact_c(const float* in0, const float* in1, long in1_frame_stride, float* out, int64_t* idx,
                        float* best, int frames, int C, int H, int W, int D, void* workspace, hipStream_t stream) {
  const bool per_frame = in1_frame_stride != 0;
  ExactWorkspace ws = exact_workspace(workspace, frames, C, H, W, D, per_frame);
  const int Dpad = (D + kND - 1) / kND * kND;
  const int W1 = W + Dpad - 1;
  {
    long total = (long)frames * C * H * W;
    hipLaunchKernelGGL(window_stats_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in0,
                       (long)C * H * W, (float*)ws.stats0, C, H, W, 0, W, BS, total);
    CTD_LAUNCH_CHECK();
    int nf1 = per_frame ? frames : 1;
    total = (long)nf1 * C * H * W1;
    hipLaunchKernelGGL(window_stats_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in1,
                       in1_frame_stride, (float*)ws.stats1, C, H, W, -(Dpad - 1), W1, BS, total);
    CTD_LAUNCH_CHECK();
  }
  const int TR = kTH + BS - 1;
  size_t lds = sizeof(float) * (size_t)TR * ((kTW + BS - 1) + (kTW + BS - 1 + Dpad - 1));
  if (lds > 160 * 1024) return CTD_ERR_UNSUPPORTED;
  dim3 grid(ceil_div(W, kTW), ceil_div(H, kTH), frames), block(kTW, kTH);
  auto kern = ncc_exact_kernel<BS, WRITE_VOL, ARGMAX, MULTI_C>;
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  timing_begin(stream);
  hipLaunchKernelGGL(kern, grid, block, lds, stream, in0, in1, in1_frame_stride, ws.stats0, ws.stats1, out, idx, best,
                     C, H, W, D);
  timing_end(stream, W);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <int BS, bool WRITE_VOL, bool ARGMAX>
static int launch_exact(const float* in0, const float* in1, long s1, float* out, int64_t* idx, float* best, int frames,
                        int C, int H, int W, int D, void* ws, hipStream_t st) {
  if (C == 1) return launch_exact_c<BS, WRITE_VOL, ARGMAX, false>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
  if (ARGMAX) return CTD_ERR_UNSUPPORTED;
  return launch_exact_c<BS, WRITE_VOL, false, true>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
}

template <bool WRITE_VOL, bool ARGMAX>
static int dispatch_exact(int bs, const float* in0, const float* in1, long s1, float* out, int64_t* idx, float* best,
                          int frames, int C, int H, int W, int D, void* ws, hipStream_t st) {
  switch (bs) {
    case 3: return launch_exact<3, WRITE_VOL, ARGMAX>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
    case 5: return launch_exact<5, WRITE_VOL, ARGMAX>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
    case 7: return launch_exact<7, WRITE_VOL, ARGMAX>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
    case 9: return launch_exact<9, WRITE_VOL, ARGMAX>(in0, in1, s1, out, idx, best, frames, C, H, W, D, ws, st);
    default: return CTD_ERR_UNSUPPORTED;
  }
}

static bool exact_has_tiled(int bs) { return bs == 3 || bs == 5 || bs == 7 || bs == 9; }

// entry points used by ctd_api.hip -------------------------------------------------
size_t ncc_exact_workspace_bytes(int frames, int C, int H, int W, int D, int bs, bool per_frame_pattern) {
  // worst case over the f32 kernels and the generic f64 one (the caller sizes one workspace per call)
  const size_t g = generic_workspace_bytes<double>(frames, C, H, W, D, per_frame_pattern);
  if (!exact_has_tiled(bs)) return g;
  const size_t t = exact_workspace(nullptr, frames, C, H, W, D, per_frame_pattern).bytes;
  return t > g ? t : g;
}

int ncc_exact_f32(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H,
                  int W, int D, int bs, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (exact_has_tiled(bs)) {
    if (workspace == nullptr || workspace_bytes < ncc_exact_workspace_bytes(frames, C, H, W, D, bs, in1_frame_stride != 0))
      return CTD_ERR_WORKSPACE;
    return dispatch_exact<true, false>(bs, in0, in1, in1_frame_stride, out, nullptr, nullptr, frames, C, H, W, D,
                                       workspace, stream);
  }
  return launch_generic<float>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, bs, workspace, workspace_bytes, stream);
}

int ncc_exact_f64(const double* in0, const double* in1, long in1_frame_stride, double* out, int frames, int C, int H,
                  int W, int D, int bs, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  return launch_generic<double>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, bs, workspace, workspace_bytes, stream);
}

int ncc_exact_argmax_f32(const float* in0, const float* in1, long in1_frame_stride, float* vol_out, int64_t* idx,
                         float* best, int frames, int H, int W, int D, int bs, void* workspace,
                         size_t workspace_bytes, hipStream_t stream) {
  if (!exact_has_tiled(bs)) return CTD_ERR_UNSUPPORTED;
  if (workspace == nullptr || workspace_bytes < ncc_exact_workspace_bytes(frames, 1, H, W, D, bs, in1_frame_stride != 0))
    return CTD_ERR_WORKSPACE;
  if (vol_out)
    return dispatch_exact<true, true>(bs, in0, in1, in1_frame_stride, vol_out, idx, best, frames, 1, H, W, D,
                                      workspace, stream);
  return dispatch_exact<false, true>(bs, in0, in1, in1_frame_stride, nullptr, idx, best, frames, 1, H, W, D, workspace,
                                     stream);
}

int argmax_disp_f32(const float* vol, int64_t* idx, float* best, int frames, int D, int H, int W, hipStream_t stream) {
  long HW = (long)H * W, total = (long)frames * HW;
  hipLaunchKernelGGL(argmax_disp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, vol, idx, best, D,
                     HW, total);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
