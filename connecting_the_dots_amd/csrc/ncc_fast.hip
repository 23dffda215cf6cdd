// ncc_fast.hip -- zero-mean NCC block-matching volume, separable window sums (CTD_NCC_FAST).
//
// Same op as ncc_exact.hip (XCorrVolFunctor, torchext/ext/ext.h:120-191) but evaluated as
//     NCC = (S_ab - n*ma*mb) / (sa*sb + 1e-8),   S_ab = sum over the bs x bs window of a*b
// so that each output costs ~20 VALU slots instead of >= 243 and the kernel is bound by
// the 4 B/output volume store (HBM roofline).  Results agree with the reference order
// to |a-b| <= 1e-5*|b| + 1e-6 (tests), not bit for bit; bit-exact indices come from the
// re-rank in ctd_xcorrvol_argmax_f32.
//
// Work decomposition (one wavefront = 64 product columns, 4 disparities per lane):
//   * a workgroup is 4 consumer wavefronts (16 adjacent disparities of one column tile)
//     plus 1 LOADER wavefront.  The loader streams each row's operands (frame sample,
//     pattern span, window statistics) global -> LDS with LDS-DMA, a few rows ahead, and
//     is the only wave that ever waits on a load.  Consumers touch global memory only to
//     store: on gfx950 loads and stores retire in order on one counter (vmcnt), so a wave
//     that both loads and stores stalls on its own stores' HBM latency every row;
//   * consumer lane l owns the UNCLAMPED product column w0 = w_lo - HALF + l and marches
//     down the rows of a band.  Per row it forms p = a'(r,w0) * b'(r,w0-d) for its ND
//     disparities;
//   * vertical bs-sum of p: registers only, as a 3+3+3 tree over a ring of past rows
//     (no running sums, so no drift: every output is a fresh <= 4-level sum);
//   * horizontal bs-sum across lanes: +-1 with DPP wave shifts, +-3 with ds_bpermute;
//     64-(bs-1) of the 64 lanes produce outputs, stored as one contiguous row segment;
//   * a', b' are the inputs minus one constant per image (the window mean at the image
//     centre): an exact-arithmetic no-op for NCC that removes the cancellation in
//     S_ab - n*ma*mb for inputs with a DC offset.  The pre-pass writes the centred copies,
//     so the main kernels never subtract; one constant per image also keeps the
//     reference's exact ties along d at the left border exact (same operands, same order).
// Window means / deviations (ma, sa, mb, sb) come from the same separable f64 pre-pass.
#include <type_traits>

#include "ctd_internal.h"
#include "ctd_ncc_point.h"
#include "ctd_prepass.h"
#include "ctd_rank.h"
#include "ctd_tail.h"
#include "ctd_wave.h"


namespace ctd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFND = 4;        // disparities per lane
constexpr int kFWaves = 4;     // consumer wavefronts per workgroup (adjacent disparity groups)
constexpr int kFDG = kFND * kFWaves;   // disparities per workgroup
constexpr int kFSpan = 64 + kFDG - 1;  // pattern columns one row of a workgroup touches (79)
constexpr int kFSpanPad = 80;
constexpr int kFPack = 3 * 64 + 3 * kFSpanPad;   // floats per staged row: A MA SA | B MB SB
constexpr int kFRows = 3;      // rows per LDS chunk (one barrier per chunk)
constexpr int kFBufs = 4;      // LDS chunks in the ring (loader runs kFBufs-1 chunks ahead)
constexpr int kFDmaPerRow = 9; // LDS-DMA instructions the loader issues per row

// ------------------------------------------------------------------------------------
// pre-pass: (mean, sqrt(sum of squared deviations)) of the clamped bs x bs window centred
// at the unclamped column x = xi + x_start, separable f64 sums through LDS.
// ------------------------------------------------------------------------------------
#ifndef CTD_PRE_TW
// (A/B of the f32 kernel, tools/ab_tail.sh, frames only: 64 x 16 24.6 us, 64 x 12 25.3, 64 x 24 26.1, 32 x 24 26.7, 32 x 32 27.6,
// 32 x 16 28.2, 128 x 16 28.3, 64 x 32 28.8, 128 x 8 30.0; the f64 kernel of rounds 1-3 preferred 32 x 24)
#define CTD_PRE_TW 64
#define CTD_PRE_TH 16
#endif
constexpr int kSTW = CTD_PRE_TW, kSTH = CTD_PRE_TH, kSRows = 256 / CTD_PRE_TW;   // (A/B, rocprofv3, with the pattern job: 64 x 16: 36.9 us, 32 x 32: 35.1, 32 x 24: 34.0, 32 x 16: 37.1, 32 x 48: 39.4)
// kDevFloor, kFlagRatio (the listing rule's two constants): ctd_prepass.h
#ifndef CTD_PREPASS_F32
#define CTD_PREPASS_F32 1
#endif
constexpr bool kPrepassF32 = CTD_PREPASS_F32 != 0;   // block 9: f32 window sums on centred samples (see ncc_prepass_kernel)

// out_mean = mean_scale * (window mean - cval), out_dev = 1 / sqrt(sum of squared deviations) (0: listed window), out_img = img - cval
// (replicate border baked in), all laid out [image][H][W_out] with column x = xi + x_start;
// cval = f64 window mean at the image centre, recomputed identically by every workgroup.
// One launch serves the frames (job a) and the pattern (job b): blockIdx.z < a.nimg -> image blockIdx.z of job a,
// else image blockIdx.z - a.nimg of job b; workgroups past a job's plane width exit at once.
struct PrepassJob {
  const float* in;
  long frame_stride;
  float *out_img, *out_mean, *out_dev;
  int x_start, W_out, nimg;
  unsigned* n_flag;
  unsigned long long* flag_list;
  int col_lo, col_hi;
  unsigned* n_runs;
  unsigned long long* run_rows;
  double mean_scale;          // out_mean = mean_scale * (window mean - cval): -bs^2 for the frames (the kernels' n*ma*mb
                              // term then needs no multiply of its own), 1 for the pattern
  double flag_ratio;          // list a window when F - 1 > flag_ratio: kFlagRatio / C (the channels' errors add up in the sum)
  int pitch, o_off, halo;     // output row pitch and column offset of xi = 0; halo > 0: the planes carry `halo` replicate
                              // columns either side of the W_out computed ones, and out_img's are filled here (copies of
                              // columns 0 and W_out - 1; the statistics planes' halo columns are never read).  The frames'
                              // planes: computing the halo columns as windows of their own made a ninth column of
                              // workgroups for 8 of 520 columns.
};

// BSC > 0: compile-time block size (tap loops unrolled); BSC == 0: run-time `bs_rt`
// F32 (round 4, block 9): the window sums in f32 on samples CENTRED by the image's constant (the planes hold centred
// values anyway): sum of squared deviations = s2' - s1' * mean', whose relative error is ~F * 2^-24 * (number of
// roundings) with F = s2' / var = 1 + n (mean - centring)^2 / var -- the very factor the listing rule bounds by sqrt(8)
// (windows above it are recomputed by the fix-up pass), so unlisted windows keep their reciprocal deviation to ~1e-6
// relative, well inside the fast path's error budget; the raw mean for the flat-window test is mean' + centring.  Half the
// LDS, no f32 -> f64 conversions, full-rate additions.
template <int BSC, bool F32 = false>
__global__ __launch_bounds__(kSTW* kSRows) void ncc_prepass_kernel(PrepassJob ja, PrepassJob jb, int H, int W, int bs_rt,
                                                                   unsigned* __restrict__ clear_counters, int n_clear) {
  // the work-list counters of a ranked call (first used two kernels later): cleared here instead of by a memset launch
  if (clear_counters && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (int)(threadIdx.y * kSTW + threadIdx.x) < n_clear)
    clear_counters[(threadIdx.y * kSTW + threadIdx.x) * kWorkListStride] = 0u;
  const bool is_a = (int)blockIdx.z < ja.nimg;
  const PrepassJob& jp = is_a ? ja : jb;
  const int img_idx = is_a ? (int)blockIdx.z : (int)blockIdx.z - ja.nimg;
  const float* __restrict__ in = jp.in;
  const long frame_stride = jp.frame_stride;
  float* __restrict__ out_img = jp.out_img;
  float* __restrict__ out_mean = jp.out_mean;
  float* __restrict__ out_dev = jp.out_dev;
  const int x_start = jp.x_start, W_out = jp.W_out, col_lo = jp.col_lo, col_hi = jp.col_hi;
  unsigned* __restrict__ n_flag = jp.n_flag;
  unsigned long long* __restrict__ flag_list = jp.flag_list;
  unsigned* __restrict__ n_runs = jp.n_runs;
  unsigned long long* __restrict__ run_rows = jp.run_rows;
  if ((int)blockIdx.x * kSTW >= W_out) return;
  extern __shared__ double lds_d[];
  __shared__ double cred[1];
  typedef typename std::conditional<F32, float, double>::type acc_t;
  const int bs = BSC > 0 ? BSC : bs_rt;
  const int half = bs / 2;
  const int TRr = kSTH + bs - 1, TCc = kSTW + bs - 1;
  acc_t* rs1 = (acc_t*)lds_d;
  acc_t* rs2 = rs1 + TRr * kSTW;
  float* tile = (float*)(rs2 + TRr * kSTW);
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * kSTW + tx;
  const int xi_lo = blockIdx.x * kSTW, h_lo = blockIdx.y * kSTH;
  const float* img = in + (long)img_idx * frame_stride;          // image = frame * C + channel
  // centring constant of the image = mean of its centre window: the first wavefront sums it with a fixed shuffle
  // butterfly (the same bits in every workgroup), everyone else goes straight to the staging loads
  if (tid < 64) {
    double t = 0;
    for (int k = tid; k < bs * bs; k += 64) {
      int hh = clampi(H / 2 + k / bs - half, 0, H - 1), ww = clampi(W / 2 + k % bs - half, 0, W - 1);
      t += (double)img[(long)hh * W + ww];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if (tid == 0) cred[0] = t;
  }
  // batches of independent loads: one memory round trip per 8 elements of a thread instead of one each
  for (int i0 = tid; i0 < TRr * TCc; i0 += kSTW * kSRows * 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + kSTW * kSRows * u, TRr * TCc - 1);
      const int r = i / TCc, c = i - r * TCc;
      const int hh = clampi(h_lo + r - half, 0, H - 1);
      const int ww = clampi(xi_lo + x_start + c - half, 0, W - 1);
      t[u] = img[(long)hh * W + ww];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + kSTW * kSRows * u < TRr * TCc) tile[i0 + kSTW * kSRows * u] = t[u];
  }
  __syncthreads();
  const double n = (double)(bs * bs), inv_n = 1.0 / n;
  const float cval = (float)(cred[0] / n);
  const acc_t shift = F32 ? (acc_t)cval : (acc_t)0;                // F32: sums of the centred samples
  for (int r = ty; r < TRr; r += kSRows) {
    const float* row = tile + r * TCc + tx;
    acc_t s1 = 0, s2 = 0;
#pragma unroll
    for (int k = 0; k < BSC; ++k) {
      acc_t v = (acc_t)row[k] - shift;
      s1 += v;
      s2 += v * v;
    }
    if (BSC == 0)
      for (int k = 0; k < bs; ++k) {
        acc_t v = (acc_t)row[k] - shift;
        s1 += v;
        s2 += v * v;
      }
    rs1[r * kSTW + tx] = s1;
    rs2[r * kSTW + tx] = s2;
  }
  __syncthreads();
  const int xi = xi_lo + tx;
  for (int r = ty; r < kSTH; r += kSRows) {
    const int h = h_lo + r;
    if (xi >= W_out || h >= H) continue;
    acc_t a1 = 0, a2 = 0;
#pragma unroll
    for (int k = 0; k < BSC; ++k) {
      a1 += rs1[(r + k) * kSTW + tx];
      a2 += rs2[(r + k) * kSTW + tx];
    }
    if (BSC == 0)
      for (int k = 0; k < bs; ++k) {
        a1 += rs1[(r + k) * kSTW + tx];
        a2 += rs2[(r + k) * kSTW + tx];
      }
    // F32: a1, a2 are sums of centred samples -- mean = centred mean + centring, var is shift-invariant
    const double s1 = (double)a1, s2 = (double)a2;
    const double mean_c = F32 ? (double)(a1 * (acc_t)inv_n) : s1 * inv_n;     // mean of the summed samples
    double var = F32 ? (double)(a2 - a1 * (acc_t)mean_c) : s2 - s1 * mean_c;  // sum of squared deviations (sigma of ext.h:180-181)
    const double mean = F32 ? mean_c + (double)cval : mean_c;
    // Windows whose outputs the fast kernel cannot deliver within tolerance are listed for ncc_fixup_kernel
    // (see there), which recomputes EVERY output they take part in:
    //  * deviation small against the offset from the centring constant: cov = S_ab - n*ma*mb cancels in f32;
    //  * (nearly) flat window, deviation below 2e-4 of its mean: the reference's own value is then decided by
    //    the rounding of its mean (ext.h:157-158) and only the same operation order reproduces it; or deviation
    //    below kDevFloor, where the 1e-8 of the reference's denominator stops being a small correction.
    // A listed window's reciprocal deviation is stored as 0: the fast kernels then produce the placeholder score 0 for
    // exactly the outputs the fix-up pass overwrites (finite, so the in-kernel ranking's integer keys stay ordered;
    // what the ranking does about placeholders: see the all-D kernel).
    const double mc = F32 ? mean_c : mean - (double)cval;
    const bool flat = 4e-8 * n * mean * mean > var || var < kDevFloor * kDevFloor;
    const bool listed = flat || n * mc * mc > jp.flag_ratio * var;
    // reciprocal deviation (see ncc_inv_norm): v_rsq_f32 and one Newton step in f32, 1e-7 relative -- the f64 square
    // root and the two f64 divisions this line and `mean` used to cost were 60 % of the kernel's instructions
    const float vf = (float)(var > 0 ? var : 1.0);
    float rdev = __builtin_amdgcn_rsqf(vf);
    rdev = fmaf(0.5f * rdev, fmaf(-vf * rdev, rdev, 1.f), rdev);
    const long o = ((long)img_idx * H + h) * jp.pitch + jp.o_off + xi;
    const int col = xi + x_start;
    out_mean[o] = (float)(jp.mean_scale * mc);
    out_dev[o] = listed ? 0.f : rdev;
    const float centred = tile[(r + half) * TCc + tx + half] - cval;
    out_img[o] = centred;
    if (jp.halo > 0) {
      if (xi == 0)
        for (int k = 1; k <= jp.halo; ++k) out_img[o - k] = centred;
      if (xi == W_out - 1)
        for (int k = 1; k <= jp.halo; ++k) out_img[o + k] = centred;
    }
    if (listed && col >= col_lo && col < col_hi) {
      flag_list[atomicAdd(n_flag, 1u)] = ((unsigned long long)img_idx << 40) | ((unsigned long long)h << 20) |
                                         (unsigned long long)(col + 0x80000);
      if (run_rows && col == col_lo) run_rows[atomicAdd(n_runs, 1u)] = ((unsigned long long)img_idx << 20) | (unsigned long long)h;
    }
  }
}

// Fix-up pass of the fast path.
//
// Error model of the fast kernel (tools/err_vs_factor.py): cov = S_ab - n*ma*mb is formed from values
// centred by one constant per image, so |fast - exact| <~ c * 2^-24 * sqrt(Fa * Fb) with
// F = 1 + n*(window mean - centring)^2 / (sum of squared deviations) per window and c <= ~6 (ten f32
// roundings along the longest summation path).  The contract |a-b| <= 1e-5|b| + 1e-6 therefore holds
// whenever Fa * Fb <= 8; LCN'd input has F ~ 1 except in flat regions and in the low-variance windows clamped
// to column 0.  The pre-pass lists every window with F > sqrt(8) (and marks it with a NaN deviation); this kernel
// visits the listed windows and recomputes every output they take part in in the reference's operation order
// (bit-identical to CTD_NCC_EXACT).
// One wavefront per listed window, lane <-> disparity; the window itself (FIX, bs x bs) and the rows of
// the other image it meets over all disparities (SPAN, bs x (bs + D - 1)) are staged in LDS per channel.
//   frame window  (f, h, w): outputs (f, d, h, w);        SPAN = pattern columns w-half-(D-1) .. w+half
//   pattern window (p, h, x): outputs (f, d, h, x + d), 0 <= x + d < W, every frame f that uses p;
//                             SPAN = frame columns x-half .. x+(D-1)+half.  x = -(bs-1-half) stands for all
//                             fully clamped windows x <= -(bs-1-half): lane d's value is written to the
//                             whole run d' >= d of pixel w = x + d (ext.h:152-154 makes the run constant).
// The NCC is symmetric in the two windows (dot and sigma0*sigma1 commute exactly), so one staging layout
// serves both cases.
// Ranked calls (ncc_fast_fixup_ranked, after the all-D kernel): the in-kernel ranking saw the placeholder score 0
// instead of these, so every recomputed one is held against the pixel's best; a pixel whose best is not clear of it by
// the re-ranking margin, or whose index IS the placeholder's disparity, joins the work list of the exact re-scoring
// (once: its flag byte is claimed atomically).  `out` may be null then (nothing was materialised).
constexpr int kFixupBlocks = 2048;

// Loops over the window rows stay rolled (a fully unrolled body is ~40 KB of straight-line code that every
// wavefront executes once -- instruction-fetch bound); BS > 0 unrolls the inner tap loop only.
// Outputs of the fully clamped run are not written here (one store per disparity plane and lane thrashes
// the TLB): the run's value goes to `run_vals[f][h][d_first]` (NaN = keep the fast value) and
// ncc_fixup_runs_kernel spreads it plane by plane.
// A pattern window shared by all frames (single channel) is one item per group of kFixFrames frames: its own side
// (window, mean, deviations) is staged once, the frames' rows follow one another with the next frame's rows already
// on their way (register prefetch) -- the pass is latency-bound, a lone wavefront per item, and this takes the global
// round trips of all but the first frame off its critical path.  Same arithmetic, same order as the generic path.
constexpr int kFixFrames = 2;
constexpr int kFixSpanRegs = 20;       // prefetched SPAN elements per lane (bs * (bs + D - 1) <= 64 * 20)

template <int BS>
__device__ __forceinline__ void fixup_grouped_item(const float* __restrict__ in0, const float* __restrict__ in1,
                                                float* __restrict__ out, float* __restrict__ run_vals,
                                                const float* __restrict__ best, unsigned long long* __restrict__ idx,
                                                float rank_eps,
                                                unsigned* __restrict__ flags, WorkList work, float* sF, float* sFq,
                                                float* sFv,
                                                float* sS, float* sSq, int f_lo, int f_hi, int h, int col, bool run_item,
                                                int H, int W, int D, int bs_rt, int lane) {
  const int bs = BS > 0 ? BS : bs_rt;
  const int half = bs / 2, span = bs + D - 1, taps = bs * bs;
  const float n = (float)taps;
  const long HW = (long)H * W;
  const int span_col0 = col - half;
  // the lane's SPAN element offsets inside a frame (the same for every frame) and the first frame's elements
  float pre[kFixSpanRegs];
  int soff[kFixSpanRegs];
#pragma unroll
  for (int k = 0; k < kFixSpanRegs; ++k) {
    const int i = min(lane + 64 * k, bs * span - 1);
    const int bh = i / span, cc = i - bh * span;
    soff[k] = clampi(h + bh - half, 0, H - 1) * W + clampi(span_col0 + cc, 0, W - 1);
    pre[k] = in0[(long)f_lo * HW + soff[k]];
  }
  // FIX side: the pattern window, its mean (every tap divided before the sum, as the reference does) and deviations
  for (int i0 = lane; i0 < taps; i0 += 64 * 2) {
    float t[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = min(i0 + 64 * u, taps - 1);
      const int bh = i / bs, bw = i - bh * bs;
      t[u] = in1[(long)clampi(h + bh - half, 0, H - 1) * W + clampi(col + bw - half, 0, W - 1)];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (i0 + 64 * u < taps) {
        sF[i0 + 64 * u] = t[u];
        sFq[i0 + 64 * u] = t[u] / n;
      }
  }
  float mu_f = 0.f;
  for (int bh = 0; bh < bs; ++bh) {
#pragma unroll
    for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) mu_f += sFq[bh * BS + bw];
    if (BS == 0)
      for (int bw = 0; bw < bs; ++bw) mu_f += sFq[bh * bs + bw];
  }
  for (int i = lane; i < taps; i += 64) sFv[i] = sF[i] - mu_f;
  float s_f = 0.f;
  for (int bh = 0; bh < bs; ++bh) {
#pragma unroll
    for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) s_f += sFv[bh * BS + bw] * sFv[bh * BS + bw];
    if (BS == 0)
      for (int bw = 0; bw < bs; ++bw) s_f += sFv[bh * bs + bw] * sFv[bh * bs + bw];
  }
  const int rounds = (D + 127) / 128;
  // ranked calls: contenders of the whole item are claimed and pushed at its end, all atomics in flight together
  constexpr int kCand = kFixFrames * 2;
  bool cand[kCand];
  long cpix[kCand];
#pragma unroll
  for (int i = 0; i < kCand; ++i) { cand[i] = false; cpix[i] = 0; }
  auto flush_candidates = [&]() {
    if (!best) return;
    bool tk[kCand];
#pragma unroll
    for (int i = 0; i < kCand; ++i) tk[i] = cand[i] && worklist_claim(flags, cpix[i]);
#pragma unroll
    for (int g = 0; g < kFixFrames; ++g)                       // the two candidates of a frame lie in one image row
      worklist_push2_same_row(tk[2 * g], cpix[2 * g], tk[2 * g + 1], cpix[2 * g + 1], work);
#pragma unroll
    for (int i = 0; i < kCand; ++i) cand[i] = false;
  };
  for (int f = f_lo; f < f_hi; ++f) {
    // this frame's rows come out of the prefetch registers; the next frame's are requested right away
#pragma unroll
    for (int k = 0; k < kFixSpanRegs; ++k)
      if (lane + 64 * k < bs * span) {
        sS[lane + 64 * k] = pre[k];
        sSq[lane + 64 * k] = pre[k] / n;
      }
    if (f + 1 < f_hi) {
#pragma unroll
      for (int k = 0; k < kFixSpanRegs; ++k) pre[k] = in0[(long)(f + 1) * HW + soff[k]];
    }
    // two disparities per lane and pass (d, d + 64): two independent serial chains in flight -- a lone wavefront
    // spends this loop waiting for its own LDS reads and dependent adds
    for (int r = 0; r < rounds; ++r) {
      int dd[2];
      bool bad[2];
      float val[2], mb[2];
      bool won[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        dd[t] = r * 128 + 64 * t + lane;
        const int w = col + dd[t];
        bad[t] = dd[t] < D && w >= 0 && w < W;
        val[t] = 0.f;
        // the pixel's best score and index (ranked calls): requested now, needed after the exact evaluation
        mb[t] = (best && bad[t]) ? best[((long)f * H + h) * W + w] : 0.f;
        won[t] = best && bad[t] && idx[((long)f * H + h) * W + w] == (unsigned long long)dd[t];   // the placeholder came out on top
      }
      const int o0 = min(dd[0], D - 1), o1 = min(dd[1], D - 1);    // clamped: lanes past D read valid LDS, results unused
      if (__any(bad[0] || bad[1])) {
        float mu0 = 0.f, mu1 = 0.f;
        for (int bh = 0; bh < bs; ++bh) {
          const float* q0 = sSq + bh * span + o0;
          const float* q1 = sSq + bh * span + o1;
#pragma unroll
          for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) { mu0 += q0[bw]; mu1 += q1[bw]; }
          if (BS == 0)
            for (int bw = 0; bw < bs; ++bw) { mu0 += q0[bw]; mu1 += q1[bw]; }
        }
        float ss0 = 0.f, ss1 = 0.f, dot0 = 0.f, dot1 = 0.f;
        for (int bh = 0; bh < bs; ++bh) {
          const float* x0 = sS + bh * span + o0;
          const float* x1 = sS + bh * span + o1;
          const float* vf = sFv + bh * bs;
#pragma unroll
          for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) {
            const float v0 = x0[bw] - mu0, v1 = x1[bw] - mu1;
            dot0 += vf[bw] * v0;
            ss0 += v0 * v0;
            dot1 += vf[bw] * v1;
            ss1 += v1 * v1;
          }
          if (BS == 0)
            for (int bw = 0; bw < bs; ++bw) {
              const float v0 = x0[bw] - mu0, v1 = x1[bw] - mu1;
              dot0 += vf[bw] * v0;
              ss0 += v0 * v0;
              dot1 += vf[bw] * v1;
              ss1 += v1 * v1;
            }
        }
        val[0] = 0.f + dot0 / ncc_norm(s_f, ss0);            // "T val = 0; val += dot / norm" (ext.h:142,186)
        val[1] = 0.f + dot1 / ncc_norm(s_f, ss1);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int d = dd[t], w = col + d;
        if (run_item) {
          if (d < D) run_vals[((long)f * H + h) * D + d] = bad[t] ? val[t] : __int_as_float(0x7fc00000);
        } else if (bad[t]) {
          if (out) out[((long)f * D + d) * HW + (long)h * W + w] = val[t];
        }
        if (best) {                                            // wave-uniform: ranked call
          const long pixc = ((long)f * H + h) * W + w;
          // clearly above everything the ranking saw: straight into the pixel's index word (ctd_tail.h); inside the
          // margin of the best, or the placeholder itself came out on top with no such lead: exact re-scoring
          const bool clear = bad[t] && val[t] > mb[t] + rank_margin(rank_eps, mb[t]);
          if (clear) atomicMax(idx + pixc, patch_key(val[t], d));
          const bool contender = bad[t] && !clear && (won[t] || !(val[t] < mb[t] - rank_margin(rank_eps, mb[t])));
          if (rounds == 1) {                                   // one slot per (frame of the item, t): flushed at the end
#pragma unroll
            for (int i = 0; i < kCand; ++i)
              if (i == (f - f_lo) * 2 + t) { cand[i] = contender; cpix[i] = ((long)f * H + h) * W + w; }
          } else {
            const long pix = ((long)f * H + h) * W + w;
            worklist_push(contender && worklist_claim(flags, pix), pix, work);
          }
        }
      }
    }
  }
  flush_candidates();
}

template <int BS>
__global__ __launch_bounds__(256, 2) void ncc_fixup_kernel(const float* __restrict__ in0, const float* __restrict__ in1,
                                                        long in1_frame_stride, float* __restrict__ out,
                                                        unsigned* __restrict__ counters,
                                                        const unsigned long long* __restrict__ list_a,
                                                        const unsigned long long* __restrict__ list_b,
                                                        float* __restrict__ run_vals,
                                                        const float* __restrict__ best,
                                                        unsigned long long* __restrict__ idx, float rank_eps,
                                                        unsigned* __restrict__ flags, WorkList work, int frames, int C, int H,
                                                        int W, int D, int bs_rt) {
  extern __shared__ float lds_fix[];
  const int bs = BS > 0 ? BS : bs_rt;
  const int lane = threadIdx.x & 63;
  const int half = bs / 2, span = bs + D - 1, taps = bs * bs;
  const float n = (float)taps;
  // per-wave staging: FIX window raw / divided by n / minus its mean, SPAN rows raw / divided by n
  float* sF = lds_fix + (threadIdx.x >> 6) * (3 * taps + 2 * bs * span);
  float* sFq = sF + taps;
  float* sFv = sFq + taps;
  float* sS = sFv + taps;
  float* sSq = sS + bs * span;
  const long HW = (long)H * W;
  const unsigned n_a = counters[0], n_b = counters[1];
  // ranked calls: the tail kernel reads the number of listed frame windows from slot 3 and clears slot 0 for the next
  // call's pre-pass (which counts in it) -- no memset launch in front of a call on a prepared pattern
  if (best && blockIdx.x == 0 && threadIdx.x == 0) counters[3] = n_a;
  const unsigned per_b = in1_frame_stride == 0 ? (unsigned)frames : 1u;   // a shared pattern window meets every frame
  // single channel, shared pattern, SPAN small enough for the prefetch registers: kFixFrames frames per item
  const bool grouped = per_b > 1u && C == 1 && bs * span <= 64 * kFixSpanRegs;
  const unsigned groups = grouped ? (per_b + kFixFrames - 1) / kFixFrames : per_b;
  const unsigned n_items = n_a + n_b * groups;
  const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
  const int rounds = (D + 63) / 64;
  for (unsigned item = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); item < n_items; item += n_waves) {
    const bool is_a = item < n_a;
    const unsigned jb = is_a ? 0u : (item - n_a) / groups;
    const unsigned long long e = is_a ? list_a[item] : list_b[jb];
    const int z = (int)(e >> 40), h = (int)((e >> 20) & 0xFFFFF), col = (int)(e & 0xFFFFF) - 0x80000;
    const bool run_item = !is_a && col == -(bs - 1 - half);
    if (grouped && !is_a) {
      const int f_lo = (int)((item - n_a) - jb * groups) * kFixFrames;
      fixup_grouped_item<BS>(in0, in1, out, run_vals, best, idx, rank_eps, flags, work, sF, sFq, sFv, sS, sSq,
                             f_lo, min(frames, f_lo + kFixFrames), h, col, run_item, H, W, D, bs, lane);
      continue;
    }
    const int f = (is_a || per_b == 1u) ? z / C : (int)((item - n_a) - jb * groups);
    const float* fix_img = is_a ? in0 + (long)f * C * HW : in1 + (long)f * in1_frame_stride;
    const float* span_img = is_a ? in1 + (long)f * in1_frame_stride : in0 + (long)f * C * HW;
    const int span_col0 = is_a ? col - half - (D - 1) : col - half;
    int staged_c = -1;
    float mu_f = 0.f, s_f = 0.f;
    for (int r = 0; r < rounds; ++r) {
      const int d = r * 64 + lane;
      const int w = is_a ? col : col + d;
      // every output of a listed window is recomputed (the fast kernels wrote NaN there)
      const bool bad = d < D && w >= 0 && w < W;
      float val = 0.f;
      const float mbest = (best && bad) ? best[((long)f * H + h) * W + w] : 0.f;   // ranked calls: needed at the end
      const bool won = best && bad && idx[((long)f * H + h) * W + w] == (unsigned long long)d;   // the placeholder came out on top
      if (__any(bad)) {
        for (int c = 0; c < C; ++c) {
          if (staged_c != c) {
            staged_c = c;
            // batches of independent loads: a lone wavefront must not pay one memory round trip per element
            for (int i0 = lane; i0 < taps; i0 += 64 * 2) {
              float t[2];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int i = min(i0 + 64 * u, taps - 1);
                const int bh = i / bs, bw = i - bh * bs;
                t[u] = fix_img[(long)c * HW + (long)clampi(h + bh - half, 0, H - 1) * W + clampi(col + bw - half, 0, W - 1)];
              }
#pragma unroll
              for (int u = 0; u < 2; ++u)
                if (i0 + 64 * u < taps) {
                  sF[i0 + 64 * u] = t[u];
                  sFq[i0 + 64 * u] = t[u] / n;            // the reference divides every tap before summing
                }
            }
            for (int i0 = lane; i0 < bs * span; i0 += 64 * 8) {
              float t[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + 64 * u, bs * span - 1);
                const int bh = i / span, cc = i - bh * span;
                t[u] = span_img[(long)c * HW + (long)clampi(h + bh - half, 0, H - 1) * W + clampi(span_col0 + cc, 0, W - 1)];
              }
#pragma unroll
              for (int u = 0; u < 8; ++u)
                if (i0 + 64 * u < bs * span) {
                  sS[i0 + 64 * u] = t[u];
                  sSq[i0 + 64 * u] = t[u] / n;
                }
            }
            // the FIX side (mean, deviations, sigma) is the same for every disparity: once per staging
            mu_f = 0.f;
            for (int bh = 0; bh < bs; ++bh) {
#pragma unroll
              for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) mu_f += sFq[bh * BS + bw];
              if (BS == 0)
                for (int bw = 0; bw < bs; ++bw) mu_f += sFq[bh * bs + bw];
            }
            for (int i = lane; i < taps; i += 64) sFv[i] = sF[i] - mu_f;
            s_f = 0.f;
            for (int bh = 0; bh < bs; ++bh) {
#pragma unroll
              for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) s_f += sFv[bh * BS + bw] * sFv[bh * BS + bw];
              if (BS == 0)
                for (int bw = 0; bw < bs; ++bw) s_f += sFv[bh * bs + bw] * sFv[bh * bs + bw];
            }
          }
          if (bad) {
            const int off = is_a ? (D - 1) - d : d;
            float mu_s = 0.f, s_s = 0.f, dot = 0.f;
            for (int bh = 0; bh < bs; ++bh) {
              const float* q = sSq + bh * span + off;
#pragma unroll
              for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) mu_s += q[bw];
              if (BS == 0)
                for (int bw = 0; bw < bs; ++bw) mu_s += q[bw];
            }
            for (int bh = 0; bh < bs; ++bh) {
              const float* x = sS + bh * span + off;
              const float* vf = sFv + bh * bs;
#pragma unroll
              for (int bw = 0; bw < (BS > 0 ? BS : 0); ++bw) {
                const float vs = x[bw] - mu_s;
                dot += vf[bw] * vs;
                s_s += vs * vs;
              }
              if (BS == 0)
                for (int bw = 0; bw < bs; ++bw) {
                  const float vs = x[bw] - mu_s;
                  dot += vf[bw] * vs;
                  s_s += vs * vs;
                }
            }
            val += dot / ncc_norm(s_f, s_s);              // ext.h:185-186 (sigma0 * sigma1 commutes)
          }
        }
      }
      if (run_item) {
        if (d < D) run_vals[((long)f * H + h) * D + d] = bad ? val : __int_as_float(0x7fc00000);
      } else if (bad) {
        if (out) out[((long)f * D + d) * HW + (long)h * W + w] = val;
      }
      if (best) {                                            // wave-uniform: ranked call
        bool take = false;
        const long pix = ((long)f * H + h) * W + w;
        const bool clear = bad && val > mbest + rank_margin(rank_eps, mbest);      // (see the grouped path)
        if (clear) atomicMax(idx + pix, patch_key(val, d));
        if (bad && !clear && (won || !(val < mbest - rank_margin(rank_eps, mbest)))) take = worklist_claim(flags, pix);
        worklist_push(take, pix, work);
      }
    }
  }
}

// Second half of the run items (ctd_tail.h: runs_role), as a kernel of its own for the unranked call; a ranked call runs
// the same role inside its tail kernel (argmax_rerank.hip).
__global__ __launch_bounds__(256) void ncc_fixup_runs_kernel(float* __restrict__ out, const float* __restrict__ run_vals,
                                                             const unsigned* __restrict__ counters,
                                                             const unsigned long long* __restrict__ run_rows, int per_frame,
                                                             int frames, int C, int H, int W, int D, int bs,
                                                             unsigned* __restrict__ rank_counter) {
  extern __shared__ int s_rows[];                          // up to C * H rows of this frame's pattern
  // the work-list counter of the ranking pass that may follow (argmax_rerank.hip) lives at the start of the
  // workspace, which the volume kernel is done with by now: cleared here instead of by a memset of its own
  if (rank_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *rank_counter = 0u;
  runs_role(out, run_vals, counters, run_rows, per_frame, C, H, W, D, bs, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y, s_rows);
}

// 1 / (sa * sb + 1e-8) from the RECIPROCAL deviations the pre-pass stores: t = ra * rb, ONE multiply.  The reference's
// 1e-8 changes the quotient by the relative amount 1e-8 * t: below 2.1e-6 because the pre-pass lists every window whose
// deviation is under kDevFloor = 7e-2 (t <= 1 / kDevFloor^2 = 204), and listed windows go through the fix-up pass in the
// reference's own arithmetic.  (Until round 3 the first-order term t * (1 - 1e-8 t) was kept and the floor was 6e-3:
// two more instructions on each of the eight scores of a lane and row -- 10 % of the volume kernels' vector work, which
// is what bounds the all-D kernel; LCN'd images have t ~ 0.01, where the term is 1e-10.)
__device__ inline float ncc_inv_norm(float ra, float rb) { return ra * rb; }

// cross-lane helpers (wave64) -----------------------------------------------------------
__device__ inline float lane_prev1(float x) {   // result[l] = x[l-1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ inline float lane_next1(float x) {   // result[l] = x[l+1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}
__device__ inline float lane_gather(float x, int byte_addr) {   // result[l] = x[byte_addr[l] / 4]
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(x)));
}

// horizontal window sum: s[l] = sum_{k=0..BS-1} x[l - HALF + k]
template <int BS>
__device__ inline float lane_window_sum(float x, int lane) {
  constexpr int HALF = BS / 2;
  if constexpr (BS == 9) {
    float s3 = x + lane_prev1(x) + lane_next1(x);
    float m3 = lane_gather(s3, ((lane - 3) & 63) * 4);
    float p3 = lane_gather(s3, ((lane + 3) & 63) * 4);
    return s3 + m3 + p3;
  } else if constexpr (BS == 3) {
    return x + lane_prev1(x) + lane_next1(x);
  } else {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < BS; ++k) s += lane_gather(x, ((lane - HALF + k) & 63) * 4);
    return s;
  }
}

// ------------------------------------------------------------------------------------
// main kernel.  grid (w tiles, bands, frames * d groups), block 64 * (kFWaves + 1).
// Vertical sum of BS rows: for BS == 9 the 3+3+3 tree (ring of 2 products + 6 triple
// sums); other BS keep a ring of the last BS-1 products.
// ------------------------------------------------------------------------------------
// (dma_dword / dma_quad: ctd_wave.h)

constexpr int gcd_ce(int a, int b) { return b == 0 ? a : gcd_ce(b, a % b); }
constexpr int lcm_ce(int a, int b) { return a / gcd_ce(a, b) * b; }

// (wait_vmcnt / wait_lgkmcnt0 / wg_barrier: ctd_wave.h)

template <int BS, bool ACCUM>
__global__ __launch_bounds__(64 * (kFWaves + 1)) void ncc_fast_kernel(
    const float* __restrict__ ac, const float* __restrict__ m0, const float* __restrict__ v0,
    const float* __restrict__ bc, const float* __restrict__ m1, const float* __restrict__ v1, long st1_frame_stride,
    float* __restrict__ out, int C, int c, int H, int W, int D, int band_rows, int n_dgroups, int Wp, int W1,
    int xoff, int w_start) {
  constexpr int HALF = BS / 2;
  constexpr int TAIL = BS - 1 - HALF;          // window rows/cols after the centre
  constexpr int WOUT = 64 - (BS - 1);          // output columns per wavefront
  constexpr int UNROLL = (BS == 9) ? 6 : (BS - 1);
  extern __shared__ float lds[];               // [kFBufs][kFRows][kFPack]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.z / n_dgroups, dg = blockIdx.z - f * n_dgroups;
  const int w_lo = w_start + blockIdx.x * WOUT;
  const int h_lo = blockIdx.y * band_rows;
  const int h_hi = min(h_lo + band_rows, H);   // exclusive
  const long HW = (long)H * W;
  const int r_begin = h_lo - HALF, r_end = h_hi - 1 + TAIL;       // inclusive product rows
  const int n_rows = r_end - r_begin + 1;
  constexpr int STEP = lcm_ce(UNROLL, kFRows);                     // rows per outer iteration
  const int n_iters = (n_rows + STEP - 1) / STEP;
  const int n_chunks = n_iters * (STEP / kFRows);

  const float* a_img = ac + ((long)f * C + c) * H * Wp + 4;      // +4: column c lives at c + 4
  const float* b_img = bc + (long)f * st1_frame_stride + (long)c * H * W1;
  const float* m0i = m0 + ((long)f * C + c) * H * Wp + 4;
  const float* v0i = v0 + ((long)f * C + c) * H * Wp + 4;
  const float* m1i = m1 + (long)f * st1_frame_stride + (long)c * H * W1;
  const float* v1i = v1 + (long)f * st1_frame_stride + (long)c * H * W1;
  const int xb = w_lo - HALF - (dg * kFDG + kFDG - 1);      // unclamped pattern column of span slot 0

  if (wave == kFWaves) {
    // ------------------------------ loader wavefront ------------------------------
    const int wa = clampi(w_lo - HALF + lane, 0, W - 1);
    const int q1 = 64 + lane;                                 // second DMA of a span: slots 64..78
    const int sc0 = clampi(xb + lane, -xoff, W - 1) + xoff, sc1 = clampi(xb + q1, -xoff, W - 1) + xoff;
    const bool second = q1 < kFSpan;
    auto issue_chunk = [&](int chunk) {
      float* buf = lds + (chunk % kFBufs) * (kFRows * kFPack);
#pragma unroll
      for (int s = 0; s < kFRows; ++s) {
        const int r = r_begin + chunk * kFRows + s;
        const int rc = clampi(r, 0, H - 1);
        const int hs = clampi(r - TAIL, 0, H - 1);
        float* pk = buf + s * kFPack;
        dma_dword(a_img + (long)rc * Wp + wa, pk);
        dma_dword(m0i + (long)hs * Wp + wa, pk + 64);
        dma_dword(v0i + (long)hs * Wp + wa, pk + 128);
        dma_dword(b_img + (long)rc * W1 + sc0, pk + 192);
        dma_dword(m1i + (long)hs * W1 + sc0, pk + 192 + kFSpanPad);
        dma_dword(v1i + (long)hs * W1 + sc0, pk + 192 + 2 * kFSpanPad);
        // lanes >= 15 re-fetch slot 78's column into the pad slot / next array's head;
        // harmless: the pad is never read and the next array is rewritten by ITS OWN DMA
        // only if issued later -- so issue the tails BEFORE nothing depends on order:
        if (second) {
          dma_dword(b_img + (long)rc * W1 + sc1, pk + 192 + 64);
          dma_dword(m1i + (long)hs * W1 + sc1, pk + 192 + kFSpanPad + 64);
          dma_dword(v1i + (long)hs * W1 + sc1, pk + 192 + 2 * kFSpanPad + 64);
        }
      }
    };
    constexpr int L = kFRows * kFDmaPerRow;                   // DMA instructions per chunk
#pragma unroll
    for (int k = 0; k < kFBufs - 1; ++k)
      if (k < n_chunks) issue_chunk(k);
    // chunk 0 landed when at most (kFBufs-2) younger chunks are still in flight
    if (n_chunks >= kFBufs - 1) wait_vmcnt<L*(kFBufs - 2)>(); else wait_vmcnt<0>();
    wg_barrier();
    for (int ch = 0; ch < n_chunks; ++ch) {
      // buffer (ch-1) % kFBufs was released by the consumers at the previous barrier
      const int nxt = ch + kFBufs - 1;
      if (nxt < n_chunks) {
        issue_chunk(nxt);
        wait_vmcnt<L*(kFBufs - 2)>();                         // chunk ch+1 has landed
      } else {
        wait_vmcnt<0>();
      }
      wg_barrier();
    }
    return;
  }

  // -------------------------------- consumer wavefronts --------------------------------
  const int d_base = dg * kFDG + wave * kFND;
  const int w0 = w_lo - HALF + lane;           // unclamped product column == output column
  float* vol = out + (long)f * D * HW;
  const bool lane_out = (lane >= HALF) && (lane < 64 - TAIL) && (w0 < W);
  const int bq = lane + (kFDG - 1) - wave * kFND;   // span slot of (lane, j = 0); j-th disparity reads bq - j

  float P[kFND][BS == 9 ? 2 : BS - 1];
  float T[kFND][BS == 9 ? 6 : 1];
#pragma unroll
  for (int j = 0; j < kFND; ++j) {
#pragma unroll
    for (int k = 0; k < (BS == 9 ? 2 : BS - 1); ++k) P[j][k] = 0.f;
#pragma unroll
    for (int k = 0; k < (BS == 9 ? 6 : 1); ++k) T[j][k] = 0.f;
  }

  wg_barrier();                                                    // chunk 0 is in LDS
  int chunk = 0;
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int u = 0; u < STEP; ++u) {
      const int r = r_begin + it * STEP + u;
      const float* pk = lds + ((chunk % kFBufs) * kFRows + (u % kFRows)) * kFPack;
      const float a = pk[lane];
      const float mav = pk[64 + lane], sav = pk[128 + lane];
      float bv[kFND], mbv[kFND], sbv[kFND];
#pragma unroll
      for (int j = 0; j < kFND; ++j) {
        bv[j] = pk[192 + bq - j];
        mbv[j] = pk[192 + kFSpanPad + bq - j];
        sbv[j] = pk[192 + 2 * kFSpanPad + bq - j];
      }
      const int h = r - TAIL;                                     // output row completed by product row r
      const bool row_out = (h >= h_lo) && (h < h_hi);             // wave-uniform
      const float nma = mav;                                      // -bs^2 * (window mean), from the pre-pass
#pragma unroll
      for (int j = 0; j < kFND; ++j) {
        const float p = a * bv[j];
        float v;
        if constexpr (BS == 9) {
          const float t3 = p + P[j][(u + 1) % 2] + P[j][u % 2];
          P[j][u % 2] = p;
          v = t3 + T[j][(u + 3) % 6] + T[j][u % 6];
          T[j][u % 6] = t3;
        } else {
          v = p;
#pragma unroll
          for (int k = 0; k < BS - 1; ++k) v += P[j][k];
          P[j][u % (BS - 1)] = p;
        }
        const float s = lane_window_sum<BS>(v, lane);
        const float cov = fmaf(nma, mbv[j], s);
        float val = cov * ncc_inv_norm(sav, sbv[j]);
        const int d = d_base + j;
        if (lane_out && row_out && d < D) {
          const long o = (long)d * HW + (long)h * W + w0;
          if (ACCUM) val += vol[o];
          vol[o] = val;
        }
      }
      if ((u % kFRows) == kFRows - 1) {                            // chunk consumed: hand the buffer back
        wait_lgkmcnt0();
        wg_barrier();
        ++chunk;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// WIDE kernel: every lane owns 4 adjacent product columns, a wavefront 256 of them.
// A bs <= 9 window then reaches only into the two neighbouring lanes, so the horizontal
// window sum needs nothing but +-1 lane DPP shifts of per-lane prefix / suffix sums
// (no LDS crossbar traffic), 62 of 64 lanes produce outputs, frame-side LDS reads are
// one ds_read_b128 per array and each lane stores 16 contiguous bytes (1 KB per wave
// instruction).  Same loader / consumer split and LDS ring as the narrow kernel above,
// which remains in use for the columns left over when W is not a multiple of 248.
// ------------------------------------------------------------------------------------
constexpr int kWCols = 4;                     // product columns per lane
constexpr int kWND = 2;                       // disparities per lane
constexpr int kWWaves = 8;                    // consumer wavefronts per workgroup
constexpr int kWDG = kWND * kWWaves;          // disparities per workgroup (8)
constexpr int kWTile = 64 * kWCols;           // product columns per wavefront (256)
constexpr int kWOut = 62 * kWCols;            // output columns per wavefront (248)
constexpr int kWSpan = kWTile + kWDG - 1;     // 263 pattern columns per row
constexpr int kWSpanPad = (kWSpan + 1 + 3) / 4 * 4;   // multiple of 4, > kWSpan
constexpr int kWPack = 3 * kWTile + 3 * kWSpanPad;   // 1560 floats per staged row
constexpr int kWRows = 3;                     // rows per LDS chunk
constexpr int kWBufs = 3;                     // chunks in the ring
constexpr int kWDmaPerRow = 3 + 3 * 2;        // dwordx4 LDS-DMA instructions per row

// Four floats starting OFF slots after the lane's own quad of a 16-byte aligned LDS array:
// one or two conflict-free ds_read_b128 (a stride-4 ds_read_b32 pattern is a 4-way bank conflict).

template <int OFF>
__device__ inline void lds_read4(const float* arr, int lane, float (&o)[4]) {
  constexpr int Q = OFF / 4, S = OFF % 4;
  // the empty asm "uses" all four elements: it keeps the compiler from narrowing the loads to the
  // elements actually consumed (ds_read_b32 / read2 at a 16-byte lane stride, which is exactly the
  // conflicting pattern this helper avoids)
  f32x4 A = *(const f32x4*)(arr + 4 * (lane + Q));
  asm("" : "+v"(A));
  const float a[4] = {A[0], A[1], A[2], A[3]};
  if constexpr (S == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = a[i];
  } else {
    f32x4 B = *(const f32x4*)(arr + 4 * (lane + Q + 1));
    asm("" : "+v"(B));
    const float b[4] = {B[0], B[1], B[2], B[3]};
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (S + i < 4) ? a[(S + i) & 3] : b[(S + i) & 3];
  }
}

// Five floats starting OFF slots after the lane's own quad, from two aligned quads: element k of `lo5` is
// slot OFF + k.  Two adjacent disparities of a lane (span offsets OFF+1 and OFF) read the same two quads.
template <int OFF>
__device__ inline void lds_read5(const float* arr, int lane, float (&o)[5]) {
  constexpr int Q = OFF / 4, S = OFF % 4;
  f32x4 A = *(const f32x4*)(arr + 4 * (lane + Q));
  f32x4 B = *(const f32x4*)(arr + 4 * (lane + Q + 1));
  asm("" : "+v"(A));
  asm("" : "+v"(B));
  const float e[8] = {A[0], A[1], A[2], A[3], B[0], B[1], B[2], B[3]};
#pragma unroll
  for (int k = 0; k < 5; ++k) o[k] = e[S + k];
}

// (window_combine4: ctd_wave.h)

template <int BS, bool ACCUM, bool VEC4, int WAVE>
__device__ __forceinline__ void wide_consume(const float* lds, float* __restrict__ out, int f, int dg, int lane,
                                             int w_lo, int c_lo, int h_lo, int h_hi, int r_begin, int n_iters, int H,
                                             int W, int D) {
  constexpr int HALF = BS / 2;
  constexpr int TAIL = BS - 1 - HALF;
  constexpr int UNROLL = (BS == 9) ? 6 : (BS - 1);
  constexpr int STEP = lcm_ce(UNROLL, kWRows);
  const long HW = (long)H * W;
  const int d_base = dg * kWDG + WAVE * kWND;
  const int c0 = c_lo + kWCols * lane;         // unclamped first column of this lane
  float* vol = out + (long)f * D * HW;
  const bool lane_out = (lane >= 1) && (lane <= 62) && (c0 < W);

  float P[kWND][kWCols][BS == 9 ? 2 : BS - 1];
  float T[kWND][kWCols][BS == 9 ? 6 : 1];
#pragma unroll
  for (int j = 0; j < kWND; ++j)
#pragma unroll
    for (int i = 0; i < kWCols; ++i) {
#pragma unroll
      for (int k = 0; k < (BS == 9 ? 2 : BS - 1); ++k) P[j][i][k] = 0.f;
#pragma unroll
      for (int k = 0; k < (BS == 9 ? 6 : 1); ++k) T[j][i][k] = 0.f;
    }

  // All LDS operands of one product row (and of the output row it completes).
  struct RowOps {
    float a[4], ma[4], sa[4];
    float b[kWND][4], mb[kWND][4], sb[kWND][4];
  };
  constexpr int kOff0 = (kWDG - 1) - WAVE * kWND;                  // span slot offset of disparity j = 0
  auto load_row = [&](const float* pk) {
    RowOps o;
    lds_read4<0>(pk, lane, o.a);
    lds_read4<0>(pk + kWTile, lane, o.ma);
    lds_read4<0>(pk + 2 * kWTile, lane, o.sa);
    lds_read4<kOff0>(pk + 3 * kWTile, lane, o.b[0]);
    lds_read4<kOff0>(pk + 3 * kWTile + kWSpanPad, lane, o.mb[0]);
    lds_read4<kOff0>(pk + 3 * kWTile + 2 * kWSpanPad, lane, o.sb[0]);
    if constexpr (kWND == 2) {
      lds_read4<(kOff0 > 0 ? kOff0 - 1 : 0)>(pk + 3 * kWTile, lane, o.b[kWND - 1]);
      lds_read4<(kOff0 > 0 ? kOff0 - 1 : 0)>(pk + 3 * kWTile + kWSpanPad, lane, o.mb[kWND - 1]);
      lds_read4<(kOff0 > 0 ? kOff0 - 1 : 0)>(pk + 3 * kWTile + 2 * kWSpanPad, lane, o.sb[kWND - 1]);
    }
    return o;
  };
  static_assert(kWND == 1 || kWND == 2, "load_row spells out one or two disparities");

  wg_barrier();                                                    // chunk 0 is in LDS
  int chunk = 0;
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int u = 0; u < STEP; ++u) {
      const int r = r_begin + it * STEP + u;
      const bool last_of_chunk = (u % kWRows) == kWRows - 1;
      // every LDS operand of the row is requested up front (15 ds_read_b128 in flight)
      const RowOps cur = load_row(lds + ((chunk % kWBufs) * kWRows + (u % kWRows)) * kWPack);
      float nma[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) nma[i] = cur.ma[i];              // -bs^2 * (window mean), from the pre-pass
      const int h = r - TAIL;
      const bool row_out = (h >= h_lo) && (h < h_hi);             // wave-uniform
#pragma unroll
      for (int j = 0; j < kWND; ++j) {
        float x[kWCols];
#pragma unroll
        for (int i = 0; i < kWCols; ++i) {
          const float p = cur.a[i] * cur.b[j][i];
          if constexpr (BS == 9) {
            const float t3 = p + P[j][i][(u + 1) % 2] + P[j][i][u % 2];
            P[j][i][u % 2] = p;
            x[i] = t3 + T[j][i][(u + 3) % 6] + T[j][i][u % 6];
            T[j][i][u % 6] = t3;
          } else {
            float v = p;
#pragma unroll
            for (int k = 0; k < BS - 1; ++k) v += P[j][i][k];
            P[j][i][u % (BS - 1)] = p;
            x[i] = v;
          }
        }
        // horizontal window sums of the lane's 4 columns from prefix / suffix sums of the
        // neighbouring lanes: out_i = suffix_prev(i - HALF + 4) + own(i-HALF .. i+TAIL) + prefix_next(i + TAIL - 4)
        float pre[kWCols], suf[kWCols];                            // pre[k] = x0..xk, suf[k] = xk..x3
        pre[0] = x[0];
#pragma unroll
        for (int k = 1; k < kWCols; ++k) pre[k] = pre[k - 1] + x[k];
        suf[kWCols - 1] = x[kWCols - 1];
#pragma unroll
        for (int k = kWCols - 2; k >= 0; --k) suf[k] = suf[k + 1] + x[k];
        float s[kWCols];
        if constexpr (BS == 9) {
          // window = previous lane's columns i..3, all four own columns, next lane's columns 0..i
          window_combine4(suf, pre[kWCols - 1], pre, s);
        } else {
#pragma unroll
          for (int i = 0; i < kWCols; ++i) {
            const int lo = i - HALF, hi = i + TAIL;                // window in own-lane column units
            const int o_lo = lo < 0 ? 0 : lo, o_hi = hi > kWCols - 1 ? kWCols - 1 : hi;
            float own;
            if (o_lo == 0) own = pre[o_hi];
            else if (o_hi == kWCols - 1) own = suf[o_lo];
            else { own = x[o_lo]; for (int k = o_lo + 1; k <= o_hi; ++k) own += x[k]; }
            float acc = own;
            if (lo < 0) acc = lane_prev1(suf[lo + kWCols]) + acc;  // previous lane's columns lo+4 .. 3
            if (hi > kWCols - 1) acc = acc + lane_next1(pre[hi - kWCols]);   // next lane's columns 0 .. hi-4
            s[i] = acc;
          }
        }
        float val[kWCols];
#pragma unroll
        for (int i = 0; i < kWCols; ++i) {
          const float cov = fmaf(nma[i], cur.mb[j][i], s[i]);
          val[i] = cov * ncc_inv_norm(cur.sa[i], cur.sb[j][i]);
        }
        const int d = d_base + j;
        if (row_out && lane_out && d < D) {
          float* o = vol + (long)d * HW + (long)h * W + c0;
          if constexpr (VEC4) {   // W % 4 == 0 and 16-byte aligned volume: c0 < W implies c0 + 3 < W
            float4 v4 = make_float4(val[0], val[1], val[2], val[3]);
            if (ACCUM) {
              const float4 old = *(const float4*)o;
              v4.x += old.x; v4.y += old.y; v4.z += old.z; v4.w += old.w;
            }
            __builtin_nontemporal_store(f32x4{v4.x, v4.y, v4.z, v4.w}, (f32x4*)o);   // see ncc_fast_t256_kernel
          } else {
#pragma unroll
            for (int i = 0; i < kWCols; ++i)
              if (c0 + i < W) o[i] = ACCUM ? o[i] + val[i] : val[i];
          }
        }
      }
      if (last_of_chunk) {
        wait_lgkmcnt0();
        wg_barrier();
        ++chunk;
      }
    }
  }
}

template <int BS, bool ACCUM, bool VEC4>
__global__ __launch_bounds__(64 * (kWWaves + 1)) void ncc_fast_wide_kernel(
    const float* __restrict__ ac, const float* __restrict__ m0, const float* __restrict__ v0,
    const float* __restrict__ bc, const float* __restrict__ m1, const float* __restrict__ v1, long st1_frame_stride,
    float* __restrict__ out, int C, int c, int H, int W, int D, int band_rows, int n_dgroups, int Wp, int W1,
    int xoff) {
  constexpr int HALF = BS / 2;
  constexpr int TAIL = BS - 1 - HALF;
  static_assert(HALF <= kWCols && TAIL <= kWCols, "window must stay inside the neighbouring lanes");
  constexpr int UNROLL = (BS == 9) ? 6 : (BS - 1);
  constexpr int STEP = lcm_ce(UNROLL, kWRows);
  extern __shared__ float lds[];               // [kWBufs][kWRows][kWPack]

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.z / n_dgroups, dg = blockIdx.z - f * n_dgroups;
  const int w_lo = blockIdx.x * kWOut;
  const int h_lo = blockIdx.y * band_rows;
  const int h_hi = min(h_lo + band_rows, H);
  const int r_begin = h_lo - HALF, r_end = h_hi - 1 + TAIL;
  const int n_rows = r_end - r_begin + 1;
  const int n_iters = (n_rows + STEP - 1) / STEP;
  const int n_chunks = n_iters * (STEP / kWRows);

  const float* a_img = ac + ((long)f * C + c) * H * Wp + 4;      // +4: column c lives at c + 4
  const float* b_img = bc + (long)f * st1_frame_stride + (long)c * H * W1;
  const float* m0i = m0 + ((long)f * C + c) * H * Wp + 4;
  const float* v0i = v0 + ((long)f * C + c) * H * Wp + 4;
  const float* m1i = m1 + (long)f * st1_frame_stride + (long)c * H * W1;
  const float* v1i = v1 + (long)f * st1_frame_stride + (long)c * H * W1;
  const int c_lo = w_lo - kWCols;                              // unclamped product column of slot 0
  const int xb = c_lo - (dg * kWDG + kWDG - 1);                // unclamped pattern column of span slot 0

  if (wave == kWWaves) {
    // ------------------------------ loader wavefront ------------------------------
    // 16 bytes per lane and DMA: one instruction moves a whole 256-column array row.  All sources are
    // 16-byte aligned by construction (padded planes, see fast_workspace); lanes past the image are
    // clamped to the last quad, which only ever feeds columns that produce no output.
    const int aq = min(c_lo + kWCols * lane, Wp - 8);                         // frame quad of this lane (>= -4)
    const int sq0 = min(xb + xoff + kWCols * lane, W1 - kWCols);              // pattern quad, slots 0..255
    const int sq1 = min(xb + xoff + kWTile + kWCols * lane, W1 - kWCols);     // slots 256.. (first 2 lanes)
    const bool tail_lane = kWTile + kWCols * lane < kWSpan;
    auto issue_chunk = [&](int chunk) {
      float* buf = lds + (chunk % kWBufs) * (kWRows * kWPack);
#pragma unroll
      for (int s = 0; s < kWRows; ++s) {
        const int r = r_begin + chunk * kWRows + s;
        const int rc = clampi(r, 0, H - 1);
        const int hs = clampi(r - TAIL, 0, H - 1);
        float* pk = buf + s * kWPack;
        dma_quad(a_img + (long)rc * Wp + aq, pk);
        dma_quad(m0i + (long)hs * Wp + aq, pk + kWTile);
        dma_quad(v0i + (long)hs * Wp + aq, pk + 2 * kWTile);
        dma_quad(b_img + (long)rc * W1 + sq0, pk + 3 * kWTile);
        dma_quad(m1i + (long)hs * W1 + sq0, pk + 3 * kWTile + kWSpanPad);
        dma_quad(v1i + (long)hs * W1 + sq0, pk + 3 * kWTile + 2 * kWSpanPad);
        if (tail_lane) {
          dma_quad(b_img + (long)rc * W1 + sq1, pk + 3 * kWTile + kWTile);
          dma_quad(m1i + (long)hs * W1 + sq1, pk + 3 * kWTile + kWSpanPad + kWTile);
          dma_quad(v1i + (long)hs * W1 + sq1, pk + 3 * kWTile + 2 * kWSpanPad + kWTile);
        }
      }
    };
    constexpr int L = kWRows * kWDmaPerRow;                   // DMA instructions per chunk
    static_assert(L * (kWBufs - 2) < 64, "in-flight DMA count must fit vmcnt");
#pragma unroll
    for (int k = 0; k < kWBufs - 1; ++k)
      if (k < n_chunks) issue_chunk(k);
    if (n_chunks >= kWBufs - 1) wait_vmcnt<L*(kWBufs - 2)>(); else wait_vmcnt<0>();
    wg_barrier();
    for (int ch = 0; ch < n_chunks; ++ch) {
      const int nxt = ch + kWBufs - 1;
      if (nxt < n_chunks) {
        issue_chunk(nxt);
        wait_vmcnt<L*(kWBufs - 2)>();
      } else {
        wait_vmcnt<0>();
      }
      wg_barrier();
    }
    return;
  }

  // -------------------------------- consumer wavefronts --------------------------------
  // the span offset of a wave's disparities is a compile-time constant of its wave index,
  // which turns the unaligned 4-float pattern reads into aligned ds_read_b128 pairs
#define CTD_WCASE(WV) \
  case WV: wide_consume<BS, ACCUM, VEC4, (WV < kWWaves ? WV : 0)>(lds, out, f, dg, lane, w_lo, c_lo, h_lo, h_hi, r_begin, n_iters, H, W, D); break;
  switch (wave) {
    CTD_WCASE(0) CTD_WCASE(1) CTD_WCASE(2) CTD_WCASE(3) CTD_WCASE(4) CTD_WCASE(5) CTD_WCASE(6) CTD_WCASE(7)
    default: break;
  }
#undef CTD_WCASE
}

// ------------------------------------------------------------------------------------
// TILE-256 kernel (bs == 9, W % 4 == 0): the production kernel.
// Same consumer pipeline as the wide kernel, but a wavefront's 64 lanes own exactly 256
// OUTPUT columns (1 KB-aligned 16-byte stores, every lane valid, W = 512 is two tiles with
// no leftover columns).  The two halo quads a tile needs (4 product columns left of lane 0,
// 4 right of lane 63) are computed by the LOADER wavefront: its lanes hold, per consumer
// wave and disparity, the vertical ring of the halo quad and publish its suffix / prefix
// sums into the staged row, one chunk ahead of the consumers and under the same barrier.
// Volume stores that are not 128-byte aligned cost ~25 % of HBM write bandwidth
// (tools/ubench_store.hip), and per-CU operand staging is limited to ~10 B/clk
// (tools/ubench_struct.hip), hence >= 12 disparities per workgroup (14: seven consumer wavefronts of two).
// ------------------------------------------------------------------------------------
constexpr int kTWaves = 7;                     // consumer wavefronts per workgroup.  7 (+ loader) = two 8-wave workgroups per CU at
                                               // 128 VGPRs = exactly 4 waves on every SIMD; with 6 two SIMDs carry 4 waves and two
                                               // carry 3, and the chunk barrier makes the lighter ones wait (measured: 7 is 9 % faster
                                               // although 10 groups of 14 disparities compute 140 for D = 128)
constexpr int kTND = 2;                        // disparities per lane
constexpr int kTDG = kTWaves * kTND;           // 14 disparities per workgroup
constexpr int kTTile = 256;                    // output columns per workgroup
constexpr int kTA = kTTile + 8;                // frame-side array: 4 halo columns either side
constexpr int kTSpanPad = (kTA + kTDG - 1 + 1 + 3) / 4 * 4;   // multiple of 4, > span
static_assert(kTA + kTDG - 1 < kTSpanPad, "pattern span must fit its padded array");
constexpr int kTHalo = kTWaves * kTND * 2 * 4; // [wave][j][side][4] halo sums
constexpr int kTPack = 3 * kTA + 3 * kTSpanPad + kTHalo;   // 1760 floats per staged row
constexpr int kTRows = 3;
constexpr int kTBufs = 3;
constexpr int kTDmaPerRow = 12;                // 3 x 2 frame-side + 3 x 2 pattern-side dwordx4 DMAs
constexpr int kTOffB = 3 * kTA, kTOffH = 3 * kTA + 3 * kTSpanPad;

// KS = sub-quad shift of the wavefront's pattern-side operands, (12 - 2 * wave) % 4: 0 for even consumer wavefronts, 2
// for odd ones.  Everything else that depends on the wavefront index (quad offset, halo slot) is a run-time scalar, so
// the kernel carries TWO copies of the consumer loop, not seven: with one copy per wavefront the seven hot loops of a
// workgroup (plus the loader's) are a 77 KB instruction working set against a 64 KB instruction cache shared by two CUs.
template <bool ACCUM, int KS>
__device__ __forceinline__ void t256_consume(float* lds, float* __restrict__ out, int WAVE, int f, int dg, int lane,
                                             int w_lo, int h_lo, int h_hi, int r_begin, int n_iters, int H, int W, int D) {
  constexpr int TAIL = 4, STEP = lcm_ce(6, kTRows);               // block size 9
  const long HW = (long)H * W;
  const int d_base = dg * kTDG + WAVE * kTND;
  // per-lane column arithmetic is kept to ONE register, 4 * lane: everything else about the column tile (w_lo) goes
  // into scalar bases -- the consumers run at the 128-VGPR limit of four wavefronts per SIMD
  const unsigned l4 = 4u * (unsigned)lane;                         // first column of the lane, relative to w_lo
  float* vol = out + (long)f * D * HW + w_lo;
  const bool lane_out = w_lo + (int)l4 < W;
  float P[kTND][4][2], T[kTND][4][6];
#pragma unroll
  for (int j = 0; j < kTND; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      P[j][i][0] = P[j][i][1] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) T[j][i][k] = 0.f;
    }
  const int kOff0 = (kTDG - 1) - WAVE * kTND;                      // span slot offset of disparity j = 0
  const int kQ = (kOff0 - 1) / 4;                                  // disparity j = 1 sits one span slot below j = 0:
  constexpr int kS = KS;                                           // both come out of the same two aligned quads
  static_assert(kTND == 2 && (kTDG - 2) % 4 == 0, "two disparities per lane; (kOff0 - 1) % 4 alternates 0, 2");
  // halo sums: lane 0 takes the left quad's suffix sums, lane 63 the right quad's prefix sums, others zero
  const int halo4 = lane == 63 ? 4 : 0;
  // applied as a multiplicative mask: hipcc 7.2 miscompiles the select form `halo_lane ? hq[i] : 0.f` here
  // (it zeroes the value for every lane < 63, lane 0 included)
  const float halo_mask = (lane == 0 || lane == 63) ? 1.f : 0.f;
  auto quad = [](const float* p) { return *(const f32x4*)p; };

  wg_barrier();                                                    // chunk 0 (operands + halos) is in LDS
  if (d_base >= D) {
    // both disparities of this wavefront lie past D (last disparity group): keep the barrier protocol, skip the work
    for (int it = 0; it < n_iters * (STEP / kTRows); ++it) wg_barrier();
    return;
  }
  int chunk = 0;
  f32x4 qa, qb0, qb1;                                              // value quads of the row (frame, pattern x 2)
  for (int it = 0; it < n_iters; ++it) {
#pragma unroll
    for (int u = 0; u < STEP; ++u) {
      const int r = r_begin + it * STEP + u;
      const bool last_of_chunk = (u % kTRows) == kTRows - 1;
      // Phase A, every row: products and the vertical 3+3+3 rings of both disparities (needs only the two value
      // quads).  Phase B, output rows only (wave-uniform branch; the (bs-1) warm-up rows of a band skip it):
      // statistics quads requested first so that they arrive under the horizontal sums, then window sums,
      // normalisation and the store.
      // One per-lane base per row, made opaque: every LDS operand of the row is then base + 16-bit immediate.
      // (Likewise every address below is an opaque per-row SCALAR plus one of two loop-invariant lane registers, l4 or
      // halo4: anything the compiler can prove loop-invariant it hoists into a register of its own, and there are none
      // to spare.)
      int row_o = ((chunk % kTBufs) * kTRows + (u % kTRows)) * kTPack + 4;
      asm("" : "+s"(row_o));
      int own_o = row_o + (int)l4;
      asm("" : "+v"(own_o));
      const float* own = lds + own_o;                              // own quad after the left halo
      int pat_s = row_o + kTOffB + 4 * kQ;
      asm("" : "+s"(pat_s));
      int pat_o = pat_s + (int)l4;
      asm("" : "+v"(pat_o));
      const float* pat = lds + pat_o;                              // first of the lane's two pattern-side quads
      int hq_s = row_o - 4 + kTOffH + WAVE * (kTND * 2 * 4);
      asm("" : "+s"(hq_s));
      int hq_o = hq_s + halo4;
      asm("" : "+v"(hq_o));
      const float* hqp = lds + hq_o;                               // halo sums of (this wave, j 0) on this lane's side
      // The value quads of a chunk's first row are read here; those of its other rows were requested under phase B of
      // the previous row.
      if ((u % kTRows) == 0) {
        qa = quad(own);
        qb0 = quad(pat);
        qb1 = quad(pat + 4);
      }
      asm("" : "+v"(qa), "+v"(qb0), "+v"(qb1));
      const float av[4] = {qa[0], qa[1], qa[2], qa[3]};
      const float be[8] = {qb0[0], qb0[1], qb0[2], qb0[3], qb1[0], qb1[1], qb1[2], qb1[3]};
      const int h = r - TAIL;
      const bool row_out = (h >= h_lo) && (h < h_hi);             // wave-uniform
      float x[kTND][4];
#pragma unroll
      for (int j = 0; j < kTND; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = av[i] * be[kS + (1 - j) + i];            // b[j][i] = slot kOff0 - j + i
          const float t3 = p + P[j][i][(u + 1) % 2] + P[j][i][u % 2];
          P[j][i][u % 2] = p;
          x[j][i] = t3 + T[j][i][(u + 3) % 6] + T[j][i][u % 6];
          T[j][i][u % 6] = t3;
        }
      auto prefetch_next = [&]() {                                 // next row of the same chunk: one ring row further
        if (!last_of_chunk) {
          qa = quad(own + kTPack);
          qb0 = quad(pat + kTPack);
          qb1 = quad(pat + kTPack + 4);
        }
      };
      if (row_out) {
        f32x4 qma = quad(own + kTA), qsa = quad(own + 2 * kTA);
        f32x4 qm0 = quad(pat + kTSpanPad), qm1 = quad(pat + kTSpanPad + 4);
        f32x4 qs0 = quad(pat + 2 * kTSpanPad), qs1 = quad(pat + 2 * kTSpanPad + 4);
        prefetch_next();                                           // requested under the whole of phase B
        float me[8], se[8];
#pragma unroll
        for (int j = 0; j < kTND; ++j) {
          float pre[4], suf[4];
          pre[0] = x[j][0];
          pre[1] = pre[0] + x[j][1];
          pre[2] = pre[1] + x[j][2];
          pre[3] = pre[2] + x[j][3];
          suf[3] = x[j][3];
          suf[2] = suf[3] + x[j][2];
          suf[1] = suf[2] + x[j][1];
          suf[0] = suf[1] + x[j][0];
          float sj[4];
          window_combine4(suf, pre[3], pre, sj);                    // wave-edge lanes get 0 from the missing neighbour
          if (j == 0) {
            // statistics quads: pinned after the first window sums (data dependency keeps the wait here)
            asm("" : "+v"(qma), "+v"(qsa), "+v"(qm0), "+v"(qm1) : "v"(sj[0]), "v"(sj[3]));
            asm("" : "+v"(qs0), "+v"(qs1) : "v"(sj[0]), "v"(sj[3]));
#pragma unroll
            for (int k = 0; k < 4; ++k) { me[k] = qm0[k]; me[4 + k] = qm1[k]; se[k] = qs0[k]; se[4 + k] = qs1[k]; }
          }
          f32x4 hq = quad(hqp + j * 2 * 4);
          asm("" : "+v"(hq));
          float val[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float sh = fmaf(halo_mask, hq[i], sj[i]);
            const float cov = fmaf(qma[i], me[kS + (1 - j) + i], sh);   // qma = -bs^2 * (window mean), from the pre-pass
            val[i] = cov * ncc_inv_norm(qsa[i], se[kS + (1 - j) + i]);
          }
          const int d = d_base + j;
          if (lane_out && d < D) {
            long ooff = (long)d * HW + (long)h * W;
            asm("" : "+s"(ooff));
            float4* o = (float4*)(vol + ooff + l4);
            float4 v4 = make_float4(val[0], val[1], val[2], val[3]);
            if (ACCUM) {
              const float4 old = *o;
              v4.x += old.x; v4.y += old.y; v4.z += old.z; v4.w += old.w;
            }
            // written once, next read by another kernel after 1.8 GB more: non-temporal (-8 % on the launch)
            __builtin_nontemporal_store(f32x4{v4.x, v4.y, v4.z, v4.w}, (f32x4*)o);
          }
        }
      } else {
        prefetch_next();
      }
      if (last_of_chunk) {
        wait_lgkmcnt0();
        wg_barrier();
        ++chunk;
      }
    }
  }
}

template <bool ACCUM>
__global__ __launch_bounds__(64 * (kTWaves + 1), 4) void ncc_fast_t256_kernel(
    const float* __restrict__ ac, const float* __restrict__ m0, const float* __restrict__ v0,
    const float* __restrict__ bc, const float* __restrict__ m1, const float* __restrict__ v1, long st1_frame_stride,
    float* __restrict__ out, int C, int c, int H, int W, int D, int band_rows, int n_dgroups, int Wp, int W1, int xoff) {
  constexpr int HALF = 4, TAIL = 4, STEP = lcm_ce(6, kTRows), CPI = STEP / kTRows;   // chunks per outer iteration
  extern __shared__ float lds[];                                  // [kTBufs][kTRows][kTPack]
  // the wave index feeds scalar arithmetic (disparity base, LDS offsets): make it a scalar for the compiler
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int f = blockIdx.z / n_dgroups, dg = blockIdx.z - f * n_dgroups;
  const int w_lo = blockIdx.x * kTTile;
  const int h_lo = blockIdx.y * band_rows;
  const int h_hi = min(h_lo + band_rows, H);
  const int r_begin = h_lo - HALF, r_end = h_hi - 1 + TAIL;
  const int n_rows = r_end - r_begin + 1;
  const int n_iters = (n_rows + STEP - 1) / STEP;
  const int n_chunks = n_iters * CPI;

  if (wave == kTWaves) {
    // every chunk barrier waits for this wavefront's DMA issue and halo sums: it goes first on its SIMD (-4 %)
    __builtin_amdgcn_s_setprio(3);
    // ------------------------------ loader + halo wavefront ------------------------------
    const float* a_img = ac + ((long)f * C + c) * H * Wp + 4;      // +4: column c lives at c + 4
    const float* m0i = m0 + ((long)f * C + c) * H * Wp + 4;
    const float* v0i = v0 + ((long)f * C + c) * H * Wp + 4;
    const float* b_img = bc + (long)f * st1_frame_stride + (long)c * H * W1;
    const float* m1i = m1 + (long)f * st1_frame_stride + (long)c * H * W1;
    const float* v1i = v1 + (long)f * st1_frame_stride + (long)c * H * W1;
    const int c_lo = w_lo - 4;
    const int xb = c_lo - (dg * kTDG + kTDG - 1);                  // unclamped pattern column of span slot 0
    const int aq0 = min(c_lo + 4 * lane, Wp - 8), aq1 = min(c_lo + 256 + 4 * lane, Wp - 8);
    const int sq0 = min(xb + xoff + 4 * lane, W1 - 4), sq1 = min(xb + xoff + 256 + 4 * lane, W1 - 4);
    const bool a_tail = 256 + 4 * lane < kTA, s_tail = 256 + 4 * lane < kTSpanPad;
    auto issue_chunk = [&](int chunk) {
      float* buf = lds + (chunk % kTBufs) * (kTRows * kTPack);
#pragma unroll
      for (int s = 0; s < kTRows; ++s) {
        const int r = r_begin + chunk * kTRows + s;
        const int rc = clampi(r, 0, H - 1);
        const int hs = clampi(r - TAIL, 0, H - 1);
        float* pk = buf + s * kTPack;
        dma_quad(a_img + (long)rc * Wp + aq0, pk);
        dma_quad(m0i + (long)hs * Wp + aq0, pk + kTA);
        dma_quad(v0i + (long)hs * Wp + aq0, pk + 2 * kTA);
        dma_quad(b_img + (long)rc * W1 + sq0, pk + kTOffB);
        dma_quad(m1i + (long)hs * W1 + sq0, pk + kTOffB + kTSpanPad);
        dma_quad(v1i + (long)hs * W1 + sq0, pk + kTOffB + 2 * kTSpanPad);
        if (a_tail) {
          dma_quad(a_img + (long)rc * Wp + aq1, pk + 256);
          dma_quad(m0i + (long)hs * Wp + aq1, pk + kTA + 256);
          dma_quad(v0i + (long)hs * Wp + aq1, pk + 2 * kTA + 256);
        }
        if (s_tail) {
          dma_quad(b_img + (long)rc * W1 + sq1, pk + kTOffB + 256);
          dma_quad(m1i + (long)hs * W1 + sq1, pk + kTOffB + kTSpanPad + 256);
          dma_quad(v1i + (long)hs * W1 + sq1, pk + kTOffB + 2 * kTSpanPad + 256);
        }
      }
    };
    // halo job of this lane: consumer wave cw, disparity j, side (0 = quad left of the tile, 1 = right of it)
    const int cw = lane >> 2, hj = (lane >> 1) & 1, side = lane & 1;
    const bool has_job = lane < 4 * kTWaves;
    const int a_slot = side ? (kTA - 4) : 0;
    const int b_slot = a_slot + (kTDG - 1) - (cw * kTND + hj);
    float hP[4][2], hT[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hP[i][0] = hP[i][1] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) hT[i][k] = 0.f;
    }
    // vertical ring update of the halo quad for the rows of one chunk; UB = ring phase of its first row
    auto halo_chunk = [&](int chunk, auto ub_tag) {
      constexpr int UB = decltype(ub_tag)::value;
      const float* buf = lds + (chunk % kTBufs) * (kTRows * kTPack);
#pragma unroll
      for (int s = 0; s < kTRows; ++s) {
        const int u = (UB + s) % 6;
        const float* pk = buf + s * kTPack;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float p = has_job ? pk[a_slot + i] * pk[kTOffB + b_slot + i] : 0.f;
          const float t3 = p + hP[i][(u + 1) % 2] + hP[i][u % 2];
          hP[i][u % 2] = p;
          x[i] = t3 + hT[i][(u + 3) % 6] + hT[i][u % 6];
          hT[i][u % 6] = t3;
        }
        float o4[4];
        if (side) {                                                // prefix sums: columns 0..i of the right quad
          o4[0] = x[0]; o4[1] = o4[0] + x[1]; o4[2] = o4[1] + x[2]; o4[3] = o4[2] + x[3];
        } else {                                                   // suffix sums: columns i..3 of the left quad
          o4[3] = x[3]; o4[2] = o4[3] + x[2]; o4[1] = o4[2] + x[1]; o4[0] = o4[1] + x[0];
        }
        if (has_job) {
          float* hq = const_cast<float*>(pk) + kTOffH + lane * 4;  // lane == ((cw*kTND + hj)*2 + side)
          hq[0] = o4[0]; hq[1] = o4[1]; hq[2] = o4[2]; hq[3] = o4[3];
        }
      }
    };
    constexpr int L = kTRows * kTDmaPerRow;
    static_assert(L * (kTBufs - 2) < 64, "in-flight DMA count must fit vmcnt");
#pragma unroll
    for (int k = 0; k < kTBufs - 1; ++k)
      if (k < n_chunks) issue_chunk(k);
    if (n_chunks >= kTBufs - 1) wait_vmcnt<L*(kTBufs - 2)>(); else wait_vmcnt<0>();
    halo_chunk(0, std::integral_constant<int, 0>{});
    wait_lgkmcnt0();
    wg_barrier();
    for (int it = 0; it < n_iters; ++it) {
#pragma unroll
      for (int cc = 0; cc < CPI; ++cc) {
        const int ch = it * CPI + cc;
        const int nxt = ch + kTBufs - 1;
        if (nxt < n_chunks) {
          issue_chunk(nxt);
          wait_vmcnt<L*(kTBufs - 2)>();                       // chunk ch+1 has landed
        } else {
          wait_vmcnt<0>();
        }
        if (ch + 1 < n_chunks) {
          if (cc == 0) halo_chunk(ch + 1, std::integral_constant<int, (1 * kTRows) % 6>{});
          else if (cc == 1) halo_chunk(ch + 1, std::integral_constant<int, (2 * kTRows) % 6>{});
          else if (cc == 2) halo_chunk(ch + 1, std::integral_constant<int, (3 * kTRows) % 6>{});
          else if (cc == 3) halo_chunk(ch + 1, std::integral_constant<int, (4 * kTRows) % 6>{});
          else if (cc == 4) halo_chunk(ch + 1, std::integral_constant<int, (5 * kTRows) % 6>{});
          else halo_chunk(ch + 1, std::integral_constant<int, (6 * kTRows) % 6>{});
        }
        wait_lgkmcnt0();
        wg_barrier();
      }
    }
    return;
  }

  // two copies of the consumer loop: the sub-quad shift of the pattern-side operands alternates with the wave index
  if (wave & 1)
    t256_consume<ACCUM, (kTDG - 2 - kTND) % 4>(lds, out, wave, f, dg, lane, w_lo, h_lo, h_hi, r_begin, n_iters, H, W, D);
  else
    t256_consume<ACCUM, (kTDG - 2) % 4>(lds, out, wave, f, dg, lane, w_lo, h_lo, h_hi, r_begin, n_iters, H, W, D);
}

// ------------------------------------------------------------------------------------
// ALL-D kernel (bs == 9, W % 4 == 0, C == 1): volume + in-kernel ranking, what ctd_xcorrvol_argmax_f32 launches.
//
// Same consumer pipeline as the tile-256 kernel, but ONE workgroup owns a (256-column tile, band of rows, frame) for
// EVERY disparity: 15 consumer wavefronts + 1 loader (1024 threads, one workgroup per CU = four wavefronts on every
// SIMD), 2 disparities per lane, so a pass over the band covers up to 30 disparities and the workgroup makes
// ceil(D / 30) passes (dealt evenly: D = 128 -> 5 passes of 26 on 13 wavefronts, see alld_plan).  What that buys:
//   * the ranking state lives in LDS for the whole band -- two u32 slots {top, runner-up} per pixel, fed by LDS
//     atomics from all consumer wavefronts across all passes -- and what leaves the kernel is the final index (int64),
//     the best score, the work-list flag: no per-group partial planes (141 MB at config 2) and no merge kernel;
//   * the frame-side operands of the band are re-read by the SAME workgroup on every pass (the same CU, the same L2)
//     instead of by ten workgroups scattered over the eight XCDs' L2s;
//   * half the passes for the loader's DMA and halo work, 15 of 16 wavefronts computing instead of 7 of 8;
//   * at most 256 workgroups are resident: the store-only ceiling of exactly this pattern is 6.0-6.7 TB/s against
//     5.8-6.0 for the per-group grid (profiles/round3_store_ceiling.txt).
// KEY of a score: t = 6 + score lies in [4, 8) for every |score| <= 1 + 1e-5, where consecutive floats are 2^-21 apart
// and ordered like their bit patterns: key = (bits(t) << 9) | (511 - d) is an unsigned integer ordered by score first
// (absolute resolution 2^-21 = 4.8e-7, against 2^-19 RELATIVE for the mantissa-tag keys it replaces) and by LOWER
// disparity second -- so u32 maxima carry the argmax with first-index-wins ties.  Per pixel:
//     old = ds_max_rtn_u32(top, hi);  ds_max_u32(second, med3(old, hi, lo))      (hi >= lo: the lane's two keys)
// -- every key that is not the final maximum is, at some point, the loser of such an exchange, so `second` ends up as
// the runner-up; the second atomic rides on the next row (its operand is the first one's return value).
// Scores of LISTED windows: the pre-pass stores a zero reciprocal deviation for them, so the kernels produce the
// placeholder score 0 for exactly the outputs the fix-up pass recomputes.  A placeholder can only matter for the
// ranking if it comes out on top or within the margin of the top: ncc_fixup_kernel sends a pixel to the exact
// re-scoring when the exact score is a contender OR when the pixel's index is the placeholder's disparity.  Scores past
// the start of the fully clamped run (ext.h:152-154 makes them copies of its first element) get key 0.
// ------------------------------------------------------------------------------------
constexpr int kAWaves = 15;                    // consumer wavefronts per workgroup
constexpr int kADGMax = kAWaves * 2;           // disparities per pass, at most
constexpr int kAA = 256 + 8;                   // frame-side array: 4 halo columns either side
constexpr int kASpanPad = (kAA + kADGMax - 1 + 1 + 3) / 4 * 4;   // 296: multiple of 4, > span
static_assert(kAA + kADGMax - 1 < kASpanPad, "pattern span must fit its padded array");
constexpr int kAHalo = kAWaves * 2 * 2 * 4;    // [wave][j][side][4] halo sums
constexpr int kAPack = 3 * kAA + 3 * kASpanPad + kAHalo;   // 1920 floats per staged row
constexpr int kABufs = 3;                     // LDS chunks in the ring; a chunk is ROWS = 3 or 2 staged rows (template parameter)
constexpr int kAOffB = 3 * kAA, kAOffH = 3 * kAA + 3 * kASpanPad;
// band height limit: rank slots (2 KB per row) + staging ring <= 160 KB -- 44 rows with 3-row chunks, 57 with 2-row chunks
constexpr int alld_max_band_rows(int rows) { return (160 * 1024 - (int)sizeof(float) * kABufs * rows * kAPack) / 2048; }
constexpr int kAStore = 1, kARank = 2;          // MODE bits of the all-D kernel: materialise the volume / rank the scores
// Block SAD / MSE cost volume through the same pipeline (with kAStore, never with kARank): the per-pixel
// plane |P[r][c - d] - I[r][c]| (squared for MSE) takes the place of the product a * b, and its 9 x 9 window sum / 81 is the
// output -- no statistics rows, no normalisation.  See costvol_sep_f32.
constexpr int kASad = 4, kAMse = 8;
constexpr int kAllowTwoRowChunks = 2;           // 3: never use 2-row chunks
constexpr double kTwoRowPenalty = 1.03;
constexpr int kTagBits = 9;                    // D <= 512
constexpr unsigned kTagMask = (1u << kTagBits) - 1u;
constexpr float kKeyBias = 6.f;
static_assert(alld_max_band_rows(3) == 46 && alld_max_band_rows(2) == 57, "LDS budget");

__device__ inline unsigned umed3(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// fixed-point units of the keys per unit of score, and the re-ranking margin in those units (ctd_rank.h: rank_margin)
__device__ inline unsigned key_margin_units(float eps) { return (unsigned)ceilf(rank_margin(eps, 1.f) * 2097152.f); }

// Diagnostic build only (-DCTD_STAMPS, tools/build_variant.sh): every wavefront of the all-D kernel notes the shader
// clock (s_memtime) when it ARRIVES at each chunk barrier of pass CTD_STAMP_PASS and when it LEAVES it, into LDS behind
// the staging ring; the workgroup dumps them at its end ([workgroup][kStampWords]: 8 header words, then
// [wave][chunk][arrive | leave]).  tools/alld_timeline.py reads them through ctd_debug_read_stamps.  No stamp executes
// in the product build.
#ifdef CTD_STAMPS
#ifndef CTD_STAMP_PASS
#define CTD_STAMP_PASS 2
#endif
constexpr int kStampChunks = 34, kStampWords = 8 + 16 * kStampChunks * 2, kStampWgs = 1024;
__device__ unsigned g_stamps[kStampWgs * kStampWords];
__device__ inline unsigned stamp_now() { return (unsigned)__builtin_amdgcn_s_memtime(); }
// (no scalar of its own: the consumer loops are at the limit of the scalar registers -- hipcc 7.2 dies with "illegal VGPR
// to SGPR copy" when their spilling fails -- so the wavefront number comes from threadIdx and the stamp area's address is
// an immediate offset from the ring's base)
__device__ inline void stamp_put(unsigned* st, int pass, int chunk, int which) {
  const unsigned t = stamp_now();
  const int idx = 8 + ((int)(threadIdx.x >> 6) * kStampChunks + chunk) * 2 + which;
  if ((threadIdx.x & 63) == 0 && pass == CTD_STAMP_PASS && chunk < kStampChunks) st[idx] = t;
}
#define CTD_STAMP_ARRIVE(st, wave, pass, chunk, lane) stamp_put(st, pass, chunk, 0)
#define CTD_STAMP_LEAVE(st, wave, pass, chunk, lane) stamp_put(st, pass, chunk, 1)
#else
#define CTD_STAMP_ARRIVE(st, wave, pass, chunk, lane) do {} while (0)
#define CTD_STAMP_LEAVE(st, wave, pass, chunk, lane) do {} while (0)
#endif

// JM: which of the pair's two disparities this wavefront works on -- 3 = both (the regular consumer), 1 = j 0 only,
// 2 = j 1 only: the two halves of a SPLIT pair, run by two wavefronts on different SIMDs (see the kernel: with 13 pairs
// per pass the thirteenth pair would otherwise put a fourth full consumer on one SIMD and the chunk barrier makes
// everybody wait for that SIMD).  WAVE is the PAIR index (span slots, halo slots, disparity base); `wave_id` the
// wavefront's own number (diagnostic stamps only).
template <int MODE, int KS, int ROWS, int JM = 3>
__device__ __forceinline__ void alld_consume(float* lds, unsigned* rank_lds, float* __restrict__ out, int WAVE, int f,
                                             int lane, int w_lo, int h_lo, int h_hi, int r_begin, int n_iters,
                                             int n_pass, int rot, int dgs, int H, int W, int D, int wave_id) {
  constexpr int J0 = (JM & 1) ? 0 : 1;                             // first active j
  constexpr int TAIL = 4, STEP = 6, CPI = STEP / ROWS;            // block size 9
  constexpr bool STORE = (MODE & kAStore) != 0, RANK = (MODE & kARank) != 0;
  static_assert(STEP % ROWS == 0, "a chunk never straddles two outer iterations");
  const long HW = (long)H * W;
  const unsigned l4 = 4u * (unsigned)lane;                         // first column of the lane, relative to w_lo
  float* vol = out + (long)f * D * HW + w_lo;
  const bool lane_out = w_lo + (int)l4 < W;
  float P[2][4][2], T[2][4][6];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      P[j][i][0] = P[j][i][1] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) T[j][i][k] = 0.f;
    }
  const int kOff0 = (dgs - 1) - WAVE * 2;                          // span slot offset of disparity j = 0
  const int kQ = (kOff0 - 1) / 4;                                  // disparity j = 1 sits one span slot below j = 0:
  constexpr int kS = KS;                                           // both come out of the same two aligned quads
  const int halo4 = lane == 63 ? 4 : 0;
  const float halo_mask = (lane == 0 || lane == 63) ? 1.f : 0.f;  // (multiplicative: see t256_consume)
  auto quad = [](const float* p) { return *(const f32x4*)p; };

  // The runner-up update needs the value the first atomic returns: it is issued at the start of the next row, behind
  // that row's operand reads -- LDS answers in order, so the returns are there by the time the operands are.
  // Unconditional (a row without outputs leaves key 0 here, a no-op for the maximum): a flag would keep these twelve
  // registers alive across the whole loop.
  unsigned rk_hi[4], rk_lo[4], rk_old[4];
  unsigned* rk_sl = rank_lds + lane;
#define st_lds ((unsigned*)(lds + kABufs * ROWS * kAPack))
#pragma unroll
  for (int i = 0; i < 4; ++i) rk_hi[i] = rk_lo[i] = rk_old[i] = 0u;
  auto rank_second = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      (void)__hip_atomic_fetch_max(rk_sl + 256 + 64 * i, umed3(rk_old[i], rk_hi[i], rk_lo[i]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  wg_barrier();                                                    // chunk 0 (operands + halos) and the cleared slots are in LDS
  int slot = 0;                                                    // ring slot of the current chunk
  f32x4 qa, qb0, qb1;                                              // value quads of the row (frame, pattern x 2)
  for (int pass = 0; pass < n_pass; ++pass) {
    const int grp = pass + rot;                                    // (rot: the workgroup's first disparity group)
    const int d_base = grp * dgs + WAVE * 2;
    if (WAVE * 2 >= dgs || d_base + J0 >= D) {
      // a wavefront without disparities in this pass (the pass is narrower than 15 pairs, or it is the last pass and
      // both disparities lie past D): keep the barrier protocol, skip the work
      for (int k = 0; k < n_iters * CPI; ++k) {
        CTD_STAMP_ARRIVE(st_lds, wave_id, pass, k, lane);
        wg_barrier();
        CTD_STAMP_LEAVE(st_lds, wave_id, pass, k, lane);
        slot = slot == kABufs - 1 ? 0 : slot + 1;
      }
      continue;
    }
    const unsigned tag0 = kTagMask - (unsigned)d_base;             // key tag of disparity j = 0 (j = 1: one less)
    // only the first column tile can reach the fully clamped run (d > w + TAIL needs d_base + 1 > w_lo + TAIL)
    const bool run_masks = d_base + 1 > w_lo + TAIL;
    for (int it = 0; it < n_iters; ++it) {
#pragma unroll
      for (int u = 0; u < STEP; ++u) {
        const int r = r_begin + it * STEP + u;
        const bool last_of_chunk = (u % ROWS) == ROWS - 1;
        // (addressing: one opaque per-row scalar plus one of two loop-invariant lane registers, see t256_consume)
        int row_o = (slot * ROWS + (u % ROWS)) * kAPack + 4;
        asm("" : "+s"(row_o));
        int own_o = row_o + (int)l4;
        asm("" : "+v"(own_o));
        const float* own = lds + own_o;                            // own quad after the left halo
        int pat_s = row_o + kAOffB + 4 * kQ;
        asm("" : "+s"(pat_s));
        int pat_o = pat_s + (int)l4;
        asm("" : "+v"(pat_o));
        const float* pat = lds + pat_o;                            // first of the lane's two pattern-side quads
        int hq_s = row_o - 4 + kAOffH + WAVE * (2 * 2 * 4);
        asm("" : "+s"(hq_s));
        int hq_o = hq_s + halo4;
        asm("" : "+v"(hq_o));
        const float* hqp = lds + hq_o;                             // halo sums of (this wave, j 0) on this lane's side
        // The value quads of a chunk's first row are read here; those of its other rows were requested at the end of
        // the previous row, AHEAD of that row's returning atomics: LDS answers in order, and a read queued behind the
        // atomics would make phase A wait for their round trip.
        if ((u % ROWS) == 0) {
          qa = quad(own);
          qb0 = quad(pat);
          qb1 = quad(pat + 4);
        }
        asm("" : "+v"(qa), "+v"(qb0), "+v"(qb1));
        const float av[4] = {qa[0], qa[1], qa[2], qa[3]};
        const float be[8] = {qb0[0], qb0[1], qb0[2], qb0[3], qb1[0], qb1[1], qb1[2], qb1[3]};
        const int h = r - TAIL;
        const bool row_out = (h >= h_lo) && (h < h_hi);           // wave-uniform
        // 9-row sums as 3 x 3: t3 = rows r-2 .. r of the products, x = t3 of rows r, r-3, r-6.  The two OLD ring entries
        // are added first and die there, so the new entry (p, t3) can take the register of the one it replaces: summed
        // as (p + P') + P the new and the old value were live together and every ring slot cost a v_mov at the loop's
        // back edge (48 of 979 vector instructions per 6 rows).
        float x[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (!(JM & (1 << j))) continue;
            float sP = P[j][i][(u + 1) % 2] + P[j][i][u % 2];
            float sT = T[j][i][(u + 3) % 6] + T[j][i][u % 6];
            asm("" : "+v"(sP), "+v"(sT));                          // (formed before the slots are reused)
            float p;                                               // b[j][i] = slot kOff0 - j + i
            if constexpr ((MODE & kASad) != 0) {
              p = fabsf(av[i] - be[kS + (1 - j) + i]);
            } else if constexpr ((MODE & kAMse) != 0) {
              const float df = av[i] - be[kS + (1 - j) + i];
              p = df * df;
            } else {
              p = av[i] * be[kS + (1 - j) + i];
            }
            const float t3 = p + sP;
            P[j][i][u % 2] = p;
            x[j][i] = t3 + sT;
            T[j][i][u % 6] = t3;
          }
        // previous output row's runner-up update (possibly of the previous chunk): its returns are in by now
        if constexpr (RANK) rank_second();
        auto prefetch_next = [&]() {                               // next row of the same chunk: one ring row further
          if (!last_of_chunk) {
            qa = quad(own + kAPack);
            qb0 = quad(pat + kAPack);
            qb1 = quad(pat + kAPack + 4);
          }
        };
        if (row_out) {
          constexpr bool COST = (MODE & (kASad | kAMse)) != 0;      // SAD / MSE cost volume: no statistics, no normalisation
          f32x4 qma, qsa, qm0, qm1, qs0, qs1;
          if constexpr (!COST) {
            qma = quad(own + kAA), qsa = quad(own + 2 * kAA);
            qm0 = quad(pat + kASpanPad), qm1 = quad(pat + kASpanPad + 4);
            qs0 = quad(pat + 2 * kASpanPad), qs1 = quad(pat + 2 * kASpanPad + 4);
          }
          float me[8], se[8];
          unsigned key[2][4];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (!(JM & (1 << j))) continue;
            float pre[4], suf[4];
            pre[0] = x[j][0];
            pre[1] = pre[0] + x[j][1];
            pre[2] = pre[1] + x[j][2];
            pre[3] = pre[2] + x[j][3];
            suf[3] = x[j][3];
            suf[2] = suf[3] + x[j][2];
            suf[1] = suf[2] + x[j][1];
            suf[0] = suf[1] + x[j][0];
            float sj[4];
            window_combine4(suf, pre[3], pre, sj);                  // wave-edge lanes get 0 from the missing neighbour
            if constexpr (!COST) {
              if (j == J0) {
                // statistics quads: pinned after the first window sums (data dependency keeps the wait here)
                asm("" : "+v"(qma), "+v"(qsa), "+v"(qm0), "+v"(qm1) : "v"(sj[0]), "v"(sj[3]));
                asm("" : "+v"(qs0), "+v"(qs1) : "v"(sj[0]), "v"(sj[3]));
#pragma unroll
                for (int k = 0; k < 4; ++k) { me[k] = qm0[k]; me[4 + k] = qm1[k]; se[k] = qs0[k]; se[4 + k] = qs1[k]; }
              }
            }
            f32x4 hq = quad(hqp + j * 2 * 4);
            asm("" : "+v"(hq));
            const int d = d_base + j;
            float val[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float sh = fmaf(halo_mask, hq[i], sj[i]);
              if constexpr (COST) {
                val[i] = sh * (1.f / 81.f);                         // mean over the 9 x 9 block
              } else {
                const float cov = fmaf(qma[i], me[kS + (1 - j) + i], sh);   // qma = -bs^2 * (window mean), from the pre-pass
                const float inv = ncc_inv_norm(qsa[i], se[kS + (1 - j) + i]);
                if (STORE) val[i] = cov * inv;                      // the same bits as the plain volume kernels'
                if constexpr (RANK) key[j][i] = (__float_as_uint(fmaf(cov, inv, kKeyBias)) << kTagBits) | (tag0 - (unsigned)j);
              }
            }
            if (STORE && lane_out && d < D) {
              long ooff = (long)d * HW + (long)h * W;
#if defined(CTD_STORE_AB) && CTD_STORE_AB == 1
              // TIMING EXPERIMENT ONLY (wrong layout): a permutation of the volume's 1-KB pieces -- workgroup-major, then
              // row, then disparity -- so that the 26 pieces a workgroup writes per row step are one contiguous run
              ooff = ((((long)blockIdx.x * (h_hi - h_lo) + (h - h_lo)) * D + d) << 8) - ((long)f * D * HW + w_lo);
#elif defined(CTD_STORE_AB) && CTD_STORE_AB == 2
              // TIMING EXPERIMENT ONLY: chip-linear -- everything the 256 workgroups write in one row step is one window
              ooff = ((((((long)pass * (h_hi - h_lo) + (h - h_lo)) * gridDim.x + blockIdx.x) * dgs + (d - grp * dgs)) %
                       ((long)gridDim.x * (h_hi - h_lo) * D)) << 8) - ((long)f * D * HW + w_lo);
#elif defined(CTD_STORE_AB) && CTD_STORE_AB == 4
              // TIMING EXPERIMENT ONLY: every store lands in one 4 MB window (stays in L2 / the Infinity Cache): the
              // instruction stream and the CU's store path as in the real kernel, no HBM write behind it
              ooff = (ooff + (long)f * D * HW + w_lo) % (1L << 20) - ((long)f * D * HW + w_lo);
#endif
              asm("" : "+s"(ooff));
              // written once, next read by another kernel after 1.8 GB more: non-temporal
#if defined(CTD_STORE_AB) && CTD_STORE_AB == 3
              *(f32x4*)(vol + ooff + l4) = f32x4{val[0], val[1], val[2], val[3]};     // TIMING EXPERIMENT: plain stores
#else
              __builtin_nontemporal_store(f32x4{val[0], val[1], val[2], val[3]}, (f32x4*)(vol + ooff + l4));
#endif
            }
            if constexpr (RANK) {
              if (d < D) {                                         // wave-uniform branch (a select would be 4 VALU slots)
                if (run_masks) {                                   // wave-uniform: first column tile only
                  // d > w + TAIL: copy of the run's first element.  Loop-invariant compare: hoisted into a lane mask.
#pragma unroll
                  for (int i = 0; i < 4; ++i)
                    if (d - TAIL - i - w_lo > (int)l4) key[j][i] = 0u;
                }
              } else {
                // (volatile: as plain assignments the compiler runs these four moves on EVERY row, ahead of the branch)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, 0" : "=v"(key[j][i]));
              }
            }
          }
          prefetch_next();
          if constexpr (RANK) {
            // slots of output row h: [row][top | second][column-in-quad][lane]
            rk_sl = rank_lds + (it * STEP + u - 2 * TAIL) * 512 + lane;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (JM == 3) {
                rk_hi[i] = max(key[0][i], key[1][i]);
                rk_lo[i] = min(key[0][i], key[1][i]);
              } else {
                rk_hi[i] = key[J0][i];                             // one key per pixel and row: the loser of the exchange
                rk_lo[i] = 0u;                                     // with the slot's top is min(old, key) = med3(old, key, 0)
              }
              rk_old[i] = __hip_atomic_fetch_max(rk_sl + 64 * i, rk_hi[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        } else {
          prefetch_next();
          if constexpr (RANK) {
            // only the returned old top is reset: the next rank_second() then offers min(hi, lo) of the last output row
            // once more -- a genuine non-top key of that pixel, harmless
#pragma unroll
            for (int i = 0; i < 4; ++i) rk_old[i] = 0u;
          }
        }
        if (last_of_chunk) {
          // the pass's last chunk has no next row to ride on: complete its slots before its barrier (wave-uniform)
          if constexpr (RANK) {
            if (it == n_iters - 1 && u == STEP - 1) {
              rank_second();
#pragma unroll
              for (int i = 0; i < 4; ++i) rk_hi[i] = rk_lo[i] = rk_old[i] = 0u;
            }
          }
          wait_lgkmcnt0();
          // (not in the volume + ranking instantiation: with the stamps' scalars on top hipcc 7.2 fails to spill its
          // scalar registers -- "illegal VGPR to SGPR copy"; the timeline is taken in the two other modes)
          if constexpr (MODE != (kAStore | kARank)) CTD_STAMP_ARRIVE(st_lds, wave_id, pass, it * CPI + u / ROWS, lane);
          wg_barrier();
          if constexpr (MODE != (kAStore | kARank)) CTD_STAMP_LEAVE(st_lds, wave_id, pass, it * CPI + u / ROWS, lane);
          slot = slot == kABufs - 1 ? 0 : slot + 1;
        }
      }
    }
  }
}

#undef st_lds

// What the statistics DMAs of a chunk need (either loader).
struct AlldStatsArgs {
  const float *m0i, *v0i, *m1i, *v1i;       // this frame's mean / reciprocal-deviation planes (frame side: column c at c + 4)
  float* lds;                               // staging ring
  int Wp, W1, xoff, c_lo, r_begin, h_lo, h_hi, n_pass, n_chunks, dgs, first_grp;
};
constexpr int kAStatsPerRow = 8, kAValuesPerRow = 4;   // dwordx4 LDS-DMA instructions per staged row

// statistics rows of chunk (pass ip, chunk ic) into ring slot `sl`
template <int ROWS>
__device__ __forceinline__ void alld_issue_stats(const AlldStatsArgs& a, int ip, int ic, int sl, int lane) {
  constexpr int TAIL = 4;
  const int aq0 = min(a.c_lo + 4 * lane, a.Wp - 8), aq1 = min(a.c_lo + 256 + 4 * lane, a.Wp - 8);
  const bool a_tail = 256 + 4 * lane < kAA, s_tail = 256 + 4 * lane < kASpanPad;
  float* buf = a.lds + sl * (ROWS * kAPack);
  const int xb = a.c_lo - ((ip + a.first_grp) * a.dgs + a.dgs - 1);   // unclamped pattern column of span slot 0
  const int sq0 = min(xb + a.xoff + 4 * lane, a.W1 - 4), sq1 = min(xb + a.xoff + 256 + 4 * lane, a.W1 - 4);
#pragma unroll
  for (int s2 = 0; s2 < ROWS; ++s2) {
    const int r = a.r_begin + ic * ROWS + s2;
    // statistics of output row r - TAIL (rows of a mixed chunk that complete no output re-read a row the band needs anyway)
    const int hs = clampi(r - TAIL, a.h_lo, a.h_hi - 1);
    float* pk = buf + s2 * kAPack;
    dma_quad(a.m0i + (long)hs * a.Wp + aq0, pk + kAA);
    dma_quad(a.v0i + (long)hs * a.Wp + aq0, pk + 2 * kAA);
    dma_quad(a.m1i + (long)hs * a.W1 + sq0, pk + kAOffB + kASpanPad);
    dma_quad(a.v1i + (long)hs * a.W1 + sq0, pk + kAOffB + 2 * kASpanPad);
    if (a_tail) {
      dma_quad(a.m0i + (long)hs * a.Wp + aq1, pk + kAA + 256);
      dma_quad(a.v0i + (long)hs * a.Wp + aq1, pk + 2 * kAA + 256);
    }
    if (s_tail) {
      dma_quad(a.m1i + (long)hs * a.W1 + sq1, pk + kAOffB + kASpanPad + 256);
      dma_quad(a.v1i + (long)hs * a.W1 + sq1, pk + kAOffB + 2 * kASpanPad + 256);
    }
  }
}

// a chunk is LIGHT when none of its rows completes an output row of the band (the 8 warm-up rows of a pass and the padding
// behind the last output row): nobody reads statistics there, none are staged
template <int ROWS>
__device__ __forceinline__ bool alld_chunk_is_light(int ch, int n_out_rows) {
  return ch * ROWS + ROWS - 1 < 8 || ch * ROWS >= 8 + n_out_rows;
}

// The second loader (a spare consumer wavefront, see the roles in the kernel): the statistics rows, two chunks ahead,
// in step with the chunk barriers; before each barrier everything but the newest chunk has landed.
template <int ROWS>
__device__ __forceinline__ void alld_stats_loader(const AlldStatsArgs& a, int lane) {
  constexpr int LS = ROWS * kAStatsPerRow;
  static_assert(LS < 64, "in-flight DMA count must fit vmcnt");
  __builtin_amdgcn_s_setprio(3);
  const int total = a.n_pass * a.n_chunks, n_out_rows = a.h_hi - a.h_lo;
  int i_slot = 0, i_pass = 0, i_ch = 0, i_n = 0;
  auto issue_next = [&]() {                                        // returns whether anything was issued
    const bool light = alld_chunk_is_light<ROWS>(i_ch, n_out_rows);
    if (!light) alld_issue_stats<ROWS>(a, i_pass, i_ch, i_slot, lane);
    ++i_n;
    i_slot = i_slot == kABufs - 1 ? 0 : i_slot + 1;
    if (++i_ch == a.n_chunks) { i_ch = 0; ++i_pass; }
    return !light;
  };
  issue_next();
  if (issue_next()) wait_vmcnt<LS>(); else wait_vmcnt<0>();        // chunk 0 has landed
  wg_barrier();
  int c_pass = 0, c_ch = 0;                                        // the chunk the consumers are working on (stamps)
  (void)c_pass;
  for (int g = 0; g < total; ++g) {
    if (i_n < total) {
      if (issue_next()) wait_vmcnt<LS>(); else wait_vmcnt<0>();    // chunk g + 1 has landed
    } else {
      wait_vmcnt<0>();
    }
#ifdef CTD_STAMPS
    unsigned* st_l = (unsigned*)(a.lds + kABufs * ROWS * kAPack);
#endif
    CTD_STAMP_ARRIVE(st_l, kAWaves - 1, c_pass, c_ch, lane);
    wg_barrier();
    CTD_STAMP_LEAVE(st_l, kAWaves - 1, c_pass, c_ch, lane);
    if (++c_ch == a.n_chunks) { c_ch = 0; ++c_pass; }
  }
}

// MODE_STORE: the volume is materialised as well; otherwise nothing but indices / best scores / work list leave.
template <int MODE, int ROWS>
__global__ __launch_bounds__(64 * (kAWaves + 1)) void ncc_fast_alld_kernel(
    const float* __restrict__ ac, const float* __restrict__ m0, const float* __restrict__ v0,
    const float* __restrict__ bc, const float* __restrict__ m1, const float* __restrict__ v1, long st1_frame_stride,
    float* __restrict__ out, int64_t* __restrict__ idx_out, float* __restrict__ best_out,
    unsigned char* __restrict__ flags_out, WorkList work, float rank_eps, int frames, int n_items, int H, int W, int D,
    int band_rows, int n_pass_all, int dgs, int Wp, int W1, int xoff, int n_psplit) {
  constexpr int HALF = 4, TAIL = 4, STEP = 6, CPI = STEP / ROWS;
  // [band_rows][top | second][256] rank slots first (their row base goes into one lane register), then the staging ring
  extern __shared__ float lds_all[];
  constexpr bool RANK = (MODE & kARank) != 0;
  unsigned* rank_lds = (unsigned*)lds_all;
  float* lds = lds_all + (RANK ? band_rows * 512 : 0);             // [kABufs][ROWS][kAPack]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Work item of this workgroup: column tile fastest, then band, then frame.  (XCD-aware orders -- every XCD a
  // contiguous range of the band-major list, so that co-resident workgroups share pattern rows in its L2 -- cut the
  // operand fetches from 2 x 99 to 2 x 59-68 MiB and were 1-2.5 % SLOWER in four A/B runs: the kernel is bound by vector
  // issue, not by its 5-10 % of operand traffic.  Pass order rotated per workgroup: slower as well.)
  // Calls with few frames (one 1024 x 1024 frame is 4 column tiles) would need short bands to fill the chip, and every
  // band pays 8 warm-up rows per pass: without a ranking the DISPARITY GROUPS of an item can go to `n_psplit` different
  // workgroups instead (pass split fastest in the item number), each walking its share of the passes over a tall band.
  int item = (int)blockIdx.x;
  if (item >= n_items) return;                                     // whole workgroup, before any barrier
  const int ps = item % n_psplit;
  item /= n_psplit;
  const int ppg = (n_pass_all + n_psplit - 1) / n_psplit;          // passes per workgroup
  const int rot = ps * ppg;                                        // first disparity group of this workgroup
  const int n_pass = min(ppg, n_pass_all - rot);
  if (n_pass <= 0) return;
  const int n_tiles = (W + 255) / 256;
  const int n_bands = n_items / (frames * n_tiles * n_psplit);
  const int w_lo = (item % n_tiles) * 256;
  const int band = (item / n_tiles) % n_bands;
  const int f = item / (n_tiles * n_bands);
  const int h_lo = band * band_rows;
  const int h_hi = min(h_lo + band_rows, H);
  const int r_begin = h_lo - HALF, r_end = h_hi - 1 + TAIL;
  const int n_rows = r_end - r_begin + 1;
  const int n_iters = (n_rows + STEP - 1) / STEP;
  const int n_chunks = n_iters * CPI;                              // per pass; a multiple of CPI
  const int n_act = dgs / 2;                                       // consumer wavefronts with work
#ifdef CTD_STAMPS
  unsigned* const st_lds = (unsigned*)(lds + kABufs * ROWS * kAPack);
  for (int k = threadIdx.x; k < kStampWords; k += 64 * (kAWaves + 1)) st_lds[k] = 0u;
  if (threadIdx.x == 0) {
    st_lds[0] = stamp_now();
    st_lds[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
    st_lds[2] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, bits 0..3
    st_lds[3] = (unsigned)n_chunks;
  }
  auto dump_stamps = [&]() {
    wait_lgkmcnt0();
    wg_barrier();
    if (threadIdx.x == 0) {
      st_lds[4] = stamp_now();
      st_lds[5] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
    wait_lgkmcnt0();
    wg_barrier();
    if ((int)blockIdx.x < kStampWgs)
      for (int k = threadIdx.x; k < kStampWords; k += 64 * (kAWaves + 1)) g_stamps[blockIdx.x * kStampWords + k] = st_lds[k];
  };
#endif

  // every wavefront clears its share of the rank slots (key 0 = below every score)
  if constexpr (RANK)
    for (int k = threadIdx.x; k < band_rows * 128; k += 64 * (kAWaves + 1)) ((uint4*)rank_lds)[k] = make_uint4(0u, 0u, 0u, 0u);
  wait_lgkmcnt0();                                                 // (the consumers' first barrier is a raw s_barrier)

  // ---- roles.  Pairs of disparities per pass: n_act = dgs / 2 (13 for D = 128).  The chunk barrier makes every
  // wavefront wait for the slowest SIMD, and wavefront w sits on SIMD w % 4 (the timeline of the -DCTD_STAMPS build:
  // profiles/round4_alld_timeline.txt), so the roles are dealt to even out the four SIMDs:
  //   * wavefront 15: loader of the VALUE rows (frame and pattern samples) + the halo sums;
  //   * with a spare wavefront (n_act <= 14), wavefront 14 is a second loader for the STATISTICS rows (mean / reciprocal
  //     deviation planes, needed by output rows only): the halo sums need the values alone, so nobody waits for it
  //     but the barrier, and the first loader's chunk drops from 24 DMA instructions to 8;
  //   * with two spare wavefronts (n_act == 13) the last pair is SPLIT: wavefront 12 takes its first disparity,
  //     wavefront 13 the second -- SIMDs 0 and 1 then carry 3.5 pairs each, 2 and 3 carry 3 pairs and a loader,
  //     instead of 4 / 3 / 3 / 3 + loader.
  constexpr bool COST = (MODE & (kASad | kAMse)) != 0;             // SAD / MSE cost volume: value rows only, no statistics
  const bool has_helper = !COST && n_act <= kAWaves - 1;
  const bool split_last = n_act == kAWaves - 2;
  const int n_out_rows = h_hi - h_lo;
  auto chunk_is_light = [&](int ch) { return alld_chunk_is_light<ROWS>(ch, n_out_rows); };
  constexpr int LV = ROWS * kAValuesPerRow, LS = ROWS * kAStatsPerRow;   // DMA instructions per chunk: values, statistics
  static_assert(LV + LS < 64, "in-flight DMA count must fit vmcnt");
  static_assert(kABufs == 3, "the loaders run two chunks ahead of the consumers");
  const int c_lo = w_lo - 4;
  const int total = n_pass * n_chunks;                             // chunks of the whole workgroup, all passes
  const AlldStatsArgs sa = {m0 + (long)f * H * Wp + 4, v0 + (long)f * H * Wp + 4, m1 + (long)f * st1_frame_stride,
                            v1 + (long)f * st1_frame_stride, lds, Wp, W1, xoff, c_lo, r_begin, h_lo, h_hi, n_pass, n_chunks, dgs, rot};

  if (wave == kAWaves - 1 && has_helper) {
    // ------------------------------ statistics loader (spare consumer wavefront) ------------------------------
    alld_stats_loader<ROWS>(sa, lane);
#ifdef CTD_STAMPS
    dump_stamps();
#endif
    return;                                                        // (the emit rows are dealt to the consumer wavefronts only)
  }
  if (wave == kAWaves) {
    // every chunk barrier waits for this wavefront's DMA issue and halo sums: it goes first on its SIMD
    __builtin_amdgcn_s_setprio(3);
    // ------------------------------ value loader + halo wavefront ------------------------------
    const float* a_img = ac + (long)f * H * Wp + 4;
    const float* b_img = bc + (long)f * st1_frame_stride;
    const int aq0 = min(c_lo + 4 * lane, Wp - 8), aq1 = min(c_lo + 256 + 4 * lane, Wp - 8);
    const bool a_tail = 256 + 4 * lane < kAA, s_tail = 256 + 4 * lane < kASpanPad;
    int i_slot = 0, i_pass = 0, i_ch = 0, i_n = 0;                 // the next chunk to issue: ring slot, pass, chunk in the pass
    auto issue_chunk = [&]() {                                     // returns whether the statistics went with it
      float* buf = lds + i_slot * (ROWS * kAPack);
      const int i_grp = i_pass + rot;
      const int xb = c_lo - (i_grp * dgs + dgs - 1);               // unclamped pattern column of span slot 0
      const int sq0 = min(xb + xoff + 4 * lane, W1 - 4), sq1 = min(xb + xoff + 256 + 4 * lane, W1 - 4);
#pragma unroll
      for (int s2 = 0; s2 < ROWS; ++s2) {
        const int r = r_begin + i_ch * ROWS + s2;
        const int rc = clampi(r, 0, H - 1);
        float* pk = buf + s2 * kAPack;
        dma_quad(a_img + (long)rc * Wp + aq0, pk);
        dma_quad(b_img + (long)rc * W1 + sq0, pk + kAOffB);
        if (a_tail) dma_quad(a_img + (long)rc * Wp + aq1, pk + 256);
        if (s_tail) dma_quad(b_img + (long)rc * W1 + sq1, pk + kAOffB + 256);
      }
      const bool with_stats = !COST && !has_helper && !chunk_is_light(i_ch);
      if (with_stats) alld_issue_stats<ROWS>(sa, i_pass, i_ch, i_slot, lane);
      ++i_n;
      i_slot = i_slot == kABufs - 1 ? 0 : i_slot + 1;
      if (++i_ch == n_chunks) { i_ch = 0; ++i_pass; }
      return with_stats;
    };
    // halo job of this lane: consumer pair cw, disparity j, side (0 = quad left of the tile, 1 = right of it)
    const int cw = lane >> 2, hj = (lane >> 1) & 1, side = lane & 1;
    const bool has_job = cw < n_act;
    const int a_slot = side ? (kAA - 4) : 0;
    const int b_slot = has_job ? a_slot + (dgs - 1) - (cw * 2 + hj) : 0;
    // SAD / MSE: right of the image the per-pixel plane is a COPY of its last column (the tap column is clamped before
    // the shift), not the pairing a[W-1], b[w0 - d] of the NCC border rule: the four halo columns right of the LAST tile
    // all take the pattern sample of column W - 1 (span slot b_slot - 1).  (Images that end inside a tile: see
    // cost_border_kernel.)
    const bool copy_last = COST && side == 1 && w_lo + 256 == W;
    float hP[4][2], hT[4][6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hP[i][0] = hP[i][1] = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) hT[i][k] = 0.f;
    }
    // vertical ring update of the halo quad for the rows of one chunk; UB = ring phase of its first row.  (The rings
    // run on across passes: the first 8 rows of a pass are warm-up rows, whatever the rings held before.)
    int h_slot = 0;                                                // ring slot of the next chunk to get its halo sums
    auto halo_chunk = [&](auto ub_tag) {
      constexpr int UB = decltype(ub_tag)::value;
      const float* buf = lds + h_slot * (ROWS * kAPack);
#pragma unroll
      for (int s2 = 0; s2 < ROWS; ++s2) {
        const int u = (UB + s2) % 6;
        const float* pk = buf + s2 * kAPack;
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float p2 = 0.f;
          if (has_job) {
            const float av = pk[a_slot + i], bv = pk[kAOffB + (copy_last ? b_slot - 1 : b_slot + i)];
            p2 = (MODE & kASad) ? fabsf(av - bv) : ((MODE & kAMse) ? (av - bv) * (av - bv) : av * bv);
          }
          const float t3 = p2 + hP[i][(u + 1) % 2] + hP[i][u % 2];
          hP[i][u % 2] = p2;
          x[i] = t3 + hT[i][(u + 3) % 6] + hT[i][u % 6];
          hT[i][u % 6] = t3;
        }
        float o4[4];
        if (side) {                                                // prefix sums: columns 0..i of the right quad
          o4[0] = x[0]; o4[1] = o4[0] + x[1]; o4[2] = o4[1] + x[2]; o4[3] = o4[2] + x[3];
        } else {                                                   // suffix sums: columns i..3 of the left quad
          o4[3] = x[3]; o4[2] = o4[3] + x[2]; o4[1] = o4[2] + x[1]; o4[0] = o4[1] + x[0];
        }
        if (has_job) {
          float* hq = const_cast<float*>(pk) + kAOffH + lane * 4;  // lane == ((cw*2 + hj)*2 + side)
          hq[0] = o4[0]; hq[1] = o4[1]; hq[2] = o4[2]; hq[3] = o4[3];
        }
      }
      h_slot = h_slot == kABufs - 1 ? 0 : h_slot + 1;
    };
    // after issuing chunk X (LV value DMAs, + LS statistics DMAs when it carried them) everything older has landed once
    // at most that many are outstanding
    auto wait_older = [&](bool with_stats) {
      if (with_stats) wait_vmcnt<LV + LS>(); else wait_vmcnt<LV>();
    };
    issue_chunk();                                                 // total >= 2 (a pass has CPI >= 2 chunks)
    wait_older(issue_chunk());                                     // chunk 0 has landed
    halo_chunk(std::integral_constant<int, 0>{});
    wait_lgkmcnt0();
    wg_barrier();
    // chunk g of the flat sequence is chunk g % n_chunks of pass g / n_chunks; n_chunks is a multiple of CPI, so the
    // ring phase of the first row of chunk g is (g % CPI) * ROWS, across pass boundaries too
    for (int g = 0; g < total; g += CPI) {
#pragma unroll
      for (int cc = 0; cc < CPI; ++cc) {
        if (i_n < total) {
          wait_older(issue_chunk());                               // chunk g + cc + 1 has landed
        } else {
          wait_vmcnt<0>();
        }
        if (g + cc + 1 < total) {
          if (cc == 0) halo_chunk(std::integral_constant<int, (1 % CPI) * ROWS>{});
          else if (cc == 1) halo_chunk(std::integral_constant<int, (2 % CPI) * ROWS>{});
          else halo_chunk(std::integral_constant<int, (3 % CPI) * ROWS>{});
        }
        wait_lgkmcnt0();
        CTD_STAMP_ARRIVE(st_lds, kAWaves, (g + cc) / n_chunks, (g + cc) % n_chunks, lane);
        wg_barrier();
        CTD_STAMP_LEAVE(st_lds, kAWaves, (g + cc) / n_chunks, (g + cc) % n_chunks, lane);
      }
    }
#ifdef CTD_STAMPS
    dump_stamps();
#endif
    return;
  }

  // Consumers.  Two copies of the regular loop -- the sub-quad shift of the pattern-side operands, (dgs - 2 - 2 * pair) % 4,
  // alternates with the pair index -- and, for a split last pair, one copy per half.
  {
    const int pair = (split_last && wave == n_act) ? n_act - 1 : wave;
    const bool ks2 = ((dgs - 2 - 2 * pair) & 2) != 0;
    if (split_last && wave >= n_act - 1) {
      if (wave == n_act - 1) {
        if (ks2) alld_consume<MODE, 2, ROWS, 1>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
        else alld_consume<MODE, 0, ROWS, 1>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
      } else {
        if (ks2) alld_consume<MODE, 2, ROWS, 2>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
        else alld_consume<MODE, 0, ROWS, 2>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
      }
    } else if (ks2) {
      alld_consume<MODE, 2, ROWS>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
    } else {
      alld_consume<MODE, 0, ROWS>(lds, rank_lds, out, pair, f, lane, w_lo, h_lo, h_hi, r_begin, n_iters, n_pass, rot, dgs, H, W, D, wave);
    }
  }

#ifdef CTD_STAMPS
  if constexpr (!RANK) { dump_stamps(); return; }
#endif
  if constexpr (!RANK) return;                                     // plain volume call: nothing to emit
  // ---- emit: the band's final {top, second} -> index, best score, work-list flag.  The last chunk barrier (behind
  // every wavefront's lgkmcnt(0)) has made all slot updates visible.
  const unsigned l4 = 4u * (unsigned)lane;
  const bool lane_out = w_lo + (int)l4 < W;
  const unsigned margin = rank_eps >= 0.f ? key_margin_units(rank_eps) : 0u;
  const int n_emit = has_helper ? kAWaves - 1 : kAWaves;           // wavefronts that reach this point
  for (int row = wave; row < h_hi - h_lo; row += n_emit) {
    const unsigned* sl = rank_lds + row * 512 + lane;
    const long p0 = ((long)f * H + h_lo + row) * W + w_lo + l4;    // first of the lane's four pixels
    long d64[4];
    f32x4 b4;
    unsigned listed4 = 0, n_hard = 0;
    bool hard[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned t = sl[64 * i], q = sl[256 + 64 * i];
      d64[i] = (long)(kTagMask - (t & kTagMask));
      const unsigned ft = t >> kTagBits, fq = q >> kTagBits;       // 23-bit fixed-point scores
      b4[i] = __uint_as_float(0x40800000u | ft) - kKeyBias;
      // runner-up within the margin of the best -> exact re-scoring (q == 0: the pixel has a single score)
      hard[i] = lane_out && rank_eps >= 0.f && q != 0u && ft - fq <= margin;
      if (hard[i]) { listed4 |= 1u << (8 * i); ++n_hard; }
    }
    if (lane_out) {
      typedef long l64x2 __attribute__((ext_vector_type(2)));
      *(l64x2*)(idx_out + p0) = l64x2{d64[0], d64[1]};
      *(l64x2*)(idx_out + p0 + 2) = l64x2{d64[2], d64[3]};
      *(f32x4*)(best_out + p0) = b4;
      *(unsigned*)(flags_out + p0) = listed4;
    }
    // all pixels of the row share the work-list key: one counter update per wavefront and row that has any
    if (__any(n_hard != 0)) {
      unsigned before = n_hard;                                    // exclusive prefix over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(before, o);
        if (lane >= o) before += v;
      }
      const unsigned total_hard = __shfl(before, 63);
      before -= n_hard;
      const int key = worklist_key(work, p0);
      unsigned base = 0;
      if (lane == 0) base = atomicAdd(work.counters + key * kWorkListStride, total_hard);
      base = __shfl(base, 0);
      int64_t* dst = work.list + (long)key * work.seg_cap + base + before;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (hard[i]) *dst++ = p0 + i;
    }
  }
#ifdef CTD_STAMPS
  dump_stamps();
#endif
}

struct FastWorkspace {
  float *ac, *m0, *v0;        // centred frames, their window mean (centred) / deviation planes   [N*C][H][W]
  float *bc, *m1, *v1;        // same for the pattern, per UNCLAMPED window-centre column          [..][H][W1]
  int Wp;                     // frame plane row pitch: W + 8, column c lives at c + 4 (replicate border baked in)
  int W1, xoff;               // pattern plane row pitch and origin: column x lives at x + xoff
  unsigned* counters;         // [0] flagged frame windows, [1] flagged pattern windows (see ncc_fixup_kernel)
  unsigned long long *flag_a, *flag_b;
  unsigned long long* run_rows;   // (pattern image << 20 | h) of the listed fully clamped pattern windows
  float* run_vals;            // [frames][H][D] exact values of the fully clamped runs (ncc_fixup_runs_kernel)
  size_t bytes;               // end of the volume pass's own workspace; the ranking buffers (RankPlan) follow
};

static FastWorkspace fast_workspace(void* base, int frames, int C, int H, int W, int D, bool per_frame_pattern) {
  FastWorkspace ws;
  // + 32: the last disparity group (tile-256 kernel) / pass (all-D kernel) stages pattern columns for up to 29
  // disparities past D; their outputs are never stored, their operands still come from inside the plane
  const int Dpad = (D + kFDG - 1) / kFDG * kFDG + 32;
  // x = w - d ranges over [-(Dpad-1) - 4, W + 3] (4 halo columns either side).  xoff = 3 (mod 4) makes
  // the first span slot of every workgroup 16-byte aligned (w_lo and the disparity-group base are
  // multiples of 4), which the dwordx4 LDS-DMA of the wide kernel relies on.
  ws.xoff = Dpad + 3;
  ws.W1 = (int)align_up((size_t)(W + 4 + ws.xoff), 4);
  ws.Wp = (int)align_up((size_t)(W + 8), 4);
  size_t n0 = align_up((size_t)frames * C * H * ws.Wp * sizeof(float), 256);
  size_t n1 = align_up((size_t)(per_frame_pattern ? frames : 1) * C * H * ws.W1 * sizeof(float), 256);
  char* p = (char*)base;
  ws.ac = (float*)p;
  ws.m0 = (float*)(p + n0);
  ws.v0 = (float*)(p + 2 * n0);
  ws.bc = (float*)(p + 3 * n0);
  ws.m1 = (float*)(p + 3 * n0 + n1);
  ws.v1 = (float*)(p + 3 * n0 + 2 * n1);
  // flag lists: room for every frame window and every pattern window the outputs can touch
  size_t nfa = align_up((size_t)frames * C * H * W * sizeof(unsigned long long), 256);
  size_t nfb = align_up((size_t)(per_frame_pattern ? frames : 1) * C * H * ws.W1 * sizeof(unsigned long long), 256);
  ws.counters = (unsigned*)(p + 3 * n0 + 3 * n1);
  ws.flag_a = (unsigned long long*)(p + 3 * n0 + 3 * n1 + 256);
  ws.flag_b = (unsigned long long*)(p + 3 * n0 + 3 * n1 + 256 + nfa);
  size_t nrr = align_up((size_t)(per_frame_pattern ? frames : 1) * C * H * sizeof(unsigned long long), 256);
  ws.run_rows = (unsigned long long*)(p + 3 * n0 + 3 * n1 + 256 + nfa + nfb);
  ws.run_vals = (float*)(p + 3 * n0 + 3 * n1 + 256 + nfa + nfb + nrr);
  ws.bytes = 3 * n0 + 3 * n1 + 256 + nfa + nfb + nrr + align_up((size_t)frames * H * D * sizeof(float), 256);
  return ws;
}

size_t ncc_fast_workspace_bytes(int frames, int C, int H, int W, int D, int bs, bool per_frame_pattern) {
  (void)bs;
  return fast_workspace(nullptr, frames, C, H, W, D, per_frame_pattern).bytes;
}

bool ncc_fast_rank_supported(int C, int H, int W, int D, int bs) {
  (void)H;
  return C == 1 && bs == 9 && W % 4 == 0 && D <= 512;      // the tile-256 kernel, single channel
}

// ranking buffers behind the volume pass's workspace
static RankPlan rank_plan(void* base, size_t offset, int frames, int H, int W, int D) {
  RankPlan rp;
  rp.eps = 0.f;
  rp.idx = nullptr;
  rp.best = nullptr;
  (void)D;
  const size_t nflag = align_up((size_t)frames * H * W, 256);
  // work list: kWorkListParts segments keyed by image row, behind their counters (one cache line each)
  const long seg_cap = (long)ceil_div(frames * H, kWorkListParts) * W;
  const size_t ncnt = align_up(sizeof(unsigned) * kWorkListStride * kWorkListParts, 256);
  const size_t nlist = ncnt + align_up((size_t)kWorkListParts * seg_cap * sizeof(int64_t), 256);
  char* p = (char*)base + offset;
  rp.flags = (unsigned char*)p;
  rp.work.counters = (unsigned*)(p + nflag);
  rp.work.list = (int64_t*)(p + nflag + ncnt);
  rp.work.parts = kWorkListParts;
  rp.work.row_width = W;
  rp.work.seg_cap = seg_cap;
  rp.best_scratch = (float*)(p + nflag + nlist);
  rp.bytes = offset + nflag + nlist + align_up((size_t)frames * H * W * sizeof(float), 256);
  return rp;
}

// How the all-D kernel cuts a call into workgroups: disparities per pass (dealt evenly over ceil(D / 30) passes), and
// the band height.  One workgroup per CU is resident (LDS), every workgroup costs about (rows + 8 warm-up rows) x passes,
// so the bands are chosen to minimise ceil(workgroups / 256) x (band rows rounded up to the 6-row unroll + 8).
struct AlldPlan {
  int n_pass, dgs, band_rows, bands, chunk_rows, n_psplit;
  size_t lds;
};
// `ranked`: the workgroup keeps the ranking of its pixels in LDS (band height limited by the slots, every disparity in one
// workgroup).  Otherwise the band may be as tall as the image and the passes may be split over workgroups.
// compute units of the current device (the plan's cost model counts rounds of one workgroup per CU); asked once per device
static int device_cu_count() {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    cus[dev] = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0 ? n : 256;
  }
  return cus[dev];
}

static AlldPlan alld_plan_compute(int frames, int H, int W, int D, bool ranked, int n_cu);
// (the unranked plan tries every band height for every pass split: ~5 K candidates at 1024 x 1024 x 256, on the launch path
// of a sub-millisecond call -- the last few shapes' plans are kept; a plan is a pure function of its key)
static AlldPlan alld_plan(int frames, int H, int W, int D, bool ranked = true) {
  struct Key { int frames, H, W, D, ranked, n_cu; AlldPlan plan; };
  static thread_local Key cache[8];
  static thread_local int next = 0;
  const int n_cu = device_cu_count();
  for (const Key& k : cache)
    if (k.frames == frames && k.H == H && k.W == W && k.D == D && k.ranked == (int)ranked && k.n_cu == n_cu && k.frames > 0) return k.plan;
  Key& k = cache[next];
  next = (next + 1) % 8;
  k = Key{frames, H, W, D, (int)ranked, n_cu, alld_plan_compute(frames, H, W, D, ranked, n_cu)};
  return k.plan;
}

static AlldPlan alld_plan_compute(int frames, int H, int W, int D, bool ranked, int n_cu) {
  AlldPlan ap;
  ap.n_psplit = 1;
  // The disparities are dealt evenly over the ceil(D / 30) passes (D = 128: 5 x 26 on 13 wavefronts).  Four full passes
  // of 30 and a last one of 8 measured 4 % SLOWER: a pass costs about the same whether 13 or 15 wavefronts work in it
  // (the row's dependent chain and the chunk barrier, not the sum of the wavefronts' instructions), so the short pass
  // is a whole pass's time for a quarter of its outputs.
  ap.n_pass = ceil_div(D, kADGMax);
  ap.dgs = 2 * ceil_div(ceil_div(D, 2), ap.n_pass);
  const long base = (long)ceil_div(W, 256) * frames;
  // Band height and chunk size: 3-row chunks allow bands of up to 46 rows, 2-row chunks (a third more chunk barriers,
  // priced at kTwoRowPenalty) up to 57 -- config 2 is then ONE round of 256 workgroups of 54 rows (62 row steps per pass,
  // 66 with the unroll) instead of two rounds of 27 (2 x 36).
  double best_cost = -1;
  ap.band_rows = H < 44 ? H : 44;
  ap.chunk_rows = 3;
  for (int cr = 3; cr >= kAllowTwoRowChunks; --cr) {
    const int max_rows = alld_max_band_rows(cr) < 44 || cr == 2 ? alld_max_band_rows(cr) : 44;
    for (int rows = max_rows; rows >= 4; --rows) {
      if (rows > H) continue;
      const long wgs = base * ceil_div(H, rows);
      const double cost = (double)((wgs + n_cu - 1) / n_cu) * (double)(ceil_div(rows + 8, 6) * 6) * (cr == 2 ? kTwoRowPenalty : 1.0);
      if (best_cost < 0 || cost < best_cost) { best_cost = cost; ap.band_rows = rows; ap.chunk_rows = cr; }
    }
  }
  if (!ranked) {
    // no rank slots: 3-row chunks, any band height; try every pass split
    ap.chunk_rows = 3;
    best_cost = -1;
    for (int sp = 1; sp <= ap.n_pass; ++sp) {
      const int ppg = ceil_div(ap.n_pass, sp);
      if (ceil_div(ap.n_pass, ppg) != sp) continue;                // (the same passes per workgroup with fewer workgroups)
      for (int rows = H; rows >= 4; --rows) {
        const long wgs = base * ceil_div(H, rows) * sp;
        const double cost = (double)((wgs + n_cu - 1) / n_cu) * ppg * (double)(ceil_div(rows + 8, 6) * 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; ap.band_rows = rows; ap.n_psplit = sp; }
      }
    }
  }
  ap.bands = ceil_div(H, ap.band_rows);
  ap.band_rows = ceil_div(H, ap.bands);                          // the same number of bands, evenly tall
  ap.lds = (size_t)ap.band_rows * 2048 + sizeof(float) * kABufs * ap.chunk_rows * kAPack;
#ifdef CTD_STAMPS
  ap.lds += sizeof(unsigned) * kStampWords;
#endif
  return ap;
}

void ncc_fast_rank_offsets(int frames, int H, int W, int D, bool per_frame_pattern, size_t* off) {
  const FastWorkspace ws = fast_workspace(nullptr, frames, 1, H, W, D, per_frame_pattern);
  const RankPlan rp = rank_plan(nullptr, ws.bytes, frames, H, W, D);
  const AlldPlan ap = alld_plan(frames, H, W, D);
  off[0] = (size_t)ap.band_rows; off[1] = (size_t)rp.flags; off[2] = (size_t)rp.work.counters; off[3] = (size_t)rp.work.list;
  off[4] = (size_t)ap.n_pass;
}

size_t ncc_fast_rank_workspace_bytes(int frames, int H, int W, int D, bool per_frame_pattern) {
  const size_t off = fast_workspace(nullptr, frames, 1, H, W, D, per_frame_pattern).bytes;
  return rank_plan(nullptr, off, frames, H, W, D).bytes;
}

static int launch_prepass(const PrepassJob& ja, const PrepassJob& jb, int H, int W, int bs, const WorkList* work,
                          hipStream_t stream) {
  const int TRr = kSTH + bs - 1, TCc = kSTW + bs - 1;
  const bool f32 = bs == 9 && kPrepassF32;
  size_t lds = (f32 ? sizeof(float) : sizeof(double)) * 2 * TRr * kSTW + sizeof(float) * (size_t)TRr * TCc;
  if (lds > 60 * 1024) return CTD_ERR_UNSUPPORTED;
  const int w_out = jb.nimg == 0 ? ja.W_out : (ja.nimg == 0 || jb.W_out > ja.W_out ? jb.W_out : ja.W_out);
  dim3 grid(ceil_div(w_out, kSTW), ceil_div(H, kSTH), ja.nimg + jb.nimg), block(kSTW, kSRows);
  auto kern = ncc_prepass_kernel<0, false>;
  if (bs == 9) kern = f32 ? ncc_prepass_kernel<9, true> : ncc_prepass_kernel<9, false>;
  hipLaunchKernelGGL(kern, grid, block, lds, stream, ja, jb, H, W, bs, work ? work->counters : nullptr, work ? work->parts : 0);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

static int pick_bands(long wg_per_band, int H, int bs) {
  // enough workgroups to fill 256 CUs a few times over, few enough that the (bs-1)-row
  // warm-up of every band stays a small fraction of its rows
  int bands = 1;
  while (bands < 8 && wg_per_band * bands < 2048 && H / (bands * 2) >= 8 * (bs - 1)) bands *= 2;
  return bands;
}

template <int BS>
static int launch_fast(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H,
                       int W, int D, const FastWorkspace& ws, const RankPlan* rank, hipStream_t stream) {
  constexpr int WOUT = 64 - (BS - 1);
  const long st1_stride = in1_frame_stride ? (long)C * H * ws.W1 : 0;
  if (rank && !(BS == 9 && W % 4 == 0 && C == 1 && ((uintptr_t)out) % 16 == 0)) return CTD_ERR_UNSUPPORTED;
  if (rank) {
    // ranked call: the all-D kernel (one workgroup per column tile, band and frame, every disparity)
    const AlldPlan ap = alld_plan(frames, H, W, D);
    const int n_items = ceil_div(W, 256) * ap.bands * frames;
    dim3 grid(n_items), block(64 * (kAWaves + 1));
    auto kern = ap.chunk_rows == 3 ? (out ? ncc_fast_alld_kernel<kAStore | kARank, 3> : ncc_fast_alld_kernel<kARank, 3>)
                                   : (out ? ncc_fast_alld_kernel<kAStore | kARank, 2> : ncc_fast_alld_kernel<kARank, 2>);
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ap.lds));
    timing_begin(stream);
    hipLaunchKernelGGL(kern, grid, block, ap.lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1, ws.v1, st1_stride, out,
                       rank->idx, rank->best, rank->flags, rank->work, rank->eps, frames, n_items, H, W, D, ap.band_rows,
                       ap.n_pass, ap.dgs,
                       ws.Wp, ws.W1, ws.xoff, 1);
    timing_end(stream, W);
    CTD_LAUNCH_CHECK();
    return CTD_OK;
  }
  if (BS == 9 && W % 4 == 0 && ((uintptr_t)out) % 16 == 0 && C == 1) {
    // plain volume, single channel: the all-D kernel without the ranking (the same bits as with it)
    const AlldPlan ap = alld_plan(frames, H, W, D, false);
    const int n_items = ceil_div(W, 256) * ap.bands * frames * ap.n_psplit;
    size_t lds = sizeof(float) * kABufs * ap.chunk_rows * kAPack;
#ifdef CTD_STAMPS
    lds += sizeof(unsigned) * kStampWords;
#endif
    dim3 grid(n_items), block(64 * (kAWaves + 1));
    auto kern = ap.chunk_rows == 3 ? ncc_fast_alld_kernel<kAStore, 3> : ncc_fast_alld_kernel<kAStore, 2>;
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    timing_begin(stream);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1, ws.v1, st1_stride, out,
                       (int64_t*)nullptr, (float*)nullptr, (unsigned char*)nullptr, WorkList{}, -1.f, frames, n_items, H, W, D,
                       ap.band_rows, ap.n_pass, ap.dgs, ws.Wp, ws.W1, ws.xoff, ap.n_psplit);
    timing_end(stream, W);
    CTD_LAUNCH_CHECK();
    return CTD_OK;
  }
  if (BS == 9 && W % 4 == 0 && ((uintptr_t)out) % 16 == 0) {
    // several channels (accumulating launches): 256-column tiles per disparity group, every store a full aligned KB
    const int n_dg = ceil_div(D, kTDG);
    const int n_tiles = ceil_div(W, kTTile);
    // Bands of ~44 rows (5.5x the (bs-1)-row warm-up).  Measured on config 2 (H = 432): 4 bands 0.448 ms, 8: 0.418,
    // 10: 0.394, 12: 0.398, 16: 0.439 -- short bands cost warm-up rows but interleave the store-free warm-up of
    // some workgroups with the store phase of others and even out the tail.
    const int bands = H >= 66 ? (H + 22) / 44 : 1;
    const int band_rows = ceil_div(H, bands);
    dim3 grid(n_tiles, ceil_div(H, band_rows), frames * n_dg), block(64 * (kTWaves + 1));
    const size_t lds = sizeof(float) * kTBufs * kTRows * kTPack;
    for (int c = 0; c < C; ++c) {
      auto kern = c == 0 ? ncc_fast_t256_kernel<false> : ncc_fast_t256_kernel<true>;
      if (lds > 64 * 1024)
        CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      timing_begin(stream);
      hipLaunchKernelGGL(kern, grid, block, lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1, ws.v1, st1_stride, out, C, c, H,
                         W, D, band_rows, n_dg, ws.Wp, ws.W1, ws.xoff);
      timing_end(stream, W);
      CTD_LAUNCH_CHECK();
    }
    return CTD_OK;
  }
  // column split: full 248-column wide tiles (plus one more when the remainder is large),
  // the rest in 64-(BS-1)-column narrow tiles
  int n_wide = W / kWOut;
  if (W - n_wide * kWOut > kWOut / 2) ++n_wide;
  const int w_rem = n_wide * kWOut < W ? n_wide * kWOut : W;     // first column of the narrow part
  for (int c = 0; c < C; ++c) {
    if (n_wide > 0) {
      const int n_dg = ceil_div(D, kWDG);
      const int bands = pick_bands((long)n_wide * frames * n_dg, H, BS);
      const int band_rows = ceil_div(H, bands);
      dim3 grid(n_wide, ceil_div(H, band_rows), frames * n_dg), block(64 * (kWWaves + 1));
      const size_t lds = sizeof(float) * kWBufs * kWRows * kWPack;
      const bool vec4 = (W % 4 == 0) && (((uintptr_t)out) % 16 == 0);
      auto kern = c == 0 ? (vec4 ? ncc_fast_wide_kernel<BS, false, true> : ncc_fast_wide_kernel<BS, false, false>)
                         : (vec4 ? ncc_fast_wide_kernel<BS, true, true> : ncc_fast_wide_kernel<BS, true, false>);
      timing_begin(stream);
      hipLaunchKernelGGL(kern, grid, block, lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1, ws.v1, st1_stride, out, C,
                         c, H, W, D, band_rows, n_dg, ws.Wp, ws.W1, ws.xoff);
      timing_end(stream, w_rem);
      CTD_LAUNCH_CHECK();
    }
    if (w_rem < W) {
      const int n_dg = ceil_div(D, kFDG);
      const int n_tiles = ceil_div(W - w_rem, WOUT);
      const int bands = pick_bands((long)n_tiles * frames * n_dg, H, BS);
      const int band_rows = ceil_div(H, bands);
      dim3 grid(n_tiles, ceil_div(H, band_rows), frames * n_dg), block(64 * (kFWaves + 1));
      const size_t lds = sizeof(float) * kFBufs * kFRows * kFPack;
      if (c == 0)
        hipLaunchKernelGGL((ncc_fast_kernel<BS, false>), grid, block, lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1,
                           ws.v1, st1_stride, out, C, c, H, W, D, band_rows, n_dg, ws.Wp, ws.W1,
                           ws.xoff, w_rem);
      else
        hipLaunchKernelGGL((ncc_fast_kernel<BS, true>), grid, block, lds, stream, ws.ac, ws.m0, ws.v0, ws.bc, ws.m1,
                           ws.v1, st1_stride, out, C, c, H, W, D, band_rows, n_dg, ws.Wp, ws.W1,
                           ws.xoff, w_rem);
      CTD_LAUNCH_CHECK();
    }
  }
  return CTD_OK;
}

static int launch_fixup(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H,
                        int W, int D, int bs, const FastWorkspace& ws, bool per_frame, const RankPlan* rank, const float* best,
                        unsigned* scan_counter, hipStream_t stream) {
  // reference-order recomputation of the outputs of listed (ill-conditioned) windows; the grid drains
  // immediately when nothing was listed
  const size_t lds = sizeof(float) * 4 * (3 * (size_t)bs * bs + 2 * (size_t)bs * (bs + D - 1));
  if (lds > 160 * 1024) return CTD_ERR_UNSUPPORTED;
  auto fix = bs == 9 ? ncc_fixup_kernel<9> : ncc_fixup_kernel<0>;
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)fix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(fix, dim3(kFixupBlocks), dim3(256), lds, stream, in0, in1, in1_frame_stride, out, ws.counters,
                     ws.flag_a, ws.flag_b, ws.run_vals, rank ? best : nullptr, rank ? (unsigned long long*)rank->idx : nullptr,
                     rank ? rank->eps : -1.f,
                     rank ? (unsigned*)rank->flags : nullptr, rank ? rank->work : WorkList{}, frames, C, H, W, D, bs);
  CTD_LAUNCH_CHECK();
  if (!out || rank) return CTD_OK;                           // nothing to spread without a volume; ranked calls spread in their tail kernel
  const size_t lds_rows = sizeof(int) * (size_t)C * H;
  if (lds_rows > 64 * 1024) return CTD_ERR_UNSUPPORTED;
  // (the old scan's work-list counter sits at the start of the workspace, which the volume kernel is done with by
  // now: cleared by the runs kernel instead of by a memset of its own)
  hipLaunchKernelGGL(ncc_fixup_runs_kernel, dim3((unsigned)(frames * ceil_div(D, kRunPlanes)), 4), dim3(256), lds_rows, stream, out, ws.run_vals,
                     ws.counters, ws.run_rows, per_frame ? 1 : 0, frames, C, H, W, D, bs, scan_counter);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// `rank` non-null (in: eps, idx, best -- best may be null): the all-D kernel computes the volume (`out` may be null:
// nothing is materialised then), ranks every pixel's scores in LDS and writes idx / best / work-list flags itself;
// *rank comes back filled with the buffers the later passes need and the call STOPS after that kernel -- the caller
// runs ncc_fast_fixup_ranked (which needs the best scores and indices), then rank_resolve_f32.
// Pattern half of the pre-pass alone (ctd_xcorrvol_pattern_prepare_f32): the pattern's planes, its list of listed windows
// and run rows stay in `workspace`; calls with `pattern_prepared` on the SAME workspace and shape then skip that half (the
// reference prepares the pattern once per run, model/exp_synph.py:64-71).  The layout depends on `frames`.
int ncc_fast_prepare_pattern_f32(const float* in1, long in1_frame_stride, int frames, int C, int H, int W, int D, int bs,
                                 void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (bs < 2 || bs > 33) return CTD_ERR_UNSUPPORTED;
  if (H >= (1 << 20) || W + D >= (1 << 19) || D > 512 || (long)frames * C >= (1 << 24)) return CTD_ERR_UNSUPPORTED;
  const bool per_frame = in1_frame_stride != 0;
  if (per_frame && in1_frame_stride != (long)C * H * W) return CTD_ERR_INVALID_ARG;
  FastWorkspace ws = fast_workspace(workspace, frames, C, H, W, D, per_frame);
  if (workspace == nullptr || workspace_bytes < ws.bytes) return CTD_ERR_WORKSPACE;
  CTD_HIP_TRY(hipMemsetAsync(ws.counters, 0, 16, stream));
  const PrepassJob ja = {in1, (long)H * W, ws.ac, ws.m0, ws.v0, 0, W, 0, ws.counters, ws.flag_a, 0, W,
                         nullptr, nullptr, -(double)(bs * bs), kFlagRatio / C, ws.Wp, 4, 4};   // (no frame images in this launch)
  const PrepassJob jb = {in1, (long)H * W, ws.bc, ws.m1, ws.v1, -ws.xoff, ws.W1, (per_frame ? frames : 1) * C,
                         ws.counters + 1, ws.flag_b, -(bs - 1 - bs / 2), W, ws.counters + 2, ws.run_rows, 1.0, kFlagRatio / C, ws.W1, 0, 0};
  return launch_prepass(ja, jb, H, W, bs, nullptr, stream);
}

int ncc_fast_f32(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H, int W,
                 int D, int bs, void* workspace, size_t workspace_bytes, RankPlan* rank, bool pattern_prepared,
                 hipStream_t stream, const FusedLcn* fused) {
  if (bs < 2 || bs > 33) return CTD_ERR_UNSUPPORTED;
  if (fused && (C != 1 || !lcn_stream_supported(H, W, fused->radius, bs))) return CTD_ERR_UNSUPPORTED;
  if (H >= (1 << 20) || W + D >= (1 << 19) || D > 512 || (long)frames * C >= (1 << 24)) return CTD_ERR_UNSUPPORTED;
  if (!out && !rank) return CTD_ERR_INVALID_ARG;
  if (rank && !ncc_fast_rank_supported(C, H, W, D, bs)) return CTD_ERR_UNSUPPORTED;
  const bool per_frame = in1_frame_stride != 0;
  FastWorkspace ws = fast_workspace(workspace, frames, C, H, W, D, per_frame);
  size_t need = ws.bytes;
  if (rank) {
    const RankPlan in = *rank;
    *rank = rank_plan(workspace, ws.bytes, frames, H, W, D);
    rank->eps = in.eps;
    rank->idx = in.idx;
    rank->best = in.best ? in.best : rank->best_scratch;
    rank->run_vals = ws.run_vals;
    rank->run_rows = ws.run_rows;
    rank->counters = ws.counters;
    rank->flag_a = ws.flag_a;
    rank->flag_b = ws.flag_b;
    rank->v1 = ws.v1;
    rank->W1 = ws.W1;
    rank->xoff = ws.xoff;
    need = rank->bytes;
  }
  if (workspace == nullptr || workspace_bytes < need) return CTD_ERR_WORKSPACE;
  if (rank && !rank->idx) return CTD_ERR_INVALID_ARG;
  // counters: [0] listed frame windows, [1] listed pattern windows, [2] listed run rows -- the last two belong to the
  // pattern and survive when it was prepared
  // (a ranked call on a prepared pattern finds [0] at zero: the prepare call and every ranked call's tail kernel leave it so)
  if (!(rank && pattern_prepared)) CTD_HIP_TRY(hipMemsetAsync(ws.counters, 0, pattern_prepared ? 4 : 16, stream));
  // window statistics of the frames (per pixel) and of the pattern (per unclamped window-centre column
  // x = w - d; windows x <= -(bs-1-bs/2) are all the same fully clamped window and are listed once), one launch
  if (per_frame && in1_frame_stride != (long)C * H * W) return CTD_ERR_INVALID_ARG;
  const PrepassJob ja = {in0, (long)H * W, ws.ac, ws.m0, ws.v0, 0, W, frames * C, ws.counters, ws.flag_a, 0, W,
                         nullptr, nullptr, -(double)(bs * bs), kFlagRatio / C, ws.Wp, 4, 4};
  const PrepassJob jb = {in1, (long)H * W, ws.bc, ws.m1, ws.v1, -ws.xoff, ws.W1,
                         pattern_prepared ? 0 : (per_frame ? frames : 1) * C, ws.counters + 1, ws.flag_b, -(bs - 1 - bs / 2), W,
                         ws.counters + 2, ws.run_rows, 1.0, kFlagRatio / C, ws.W1, 0, 0};
  int st;
  if (fused) {
    // fused call: `in0` is the LCN OUTPUT buffer -- the streaming kernel (lcn_stream.hip) writes it and the frames'
    // planes from the raw frames in one launch; the pattern's half (if not prepared) keeps the tiled kernel
    const StatPlanes sp = {ws.ac, ws.m0, ws.v0, ws.Wp, 4, ws.counters, ws.flag_a, -(float)(bs * bs), (float)kFlagRatio};
    st = lcn_stream_f32(fused->raw, const_cast<float*>(in0), fused->stds, frames, H, W, fused->eps, sp,
                        rank ? rank->work.counters : nullptr, rank ? rank->work.parts : 0, fused->exact, stream);
    if (st) return st;
    if (!pattern_prepared) {
      PrepassJob none = ja;
      none.nimg = 0;
      st = launch_prepass(none, jb, H, W, bs, nullptr, stream);
    }
  } else {
    st = launch_prepass(ja, jb, H, W, bs, rank ? &rank->work : nullptr, stream);
  }
  if (st) return st;
  switch (bs) {
    case 3: st = launch_fast<3>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, ws, rank, stream); break;
    case 5: st = launch_fast<5>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, ws, rank, stream); break;
    case 7: st = launch_fast<7>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, ws, rank, stream); break;
    case 9: st = launch_fast<9>(in0, in1, in1_frame_stride, out, frames, C, H, W, D, ws, rank, stream); break;
    default: return CTD_ERR_UNSUPPORTED;
  }
  if (st || rank) return st;
  st = launch_fixup(in0, in1, in1_frame_stride, out, frames, C, H, W, D, bs, ws, per_frame, nullptr, nullptr,
                    (unsigned*)workspace, stream);
  if (st == CTD_OK && pattern_prepared) CTD_HIP_TRY(hipMemsetAsync(ws.counters, 0, 4, stream));   // (see above: [0] stays zero between calls)
  return st;
}

// Second half of a ranked call, after the all-D kernel: fix-up of the listed windows (volume patch when there is one,
// run values) with every recomputed score held against the `best` / `idx` of its pixel.
int ncc_fast_fixup_ranked(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int H, int W,
                          int D, int bs, void* workspace, const RankPlan& rank, const float* best, hipStream_t stream) {
  const bool per_frame = in1_frame_stride != 0;
  FastWorkspace ws = fast_workspace(workspace, frames, 1, H, W, D, per_frame);
  return launch_fixup(in0, in1, in1_frame_stride, out, frames, 1, H, W, D, bs, ws, per_frame, &rank, best, nullptr, stream);
}

// ------------------------------------------------------------------------------------
// Separable block SAD / MSE cost volume (block 9):
//     cost[f][d][h][x] = 1/81 * sum over the 9 x 9 taps of g(P[r][clamp(c - d)] - I[r][c]),  r = clamp(h + dy), c = clamp(x + dx)
// (the tap column is clamped BEFORE the shift, ext.h:231-243 composed with P_d[h][x] = P[h][clamp(x - d)]), i.e. the
// replicate-border 9 x 9 box filter of the per-pixel plane q_d[r][c] = g(P[r][clamp(c - d)] - I[r][c]), g = |.| or (.)^2.
// The all-D kernel in its kASad / kAMse mode filters that plane exactly like the NCC products: 3+3+3 vertical sums in
// registers, the horizontal 9-sum by DPP, one subtract instead of 81 per output.  Its operand planes are plain padded
// copies: frames with 4 replicate columns either side, the pattern per UNCLAMPED column x = c - d with the replicate
// border baked in (cost_planes_kernel).  One difference to the NCC border rule: right of the image the NCC product
// column w0 > W-1 pairs a[W-1] with b[w0 - d], here it must be a COPY of column W-1 (clamp before the shift) -- so the
// halo right of the last column tile takes the pattern sample of column W-1 (loader), and for an image that ends INSIDE a
// tile (W % 256 != 0) the outputs of its last four columns are recomputed by cost_border_kernel (taps summed directly).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cost_planes_kernel(const float* __restrict__ im, const float* __restrict__ pat,
                                                          long pat_frame_stride, float* __restrict__ ac,
                                                          float* __restrict__ bc, int frames, int n_pat, int H, int W, int Wp,
                                                          int W1, int xoff) {
  const long na = (long)frames * H * Wp, nb = (long)n_pat * H * W1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += (long)gridDim.x * blockDim.x) {
    if (i < na) {
      const int c = (int)(i % Wp) - 4;
      const long fh = i / Wp;
      ac[i] = im[fh * W + clampi(c, 0, W - 1)];
    } else {
      const long k = i - na;
      const int x = (int)(k % W1) - xoff;
      const long ph = k / W1;                                      // pattern image * H + row
      const long pimg = ph / H, h = ph - pimg * H;
      bc[k] = pat[pimg * pat_frame_stride + h * W + clampi(x, 0, W - 1)];
    }
  }
}

// outputs (f, d, h, x) for the last four image columns x = W-4 .. W-1: thread per (f, d, h), the taps of its 9 x 8
// neighbourhood summed in the reference's composition (tap column clamped, then shifted and clamped again)
template <int TYPE>
__global__ __launch_bounds__(256) void cost_border_kernel(const float* __restrict__ im, const float* __restrict__ pat,
                                                          long pat_frame_stride, float* __restrict__ cost, int frames, int H,
                                                          int W, int D) {
  const long n = (long)frames * D * H;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  // (d fastest: neighbouring lanes read neighbouring pattern samples and the same image sample)
  const int d = (int)(t % D), h = (int)((t / D) % H), f = (int)(t / ((long)H * D));
  const float* I = im + (long)f * H * W;
  const float* P = pat + (long)f * pat_frame_stride;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int dy = -4; dy <= 4; ++dy) {
    const int r = clampi(h + dy, 0, H - 1);
    float q[12];                                                   // q_d at columns W-8 .. W+3 (the last four: copies of W-1)
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int c = clampi(W - 8 + k, 0, W - 1);
      const float df = P[(long)r * W + clampi(c - d, 0, W - 1)] - I[(long)r * W + c];
      q[k] = TYPE == 0 ? df * df : fabsf(df);
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {                                   // output column W-4+o: q index 4+o, window o .. o+8
      float s9 = 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) s9 += q[o + k];
      acc[o] += s9;
    }
  }
  float* out = cost + (((long)f * D + d) * H + h) * W + (W - 4);
#pragma unroll
  for (int o = 0; o < 4; ++o)
    if (W - 4 + o >= 0) out[o] = acc[o] * (1.f / 81.f);
}

struct CostPlanes {
  int Wp, W1, xoff;
  size_t off_b, bytes;
};
static CostPlanes cost_planes(int frames, int H, int W, int D, bool per_frame_pattern) {
  CostPlanes cp;
  const int Dpad = (D + kFDG - 1) / kFDG * kFDG + 32;              // (as fast_workspace: a last pass stages columns past D)
  cp.xoff = Dpad + 3;
  cp.W1 = (int)align_up((size_t)(W + 4 + cp.xoff), 4);
  cp.Wp = (int)align_up((size_t)(W + 8), 4);
  cp.off_b = align_up((size_t)frames * H * cp.Wp * sizeof(float), 256);
  cp.bytes = cp.off_b + align_up((size_t)(per_frame_pattern ? frames : 1) * H * cp.W1 * sizeof(float), 256);
  return cp;
}

bool costvol_sep_supported(int H, int W, int D, int bs, int type) {
  return bs == 9 && (type == 0 || type == 1) && W % 4 == 0 && W >= 8 && H >= 1 && D <= 512;
}

size_t costvol_sep_workspace_bytes(int frames, int H, int W, int D, bool per_frame_pattern) {
  return cost_planes(frames, H, W, D, per_frame_pattern).bytes;
}

int costvol_sep_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W, int D,
                    int type, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  const bool per_frame = pat_frame_stride != 0;
  const CostPlanes cp = cost_planes(frames, H, W, D, per_frame);
  if (!workspace || workspace_bytes < cp.bytes || ((uintptr_t)workspace & 15)) return CTD_ERR_WORKSPACE;
  if (((uintptr_t)cost) % 16 != 0) return CTD_ERR_UNSUPPORTED;
  float* ac = (float*)workspace;
  float* bc = (float*)((char*)workspace + cp.off_b);
  const int n_pat = per_frame ? frames : 1;
  hipLaunchKernelGGL(cost_planes_kernel, dim3(1024), dim3(256), 0, stream, im, pat, pat_frame_stride, ac, bc, frames, n_pat, H, W,
                     cp.Wp, cp.W1, cp.xoff);
  CTD_LAUNCH_CHECK();
  const AlldPlan ap = alld_plan(frames, H, W, D, false);
  const int n_items = ceil_div(W, 256) * ap.bands * frames * ap.n_psplit;
  size_t lds = sizeof(float) * kABufs * ap.chunk_rows * kAPack;
#ifdef CTD_STAMPS
  lds += sizeof(unsigned) * kStampWords;
#endif
  dim3 grid(n_items), block(64 * (kAWaves + 1));
  auto kern = type == 1 ? (ap.chunk_rows == 3 ? ncc_fast_alld_kernel<kAStore | kASad, 3> : ncc_fast_alld_kernel<kAStore | kASad, 2>)
                        : (ap.chunk_rows == 3 ? ncc_fast_alld_kernel<kAStore | kAMse, 3> : ncc_fast_alld_kernel<kAStore | kAMse, 2>);
  CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long st1_stride = per_frame ? (long)H * cp.W1 : 0;
  hipLaunchKernelGGL(kern, grid, block, lds, stream, (const float*)ac, (const float*)nullptr, (const float*)nullptr,
                     (const float*)bc, (const float*)nullptr, (const float*)nullptr, st1_stride, cost, (int64_t*)nullptr,
                     (float*)nullptr, (unsigned char*)nullptr, WorkList{}, -1.f, frames, n_items, H, W, D, ap.band_rows, ap.n_pass,
                     ap.dgs, cp.Wp, cp.W1, cp.xoff, ap.n_psplit);
  CTD_LAUNCH_CHECK();
  if (W % 256 == 0) return CTD_OK;                                // (the loader's halo rule covers the right border)
  const long nb = (long)frames * D * H;
  if (type == 0)
    hipLaunchKernelGGL(cost_border_kernel<0>, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, stream, im, pat, pat_frame_stride, cost, frames, H, W, D);
  else
    hipLaunchKernelGGL(cost_border_kernel<1>, dim3((unsigned)ceil_div(nb, 256)), dim3(256), 0, stream, im, pat, pat_frame_stride, cost, frames, H, W, D);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd

#ifdef CTD_STAMPS
extern "C" int ctd_debug_read_stamps(void* dst, size_t bytes) {
  const size_t have = sizeof(unsigned) * ctd::kStampWgs * ctd::kStampWords;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ctd::g_stamps), bytes < have ? bytes : have);
}
extern "C" int ctd_debug_stamp_layout(int* words_per_wg, int* chunks, int* pass) {
  *words_per_wg = ctd::kStampWords; *chunks = ctd::kStampChunks; *pass = CTD_STAMP_PASS;
  return 0;
}
#endif
