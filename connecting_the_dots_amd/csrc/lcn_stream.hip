// lcn_stream.hip -- LCN of the frames and the window statistics of the matcher's pre-pass in ONE streaming kernel.
//
// Replaces, for the frames of a fused call (ctd_lcn_xcorrvol_argmax_f32), the two LDS-tiled launches lcn_kernel /
// lcn_fast_kernel (lcn.hip; LCN.tforward, model/networks.py:507-533) and ncc_prepass_kernel (ncc_fast.hip; the window
// mean / deviation of XCorrVolFunctor, torchext/ext/ext.h:143-181, hoisted out of the disparity loop).  Both of those
// are tiles with two barriers and a halo of 10 / 8 pixels each way, and both live on occupancy (docs/history.md);
// here a workgroup owns a strip of 256 columns (lane = 4 adjacent columns: 16-byte loads and stores) and marches down
// a band of rows with everything it needs of earlier rows in registers and LDS rings -- no tile phases, no second
// launch, and the LCN output feeds the 9 x 9 statistics without a trip through memory.  Its three wavefronts are two
// pipeline stages (LCN | statistics, one row apart) and a LOADER that streams the raw rows global -> LDS (LDS-DMA,
// kLsPF rows ahead) and is the only one that ever waits for a load: on gfx950 loads and stores retire in order on one
// counter, so a wavefront that did both would wait for its newest stores every row (the same split as the volume
// kernels of ncc_fast.hip); the three meet at one s_barrier per row.
//   raw row u  --vertical 11-row sums V (sliding, per column)-->  horizontal 11-column sums (DPP wave shifts)
//              --> y = (x - avg) / std of row u - 5 --> stored (LCN output + the matcher's padded copy)
//              --> vertical 9-row sums of y, y^2 as 3 + 3 + 3 (no running sum: every window is a fresh <= 4-level sum)
//              --> horizontal 9-column sums (DPP) --> mean / reciprocal-deviation planes of row u - 9, listing rule.
// Of a strip's 64 lanes the outer three each side are halo (5 + 4 columns, rounded to whole lanes): 232 valid columns.
//
// Numerics.  ACC = double ("exact"): the LCN box sums are formed in f64 -- exact, whatever the order, unless one
// window spans more than 2^29 in magnitude -- and rounded once; the elementwise tail is the reference's f32 operation
// order: the same bits as lcn_kernel / the oracle.  ACC = float ("fast"): f32 sums of samples centred by one constant
// per wavefront, v_rcp / v_sqrt in the tail: tolerance level, and only where E[(x - c)^2] is not large against the
// window's variance (see DESIGN: an f32 one-pass variance cannot be better than that).
// The pre-pass statistics follow ncc_prepass_kernel<9, true> (f32 sums of the LCN output, which is centred by
// construction: the image constant is 0 here) with the same listing rule.
#include <cstdlib>

#include "ctd_internal.h"
#include "ctd_prepass.h"
#include "ctd_wave.h"

namespace ctd {

typedef float ls_f32x4 __attribute__((ext_vector_type(4)));

#ifndef CTD_LS_VALID
#define CTD_LS_VALID 232
#endif
constexpr int kLsValid = CTD_LS_VALID;             // output columns of a strip (58 lanes)
constexpr int kLsHalo = (256 - kLsValid) / 2;      // halo columns either side (3 lanes: 5 + 4 columns rounded to whole lanes)
static_assert(kLsValid % 4 == 0 && kLsHalo % 4 == 0 && kLsHalo >= 12, "strip geometry");
constexpr int kLsPF = 4;           // raw rows the loader runs ahead of the consumer
constexpr int kLsXRing = 16;       // raw rows in LDS: the 11 of the window + the one the sliding sum drops + kLsPF in flight
constexpr int kLsYRing = 7;        // triple sums of y rows kept in LDS: T3(j), T3(j - 3), T3(j - 6)
constexpr int kLsMaxBand = 64;     // rows of a band at most (bounds the drift of the fast variant's f32 sliding 11-row sums)
constexpr int kLsR = 5;            // LCN radius
constexpr int kLsHalf = 4;         // half block of the matcher (block 9)

__device__ inline int ls_reflect(int i, int n) {
  if (i < 0) i = -i;
  if (i > n - 1) i = 2 * (n - 1) - i;
  return i;
}

// lane l gets the value of lane l - 1 (l + 1); the first (last) lane gets 0
__device__ inline double ls_prev(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double ls_next(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// s[i] = sum of the 11 columns around column i of the lane's quad a[0..3] (columns 4l .. 4l + 3): every sum is a fresh
// tree of partial sums of the lane and of its two neighbours either side -- 15 additions, 10 of them on a shifted operand.
__device__ __forceinline__ void ls_hsum11(const double (&a)[4], double (&s)[4]) {
  const double E = a[0] + a[1], Tq = E + a[2], Q = Tq + a[3], R1 = a[2] + a[3], R = a[1] + R1;
  const double U = ls_prev(a[3]) + Q;         // a3[l-1] + Q[l]
  const double Z = Q + ls_next(a[0]);         // Q[l] + a0[l+1]
  s[0] = (ls_prev(U) + Q) + ls_next(E);       // 4l-5 .. 4l+5
  s[1] = (ls_prev(Q) + Q) + ls_next(Tq);      // 4l-4 .. 4l+6
  s[2] = (ls_prev(R) + Q) + ls_next(Q);       // 4l-3 .. 4l+7
  s[3] = (ls_prev(R1) + Q) + ls_next(Z);      // 4l-2 .. 4l+8
}
// f32: the shifted operands ride on the additions (v_add_f32_dpp, window_combine4 of ctd_wave.h)
__device__ __forceinline__ void ls_hsum11(const float (&a)[4], float (&s)[4]) {
  const float E = a[0] + a[1], Tq = E + a[2], Q = Tq + a[3], R1 = a[2] + a[3], R = a[1] + R1;
  float U, Z;
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
               "v_add_f32_dpp %1, %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "=&v"(U), "=&v"(Z) : "v"(a[3]), "v"(a[0]), "v"(Q));
  const float sp[4] = {U, Q, R, R1}, pn[4] = {E, Tq, Q, Z};
  window_combine4(sp, Q, pn, s);
}
// the same for 9 columns: 4l-4+i .. 4l+4+i -- 13 additions, 8 of them on a shifted operand
__device__ __forceinline__ void ls_hsum9(const float (&a)[4], float (&s)[4]) {
  const float E = a[0] + a[1], Tq = E + a[2], Q = Tq + a[3], R1 = a[2] + a[3], R = a[1] + R1;
  const float sp[4] = {Q, R, R1, a[3]}, pn[4] = {a[0], E, Tq, Q};
  window_combine4(sp, Q, pn, s);
}

struct LcnStreamArgs {
  const float* x;             // raw frames [N][H][W]
  float *y, *stds;            // LCN outputs [N][H][W]
  StatPlanes sp;              // the matcher's frame-side planes (ctd_prepass.h)
  unsigned* clear_counters;   // work-list counters of a ranked call, cleared here (see ncc_prepass_kernel)
  int n_clear;
  int N, H, W, band_rows, n_bands, n_strips;
  float eps;
};

#ifdef CTD_LS_NOSTORE   // TIMING EXPERIMENT ONLY: the consumer's global stores go away
#define CTD_LS_ST(cond) if ((cond) && a.N < 0)
#else
#define CTD_LS_ST(cond) if (cond)
#endif

// Roles of a workgroup's three wavefronts (one (strip, band, frame) item per workgroup):
//   0  LCN stage: 11-row sums, 11-column sums, the elementwise tail; stores the LCN outputs and the padded copy and
//      hands the row (after the replicate rule) to wavefront 1 through LDS;
//   1  statistics stage, one row behind: 9-row sums, 9-column sums, mean / reciprocal deviation, listing;
//   2  loader of the raw rows.
// The kernel is bound by vector / scalar issue (about 400 instructions per full row of a strip, 18 warm-up rows per
// band; without its stores 28.7 us against 34.5): with both stages in one wavefront and 1152 workgroups it took 34.5 us,
// as two stages on two SIMDs with at most four workgroups per CU 31.4 us -- against 18.2 + 24.0 us for the two tiled
// kernels it replaces (profiles/round5_lcn_stream_ab.txt).
template <class ACC>
__global__ __launch_bounds__(192) void lcn_prepass_stream_kernel(LcnStreamArgs a) {
  constexpr bool EXACT = sizeof(ACC) == 8;
  constexpr int NR = 2 * kLsR + 1, NB = 2 * kLsHalf + 1;         // 11 rows of the LCN window, 9 of the matcher's
  static_assert(kLsXRing == 16 && kLsXRing >= NR + 1 + kLsPF, "ring: window + dropped row + rows in flight, slot = feed & 15");
  __shared__ float xring[kLsXRing][256];
  __shared__ float yring[kLsYRing][2][256];
  __shared__ float yhand[2][256];                                // the LCN stage's newest rows, for the statistics stage
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (a.clear_counters && blockIdx.x == 0 && (int)threadIdx.x < a.n_clear) a.clear_counters[threadIdx.x * kWorkListStride] = 0u;
  int item = (int)blockIdx.x;
  const int strip = item % a.n_strips;
  item /= a.n_strips;
  const int band = item % a.n_bands, f = item / a.n_bands;
  const int H = a.H, W = a.W;
  const int h_lo = band * a.band_rows, h_hi = min(h_lo + a.band_rows, H);
  if (h_lo >= h_hi) return;
  const int strip_lo = strip * kLsValid - kLsHalo;               // column of the strip's position 0
  const int c0 = strip_lo + 4 * lane;                            // first column of the lane (may lie outside the image)
  const bool in_img = c0 >= 0 && c0 < W;                         // W % 4 == 0: the whole quad is inside or outside
  const bool own = lane >= kLsHalo / 4 && lane < 64 - kLsHalo / 4 && in_img;            // this lane's outputs belong to the strip
  const int t0 = h_lo - kLsHalf, t1 = h_hi - 1 + kLsHalf;        // rows of y the band's windows take (unclamped)
  const int ry_first = max(t0, 0), ry_last = min(t1, H - 1);
  const int u0 = ry_first - kLsR;                                // first raw row fed (unreflected)
  const int n_raw = ry_last - ry_first + NR;
  // Barriers: number k (0 <= k < n_raw) says "raw feed k has landed", and, for k > NR - 1, "LCN row k - NR is in
  // yhand[(k - NR) & 1]"; one more at the end hands over the last LCN row.  Every role passes n_raw + 1 of them.

  if (role == 2) {
    // ------------------------------------------------ loader ------------------------------------------------
    // Feed k goes to ring slot k & 15.  Before barrier k everything but the kLsPF - 1 newest DMAs has landed (feed k
    // among it); DMA k + kLsPF is issued behind barrier k: its slot held feed k + kLsPF - 16 <= k - 12, which the LCN
    // stage read last in iteration k - 1.  Past the last row the loader keeps issuing (the last row again, into slots
    // nobody reads any more) so that the count in flight stays the same to the end.
    const float* xf = a.x + (long)f * H * W + clampi(c0, 0, W - 4);
    __builtin_amdgcn_s_setprio(3);
    auto issue = [&](int k) {
      const int row = ls_reflect(u0 + min(k, n_raw - 1), H);
      dma_quad(xf + (long)row * W, &xring[k & (kLsXRing - 1)][0]);
    };
#pragma unroll
    for (int i = 0; i < kLsPF; ++i) issue(i);
    for (int k = 0; k < n_raw; ++k) {
      wait_vmcnt<kLsPF - 1>();
      wg_barrier();
      issue(k + kLsPF);
    }
    wait_vmcnt<0>();                                             // nothing of this workgroup's may land in LDS after it has gone
    wg_barrier();
    return;
  }

  if (role == 0) {
    // ------------------------------------------------ LCN stage ------------------------------------------------
    // the padded copy also takes the replicate columns -4 .. -1 and W .. W + 3 (their quads hold copies after the border rule)
    const bool own_pad = own || (lane >= kLsHalo / 4 - 1 && lane <= 64 - kLsHalo / 4 && (c0 == -4 || c0 == W));
    // reflect rule of the raw columns (ReflectionPad2d, networks.py:524): strip positions the outside quads copy from
    int src[4];
    bool foreign = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      src[i] = clampi(ls_reflect(c0 + i, W) - strip_lo, 0, 255);
      foreign |= src[i] != 4 * lane + i;
    }
    const bool patch = __any(foreign);                           // wave-uniform: the strip touches an image border
    // replicate rule of the LCN output's columns (the matcher clamps its taps, ext.h:147-150)
    const int lane_l = (0 - strip_lo) >> 2, lane_r = (W - 4 - strip_lo) >> 2;   // lanes holding columns 0 and W - 1 (if in this strip)
    auto raw_row = [&](int k) {                                  // feed k, border rule applied (and kept in the ring)
      float* row = &xring[k & (kLsXRing - 1)][0];
      ls_f32x4 v = *(const ls_f32x4*)(row + 4 * lane);
      if (patch) {                                               // wave-uniform
        if (foreign) {
          v = ls_f32x4{row[src[0]], row[src[1]], row[src[2]], row[src[3]]};
          *(ls_f32x4*)(row + 4 * lane) = v;
        }
      }
      return v;
    };
    ACC V1[4], V2[4];                                            // sums over the 11 newest raw rows, per column
#pragma unroll
    for (int i = 0; i < 4; ++i) V1[i] = V2[i] = (ACC)0;
    // fast variant: samples centred by one constant per wavefront, the in-image value of the first row closest to zero:
    // |x - c| <= |x| for every sample of that row, and exactly 0 on a zero background
    float cen = 0.f;
    for (int k = 0; k < NR; ++k) {                               // the first 11 rows only accumulate
      wait_lgkmcnt0();
      wg_barrier();                                              // feed k has landed
      const ls_f32x4 xn = raw_row(k);
      if (!EXACT && k == 0) {
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          mn = fminf(mn, in_img ? xn[i] : INFINITY);
          mx = fmaxf(mx, in_img ? xn[i] : -INFINITY);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          mn = fminf(mn, __shfl_xor(mn, o));
          mx = fmaxf(mx, __shfl_xor(mx, o));
        }
        cen = __builtin_amdgcn_fmed3f(0.f, mn, mx);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float vn = EXACT ? xn[i] : xn[i] - cen;
        const float qn = vn * vn;                                // data**2 is an f32 tensor (networks.py:528)
        V1[i] += (ACC)vn;
        V2[i] += (ACC)qn;
      }
    }
    // byte offsets of the lane's quad inside this frame's tensors / planes (32 bits: lcn_stream_supported; unsigned
    // offsets from the start of the frame's plane: the replicate quad left of column 0 sits at o_off - 4 >= 0, lanes
    // whose offset would be negative never store)
    const char* const yb = (const char*)(a.y + (long)f * H * W);
    const char* const sb = (const char*)(a.stds + (long)f * H * W);
    const char* const ib = (const char*)(a.sp.img + (long)f * H * a.sp.pitch);
    unsigned o_y = (unsigned)(((long)ry_first * W + c0) * 4);                                    // LCN row `cur`
    unsigned o_p = (unsigned)(((long)ry_first * a.sp.pitch + a.sp.o_off + c0) * 4);             // the same row of the padded copy
    for (int k = NR - 1;;) {
      // ---- LCN of row cur = ry_first + k - 10 (its own sample: the feed 5 steps ago)
      const int cur = ry_first + k - (NR - 1);
      const ls_f32x4 xc = *(const ls_f32x4*)&xring[(k - kLsR) & (kLsXRing - 1)][4 * lane];
      ACC S1[4], S2[4];
      ls_hsum11(V1, S1);
      ls_hsum11(V2, S2);
      float yq[4], sq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (EXACT) {
          // the reference's f32 tail (networks.py:529-532), as lcn_kernel evaluates it
          const float boxs = (float)S1[i], boxs2 = (float)S2[i];
          const float avgs = boxs / 121.f;
          const float var = boxs2 / 121.f - avgs * avgs + 1e-6f;
          const float sd = sqrtf(var) + a.eps;
          yq[i] = (xc[i] - avgs) / sd;
          sq[i] = sd;
        } else {
          const float avgs = S1[i] * (1.f / 121.f);
          const float var = fmaf(-avgs, avgs, fmaf(S2[i], 1.f / 121.f, 1e-6f));
          const float sd = __builtin_amdgcn_sqrtf(var) + a.eps;
          yq[i] = ((xc[i] - cen) - avgs) * __builtin_amdgcn_rcpf(sd);
          sq[i] = sd;
        }
      }
      const bool row_own = cur >= h_lo && cur < h_hi;            // wave-uniform: rows of the neighbouring bands are theirs
      CTD_LS_ST(row_own && own) {
        *(ls_f32x4*)(const_cast<char*>(yb) + o_y) = ls_f32x4{yq[0], yq[1], yq[2], yq[3]};
        *(ls_f32x4*)(const_cast<char*>(sb) + o_y) = ls_f32x4{sq[0], sq[1], sq[2], sq[3]};
      }
      // replicate rule: columns left of 0 / right of W - 1 take the LCN output of column 0 / W - 1
      if (patch) {
        if (lane_l >= 0 && lane_l < 64) {
          const float yl = __shfl(yq[0], lane_l);
          if (c0 < 0) yq[0] = yq[1] = yq[2] = yq[3] = yl;
        }
        if (lane_r >= 0 && lane_r < 64) {
          const float yr = __shfl(yq[3], lane_r);
          if (c0 >= W) yq[0] = yq[1] = yq[2] = yq[3] = yr;
        }
      }
      CTD_LS_ST(row_own && own_pad) *(ls_f32x4*)(const_cast<char*>(ib) + o_p) = ls_f32x4{yq[0], yq[1], yq[2], yq[3]};
      o_y += (unsigned)(W * 4);
      o_p += (unsigned)(a.sp.pitch * 4);
      *(ls_f32x4*)&yhand[(k - (NR - 1)) & 1][4 * lane] = ls_f32x4{yq[0], yq[1], yq[2], yq[3]};
      ++k;
      wait_lgkmcnt0();                                           // ring reads done, the hand-over row written: raw barrier
      wg_barrier();                                              // feed k has landed (k == n_raw: the last hand-over)
      if (k == n_raw) break;
      // ---- next raw row: add it, drop the row that leaves the window (the feed 11 steps ago)
      const ls_f32x4 xn = raw_row(k);
      const ls_f32x4 xo = *(const ls_f32x4*)&xring[(k - NR) & (kLsXRing - 1)][4 * lane];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float vn = EXACT ? xn[i] : xn[i] - cen, vo = EXACT ? xo[i] : xo[i] - cen;
        const float qn = vn * vn, qo = vo * vo;
        V1[i] = (V1[i] + (ACC)vn) - (ACC)vo;
        V2[i] = (V2[i] + (ACC)qn) - (ACC)qo;
      }
    }
    return;
  }

  // ------------------------------------------------ statistics stage ------------------------------------------------
  const char* const mb = (const char*)(a.sp.mean + (long)f * H * a.sp.pitch);
  const char* const db = (const char*)(a.sp.dev + (long)f * H * a.sp.pitch);
  unsigned o_s = (unsigned)(((long)h_lo * a.sp.pitch + a.sp.o_off + c0) * 4);                   // the statistics' output row
  float y1[4], q1[4], p2y[4], p2q[4];                            // y, y^2 of the last feed; their sums over the last two feeds
#pragma unroll
  for (int i = 0; i < 4; ++i) y1[i] = q1[i] = p2y[i] = p2q[i] = 0.f;
  int ys = 0, j = 0;                                             // ring slot and number of the next feed of the 9-row stage
  for (int k = 0; k < NR; ++k) wg_barrier();                     // nothing to do before the first LCN row exists
  for (int k = NR;; ++k) {
    wait_lgkmcnt0();
    wg_barrier();                                                // LCN row k - 11 is in yhand
    const int cur = ry_first + k - NR;
    const ls_f32x4 yv = *(const ls_f32x4*)&yhand[(k - NR) & 1][4 * lane];
    const float yq[4] = {yv[0], yv[1], yv[2], yv[3]};
    // ---- feeds of the 9-row stage: once, and once more for every window row clamped to this one
    int n_feed = 1;
    if (cur == 0) n_feed += max(0, -t0);
    if (cur == H - 1) n_feed += max(0, t1 - (H - 1));
    float q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = yq[i] * yq[i];
#pragma unroll 1
    for (int r = 0; r < n_feed; ++r) {
      // 9-row sums as 3 + 3 + 3, every one a fresh <= 4-level sum: T3(j) = y(j-2) + y(j-1) + y(j) goes to the ring, the
      // window is T3(j) + T3(j-3) + T3(j-6).  (No running sum: after a bright sample has left the window a running sum
      // keeps the rounding errors of ITS magnitude, and the statistics of a flat window behind it would carry them.)
      float t3y[4], t3q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t3y[i] = p2y[i] + yq[i];
        t3q[i] = p2q[i] + q[i];
        p2y[i] = y1[i] + yq[i];
        p2q[i] = q1[i] + q[i];
        y1[i] = yq[i];
        q1[i] = q[i];
      }
      const int s3 = ys >= 3 ? ys - 3 : ys + kLsYRing - 3, s6 = ys >= 6 ? ys - 6 : ys + kLsYRing - 6;
      const ls_f32x4 ay = *(const ls_f32x4*)&yring[s3][0][4 * lane], by = *(const ls_f32x4*)&yring[s6][0][4 * lane];
      const ls_f32x4 aq = *(const ls_f32x4*)&yring[s3][1][4 * lane], bq = *(const ls_f32x4*)&yring[s6][1][4 * lane];
      *(ls_f32x4*)&yring[ys][0][4 * lane] = ls_f32x4{t3y[0], t3y[1], t3y[2], t3y[3]};
      *(ls_f32x4*)&yring[ys][1][4 * lane] = ls_f32x4{t3q[0], t3q[1], t3q[2], t3q[3]};
      ys = ys == kLsYRing - 1 ? 0 : ys + 1;
      if (j >= NB - 1) {
        // statistics of output row h = t0 + j - 4 = h_lo + j - 8: the arithmetic of ncc_prepass_kernel<9, true> with the
        // image constant 0 (the LCN output is centred by construction)
        float W1[4], W2[4], s1[4], s2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          W1[i] = t3y[i] + (ay[i] + by[i]);
          W2[i] = t3q[i] + (aq[i] + bq[i]);
        }
        ls_hsum9(W1, s1);
        ls_hsum9(W2, s2);
        float om[4], od[4];
        bool listed[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float mean_c = s1[i] * (float)(1.0 / 81.0);
          const float nm2 = s1[i] * mean_c;                      // n * mean^2
          const float var = s2[i] - nm2;                         // sum of squared deviations (sigma of ext.h:180-181)
          // listing rule (ncc_prepass_kernel): deviation under the floor, or small against the mean's offset from the
          // centring constant (0 here, which also makes the flat-window clause a special case of the second)
          listed[i] = var < (float)(kDevFloor * kDevFloor) || nm2 > a.sp.flag_ratio * var;
          const float rdev = __builtin_amdgcn_rsqf(var > 0.f ? var : 1.f);
          om[i] = a.sp.mean_scale * mean_c;
          od[i] = listed[i] ? 0.f : rdev;
        }
        CTD_LS_ST(own) {
          *(ls_f32x4*)(const_cast<char*>(mb) + o_s) = ls_f32x4{om[0], om[1], om[2], om[3]};
          *(ls_f32x4*)(const_cast<char*>(db) + o_s) = ls_f32x4{od[0], od[1], od[2], od[3]};
          if (listed[0] | listed[1] | listed[2] | listed[3]) {
            const int h = t0 + j - kLsHalf;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (listed[i])
                a.sp.flag_list[atomicAdd(a.sp.n_flag, 1u)] = ((unsigned long long)f << 40) | ((unsigned long long)h << 20) |
                                                             (unsigned long long)(c0 + i + 0x80000);
          }
        }
        o_s += (unsigned)(a.sp.pitch * 4);
      }
      ++j;
    }
    if (k == n_raw) break;
  }
}

bool lcn_stream_supported(int H, int W, int radius, int bs) {
  return radius == kLsR && bs == 2 * kLsHalf + 1 && W % 4 == 0 && W >= 16 && H >= 2 * kLsR + 1 &&
         (double)H * (W + 8) * 4 < 4294967296.0;                // 32-bit byte offsets inside a frame's planes
}

int lcn_stream_f32(const float* x, float* y, float* stds, int N, int H, int W, float eps, const StatPlanes& sp,
                   unsigned* clear_counters, int n_clear, bool exact, hipStream_t stream) {
  if (!lcn_stream_supported(H, W, kLsR, 2 * kLsHalf + 1) || n_clear > 64) return CTD_ERR_UNSUPPORTED;
  LcnStreamArgs a;
  a.x = x; a.y = y; a.stds = stds; a.sp = sp; a.clear_counters = clear_counters; a.n_clear = n_clear;
  a.N = N; a.H = H; a.W = W; a.eps = eps;
  a.n_strips = ceil_div(W, kLsValid);
  // one workgroup (three wavefronts) per (strip, band, frame); every band pays 18 warm-up rows, and four workgroups per
  // CU are resident side by side on its four SIMDs: as many bands as give at most 4 x CUs workgroups (A/B at config 2,
  // rocprofv3, f32 sums: 768 workgroups 33.9 us, 864 32.5, 960 / 1008 31.4, 1152 38.0, 1280 38.2)
  int target = 1024, dev = 0, n_cu = 0;
  if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
    target = 4 * n_cu;
#ifdef CTD_LS_KNOBS   // (variant builds only: the workgroup-count sweep of tools/r5_prof_fused.sh)
  if (const char* e = getenv("CTD_LS_WAVES")) target = atoi(e);
#endif
  int n_bands = target / (N * a.n_strips);
  if (n_bands > H / 8) n_bands = H / 8;
  if (n_bands < 1) n_bands = 1;
  if (n_bands < ceil_div(H, kLsMaxBand)) n_bands = ceil_div(H, kLsMaxBand);
  a.band_rows = ceil_div(H, n_bands);
  a.n_bands = ceil_div(H, a.band_rows);
  const unsigned grid = (unsigned)(a.n_strips * a.n_bands * N);
  if (exact) hipLaunchKernelGGL(lcn_prepass_stream_kernel<double>, dim3(grid), dim3(192), 0, stream, a);
  else hipLaunchKernelGGL(lcn_prepass_stream_kernel<float>, dim3(grid), dim3(192), 0, stream, a);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

}  // namespace ctd
