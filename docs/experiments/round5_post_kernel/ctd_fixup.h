// ctd_fixup.h -- the fix-up of the fast NCC path's listed windows, as device code shared by its two users: the unranked
// call's ncc_fixup_kernel (ncc_fast.hip) and the ranked call's post kernel (argmax_rerank.hip, POST).
//
// Error model of the fast kernels (tools/err_vs_factor.py): cov = S_ab - n*ma*mb is formed from values centred by one
// constant per image, so |fast - exact| <~ c * 2^-24 * sqrt(Fa * Fb) with F = 1 + n*(window mean - centring)^2 / (sum of
// squared deviations) per window.  The pre-pass lists every window with F > sqrt(8) (and the nearly flat ones) and stores 0
// as its reciprocal deviation; here every output a listed window takes part in is recomputed in the reference's operation
// order (bit-identical to CTD_NCC_EXACT).  One wavefront per item, lane <-> disparity; the window itself (FIX, bs x bs)
// and the rows of the other image it meets over all disparities (SPAN, bs x (bs + D - 1)) are staged in LDS.
//   frame window  (f, h, w): outputs (f, d, h, w);        SPAN = pattern columns w-half-(D-1) .. w+half
//   pattern window (p, h, x): outputs (f, d, h, x + d), 0 <= x + d < W, every frame f that uses p;
//                             SPAN = frame columns x-half .. x+(D-1)+half.  x = -(bs-1-half) stands for all
//                             fully clamped windows x <= -(bs-1-half): lane d's value belongs to the whole run
//                             d' >= d of pixel w = x + d (ext.h:152-154 makes the run constant).
// The NCC is symmetric in the two windows (dot and sigma0*sigma1 commute exactly), so one staging layout serves both.
//
// POST (ranked call, after the all-D kernel): the in-kernel ranking saw the placeholder score 0 instead of these values, so
// every recomputed one is held against the pixel's ranked best.  Clearly above it: straight into the pixel's int64 index
// word as the patch key (ordered score << 32 | ~d) by a 64-bit atomic maximum (plain indices have a zero high word; the
// decode pass of the tail kernel turns keys back into indices).  Inside the margin of the best, or a placeholder that came
// out on top: the pixel is a case for the exact re-scoring, and -- round 5 -- the wavefront that claims it settles it ITSELF
// once its item is done (resolve_pixel treats every listed entry of the pixel's column as a candidate and so depends on
// nobody's patches): the re-scoring runs in THIS launch, beside the fix-up, instead of in a dependent one behind it.
// (Tried in round 5 and dropped, profiles/round5_post_kernel_ab.txt: settling the clear winners the same way -- listed windows
// are nearly flat, their exact scores are noise and beat the ranked best tens of thousands of times per call: 295 us; the
// item's own wavefront spreading a listed run over its 124 planes -- one plane row per store instruction, a new page each:
// +135 us.)
#pragma once
#include "ctd_common.h"
#include "ctd_ncc_point.h"
#include "ctd_rank.h"
#include "ctd_resolve.h"
#include "ctd_tail.h"

namespace ctd {

typedef float fix_f32x4 __attribute__((ext_vector_type(4)));

// A pattern window shared by all frames (single channel) is one item per group of kFixFrames frames: its own side
// (window, mean, deviations) is staged once, the frames' rows follow one another with the next frame's rows already
// on their way (register prefetch) -- the pass is latency-bound, a lone wavefront per item.
constexpr int kFixFrames = 2;
constexpr int kFixSpanRegs = 20;       // prefetched SPAN elements per lane (bs * (bs + D - 1) <= 64 * 20)

// per-wave LDS of an item: FIX raw / divided by n / minus its mean, SPAN rows raw / divided by n
__host__ __device__ inline size_t fixup_wave_floats(int bs, int D) {
  return 3 * (size_t)bs * bs + 2 * (size_t)bs * (bs + D - 1);
}

struct FixPost {                       // what the ranked post kernel's items need beyond the unranked fix-up's
  const float* best;                   // the all-D kernel's best scores
  int64_t* idx;                        // ... and indices (rewritten for the pixels settled here)
  float* best_out;
  unsigned* flags;                     // work-list flag bytes: a pixel is settled by exactly one wavefront
  float rank_eps;
  ListedPlanes lp;
};

// What an item of the post kernel hands to the ONE place that settles pixels (the caller's loop: a second inlined copy of
// resolve_pixel per call site made the kernel 140 KB of code and instruction-fetch bound): `cmask` bit s of a lane = the
// lane's pixel of slot s is to be settled; the pixel of (slot s, lane l) is
//   mode 0: base                                    (a listed frame window's pixel; lane 0, slot 0)
//   mode 1: base + (s / 8) * HW + ((s % 8) / 2) * 128 + 64 * (s % 2) + l     (grouped item: frame of the item, round, t)
//   mode 2: base + s * 64 + l                                              (generic item: round)
struct FixClaims {
  unsigned cmask;
  int n_slots, mode;
  long base;
};
__device__ inline long fix_claim_pixel(const FixClaims& c, int s, int lane, long HW) {
  if (c.mode == 0) return c.base;
  if (c.mode == 1) return c.base + (long)(s / 8) * HW + ((s % 8) / 2) * 128 + 64 * (s % 2) + lane;
  return c.base + (long)s * 64 + lane;
}

// Loops over the window rows stay rolled (a fully unrolled body is ~40 KB of straight-line code that every
// wavefront executes once -- instruction-fetch bound); BS > 0 unrolls the inner tap loop only.
template <int BS, bool POST, bool VOL>
__device__ __forceinline__ unsigned fixup_grouped_item(const float* __restrict__ in0, const float* __restrict__ in1,
                                                       long in1_frame_stride, float* __restrict__ out,
                                                       float* __restrict__ run_vals, const FixPost& post, float* lds, int f_lo,
                                                       int f_hi, int h, int col, bool run_item, int H, int W, int D, int bs_rt,
                                                       int lane) {
  const int bs = BS > 0 ? BS : bs_rt;
  const int half = bs / 2, span = bs + D - 1, taps = bs * bs;
  float* sF = lds;
  float* sFq = sF + taps;
  float* sFv = sFq + taps;
  float* sS = sFv + taps;
  float* sSq = sS + bs * span;
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
  unsigned cmask = 0u;                                           // POST: bit (frame of the item * 8 + round * 2 + t)
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
        mb[t] = (POST && bad[t]) ? post.best[((long)f * H + h) * W + w] : 0.f;
        won[t] = POST && bad[t] && post.idx[((long)f * H + h) * W + w] == (int64_t)dd[t];   // the placeholder came out on top
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
        if (POST) {
          const long pixc = ((long)f * H + h) * W + w;
          // clearly above everything the ranking saw: the patch key; inside the margin of the best (ctd_rank.h:
          // rank_margin), or the placeholder itself came out on top with no such lead: settled by the exact re-scoring, by
          // whoever claims the pixel first
          const bool clear = bad[t] && val[t] > mb[t] + rank_margin(post.rank_eps, mb[t]);
          if (clear) atomicMax((unsigned long long*)post.idx + pixc, patch_key(val[t], d));
          const bool contender = bad[t] && !clear && (won[t] || !(val[t] < mb[t] - rank_margin(post.rank_eps, mb[t])));
          if (contender && worklist_claim(post.flags, pixc)) cmask |= 1u << ((f - f_lo) * 8 + r * 2 + t);
        }
      }
    }
  }
  (void)in1_frame_stride;
  return cmask;                                                  // slots: (frame of the item) * 8 + round * 2 + t
}

// How the call's fix-up list is cut into items, one per wavefront: n_a listed frame windows (list_a), n_b listed pattern
// windows (list_b; a shared pattern window meets every frame, kFixFrames frames per item where the grouped path applies).
struct FixItems {
  unsigned n_a, n_b, groups, n_items;
  bool grouped;
};
template <bool POST>
__device__ inline FixItems fixup_item_count(unsigned n_a, unsigned n_b, long in1_frame_stride, int frames, int C, int bs, int D) {
  FixItems fi;
  const unsigned per_b = in1_frame_stride == 0 ? (unsigned)frames : 1u;   // a shared pattern window meets every frame
  // single channel, shared pattern, SPAN small enough for the prefetch registers: kFixFrames frames per item
  fi.grouped = per_b > 1u && C == 1 && bs * (bs + D - 1) <= 64 * kFixSpanRegs && (!POST || D <= 512);
  fi.groups = fi.grouped ? (per_b + kFixFrames - 1) / kFixFrames : per_b;
  fi.n_a = n_a;
  fi.n_b = n_b;
  fi.n_items = n_a + n_b * fi.groups;
  return fi;
}

// Item `item` of the list.  POST: returns the pixels its wavefront has claimed for the exact re-scoring (the caller
// settles them, FixClaims).
template <int BS, bool POST, bool VOL>
__device__ __forceinline__ FixClaims fixup_one_item(const float* __restrict__ in0, const float* __restrict__ in1,
                                                    long in1_frame_stride, float* __restrict__ out, const FixItems& fi,
                                                    const unsigned long long* __restrict__ list_a,
                                                    const unsigned long long* __restrict__ list_b,
                                                    float* __restrict__ run_vals, const FixPost& post, float* lds,
                                                    unsigned item, int frames, int C, int H, int W, int D, int bs_rt) {
  const int bs = BS > 0 ? BS : bs_rt;
  const int lane = threadIdx.x & 63;
  const int half = bs / 2, span = bs + D - 1, taps = bs * bs;
  const float n = (float)taps;
  float* sF = lds;
  float* sFq = sF + taps;
  float* sFv = sFq + taps;
  float* sS = sFv + taps;
  float* sSq = sS + bs * span;
  const long HW = (long)H * W;
  const unsigned per_b = in1_frame_stride == 0 ? (unsigned)frames : 1u;
  const unsigned n_a = fi.n_a, groups = fi.groups;
  const bool grouped = fi.grouped;
  const int rounds = (D + 63) / 64;
  FixClaims claims = {0u, 0, 0, 0};
  {
    const bool is_a = item < n_a;
    const unsigned jb = is_a ? 0u : (item - n_a) / groups;
    const unsigned long long e = is_a ? list_a[item] : list_b[jb];
    const int z = (int)(e >> 40), h = (int)((e >> 20) & 0xFFFFF), col = (int)(e & 0xFFFFF) - 0x80000;
    const bool run_item = !is_a && col == -(bs - 1 - half);
    if (POST && is_a) {
      // a listed frame window: every score of its pixel is a placeholder -- the exact re-scoring of that one pixel
      // recomputes (and, with a volume, writes) its whole column
      const long pix = ((long)z * H + h) * W + col;              // (ranked calls are single channel: z = frame)
      claims.cmask = (lane == 0 && worklist_claim(post.flags, pix)) ? 1u : 0u;
      claims.n_slots = 1;
      claims.mode = 0;
      claims.base = pix;
      return claims;
    }
    if (grouped && !is_a) {
      const int f_lo = (int)((item - n_a) - jb * groups) * kFixFrames;
      claims.cmask = fixup_grouped_item<BS, POST, VOL>(in0, in1, in1_frame_stride, out, run_vals, post, lds, f_lo,
                                                        min(frames, f_lo + kFixFrames), h, col, run_item, H, W, D, bs, lane);
      claims.n_slots = kFixFrames * 8;
      claims.mode = 1;
      claims.base = ((long)f_lo * H + h) * W + col;
      return claims;
    }
    const int f = (is_a || per_b == 1u) ? z / C : (int)((item - n_a) - jb * groups);
    const float* fix_img = is_a ? in0 + (long)f * C * HW : in1 + (long)f * in1_frame_stride;
    const float* span_img = is_a ? in1 + (long)f * in1_frame_stride : in0 + (long)f * C * HW;
    const int span_col0 = is_a ? col - half - (D - 1) : col - half;
    int staged_c = -1;
    float mu_f = 0.f, s_f = 0.f;
    unsigned cmask = 0u;                                         // POST: bit r = this lane's pixel of round r
    for (int r = 0; r < rounds; ++r) {
      const int d = r * 64 + lane;
      const int w = is_a ? col : col + d;
      // every output of a listed window is recomputed (the fast kernels wrote the placeholder there)
      const bool bad = d < D && w >= 0 && w < W;
      float val = 0.f;
      const float mbest = (POST && bad) ? post.best[((long)f * H + h) * W + w] : 0.f;   // ranked calls: needed at the end
      const bool won = POST && bad && post.idx[((long)f * H + h) * W + w] == (int64_t)d;   // the placeholder came out on top
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
      if (POST) {
        const long pixc = ((long)f * H + h) * W + w;
        const bool clear = bad && val > mbest + rank_margin(post.rank_eps, mbest);      // (see the grouped path)
        if (clear) atomicMax((unsigned long long*)post.idx + pixc, patch_key(val, d));
        const bool contender = bad && !clear && (won || !(val < mbest - rank_margin(post.rank_eps, mbest)));
        if (contender && worklist_claim(post.flags, pixc)) cmask |= 1u << r;
      }
    }
    if (POST) {
      claims.cmask = cmask;
      claims.n_slots = rounds;
      claims.mode = 2;
      claims.base = ((long)f * H + h) * W + col;
    }
  }
  return claims;
}

}  // namespace ctd
