// ctd_resolve.h -- exact re-scoring of one pixel by one wavefront (argmax_rerank.hip: the resolve pass of the plain ranked
// argmax and both roles of the ranked fast call's post kernel).
#pragma once
#include "ctd_common.h"
#include "ctd_ncc_point.h"

namespace ctd {

constexpr int kMaskWords = 8;            // at most 512 disparities per candidate mask

__device__ inline float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Which scores of the fast volume are PLACEHOLDERS (outputs of listed windows: the pre-pass stored a zero reciprocal
// deviation for them, ncc_fast.hip).  v1 == nullptr: none (the plain ranked argmax works on a fully patched volume).
// A ranked fast call's post kernel re-scores pixels while other wavefronts are still patching the volume's listed
// entries, so the re-scoring never trusts such an entry: every listed disparity of the pixel is a candidate by itself.
struct ListedPlanes {
  const float* v0;            // frames' reciprocal-deviation planes [frame][H][Wp], column c at c + 4
  const float* v1;            // pattern's [pattern image][H][W1], unclamped window-centre column x at x + xoff
  int Wp, W1, xoff, per_frame;
};

// One wavefront settles pixel `pj` (flat over frames): lane <-> disparity collects the candidate set -- VOL: scores
// within eps of the best fast score, the run of disparities whose window is clamped to column 0 counted once (lowest d),
// plus every listed disparity; !VOL (nothing materialised): every disparity up to the start of the clamped run.  A lone
// unlisted candidate is final; otherwise the frame window and the reachable pattern rows are staged in LDS and the
// candidates are re-scored in the reference's operation order (lane <-> candidate), lowest d among the best.
// Writes idx[pj] and best[pj] (the reference-order score when the pixel was re-scored, else the fast score); VOL: the
// re-scored values of LISTED candidates go into the volume as well (a frame-listed pixel gets its whole column here).
// `sA`: per-wave LDS, 2 * (bs * bs + bs * (bs + D - 1)) floats.
template <int WORDS, bool VOL>
__device__ __forceinline__ void resolve_pixel(float* sA, float* __restrict__ vol, const float* __restrict__ in0,
                                              const float* __restrict__ in1, long in1_frame_stride,
                                              int64_t* __restrict__ idx, float* __restrict__ best, int D, int H, int W, int bs,
                                              float eps, const ListedPlanes& lp, long pj, int lane) {
  const int half = bs / 2, span = bs + D - 1, tailc = bs - 1 - bs / 2;
  float* sB = sA + bs * bs;
  float* sAq = sB + bs * span;
  float* sBq = sAq + bs * bs;
  const float bs2f = (float)(bs * bs);
  const long HW = (long)H * W;
  const long fj = pj / HW, qj = pj - fj * HW;
  const int hj = (int)(qj / W), wj = (int)(qj - (long)hj * W);
  float* v = VOL ? vol + fj * D * HW + qj : nullptr;
  const int d_clamped = wj + tailc;
  unsigned long long mask[WORDS], lmask[WORDS];
  int n_cand = 0;
  float x[WORDS];
  bool lst[WORDS];
  bool any_listed = false;                                 // wave-uniform
  const bool frame_listed = lp.v1 && lp.v0[(fj * H + hj) * lp.Wp + 4 + wj] == 0.f;     // wave-uniform: every entry is a placeholder
  {
    const float* v1row = lp.v1 ? lp.v1 + ((lp.per_frame ? fj : 0) * H + hj) * lp.W1 + lp.xoff : nullptr;
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd) {
      const int d = min(wd * 64 + lane, D - 1);
      lst[wd] = lp.v1 && (frame_listed || v1row[max(wj - d, -tailc)] == 0.f);
      lmask[wd] = __ballot(lst[wd] && wd * 64 + lane < D);
      any_listed |= lmask[wd] != 0ull;
      if constexpr (VOL) x[wd] = v[(long)min(d, d_clamped) * HW];   // (entries past d_clamped are copies of the run's first)
    }
  }
  // !VOL (every pixel is re-scored): the frame window and the pattern rows are requested at once, one batch of
  // independent loads per lane held in registers (block 9, D <= 128).  With a volume most pixels turn out to have a
  // single candidate once their column is read, and requesting the rows ahead of that decision measured no gain.
  constexpr int kPreA = 2, kPreB = 20;
  const bool pre = !VOL && bs * bs <= 64 * kPreA && bs * span <= 64 * kPreB;  // wave-uniform
  const float* a = in0 + fj * HW;
  const float* b = in1 + fj * in1_frame_stride;
  float ta[kPreA], tb[kPreB];
  if (pre) {
#pragma unroll
    for (int k = 0; k < kPreA; ++k) {
      const int i = min(lane + 64 * k, bs * bs - 1);
      const int bh = i / bs, bw = i - bh * bs;
      ta[k] = a[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj + bw - half, 0, W - 1)];
    }
#pragma unroll
    for (int k = 0; k < kPreB; ++k) {
      const int i = min(lane + 64 * k, bs * span - 1);
      const int bh = i / span, c = i - bh * span;
      tb[k] = b[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj - half - (D - 1) + c, 0, W - 1)];
    }
  }
  if constexpr (VOL) {
    // best fast score among the UNLISTED entries; of the clamped run (copies of one score) only the first element takes part
    float m = -INFINITY;
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd)
      if (wd * 64 + lane < D && wd * 64 + lane <= d_clamped && !lst[wd]) m = fmaxf(m, x[wd]);
    m = wave_maxf(m);
    bool have_clamped = false;
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd) {
      unsigned long long bits = __ballot(wd * 64 + lane < D && (lst[wd] || x[wd] >= m - eps));
      const int c0 = d_clamped - wd * 64;                  // bits >= c0 belong to the clamped run
      if (c0 < 64) {
        const unsigned long long run = c0 <= 0 ? bits : bits & ~((1ull << c0) - 1ull);
        bits &= ~run;
        if (!have_clamped && run) {
          bits |= run & (0ull - run);                      // lowest disparity of the run stands for all of it
          have_clamped = true;
        }
      }
      mask[wd] = bits;
      n_cand += __popcll(bits);
    }
  } else {
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd) {
      const int last = min(D - 1, d_clamped) - wd * 64;    // candidates: bits 0 .. last of this word
      mask[wd] = last < 0 ? 0ull : (last >= 63 ? ~0ull : ((1ull << (last + 1)) - 1ull));
      n_cand += __popcll(mask[wd]);
    }
  }
  float eb = 0.f;
  int ei = 0x7fffffff;
  // with a volume a single UNLISTED candidate is final and its score is read back; without one (or when the lone
  // candidate is a placeholder) the exact score still has to be formed
  bool cand_listed = false;
#pragma unroll
  for (int wd = 0; wd < WORDS; ++wd) cand_listed |= (mask[wd] & lmask[wd]) != 0ull;
  const bool rescore = VOL ? (n_cand > 1 || cand_listed) : n_cand >= 1;            // wave-uniform
  if (rescore && pre) {
#pragma unroll
    for (int k = 0; k < kPreA; ++k)
      if (lane + 64 * k < bs * bs) {
        sA[lane + 64 * k] = ta[k];
        sAq[lane + 64 * k] = ta[k] / bs2f;
      }
#pragma unroll
    for (int k = 0; k < kPreB; ++k)
      if (lane + 64 * k < bs * span) {
        sB[lane + 64 * k] = tb[k];
        sBq[lane + 64 * k] = tb[k] / bs2f;
      }
  } else if (rescore) {
    for (int i = lane; i < bs * bs; i += 64) {
      const int bh = i / bs, bw = i - bh * bs;
      const float xa = a[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj + bw - half, 0, W - 1)];
      sA[i] = xa;
      sAq[i] = xa / bs2f;
    }
    constexpr int NB = 20;                  // block 9, D <= 128: all 1224 elements in ONE round trip
    for (int i0 = lane; i0 < bs * span; i0 += 64 * NB) {         // NB independent loads in flight per lane
      float t[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int i = min(i0 + 64 * k, bs * span - 1);
        const int bh = i / span, c = i - bh * span;
        t[k] = b[(long)clampi(hj + bh - half, 0, H - 1) * W + clampi(wj - half - (D - 1) + c, 0, W - 1)];
      }
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (i0 + 64 * k < bs * span) {
          sB[i0 + 64 * k] = t[k];
          sBq[i0 + 64 * k] = t[k] / bs2f;
        }
    }
  }
  // Exact re-scoring, lane <-> candidate (ascending d): every lane runs the reference's serial accumulations for
  // its own candidate out of LDS.  Then the lowest d among the best exact scores.
  float run_e = 0.f;                                       // exact score of the clamped run's first element (frame-listed pixels)
  if (rescore) {
    for (int c0 = 0; c0 < n_cand; c0 += 64) {
      int my_d = -1, k = 0;
      bool my_listed = false;
#pragma unroll
      for (int wd = 0; wd < WORDS; ++wd) {
        unsigned long long mj = mask[wd];
        if (k + __popcll(mj) <= c0 || k >= c0 + 64) { k += __popcll(mj); continue; }   // word outside this round
        while (mj) {
          const int bit = __ffsll((long long)mj) - 1;
          mj &= mj - 1;
          if (k - c0 == lane) { my_d = wd * 64 + bit; my_listed = ((lmask[wd] >> bit) & 1ull) != 0ull; }
          ++k;
        }
      }
      float e = -INFINITY;
      if (my_d >= 0) e = bs == 9 ? ncc_exact_point_lds_bs<9>(sA, sB, sAq, sBq, span, (D - 1) - my_d)
                                 : ncc_exact_point_lds(sA, sB, bs, span, (D - 1) - my_d);
      if constexpr (VOL) {
        // listed entries of this pixel's column get their reference-order value here as well (the same bits the fix-up of
        // their window writes; a frame-listed pixel has no other writer).  A listed clamped run is spread by its window's item.
        if (my_d >= 0 && my_listed && my_d < d_clamped) v[(long)my_d * HW] = e;
        if (my_d >= 0 && my_listed && my_d == d_clamped && d_clamped < D) v[(long)my_d * HW] = e;
        const unsigned long long at_run = __ballot(my_d == d_clamped && my_d >= 0);
        if (at_run) run_e = __shfl(e, __ffsll((long long)at_run) - 1);
      }
      // best exact score of the round, lowest d among its holders; strict > across rounds keeps the lowest index on ties
      const float em = wave_maxf(e);
      const unsigned long long holders = __ballot(my_d >= 0 && e == em);
      const int dl = __shfl(my_d, holders ? __ffsll((long long)holders) - 1 : 0);
      if (ei == 0x7fffffff || em > eb) { eb = em; ei = dl; }
    }
    if constexpr (VOL) {
      // a frame-listed pixel's copies of the run's first element (no window item of the pattern spreads them)
      if (frame_listed)
        for (int d = d_clamped + 1 + lane; d < D; d += 64) v[(long)d * HW] = run_e;
    }
  } else {
#pragma unroll
    for (int wd = 0; wd < WORDS; ++wd)
      if (mask[wd]) ei = min(ei, wd * 64 + __ffsll((long long)mask[wd]) - 1);
  }
  if (lane == 0) {
    // A pixel with listed entries may still receive the patch key of another window's clear winner (ctd_fixup.h,
    // atomic maximum on its index word): its settled index goes in as the key that beats them all, (0xFFFFFFFF, ~d),
    // and the decode pass turns it into a plain index.  Every other pixel's index word is nobody else's: plain store.
    if (any_listed) atomicMax((unsigned long long*)idx + pj, (0xFFFFFFFFull << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ei));
    else idx[pj] = ei;
    if (best) {
      if (rescore) best[pj] = eb;
      else if constexpr (VOL) best[pj] = v[(long)min(ei, d_clamped) * HW];
    }
  }
}

}  // namespace ctd
