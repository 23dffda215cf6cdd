// argmax_rerank.hip -- argmax over disparity of a FAST (tolerance-level) NCC volume with the
// indices of the REFERENCE-ORDER volume.
//
// No reference code exists for the argmax; the contract is
// torch.argmax(xcorrvol_cpu(in0, in1, D, bs), 0): first index wins ties.  The fast volume
// differs from the reference-order one by <= ~1e-6, which can flip near-ties, so every
// disparity whose fast score lies within `eps` of the pixel's best fast score is re-scored
// with the reference's own operation order (XCorrVolFunctor, torchext/ext/ext.h:120-191,
// two passes over the window, no FMA) and the best exact score wins, lowest d first.
// (Along the left border, where the reference volume is exactly constant in d, only the
// lowest disparity of the constant run is a candidate.)  Two kernels: a coalesced sweep that
// finalises every pixel whose runner-up is farther than eps from its best, and a resolve pass
// in which a whole wavefront (lane <-> disparity) re-scores each remaining pixel.
#include "ctd_internal.h"
#include "ctd_ncc_point.h"
#include "ctd_rank.h"
#include "ctd_tail.h"


namespace ctd {

// Pass 1: one coalesced sweep over d, 4 adjacent pixels per thread (16-byte loads), branch-free tracking of
// the best score / index and of the runner-up score.  Pixels whose runner-up is within eps of the best are
// marked (idx = -1 - argmax) for pass 2; all others are final.
constexpr int kScanBatch = 8;
template <int PX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void argmax_scan_kernel(const float* __restrict__ vol, int64_t* __restrict__ idx,
                                                          float* __restrict__ best, int D, long HW, int W,
                                                          int bs, float eps, long total_threads, WorkList work) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total_threads) return;
  const long p0 = t * PX;                         // first pixel (flat over frames); HW % 4 == 0 when VEC4
  const long f = p0 / HW, q0 = p0 - f * HW;
  const float* v = vol + f * D * HW + q0;
  // Past d_clamped = w + (bs-1-bs/2) every pattern tap clamps to column 0 (ext.h:152-154): the scores of
  // that run are one window's score repeated, so only its first element takes part in the ranking.
  const int w0 = (int)(q0 % W);                   // VEC4 implies W % 4 == 0: the 4 pixels share a row
  float v0[PX], v1[PX];
  int i0[PX], dc[PX];
#pragma unroll
  for (int k = 0; k < PX; ++k) { v0[k] = -INFINITY; v1[k] = -INFINITY; i0[k] = 0; dc[k] = w0 + k + (bs - 1 - bs / 2); }
  // explicit batches of kBatch independent loads (the compiler otherwise waits for each plane before
  // requesting the next); streamed once, so non-temporal
  constexpr int kBatch = kScanBatch;
  const bool wave_masks = __any(dc[0] < D - 1);
  for (int d0 = 0; d0 < D; d0 += kBatch) {
    float x[kBatch][PX];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const float* src = v + (long)min(d0 + u, D - 1) * HW;
      if constexpr (PX == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 q = __builtin_nontemporal_load((const f4*)src);
        x[u][0] = q.x; x[u][1] = q.y; x[u][2] = q.z; x[u][3] = q.w;
      } else if constexpr (PX == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 q = __builtin_nontemporal_load((const f2*)src);
        x[u][0] = q.x; x[u][1] = q.y;
      } else {
        x[u][0] = __builtin_nontemporal_load(src);
      }
    }
    // a wave whose columns all lie past D - 1 - (bs-1-bs/2) has no clamped run to mask (wave-uniform)
    if (wave_masks || d0 + kBatch > D) {
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int d = d0 + u;
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          const float y = (d > dc[k] || d >= D) ? -INFINITY : x[u][k];
          v1[k] = __builtin_amdgcn_fmed3f(v0[k], v1[k], y);   // runner-up = second largest of {v0 >= v1, y}
          i0[k] = y > v0[k] ? d : i0[k];                      // strict >: first index wins ties
          v0[k] = fmaxf(v0[k], y);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
#pragma unroll
        for (int k = 0; k < PX; ++k) {
          const float y = x[u][k];
          v1[k] = __builtin_amdgcn_fmed3f(v0[k], v1[k], y);
          i0[k] = y > v0[k] ? d0 + u : i0[k];
          v0[k] = fmaxf(v0[k], y);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < PX; ++k) {
    const bool hard = v1[k] >= v0[k] - eps;
    idx[p0 + k] = hard ? (int64_t)(-1 - i0[k]) : (int64_t)i0[k];
    if (best) best[p0 + k] = v0[k];
    if (hard) work.list[atomicAdd(work.counters, 1u)] = p0 + k;   // rare (a few pixels in 10^4): pass 2's work list
  }
}

// Pass 2: one wavefront per pixel of the work list.  VOL: lane <-> disparity collects the pixel's candidate set from
// the (patched) fast volume with one round of loads (scores within eps of the best; the run of disparities whose
// window is clamped to column 0 counted once, lowest d); a single candidate is final, otherwise the frame window
// and the reachable pattern rows are staged in LDS and the wave re-scores each candidate in reference order.
// !VOL (no volume was materialised): every disparity up to the start of the clamped run is a candidate.
constexpr int kMaskWords = 8;            // at most 512 disparities per candidate mask

__device__ inline float wave_maxf(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// WORDS = 64-disparity words of the candidate mask (2 for D <= 128 ... 8 for D <= 512)
// Fully clamped runs in a ranked call: the run spreading (runs_role) works beside this pass, so the volume's entries
// past d_clamped are not read at all -- where the run's window is listed (zero reciprocal deviation in the pattern's
// plane) the run's exact value comes from run_vals, elsewhere the copies equal the first element anyway.
struct RunSource {
  const float* run_vals;      // null: the volume is fully patched (unranked call), read it everywhere
  const float* v1;
  int W1, xoff, per_frame;
};

template <int WORDS, bool VOL>
__device__ __forceinline__ void resolve_role(float* lds_resolve, const float* __restrict__ vol, const float* __restrict__ in0,
                                             const float* __restrict__ in1, long in1_frame_stride,
                                             int64_t* __restrict__ idx, float* __restrict__ best, int D, int H, int W, int bs,
                                             float eps, WorkList work, RunSource rsrc, unsigned role_block,
                                             unsigned n_role_blocks, unsigned waves_used = 4) {
  // `waves_used` of the workgroup's four wavefronts work (the tail kernel uses two: half the staging LDS per workgroup,
  // so that seven workgroups fit a CU instead of three and the other roles of that launch keep their concurrency)
  if ((threadIdx.x >> 6) >= waves_used) return;
  const int lane = threadIdx.x & 63;
  const int half = bs / 2, span = bs + D - 1;
  // per-wave staging area: frame window / pattern rows, raw and divided by bs^2 (the reference divides every tap
  // before it sums the means, ext.h:157-158: done once per staged element, not once per candidate and tap)
  float* sA = lds_resolve + (threadIdx.x >> 6) * 2 * (bs * bs + bs * span);
  float* sB = sA + bs * bs;
  float* sAq = sB + bs * span;
  float* sBq = sAq + bs * bs;
  const float bs2f = (float)(bs * bs);
  const long HW = (long)H * W;
  // The list's segments are walked side by side (slot i = entry i / parts of segment i % parts): taken end to end,
  // one image-row key after the other, the pass measured 42 instead of 28 us.  All counters are requested at once (the
  // unused ones of a one-segment list read counter 0 again): as a conditional load per segment they were a chain of
  // sixteen dependent scalar loads.
  unsigned seg_cnt[kWorkListParts];
  unsigned longest = 0;
#pragma unroll
  for (int k = 0; k < kWorkListParts; ++k) seg_cnt[k] = work.counters[min(k, work.parts - 1) * kWorkListStride];
#pragma unroll
  for (int k = 0; k < kWorkListParts; ++k) {
    if (k >= work.parts) seg_cnt[k] = 0u;
    longest = max(longest, seg_cnt[k]);
  }
  const unsigned n_slots = longest * (unsigned)work.parts;
  const unsigned n_waves = n_role_blocks * waves_used;
  for (unsigned slot = role_block * waves_used + (threadIdx.x >> 6); slot < n_slots; slot += n_waves) {
    const unsigned seg = slot & (unsigned)(work.parts - 1), entry = slot / (unsigned)work.parts;
    unsigned cnt = 0;
#pragma unroll
    for (int k = 0; k < kWorkListParts; ++k) cnt = seg == (unsigned)k ? seg_cnt[k] : cnt;
    if (entry >= cnt) continue;                            // wave-uniform: this segment is shorter
    const long pj = work.list[(long)seg * work.seg_cap + entry];   // wave-uniform from here on
    const long fj = pj / HW, qj = pj - fj * HW;
    const int hj = (int)(qj / W), wj = (int)(qj - (long)hj * W);
    const float* v = VOL ? vol + fj * D * HW + qj : nullptr;
    const int d_clamped = wj + (bs - 1 - bs / 2);
    unsigned long long mask[WORDS];
    int n_cand = 0;
    float x[WORDS];
    bool run_listed = false;                               // wave-uniform
    float rv = 0.f;
    if constexpr (VOL) {
      if (rsrc.run_vals && d_clamped < D) {
        const long z = rsrc.per_frame ? fj : 0;              // (ranked calls are single channel)
        run_listed = rsrc.v1[(z * H + hj) * rsrc.W1 + (rsrc.xoff - (bs - 1 - bs / 2))] == 0.f;
        if (run_listed) rv = rsrc.run_vals[(fj * H + hj) * D + d_clamped];
      }
#pragma unroll
      for (int wd = 0; wd < WORDS; ++wd) {
        const int d = min(wd * 64 + lane, D - 1);
        // (a ranked call's run entries past d_clamped are being written by runs_role right now: not read)
        x[wd] = v[(long)(rsrc.run_vals ? min(d, d_clamped) : d) * HW];
        if (run_listed && d >= d_clamped) x[wd] = rv;
      }
    }
    // !VOL (every pixel of the list is re-scored): the frame window and the pattern rows are requested at once, one
    // batch of independent loads per lane held in registers (block 9, D <= 128: kPreA + kPreB of them; -5 us of 0.53 ms).
    // With a volume most listed pixels turn out to have a single candidate once their column is read, and requesting
    // the rows ahead of that decision measured no gain: those calls load them after the count, as before.
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
      // best fast score; of the clamped run (copies of one score) only the first element takes part
      float m = -INFINITY;
#pragma unroll
      for (int wd = 0; wd < WORDS; ++wd)
        if (wd * 64 + lane < D && wd * 64 + lane <= d_clamped) m = fmaxf(m, x[wd]);
      m = wave_maxf(m);
      bool have_clamped = false;
#pragma unroll
      for (int wd = 0; wd < WORDS; ++wd) {
        unsigned long long bits = __ballot(wd * 64 + lane < D && x[wd] >= m - eps);
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
    // with a volume a single candidate is final and its (patched) score is read back; without one the exact score of a
    // lone candidate still has to be formed (D = 1 with a listed window: the ranking only saw the placeholder)
    const bool rescore = VOL ? n_cand > 1 : n_cand >= 1;            // wave-uniform
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
      constexpr int NB = 20;                  // block 9, D <= 128: all 1224 elements in ONE round trip (8 deep: tail kernel +2 us)
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
    // its own candidate out of LDS (a whole-wave evaluation of one candidate at a time, fed by v_readlane, costs
    // ~15 us per candidate; this costs ~3 us for all of them).  Then the lowest d among the best exact scores.
    if (rescore) {
      for (int c0 = 0; c0 < n_cand; c0 += 64) {
        int my_d = -1, k = 0;
#pragma unroll
        for (int wd = 0; wd < WORDS; ++wd) {
          unsigned long long mj = mask[wd];
          if (k + __popcll(mj) <= c0 || k >= c0 + 64) { k += __popcll(mj); continue; }   // word outside this round
          while (mj) {
            const int d = wd * 64 + __ffsll((long long)mj) - 1;
            mj &= mj - 1;
            if (k - c0 == lane) my_d = d;
            ++k;
          }
        }
        float e = -INFINITY;
        if (my_d >= 0) e = bs == 9 ? ncc_exact_point_lds_bs<9>(sA, sB, sAq, sBq, span, (D - 1) - my_d)
                                   : ncc_exact_point_lds(sA, sB, bs, span, (D - 1) - my_d);
        // best exact score of the round, lowest d among its holders; strict > across rounds keeps the lowest index on ties
        const float em = wave_maxf(e);
        const unsigned long long holders = __ballot(my_d >= 0 && e == em);
        const int dl = __shfl(my_d, holders ? __ffsll((long long)holders) - 1 : 0);
        if (ei == 0x7fffffff || em > eb) { eb = em; ei = dl; }
      }
    } else {
#pragma unroll
      for (int wd = 0; wd < WORDS; ++wd)
        if (mask[wd]) ei = min(ei, wd * 64 + __ffsll((long long)mask[wd]) - 1);
    }
    if (lane == 0) {
      idx[pj] = ei;
      if (best) {
        if constexpr (VOL) best[pj] = (run_listed && ei >= d_clamped) ? rv : v[(long)(rsrc.run_vals ? min(ei, d_clamped) : ei) * HW];
        else if (rescore) best[pj] = eb;                     // no fast score exists: the reference-order one
      }
    }
  }
}

template <int WORDS, bool VOL>
__global__ __launch_bounds__(256) void argmax_resolve_kernel(const float* __restrict__ vol, const float* __restrict__ in0,
                                                             const float* __restrict__ in1, long in1_frame_stride,
                                                             int64_t* __restrict__ idx, float* __restrict__ best, int D,
                                                             int H, int W, int bs, float eps, WorkList work) {
  extern __shared__ float lds_dyn[];
  resolve_role<WORDS, VOL>(lds_dyn, vol, in0, in1, in1_frame_stride, idx, best, D, H, W, bs, eps, work,
                           RunSource{nullptr, nullptr, 0, 0, 0}, blockIdx.x, gridDim.x);
}

constexpr long kResolveBlocks = 1024;
constexpr unsigned kTailResolveWaves = 2;       // working wavefronts of a resolve workgroup of the tail kernel

// Tail of a ranked call, ONE launch behind the fix-up kernel, three independent roles by workgroup number:
//   [0, n_resolve)            exact re-scoring of the work-list pixels (resolve_role);
//   [.., + n_runs)            spreading of the listed fully clamped runs into the volume (runs_role; volume calls only);
//   [.., + n_decode)          patched index words -> plain indices and best scores (decode_role).
// All three need the fix-up kernel complete and nothing of each other: the resolve role reads run values from run_vals,
// never from the part of the volume the runs role is writing, and the decode role leaves work-list pixels alone.
// (As three kernels they were 36 + 24 us of dependent launches in round 2; each is a chain of global round trips with
// most of the chip idle.)
template <int WORDS, bool VOL>
__global__ __launch_bounds__(256) void rank_tail_kernel(float* __restrict__ vol, const float* __restrict__ in0,
                                                        const float* __restrict__ in1, long in1_frame_stride,
                                                        int64_t* __restrict__ idx, float* __restrict__ best,
                                                        const unsigned char* __restrict__ flags, int frames, int D, int H,
                                                        int W, int bs, float eps, WorkList work, RunSource rsrc,
                                                        unsigned* __restrict__ counters,
                                                        const unsigned long long* __restrict__ run_rows,
                                                        const unsigned long long* __restrict__ flag_a,
                                                        const unsigned long long* __restrict__ flag_b, unsigned n_resolve,
                                                        unsigned n_runs, unsigned n_decode) {
  extern __shared__ float lds_dyn[];
  // Roles by workgroup number, resolve and runs workgroups alternating while both last: the resolve role is the longest
  // chain and must not queue behind the others for a CU slot, nor they behind it (all share one LDS size).
  const unsigned n_pair = min(n_resolve, n_runs);
  unsigned role, rb;                                          // 0 resolve, 1 runs, 2 decode; number within the role
  if (blockIdx.x < n_decode) { role = 2; rb = blockIdx.x; }
  else {
    const unsigned b = blockIdx.x - n_decode;
    if (b < 2 * n_pair) { role = b & 1; rb = b >> 1; }
    else { role = n_resolve > n_runs ? 0 : 1; rb = b - n_pair; }
  }
  // (nobody reads slot 0 in this launch -- the decode role takes the count from slot 3, ncc_fixup_kernel -- and the next
  // call's pre-pass counts its listed frame windows from zero in it)
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[0] = 0u;
  if (role == 0) {
    resolve_role<WORDS, VOL>(lds_dyn, vol, in0, in1, in1_frame_stride, idx, best, D, H, W, bs, eps, work, rsrc, rb, n_resolve,
                             kTailResolveWaves);
  } else if (role == 1) {
    if constexpr (VOL)
      runs_role(vol, rsrc.run_vals, counters, run_rows, rsrc.per_frame, 1, H, W, D, bs, (int)(rb >> 2), (int)(rb & 3), 4,
                (int*)lds_dyn);
  } else {
    decode_role((unsigned long long*)idx, best, flags, counters, flag_a, flag_b, rsrc.per_frame, frames, H, W, D,
                rb * 4 + (threadIdx.x >> 6), n_decode * 4);
  }
}


template <bool VOL>
static int launch_resolve(const float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                          float* best, long total, int D, int H, int W, int bs, float eps, const WorkList& work,
                          hipStream_t stream) {
  const size_t lds = sizeof(float) * 4 * 2 * ((size_t)bs * bs + (size_t)bs * (bs + D - 1));
  if (lds > 160 * 1024) return CTD_ERR_UNSUPPORTED;
  const long chunks = (total + 255) / 256;
  auto resolve = D <= 128 ? argmax_resolve_kernel<2, VOL> : (D <= 256 ? argmax_resolve_kernel<4, VOL> : argmax_resolve_kernel<8, VOL>);
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(resolve, dim3((unsigned)(chunks < kResolveBlocks ? chunks : kResolveBlocks)), dim3(256), lds, stream, vol,
                     in0, in1, in1_frame_stride, idx, best, D, H, W, bs, eps, work);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <bool VOL>
static int launch_tail(const RankPlan& rp, float* vol, const float* in0, const float* in1, long in1_frame_stride,
                       int64_t* idx, float* best, int frames, int D, int H, int W, int bs, hipStream_t stream) {
  size_t lds = sizeof(float) * kTailResolveWaves * 2 * ((size_t)bs * bs + (size_t)bs * (bs + D - 1));
  if (lds < sizeof(int) * (size_t)H) lds = sizeof(int) * (size_t)H;          // runs role: the rows of a frame's pattern
  if (lds > 160 * 1024) return CTD_ERR_UNSUPPORTED;
  const long total = (long)frames * H * W;
  const long chunks = (total + 255) / 256;
  // (only the pixels inside the margin come here, ctd_tail.h patch_key: a few thousand of millions; workgroups without
  // an item leave at once)
  const unsigned n_resolve = rp.eps >= 0.f ? (unsigned)(chunks < kResolveBlocks ? chunks : kResolveBlocks) : 0u;
  const unsigned n_runs = VOL ? (unsigned)(frames * ceil_div(D, kRunPlanes)) * 4u : 0u;
  const unsigned n_decode = 128u;
  auto kern = D <= 128 ? rank_tail_kernel<2, VOL> : (D <= 256 ? rank_tail_kernel<4, VOL> : rank_tail_kernel<8, VOL>);
  if (lds > 64 * 1024)
    CTD_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const RunSource rsrc = {rp.run_vals, rp.v1, rp.W1, rp.xoff, in1_frame_stride != 0 ? 1 : 0};
  hipLaunchKernelGGL(kern, dim3(n_resolve + n_runs + n_decode), dim3(256), lds, stream, vol, in0, in1, in1_frame_stride, idx,
                     best, rp.flags, frames, D, H, W, bs, rp.eps, rp.work, rsrc, rp.counters, rp.run_rows, rp.flag_a, rp.flag_b,
                     n_resolve,
                     n_runs, n_decode);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

// Last pass of a ranked call (after ncc_fast_fixup_ranked): see rank_tail_kernel.
int rank_tail_f32(const RankPlan& rp, float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                  float* best, int frames, int D, int H, int W, int bs, hipStream_t stream) {
  return vol ? launch_tail<true>(rp, vol, in0, in1, in1_frame_stride, idx, best, frames, D, H, W, bs, stream)
             : launch_tail<false>(rp, nullptr, in0, in1, in1_frame_stride, idx, best, frames, D, H, W, bs, stream);
}

int argmax_rerank_f32(const float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                      float* best, int frames, int D, int H, int W, int bs, float eps, void* workspace,
                      size_t workspace_bytes, bool counter_cleared, hipStream_t stream) {
  if (D > kMaskWords * 64) return CTD_ERR_UNSUPPORTED;
  const long total = (long)frames * H * W;
  const long HW = (long)H * W;
  // work list of marked pixels: a counter and room for every pixel (the volume kernel is done with the
  // workspace by the time these kernels run on the same stream)
  if (!workspace || workspace_bytes < 16 + sizeof(int64_t) * (size_t)total) return CTD_ERR_WORKSPACE;
  unsigned* n_hard = (unsigned*)workspace;
  const WorkList work = {n_hard, (int64_t*)((char*)workspace + 16), 1, W, total};     // one plain list
  if (!counter_cleared) CTD_HIP_TRY(hipMemsetAsync(n_hard, 0, 16, stream));   // ncc_fixup_runs_kernel clears it
  const bool vec4 = W % 4 == 0 && ((uintptr_t)vol % 16) == 0;
  const int px = vec4 ? 4 : 1;
  if (px == 4)
    hipLaunchKernelGGL(argmax_scan_kernel<4>, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream, vol,
                       idx, best, D, HW, W, bs, eps, total / 4, work);
  else
    hipLaunchKernelGGL(argmax_scan_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, vol, idx,
                       best, D, HW, W, bs, eps, total, work);
  CTD_LAUNCH_CHECK();
  if (eps < 0.f) return CTD_OK;                            // nothing is marked: plain argmax of the fast volume
  return launch_resolve<true>(vol, in0, in1, in1_frame_stride, idx, best, total, D, H, W, bs, eps, work, stream);
}

}  // namespace ctd
