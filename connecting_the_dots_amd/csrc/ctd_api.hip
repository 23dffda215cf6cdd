// ctd_api.hip -- the extern "C" surface declared in include/ctd_hip.h.
// Argument validation lives here; kernels assume validated shapes.
#include "ctd_internal.h"
#include "ctd_prepass.h"
#include "../../include/ctd_hip_bench.h"

#include <deque>
#include <utility>
#include <vector>

using namespace ctd;

namespace ctd {
// Bench instrumentation (include/ctd_hip_bench.h, ctd_kernel_timing_*): NOT part of the drop-in ABI of
// include/ctd_hip.h.  The state is per calling thread (thread_local): the thread that enables it sees its own launches
// only, other threads of the process launch un-instrumented and race with nothing.
// The events come from a pool created when timing is switched on (no hipEventCreate between launches) and carry
// hipEventDisableSystemFence: a default event makes the queue release to system scope at every record (an L2
// write-back of whatever the previous kernel left dirty), which the un-instrumented path never pays.
static thread_local bool g_timing = false;
static thread_local int g_timing_columns = 0;
static thread_local std::deque<hipEvent_t> g_pool;                                    // free events, reused first-in first-out
static thread_local std::vector<std::pair<hipEvent_t, hipEvent_t>> g_events;          // recorded (start, stop) pairs
static thread_local hipEvent_t g_pending = nullptr;
static hipEvent_t pool_get() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.front();
    g_pool.pop_front();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
  return e;
}
bool timing_enabled() { return g_timing; }
void timing_begin(hipStream_t stream) {
  if (!g_timing) return;
  if (!g_pending) g_pending = pool_get();            // (a launch that failed between begin and end left its event here)
  if (g_pending) (void)hipEventRecord(g_pending, stream);
}
void timing_end(hipStream_t stream, int columns) {
  if (!g_timing || !g_pending) return;
  hipEvent_t stop = pool_get();
  if (!stop) return;
  (void)hipEventRecord(stop, stream);
  g_events.emplace_back(g_pending, stop);
  g_pending = nullptr;
  g_timing_columns = columns;
}
}  // namespace ctd

extern "C" {

int ctd_version(void) { return 5; }

void ctd_kernel_timing_enable(int enable) {
  g_timing = enable != 0;
  if (g_timing) {
    // fill the pool up front (enough for a default bench region) and record every new event once: the runtime
    // allocates an event's signal at its first record, and that must not happen between the start event and the
    // kernel it brackets
    bool fresh = false;
    const size_t want = enable > 128 ? (size_t)enable : 128;          // `enable` doubles as the number of events to hold ready
    while (g_pool.size() < want) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) break;
      (void)hipEventRecord(e, nullptr);
      g_pool.push_back(e);
      fresh = true;
    }
    if (fresh) (void)hipStreamSynchronize(nullptr);
  } else if (g_pending) {
    g_pool.push_back(g_pending);
    g_pending = nullptr;
  }
}

int ctd_kernel_timing_collect(double* avg_ms, int* columns) {
  double total = 0;
  int n = 0;
  for (auto& ev : g_events) {
    float ms = 0.f;
    if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
      total += ms;
      ++n;
    }
    g_pool.push_back(ev.first);
    g_pool.push_back(ev.second);
  }
  g_events.clear();
  if (avg_ms) *avg_ms = n ? total / n : 0.0;
  if (columns) *columns = g_timing_columns;
  return n;
}

const char* ctd_status_string(int status) {
  switch (status) {
    case CTD_OK: return "ok";
    case CTD_ERR_INVALID_ARG: return "invalid argument";
    case CTD_ERR_WORKSPACE: return "workspace missing or too small";
    case CTD_ERR_UNSUPPORTED: return "unsupported parameter combination";
    default: break;
  }
  if (status >= CTD_ERR_HIP) return hipGetErrorString((hipError_t)(status - CTD_ERR_HIP));
  return "unknown status";
}

static bool vol_shape_ok(int frames, int C, int H, int W, int D, int bs) {
  if (frames < 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0 || bs <= 0) return false;
  // the reference indexes outputs with int (common_cuda.h:65-66, ext_cpu.cpp:8-9)
  if ((double)D * H * W >= 2147483648.0) return false;
  return true;
}

size_t ctd_xcorrvol_workspace_bytes(int frames, int C, int H, int W, int D, int block_size, int algo) {
  if (!vol_shape_ok(frames, C, H, W, D, block_size)) return 0;
  // worst case over "pattern shared" / "pattern per frame"
  size_t exact = ncc_exact_workspace_bytes(frames, C, H, W, D, block_size, true);
  if (algo == CTD_NCC_EXACT) return exact;
  size_t fast = ncc_fast_workspace_bytes(frames, C, H, W, D, block_size, true);
  return fast > exact ? fast : exact;
}

int ctd_xcorrvol_pattern_prepare_f32(const float* in1, long in1_frame_stride, int frames, int C, int H, int W, int D,
                                     int block_size, void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (!vol_shape_ok(frames, C, H, W, D, block_size) || in1_frame_stride < 0 || frames == 0 || !in1) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return ncc_fast_prepare_pattern_f32(in1, in1_frame_stride, frames, C, H, W, D, block_size, workspace, workspace_bytes,
                                      (hipStream_t)stream);
}

int ctd_xcorrvol_f32(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H,
                     int W, int D, int block_size, int algo, void* workspace, size_t workspace_bytes, int device,
                     void* stream) {
  const bool prepared = (algo & CTD_PATTERN_PREPARED) != 0;
  algo &= ~CTD_PATTERN_PREPARED;
  if (prepared && algo != CTD_NCC_FAST) return CTD_ERR_INVALID_ARG;
  if (!vol_shape_ok(frames, C, H, W, D, block_size) || in1_frame_stride < 0) return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!in0 || !in1 || !out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  if (algo == CTD_NCC_EXACT)
    return ncc_exact_f32(in0, in1, in1_frame_stride, out, frames, C, H, W, D, block_size, workspace, workspace_bytes,
                         (hipStream_t)stream);
  if (algo == CTD_NCC_FAST)
    return ncc_fast_f32(in0, in1, in1_frame_stride, out, frames, C, H, W, D, block_size, workspace, workspace_bytes,
                        nullptr, prepared, (hipStream_t)stream);
  return CTD_ERR_INVALID_ARG;
}

int ctd_xcorrvol_f64(const double* in0, const double* in1, long in1_frame_stride, double* out, int frames, int C,
                     int H, int W, int D, int block_size, void* workspace, size_t workspace_bytes, int device,
                     void* stream) {
  if (!vol_shape_ok(frames, C, H, W, D, block_size) || in1_frame_stride < 0) return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!in0 || !in1 || !out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return ncc_exact_f64(in0, in1, in1_frame_stride, out, frames, C, H, W, D, block_size, workspace, workspace_bytes,
                       (hipStream_t)stream);
}

int ctd_argmax_disp_f32(const float* vol, int64_t* idx, float* best, int frames, int D, int H, int W, int device,
                        void* stream) {
  if (frames < 0 || D <= 0 || H <= 0 || W <= 0) return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!vol || !idx) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return argmax_disp_f32(vol, idx, best, frames, D, H, W, (hipStream_t)stream);
}

int ctd_xcorrvol_rank_supported(int C, int H, int W, int D, int block_size) {
  return vol_shape_ok(1, C, H, W, D, block_size) && ncc_fast_rank_supported(C, H, W, D, block_size) ? 1 : 0;
}

int ctd_xcorrvol_rank_layout(int frames, int H, int W, int D, int per_frame_pattern, size_t* offsets) {
  if (!offsets || frames <= 0 || !vol_shape_ok(frames, 1, H, W, D, 9)) return CTD_ERR_INVALID_ARG;
  ncc_fast_rank_offsets(frames, H, W, D, per_frame_pattern != 0, offsets);
  return CTD_OK;
}

size_t ctd_xcorrvol_argmax_workspace_bytes(int frames, int C, int H, int W, int D, int block_size, int algo) {
  size_t base = ctd_xcorrvol_workspace_bytes(frames, C, H, W, D, block_size, algo);
  if (base == 0 || algo != CTD_NCC_FAST) return base;
  // old path: work list behind a 16-byte counter at the start of the workspace
  size_t need = 16 + sizeof(int64_t) * (size_t)frames * H * W;
  if (ncc_fast_rank_supported(C, H, W, D, block_size)) need = ncc_fast_rank_workspace_bytes(frames, H, W, D, true);
  return need > base ? need : base;
}

int ctd_xcorrvol_argmax_f32(const float* in0, const float* in1, long in1_frame_stride, float* vol_out, int64_t* idx,
                            float* best, int frames, int C, int H, int W, int D, int block_size, int algo,
                            float rerank_eps, void* workspace, size_t workspace_bytes, int device, void* stream) {
  const bool prepared = (algo & CTD_PATTERN_PREPARED) != 0;
  algo &= ~CTD_PATTERN_PREPARED;
  if (prepared && algo != CTD_NCC_FAST) return CTD_ERR_INVALID_ARG;
  if (!vol_shape_ok(frames, C, H, W, D, block_size) || in1_frame_stride < 0) return CTD_ERR_INVALID_ARG;
  if (C != 1) return CTD_ERR_UNSUPPORTED;
  if (frames == 0) return CTD_OK;
  if (!in0 || !in1 || !idx) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  if (algo == CTD_NCC_EXACT)
    return ncc_exact_argmax_f32(in0, in1, in1_frame_stride, vol_out, idx, best, frames, H, W, D, block_size, workspace,
                                workspace_bytes, (hipStream_t)stream);
  if (algo == CTD_NCC_FAST) {
    if (rerank_eps != rerank_eps) return CTD_ERR_INVALID_ARG;
    // rerank_eps < 0 (plain argmax of the fast scores, no exact re-scoring) is defined on a materialised volume: the
    // scores of listed windows exist only there (fix-up pass), so such a call ranks the patched volume in one more pass
    const bool plain = rerank_eps < 0.f && vol_out;
    if (!plain && ncc_fast_rank_supported(1, H, W, D, block_size) && ((uintptr_t)vol_out) % 16 == 0) {
      // ranked inside the all-D volume kernel: {top, runner-up} per pixel in LDS across every disparity; the kernel
      // writes idx / best / work list itself -- no partial planes, no merge, no pass over the volume
      RankPlan rp;
      rp.eps = rerank_eps < 0.f ? 0.f : rerank_eps;                        // (no volume: negative eps means 0)
      rp.idx = idx;
      rp.best = best;
      const hipStream_t hs = (hipStream_t)stream;
      int st = ncc_fast_f32(in0, in1, in1_frame_stride, vol_out, frames, 1, H, W, D, block_size, workspace,
                            workspace_bytes, &rp, prepared, hs);       // pre-pass + all-D kernel
      if (st) return st;
      st = ncc_fast_fixup_ranked(in0, in1, in1_frame_stride, vol_out, frames, H, W, D, block_size, workspace, rp, rp.best, hs);
      if (st) return st;
      return rank_tail_f32(rp, vol_out, in0, in1, in1_frame_stride, idx, rp.best, frames, D, H, W, block_size, hs);
    }
    if (!vol_out) return CTD_ERR_INVALID_ARG;                              // this shape ranks a materialised volume
    int st = ncc_fast_f32(in0, in1, in1_frame_stride, vol_out, frames, 1, H, W, D, block_size, workspace,
                          workspace_bytes, nullptr, prepared, (hipStream_t)stream);
    if (st) return st;
    return argmax_rerank_f32(vol_out, in0, in1, in1_frame_stride, idx, best, frames, D, H, W, block_size, rerank_eps,
                             workspace, workspace_bytes, /*counter_cleared=*/true, (hipStream_t)stream);
  }
  return CTD_ERR_INVALID_ARG;
}

int ctd_lcn_xcorrvol_supported(int H, int W, int D, int radius, int block_size) {
  return vol_shape_ok(1, 1, H, W, D, block_size) && ncc_fast_rank_supported(1, H, W, D, block_size) &&
                 lcn_stream_supported(H, W, radius, block_size) ? 1 : 0;
}

int ctd_lcn_xcorrvol_argmax_f32(const float* raw, float* lcn_out, float* std_out, int radius, float lcn_eps, int lcn_algo,
                                const float* in1, long in1_frame_stride, float* vol_out, int64_t* idx, float* best,
                                int frames, int H, int W, int D, int block_size, int algo, float rerank_eps,
                                void* workspace, size_t workspace_bytes, int device, void* stream) {
  const bool prepared = (algo & CTD_PATTERN_PREPARED) != 0;
  algo &= ~CTD_PATTERN_PREPARED;
  if (algo != CTD_NCC_FAST || (lcn_algo != CTD_LCN_EXACT && lcn_algo != CTD_LCN_FAST)) return CTD_ERR_INVALID_ARG;
  if (!vol_shape_ok(frames, 1, H, W, D, block_size) || in1_frame_stride < 0 || radius < 0 || rerank_eps != rerank_eps)
    return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!raw || !lcn_out || !std_out || !in1 || !idx) return CTD_ERR_INVALID_ARG;
  if (!ctd_lcn_xcorrvol_supported(H, W, D, radius, block_size) || ((uintptr_t)vol_out) % 16 != 0) return CTD_ERR_UNSUPPORTED;
  DeviceGuard g(device);
  if (g.status) return g.status;
  const hipStream_t hs = (hipStream_t)stream;
  const FusedLcn fused = {raw, std_out, radius, lcn_eps, lcn_algo == CTD_LCN_EXACT};
  RankPlan rp;
  rp.eps = rerank_eps < 0.f ? 0.f : rerank_eps;             // (as ctd_xcorrvol_argmax_f32 without a plain-argmax pass)
  rp.idx = idx;
  rp.best = best;
  int st = ncc_fast_f32(lcn_out, in1, in1_frame_stride, vol_out, frames, 1, H, W, D, block_size, workspace, workspace_bytes,
                        &rp, prepared, hs, &fused);           // streaming LCN + statistics, then the all-D kernel
  if (st) return st;
  st = ncc_fast_fixup_ranked(lcn_out, in1, in1_frame_stride, vol_out, frames, H, W, D, block_size, workspace, rp, rp.best, hs);
  if (st) return st;
  return rank_tail_f32(rp, vol_out, lcn_out, in1, in1_frame_stride, idx, rp.best, frames, D, H, W, block_size, hs);
}

int ctd_lcn_f32(const float* x, float* y, float* std_out, int N, int H, int W, int radius, float eps, int device,
                void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || radius < 0 || radius >= H || radius >= W || (double)H * W >= 2147483648.0)
    return CTD_ERR_INVALID_ARG;
  if (N == 0) return CTD_OK;
  if (!x || !y || !std_out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return lcn_f32(x, y, std_out, N, H, W, radius, eps, (hipStream_t)stream);
}

int ctd_lcn_fast_f32(const float* x, float* y, float* std_out, int N, int H, int W, int radius, float eps, int device,
                     void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || radius < 0 || radius >= H || radius >= W || (double)H * W >= 2147483648.0)
    return CTD_ERR_INVALID_ARG;
  if (N == 0) return CTD_OK;
  if (!x || !y || !std_out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return lcn_fast_f32(x, y, std_out, N, H, W, radius, eps, (hipStream_t)stream);
}

int ctd_lcn_datagen_f32(const float* img, float* out, float* out_std, int N, int H, int W, int kernel_size, float eps,
                        int device, void* stream) {
  if (N < 0 || H <= 0 || W <= 0 || kernel_size < 0 || N > 65535) return CTD_ERR_INVALID_ARG;
  if (N == 0) return CTD_OK;
  if (!img || !out || !out_std) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return lcn_datagen_f32(img, out, out_std, N, H, W, kernel_size, eps, (hipStream_t)stream);
}

static bool photo_shape_ok(int B, int C, int H, int W, int bs, int type) {
  return B >= 0 && C > 0 && H > 0 && W > 0 && bs > 0 && type >= 0 && type <= 3 &&
         (double)B * C * H * W < 2147483648.0;                 // int indices in the reference (ext.h:220-235)
}

#define CTD_PHOTO_ENTRY(SFX, T)                                                                                   \
  int ctd_photometric_fwd_##SFX(const T* es, const T* ta, T* out, int B, int C, int H, int W, int block_size,      \
                                int type, float eps, int device, void* stream) {                                   \
    if (!photo_shape_ok(B, C, H, W, block_size, type)) return CTD_ERR_INVALID_ARG;                                 \
    if (B == 0) return CTD_OK;                                                                                     \
    if (!es || !ta || !out) return CTD_ERR_INVALID_ARG;                                                            \
    DeviceGuard g(device);                                                                                         \
    if (g.status) return g.status;                                                                                 \
    return photometric_fwd_##SFX(es, ta, out, B, C, H, W, block_size, type, eps, (hipStream_t)stream);             \
  }                                                                                                                \
  int ctd_photometric_bwd_##SFX(const T* es, const T* ta, const T* grad_out, T* grad_es, int B, int C, int H,      \
                                int W, int block_size, int type, float eps, int device, void* stream) {            \
    if (!photo_shape_ok(B, C, H, W, block_size, type)) return CTD_ERR_INVALID_ARG;                                 \
    if (B == 0) return CTD_OK;                                                                                     \
    if (!es || !ta || !grad_out || !grad_es) return CTD_ERR_INVALID_ARG;                                           \
    DeviceGuard g(device);                                                                                         \
    if (g.status) return g.status;                                                                                 \
    return photometric_bwd_##SFX(es, ta, grad_out, grad_es, B, C, H, W, block_size, type, eps, (hipStream_t)stream); \
  }
CTD_PHOTO_ENTRY(f32, float)
CTD_PHOTO_ENTRY(f64, double)
CTD_PHOTO_ENTRY(fast_f32, float)
#undef CTD_PHOTO_ENTRY

static bool img_shape_ok(int B, int H, int W);

size_t ctd_pattern_loss_workspace_bytes(int B, int H, int W) {
  return img_shape_ok(B, H, W) ? pattern_loss_workspace_bytes(B, H, W) : 0;
}

int ctd_pattern_loss_fwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                             float* pattern_proj, float* terms, int B, int H, int W, int type, float eps,
                             void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (!img_shape_ok(B, H, W) || H < 2 || W < 2 || type < 0 || type > 3) return CTD_ERR_INVALID_ARG;
  if (!disp || !im || !pattern || !pattern_proj || !terms) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return pattern_loss_fwd_f32(disp, im, mask, pattern, pattern_proj, terms, B, H, W, type, eps, workspace, workspace_bytes,
                              (hipStream_t)stream);
}

int ctd_pattern_loss_bwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                             const float* terms, const float* grad_val, const float* grad_proj, float* grad_disp,
                             int B, int H, int W, int type, float eps, int device, void* stream) {
  if (!img_shape_ok(B, H, W) || H < 2 || W < 2 || type < 0 || type > 3) return CTD_ERR_INVALID_ARG;
  if (!disp || !im || !pattern || !terms || !grad_val || !grad_disp) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return pattern_loss_bwd_f32(disp, im, mask, pattern, terms, grad_val, grad_proj, grad_disp, B, H, W, type, eps,
                              (hipStream_t)stream);
}

size_t ctd_pattern_loss_multi_workspace_bytes(int n_levels, const ctd_pattern_level* levels) {
  return pattern_loss_multi_workspace_bytes(n_levels, levels);
}

int ctd_pattern_loss_multi_fwd_f32(int n_levels, const ctd_pattern_level* levels, float* terms, int type, float eps,
                                   void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (!terms || type < 0 || type > 3) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return pattern_loss_multi_fwd_f32(n_levels, levels, terms, type, eps, workspace, workspace_bytes, (hipStream_t)stream);
}

int ctd_pattern_loss_multi_bwd_f32(int n_levels, const ctd_pattern_level* levels, const float* terms,
                                   const float* grad_vals, int type, float eps, int device, void* stream) {
  if (!terms || !grad_vals || type < 0 || type > 3) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return pattern_loss_multi_bwd_f32(n_levels, levels, terms, grad_vals, type, eps, (hipStream_t)stream);
}

int ctd_costvol_f32(const float* im, const float* pattern, long pattern_frame_stride, float* cost, int frames, int H,
                    int W, int D, int block_size, int type, float eps, int device, void* stream) {
  if (!vol_shape_ok(frames, 1, H, W, D, block_size) || type < 0 || type > 3 || pattern_frame_stride < 0 ||
      (long)frames * D > 65535)
    return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!im || !pattern || !cost) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return costvol_f32(im, pattern, pattern_frame_stride, cost, frames, H, W, D, block_size, type, eps,
                     (hipStream_t)stream);
}

size_t ctd_costvol_workspace_bytes(int frames, int H, int W, int D, int block_size, int type, int per_frame_pattern) {
  if (!vol_shape_ok(frames, 1, H, W, D, block_size) || type < 0 || type > 3) return 0;
  if (!costvol_sep_supported(H, W, D, block_size, type)) return 0;          // the other kernels need none
  return costvol_sep_workspace_bytes(frames, H, W, D, per_frame_pattern != 0);
}

int ctd_costvol_fast_f32(const float* im, const float* pattern, long pattern_frame_stride, float* cost, int frames, int H,
                         int W, int D, int block_size, int type, float eps, void* workspace, size_t workspace_bytes,
                         int device, void* stream) {
  if (!vol_shape_ok(frames, 1, H, W, D, block_size) || type < 0 || type > 3 || pattern_frame_stride < 0)
    return CTD_ERR_INVALID_ARG;
  if (frames == 0) return CTD_OK;
  if (!im || !pattern || !cost) return CTD_ERR_INVALID_ARG;
  if (pattern_frame_stride != 0 && pattern_frame_stride != (long)H * W) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return costvol_fast_f32(im, pattern, pattern_frame_stride, cost, frames, H, W, D, block_size, type, eps, workspace,
                          workspace_bytes, (hipStream_t)stream);
}

int ctd_disp_to_depth_fwd_f32(const float* disp, float* depth, long n, float baseline_focal, int device, void* stream) {
  if (n < 0) return CTD_ERR_INVALID_ARG;
  if (n == 0) return CTD_OK;
  if (!disp || !depth) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return disp_to_depth_fwd_f32(disp, depth, n, baseline_focal, (hipStream_t)stream);
}

int ctd_idx_to_depth_f32(const int64_t* idx, float* depth, long n, float baseline_focal, float disp_offset, int device,
                         void* stream) {
  if (n < 0) return CTD_ERR_INVALID_ARG;
  if (n == 0) return CTD_OK;
  if (!idx || !depth) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return idx_to_depth_f32(idx, depth, n, baseline_focal, disp_offset, (hipStream_t)stream);
}

int ctd_disp_to_depth_bwd_f32(const float* disp, const float* grad_depth, float* grad_disp, long n,
                              float baseline_focal, int device, void* stream) {
  if (n < 0) return CTD_ERR_INVALID_ARG;
  if (n == 0) return CTD_OK;
  if (!disp || !grad_depth || !grad_disp) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return disp_to_depth_bwd_f32(disp, grad_depth, grad_disp, n, baseline_focal, (hipStream_t)stream);
}

static bool img_shape_ok(int B, int H, int W) {
  return B > 0 && H > 0 && W > 0 && B <= 65535 && (double)B * H * W < 2147483648.0;
}

size_t ctd_disparity_loss_workspace_bytes(int B, int H, int W) {
  return img_shape_ok(B, H, W) ? disparity_loss_workspace_bytes(B, H, W) : 0;
}

int ctd_disparity_loss_fwd_f32(const float* disp, const float* edge, float* loss, int B, int H, int W, void* workspace,
                               size_t workspace_bytes, int device, void* stream) {
  if (!img_shape_ok(B, H, W) || !disp || !loss) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return disparity_loss_fwd_f32(disp, edge, loss, B, H, W, workspace, workspace_bytes, (hipStream_t)stream);
}

int ctd_disparity_loss_bwd_f32(const float* disp, const float* edge, const float* grad_loss, float* grad_disp,
                               float* grad_edge, int B, int H, int W, void* workspace, size_t workspace_bytes,
                               int device, void* stream) {
  if (!img_shape_ok(B, H, W) || !disp || !grad_loss || !grad_disp) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return disparity_loss_bwd_f32(disp, edge, grad_loss, grad_disp, grad_edge, B, H, W, workspace, workspace_bytes,
                                (hipStream_t)stream);
}

size_t ctd_geometric_workspace_bytes(int B, int H, int W) {
  return img_shape_ok(B, H, W) ? geometric_workspace_bytes(B, H, W) : 0;
}

int ctd_geometric_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                          const float* t0, const float* R1, const float* t1, float* loss, int accumulate, int B, int H,
                          int W, float clamp, void* workspace, size_t workspace_bytes, int device, void* stream) {
  if (!img_shape_ok(B, H, W) || H < 2 || W < 2 || !depth0 || !depth1 || !ray || !K || !R0 || !t0 || !R1 || !t1 || !loss)
    return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return geometric_fwd_f32(depth0, depth1, ray, K, R0, t0, R1, t1, loss, accumulate, B, H, W, clamp, workspace,
                           workspace_bytes, (hipStream_t)stream);
}

int ctd_geometric_sym_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                              const float* t0, const float* R1, const float* t1, float* loss, int B, int H, int W,
                              float clamp, void* workspace, size_t workspace_bytes, unsigned* ticket, int device,
                              void* stream) {
  if (!img_shape_ok(B, H, W) || H < 2 || W < 2 || !depth0 || !depth1 || !ray || !K || !R0 || !t0 || !R1 || !t1 || !loss ||
      !ticket)
    return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return geometric_sym_fwd_f32(depth0, depth1, ray, K, R0, t0, R1, t1, loss, B, H, W, clamp, workspace, workspace_bytes,
                               ticket, (hipStream_t)stream);
}

int ctd_geometric_bwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                          const float* t0, const float* R1, const float* t1, const float* grad_loss,
                          float* grad_depth0, int accumulate0, float* grad_depth1, int B, int H, int W, float clamp,
                          int device, void* stream) {
  if (!img_shape_ok(B, H, W) || H < 2 || W < 2 || !depth0 || !depth1 || !ray || !K || !R0 || !t0 || !R1 || !t1 ||
      !grad_loss || !grad_depth0 || !grad_depth1)
    return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return geometric_bwd_f32(depth0, depth1, ray, K, R0, t0, R1, t1, grad_loss, grad_depth0, accumulate0, grad_depth1, B,
                           H, W, clamp, (hipStream_t)stream);
}

int ctd_nn_f32(const float* in0, const float* in1, long n0, long n1, int64_t* out, int device, void* stream) {
  if (n0 < 0 || n1 < 0) return CTD_ERR_INVALID_ARG;
  if (n0 == 0) return CTD_OK;
  if (!in0 || !out || (n1 > 0 && !in1)) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return nn_f32(in0, in1, n0, n1, out, (hipStream_t)stream);
}

int ctd_nn_f64(const double* in0, const double* in1, long n0, long n1, int64_t* out, int device, void* stream) {
  if (n0 < 0 || n1 < 0) return CTD_ERR_INVALID_ARG;
  if (n0 == 0) return CTD_OK;
  if (!in0 || !out || (n1 > 0 && !in1)) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return nn_f64(in0, in1, n0, n1, out, (hipStream_t)stream);
}

int ctd_crosscheck(const int64_t* in0, const int64_t* in1, long n0, long n1, uint8_t* out, int device, void* stream) {
  if (n0 < 0 || n1 < 0) return CTD_ERR_INVALID_ARG;
  if (n0 == 0) return CTD_OK;
  if (!in0 || !out || (n1 > 0 && !in1)) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return crosscheck_i64(in0, in1, n0, n1, out, (hipStream_t)stream);
}

int ctd_proj_nn_f32(const float* xyz0, const float* xyz1, const float* K, int B, int H, int W, int patch_size,
                    int64_t* out, int device, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || patch_size < 0) return CTD_ERR_INVALID_ARG;
  if (B == 0) return CTD_OK;
  if (!xyz0 || !xyz1 || !K || !out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return proj_nn_f32(xyz0, xyz1, K, B, H, W, patch_size, out, (hipStream_t)stream);
}

int ctd_proj_nn_f64(const double* xyz0, const double* xyz1, const double* K, int B, int H, int W, int patch_size,
                    int64_t* out, int device, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || patch_size < 0) return CTD_ERR_INVALID_ARG;
  if (B == 0) return CTD_OK;
  if (!xyz0 || !xyz1 || !K || !out) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return proj_nn_f64(xyz0, xyz1, K, B, H, W, patch_size, out, (hipStream_t)stream);
}

int ctd_render_mesh_proj_f32(const float* verts, const float* colors, int n_verts, const int* faces, int n_faces,
                             const float* cam, int cam_width, int cam_height, const float* proj, int proj_width,
                             int proj_height, const float* shader, const float* pattern, float d_alpha, float d_beta,
                             float* depth, float* color, float* normal, int device, void* stream) {
  if (n_verts < 0 || n_faces < 0 || cam_width <= 0 || cam_height <= 0 || proj_width <= 0 || proj_height <= 0 ||
      (double)cam_width * cam_height * 3 >= 2147483648.0)
    return CTD_ERR_INVALID_ARG;
  if (!cam || !proj || !shader || !pattern || !color || (n_faces > 0 && (!verts || !colors || !faces)))
    return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return render_mesh_proj_f32(verts, colors, faces, n_faces, cam, cam_width, cam_height, proj, proj_width, proj_height,
                              shader, pattern, d_alpha, d_beta, depth, color, normal, (hipStream_t)stream);
}

int ctd_render_mesh_f32(const float* verts, const float* colors, const float* normals, int n_verts, const int* faces,
                        int n_faces, const float* cam, int cam_width, int cam_height, const float* shader, float* depth,
                        float* color, float* normal, int device, void* stream) {
  if (n_verts < 0 || n_faces < 0 || cam_width <= 0 || cam_height <= 0 || (double)cam_width * cam_height * 3 >= 2147483648.0)
    return CTD_ERR_INVALID_ARG;
  if (!cam || !shader || (n_faces > 0 && (!verts || !faces))) return CTD_ERR_INVALID_ARG;
  if (n_faces > 0 && ((color && !colors) || ((color || normal) && !normals))) return CTD_ERR_INVALID_ARG;
  DeviceGuard g(device);
  if (g.status) return g.status;
  return render_mesh_f32(verts, colors, normals, faces, n_faces, cam, cam_width, cam_height, shader, depth, color, normal,
                         (hipStream_t)stream);
}

}  // extern "C"
