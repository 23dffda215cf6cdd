// ctd_internal.h -- C++ entry points behind the C ABI (one per kernel family).
#pragma once
#include "ctd_common.h"

namespace ctd {

// kernel timing hooks (ctd_api.hip)
bool timing_enabled();
void timing_begin(hipStream_t stream);          // records the start event of the dominant kernel
void timing_end(hipStream_t stream, int columns);

// ncc_exact.hip
size_t ncc_exact_workspace_bytes(int frames, int C, int H, int W, int D, int bs, bool per_frame_pattern);
int ncc_exact_f32(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H,
                  int W, int D, int bs, void* workspace, size_t workspace_bytes, hipStream_t stream);
int ncc_exact_f64(const double* in0, const double* in1, long in1_frame_stride, double* out, int frames, int C, int H,
                  int W, int D, int bs, void* workspace, size_t workspace_bytes, hipStream_t stream);
int ncc_exact_argmax_f32(const float* in0, const float* in1, long in1_frame_stride, float* vol_out, int64_t* idx,
                         float* best, int frames, int H, int W, int D, int bs, void* workspace,
                         size_t workspace_bytes, hipStream_t stream);
int argmax_disp_f32(const float* vol, int64_t* idx, float* best, int frames, int D, int H, int W, hipStream_t stream);

// ncc_fast.hip
// Buffers of the in-kernel ranking (all inside the caller's workspace, laid out by ncc_fast_f32).
struct RankPlan {
  float eps;                  // in: re-ranking margin requested by the caller
  int64_t* idx;               // in: [frames][H][W] indices (written by the all-D kernel, corrected by the resolve pass)
  float* best;                // in: [frames][H][W] best scores, or null (out: best_scratch then)
  unsigned char* flags;       // [frames][H][W] 1 = pixel is on the work list (written by the all-D kernel)
  WorkList work;              // work list of the pixels the exact re-scoring has to settle (counters cleared by the
                              // pre-pass kernel)
  float* best_scratch;        // [frames][H][W] best score when the caller does not ask for it
  // what the tail kernel (runs | decode | resolve roles, argmax_rerank.hip) needs of the volume pass's workspace
  const float* run_vals;      // [frames][H][D] exact values of the listed fully clamped runs
  const unsigned long long* run_rows;
  unsigned* counters;         // [0] listed frame windows (the tail kernel clears it), [1] listed pattern windows, [2] listed run
                              // rows, [3] copy of [0] made by the ranked fix-up kernel for the tail kernel
  const unsigned long long *flag_a, *flag_b;   // lists of the listed frame / pattern windows
  const float* v1;            // pattern reciprocal-deviation planes (0 = listed window), row pitch W1, column x at x + xoff
  int W1, xoff;
  size_t bytes;               // workspace bytes up to the end of these buffers
};
size_t ncc_fast_workspace_bytes(int frames, int C, int H, int W, int D, int bs, bool per_frame_pattern);
bool ncc_fast_rank_supported(int C, int H, int W, int D, int bs);
size_t ncc_fast_rank_workspace_bytes(int frames, int H, int W, int D, bool per_frame_pattern);
void ncc_fast_rank_offsets(int frames, int H, int W, int D, bool per_frame_pattern, size_t* off);
// A fused call: the frames arrive raw and `in0` of ncc_fast_f32 is the buffer their LCN goes to (lcn_stream.hip).
struct FusedLcn {
  const float* raw;           // [frames][H][W] raw frames
  float* stds;                // [frames][H][W] LCN deviation output (the LCN output itself is `in0`)
  int radius;
  float eps;
  bool exact;                 // f64 box sums + the reference's f32 tail (the oracle's bits) | f32 sums, v_rcp / v_sqrt tail
};
int ncc_fast_f32(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int C, int H, int W,
                 int D, int bs, void* workspace, size_t workspace_bytes, RankPlan* rank, bool pattern_prepared,
                 hipStream_t stream, const FusedLcn* fused = nullptr);
int ncc_fast_prepare_pattern_f32(const float* in1, long in1_frame_stride, int frames, int C, int H, int W, int D, int bs,
                                 void* workspace, size_t workspace_bytes, hipStream_t stream);
int ncc_fast_fixup_ranked(const float* in0, const float* in1, long in1_frame_stride, float* out, int frames, int H, int W,
                          int D, int bs, void* workspace, const RankPlan& rank, const float* best, hipStream_t stream);

// separable block SAD / MSE cost volume through the all-D pipeline (block 9, W % 4 == 0); workspace = padded operand planes
bool costvol_sep_supported(int H, int W, int D, int bs, int type);
size_t costvol_sep_workspace_bytes(int frames, int H, int W, int D, bool per_frame_pattern);
int costvol_sep_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W, int D,
                    int type, void* workspace, size_t workspace_bytes, hipStream_t stream);

// argmax_rerank.hip
int argmax_rerank_f32(const float* vol, const float* in0, const float* in1, long in1_frame_stride, int64_t* idx,
                      float* best, int frames, int D, int H, int W, int bs, float eps, void* workspace,
                      size_t workspace_bytes, bool counter_cleared, hipStream_t stream);
int rank_tail_f32(const RankPlan& rp, float* vol, const float* in0, const float* in1, long in1_frame_stride,
                  int64_t* idx, float* best, int frames, int D, int H, int W, int bs, hipStream_t stream);

// photometric.hip
int photometric_fwd_f32(const float* es, const float* ta, float* out, int B, int C, int H, int W, int bs, int type,
                        float eps, hipStream_t s);
int photometric_fwd_f64(const double* es, const double* ta, double* out, int B, int C, int H, int W, int bs, int type,
                        float eps, hipStream_t s);
int photometric_bwd_f32(const float* es, const float* ta, const float* go, float* gi, int B, int C, int H, int W,
                        int bs, int type, float eps, hipStream_t s);
int photometric_bwd_f64(const double* es, const double* ta, const double* go, double* gi, int B, int C, int H, int W,
                        int bs, int type, float eps, hipStream_t s);
int costvol_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W, int D,
                int bs, int type, float eps, hipStream_t stream);

// photometric_fast.hip
int photometric_fwd_fast_f32(const float* es, const float* ta, float* out, int B, int C, int H, int W, int bs, int type,
                             float eps, hipStream_t s);
int photometric_bwd_fast_f32(const float* es, const float* ta, const float* go, float* gi, int B, int C, int H, int W,
                             int bs, int type, float eps, hipStream_t s);

size_t pattern_loss_workspace_bytes(int B, int H, int W);
int pattern_loss_fwd_f32(const float* disp, const float* im, const float* mask, const float* pattern, float* proj,
                         float* terms, int B, int H, int W, int type, float eps, void* ws, size_t ws_bytes,
                         hipStream_t s);
int pattern_loss_bwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                         const float* terms, const float* grad_val, const float* grad_proj, float* grad_disp, int B,
                         int H, int W, int type, float eps, hipStream_t s);

size_t pattern_loss_multi_workspace_bytes(int n_levels, const ctd_pattern_level* levels);
int pattern_loss_multi_fwd_f32(int n_levels, const ctd_pattern_level* levels, float* terms, int type, float eps, void* ws,
                               size_t ws_bytes, hipStream_t stream);
int pattern_loss_multi_bwd_f32(int n_levels, const ctd_pattern_level* levels, const float* terms, const float* grad_vals,
                               int type, float eps, hipStream_t stream);

int costvol_fast_f32(const float* im, const float* pat, long pat_frame_stride, float* cost, int frames, int H, int W,
                     int D, int bs, int type, float eps, void* workspace, size_t workspace_bytes, hipStream_t stream);

// lcn.hip
int lcn_f32(const float* x, float* y, float* stds, int N, int H, int W, int radius, float eps, hipStream_t stream);
int lcn_fast_f32(const float* x, float* y, float* stds, int N, int H, int W, int radius, float eps, hipStream_t stream);

int lcn_datagen_f32(const float* img, float* out, float* out_std, int N, int H, int W, int ks, float eps,
                    hipStream_t stream);

// losses.hip
int disp_to_depth_fwd_f32(const float* disp, float* depth, long n, float bf, hipStream_t s);
int idx_to_depth_f32(const int64_t* idx, float* depth, long n, float bf, float offset, hipStream_t s);
int disp_to_depth_bwd_f32(const float* disp, const float* go, float* gi, long n, float bf, hipStream_t s);
size_t disparity_loss_workspace_bytes(int B, int H, int W);
int disparity_loss_fwd_f32(const float* disp, const float* edge, float* loss, int B, int H, int W, void* ws,
                           size_t ws_bytes, hipStream_t s);
int disparity_loss_bwd_f32(const float* disp, const float* edge, const float* grad_loss, float* grad_disp,
                           float* grad_edge, int B, int H, int W, void* ws, size_t ws_bytes, hipStream_t s);
size_t geometric_workspace_bytes(int B, int H, int W);
int geometric_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                      const float* t0, const float* R1, const float* t1, float* loss, int accumulate, int B, int H,
                      int W, float clamp, void* ws, size_t ws_bytes, hipStream_t s);
int geometric_sym_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                          const float* t0, const float* R1, const float* t1, float* loss, int B, int H, int W, float clamp,
                          void* ws, size_t ws_bytes, unsigned* ticket, hipStream_t s);
int geometric_bwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K, const float* R0,
                      const float* t0, const float* R1, const float* t1, const float* grad_loss, float* grad_depth0,
                      int accumulate0, float* grad_depth1, int B, int H, int W, float clamp, hipStream_t s);

// nn_ops.hip
int nn_f32(const float* in0, const float* in1, long n0, long n1, int64_t* out, hipStream_t s);
int nn_f64(const double* in0, const double* in1, long n0, long n1, int64_t* out, hipStream_t s);
int crosscheck_i64(const int64_t* in0, const int64_t* in1, long n0, long n1, uint8_t* out, hipStream_t s);
int proj_nn_f32(const float* xyz0, const float* xyz1, const float* K, long B, long H, long W, int patch_size,
                int64_t* out, hipStream_t s);
int proj_nn_f64(const double* xyz0, const double* xyz1, const double* K, long B, long H, long W, int patch_size,
                int64_t* out, hipStream_t s);

// render.hip
int render_mesh_proj_f32(const float* verts, const float* colors, const int* faces, int n_faces, const float* cam_p,
                         int cam_w, int cam_h, const float* proj_p, int proj_w, int proj_h, const float* shader,
                         const float* pattern, float d_alpha, float d_beta, float* depth, float* color, float* normal,
                         hipStream_t stream);
int render_mesh_f32(const float* verts, const float* colors, const float* normals, const int* faces, int n_faces,
                    const float* cam_p, int cam_w, int cam_h, const float* shader, float* depth, float* color, float* normal,
                    hipStream_t stream);

}  // namespace ctd
