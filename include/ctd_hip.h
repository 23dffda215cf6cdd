/*
 * ctd_hip.h -- C ABI of libctd_hip.so, the MI355X (gfx950) implementation of the
 * disparity hot path of autonomousvision/connecting_the_dots.
 *
 * This is the drop-in boundary: each entry point replaces one pybind11 function of
 * the reference's CUDA extension (torchext/ext/ext_cuda.cpp:126-135) or fuses a chain
 * of stock PyTorch ops of model/networks.py.  Signatures carry plain device pointers,
 * sizes, a device ordinal and a hipStream_t (as void*); no torch / pybind types.
 *
 * Conventions
 *   - all tensors are dense, row-major, already resident in device memory;
 *   - the callee never allocates, frees or synchronises: outputs and workspaces are
 *     provided by the caller (the reference allocated outputs with ATen inside the
 *     call, ext_cuda.cpp:81,100,119; here the Python wrapper does it with torch.empty);
 *   - `device` is the HIP device ordinal the pointers live on (-1 = current device);
 *     `stream` is the hipStream_t to launch on (NULL = default stream).  The reference
 *     launched on the legacy default stream with no device guard (common_cuda.h:168);
 *   - return value: 0 on success, otherwise a ctd_status code (never exit(), unlike
 *     common_cuda.h:11-20); ctd_status_string() names it;
 *   - kernels are deterministic (no floating-point atomics) except ctd_geometric_bwd_f32, whose
 *     bilinear scatter into grad_depth1 uses float atomics like ATen's grid_sample backward.
 */
#ifndef CTD_HIP_H
#define CTD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum ctd_status {
  CTD_OK = 0,
  CTD_ERR_INVALID_ARG = 1,     /* bad size / null pointer / unsupported parameter        */
  CTD_ERR_WORKSPACE = 2,       /* workspace too small                                     */
  CTD_ERR_UNSUPPORTED = 3,     /* valid request this build has no kernel for              */
  CTD_ERR_HIP = 1000           /* 1000 + hipError_t of the failing runtime call           */
};

int ctd_version(void);                       /* ABI version, currently 5 (5: + ctd_lcn_xcorrvol_argmax_f32 / ctd_lcn_xcorrvol_supported; 4: ctd_costvol_fast_f32 takes a workspace, ctd_costvol_workspace_bytes; 2: ranked argmax inside the all-D volume kernel, its workspace is ctd_xcorrvol_argmax_workspace_bytes(); 3: + ctd_xcorrvol_pattern_prepare_f32 / CTD_PATTERN_PREPARED, ctd_geometric_sym_fwd_f32) */
const char* ctd_status_string(int status);

/* (Bench instrumentation -- per-kernel device timing of the volume kernel -- is declared in ctd_hip_bench.h: it is not
 * part of the drop-in interface.) */

/* photometric loss types -- torchext/ext/ext.h:196-199, torchext/functions.py:106-118 */
#define CTD_PHOTOMETRIC_MSE 0
#define CTD_PHOTOMETRIC_SAD 1
#define CTD_PHOTOMETRIC_CENSUS_MSE 2
#define CTD_PHOTOMETRIC_CENSUS_SAD 3

/* algorithm selector of the NCC volume */
#define CTD_NCC_EXACT 0   /* reference operation order, bit-identical to ext_cpu.cpp        */
#define CTD_NCC_FAST 1    /* separable window sums; |a-b| <= 1e-5|b| + 1e-6 of the exact (C > 1: the sum of the
                             channels' bounds -- the volume is the sum of per-channel NCCs, and where two channels
                             cancel no f32 order keeps 1e-5 of the sum)                                     */

/* --------------------------------------------------------------------------------------
 * Zero-mean NCC block-matching volume.
 * Replaces  xcorrvol_cuda(in0, in1, n_disps, block_size)  -- ext_cuda.cpp:73-86,
 * kernel ext_kernel.cu:40-50, functor ext.h:120-191.
 *
 *   in0  [frames][C][H][W]   IR frames
 *   in1  [C][H][W]           pattern, shared by all frames when in1_frame_stride == 0,
 *                            else in1 + f*in1_frame_stride (elements) is frame f's pattern
 *   out  [frames][D][H][W]
 * The reference op has no batch axis (functions.py:73-74 is called per frame);
 * frames == 1 reproduces it exactly, frames > 1 is the same op applied per frame in
 * one launch.  `algo` = CTD_NCC_EXACT | CTD_NCC_FAST.
 * Workspace: ctd_xcorrvol_workspace_bytes() bytes, 256-byte aligned, contents scratch.
 * -------------------------------------------------------------------------------------- */
size_t ctd_xcorrvol_workspace_bytes(int frames, int C, int H, int W, int D, int block_size, int algo);

int ctd_xcorrvol_f32(const float* in0, const float* in1, long in1_frame_stride, float* out,
                     int frames, int C, int H, int W, int D, int block_size, int algo,
                     void* workspace, size_t workspace_bytes, int device, void* stream);

/* Pattern half of the fast path's pre-pass, ONCE per pattern: the reference prepares the pattern once per run
 * (model/exp_synph.py:64-71: LCN of the pattern in the Worker's constructor), its xcorrvol op re-reads it on every call.
 * Writes the pattern's window statistics, its list of windows for the fix-up pass and run rows into `workspace`
 * (ctd_xcorrvol_argmax_workspace_bytes() for the same frames / shape: the layout depends on all of them).  Afterwards
 * ctd_xcorrvol_f32 / ctd_xcorrvol_argmax_f32 calls with algo = CTD_NCC_FAST | CTD_PATTERN_PREPARED, the SAME workspace,
 * in1, in1_frame_stride, frames and shape skip that half.  The caller keeps the workspace to these calls in between. */
#define CTD_PATTERN_PREPARED 0x100
int ctd_xcorrvol_pattern_prepare_f32(const float* in1, long in1_frame_stride, int frames, int C, int H,
                                     int W, int D, int block_size, void* workspace,
                                     size_t workspace_bytes, int device, void* stream);

int ctd_xcorrvol_f64(const double* in0, const double* in1, long in1_frame_stride, double* out,
                     int frames, int C, int H, int W, int D, int block_size,
                     void* workspace, size_t workspace_bytes, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * argmax over the disparity axis:  idx = torch.argmax(vol, dim=0)  (first index wins
 * ties), best = vol.max(0).  No reference code.
 *   vol [frames][D][H][W] -> idx int64 [frames][H][W], best f32 [frames][H][W] (may be NULL)
 * -------------------------------------------------------------------------------------- */
int ctd_argmax_disp_f32(const float* vol, int64_t* idx, float* best, int frames, int D, int H,
                        int W, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * NCC volume + argmax over disparity (C == 1 only).
 * CTD_NCC_EXACT: fused, vol_out may be NULL (no volume written) or [frames][D][H][W]
 *   (written as well); indices equal torch.argmax(xcorrvol_cpu(...), 0) bit for bit and
 *   best is the reference-order score.
 * CTD_NCC_FAST: every disparity whose fast score lies within `rerank_eps` of the pixel's
 *   best is re-scored in reference order, so the indices are those of the reference-order
 *   volume whenever |fast - exact| <= rerank_eps / 2.  Where ctd_xcorrvol_rank_supported()
 *   holds (block 9, W % 4 == 0, D <= 512) the all-D volume kernel ranks the scores itself:
 *   one workgroup walks every disparity of its (column tile, band, frame) and keeps the best
 *   and the runner-up of every pixel in LDS, so idx / best leave that kernel directly (no
 *   partial planes, no second pass over the volume) and vol_out may be NULL: nothing is
 *   materialised then.  Otherwise vol_out is required (CTD_ERR_INVALID_ARG if NULL) and
 *   ranked in one more pass.  best = the fast score of idx to within 2.4e-7 absolute (the
 *   ranking keys are fixed point, 2^-21 apart; it is the reference-order score for re-scored
 *   pixels when no volume is written).
 *   rerank_eps < 0 = plain argmax of the fast scores, no exact re-scoring: defined on a
 *   materialised volume (vol_out != NULL: the patched volume is ranked in one more pass);
 *   without a volume a negative value is taken as 0.
 * Workspace: ctd_xcorrvol_argmax_workspace_bytes().
 * -------------------------------------------------------------------------------------- */
int ctd_xcorrvol_rank_supported(int C, int H, int W, int D, int block_size);
/* Inspection aid for tests / tools: how a ranked ctd_xcorrvol_argmax_f32 call is laid out.  offsets[0] = rows per band
 * of the all-D kernel, offsets[4] = its passes over the disparities; byte offsets inside the workspace of
 * offsets[1] = flag bytes [frames][H][W], offsets[2] = the work-list counters (16 x u32, 256 bytes apart, one per key =
 * image row & 15), offsets[3] = the work list (i64 flat pixel indices, 16 segments of ceil(frames * H / 16) * W entries,
 * one per key).  `offsets` has 5 entries. */
int ctd_xcorrvol_rank_layout(int frames, int H, int W, int D, int per_frame_pattern, size_t* offsets);
size_t ctd_xcorrvol_argmax_workspace_bytes(int frames, int C, int H, int W, int D, int block_size, int algo);
int ctd_xcorrvol_argmax_f32(const float* in0, const float* in1, long in1_frame_stride,
                            float* vol_out, int64_t* idx, float* best, int frames, int C, int H,
                            int W, int D, int block_size, int algo, float rerank_eps,
                            void* workspace, size_t workspace_bytes, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * Photometric block loss, forward and backward.
 * Replaces photometric_loss_forward / photometric_loss_backward -- ext_cuda.cpp:92-123,
 * kernels ext_kernel.cu:54-112, functors ext.h:201-344.
 *   es, ta [B][C][H][W] -> out [B][1][H][W];  grad_out [B][1][H][W] -> grad_es [B][C][H][W]
 * `eps` is a C float for both dtypes, as in the reference binding (ext_cuda.cpp:92).
 * The backward is a deterministic gather (no atomics, no pre-zeroed buffer needed); the
 * reference scatter-adds with atomicAdd into at::zeros (ext.h:315,338-339).
 * -------------------------------------------------------------------------------------- */
int ctd_photometric_fwd_f32(const float* es, const float* ta, float* out, int B, int C, int H,
                            int W, int block_size, int type, float eps, int device, void* stream);
int ctd_photometric_bwd_f32(const float* es, const float* ta, const float* grad_out,
                            float* grad_es, int B, int C, int H, int W, int block_size, int type,
                            float eps, int device, void* stream);
/* Tolerance-level variants (f32, odd block sizes 3/5/7/9; CTD_ERR_UNSUPPORTED otherwise): same functions with
 * the summation order left free and v_rsq_f32 in place of the correctly rounded sqrt/divide chains;
 * |fast - reference| <= 1e-5 |reference| + 1e-6 (for the gradient: of the gradient's scale).  The backward
 * evaluates one term per pixel pair using K(q,p) = -K(p,q) (see photometric_fast.hip), no atomics. */
int ctd_photometric_fwd_fast_f32(const float* es, const float* ta, float* out, int B, int C, int H,
                                 int W, int block_size, int type, float eps, int device,
                                 void* stream);
int ctd_photometric_bwd_fast_f32(const float* es, const float* ta, const float* grad_out,
                                 float* grad_es, int B, int C, int H, int W, int block_size,
                                 int type, float eps, int device, void* stream);
int ctd_photometric_fwd_f64(const double* es, const double* ta, double* out, int B, int C, int H,
                            int W, int block_size, int type, float eps, int device, void* stream);
int ctd_photometric_bwd_f64(const double* es, const double* ta, const double* grad_out,
                            double* grad_es, int B, int C, int H, int W, int block_size, int type,
                            float eps, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * SAD / soft-census cost volume, defined by composition of reference ops:
 *   cost[d] = photometric_loss_forward(es = P_d, ta = I)[0,0],  P_d[h,x] = P[h, clamp(x-d)]
 *   im [frames][H][W], pattern [H][W] (stride as above) -> cost [frames][D][H][W]
 * -------------------------------------------------------------------------------------- */
int ctd_costvol_f32(const float* im, const float* pattern, long pattern_frame_stride, float* cost,
                    int frames, int H, int W, int D, int block_size, int type, float eps,
                    int device, void* stream);
/* tolerance-level variant (odd block sizes 3/5/7/9): |fast - exact| <= 1e-5 |exact| + 1e-6.
 * SAD / MSE with block 9 and W % 4 == 0 are evaluated as a replicate-border 9 x 9 box filter of the per-pixel plane
 * |P[r][clamp(c - d)] - I[r][c]| (one subtract per output instead of 81) by the NCC volume kernel's pipeline; that path
 * needs `workspace` (ctd_costvol_workspace_bytes(), 16-byte aligned; 0 bytes / NULL: the LDS-tiled 81-tap kernel runs
 * instead).  The census types use the census-transform kernel and no workspace.  (Signature since ABI version 4.)
 * pattern_frame_stride: 0 (one pattern for all frames) or H * W (dense per-frame patterns); anything else is
 * CTD_ERR_INVALID_ARG here -- ctd_costvol_f32 takes arbitrary strides. */
size_t ctd_costvol_workspace_bytes(int frames, int H, int W, int D, int block_size, int type, int per_frame_pattern);
int ctd_costvol_fast_f32(const float* im, const float* pattern, long pattern_frame_stride, float* cost,
                    int frames, int H, int W, int D, int block_size, int type, float eps,
                    void* workspace, size_t workspace_bytes, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * Local contrast normalisation, fused.  Replaces the op chain of LCN.tforward,
 * model/networks.py:507-533 (ReflectionPad2d + two all-ones Conv2d + 6 elementwise ops).
 *   x [N][1][H][W] -> y = (x-avg)/std, std  (both [N][1][H][W]);  radius < min(H, W)
 * -------------------------------------------------------------------------------------- */
int ctd_lcn_f32(const float* x, float* y, float* std_out, int N, int H, int W, int radius,
                float eps, int device, void* stream);
/* tolerance-level variant (radius 1 .. 7, else CTD_ERR_UNSUPPORTED): f32 sliding-window box sums instead of f64 ones, of
 * samples centred per 64 x 16 tile (0 where the tile reaches zero, else the tile's mean).  Every output within
 * 1e-5 |b| + 1e-6 of ctd_lcn_f32 and of the reference's networks.LCN (whose conv2d summation order is unspecified) where a
 * window's variance is not small against its mean square about that centre -- uniform / textured frames, frames with a DC
 * level, dots on a zero background; windows of a low-noise non-zero level with no sample in them sit on the 1e-6 variance
 * floor, where ANY f32 one-pass variance (the reference's included) is good to E[(x - c)^2] * 2^-24 per rounding: up to
 * 3e-6 absolute of std there (tools/fuzz_lcn.py). */
int ctd_lcn_fast_f32(const float* x, float* y, float* std_out, int N, int H, int W, int radius,
                     float eps, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * LCN of the frames + NCC volume + argmax over disparity in one call (C == 1): the step
 *   ir = LCN(raw)  (model/networks.py:507-533, applied to the input images at model/exp_synph.py:84-91)
 *   vol = xcorrvol(ir, pattern)  (ext_cuda.cpp:73-86);  idx = argmax_d vol
 * of a matcher built on the reference's ops, without the round trip of the LCN output through memory before the
 * matcher's window statistics: one streaming kernel (a wavefront per 232-column strip and band of rows) reads the raw
 * frames once and writes lcn_out, std_out and the frame-side planes of the fast NCC path.
 *   raw [frames][1][H][W] -> lcn_out, std_out [frames][1][H][W] (both required);  in1 = the LCN'd pattern, as for
 *   ctd_xcorrvol_argmax_f32, whose remaining arguments, outputs, workspace and CTD_PATTERN_PREPARED rule apply unchanged
 *   (algo = CTD_NCC_FAST [| CTD_PATTERN_PREPARED]).
 * lcn_algo: CTD_LCN_EXACT -- f64 box sums and the reference's f32 elementwise tail: lcn_out / std_out carry the bits of
 *   ctd_lcn_f32 (f64 sums of f32 samples are exact in any order unless one window spans more than 2^29 in magnitude);
 *   CTD_LCN_FAST -- f32 sums of samples centred by one constant per wavefront, v_rcp / v_sqrt tail: tolerance level,
 *   and only for windows whose variance is not small against their mean square (an f32 one-pass variance cannot do
 *   better: E[x^2] - avg^2 cancels; the reference's own f32 conv2d has the same limit).
 * Supported where ctd_lcn_xcorrvol_supported() says so (radius 5, block 9, W % 4 == 0, W >= 16, H >= 11 and the ranked
 * fast path of ctd_xcorrvol_rank_supported()); CTD_ERR_UNSUPPORTED otherwise (call ctd_lcn_f32 + ctd_xcorrvol_argmax_f32).
 * -------------------------------------------------------------------------------------- */
#define CTD_LCN_EXACT 0
#define CTD_LCN_FAST 1
int ctd_lcn_xcorrvol_supported(int H, int W, int D, int radius, int block_size);
int ctd_lcn_xcorrvol_argmax_f32(const float* raw, float* lcn_out, float* std_out, int radius, float lcn_eps, int lcn_algo,
                                const float* in1, long in1_frame_stride, float* vol_out, int64_t* idx, float* best,
                                int frames, int H, int W, int D, int block_size, int algo, float rerank_eps,
                                void* workspace, size_t workspace_bytes, int device, void* stream);

/* Data-generation variant, data/lcn/lcn.pyx:16-58 (`lcn.normalize(img, kernel_size, epsilon)`): two-pass window mean
 * / std in f32 in the Cython loop's tap order (bit-identical), out = (x - mean) / (std + eps), out_std = raw std, a
 * border of width kernel_size stays zero.  img, out, out_std [N][H][W]. */
int ctd_lcn_datagen_f32(const float* img, float* out, float* out_std, int N, int H, int W,
                        int kernel_size, float eps, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * DispToDepth.tforward, model/networks.py:313-321, and its backward.
 *   depth = (baseline*focal) / (relu(disp) + 1e-12)
 * -------------------------------------------------------------------------------------- */
int ctd_disp_to_depth_fwd_f32(const float* disp, float* depth, long n, float baseline_focal,
                              int device, void* stream);
int ctd_disp_to_depth_bwd_f32(const float* disp, const float* grad_depth, float* grad_disp, long n,
                              float baseline_focal, int device, void* stream);
/* Same forward for a disparity given as the argmax index of ctd_xcorrvol_argmax_f32 (int64) plus a constant offset:
 * depth = (baseline*focal) / (relu(float(idx) + disp_offset) + 1e-12).  Not differentiable (additive, no reference
 * counterpart: saves the int64 -> float and offset passes between the matcher and the geometric loss). */
int ctd_idx_to_depth_f32(const int64_t* idx, float* depth, long n, float baseline_focal, float disp_offset,
                         int device, void* stream);

/* --------------------------------------------------------------------------------------
 * Edge-aware disparity loss: SobelFilter (5x5, replicate pad) + DisparityLoss.tforward,
 * model/networks.py:380-412, 537-565, fused.
 *   disp [B][1][H][W], edge [B][1][H][W] or NULL  ->  loss[0] (device scalar, mean over B*H*W)
 *   with edge   : mean(-log(clamp((1-e)/b0*exp(-g/b0) + e/b1*exp(-g/b1), 1e-4)))
 *   without edge: mean(clamp(g, 0, 1)),   g = sqrt(gx^2 + gy^2 + 1e-8)
 * Backward: grad_loss is a device scalar; grad_disp always written, grad_edge optional (NULL).
 * Workspace: ctd_disparity_loss_workspace_bytes() bytes, shared layout for both directions.
 * -------------------------------------------------------------------------------------- */
size_t ctd_disparity_loss_workspace_bytes(int B, int H, int W);
int ctd_disparity_loss_fwd_f32(const float* disp, const float* edge, float* loss, int B, int H, int W,
                               void* workspace, size_t workspace_bytes, int device, void* stream);
int ctd_disparity_loss_bwd_f32(const float* disp, const float* edge, const float* grad_loss,
                               float* grad_disp, float* grad_edge, int B, int H, int W,
                               void* workspace, size_t workspace_bytes, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * Two-view geometric loss, ONE direction: ProjectionDepthSimilarityLoss.fwd,
 * model/networks.py:483-498 (unproject depth0 along `ray`, rigid transform by (R0,t0) then
 * (R1,t1), project with K, bilinear border-mode sample of depth1, mean clamped |d - sample|).
 *   depth0, depth1 [B][1][H][W]; ray [H*W][3] = [u v 1] @ Ki^T (networks.py:428-434);
 *   K [3][3]; R0, R1 [B][3][3]; t0, t1 [B][3]  (all device, f32);  clamp <= 0 disables clamping
 *   loss[0] = (accumulate ? loss[0] : 0) + mean            (the module sums both directions)
 * Backward: grad_depth0 = (accumulate0 ? grad_depth0 : 0) + d/d depth0; grad_depth1 is
 * ACCUMULATED with float atomics (caller zero-fills once per backward).
 * -------------------------------------------------------------------------------------- */
size_t ctd_geometric_workspace_bytes(int B, int H, int W);
int ctd_geometric_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K,
                          const float* R0, const float* t0, const float* R1, const float* t1,
                          float* loss, int accumulate, int B, int H, int W, float clamp,
                          void* workspace, size_t workspace_bytes, int device, void* stream);
/* Additive: BOTH directions (the module's tforward, model/networks.py:500-503) in one launch, the two means formed by the
 * last workgroup to finish: loss[0] = mean(depth0 -> view 1) + mean(depth1 -> view 0), the same bits as the two
 * ctd_geometric_fwd_f32 calls (accumulate 0, then 1 with the views swapped).
 *   workspace: ctd_geometric_workspace_bytes(2 * B, H, W) bytes (contents irrelevant);
 *   ticket: 65 4-byte device words (260 bytes) owned by the caller, ZERO before the first call on them; every call
 *           leaves them zero.  Calls that may run concurrently (different streams) need tickets (and a workspace) each. */
int ctd_geometric_sym_fwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K,
                              const float* R0, const float* t0, const float* R1, const float* t1,
                              float* loss, int B, int H, int W, float clamp, void* workspace,
                              size_t workspace_bytes, unsigned* ticket, int device, void* stream);
int ctd_geometric_bwd_f32(const float* depth0, const float* depth1, const float* ray, const float* K,
                          const float* R0, const float* t0, const float* R1, const float* t1,
                          const float* grad_loss, float* grad_depth0, int accumulate0,
                          float* grad_depth1, int B, int H, int W, float clamp, int device,
                          void* stream);

/* --------------------------------------------------------------------------------------
 * Fused pattern similarity loss (tolerance level, f32, block 9): RectifiedPatternSimilarityLoss.tforward,
 * model/networks.py:358-378 -- warp of the reference pattern by the predicted disparity
 * (grid_sample bilinear / border / align_corners=False on the grid of :362-369), block photometric
 * loss against the image (:376) and masked mean (:377) in one forward and one backward kernel.
 *   disp, im [B][1][H][W]; mask [B][1][H][W] or NULL (= ones); pattern [H][W] shared by the batch
 *   pattern_proj [B][1][H][W] (written); terms[3] (device) = { sum(mask*diff), sum(mask), their ratio }
 *   backward: grad_val (device scalar, d/d terms[2]), grad_proj [B][1][H][W] or NULL -> grad_disp
 * Workspace (forward only): ctd_pattern_loss_workspace_bytes().  The reduction is a fixed-order
 * tree: bitwise reproducible run to run.
 * -------------------------------------------------------------------------------------- */
size_t ctd_pattern_loss_workspace_bytes(int B, int H, int W);
int ctd_pattern_loss_fwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                             float* pattern_proj, float* terms, int B, int H, int W, int type,
                             float eps, void* workspace, size_t workspace_bytes, int device,
                             void* stream);
int ctd_pattern_loss_bwd_f32(const float* disp, const float* im, const float* mask, const float* pattern,
                             const float* terms, const float* grad_val, const float* grad_proj,
                             float* grad_disp, int B, int H, int W, int type, float eps, int device,
                             void* stream);

/* Several pyramid levels (and any number of track frames stacked in B) in ONE forward and ONE backward launch
 * (the reference loops over the scales in Python, model/exp_synph.py:107-111,
 * model/exp_synphge.py:141-150).  Results per level are identical to the single-level calls.
 *   levels[l]: tensors of level l as in ctd_pattern_loss_fwd/bwd_f32 (mask, grad_proj may be NULL;
 *   pattern_proj is written by the forward, grad_disp by the backward); at most 8 levels.
 *   terms [n_levels][3], grad_vals [n_levels] (device). */
typedef struct ctd_pattern_level {
  const float *disp, *im, *mask, *pattern;
  float* pattern_proj;
  const float* grad_proj;
  float* grad_disp;
  int B, H, W;
} ctd_pattern_level;
size_t ctd_pattern_loss_multi_workspace_bytes(int n_levels, const ctd_pattern_level* levels);
int ctd_pattern_loss_multi_fwd_f32(int n_levels, const ctd_pattern_level* levels, float* terms, int type,
                                   float eps, void* workspace, size_t workspace_bytes, int device,
                                   void* stream);
int ctd_pattern_loss_multi_bwd_f32(int n_levels, const ctd_pattern_level* levels, const float* terms,
                                   const float* grad_vals, int type, float eps, int device,
                                   void* stream);

/* --------------------------------------------------------------------------------------
 * Nearest-neighbour consistency ops (integer results, bit-exact).
 * Replace nn_cuda / crosscheck_cuda / proj_nn_cuda -- torchext/ext/ext_cuda.cpp:17-68,
 * functors torchext/ext/ext.h:13-117, Python torchext/functions.py:5-56.
 *   ctd_nn:         in0 [n0][3], in1 [n1][3] -> out int64 [n0] = argmin_j |in0[i] - in1[j]|^2
 *                   (first index wins ties; -1 if no squared distance is below 1e9)
 *   ctd_crosscheck: in0 int64 [n0], in1 int64 [n1] -> out uint8 [n0] = 1 where
 *                   in1[in0[i]] == i (in0[i] truncated to int as in ext.h:61; an index
 *                   >= n1, which the reference reads out of bounds, gives 0)
 *   ctd_proj_nn:    xyz0, xyz1 [B][H][W][3], K [3][3] (device) -> out int64 [B][H][W] =
 *                   flat index of the closest xyz1 point in the patch_size^2 patch around
 *                   the projection of xyz0 (ext.h:86-114), -1 if the patch is empty
 * -------------------------------------------------------------------------------------- */
int ctd_nn_f32(const float* in0, const float* in1, long n0, long n1, int64_t* out, int device,
               void* stream);
int ctd_nn_f64(const double* in0, const double* in1, long n0, long n1, int64_t* out, int device,
               void* stream);
int ctd_crosscheck(const int64_t* in0, const int64_t* in1, long n0, long n1, uint8_t* out,
                   int device, void* stream);
int ctd_proj_nn_f32(const float* xyz0, const float* xyz1, const float* K, int B, int H, int W,
                    int patch_size, int64_t* out, int device, void* stream);
int ctd_proj_nn_f64(const double* xyz0, const double* xyz1, const double* K, int B, int H, int W,
                    int patch_size, int64_t* out, int device, void* stream);

/* --------------------------------------------------------------------------------------
 * Synthetic structured-light rendering (data side of the path).
 * Replaces RendererGpu<float>::render_mesh_proj -- renderer/render/render_gpu.h,
 * functor RenderProjectorFunctor renderer/render/render.h:251-364 (Python: PyRenderer.mesh_proj,
 * renderer/cyrender.pyx:196-199): brute-force ray casting of a triangle mesh from the camera,
 * shadow ray from the projector, bilinear fetch of the projected pattern with distance decay
 * max(1, (d_alpha + d_beta * d)^2), Phong-shaded vertex colours as the ambient image.
 *   verts, colors [n_verts][3] f32, faces [n_faces][3] int32, pattern [proj_height][proj_width][3]: device
 *   cam, proj = { fx, fy, px, py, R[9] row-major, t[3] } (16 floats), shader = { ka, kd, ks, alpha }: HOST
 *   depth [H][W] (-1 where nothing is hit; may be NULL), color [H][W][3], normal [H][W][3] (may be NULL;
 *   left untouched where nothing is hit, as in the reference): device
 * depth / color are bit-identical to the reference CPU build; normal too when ks == 0.
 * -------------------------------------------------------------------------------------- */
int ctd_render_mesh_proj_f32(const float* verts, const float* colors, int n_verts, const int* faces,
                             int n_faces, const float* cam, int cam_width, int cam_height,
                             const float* proj, int proj_width, int proj_height, const float* shader,
                             const float* pattern, float d_alpha, float d_beta, float* depth,
                             float* color, float* normal, int device, void* stream);

/* Replaces RendererGpu<float>::render_mesh -- functor RenderMeshFunctor renderer/render/render.h:150-223
 * (Python: PyRenderer.mesh, renderer/cyrender.pyx:193-194): camera rays only.
 *   normals [n_verts][3] f32: device (the reference interpolates the given vertex normals, unnormalised)
 *   depth [H][W] (-1 where nothing is hit), color [H][W][3] = clamp(phong * interpolated vertex colour, 0, 1),
 *   normal [H][W][3] = interpolated vertex normal flipped towards the camera; colour and normal are 0 where
 *   nothing is hit.  Each of the three outputs may be NULL (skipped), as the reference's Buffer allows.
 * Bit-identical to the reference CPU build (colour: when ks == 0, else up to powf's last bits). */
int ctd_render_mesh_f32(const float* verts, const float* colors, const float* normals, int n_verts,
                        const int* faces, int n_faces, const float* cam, int cam_width, int cam_height,
                        const float* shader, float* depth, float* color, float* normal, int device,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTD_HIP_H */
