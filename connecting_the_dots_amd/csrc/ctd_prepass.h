// ctd_prepass.h -- what the kernels that produce the matcher's frame-side window statistics share (ncc_fast.hip:
// ncc_prepass_kernel; lcn_stream.hip: the fused LCN + statistics kernel).
#pragma once
#include "ctd_common.h"

namespace ctd {

constexpr double kDevFloor = 7e-2;     // windows with a smaller deviation are listed: keeps 1e-8 / (sa * sb) <= 2.1e-6 (ncc_inv_norm)
constexpr double kFlagRatio = 1.8284;  // list a window when F - 1 = n*(mean - centring)^2 / (sum sq. dev.) > sqrt(8) - 1

// The frame-side planes of the fast NCC path ([image][H][pitch], column c at o_off + c; `img` carries o_off replicate
// columns either side) and the list of windows the fix-up pass has to recompute.
struct StatPlanes {
  float *img, *mean, *dev;          // centred copy, mean_scale * (window mean - centring), reciprocal deviation (0: listed)
  int pitch, o_off;
  unsigned* n_flag;
  unsigned long long* flag_list;    // (image << 40) | (row << 20) | (column + 0x80000)
  float mean_scale, flag_ratio;
};

// lcn_stream.hip
bool lcn_stream_supported(int H, int W, int radius, int bs);
int lcn_stream_f32(const float* x, float* y, float* stds, int N, int H, int W, float eps, const StatPlanes& sp,
                   unsigned* clear_counters, int n_clear, bool exact, hipStream_t stream);

}  // namespace ctd
