// nn_ops.hip -- nearest-neighbour consistency ops of the torchext surface.
//
//   nn          NNFunctor<T,3>      /root/reference/torchext/ext/ext.h:13-47
//   crosscheck  CrossCheckFunctor   ext.h:49-66
//   proj_nn     ProjNNFunctor<T,3>  ext.h:68-117
//
// Integer results, bit-exact by construction: every distance is evaluated with the reference's operation
// order (no FMA: the library is built with -ffp-contract=off) and candidates are visited in ascending index
// order with the reference's strict `<`, so ties resolve to the same index.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ctd_common.h"
#include "ctd_internal.h"

namespace ctd {

// One query point per thread; the candidate set streams through LDS in tiles that every lane reads at the
// same address (a broadcast, conflict-free), padded to 4 components so that a candidate is one ds_read_b128
// (f32) / two (f64).  O(n0 * n1) VALU work, ~11 instructions per pair; HBM traffic is negligible.
constexpr int kNNTile = 1024;

template <typename T>
__global__ __launch_bounds__(256) void nn_kernel(const T* __restrict__ in0, const T* __restrict__ in1, long n0, long n1,
                                                 int64_t* __restrict__ out) {
  __shared__ T tile[kNNTile][4];
  const long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = idx0 < n0;
  T q0 = 0, q1 = 0, q2 = 0;
  if (live) { q0 = in0[idx0 * 3 + 0]; q1 = in0[idx0 * 3 + 1]; q2 = in0[idx0 * 3 + 2]; }
  T min_dist = (T)1e9;                                              // ext.h:29
  long min_arg = -1;
  for (long base = 0; base < n1; base += kNNTile) {
    const int n = (int)min((long)kNNTile, n1 - base);
    __syncthreads();
    for (int i = threadIdx.x; i < n * 3; i += blockDim.x) tile[i / 3][i % 3] = in1[base * 3 + i];
    __syncthreads();
    if (!live) continue;
#pragma unroll 4
    for (int i = 0; i < n; ++i) {
      const T d0 = q0 - tile[i][0], d1 = q1 - tile[i][1], d2 = q2 - tile[i][2];
      T dist = d0 * d0;                                             // 0 + d0^2 == d0^2 exactly
      dist += d1 * d1;
      dist += d2 * d2;                                              // ext.h:33-37
      if (dist < min_dist) {                                        // ext.h:39: strict, first index wins
        min_dist = dist;
        min_arg = base + i;
      }
    }
  }
  if (live) out[idx0] = min_arg;
}

__global__ __launch_bounds__(256) void crosscheck_kernel(const int64_t* __restrict__ in0,
                                                         const int64_t* __restrict__ in1, long n0, long n1,
                                                         uint8_t* __restrict__ out) {
  const long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx0 >= n0) return;
  const int idx1 = (int)in0[idx0];                                  // ext.h:61 truncates to int
  // the reference reads in1[idx1] unchecked; an index past the end counts as "not mutual" here
  bool ok = idx1 >= 0 && idx1 < n1;
  if (ok) {
    const int64_t back = in1[idx1];
    ok = back >= 0 && back == idx0;
  }
  out[idx0] = ok ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void proj_nn_kernel(const T* __restrict__ xyz0, const T* __restrict__ xyz1,
                                                      const T* __restrict__ K, long batch_size, long height,
                                                      long width, int patch_size, int64_t* __restrict__ out) {
  const long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx0 >= batch_size * height * width) return;
  const long bs = idx0 / (height * width);
  const T x = xyz0[idx0 * 3 + 0], y = xyz0[idx0 * 3 + 1], z = xyz0[idx0 * 3 + 2];
  const T d = K[6] * x + K[7] * y + K[8] * z;                       // ext.h:89-91
  const T u = (K[0] * x + K[1] * y + K[2] * z) / d;
  const T v = (K[3] * x + K[4] * y + K[5] * z) / d;
  const double ud = (double)u + 0.5, vd = (double)v + 0.5;          // `int u0 = u + 0.5` promotes to double
  long min_idx1 = -1;
  // projections that do not fit an int (d == 0, NaN) are undefined in the reference and leave the patch
  // outside the image on x86; here they yield "no candidate" explicitly
  if (ud > -2147483649.0 && ud < 2147483648.0 && vd > -2147483649.0 && vd < 2147483648.0) {
    const int u0 = (int)ud, v0 = (int)vd;                           // truncation toward zero, ext.h:94-95
    T min_dist = (T)1e9;
    for (int pv = 0; pv < patch_size; ++pv) {                       // pidx = pv * patch_size + pu ascending
      const long v1 = (long)v0 + pv - patch_size / 2;
      if (v1 < 0 || v1 >= height) continue;
      for (int pu = 0; pu < patch_size; ++pu) {
        const long u1 = (long)u0 + pu - patch_size / 2;
        if (u1 < 0 || u1 >= width) continue;
        const long idx1 = (bs * height + v1) * width + u1;
        const T a = xyz1[idx1 * 3 + 0], b = xyz1[idx1 * 3 + 1], c = xyz1[idx1 * 3 + 2];
        const T dd = (x - a) * (x - a) + (y - b) * (y - b) + (z - c) * (z - c);   // ext.h:108
        if (dd < min_dist) {
          min_dist = dd;
          min_idx1 = idx1;
        }
      }
    }
  }
  out[idx0] = min_idx1;
}

template <typename T>
static int nn_launch(const T* in0, const T* in1, long n0, long n1, int64_t* out, hipStream_t stream) {
  if (n0 == 0) return CTD_OK;
  hipLaunchKernelGGL(nn_kernel<T>, dim3((unsigned)ceil_div(n0, 256L)), dim3(256), 0, stream, in0, in1, n0, n1, out);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}
int nn_f32(const float* in0, const float* in1, long n0, long n1, int64_t* out, hipStream_t s) {
  return nn_launch(in0, in1, n0, n1, out, s);
}
int nn_f64(const double* in0, const double* in1, long n0, long n1, int64_t* out, hipStream_t s) {
  return nn_launch(in0, in1, n0, n1, out, s);
}

int crosscheck_i64(const int64_t* in0, const int64_t* in1, long n0, long n1, uint8_t* out, hipStream_t stream) {
  if (n0 == 0) return CTD_OK;
  hipLaunchKernelGGL(crosscheck_kernel, dim3((unsigned)ceil_div(n0, 256L)), dim3(256), 0, stream, in0, in1, n0, n1, out);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}

template <typename T>
static int proj_nn_launch(const T* xyz0, const T* xyz1, const T* K, long B, long H, long W, int patch_size,
                          int64_t* out, hipStream_t stream) {
  const long N = B * H * W;
  if (N == 0) return CTD_OK;
  hipLaunchKernelGGL(proj_nn_kernel<T>, dim3((unsigned)ceil_div(N, 256L)), dim3(256), 0, stream, xyz0, xyz1, K, B, H, W,
                     patch_size, out);
  CTD_LAUNCH_CHECK();
  return CTD_OK;
}
int proj_nn_f32(const float* xyz0, const float* xyz1, const float* K, long B, long H, long W, int patch_size,
                int64_t* out, hipStream_t s) {
  return proj_nn_launch(xyz0, xyz1, K, B, H, W, patch_size, out, s);
}
int proj_nn_f64(const double* xyz0, const double* xyz1, const double* K, long B, long H, long W, int patch_size,
                int64_t* out, hipStream_t s) {
  return proj_nn_launch(xyz0, xyz1, K, B, H, W, patch_size, out, s);
}

}  // namespace ctd
