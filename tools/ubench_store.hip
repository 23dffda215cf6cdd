// Pure-store microbenchmark replaying the volume store pattern of ncc_fast_wide_kernel (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
// MODE 0: wide-kernel pattern: block = 4 waves x 2 disparities, wave writes 992 B per (d, h) row, tiles of 248 columns
// MODE 1: same but a block covers both column tiles (8 waves: 2 tiles x 4 d-pairs) -> 2 KB contiguous per (d,h)
// MODE 2: fully sequential: every wave streams a contiguous slab (1 KB per instruction)
__device__ inline float spin4(float x, int spin) {
  float a = x, b = x + 1, c = x + 2, d = x + 3;
  for (int s = 0; s < spin; ++s) { a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f; }
  return a + b + c + d;
}
template <int MODE, bool NT = false>
__global__ void k(float* out, int H, int W, int D, int band_rows, int n_dg, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long HW = (long)H * W;
  if (MODE == 2) {
    long total4 = (long)(gridDim.z / n_dg) * D * HW / 4;   // gridDim.z = frames * n_dg
    long nwaves = (long)gridDim.x * gridDim.y * gridDim.z * (blockDim.x / 64);
    long wid = ((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x * (blockDim.x / 64) + blockIdx.x * (blockDim.x / 64) + wave;
    long per = (total4 / 64 + nwaves - 1) / nwaves;
    float4* o = (float4*)out;
    for (long i = 0; i < per; ++i) {
      long idx = (wid * per + i) * 64 + lane;
      float x = (float)i; for (int s = 0; s < spin; ++s) x = x * 1.0001f + 0.5f;
      if (idx < total4) o[idx] = make_float4(x, 1.f, 2.f, 3.f);
    }
    return;
  }
  const int f = blockIdx.z / n_dg, dg = blockIdx.z % n_dg;
  const int tile = MODE == 1 ? (wave >> 2) : blockIdx.x;
  const int w4 = MODE == 1 ? (wave & 3) : wave;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int c0 = (MODE == 3 || MODE == 4) ? tile * 256 + 4 * lane : tile * 248 - 4 + 4 * lane;
  const bool lane_out = (MODE == 3 || MODE == 4) ? true : (lane >= 1 && lane <= 62 && c0 < W);
  float* vol = out + (long)f * D * HW;
  for (int h = h_lo; h < h_hi; ++h) {
    float x = spin4((float)h, spin);
    for (int j = 0; j < 2; ++j) {
      int d = MODE == 4 ? (dg * 8 + w4 + 4 * j) : (dg * 8 + w4 * 2 + j);
      if (lane_out) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {x, 1.f, 2.f, 3.f};
        f4* o = (f4*)(vol + (long)d * HW + (long)h * W + c0);
        if (NT) __builtin_nontemporal_store(v, o); else *o = v;
      }
    }
  }
}
template <int MODE, bool NT = false> void run(const char* name, float* d, int N, int H, int W, int D, int spin) {
  int n_dg = D / 8, bands = 4, band_rows = (H + bands - 1) / bands;
  dim3 grid(MODE == 1 ? 1 : 2, bands, N * n_dg), block(MODE == 1 ? 512 : 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NT><<<grid, block>>>(d, H, W, D, band_rows, n_dg, spin); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) k<MODE, NT><<<grid, block>>>(d, H, W, D, band_rows, n_dg, spin); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double bytes = (double)N * D * H * ((MODE == 2 || MODE == 3 || MODE == 4) ? W : 496) * 4;
  printf("%-28s spin=%3d %8.3f ms  %.2f TB/s  (%.1f us/frame)\n", name, spin, ms, bytes / ms / 1e9, ms * 1e3 / N);
}
int main() {
  int N = 16, H = 432, W = 512, D = 128;
  float* d; hipMalloc(&d, (size_t)N * D * H * W * 4);
  for (int spin : {0, 50, 100, 200}) {
    run<0>("wide pattern (992 B pieces)", d, N, H, W, D, spin);
    run<0, true>("wide pattern, nontemporal", d, N, H, W, D, spin);
    run<3>("aligned 1 KB pieces", d, N, H, W, D, spin);
    run<3, true>("aligned 1 KB, nontemporal", d, N, H, W, D, spin);
  }
  return 0;
}
