// Store-bandwidth ceiling of the t256 volume kernel's write pattern (no compute): every workgroup of 7 waves
// writes its 14 disparity planes x band rows x 256 columns, 1 KB per wave-instruction, exactly like the kernel's
// epilogue; compared with a linear fill of the same 1.84 GB.  hipcc --offload-arch=gfx950 -O3 store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(512) void pattern_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int dg,
                                                      int waves, int nd, int rows_per_iter) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const int f = blockIdx.z / n_dg, g = blockIdx.z % n_dg;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int col = blockIdx.x * 256 + lane * 4;
  typedef float f32x4 __attribute__((ext_vector_type(4))); const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int h = h_lo; h < h_hi; ++h)
    for (int j = 0; j < nd; ++j) {
      const int d = g * dg + wave * nd + j;
      if (d < D) __builtin_nontemporal_store(v, (f32x4*)(out + (((long)f * D + d) * H + h) * W + col));
    }
}

// same bytes, same WG count, but every WG writes ONE contiguous chunk (the layout-free ceiling)
__global__ __launch_bounds__(512) void linear_kernel(float4* out, long n4_per_wg) {
  float4* p = out + (long)blockIdx.x * n4_per_wg;
  typedef float f32x4 __attribute__((ext_vector_type(4))); const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = threadIdx.x; i < n4_per_wg; i += blockDim.x) __builtin_nontemporal_store(v, (f32x4*)(p + i));
}

__global__ __launch_bounds__(512) void pattern_plain_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int dg,
                                                            int waves, int nd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const int f = blockIdx.z / n_dg, g = blockIdx.z % n_dg;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int col = blockIdx.x * 256 + lane * 4;
  typedef float f32x4 __attribute__((ext_vector_type(4))); const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int h = h_lo; h < h_hi; ++h)
    for (int j = 0; j < nd; ++j) {
      const int d = g * dg + wave * nd + j;
      if (d < D) *(f32x4*)(out + (((long)f * D + d) * H + h) * W + col) = v;
    }
}


// cache-policy variants of the same pattern (gfx950 store modifiers)
template <int POLICY>
__global__ __launch_bounds__(512) void pattern_policy_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int dg,
                                                             int waves, int nd) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const int f = blockIdx.z / n_dg, g = blockIdx.z % n_dg;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int col = blockIdx.x * 256 + lane * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int h = h_lo; h < h_hi; ++h)
    for (int j = 0; j < nd; ++j) {
      const int d = g * dg + wave * nd + j;
      if (d >= D) continue;
      float* p = out + (((long)f * D + d) * H + h) * W + col;
      if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
      if (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    }
}

int main() {
  const int F = 16, D = 128, H = 432, W = 512;
  const long n = (long)F * D * H * W;
  float* out;
  CK(hipMalloc(&out, n * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 60; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-58s %.4f ms  %.0f GB/s\n", name, ms / 20, n * 4 / (ms / 20 * 1e-3) / 1e9);
  };
  for (int nt = 0; nt < 2; ++nt) {
    struct Cfg { int waves, nd, bands; } cfgs[] = {{7, 2, 10}, {8, 2, 10}, {7, 2, 5}, {7, 2, 20}, {8, 2, 8}, {8, 2, 16}};
    for (auto c : cfgs) {
      const int dg = c.waves * c.nd, n_dg = (D + dg - 1) / dg, band_rows = (H + c.bands - 1) / c.bands;
      dim3 grid(W / 256, (H + band_rows - 1) / band_rows, F * n_dg);
      char name[128];
      snprintf(name, sizeof name, "%s pattern waves %d nd %d bands %d (%d WGs)", nt ? "plain" : "nontemporal", c.waves, c.nd, c.bands,
               grid.x * grid.y * grid.z);
      if (nt == 0) time(name, [&] { hipLaunchKernelGGL(pattern_kernel, grid, dim3(512), 0, 0, out, H, W, D, band_rows, n_dg, dg, c.waves, c.nd, 1); });
      else time(name, [&] { hipLaunchKernelGGL(pattern_plain_kernel, grid, dim3(512), 0, 0, out, H, W, D, band_rows, n_dg, dg, c.waves, c.nd); });
    }
  }

  {
    const int waves = 7, nd = 2, bands = 10, dg = 14, n_dg = 10, band_rows = 44;
    dim3 grid(W / 256, bands, F * n_dg);
    const char* names[8] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt"};
#define RUN(P) { char name[128]; snprintf(name, sizeof name, "policy %-12s 7 waves 10 bands", names[P]); \
    time(name, [&] { hipLaunchKernelGGL(pattern_policy_kernel<P>, grid, dim3(512), 0, 0, out, H, W, D, band_rows, n_dg, dg, waves, nd); }); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
  }
  for (int wgs : {2560, 3200, 8192, 32768}) {
    char name[128];
    snprintf(name, sizeof name, "linear fill, %d WGs", wgs);
    const long per = n / 4 / wgs;
    time(name, [&] { hipLaunchKernelGGL(linear_kernel, dim3(wgs), dim3(512), 0, 0, (float4*)out, per); });
  }
  return 0;
}
