// store_ceiling.hip -- what a store-only kernel reaches on this card, by footprint, bytes per wave-instruction, cache
// policy, grid shape and access pattern (not part of the product; its output is committed as
// profiles/round3_store_ceiling.txt).
//
//   hipcc --offload-arch=gfx950 -O3 store_ceiling.hip -o store_ceiling && ./store_ceiling
//
// Every row: pattern, policy, grid x block, footprint, bytes written per launch, microseconds per launch (average of
// 20 launches after 40 untimed ones, HIP events on the null stream), TB/s.
//
// Patterns
//   linear256   one dword per lane, 256 B per wave-instruction, grid-stride over the buffer   (the shape of the guide's
//               "plain stores 6.0-6.2 TB/s" row, MI355X_MICROARCH.md, Global float atomics table)
//   linear1k    16 B per lane, 1 KB per wave-instruction, grid-stride
//   chunk1k     16 B per lane, every workgroup writes ONE contiguous chunk (no interleaving between workgroups)
//   rows2304    one dword per lane, 256 B per wave-instruction, random 2,304-B rows of the table (the guide's row)
//   copy16      float4 copy, bytes = read + written (the guide's 6.29 TB/s "float4 copy" row)
//   volume      the NCC volume kernel's own epilogue pattern: a workgroup owns `waves` x 2 disparity planes x a band of
//               rows x 256 columns and writes one 1 KB row segment per plane and row (ncc_fast.hip, t256 / all-D kernels)
#include <hip/hip_runtime.h>
#include <chrono>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                      \
    }                                                                \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ inline void store16(float* p, f32x4 v) {   // 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt
  if (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
}
template <int POLICY>
__device__ inline void store4(float* p, float v) {
  if (POLICY == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (POLICY == 1) asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  if (POLICY >= 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
static const char* kPolicy[5] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt"};

template <int POLICY>
__global__ __launch_bounds__(256) void linear256_kernel(float* out, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) store4<POLICY>(out + i, (float)i);
}

template <int POLICY>
__global__ __launch_bounds__(256) void linear1k_kernel(float* out, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) store16<POLICY>(out + 4 * i, v);
}

template <int POLICY>
__global__ __launch_bounds__(256) void chunk1k_kernel(float* out, long n4_per_wg) {
  float* p = out + (long)blockIdx.x * n4_per_wg * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (long i = threadIdx.x; i < n4_per_wg; i += blockDim.x) store16<POLICY>(p + 4 * i, v);
}

// random 2,304-B rows (576 floats = 9 wave-instructions of 256 B), one wave per row at a time, rows dealt by a hash
template <int POLICY>
__global__ __launch_bounds__(512) void rows2304_kernel(float* out, long n_rows, long rows_per_wave) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (long k = 0; k < rows_per_wave; ++k) {
    const unsigned long long j = (unsigned long long)(wave * rows_per_wave + k);
    const long row = (long)((j * 0x9E3779B97F4A7C15ull >> 20) % (unsigned long long)n_rows);   // spread, not a bijection
    float* p = out + row * 576;
#pragma unroll
    for (int s = 0; s < 9; ++s) store4<POLICY>(p + 64 * s + lane, (float)s);
  }
}

__global__ __launch_bounds__(256) void copy16_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// 512 streams (256 workgroups x 2 wavefronts), groups of k streams interleave their 1-KB segments into one contiguous
// region (store_streams.hip's model with T = 2, S = 1): stream sg, segment r -> KB (sg / k) * k * R + r * k + sg % k
template <int POLICY>
__global__ __launch_bounds__(128) void interleaved_kernel(float* out, long R, long k) {
  const long sg = (long)blockIdx.x * 2 + (threadIdx.x >> 6);
  float* base = out + ((sg / k) * k * R + (sg % k)) * 256 + (threadIdx.x & 63) * 4;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long r = 0; r < R; ++r) store16<POLICY>(base + r * k * 256, v);
}

// the volume kernel's pattern: grid (column tiles of 256, bands, frames * disparity groups); all_d: one workgroup walks
// every disparity group of its (tile, band, frame) in turn (the round-3 kernel), grid.z = frames
template <int POLICY>
__global__ __launch_bounds__(1024) void volume_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int waves,
                                                      int all_d) {
  extern __shared__ float lds_unused[];                  // (declared so that the launch's dynamic LDS size limits residency)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const int f = all_d ? blockIdx.z : blockIdx.z / n_dg;
  const int g_lo = all_d ? 0 : blockIdx.z % n_dg, g_hi = all_d ? n_dg : g_lo + 1;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int col = blockIdx.x * 256 + lane * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int g = g_lo; g < g_hi; ++g)
    for (int h = h_lo; h < h_hi; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = g * waves * 2 + wave * 2 + j;
        if (d < D) store16<POLICY>(out + (((long)f * D + d) * H + h) * W + col, v);
      }
}

// all-D pattern with `burst` consecutive rows of a plane written back to back, optionally 512 columns per workgroup
template <int POLICY>
__global__ __launch_bounds__(1024) void burst_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int waves,
                                                     int burst, int wide) {
  extern __shared__ float lds_unused[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= waves) return;
  const int f = blockIdx.z;
  const int h_lo = blockIdx.y * band_rows, h_hi = min(h_lo + band_rows, H);
  const int col = blockIdx.x * 256 + lane * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int g = 0; g < n_dg; ++g)
    for (int h0 = h_lo; h0 < h_hi; h0 += burst)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = g * waves * 2 + wave * 2 + j;
        if (d >= D) continue;
        for (int h = h0; h < min(h0 + burst, h_hi); ++h) {
          float* p = out + (((long)f * D + d) * H + h) * W + col;
          store16<POLICY>(p, v);
          if (wide) store16<POLICY>(p + 256, v);
        }
      }
}

static hipEvent_t ev_a, ev_b;
template <class L>
static void timeit(const char* pattern, const char* policy, unsigned grid, unsigned block, double footprint, double bytes,
                   L launch) {
  for (int i = 0; i < 40; ++i) launch();
  hipEventRecord(ev_a);
  for (int i = 0; i < 20; ++i) launch();
  hipEventRecord(ev_b);
  hipEventSynchronize(ev_b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, ev_a, ev_b);
  const double us = ms * 1e3 / 20;
  printf("%-34s %-8s grid %6u x %4u  footprint %7.1f MB  bytes %7.1f MB  %8.1f us  %5.2f TB/s\n", pattern, policy, grid,
         block, footprint / 1e6, bytes / 1e6, us, bytes / us / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int F = 16, D = 128, H = 432, W = 512;
  if (argc > 1 && argv[1][0] == 'p') {
    // `store_ceiling pattern [band_rows [seconds]]`: only the all-D kernel's own pattern (13 storing wavefronts, one
    // workgroup per CU, non-temporal; band_rows = the kernel's plan, default config 2's 54 = 8 bands), `key = value`
    // lines -- what bench.py runs beside its timed region for the second denominator.  Two rates: the first 20 launches
    // after a short warm-up (clocks up, card not yet power-limited) and the average of 20 launches after `seconds`
    // (default 0.6) of continuous launches -- the card sheds ~8 % of its rate within ~0.2 s of sustained load, and the
    // bench's timed region runs in that state.
    const long n = (long)F * D * H * W;
    const int band_rows = argc > 2 ? atoi(argv[2]) : 54;
    const double seconds = argc > 3 ? atof(argv[3]) : 0.6;
    float* o;
    CK(hipMalloc(&o, n * 4));
    CK(hipEventCreate(&ev_a));
    CK(hipEventCreate(&ev_b));
    CK(hipFuncSetAttribute((const void*)volume_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int waves = 13, n_dg = 5, bands = (H + band_rows - 1) / band_rows;
    dim3 grid(W / 256, bands, F);
    auto region = [&]() {
      hipEventRecord(ev_a);
      for (int i = 0; i < 20; ++i)
        hipLaunchKernelGGL(volume_kernel<1>, grid, dim3(1024), (size_t)124 * 1024, 0, o, H, W, D, band_rows, n_dg, waves, 1);
      hipEventRecord(ev_b);
      hipEventSynchronize(ev_b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev_a, ev_b);
      return ms * 50.f;                                            // us per launch
    };
    for (int i = 0; i < 100; ++i)
      hipLaunchKernelGGL(volume_kernel<1>, grid, dim3(1024), (size_t)124 * 1024, 0, o, H, W, D, band_rows, n_dg, waves, 1);
    const float burst_us = region();
    float best_us = burst_us;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) region();
    best_us = 0.5f * (region() + region());
    printf("all_d_pattern_bands = %d\nall_d_pattern_store_only_burst_TBs = %.3f\n", bands, n * 4.0 / burst_us / 1e6);
    printf("all_d_pattern_store_only_us = %.1f\nall_d_pattern_store_only_TBs = %.3f\n", best_us, n * 4.0 / best_us / 1e6);
    // The CARD's best store-only rate at the same 1.81 GB, sustained state (round 4, tools/ubench_src/store_streams.hip:
    // profiles/round4_store_streams.txt): few wavefronts per CU writing interleaved contiguous pieces in step -- (i) the
    // 256-workgroup x 256-thread grid-stride fill, (ii) 2 wavefronts per CU, groups of 64 streams interleaving 1-KB pieces.
    // No kernel that owns 26 disparity planes per workgroup can write that way; reported as the second denominator.
    auto timed = [&](auto launch) {
      for (int i = 0; i < 40; ++i) launch();
      hipEventRecord(ev_a);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(ev_b);
      hipEventSynchronize(ev_b);
      float ms = 0.f;
      hipEventElapsedTime(&ms, ev_a, ev_b);
      return ms * 50.f;
    };
    const float fill_us = timed([&] { hipLaunchKernelGGL(linear1k_kernel<0>, dim3(256), dim3(256), 0, 0, o, n / 4); });
    const float fill_nt_us = timed([&] { hipLaunchKernelGGL(linear1k_kernel<1>, dim3(256), dim3(256), 0, 0, o, n / 4); });
    const float il_us = timed([&] { hipLaunchKernelGGL(interleaved_kernel<0>, dim3(256), dim3(128), 0, 0, o, n / 256 / 512, 64); });
    const float il_nt_us = timed([&] { hipLaunchKernelGGL(interleaved_kernel<1>, dim3(256), dim3(128), 0, 0, o, n / 256 / 512, 64); });
    float card_us = fill_us;
    const char* which = "256 x 256-thread grid-stride fill, plain";
    if (fill_nt_us < card_us) { card_us = fill_nt_us; which = "256 x 256-thread grid-stride fill, nt"; }
    if (il_us < card_us) { card_us = il_us; which = "2 wavefronts per CU, 64 streams interleaved, plain"; }
    if (il_nt_us < card_us) { card_us = il_nt_us; which = "2 wavefronts per CU, 64 streams interleaved, nt"; }
    printf("card_best_store_only_us = %.1f\ncard_best_store_only_TBs = %.3f\ncard_best_pattern = %s\n", card_us,
           n * 4.0 / card_us / 1e6, which);
    printf("card_fill_256wg_TBs = %.3f\ncard_fill_256wg_nt_TBs = %.3f\ncard_interleaved_2waves_TBs = %.3f\ncard_interleaved_2waves_nt_TBs = %.3f\n",
           n * 4.0 / fill_us / 1e6, n * 4.0 / fill_nt_us / 1e6, n * 4.0 / il_us / 1e6, n * 4.0 / il_nt_us / 1e6);
    return 0;
  }
  const long n_vol = (long)F * D * H * W;            // 1.81 GB: config 2's volume
  float *out, *in;
  CK(hipMalloc(&out, n_vol * 4));
  CK(hipMalloc(&in, n_vol * 2));                     // copy source: half the volume
  CK(hipMemset(in, 0, n_vol * 2));
  CK(hipEventCreate(&ev_a));
  CK(hipEventCreate(&ev_b));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs, clock %d MHz, memory clock %d MHz, bus %d bits\n", prop.name, prop.multiProcessorCount,
         prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.memoryBusWidth);

#define POL(P, CALL) { constexpr int PP = P; const char* pname = kPolicy[P]; CALL; }
  // 1. footprint sweep, linear: where the Infinity Cache (256 MiB) stops absorbing the writes
  for (long mb : {75L, 151L, 302L, 604L, 1812L}) {
    long n = mb * 1000000 / 4 / 1024 * 1024;
    if (n > n_vol) n = n_vol / 1024 * 1024;          // the last entry is the whole volume buffer, never past it
    for (int grid : {2048, 8192})
      POL(0, timeit("linear256 (256 B per wave-instr)", pname, grid, 256, n * 4.0, n * 4.0,
                    [&] { hipLaunchKernelGGL(linear256_kernel<PP>, dim3(grid), dim3(256), 0, 0, out, n); }));
    POL(0, timeit("linear1k  (1 KB per wave-instr)", pname, 2048, 256, n * 4.0, n * 4.0,
                  [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(2048), dim3(256), 0, 0, out, n / 4); }));
    POL(1, timeit("linear1k  (1 KB per wave-instr)", pname, 2048, 256, n * 4.0, n * 4.0,
                  [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(2048), dim3(256), 0, 0, out, n / 4); }));
  }
  // 2. the guide's row shape: random 2,304-B rows of a 75 MB / 302 MB table and of the 1.81 GB volume
  for (long mb : {75L, 302L, 1812L}) {
    long n_rows = mb * 1000000 / 2304;
    if (n_rows * 576 > n_vol) n_rows = n_vol / 576;  // rows stay inside the buffer
    const long waves = 256L * 8;                     // 8 waves per CU
    const long rows_per_wave = 4 * n_rows / waves;   // four table-sizes of rows per launch
    POL(0, timeit("rows2304 (random rows, 256 B instr)", pname, 256, 512, n_rows * 2304.0, waves * rows_per_wave * 2304.0,
                  [&] { hipLaunchKernelGGL(rows2304_kernel<PP>, dim3(256), dim3(512), 0, 0, out, n_rows, rows_per_wave); }));
  }
  // 3. the full 1.81 GB, linear, by policy and grid
  for (int grid : {256, 512, 2048, 8192, 32768}) {
    POL(0, timeit("linear1k  1.81 GB", pname, grid, 256, n_vol * 4.0, n_vol * 4.0,
                  [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(grid), dim3(256), 0, 0, out, n_vol / 4); }));
    POL(1, timeit("linear1k  1.81 GB", pname, grid, 256, n_vol * 4.0, n_vol * 4.0,
                  [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(grid), dim3(256), 0, 0, out, n_vol / 4); }));
  }
  for (int grid : {512, 3200, 8192}) {
    POL(0, timeit("chunk1k   1.81 GB (one chunk per WG)", pname, grid, 256, n_vol * 4.0, n_vol * 4.0,
                  [&] { hipLaunchKernelGGL(chunk1k_kernel<PP>, dim3(grid), dim3(256), 0, 0, out, n_vol / 4 / grid); }));
    POL(1, timeit("chunk1k   1.81 GB (one chunk per WG)", pname, grid, 256, n_vol * 4.0, n_vol * 4.0,
                  [&] { hipLaunchKernelGGL(chunk1k_kernel<PP>, dim3(grid), dim3(256), 0, 0, out, n_vol / 4 / grid); }));
  }
  POL(2, timeit("linear1k  1.81 GB", pname, 2048, 256, n_vol * 4.0, n_vol * 4.0,
                [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(2048), dim3(256), 0, 0, out, n_vol / 4); }));
  POL(3, timeit("linear1k  1.81 GB", pname, 2048, 256, n_vol * 4.0, n_vol * 4.0,
                [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(2048), dim3(256), 0, 0, out, n_vol / 4); }));
  POL(4, timeit("linear1k  1.81 GB", pname, 2048, 256, n_vol * 4.0, n_vol * 4.0,
                [&] { hipLaunchKernelGGL(linear1k_kernel<PP>, dim3(2048), dim3(256), 0, 0, out, n_vol / 4); }));
  // 4. float4 copy (bytes = read + written), 0.9 GB each way
  timeit("copy16 (bytes = read + written)", "nt", 8192, 256, n_vol * 4.0, n_vol * 4.0,
         [&] { hipLaunchKernelGGL(copy16_kernel, dim3(8192), dim3(256), 0, 0, (const f32x4*)in, (f32x4*)out, n_vol / 8); });
  // 5. the volume kernel's own pattern
  // lds_kb > 0: dynamic LDS the workgroup declares (nothing is stored there): it limits the workgroups RESIDENT per CU the
  // way the real kernels' staging does -- the per-group kernel holds 63 KB (two workgroups per CU), the all-D kernel
  // 124 KB (one per CU, so its 512 workgroups run as two rounds of 256)
  struct Cfg { int waves, bands, all_d, lds_kb; } cfgs[] = {{7, 10, 0, 0}, {7, 10, 0, 63}, {8, 10, 0, 0}, {15, 16, 1, 0},
                                                            {13, 16, 1, 124}, {15, 16, 1, 124}, {15, 8, 1, 0}, {15, 8, 1, 124},
                                                            {15, 24, 1, 0}, {14, 16, 1, 0}, {16, 16, 1, 0}, {7, 16, 1, 0}};
  CK(hipFuncSetAttribute((const void*)volume_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)volume_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)volume_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)burst_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)burst_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (auto c : cfgs) {
    const int dg = c.waves * 2, n_dg = (D + dg - 1) / dg, band_rows = (H + c.bands - 1) / c.bands;
    dim3 grid(W / 256, (H + band_rows - 1) / band_rows, c.all_d ? F : F * n_dg);
    char name[96];
    snprintf(name, sizeof name, "volume %2d waves %2d bands %s LDS %3d KB", c.waves, c.bands, c.all_d ? "all-D" : "group", c.lds_kb);
    const unsigned block = c.waves <= 8 ? 512 : 1024;
    const size_t lds = (size_t)c.lds_kb * 1024;
    POL(0, timeit(name, pname, grid.x * grid.y * grid.z, block, n_vol * 4.0, n_vol * 4.0, [&] {
          hipLaunchKernelGGL(volume_kernel<PP>, grid, dim3(block), lds, 0, out, H, W, D, band_rows, n_dg, c.waves, c.all_d); }));
    POL(1, timeit(name, pname, grid.x * grid.y * grid.z, block, n_vol * 4.0, n_vol * 4.0, [&] {
          hipLaunchKernelGGL(volume_kernel<PP>, grid, dim3(block), lds, 0, out, H, W, D, band_rows, n_dg, c.waves, c.all_d); }));
    POL(4, timeit(name, pname, grid.x * grid.y * grid.z, block, n_vol * 4.0, n_vol * 4.0, [&] {
          hipLaunchKernelGGL(volume_kernel<PP>, grid, dim3(block), lds, 0, out, H, W, D, band_rows, n_dg, c.waves, c.all_d); }));
  }
  // 6. would a more sequential version of the all-D pattern pay?  burst = consecutive rows of one plane written back to
  //    back by a wave (the kernel writes one row per plane and ~1 us); wide = the workgroup covers both 256-column tiles
  //    (2 KB contiguous per plane row, written as two stores back to back)
  for (int wide = 0; wide < 2; ++wide)
    for (int burst : {1, 2, 4, 9}) {
      const int waves = 13, bands = 16, dg = 26, n_dg = 5, band_rows = 27;
      dim3 grid(wide ? 1 : W / 256, bands, F);
      char name[96];
      snprintf(name, sizeof name, "all-D 13 waves burst %d rows%s LDS 124", burst, wide ? ", 512 wide" : "");
      (void)dg;
      POL(1, timeit(name, pname, grid.x * grid.y * grid.z, 1024, n_vol * 4.0, n_vol * 4.0, [&] {
            hipLaunchKernelGGL(burst_kernel<PP>, grid, dim3(1024), (size_t)124 * 1024, 0, out, H, W, D, band_rows, n_dg, waves, burst, wide); }));
      POL(0, timeit(name, pname, grid.x * grid.y * grid.z, 1024, n_vol * 4.0, n_vol * 4.0, [&] {
            hipLaunchKernelGGL(burst_kernel<PP>, grid, dim3(1024), (size_t)124 * 1024, 0, out, H, W, D, band_rows, n_dg, waves, burst, wide); }));
    }
  return 0;
}
