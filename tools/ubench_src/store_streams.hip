// store_streams.hip -- what decides the rate of a store-only kernel at the volume's 1.81 GB: how many 1-KB streams
// are written at the same time, and how far apart they lie (round-4 follow-up of store_ceiling.hip: a 256-workgroup
// grid-stride fill reached 6.5-6.8 TB/s where the all-D kernel's own pattern reached 5.3-5.6 and the same fill with
// 512-8,192 workgroups 4.1-5.6).  Not part of the product.
//
//   hipcc --offload-arch=gfx950 -O3 store_streams.hip -o store_streams && ./store_streams [quick]
//
// Model: G workgroups (one per CU: each declares 100 KB of LDS) x T wavefronts x S streams per wavefront.  A STREAM is a
// sequence of R segments of 1 KB (one dwordx4 wave-instruction each).  Per step a wavefront writes one segment of each of
// its S streams, then `pace` x 64 idle cycles (s_sleep) -- pace 0 = as fast as the memory system takes them.
//   stream number  sg = (wg * T + wave) * S + s                   ("within": neighbours in sg belong to one workgroup)
//                  sg = (s * T + wave) * G + wg                   ("across": neighbours in sg belong to different CUs)
//   segment r of stream sg lives at    (sg / k) * k * R  +  r * k  +  (sg % k)      [KB]
// i.e. groups of k streams interleave their segments into one contiguous region of k * R KB: k = 1 every stream owns a
// contiguous chunk of R KB; k = all streams is the linear grid-stride fill.  The all-D kernel's own pattern is one more
// row of the table ("alld": (frame, d-plane, band, tile) addressing, 13 waves x 2 planes, 54-row bands), with variants of
// what a workgroup's wavefronts write at the same time:
//   skew   every wavefront starts at another row of the band (wave w at row (w * skew) % rows)
//   both   one workgroup writes BOTH 256-column tiles of its rows (2 KB contiguous per plane row, 128 workgroups x 2)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                        \
  do {                                                               \
    hipError_t e_ = (x);                                             \
    if (e_ != hipSuccess) {                                          \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                      \
    }                                                                \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ inline void store16(float* p, f32x4 v) {
  if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

struct Cfg {
  int G, T, S;        // workgroups, wavefronts per workgroup, streams per wavefront
  long R;             // segments (KB) per stream
  long k;             // interleave factor (streams per contiguous region)
  int across;         // stream numbering
  int pace;           // s_sleep units between steps
};

template <int NT>
__global__ __launch_bounds__(1024) void streams_kernel(float* out, Cfg c) {
  extern __shared__ float lds_unused[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= c.T) return;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  float* base[4];
  for (int s = 0; s < c.S; ++s) {
    const long sg = c.across ? ((long)s * c.T + wave) * c.G + blockIdx.x : ((long)blockIdx.x * c.T + wave) * c.S + s;
    base[s] = out + ((sg / c.k) * c.k * c.R + (sg % c.k)) * 256 + lane * 4;
  }
  const long adv = c.k * 256;
  for (long r = 0; r < c.R; ++r) {
    for (int s = 0; s < c.S; ++s) store16<NT>(base[s] + r * adv, v);
    for (int q = 0; q < c.pace; ++q) __builtin_amdgcn_s_sleep(1);
  }
}

// scattered streams with BURSTS: a wavefront visits its S streams in turn and writes `seg` contiguous KB per visit (seg
// store instructions back to back) -- what store-only wavefronts fed from an LDS staging buffer would do (round 5 idea:
// two such wavefronts per CU, 26 planes, both column tiles and 2-4 rows per visit)
template <int NT>
__global__ __launch_bounds__(1024) void burst_streams_kernel(float* out, int T, int S, long R, int seg) {
  extern __shared__ float lds_unused[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= T) return;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  const long sg0 = ((long)blockIdx.x * T + wave) * S;
  for (long r = 0; r < R; r += seg)
    for (int s = 0; s < S; ++s) {
      float* p = out + ((sg0 + s) * R + r) * 256 + lane * 4;
      for (int q = 0; q < seg; ++q) store16<NT>(p + q * 256, v);
    }
}

// the all-D kernel's pattern: item = (tile, band, frame), tile fastest; pass g of n_dg; wave w writes planes g*2T + 2w + j
template <int NT>
__global__ __launch_bounds__(1024) void alld_kernel(float* out, int H, int W, int D, int band_rows, int n_dg, int T, int skew,
                                                    int both, int pace, int order) {
  extern __shared__ float lds_unused[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave >= T) return;
  const int n_tiles = both ? 1 : W / 256, n_bands = (H + band_rows - 1) / band_rows;
  int item = blockIdx.x;
  if (order == 1) {                                   // band-major per XCD: workgroups b, b + 8, ... are neighbours in the list
    const int per = gridDim.x / 8;
    item = (blockIdx.x % 8) * per + blockIdx.x / 8;
  }
  const int tile = item % n_tiles, band = (item / n_tiles) % n_bands, f = item / (n_tiles * n_bands);
  const int h_lo = band * band_rows, rows = min(band_rows, H - h_lo);
  const int col = tile * 256 + lane * 4;
  const f32x4 v = {1.f, 2.f, 3.f, (float)wave};
  for (int g = 0; g < n_dg; ++g)
    for (int r = 0; r < rows; ++r) {
      const int h = h_lo + (skew ? (r + wave * skew) % rows : r);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int d = g * T * 2 + wave * 2 + j;
        if (d < D) {
          float* p = out + (((long)f * D + d) * H + h) * W + col;
          store16<NT>(p, v);
          if (both) store16<NT>(p + 256, v);
        }
      }
      for (int q = 0; q < pace; ++q) __builtin_amdgcn_s_sleep(1);
    }
}

static hipEvent_t ev_a, ev_b;
template <class L>
static double time_us(L launch, int warm, int reps) {
  for (int i = 0; i < warm; ++i) launch();
  hipEventRecord(ev_a);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(ev_b);
  hipEventSynchronize(ev_b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, ev_a, ev_b);
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  const long total_kb = 16L * 128 * 432 * 512 * 4 / 1024;           // 1,769,472 KB = 1.81 GB
  float* out;
  CK(hipMalloc(&out, total_kb * 1024 + (64 << 20)));
  CK(hipEventCreate(&ev_a));
  CK(hipEventCreate(&ev_b));
  const size_t lds = 100 * 1024;
  CK(hipFuncSetAttribute((const void*)streams_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)streams_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)alld_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)alld_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int warm = quick ? 10 : 30, reps = quick ? 10 : 20;
  printf("# 1.81 GB store-only sweep; every row: us per launch and TB/s, plain | nt\n");
  printf("# %-3s %-3s %-2s %-8s %-9s %-7s %-5s | %9s %6s | %9s %6s\n", "G", "T", "S", "streams", "k", "order", "pace", "plain us",
         "TB/s", "nt us", "TB/s");
  auto row = [&](Cfg c) {
    const long n_streams = (long)c.G * c.T * c.S;
    c.R = total_kb / n_streams;
    if (c.k > n_streams) c.k = n_streams;
    while (n_streams % c.k) --c.k;
    const double bytes = (double)n_streams * c.R * 1024.0;
    const double a = time_us([&] { hipLaunchKernelGGL(streams_kernel<0>, dim3(c.G), dim3(64 * c.T), lds, 0, out, c); }, warm, reps);
    const double b = time_us([&] { hipLaunchKernelGGL(streams_kernel<1>, dim3(c.G), dim3(64 * c.T), lds, 0, out, c); }, warm, reps);
    printf("  %-3d %-3d %-2d %-8ld %-9ld %-7s %-5d | %9.1f %6.2f | %9.1f %6.2f\n", c.G, c.T, c.S, n_streams, c.k,
           c.across ? "across" : "within", c.pace, a, bytes / a / 1e6, b, bytes / b / 1e6);
    fflush(stdout);
  };
  // 1. number of concurrent streams x interleave factor (k = 1: own chunk per stream ... k = all: linear fill)
  for (int T : {1, 2, 4, 8, 13, 16})
    for (int S : {1, 2}) {
      if (quick && (T == 1 || T == 2 || (T == 8 && S == 1))) continue;
      const long n = 256L * T * S;
      for (long k : {1L, 2L, 8L, 64L, 256L, 1024L, n})
        for (int across : {0, 1}) {
          if (k == 1 && across) continue;                           // (the numbering does not matter when nothing interleaves)
          if (k == n && across) continue;
          if (k > n) continue;
          row(Cfg{256, T, S, 0, k, across, 0});
        }
    }
  // 2. more than one workgroup per CU does not exist in the all-D kernel; fewer CUs writing: does the ceiling need all 256?
  for (int G : {64, 128, 192}) row(Cfg{G, 13, 2, 0, 1, 0, 0});
  // 3. paced: the same streams when every wavefront idles between steps (the kernel computes ~0.9 us per row step)
  for (int pace : {16, 24, 30, 34, 38}) {
    row(Cfg{256, 13, 2, 0, 1, 0, pace});
    row(Cfg{256, 13, 2, 0, 1L << 40, 0, pace});
  }
  // 3b. bursts: T store wavefronts per CU, 26 (or 52) streams per CU, seg KB contiguous per visit
  CK(hipFuncSetAttribute((const void*)burst_streams_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void*)burst_streams_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  printf("# bursts: G workgroups x T store wavefronts x S streams, seg KB contiguous per visit\n");
  printf("# %-4s %-3s %-3s %-4s | %9s %6s | %9s %6s\n", "G", "T", "S", "seg", "plain us", "TB/s", "nt us", "TB/s");
  for (int G : {256, 128})
    for (int T : {1, 2, 4})
      for (int seg : {1, 2, 4, 8, 16}) {
        const int S = (G == 256 ? 26 : 52) / T;
        const long n_streams = (long)G * T * S;
        long R = total_kb / n_streams / seg * seg;
        const double bytes = (double)n_streams * R * 1024.0;
        const double a = time_us([&] { hipLaunchKernelGGL(burst_streams_kernel<0>, dim3(G), dim3(64 * T), lds, 0, out, T, S, R, seg); }, warm, reps);
        const double b = time_us([&] { hipLaunchKernelGGL(burst_streams_kernel<1>, dim3(G), dim3(64 * T), lds, 0, out, T, S, R, seg); }, warm, reps);
        printf("  %-4d %-3d %-3d %-4d | %9.1f %6.2f | %9.1f %6.2f\n", G, T, S, seg, a, bytes / a / 1e6, b, bytes / b / 1e6);
        fflush(stdout);
      }
  // 4. the all-D kernel's own addressing and what its wavefronts could write at the same time instead
  printf("# all-D pattern: 13 waves x 2 planes, 5 passes, 54-row bands; skew / both tiles / pace / workgroup order\n");
  printf("# %-5s %-5s %-5s %-6s | %9s %6s | %9s %6s\n", "skew", "both", "pace", "order", "plain us", "TB/s", "nt us", "TB/s");
  const int H = 432, W = 512, D = 128, F = 16, T = 13, n_dg = 5;
  const double vol_bytes = (double)F * D * H * W * 4;
  auto arow = [&](int band_rows, int skew, int both, int pace, int order) {
    const int n_bands = (H + band_rows - 1) / band_rows;
    const int grid = (both ? 1 : 2) * n_bands * F;
    const double a = time_us([&] { hipLaunchKernelGGL(alld_kernel<0>, dim3(grid), dim3(1024), lds, 0, out, H, W, D, band_rows, n_dg, T, skew, both, pace, order); }, warm, reps);
    const double b = time_us([&] { hipLaunchKernelGGL(alld_kernel<1>, dim3(grid), dim3(1024), lds, 0, out, H, W, D, band_rows, n_dg, T, skew, both, pace, order); }, warm, reps);
    printf("  %-5d %-5d %-5d %-6d | %9.1f %6.2f | %9.1f %6.2f   (bands of %d rows, %d workgroups)\n", skew, both, pace, order, a,
           vol_bytes / a / 1e6, b, vol_bytes / b / 1e6, band_rows, grid);
    fflush(stdout);
  };
  for (int order : {0, 1})
    for (int skew : {0, 1, 4, 7}) arow(54, skew, 0, 0, order);
  arow(27, 0, 1, 0, 0);
  arow(27, 4, 1, 0, 0);
  arow(54, 0, 1, 0, 0);                      // 128 workgroups only
  for (int pace : {16, 24, 30, 34, 38}) {
    arow(54, 0, 0, pace, 0);
    arow(54, 4, 0, pace, 0);
  }
  return 0;
}
