// Streaming-read microbenchmark (not part of the product): what read bandwidth can a volume sweep reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
// MODE 0: scan pattern: thread <-> 4 pixels, loop over D planes HW apart (like argmax_scan_kernel)
// MODE 1: fully sequential: each wave streams a contiguous slab
// MODE 2: scan pattern, but a block walks only D/SPLIT planes (more, shorter blocks)
template <int MODE, int UNROLL, bool NT>
__global__ __launch_bounds__(256) void k(const float* __restrict__ vol, float* __restrict__ out, int D, long HW, long total4, int split) {
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 1) {
    const long nthreads = (long)gridDim.x * blockDim.x;
    const long n4 = total4 * D;
    const f4* p = (const f4*)vol;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, nwaves = nthreads >> 6;
    const long per = (n4 / 64 + nwaves - 1) / nwaves;
    for (long i = 0; i < per; i += UNROLL) {
      f4 x[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { long j = (wave * per + i + u) * 64 + lane; x[u] = j < n4 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : acc; }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc = __builtin_elementwise_max(acc, x[u]);
    }
  } else {
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int d0 = 0, d1 = D;
    if (MODE == 2) { const long per = (total4 + 255) / 256 * 256; d0 = (int)(t / per) * (D / split); d1 = d0 + D / split; t = t % per; }
    if (t >= total4) return;
    const long p0 = t * 4, f = p0 / HW, q0 = p0 - f * HW;
    const float* v = vol + f * D * HW + q0;
    for (int d = d0; d < d1; d += UNROLL) {
      f4 x[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { const f4* a = (const f4*)(v + (long)(d + u) * HW); x[u] = NT ? __builtin_nontemporal_load(a) : *a; }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc = __builtin_elementwise_max(acc, x[u]);
    }
  }
  if (acc.x == 12345.f) out[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <int MODE, int UNROLL, bool NT> void run(const char* name, const float* vol, float* out, int N, int D, int H, int W, int blocks, int split) {
  const long HW = (long)H * W, total4 = (long)N * HW / 4;
  long nb = MODE == 1 ? blocks : (total4 + 255) / 256 * (MODE == 2 ? split : 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, UNROLL, NT><<<dim3((unsigned)nb), 256>>>(vol, out, D, HW, total4, split); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 5; ++i) k<MODE, UNROLL, NT><<<dim3((unsigned)nb), 256>>>(vol, out, D, HW, total4, split); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("%-40s blocks=%6ld %8.3f ms  %.2f TB/s\n", name, nb, ms, (double)N * D * HW * 4 / ms / 1e9);
}
int main() {
  const int N = 16, D = 128, H = 432, W = 512;
  float *vol, *out; size_t bytes = (size_t)N * D * H * W * 4;
  hipMalloc(&vol, bytes); hipMalloc(&out, 4096); hipMemset(vol, 0, bytes);
  run<0, 8, false>("scan pattern unroll 8", vol, out, N, D, H, W, 0, 1);
  run<0, 16, false>("scan pattern unroll 16", vol, out, N, D, H, W, 0, 1);
  run<0, 4, false>("scan pattern unroll 4", vol, out, N, D, H, W, 0, 1);
  run<0, 8, true>("scan pattern unroll 8 nontemporal", vol, out, N, D, H, W, 0, 1);
  run<2, 8, false>("scan pattern split 2", vol, out, N, D, H, W, 0, 2);
  run<2, 8, false>("scan pattern split 4", vol, out, N, D, H, W, 0, 4);
  run<2, 8, false>("scan pattern split 8", vol, out, N, D, H, W, 0, 8);
  run<1, 8, false>("sequential slabs, 2048 blocks", vol, out, N, D, H, W, 2048, 1);
  run<1, 8, false>("sequential slabs, 4096 blocks", vol, out, N, D, H, W, 4096, 1);
  run<1, 8, false>("sequential slabs, 16384 blocks", vol, out, N, D, H, W, 16384, 1);
  run<1, 4, false>("sequential slabs u4, 65536 blocks", vol, out, N, D, H, W, 65536, 1);
  run<1, 8, true>("sequential slabs nt, 4096 blocks", vol, out, N, D, H, W, 4096, 1);
  run<1, 16, false>("sequential slabs u16, 2048 blocks", vol, out, N, D, H, W, 2048, 1);
  return 0;
}
