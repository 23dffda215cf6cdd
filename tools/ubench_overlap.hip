// Does VALU work overlap with HBM-saturating stores?  (a) same waves compute then store,
// (b) specialised waves: half the waves of a block only compute, the other half only store.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline float spin_work(float x, int spin) {
  float a = x, b = x + 1, c = x + 2, d = x + 3;       // 4 independent chains
  for (int s = 0; s < spin; ++s) { a = a * 1.0001f + 0.5f; b = b * 1.0001f + 0.5f; c = c * 1.0001f + 0.5f; d = d * 1.0001f + 0.5f; }
  return a + b + c + d;
}
// each wave: rows of 1 KB aligned stores, 2 per row (like the volume kernel), `spin` FMAs x4 per row
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, float* sink, long rows_per_wave, int spin) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 8 + wave;
  float acc = 0;
  if (MODE == 0) {            // all 8 waves: compute + store
    float4* o = (float4*)out + gw * rows_per_wave * 2 * 64;
    for (long r = 0; r < rows_per_wave; ++r) {
      float x = spin_work((float)r, spin);
      o[(r * 2) * 64 + lane] = make_float4(x, 1, 2, 3);
      o[(r * 2 + 1) * 64 + lane] = make_float4(x, 4, 5, 6);
    }
  } else if (MODE == 1) {     // waves 0-3 compute only (2x rows), waves 4-7 store only (2x rows)
    if (wave < 4) {
      for (long r = 0; r < 2 * rows_per_wave; ++r) acc += spin_work((float)r, spin);
    } else {
      float4* o = (float4*)out + ((long)blockIdx.x * 4 + (wave - 4)) * rows_per_wave * 4 * 64;
      for (long r = 0; r < 2 * rows_per_wave; ++r) {
        o[(r * 2) * 64 + lane] = make_float4((float)r, 1, 2, 3);
        o[(r * 2 + 1) * 64 + lane] = make_float4((float)r, 4, 5, 6);
      }
    }
  } else if (MODE == 2) {     // compute only
    for (long r = 0; r < rows_per_wave; ++r) acc += spin_work((float)r, spin);
  } else {                    // store only
    float4* o = (float4*)out + gw * rows_per_wave * 2 * 64;
    for (long r = 0; r < rows_per_wave; ++r) {
      o[(r * 2) * 64 + lane] = make_float4((float)r, 1, 2, 3);
      o[(r * 2 + 1) * 64 + lane] = make_float4((float)r, 4, 5, 6);
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}
template <int MODE> float run(float* d, float* sink, int blocks, long rows, int spin) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 512>>>(d, sink, rows, spin); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 3; ++i) k<MODE><<<blocks, 512>>>(d, sink, rows, spin); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 3;
}
int main() {
  const int blocks = 1024; const long rows = 110;    // 1024 x 8 waves x 110 rows x 2 KB = 1.8 GB
  size_t bytes = (size_t)blocks * 8 * rows * 2048;
  float *d, *sink; hipMalloc(&d, bytes); hipMalloc(&sink, 4);
  for (int spin : {16, 32, 64, 128}) {
    float both = run<0>(d, sink, blocks, rows, spin), spec = run<1>(d, sink, blocks, rows, spin), comp = run<2>(d, sink, blocks, rows, spin), st = run<3>(d, sink, blocks, rows, spin);
    printf("spin=%3d  compute-only %.3f ms | store-only %.3f ms (%.2f TB/s) | same-wave compute+store %.3f ms | specialised waves %.3f ms\n", spin, comp, st, bytes / st / 1e9, both, spec);
  }
  return 0;
}
