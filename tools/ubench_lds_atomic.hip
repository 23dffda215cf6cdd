// LDS atomic issue-rate microbenchmark for gfx950 (not part of the product): float vs integer max, with and without
// return, against plain b32 writes and b128 reads.  4 waves per workgroup, 2 workgroups per CU-worth of grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = -1e30f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned addr = (unsigned)(size_t)s + lane * 4;        // conflict-free b32, all waves on the same slots (as the kernel)
  unsigned addr16 = (unsigned)(size_t)s + lane * 16 + wave * 1024;
  float v0 = lane + wave * 0.25f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  float4 q = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP8(asm volatile("ds_max_f32 %0, %1\n ds_max_f32 %0, %2 offset:256\n ds_max_f32 %0, %3 offset:512\n ds_max_f32 %0, %4 offset:768\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (MODE == 1) { REP8(asm volatile("ds_max_rtn_f32 %0, %4, %5\n ds_max_rtn_f32 %1, %4, %6 offset:256\n ds_max_rtn_f32 %2, %4, %7 offset:512\n ds_max_rtn_f32 %3, %4, %8 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (MODE == 2) { REP8(asm volatile("ds_max_i32 %0, %1\n ds_max_i32 %0, %2 offset:256\n ds_max_i32 %0, %3 offset:512\n ds_max_i32 %0, %4 offset:768\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (MODE == 3) { REP8(asm volatile("ds_max_rtn_i32 %0, %4, %5\n ds_max_rtn_i32 %1, %4, %6 offset:256\n ds_max_rtn_i32 %2, %4, %7 offset:512\n ds_max_rtn_i32 %3, %4, %8 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (MODE == 4) { REP8(asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %2 offset:256\n ds_write_b32 %0, %3 offset:512\n ds_write_b32 %0, %4 offset:768\n s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3));) }
    if (MODE == 5) { REP8(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1 offset:4096\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1 offset:8192\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %1 offset:12288\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr16));) }
    if (MODE == 6) { REP8(asm volatile("ds_max_rtn_u64 %0, %2, %3\n ds_max_rtn_u64 %1, %2, %4 offset:512\n s_waitcnt lgkmcnt(0)" : "=v"(*(double*)&r0), "=v"(*(double*)&r2) : "v"((unsigned)(size_t)s + lane * 8), "v"(*(double*)&v0), "v"(*(double*)&v2));) }
    v0 += 1e-3f; v1 += 1e-3f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + q.x + q.w + s[threadIdx.x];
}
template <int MODE> void run(const char* name, float* d, int per_rep) {
  int blocks = 256 * 4, iters = 100;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 5); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double insts = (double)blocks * 4 * iters * 8 * per_rep;        // wave instructions
  printf("%-18s %8.3f ms  %.1f nominal-cycles per wave-instruction per CU\n", name, ms, ms * 1e-3 * 2.4e9 * 256 / insts);
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<4>("ds_write_b32", d, 4); run<0>("ds_max_f32", d, 4); run<1>("ds_max_rtn_f32", d, 4); run<2>("ds_max_i32", d, 4);
  run<3>("ds_max_rtn_i32", d, 4); run<6>("ds_max_rtn_u64", d, 2); run<5>("ds_read_b128", d, 4);
  return 0;
}
