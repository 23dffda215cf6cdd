// ctd_wave.h -- wavefront-level helpers of the loader / consumer kernels (ncc_fast.hip, lcn_stream.hip): LDS-DMA, counted
// waits, the raw workgroup barrier.
#pragma once
#include "ctd_common.h"

namespace ctd {

typedef const void __attribute__((address_space(1))) * gptr_t;
typedef void __attribute__((address_space(3))) * lptr_t;

__device__ inline void dma_dword(const float* g, float* l) {   // LDS[l + 4*lane] <- *g (per-lane address)
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, 0);
}
__device__ inline void dma_quad(const float* g, float* l) {    // LDS[l + 16*lane .. +15] <- g[0..3] (16-byte aligned)
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int N>
__device__ inline void wait_vmcnt() {   // s_waitcnt vmcnt(N) only
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0070 | 0x0F00);
}
__device__ inline void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(0xC07F); }
// raw s_barrier (no vmcnt drain, unlike __syncthreads) fenced against compiler motion of LDS accesses
__device__ inline void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// out[i] = prev_lane(sp[i]) + own + next_lane(pn[i]) for the lane's 4 columns: 8 v_add_f32_dpp.
// Without bound_ctrl a lane whose shifted source does not exist (lane 0 for wave_shr, lane 63 for wave_shl) is
// skipped by the hardware (its destination keeps the old value); with bound_ctrl:1 it reads 0, which is the
// missing neighbour's contribution, so the three-operand form needs no preset moves.
// One s_nop 1 covers the VALU-write -> DPP-read hazard of the operands (2 wait states).
__device__ inline void window_combine4(const float (&sp)[4], float own, const float (&pn)[4], float (&o)[4]) {
  // bound_ctrl:1 -- a lane whose shifted source does not exist reads 0: three-operand form, no preset moves
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %4, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %5, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %6, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %7, %12 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %0, %8, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %1, %9, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %2, %10, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_add_f32_dpp %3, %11, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])
      : "v"(sp[0]), "v"(sp[1]), "v"(sp[2]), "v"(sp[3]), "v"(pn[0]), "v"(pn[1]), "v"(pn[2]), "v"(pn[3]), "v"(own));
}

}  // namespace ctd
