// ctd_common.h -- shared host/device helpers of libctd_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/ctd_hip.h"

#define CTD_HIP_TRY(expr)                                   \
  do {                                                      \
    hipError_t e_ = (expr);                                 \
    if (e_ != hipSuccess) return CTD_ERR_HIP + (int)e_;     \
  } while (0)

// Every launch is followed by a non-blocking error peek; nothing synchronises.
#define CTD_LAUNCH_CHECK()                                  \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return CTD_ERR_HIP + (int)e_;     \
  } while (0)

namespace ctd {

// Selects the device the pointers live on for the duration of one entry point and
// restores the caller's device afterwards (the reference had no device guard).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  int status = CTD_OK;
  explicit DeviceGuard(int device) {
    if (device < 0) return;
    hipError_t e = hipGetDevice(&prev);
    if (e != hipSuccess) { status = CTD_ERR_HIP + (int)e; return; }
    if (prev != device) {
      e = hipSetDevice(device);
      if (e != hipSuccess) { status = CTD_ERR_HIP + (int)e; return; }
      switched = true;
    }
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

__host__ __device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// work list of the ranked argmax (device helpers and the why: ctd_rank.h)
constexpr int kWorkListStride = 64;            // unsigned between two counters
constexpr int kWorkListParts = 16;
struct WorkList {
  unsigned* counters;                          // `parts` counters, kWorkListStride apart
  int64_t* list;                               // `parts` segments of `seg_cap` entries
  int parts;                                   // power of two (1: one plain list)
  int row_width;                               // key = (pixel / row_width) & (parts - 1)
  long seg_cap;
};

}  // namespace ctd
