// Shared helpers for the cap2det HIP kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cap2det_hip.h"

#define C2D_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return C2D_ERR_INVALID_ARG; \
  } while (0)

static inline int c2d_launch_status() {
  return hipGetLastError() == hipSuccess ? C2D_OK : C2D_ERR_LAUNCH;
}

static inline int c2d_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Wave64 reductions (CDNA4 wavefront = 64 lanes).
__device__ __forceinline__ float c2d_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float c2d_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float c2d_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
