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

// Tuning / ablation hooks of the tools (tools/sweep_*.sh, tools/ablate_x9.sh ...): ONE environment
// variable, C2D_TUNE="key=value,key=value", parsed once at first use; unset in production, where
// every hook takes its default.  c2d_tune_get: the value string of `key`, or null.
bool c2d_tune_on();
const char* c2d_tune_get(const char* key);

// CUs this process may count on (c2d_set_available_cus; default 256 = the whole MI355X).  The launch
// plans that size a launch as "one round of resident workgroups" — split counts of the filter
// gradients, the small-problem threshold — scale their budgets with it: a rank whose RCCL channel
// kernels hold CUs under the backward pass would otherwise launch a round and a bit.
int c2d_available_cus();
static inline int c2d_cu_scaled(int per_256_cus) {
  const int n = c2d_available_cus();
  const int v = (int)((long long)per_256_cus * n / 256);
  return v > 0 ? v : 1;
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

// ---- element access for the two activation storage types (fp32 / bf16): 4 channels per lane ----
typedef __bf16 c2d_bf16;
typedef __bf16 c2d_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 c2d_ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ float4 c2d_ld4(const c2d_bf16* p) {
  const c2d_bf16x4 v = *reinterpret_cast<const c2d_bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void c2d_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void c2d_st4(c2d_bf16* p, float4 v) {
  c2d_bf16x4 o;
  o[0] = (c2d_bf16)v.x; o[1] = (c2d_bf16)v.y; o[2] = (c2d_bf16)v.z; o[3] = (c2d_bf16)v.w;
  *reinterpret_cast<c2d_bf16x4*>(p) = o;
}
