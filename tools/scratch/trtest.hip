#include <hip/hip_runtime.h>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void k(const short* in, short* out, int rs) {
  __shared__ __attribute__((aligned(16))) short T[64 * 256];
  for (int i = threadIdx.x; i < 64 * 256; i += 64) T[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3, h = g >> 1;
  const char* base = (const char*)T + (8 * h + q) * rs + (16 * (g & 1) + 4 * p) * 2;
  v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)base);
  v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(base + 4 * rs));
  for (int e = 0; e < 4; ++e) { out[l * 8 + e] = a[e]; out[l * 8 + 4 + e] = b[e]; }
}
