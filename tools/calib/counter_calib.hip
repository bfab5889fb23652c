// Known-answer kernels for the two rocprofv3 counters every roofline paragraph of DESIGN.md leans
// on ("next" #2).  Diagnostic program: not part of libcap2det_hip.so, never loaded by
// the product path.
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/calib/counter_calib tools/calib/counter_calib.hip
//   bash tools/calib/run_calibration.sh          (on a GPU box: four rocprofv3 passes)
//
// (a) matrix-pipe busy: pure MFMA loops whose matrix pipe is busy 100 % (every SIMD issues
//     back-to-back MFMAs: 1 or 4 waves per SIMD) or exactly 50 % of the SIMDs (two of a block's four
//     waves leave at once) -> what SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024) reads.
// (b) HBM bytes: kernels that read / write a known byte count ONCE from a 1 GiB buffer (4x the
//     Infinity Cache) at 4, 8 and 16 B per lane, in contiguous runs of 1024 / 256 / 128 / 64 B per
//     row (the shapes of the step's kernels: float4 streams, bf16x4 streams, one dword per lane per
//     channel, 128- and 64-byte operand rows of the GEMM stages), through global loads and
//     through `buffer_load ... lds`, plus stores and float atomics -> where FETCH_SIZE needs x2.
// Prints one JSON line with the expected bytes / MFMA cycles per kernel name.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void_t;

// ---------------------------------------------------------------------------------------------
// (a) MFMA loops.  One block per CU (96 KiB of dynamic LDS keeps a second block off the CU).
// ACTIVE = number of the block's waves (of 64 lanes) that issue MFMAs; the others return at once.
// ---------------------------------------------------------------------------------------------
template <int ACTIVE>
__global__ void calib_mfma_f32(float* out, int iters) {
  extern __shared__ char lds_pad[];
  const int wave = threadIdx.x >> 6;
  if (wave % 4 >= ACTIVE && ACTIVE < 4) return;
  f32x16 acc = {0};
  float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
  if (acc[0] == 12345.678f) out[threadIdx.x] = acc[3] + lds_pad[0];
}

template <int ACTIVE>
__global__ void calib_mfma_bf16(float* out, int iters) {
  extern __shared__ char lds_pad[];
  const int wave = threadIdx.x >> 6;
  if (wave % 4 >= ACTIVE && ACTIVE < 4) return;
  f32x16 acc = {0};
  bf16x8 a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(1.0f + 0.01f * k); b[k] = (__bf16)(0.5f - 0.01f * (threadIdx.x & 7)); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  if (acc[0] == 12345.678f) out[threadIdx.x] = acc[3] + lds_pad[0];
}

// ---------------------------------------------------------------------------------------------
// (b) reads.  Every lane reads W bytes per access; a wave-instruction covers 64 * W bytes laid out
// as contiguous runs of RUN bytes, one run per "row", rows RUN_STRIDE apart (RUN = 64 * W: one
// contiguous segment).  The whole buffer is touched exactly once.
// ---------------------------------------------------------------------------------------------
template <int W> struct Vec;
template <> struct Vec<4> { typedef unsigned T; };
template <> struct Vec<8> { typedef u32x2 T; };
template <> struct Vec<16> { typedef u32x4 T; };
__device__ __forceinline__ unsigned fold(unsigned v) { return v; }
__device__ __forceinline__ unsigned fold(u32x2 v) { return v.x ^ v.y; }
__device__ __forceinline__ unsigned fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// rows of RUN bytes: the buffer is viewed as [rows][RUN]; a wave reads (64 * W / RUN) rows per
// access, and consecutive accesses of a wave go DOWN the rows — but the rows a wave reads at once
// are PERMUTED far apart (row r -> a row in another 1/LPR-th of the buffer), so that a run is the
// only contiguous piece, as in a GEMM operand tile whose rows are a leading dimension apart.
template <int W, int RUN>
__global__ __launch_bounds__(256) void calib_read(const char* __restrict__ buf, size_t bytes,
                                                  unsigned* out) {
  typedef typename Vec<W>::T T;
  constexpr int LPR = RUN / W;            // lanes per run
  constexpr int RPA = 64 / LPR;           // rows per wave access
  const size_t rows = bytes / RUN;
  const size_t part = rows / RPA;         // the j-th row of an access comes from part j
  const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int j = lane / LPR, c = lane % LPR;
  unsigned acc = 0;
  for (size_t r = wave_global; r < part; r += waves) {
    const T v = *reinterpret_cast<const T*>(buf + ((size_t)j * part + r) * RUN + (size_t)c * W);
    acc ^= fold(v);
  }
  if (acc == 0x9e3779b9u) out[0] = acc;
}

// the same through the LDS-DMA path (buffer_load_dword{,x4} ... lds): W = 4 or 16
template <int W, int RUN>
__global__ __launch_bounds__(256) void calib_read_lds(const char* __restrict__ buf, size_t bytes,
                                                      unsigned* out) {
  __shared__ __attribute__((aligned(16))) char stage[4][2][64 * 16];
  constexpr int LPR = RUN / W;
  constexpr int RPA = 64 / LPR;
  const size_t rows = bytes / RUN;
  const size_t part = rows / RPA;
  const int wave = threadIdx.x >> 6;
  const size_t wave_global = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const int lane = threadIdx.x & 63;
  const int j = lane / LPR, c = lane % LPR;
  // one descriptor per part would need RPA descriptors: instead the descriptor covers the whole
  // buffer in 2 GiB windows (32-bit offsets) — the calibration buffer is 1 GiB
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, (short)0, (int)0x7fffffff, 0x00020000);
  int slot = 0;
  for (size_t r = wave_global; r < part; r += waves) {
    const unsigned off = (unsigned)(((size_t)j * part + r) * RUN + (size_t)c * W);
    if constexpr (W == 16)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)&stage[wave][slot][0], 16, (int)off, 0, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)&stage[wave][slot][0], 4, (int)off, 0, 0, 0);
    slot ^= 1;
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*(volatile unsigned*)&stage[wave][0][lane * 4] == 0x9e3779b9u && bytes == 1) out[0] = 1;
}

// ---------------------------------------------------------------------------------------------
// (b) writes: W bytes per lane, contiguous; float atomics: one dword per lane
// ---------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(256) void calib_write(char* __restrict__ buf, size_t bytes) {
  typedef typename Vec<W>::T T;
  const size_t n = bytes / W;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  T v;
  memset(&v, 0x3c, sizeof(v));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    reinterpret_cast<T*>(buf)[i] = v;
}

__global__ __launch_bounds__(256) void calib_atomic_f32(float* __restrict__ buf, size_t bytes) {
  const size_t n = bytes / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(buf + i), 1.0f);
}

// a read-modify-write stream (the accumulate operand of a GEMM epilogue): 16 B per lane
__global__ __launch_bounds__(256) void calib_rmw16(float* __restrict__ buf, size_t bytes) {
  const size_t n = bytes / 16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    f32x4 v = reinterpret_cast<f32x4*>(buf)[i];
    v += 1.0f;
    reinterpret_cast<f32x4*>(buf)[i] = v;
  }
}

struct Timer {
  hipEvent_t a, b;
  Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); }
  void start() { CHECK(hipEventRecord(a, 0)); }
  float stop_ms() { CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }
};

static bool first_item = true;
static void report(const char* name, const char* kind, double expected, float ms, const char* extra = "") {
  printf("%s\"%s\": {\"kind\": \"%s\", \"expected\": %.0f, \"ms\": %.4f%s}", first_item ? "" : ", ", name,
         kind, expected, ms, extra);
  first_item = false;
}

template <int W, int RUN>
static void run_read(const char* buf, size_t bytes, unsigned* out, Timer& t) {
  char name[64];
  snprintf(name, sizeof name, "calib_read<%d, %d>", W, RUN);
  t.start();
  hipLaunchKernelGGL((calib_read<W, RUN>), dim3(256 * 8), dim3(256), 0, 0, buf, bytes, out);
  const float ms = t.stop_ms();
  char extra[64];
  snprintf(extra, sizeof extra, ", \"TBps\": %.3f", bytes / ms * 1e-9);
  report(name, "read", (double)bytes, ms, extra);
}

template <int W, int RUN>
static void run_read_lds(const char* buf, size_t bytes, unsigned* out, Timer& t) {
  char name[64];
  snprintf(name, sizeof name, "calib_read_lds<%d, %d>", W, RUN);
  t.start();
  hipLaunchKernelGGL((calib_read_lds<W, RUN>), dim3(256 * 8), dim3(256), 0, 0, buf, bytes, out);
  const float ms = t.stop_ms();
  char extra[64];
  snprintf(extra, sizeof extra, ", \"TBps\": %.3f", bytes / ms * 1e-9);
  report(name, "read", (double)bytes, ms, extra);
}

int main(int argc, char** argv) {
  const size_t bytes = (size_t)1 << 30;
  char* buf;
  float* out;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(buf, 0x11, bytes));
  CHECK(hipMemset(out, 0, 1 << 20));
  CHECK(hipDeviceSynchronize());
  Timer t;
  printf("{");
  // ---- (a) MFMA: ~10 ms per launch, so that GRBM_GUI_ACTIVE / 8 / time is the clock ------------
  const int lds = 96 * 1024;
  CHECK(hipFuncSetAttribute((const void*)calib_mfma_f32<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)calib_mfma_f32<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)calib_mfma_bf16<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHECK(hipFuncSetAttribute((const void*)calib_mfma_bf16<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  struct { const char* name; int threads; int active; bool bf16; int iters; } mf[] = {
      {"calib_mfma_f32<4>@1wave", 256, 4, false, 20000},   // 1 wave / SIMD, all SIMDs
      {"calib_mfma_f32<4>@4waves", 1024, 4, false, 5000},  // 4 waves / SIMD
      {"calib_mfma_f32<2>@1wave", 256, 2, false, 20000},   // half of the SIMDs
      {"calib_mfma_bf16<4>@1wave", 256, 4, true, 40000},
      {"calib_mfma_bf16<4>@4waves", 1024, 4, true, 10000},
      {"calib_mfma_bf16<2>@1wave", 256, 2, true, 40000},
  };
  for (auto& m : mf) {
    for (int rep = 0; rep < 2; ++rep) {          // (first launch: warm-up, reported too)
      t.start();
      if (!m.bf16 && m.active == 4) hipLaunchKernelGGL(calib_mfma_f32<4>, dim3(256), dim3(m.threads), lds, 0, out, m.iters);
      if (!m.bf16 && m.active == 2) hipLaunchKernelGGL(calib_mfma_f32<2>, dim3(256), dim3(m.threads), lds, 0, out, m.iters);
      if (m.bf16 && m.active == 4) hipLaunchKernelGGL(calib_mfma_bf16<4>, dim3(256), dim3(m.threads), lds, 0, out, m.iters);
      if (m.bf16 && m.active == 2) hipLaunchKernelGGL(calib_mfma_bf16<2>, dim3(256), dim3(m.threads), lds, 0, out, m.iters);
      const float ms = t.stop_ms();
      if (rep == 0) continue;
      const double waves = 256.0 * (m.threads / 64) * m.active / 4.0;
      const double mfmas = waves * m.iters * 16.0;
      const double cycles = mfmas * (m.bf16 ? 32.0 : 64.0);       // matrix-pipe cycles, summed over SIMDs
      const double flops = mfmas * (m.bf16 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2);
      char extra[160];
      snprintf(extra, sizeof extra, ", \"threads\": %d, \"simd_share\": %.2f, \"TFLOPs\": %.1f, \"mfmas\": %.0f",
               m.threads, m.active / 4.0, flops / ms * 1e-9, mfmas);
      report(m.name, "mfma_cycles", cycles, ms, extra);
    }
  }
  // ---- (b) reads ---------------------------------------------------------------------------
  run_read<16, 1024>(buf, bytes, (unsigned*)out, t);
  run_read<16, 256>(buf, bytes, (unsigned*)out, t);
  run_read<16, 128>(buf, bytes, (unsigned*)out, t);
  run_read<16, 64>(buf, bytes, (unsigned*)out, t);
  run_read<8, 512>(buf, bytes, (unsigned*)out, t);
  run_read<8, 128>(buf, bytes, (unsigned*)out, t);
  run_read<8, 64>(buf, bytes, (unsigned*)out, t);
  run_read<4, 256>(buf, bytes, (unsigned*)out, t);
  run_read<4, 128>(buf, bytes, (unsigned*)out, t);
  run_read<4, 64>(buf, bytes, (unsigned*)out, t);
  run_read_lds<16, 1024>(buf, bytes, (unsigned*)out, t);
  run_read_lds<16, 128>(buf, bytes, (unsigned*)out, t);
  run_read_lds<16, 64>(buf, bytes, (unsigned*)out, t);
  run_read_lds<4, 256>(buf, bytes, (unsigned*)out, t);
  // ---- (b) writes --------------------------------------------------------------------------
  t.start(); hipLaunchKernelGGL(calib_write<16>, dim3(2048), dim3(256), 0, 0, buf, bytes); report("calib_write<16>", "write", (double)bytes, t.stop_ms());
  t.start(); hipLaunchKernelGGL(calib_write<8>, dim3(2048), dim3(256), 0, 0, buf, bytes); report("calib_write<8>", "write", (double)bytes, t.stop_ms());
  t.start(); hipLaunchKernelGGL(calib_write<4>, dim3(2048), dim3(256), 0, 0, buf, bytes); report("calib_write<4>", "write", (double)bytes, t.stop_ms());
  CHECK(hipMemset(buf, 0, bytes));
  t.start(); hipLaunchKernelGGL(calib_atomic_f32, dim3(2048), dim3(256), 0, 0, (float*)buf, bytes / 4); report("calib_atomic_f32", "write", (double)(bytes / 4), t.stop_ms(), ", \"note\": \"also read at the memory side: expected FETCH unknown\"");
  t.start(); hipLaunchKernelGGL(calib_rmw16, dim3(2048), dim3(256), 0, 0, (float*)buf, bytes); report("calib_rmw16", "read+write", (double)bytes, t.stop_ms());
  printf("}\n");
  CHECK(hipDeviceSynchronize());
  return 0;
}
