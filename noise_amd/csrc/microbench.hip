// Micro-benchmarks that measure what this GPU achieves on the two rooflines the kernels are priced against: HBM streaming
// (copy / fill) and the matrix cores (fp32-input and half-precision MFMA issue rate).  Used by scripts/peaks.py; not on
// the product path.
#include "common.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = src[i];
}

// every wave issues `iters` x 4 independent MFMAs back to back; the result is stored so nothing is optimised away
template <int KIND>
__global__ __launch_bounds__(256) void mfma_kernel(float* __restrict__ out, int iters, float seed) {
  if (KIND == 2) {   // v_mfma_f32_32x32x2_f32: 16 accumulator registers per instruction
    f32x16 big[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) big[j][e] = e == 0 ? seed : 0.f;
    const float a = seed * (float)(threadIdx.x & 7), b = 1.f - seed;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) big[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += big[j][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    return;
  }
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{seed, 0.f, 0.f, 0.f};
  const float a = seed * (float)(threadIdx.x & 7), b = 1.f - seed;
  f16x8 ha, hb;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ha[j] = (_Float16)(a + (float)j * 0.125f);
    hb[j] = (_Float16)(b - (float)j * 0.0625f);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

// dst = src, n floats (n % 4 == 0): 2 * 4 * n bytes of HBM traffic
NOISE_EXPORT int noise_bench_copy(const float* src, float* dst, long n, void* stream) {
  NOISE_REQUIRE(src && dst && n > 0 && n % 4 == 0, "noise_bench_copy: bad argument");
  hipLaunchKernelGGL(copy_kernel, dim3(noise_num_cus() * 8), dim3(256), 0, (hipStream_t)stream, (const float4*)src, (float4*)dst,
                     n / 4);
  NOISE_CHECK_LAUNCH("noise_bench_copy");
  return 0;
}

// kind 0: v_mfma_f32_16x16x4_f32 (2048 flop per wave-instruction); kind 1: v_mfma_f32_16x16x32_f16 (16384 flop);
// kind 2: v_mfma_f32_32x32x2_f32 (4096 flop).
// blocks x 4 waves x iters x 4 instructions; out: blocks * 256 floats.
NOISE_EXPORT int noise_bench_mfma(int kind, float* out, int blocks, int iters, void* stream) {
  NOISE_REQUIRE(out && blocks > 0 && iters > 0 && (kind >= 0 && kind <= 2), "noise_bench_mfma: bad argument");
  if (kind == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.5f);
  else if (kind == 1) hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.5f);
  else hipLaunchKernelGGL(mfma_kernel<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.5f);
  NOISE_CHECK_LAUNCH("noise_bench_mfma");
  return 0;
}
