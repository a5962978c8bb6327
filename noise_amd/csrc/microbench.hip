// Micro-benchmarks that measure what this GPU achieves on the two rooflines the kernels are priced against: HBM streaming
// (copy / fill) and the matrix cores (fp32-input and half-precision MFMA issue rate).  Used by scripts/peaks.py; not on
// the product path.
#include <cstdlib>

#include "common.h"

namespace {

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) dst[i] = src[i];
}

// Variants of the same copy (NOISE_COPY_MODE, diagnostic): bit 0 = non-temporal stores, bit 1 = non-temporal loads,
// bit 2 = four 16-byte pieces per thread and iteration in flight, bit 3 = one workgroup per 4 KiB instead of a persistent grid,
// bit 4 = persistent grid that requests the next pieces before it stores the current ones.
using v4f = __attribute__((ext_vector_type(4))) float;

// Persistent workgroups that take their 4 KiB pieces from a ticket counter instead of a fixed stride: the pieces in
// flight stay a compact, advancing window however far the workgroups drift apart.
__device__ unsigned long long copy_ticket;
__global__ __launch_bounds__(256) void copy_ticket_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, long n4,
                                                          unsigned long long* __restrict__ ticket, int batch) {
  __shared__ unsigned long long first;
  const long npieces = (n4 + 255) / 256;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) first = atomicAdd(ticket, (unsigned long long)batch);
    __syncthreads();
    const long p0 = (long)first;
    if (p0 >= npieces) break;            // every workgroup reaches this: the counter only grows
    for (int b = 0; b < batch; ++b) {
      const long i = (p0 + b) * 256 + threadIdx.x;
      if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    }
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void copy_variant_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, long n4,
                                                           int chunked) {
  constexpr int U = (MODE & 4) ? 4 : 1;
  if (chunked) {   // every workgroup copies one contiguous share of the buffer, front to back
    const long pieces = (n4 + 256 * U - 1) / (256 * U), per = (pieces + gridDim.x - 1) / gridDim.x;
    const long p1 = min(pieces, (long)(blockIdx.x + 1) * per);
    for (long pc = (long)blockIdx.x * per; pc < p1; ++pc) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = pc * 256 * U + 256 * u + threadIdx.x;
        if (i < n4) dst[i] = src[i];
      }
    }
    return;
  }
  const long step = (MODE & 8) ? n4 : (long)gridDim.x * 256 * U;
  if (MODE & 16) {   // persistent and software-pipelined: the next pieces are requested before the current ones are stored
    long i0 = (long)blockIdx.x * 256 * U + threadIdx.x;
    v4f cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + 256 * u;
      cur[u] = src[i < n4 ? i : n4 - 1];
    }
    for (; i0 < n4; i0 += step) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = i0 + step + 256 * u;
        nxt[u] = (MODE & 2) ? __builtin_nontemporal_load(src + (i < n4 ? i : n4 - 1)) : src[i < n4 ? i : n4 - 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long i = i0 + 256 * u;
        if (i < n4) {
          if (MODE & 1) __builtin_nontemporal_store(cur[u], dst + i);
          else dst[i] = cur[u];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
    return;
  }
  for (long i0 = (long)blockIdx.x * 256 * U + threadIdx.x; i0 < n4; i0 += step) {
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + 256 * u;
      const long ic = i < n4 ? i : n4 - 1;
      v[u] = (MODE & 2) ? __builtin_nontemporal_load(src + ic) : src[ic];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + 256 * u;
      if (i < n4) {
        if (MODE & 1) __builtin_nontemporal_store(v[u], dst + i);
        else dst[i] = v[u];
      }
    }
  }
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
  const char* mode_s = getenv("NOISE_COPY_MODE");
  const int mode = mode_s ? atoi(mode_s) : -1;
  if (mode >= 32 && mode < 32 + 64) {   // ticket counter, mode - 32 = pieces per ticket (0 -> 1)
    static unsigned long long* ticket = nullptr;
    if (ticket == nullptr && hipGetSymbolAddress((void**)&ticket, HIP_SYMBOL(copy_ticket)) != hipSuccess) {
      NOISE_REQUIRE(false, "noise_bench_copy: no ticket counter");
    }
    const char* grid_s = getenv("NOISE_COPY_GRID");
    const long persistent = grid_s ? atol(grid_s) : noise_num_cus() * 8;
    NOISE_REQUIRE(persistent > 0 && persistent < (1l << 31), "noise_bench_copy: bad NOISE_COPY_GRID");
    if (hipMemsetAsync(ticket, 0, sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess) return -1;
    const int batch = mode - 32 > 0 ? mode - 32 : 1;
    hipLaunchKernelGGL(copy_ticket_kernel, dim3((unsigned)persistent), dim3(256), 0, (hipStream_t)stream, (const v4f*)src,
                       (v4f*)dst, n / 4, ticket, batch);
    NOISE_CHECK_LAUNCH("noise_bench_copy/ticket");
    return 0;
  }
  if (mode >= 0 && mode < 24) {
    const long n4 = n / 4;
    const int U = (mode & 4) ? 4 : 1;
    const long oneshot = (n4 + 256 * U - 1) / (256 * U);
    NOISE_REQUIRE(oneshot < (1l << 31), "noise_bench_copy: too large for a one-shot grid");
    const char* grid_s = getenv("NOISE_COPY_GRID");   // workgroups of the persistent forms (default: 8 per CU)
    const long persistent = grid_s ? atol(grid_s) : noise_num_cus() * 8;
    NOISE_REQUIRE(persistent > 0 && persistent < (1l << 31), "noise_bench_copy: bad NOISE_COPY_GRID");
    const int chunked = getenv("NOISE_COPY_CHUNKED") != nullptr;   // contiguous share per workgroup instead of a stride
    const dim3 grid((unsigned)((mode & 8) ? oneshot : persistent));
#define X(M)                                                                                                            \
  if (mode == M)                                                                                                        \
    hipLaunchKernelGGL(copy_variant_kernel<M>, grid, dim3(256), 0, (hipStream_t)stream, (const v4f*)src, (v4f*)dst, n4, \
                       chunked);
    X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)
#undef X
    NOISE_CHECK_LAUNCH("noise_bench_copy/variant");
    return 0;
  }
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
