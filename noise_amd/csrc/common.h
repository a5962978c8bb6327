// Shared host/device helpers for the noise_hip kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#define NOISE_EXPORT extern "C" __attribute__((visibility("default")))

// ---- error plumbing: every entry point returns 0 on success; the text of the last failure
// on the calling thread is available through noise_last_error().
void noise_set_error(const char* fmt, ...);

#define NOISE_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      noise_set_error(__VA_ARGS__);         \
      return -1;                            \
    }                                       \
  } while (0)

#define NOISE_CHECK_LAUNCH(name)                                                   \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      noise_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));      \
      return (int)e__;                                                             \
    }                                                                              \
  } while (0)

// Kernels that stage more than 64 KiB of dynamic LDS opt in explicitly (gfx950 has 160 KiB per CU).
#define NOISE_ALLOW_LDS(kernel, bytes)                                                                         \
  do {                                                                                                         \
    if ((bytes) > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)(bytes));                                                                 \
  } while (0)

static inline int noise_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Memory-bound elementwise/reduction kernels: cap the grid and grid-stride the rest.
static inline int noise_grid(long work_items, int per_block, int cap = 2048) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---- lazily normalised activations --------------------------------------------------------
// A convolution writes its RAW output y; BatchNorm (+ReLU6) is applied by whoever reads it:
//     a[p][c] = act(scale[c] * y[p][c] + shift[c])
// `coef` points at [scale(C) | shift(C) | mean(C) | invstd(C)] or is null for a plain tensor.
struct LazySrc {
  const float* y;
  const float* coef;
  int relu6;
};

// Gradient w.r.t. a raw conv output, reconstructed on the fly (BatchNorm backward, centred form):
//     dy[p][c] = s[c]*(dz[p][c] - k1[c]) + a2[c]*(y[p][c] - mean[c])
// where dz is the (ReLU6-masked) gradient w.r.t. the BatchNorm output.  `coef3` points at six vectors
// [s | k1_hi | k1_lo | a2 | mean_hi | mean_lo] (C floats each) or is null when dy == dz.
struct LazyGrad {
  const float* dz;
  const float* y;
  const float* coef3;
};

// What a backward kernel does with the input gradient g it has just computed for source A:
//   g += g_other (second gradient of the same tensor, e.g. the decoder's skip branch)     [optional]
//   g  = g * [0 < scale*y_A + shift < 6]            (ReLU6 of the producer, when relu6)    [needs coef]
//   partial sums of g and g*xhat_A go to stats_slab                                       [needs coef]
// so that the producer's BatchNorm backward needs no pass of its own.  coef is the PRODUCER's
// [scale|shift|mean|invstd|mean_lo]; all members may be null / 0.
struct BwdEpilogue {
  const float* g_other;
  const float* coef;
  int relu6;
  float* stats_slab;
};

#ifdef __HIPCC__
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float relu6f(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_reduce_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Column sums of a [nrows][row_stride] fp32 slab in double.  Call with 256 threads laid out as
// 32 columns x 8 row lanes (col = first_col + (tid & 31), row lane = tid >> 5): reads are 128-byte
// coalesced and every thread walks nrows/8 rows.  `red` is 256 doubles of LDS; the result is valid in
// the threads with tid < 32.  Contains barriers: every thread of the block must call it.
__device__ __forceinline__ double slab_column_sum(const float* __restrict__ slab, int nrows, int row_stride, int col,
                                                  bool col_valid, double* red) {
  const int rl = threadIdx.x >> 5;
  double s = 0.0;
  if (col_valid) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four independent chains keep several loads in flight
    int r = rl;
    for (; r + 24 < nrows; r += 32) {
      s0 += (double)slab[(size_t)r * row_stride + col];
      s1 += (double)slab[(size_t)(r + 8) * row_stride + col];
      s2 += (double)slab[(size_t)(r + 16) * row_stride + col];
      s3 += (double)slab[(size_t)(r + 24) * row_stride + col];
    }
    for (; r < nrows; r += 8) s0 += (double)slab[(size_t)r * row_stride + col];
    s = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[threadIdx.x + 32 * k];
  }
  return s;
}
#endif
