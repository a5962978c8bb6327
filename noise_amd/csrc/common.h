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
static inline int noise_pad8(long n) { return (int)((n + 7) / 8 * 8); }   // grid size for xcd_chunked_block()

// Memory-bound elementwise/reduction kernels: cap the grid and grid-stride the rest.
static inline int noise_grid(long work_items, int per_block, int cap = 2048) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// A persistent grid larger than what the GPU holds at once runs in rounds, and a last round that is only partly full
// costs as much as a full one: round the grid down to a whole number of rounds (CUs x resident workgroups per CU, as
// the runtime reports them for this kernel, block size and dynamic LDS).
static inline int noise_num_cus() {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    return n;
  }();
  return cus;
}
template <typename KernelT>
static inline int noise_whole_rounds(int grid, KernelT kernel, int threads, size_t smem) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, smem) != hipSuccess || per_cu < 1) return grid;
  const int round = noise_num_cus() * per_cu;
  return grid > round ? grid / round * round : grid;
}
#define NOISE_LAUNCH_ROUNDS(kernel, grid, smem, st, ...)                               \
  do {                                                                                 \
    auto kfn_ = kernel;                                                                \
    grid = noise_whole_rounds(grid, kfn_, 256, smem);                                  \
    hipLaunchKernelGGL(kfn_, dim3(grid), dim3(256), smem, st, __VA_ARGS__);            \
  } while (0)

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

// Workgroups are dealt round-robin over the 8 XCDs (observed behaviour, used for L2 locality only, never for
// correctness): blocks b and b + 8 share an L2.  With a grid that is a multiple of 8 this maps blockIdx.x to a virtual
// index such that CONSECUTIVE virtual indices run on the same XCD, so neighbours that read the same cache lines fetch
// them from HBM once instead of once per XCD.  The launcher pads the grid to a multiple of 8; padding blocks get an
// index >= the real count and return at once.
__device__ __forceinline__ int xcd_chunked_block() { return (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)); }

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

// Column sums of a slab of per-workgroup partials: slab[r][col], r < nrows (row_stride floats apart), in double.
// A block covers SLAB_COLS columns x (256 / SLAB_COLS) row lanes (col = first_col + (tid & (SLAB_COLS-1)), row
// lane = tid / SLAB_COLS): a row lane reads 32 contiguous bytes of every 32nd row.  The kernels that call this are
// pure latency (a few workgroups, a slab that the previous kernel has just written): what they cost is the number of
// DEPENDENT memory round trips, so a lane requests SLAB_BATCH rows of every summed column before it adds the first one
// (clamped addresses, no branch around the loads): a 1024-row slab is two round trips instead of eight (measured: 6.6 ->
// 5.8 us per finalize launch; a deeper batch and requesting the per-channel operands ahead add nothing: what is left is
// the launch boundary behind a kernel that leaves megabytes of dirty lines in the L2s).  The additions
// run in a fixed order, so the result does not change from run to run.  `red` is 256 doubles of LDS per summed column
// set; the result is valid in the threads with tid < SLAB_COLS.  Contains barriers: every thread of the block must
// call it.
#define SLAB_COLS 8
#define SLAB_BATCH 16
template <int NSUM>
__device__ __forceinline__ void slab_column_sums(const float* __restrict__ slab, int nrows, int row_stride,
                                                 const int (&col)[NSUM], bool col_valid, double* red, double (&out)[NSUM]) {
  constexpr int LANES = 256 / SLAB_COLS;
  const int rl = threadIdx.x / SLAB_COLS;
  double acc[NSUM];
#pragma unroll
  for (int j = 0; j < NSUM; ++j) acc[j] = 0.0;
  if (col_valid) {
    for (int r0 = rl; r0 < nrows; r0 += SLAB_BATCH * LANES) {
      float v[NSUM][SLAB_BATCH];
#pragma unroll
      for (int u = 0; u < SLAB_BATCH; ++u) {
        const int r = r0 + u * LANES;
        const size_t row = (size_t)(r < nrows ? r : nrows - 1) * row_stride;
#pragma unroll
        for (int j = 0; j < NSUM; ++j) v[j][u] = slab[row + col[j]];
      }
#pragma unroll
      for (int u = 0; u < SLAB_BATCH; ++u) {
        const bool in = r0 + u * LANES < nrows;
#pragma unroll
        for (int j = 0; j < NSUM; ++j) acc[j] += in ? (double)v[j][u] : 0.0;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NSUM; ++j) {
    __syncthreads();
    red[threadIdx.x] = acc[j];
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x < SLAB_COLS) {
      for (int k = 0; k < LANES; ++k) s += red[threadIdx.x + SLAB_COLS * k];
    }
    out[j] = s;
  }
}

__device__ __forceinline__ double slab_column_sum(const float* __restrict__ slab, int nrows, int row_stride, int col,
                                                  bool col_valid, double* red) {
  const int cols[1] = {col};
  double out[1];
  slab_column_sums<1>(slab, nrows, row_stride, cols, col_valid, red, out);
  return out[0];
}
#endif
