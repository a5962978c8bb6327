// BatchNorm bookkeeping for lazily normalised activations (train/models.py: every nn.BatchNorm2d
// created through Conv2dNormActivation / InvertedResidual, :371-397, :1151-1159).
//
// forward  : per-workgroup (sum, sum-of-squares) partials written by the conv kernels are folded
//            into [scale | shift | mean | invstd | mean_lo] and the running statistics are updated
//            (momentum 0.1, unbiased running variance, num_batches_tracked += 1).
// backward : one pass over (g1 [+ g2], y) applies the ReLU6 mask, writes dz and the partial sums
//            of dz and dz*xhat; the finalize kernel turns them into dgamma, dbeta and the six
//            per-channel coefficients with which consumers rebuild dL/dy (see LazyGrad in common.h).
#include "lazy.h"

namespace {

constexpr int BN_BWD_MAX_ROWS = 1024;   // grid cap of bn_bwd_reduce_kernel = rows of its partial slab

__global__ __launch_bounds__(256) void bn_finalize_fwd_kernel(const float* __restrict__ slab, int nrows, int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              long long* __restrict__ num_batches_tracked, float momentum,
                                                              float eps, float* __restrict__ coef) {
  __shared__ double red[256];
  const int c = blockIdx.x * SLAB_COLS + (threadIdx.x & (SLAB_COLS - 1));
  const bool ok = c < C;
  const int cols[2] = {c, C + c};
  // the per-channel operands are requested together with the slab rows (one memory round trip for the whole kernel)
  const int cc = ok ? c : 0;
  const float g_c = gamma[cc], b_c = beta[cc];
  const float rm_c = running_mean ? running_mean[cc] : 0.f, rv_c = running_mean ? running_var[cc] : 0.f;
  double sums[2];
  slab_column_sums<2>(slab, nrows, 2 * C, cols, ok, red, sums);
  const double s1 = sums[0], s2 = sums[1];
  if (threadIdx.x >= SLAB_COLS || !ok) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = g_c * invstd;
  coef[c] = sc;
  coef[C + c] = b_c - (float)mean * sc;
  coef[2 * C + c] = (float)mean;
  coef[3 * C + c] = invstd;
  coef[4 * C + c] = (float)(mean - (double)(float)mean);   // low part of the batch mean (used by the backward pass)
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * rm_c + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * rv_c + momentum * (float)unbiased;
  }
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
}

__global__ void bn_coef_eval_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                    float eps, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * invstd;
  coef[c] = sc;
  coef[C + c] = beta[c] - running_mean[c] * sc;
  coef[2 * C + c] = running_mean[c];
  coef[3 * C + c] = invstd;
  coef[4 * C + c] = 0.f;
}

// dz = (g1 + g2) * [0 < scale*y+shift < 6]   (mask only when relu6);  partials of sum dz, sum dz*xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                            const float* __restrict__ y, const float* __restrict__ coef,
                                                            int relu6, int C, long npix, float* __restrict__ dz,
                                                            float* __restrict__ slab) {
  extern __shared__ float red[];
  const int CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (pl < PPB) {
    const int c = 4 * cg;
    const float4 sc = ld4(coef + c), sh = ld4(coef + C + c), mean = ld4(coef + 2 * C + c), istd = ld4(coef + 3 * C + c);
    const float4 mlo = ld4(coef + 4 * C + c);
    for (long pix = (long)blockIdx.x * PPB + pl; pix < npix; pix += (long)gridDim.x * PPB) {
      const size_t off = (size_t)pix * C + c;
      float4 g = ld4(g1 + off);
      if (g2) g = g + ld4(g2 + off);
      const float4 yv = ld4(y + off);
      if (relu6) {
        const float4 v = fma4(yv, sc, sh);
        g.x = (v.x > 0.f && v.x < 6.f) ? g.x : 0.f;
        g.y = (v.y > 0.f && v.y < 6.f) ? g.y : 0.f;
        g.z = (v.z > 0.f && v.z < 6.f) ? g.z : 0.f;
        g.w = (v.w > 0.f && v.w < 6.f) ? g.w : 0.f;
      }
      if (dz) st4(dz + off, g);
      const float4 xh = make_float4(((yv.x - mean.x) - mlo.x) * istd.x, ((yv.y - mean.y) - mlo.y) * istd.y,
                                    ((yv.z - mean.z) - mlo.z) * istd.z, ((yv.w - mean.w) - mlo.w) * istd.w);
      s1 = s1 + g;
      s2 = fma4(g, xh, s2);
    }
  }
  __syncthreads();
  if (pl < PPB) {
    st4(red + ((size_t)pl * 2 + 0) * C + 4 * cg, s1);
    st4(red + ((size_t)pl * 2 + 1) * C + 4 * cg, s2);
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += 256) {
    float s = 0.f;
    for (int p = 0; p < PPB; ++p) s += red[(size_t)p * 2 * C + i];
    slab[(size_t)blockIdx.x * 2 * C + i] = s;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ slab, int nrows, int row_stride,
                                                              int C, double count,
                                                              const float* __restrict__ gamma, const float* __restrict__ coef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef3) {
  __shared__ double red[256];
  const int c = blockIdx.x * SLAB_COLS + (threadIdx.x & (SLAB_COLS - 1));
  const bool ok = c < C;
  const int cols[2] = {c, C + c};
  const int cc = ok ? c : 0;     // per-channel operands requested together with the slab rows
  const float g_c = gamma[cc], m_hi = coef[2 * C + cc], m_lo = coef[4 * C + cc], is_c = coef[3 * C + cc];
  double sums[2];
  slab_column_sums<2>(slab, nrows, row_stride, cols, ok, red, sums);
  const double s1 = sums[0], s2 = sums[1];
  if (threadIdx.x >= SLAB_COLS || !ok) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  const double mean = (double)m_hi + (double)m_lo, istd = is_c;
  const double s = (double)g_c * istd;
  const double k1 = s1 / count, k2 = s2 / count;
  // dy = s*(dz - k1) - s*k2*istd*(y - mean);  k1 and mean as hi+lo float pairs
  const float k1h = (float)k1, mh = (float)mean;
  coef3[c] = (float)s;
  coef3[C + c] = k1h;
  coef3[2 * C + c] = (float)(k1 - (double)k1h);
  coef3[3 * C + c] = (float)(-s * k2 * istd);
  coef3[4 * C + c] = mh;
  coef3[5 * C + c] = (float)(mean - (double)mh);
}

}  // namespace

NOISE_EXPORT int noise_bn_finalize_fwd(const float* slab, int slab_rows, int C, long count, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var,
                                       long long* num_batches_tracked, float momentum, float eps, float* coef,
                                       void* stream) {
  NOISE_REQUIRE(slab && gamma && beta && coef && C > 0 && slab_rows > 0 && count > 0, "noise_bn_finalize_fwd: bad argument");
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(noise_cdiv(C, SLAB_COLS)), dim3(256), 0, (hipStream_t)stream, slab, slab_rows, C,
                     (double)count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, coef);
  NOISE_CHECK_LAUNCH("noise_bn_finalize_fwd");
  return 0;
}

NOISE_EXPORT int noise_bn_coef_eval(int C, const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, float* coef, void* stream) {
  NOISE_REQUIRE(gamma && beta && running_mean && running_var && coef && C > 0, "noise_bn_coef_eval: bad argument");
  hipLaunchKernelGGL(bn_coef_eval_kernel, dim3(noise_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                     running_mean, running_var, eps, coef);
  NOISE_CHECK_LAUNCH("noise_bn_coef_eval");
  return 0;
}

// dz may alias g1 (in-place) or be null when there is neither a mask nor a second gradient.
// scratch: noise_conv_max_blocks()*2*C floats.
NOISE_EXPORT int noise_bn_bwd(const float* g1, const float* g2, const float* y, const float* coef, int relu6,
                              const float* gamma, float* dz, float* dgamma, float* dbeta, float* coef3, float* scratch,
                              long npix, int C, void* stream) {
  NOISE_REQUIRE(g1 && y && coef && gamma && dgamma && dbeta && coef3 && scratch, "noise_bn_bwd: null pointer");
  NOISE_REQUIRE(C >= 4 && C % 4 == 0 && npix > 0, "noise_bn_bwd: bad shape");
  const int PPB = 256 / (C / 4);
  const int grid = noise_grid(npix, PPB * 4, BN_BWD_MAX_ROWS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid), dim3(256), (size_t)PPB * 2 * C * sizeof(float), st, g1, g2, y, coef,
                     relu6, C, npix, dz, scratch);
  NOISE_CHECK_LAUNCH("noise_bn_bwd/reduce");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(noise_cdiv(C, SLAB_COLS)), dim3(256), 0, st, scratch, grid, 2 * C, C, (double)npix,
                     gamma, coef, dgamma, dbeta, coef3);
  NOISE_CHECK_LAUNCH("noise_bn_bwd/finalize");
  return 0;
}

// ---- noise_bn_bwd for callers that cannot allocate ---------------------------------------------------------------
// Same two kernels; the per-workgroup partial slab lives in a file-scope device array instead of a caller workspace.
// Each GPU the library runs on holds its own copy of the array (at its own address) until the process ends; the
// address is looked up once per device.  The reduce kernel writes rows [0, grid) of the slab before the finalize
// kernel reads exactly those rows, so nothing has to start at zero; calls that overlap on one GPU (two streams)
// share the array and race.
constexpr int BN_STATIC_MAX_C = 256;      // widest BatchNorm served
static __device__ __align__(256) float scratch[BN_BWD_MAX_ROWS * 2 * BN_STATIC_MAX_C];   // 2 MiB per GPU

constexpr int BN_STATIC_MAX_DEVICES = 64;
static float* g_scratch_ptr[BN_STATIC_MAX_DEVICES];   // device address of `scratch` on each GPU (null: not looked up yet)

NOISE_EXPORT int noise_bn_bwd_static(const float* g1, const float* g2, const float* y, const float* coef, int relu6,
                                     const float* gamma, float* dz, float* dgamma, float* dbeta, float* coef3,
                                     long npix, int C, void* stream) {
  NOISE_REQUIRE(C <= BN_STATIC_MAX_C, "noise_bn_bwd_static: C=%d exceeds the static workspace (at most %d channels); "
                "use noise_bn_bwd with a caller workspace", C, BN_STATIC_MAX_C);
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  NOISE_REQUIRE(e == hipSuccess && dev >= 0 && dev < BN_STATIC_MAX_DEVICES, "noise_bn_bwd_static: no current device (%s)",
                hipGetErrorString(e));
  float* ws = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (ws == nullptr) {
    e = hipGetSymbolAddress((void**)&ws, HIP_SYMBOL(scratch));
    NOISE_REQUIRE(e == hipSuccess && ws != nullptr, "noise_bn_bwd_static: hipGetSymbolAddress failed on device %d: %s", dev,
                  hipGetErrorString(e));
    __atomic_store_n(&g_scratch_ptr[dev], ws, __ATOMIC_RELEASE);   // concurrent first calls look up the same address
  }
  return noise_bn_bwd(g1, g2, y, coef, relu6, gamma, dz, dgamma, dbeta, coef3, ws, npix, C, stream);
}

// Second half of noise_bn_bwd alone, for sums that a consumer's backward epilogue already left in `slab`
// (slab_rows rows of row_stride floats, [sum dz(C) | sum dz*xhat(C)] starting at the given pointer).
NOISE_EXPORT int noise_bn_bwd_finalize(const float* slab, int slab_rows, int row_stride, const float* coef,
                                       const float* gamma, float* dgamma, float* dbeta, float* coef3, long npix, int C,
                                       void* stream) {
  NOISE_REQUIRE(slab && coef && gamma && dgamma && dbeta && coef3 && slab_rows > 0 && row_stride >= 2 * C && npix > 0,
                "noise_bn_bwd_finalize: bad argument");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(noise_cdiv(C, SLAB_COLS)), dim3(256), 0, (hipStream_t)stream, slab, slab_rows,
                     row_stride, C, (double)npix, gamma, coef, dgamma, dbeta, coef3);
  NOISE_CHECK_LAUNCH("noise_bn_bwd_finalize");
  return 0;
}
