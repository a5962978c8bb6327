// Crop-from-the-end of a skip tensor (train/models.py:608: `x + skip[..., -h:, -w:]`), gfx950.
//
// When the searched padding was found for another input length, an encoder output is larger than the
// decoder tensor it is added to and the reference keeps its last h rows / w columns.  The decoder
// kernels read their second source with the first source's geometry, so the window is copied once into
// a contiguous NHWC buffer (raw values: the lazy BatchNorm coefficients are per channel and carry over).
#include "lazy.h"

namespace {

__global__ __launch_bounds__(256) void crop_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int Hs,
                                                        int Ws, int Hd, int Wd, int oy, int ox, int C4, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long p = i / C4;
    const int x = (int)(p % Wd);
    p /= Wd;
    const int y = (int)(p % Hd);
    const long b = p / Hd;
    const size_t so = (((size_t)b * Hs + (y + oy)) * Ws + (x + ox)) * C4 + c;
    reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[so];
  }
}

// Adjoint of the crop: the gradient of the kept window, zero on the rows and columns that were cut off.
__global__ __launch_bounds__(256) void crop_nhwc_bwd_kernel(const float* __restrict__ g_dst, float* __restrict__ g_src, int Hs,
                                                            int Ws, int Hd, int Wd, int oy, int ox, int C4, long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long p = i / C4;
    const int x = (int)(p % Ws) - ox;
    p /= Ws;
    const int y = (int)(p % Hs) - oy;
    const long b = p / Hs;
    const bool in = x >= 0 && y >= 0;   // the window reaches the last row and column
    const size_t go = (((size_t)b * Hd + (in ? y : 0)) * Wd + (in ? x : 0)) * C4 + c;
    const float4 g = reinterpret_cast<const float4*>(g_dst)[go];
    reinterpret_cast<float4*>(g_src)[i] = in ? g : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

}  // namespace

// g_src [B][Hs][Ws][C] = gradient of noise_crop_nhwc's source given g_dst [B][Hd][Wd][C], the gradient of its window
// (which ends at the last row and column of the source: oy + Hd == Hs, ox + Wd == Ws); every element is written.
NOISE_EXPORT int noise_crop_nhwc_bwd(const float* g_dst, float* g_src, int B, int Hs, int Ws, int C, int oy, int ox, int Hd,
                                     int Wd, void* stream) {
  NOISE_REQUIRE(g_dst && g_src && B > 0 && C > 0 && C % 4 == 0, "noise_crop_nhwc_bwd: bad argument (C=%d must be a multiple of 4)", C);
  NOISE_REQUIRE(oy >= 0 && ox >= 0 && Hd > 0 && Wd > 0 && oy + Hd == Hs && ox + Wd == Ws,
                "noise_crop_nhwc_bwd: window %dx%d at (%d,%d) must end at the last row and column of the %dx%d source", Hd, Wd,
                oy, ox, Hs, Ws);
  const long total = (long)B * Hs * Ws * (C / 4);
  hipLaunchKernelGGL(crop_nhwc_bwd_kernel, dim3(noise_grid(total, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, g_dst,
                     g_src, Hs, Ws, Hd, Wd, oy, ox, C / 4, total);
  NOISE_CHECK_LAUNCH("noise_crop_nhwc_bwd");
  return 0;
}

// dst [B][Hd][Wd][C] = src [B][Hs][Ws][C] at rows oy.., columns ox..
NOISE_EXPORT int noise_crop_nhwc(const float* src, float* dst, int B, int Hs, int Ws, int C, int oy, int ox, int Hd,
                                 int Wd, void* stream) {
  NOISE_REQUIRE(src && dst && B > 0 && C > 0 && C % 4 == 0, "noise_crop_nhwc: bad argument (C=%d must be a multiple of 4)", C);
  NOISE_REQUIRE(oy >= 0 && ox >= 0 && Hd > 0 && Wd > 0 && oy + Hd <= Hs && ox + Wd <= Ws,
                "noise_crop_nhwc: window %dx%d at (%d,%d) leaves the %dx%d source", Hd, Wd, oy, ox, Hs, Ws);
  const long total = (long)B * Hd * Wd * (C / 4);
  hipLaunchKernelGGL(crop_nhwc_kernel, dim3(noise_grid(total, 256 * 4, 4096)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     Hs, Ws, Hd, Wd, oy, ox, C / 4, total);
  NOISE_CHECK_LAUNCH("noise_crop_nhwc");
  return 0;
}
