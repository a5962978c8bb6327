// 3x3 stencil layers of the denoiser U-Net (NHWC fp32), gfx950.
//
// Replaces, per reference layer (train/models.py):
//   * depthwise Conv2d k3 p1 s{1,2}  + BN + ReLU6            :385-393 (InvertedResidual "dw")
//   * depthwise ConvTranspose2d k3 p1 s{1,2} + BN + ReLU6    :299-345 used at :1193-1201
//   * first Conv2d 1->C0 k3 s2 p1                            :1151-1159 (with :589 pad, :614-622 normalise)
//   * last ConvTranspose2d C0->1 k3 s2 p1 with bias          :1164-1171 (with :610 crop, :762-763 mask)
// Every kernel writes the RAW convolution output; BatchNorm(+ReLU6) of the producer is applied by
// the consumer's loader (LazySrc), and the per-channel sum / sum-of-squares needed for train-mode
// batch statistics come out of the same pass as per-workgroup partials (no atomics).
#include "lazy.h"
#include "noise_hip.h"

namespace {

struct ConvDims {
  int B, Hi, Wi, Ho, Wo, C, sy, sx;
};

// Pixel range [start, end) of this workgroup: contiguous chunks (a multiple of the PPB pixels one pass covers),
// so the rows a 3x3 stencil shares between neighbouring pixels are re-read by the same CU a few passes
// later (from its L1 / its XCD's L2) instead of by workgroups on other XCDs.
__device__ __forceinline__ void pixel_chunk(int npix, int PPB, int& start, int& end) {
  int chunk = (npix + (int)gridDim.x - 1) / (int)gridDim.x;
  chunk = (chunk + PPB - 1) / PPB * PPB;
  const long s0 = (long)blockIdx.x * chunk;
  start = s0 < npix ? (int)s0 : npix;
  end = s0 + chunk < npix ? (int)(s0 + chunk) : npix;
}

// Reduce per-thread float4 partials over the pixel lanes of a workgroup and write one row of the
// slab: slab[block][j*C + c] for j < NV.  Threads are laid out tid = pl*CG + cg.
template <int NV>
__device__ __forceinline__ void block_channel_reduce(const float4 (&v)[NV], int C, int CG, int PPB, float* red,
                                                     float* slab_row) {
  const int tid = threadIdx.x;
  const int pl = tid / CG, cg = tid - pl * CG;
  __syncthreads();
  if (pl < PPB) {
#pragma unroll
    for (int j = 0; j < NV; ++j) st4(red + ((size_t)pl * NV + j) * C + 4 * cg, v[j]);
  }
  __syncthreads();
  for (int i = tid; i < NV * C; i += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < PPB; ++p) s += red[(size_t)p * NV * C + i];
    slab_row[i] = s;
  }
}

}  // namespace
#include "stencil_walk.h"
namespace {

// ------------------------------------------------------------------------------------------------
// Generic one-thread-per-pixel kernels: any mix of strides 1 and 2 (the column walkers above take the
// equal-stride layers, which are all the reference network has).
// depthwise forward: out[o][c] = sum_k in[src(o,k)][c] * w[c][k]
template <int GEOM>
__global__ __launch_bounds__(256) void dw_fwd_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                     float* __restrict__ out, float* __restrict__ slab, ConvDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  const bool active = pl < PPB;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (active) {
    const int c = 4 * cg;
    const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = make_float4(w[(c + 0) * 9 + k], w[(c + 1) * 9 + k], w[(c + 2) * 9 + k], w[(c + 3) * 9 + k]);
    int p0, p1;
    pixel_chunk(d.B * d.Ho * d.Wo, PPB, p0, p1);
    for (int pix = p0 + pl; pix < p1; pix += PPB) {
      const int ox = pix % d.Wo;
      const int r = pix / d.Wo;
      const int oy = r % d.Ho, bi = r / d.Ho;
      float4 acc = f4(0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = tap_src<GEOM>(oy, ky, d.sy, d.Hi);
        if (iy < 0) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = tap_src<GEOM>(ox, kx, d.sx, d.Wi);
          if (ix < 0) continue;
          const size_t off = (((size_t)bi * d.Hi + iy) * d.Wi + ix) * C + c;
          float4 v = lazy4_at(a, off);
          if (b.present) v = v + lazy4_at(b, off);
          acc = fma4(v, wk[ky * 3 + kx], acc);
        }
      }
      st4(out + (size_t)pix * C + c, acc);
      s1 = s1 + acc;
      s2 = fma4(acc, acc, s2);
    }
  }
  if (slab != nullptr) {
    const float4 v[2] = {s1, s2};
    block_channel_reduce<2>(v, C, CG, PPB, red, slab + (size_t)blockIdx.x * 2 * C);
  }
}

// depthwise backward: for every INPUT pixel i gather the output gradients that touched it.
//   g_in[i][c]  = sum_k dy[src(i,k)][c] * w[c][k]        (GEOM = adjoint geometry of the forward)
//   dW[c][k]   += dy[src(i,k)][c] * a[i][c]
template <int GEOM>
__global__ __launch_bounds__(256) void dw_bwd_kernel(LazyGrad G, LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                     float* __restrict__ g_in, float* __restrict__ dw_slab,
                                                     BwdEpilogue E, ConvDims d) {
  // d.Hi/Wi: the layer's INPUT size (destination of this gather); d.Ho/Wo: its output size (source of dy)
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  const bool active = pl < PPB;
  float4 dwk[11];   // 9 taps + the two BatchNorm-backward sums of the producer (slots 9, 10)
#pragma unroll
  for (int k = 0; k < 11; ++k) dwk[k] = f4(0.f);
  if (active) {
    const int c = 4 * cg;
    const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
    const LazyGrad4 g = lazygrad4_init(G, C, c);
    const Epi4 epi = epi4_init(E, C, c);
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = make_float4(w[(c + 0) * 9 + k], w[(c + 1) * 9 + k], w[(c + 2) * 9 + k], w[(c + 3) * 9 + k]);
    int p0, p1;
    pixel_chunk(d.B * d.Hi * d.Wi, PPB, p0, p1);
    for (int pix = p0 + pl; pix < p1; pix += PPB) {
      const int ix = pix % d.Wi;
      const int r = pix / d.Wi;
      const int iy = r % d.Hi, bi = r / d.Hi;
      const size_t ioff = (size_t)pix * C + c;
      const float4 yraw = ld4(a.y + ioff);
      float4 av = fma4(yraw, a.scale, a.shift);
      if (a.act == ACT_RELU6) av = relu6_4(av);
      if (b.present) av = av + lazy4_at(b, ioff);
      float4 acc = f4(0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = tap_src<GEOM>(iy, ky, d.sy, d.Ho);
        if (oy < 0) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = tap_src<GEOM>(ix, kx, d.sx, d.Wo);
          if (ox < 0) continue;
          const size_t off = (((size_t)bi * d.Ho + oy) * d.Wo + ox) * C + c;
          const float4 dy = lazygrad4_at(g, off);
          acc = fma4(dy, wk[ky * 3 + kx], acc);
          dwk[ky * 3 + kx] = fma4(dy, av, dwk[ky * 3 + kx]);
        }
      }
      acc = epi4_apply(epi, acc, yraw, ioff, dwk[9], dwk[10]);
      if (g_in != nullptr) st4(g_in + ioff, acc);
    }
  }
  // slab row layout: [k][C] for the 9 taps (the reduce kernel transposes to (C,1,3,3)), then [sum dz | sum dz*xhat]
  block_channel_reduce<11>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// ------------------------------------------------------------------------------------------------
// Backward of a stride-2 transposed layer on one-row tensors (B rows of Wi pixels -> Wo = 2 Wi - 1; the 1 x 3 kernel is
// the middle row of the 3 x 3 one).  The column walkers have a single row to walk there and the per-pixel kernel reads
// six gradient vectors per input pixel; here a thread owns 4 channels of NPX consecutive input pixels: it loads their
// NPX + 2 inputs and 2 NPX + 1 output gradients once, rebuilds the raw layer output the lazy gradient needs from the
// inputs (even column 2x: a[x] w1; odd column 2x + 1: a[x + 1] w0, then a[x] w2 -- the forward's order), and applies the
// producer's epilogue.  The workgroup has PPB * C/4 threads (every lane has a channel quad); slab rows as dw_bwd_kernel.
template <int NPX>
__global__ __launch_bounds__(512) void dw_row_t2_bwd_kernel(LazyGrad G, LazySrc A, const float* __restrict__ w,
                                                            float* __restrict__ g_in, float* __restrict__ dw_slab,
                                                            BwdEpilogue E, int B, int Wi, int C, int PPB) {
  extern __shared__ float red[];
  const int CG = C >> 2;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  const Lazy4 a = lazy4_init(A, C, c);
  const LazyGrad4 g = lazygrad4_init(G, C, c);
  const Epi4 epi = epi4_init(E, C, c);
  float4 wk[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) wk[k] = make_float4(w[(c + 0) * 9 + 3 + k], w[(c + 1) * 9 + 3 + k], w[(c + 2) * 9 + 3 + k], w[(c + 3) * 9 + 3 + k]);
  float4 dwk[3] = {f4(0.f), f4(0.f), f4(0.f)}, s1 = f4(0.f), s2 = f4(0.f);
  const int Wo = 2 * Wi - 1, GX = (Wi + NPX - 1) / NPX;
  int g0, g1;
  pixel_chunk(B * GX, PPB, g0, g1);
  for (int grp = g0 + pl; grp < g1; grp += PPB) {
    const int b = grp / GX, x0 = (grp - b * GX) * NPX;
    // inputs x0 - 1 .. x0 + NPX (raw and activated; zero where the pixel does not exist)
    float4 araw[NPX + 2], av[NPX + 2];
#pragma unroll
    for (int u = 0; u < NPX + 2; ++u) {
      const int ix = x0 - 1 + u;
      const bool ok = ix >= 0 && ix < Wi;
      araw[u] = ld4(a.y + ((size_t)b * Wi + (ok ? ix : 0)) * C + c);
      const float4 v = lazy4_act(a, araw[u]);
      av[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    // output gradients at columns 2 x0 - 1 .. 2 x0 + 2 NPX - 1
    float4 dy[2 * NPX + 1];
#pragma unroll
    for (int j = 0; j < 2 * NPX + 1; ++j) {
      const int ox = 2 * x0 - 1 + j;
      const bool ok = ox >= 0 && ox < Wo;
      const float4 dz = ld4(g.dz + ((size_t)b * Wo + (ok ? ox : 0)) * C + c);
      // j odd: even column 2 (x0 + (j - 1) / 2); j even: odd column 2 x + 1 with x = x0 + j / 2 - 1
      const float4 y = (j & 1) ? av[(j - 1) / 2 + 1] * wk[1] : fma4(av[j / 2], wk[2], av[j / 2 + 1] * wk[0]);
      const float4 v = lazygrad4_eval(g, dz, y);
      dy[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int ix = x0 + u;
      if (ix < Wi) {
        float4 acc = f4(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc = fma4(dy[2 * u + k], wk[k], acc);
          dwk[k] = fma4(dy[2 * u + k], av[u + 1], dwk[k]);
        }
        const size_t ioff = ((size_t)b * Wi + ix) * C + c;
        acc = epi4_apply(epi, acc, araw[u + 1], ioff, s1, s2);
        if (g_in != nullptr) st4(g_in + ioff, acc);
      }
    }
  }
  const float4 v[11] = {f4(0.f), f4(0.f), f4(0.f), dwk[0], dwk[1], dwk[2], f4(0.f), f4(0.f), f4(0.f), s1, s2};
  block_channel_reduce<11>(v, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// The same for a plain stride-2 layer on one-row tensors (Wi -> Wo = (Wi - 1) / 2 + 1): a thread owns 4 channels of NOX
// output pixels = 2 NOX input pixels.  Input 2m receives dy[m] w1, input 2m + 1 receives dy[m + 1] w0, then dy[m] w2; the raw
// layer output of the lazy gradient is rebuilt as a[2o - 1] w0, a[2o] w1, a[2o + 1] w2 in that order (the forward's).
template <int NOX>
__global__ __launch_bounds__(512) void dw_row_c2_bwd_kernel(LazyGrad G, LazySrc A, const float* __restrict__ w,
                                                            float* __restrict__ g_in, float* __restrict__ dw_slab,
                                                            BwdEpilogue E, int B, int Wi, int C, int PPB) {
  extern __shared__ float red[];
  const int CG = C >> 2;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  const Lazy4 a = lazy4_init(A, C, c);
  const LazyGrad4 g = lazygrad4_init(G, C, c);
  const Epi4 epi = epi4_init(E, C, c);
  float4 wk[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) wk[k] = make_float4(w[(c + 0) * 9 + 3 + k], w[(c + 1) * 9 + 3 + k], w[(c + 2) * 9 + 3 + k], w[(c + 3) * 9 + 3 + k]);
  float4 dwk[3] = {f4(0.f), f4(0.f), f4(0.f)}, s1 = f4(0.f), s2 = f4(0.f);
  const int Wo = (Wi - 1) / 2 + 1, GX = (Wo + NOX - 1) / NOX;
  int g0, g1;
  pixel_chunk(B * GX, PPB, g0, g1);
  for (int grp = g0 + pl; grp < g1; grp += PPB) {
    const int b = grp / GX, o0 = (grp - b * GX) * NOX;
    // inputs 2 o0 - 1 .. 2 o0 + 2 NOX + 1 (raw and activated; zero where the pixel does not exist)
    float4 araw[2 * NOX + 3], av[2 * NOX + 3];
#pragma unroll
    for (int u = 0; u < 2 * NOX + 3; ++u) {
      const int ix = 2 * o0 - 1 + u;
      const bool ok = ix >= 0 && ix < Wi;
      araw[u] = ld4(a.y + ((size_t)b * Wi + (ok ? ix : 0)) * C + c);
      const float4 v = lazy4_act(a, araw[u]);
      av[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    // output gradients o0 .. o0 + NOX (the last one only feeds the odd input before it)
    float4 dy[NOX + 1];
#pragma unroll
    for (int j = 0; j < NOX + 1; ++j) {
      const int ox = o0 + j;
      const bool ok = ox < Wo;
      const float4 dz = ld4(g.dz + ((size_t)b * Wo + (ok ? ox : 0)) * C + c);
      float4 y = av[2 * j] * wk[0];
      y = fma4(av[2 * j + 1], wk[1], y);
      y = fma4(av[2 * j + 2], wk[2], y);
      const float4 v = lazygrad4_eval(g, dz, y);
      dy[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
#pragma unroll
    for (int j = 0; j < NOX; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) dwk[k] = fma4(dy[j], av[2 * j + k], dwk[k]);
#pragma unroll
    for (int u = 0; u < 2 * NOX; ++u) {
      const int ix = 2 * o0 + u;
      if (ix < Wi) {
        const float4 acc = (u & 1) ? fma4(dy[u / 2], wk[2], dy[u / 2 + 1] * wk[0]) : dy[u / 2] * wk[1];
        const size_t ioff = ((size_t)b * Wi + ix) * C + c;
        const float4 o = epi4_apply(epi, acc, araw[u + 1], ioff, s1, s2);
        if (g_in != nullptr) st4(g_in + ioff, o);
      }
    }
  }
  const float4 v[11] = {f4(0.f), f4(0.f), f4(0.f), dwk[0], dwk[1], dwk[2], f4(0.f), f4(0.f), f4(0.f), s1, s2};
  block_channel_reduce<11>(v, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// Stride 1 on one-row tensors, plain (TR = false: out[o] = sum_k a[o - 1 + k] w[k]) or transposed (out[o] = sum_k
// a[o + 1 - k] w[k]): a thread owns 4 channels of NPX consecutive pixels.  The column walkers serve these layers as one
// image of B rows, but their 256-thread blocks leave lanes without a channel quad at 576 / 768 channels, and they read
// the raw output this kernel rebuilds (ascending source index, the forward's order).
template <bool TR, int NPX>
__global__ __launch_bounds__(512) void dw_row_s1_bwd_kernel(LazyGrad G, LazySrc A, const float* __restrict__ w,
                                                            float* __restrict__ g_in, float* __restrict__ dw_slab,
                                                            BwdEpilogue E, int B, int Wi, int C, int PPB) {
  extern __shared__ float red[];
  const int CG = C >> 2;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  const Lazy4 a = lazy4_init(A, C, c);
  const LazyGrad4 g = lazygrad4_init(G, C, c);
  const Epi4 epi = epi4_init(E, C, c);
  float4 wk[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) wk[k] = make_float4(w[(c + 0) * 9 + 3 + k], w[(c + 1) * 9 + 3 + k], w[(c + 2) * 9 + 3 + k], w[(c + 3) * 9 + 3 + k]);
  const float4 wlo = TR ? wk[2] : wk[0], whi = TR ? wk[0] : wk[2];   // weights of the sources o - 1 and o + 1
  float4 dwk[3] = {f4(0.f), f4(0.f), f4(0.f)}, s1 = f4(0.f), s2 = f4(0.f);
  const int GX = (Wi + NPX - 1) / NPX;
  int g0, g1;
  pixel_chunk(B * GX, PPB, g0, g1);
  for (int grp = g0 + pl; grp < g1; grp += PPB) {
    const int b = grp / GX, x0 = (grp - b * GX) * NPX;
    float4 araw[NPX + 4], av[NPX + 4];   // pixels x0 - 2 .. x0 + NPX + 1
#pragma unroll
    for (int u = 0; u < NPX + 4; ++u) {
      const int ix = x0 - 2 + u;
      const bool ok = ix >= 0 && ix < Wi;
      araw[u] = ld4(a.y + ((size_t)b * Wi + (ok ? ix : 0)) * C + c);
      const float4 v = lazy4_act(a, araw[u]);
      av[u] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    float4 dy[NPX + 2];                   // output pixels x0 - 1 .. x0 + NPX
#pragma unroll
    for (int j = 0; j < NPX + 2; ++j) {
      const int ox = x0 - 1 + j;
      const bool ok = ox >= 0 && ox < Wi;
      const float4 dz = ld4(g.dz + ((size_t)b * Wi + (ok ? ox : 0)) * C + c);
      float4 y = av[j] * wlo;
      y = fma4(av[j + 1], wk[1], y);
      y = fma4(av[j + 2], whi, y);
      const float4 v = lazygrad4_eval(g, dz, y);
      dy[j] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
#pragma unroll
    for (int u = 0; u < NPX; ++u) {
      const int ix = x0 + u;
      if (ix < Wi) {
        float4 acc = f4(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          acc = fma4(TR ? dy[u + k] : dy[u + 2 - k], wk[k], acc);
          dwk[k] = TR ? fma4(dy[u + k], av[u + 2], dwk[k]) : fma4(dy[u + 1], av[u + 1 + k], dwk[k]);
        }
        const size_t ioff = ((size_t)b * Wi + ix) * C + c;
        acc = epi4_apply(epi, acc, araw[u + 2], ioff, s1, s2);
        if (g_in != nullptr) st4(g_in + ioff, acc);
      }
    }
  }
  const float4 v[11] = {f4(0.f), f4(0.f), f4(0.f), dwk[0], dwk[1], dwk[2], f4(0.f), f4(0.f), f4(0.f), s1, s2};
  block_channel_reduce<11>(v, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// ------------------------------------------------------------------------------------------------
// first layer: single-channel input, normalised by its per-clip max and padded on the fly.
//   x_in(b, py, px) = pad_value if py < hp or px < wp else logmag[b][py-hp][px-wp] / (absmax[b] + 1e-6)
struct FirstDims {
  int B, H, W, hp, wp, Ho, Wo, C, sy, sx;
  float pad_value;
};

__device__ __forceinline__ float first_input(const float* __restrict__ x, float mx, const FirstDims& d, int bi,
                                             int py, int px) {
  if (py < 0 || px < 0 || py >= d.H + d.hp || px >= d.W + d.wp) return 0.f;  // the conv's own zero padding
  if (py < d.hp || px < d.wp) return d.pad_value;
  return x[((size_t)bi * d.H + (py - d.hp)) * d.W + (px - d.wp)] / mx;
}

__global__ __launch_bounds__(256) void first_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ absmax,
                                                             const float* __restrict__ w, float* __restrict__ out,
                                                             float* __restrict__ slab, FirstDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  const bool active = pl < PPB;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (active) {
    const int c = 4 * cg;
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = make_float4(w[(c + 0) * 9 + k], w[(c + 1) * 9 + k], w[(c + 2) * 9 + k], w[(c + 3) * 9 + k]);
    int p0, p1;
    pixel_chunk(d.B * d.Ho * d.Wo, PPB, p0, p1);
    for (int pix = p0 + pl; pix < p1; pix += PPB) {
      const int ox = pix % d.Wo;
      const int r = pix / d.Wo;
      const int oy = r % d.Ho, bi = r / d.Ho;
      const float mx = absmax[bi] + 1e-6f;
      float4 acc = f4(0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = first_input(x, mx, d, bi, oy * d.sy - 1 + ky, ox * d.sx - 1 + kx);
          acc = fma4(f4(v), wk[ky * 3 + kx], acc);
        }
      st4(out + (size_t)pix * C + c, acc);
      s1 = s1 + acc;
      s2 = fma4(acc, acc, s2);
    }
  }
  if (slab != nullptr) {
    const float4 v[2] = {s1, s2};
    block_channel_reduce<2>(v, C, CG, PPB, red, slab + (size_t)blockIdx.x * 2 * C);
  }
}

// dW[c][k] = sum_o dy[o][c] * x_in(o*s - 1 + k)   (the input is data: no input gradient)
__global__ __launch_bounds__(256) void first_conv_wgrad_kernel(LazyGrad G, const float* __restrict__ x,
                                                               const float* __restrict__ absmax,
                                                               float* __restrict__ dw_slab, FirstDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  float4 dwk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) dwk[k] = f4(0.f);
  if (pl < PPB) {
    const int c = 4 * cg;
    const LazyGrad4 g = lazygrad4_init(G, C, c);
    int p0, p1;
    pixel_chunk(d.B * d.Ho * d.Wo, PPB, p0, p1);
    for (int pix = p0 + pl; pix < p1; pix += PPB) {
      const int ox = pix % d.Wo;
      const int r = pix / d.Wo;
      const int oy = r % d.Ho, bi = r / d.Ho;
      const float mx = absmax[bi] + 1e-6f;
      const float4 dy = lazygrad4_at(g, (size_t)pix * C + c);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = first_input(x, mx, d, bi, oy * d.sy - 1 + ky, ox * d.sx - 1 + kx);
          dwk[ky * 3 + kx] = fma4(dy, f4(v), dwk[ky * 3 + kx]);
        }
    }
  }
  block_channel_reduce<9>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 9 * C);
}

// ------------------------------------------------------------------------------------------------
// last layer: dense transposed conv C->1 with bias, cropped to the last H x W outputs, fused with
// the additive mask:  enh = clamp(x/max + filt, -1, 1) * max        (train/models.py:762-763)
struct LastDims {
  int B, Hi, Wi, C, sy, sx;  // input (B,Hi,Wi,C)
  int H, W, oy0, ox0;        // output crop: H x W starting at (oy0, ox0) of the full transposed-conv output
};

__global__ __launch_bounds__(256) void last_convT_fwd_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ logmag,
                                                             const float* __restrict__ absmax, float* __restrict__ enh,
                                                             float* __restrict__ filt_out, LastDims d) {
  extern __shared__ float wl[];  // [9][C] + scaleA/shiftA/scaleB/shiftB [4][C]
  const int C = d.C;
  for (int i = threadIdx.x; i < 9 * C; i += blockDim.x) wl[i] = w[(i % C) * 9 + i / C];
  float* cf = wl + 9 * C;
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    cf[i] = A.coef ? A.coef[i] : 1.f;
    cf[C + i] = A.coef ? A.coef[C + i] : 0.f;
    cf[2 * C + i] = (Bsrc.y && Bsrc.coef) ? Bsrc.coef[i] : 1.f;
    cf[3 * C + i] = (Bsrc.y && Bsrc.coef) ? Bsrc.coef[C + i] : 0.f;
  }
  __syncthreads();
  int p0, p1;
  pixel_chunk(d.B * d.H * d.W, blockDim.x, p0, p1);
  for (int pix = p0 + threadIdx.x; pix < p1; pix += blockDim.x) {
    const int ox = pix % d.W;
    const int r = pix / d.W;
    const int oy = r % d.H, bi = r / d.H;
    float acc = bias[0];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = tap_src<GEOM_CONVT>(oy + d.oy0, ky, d.sy, d.Hi);
      if (iy < 0) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = tap_src<GEOM_CONVT>(ox + d.ox0, kx, d.sx, d.Wi);
        if (ix < 0) continue;
        const size_t off = (((size_t)bi * d.Hi + iy) * d.Wi + ix) * C;
        const float* wk = wl + (ky * 3 + kx) * C;
        for (int c = 0; c < C; c += 4) {
          float4 v = fma4(ld4(A.y + off + c), ld4(cf + c), ld4(cf + C + c));
          if (A.relu6) v = relu6_4(v);
          if (Bsrc.y) {
            float4 u = fma4(ld4(Bsrc.y + off + c), ld4(cf + 2 * C + c), ld4(cf + 3 * C + c));
            if (Bsrc.relu6) u = relu6_4(u);
            v = v + u;
          }
          const float4 ww = ld4(wk + c);
          acc = fmaf(v.x, ww.x, acc);
          acc = fmaf(v.y, ww.y, acc);
          acc = fmaf(v.z, ww.z, acc);
          acc = fmaf(v.w, ww.w, acc);
        }
      }
    }
    if (filt_out) filt_out[pix] = acc;
    if (enh) {
      const float mx = absmax[bi] + 1e-6f;
      const float v = logmag[pix] / mx + acc;
      enh[pix] = fminf(fmaxf(v, -1.f), 1.f) * mx;
    }
  }
}

// Same computation with the channel sum spread over the C/4 lanes of a pixel (C/4 a power of two <= 16):
// every load is a coalesced float4 and the per-pixel dot product finishes with a few lane shuffles.
__global__ __launch_bounds__(256) void last_convT_fwd_split_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ logmag,
                                                                   const float* __restrict__ absmax,
                                                                   float* __restrict__ enh, float* __restrict__ filt_out,
                                                                   LastDims d) {
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
  float4 wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = make_float4(w[(c + 0) * 9 + k], w[(c + 1) * 9 + k], w[(c + 2) * 9 + k], w[(c + 3) * 9 + k]);
  const float bv = bias[0];
  int p0, p1;
  pixel_chunk(d.B * d.H * d.W, PPB, p0, p1);
  for (int base = p0; base < p1; base += PPB) {   // uniform trip count: the shuffles need whole pixel groups
    const int pix = base + pl;
    const bool valid = pix < p1;
    float acc = 0.f;
    int bi = 0;
    if (valid) {
      const int ox = pix % d.W;
      const int r = pix / d.W;
      const int oy = r % d.H;
      bi = r / d.H;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = tap_src<GEOM_CONVT>(oy + d.oy0, ky, d.sy, d.Hi);
        if (iy < 0) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ix = tap_src<GEOM_CONVT>(ox + d.ox0, kx, d.sx, d.Wi);
          if (ix < 0) continue;
          const size_t off = (((size_t)bi * d.Hi + iy) * d.Wi + ix) * C + c;
          float4 v = lazy4_at(a, off);
          if (b.present) v = v + lazy4_at(b, off);
          const float4 ww = wk[ky * 3 + kx];
          acc = fmaf(v.x, ww.x, acc);
          acc = fmaf(v.y, ww.y, acc);
          acc = fmaf(v.z, ww.z, acc);
          acc = fmaf(v.w, ww.w, acc);
        }
      }
    }
    for (int o = CG >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (valid && cg == 0) {
      acc += bv;
      if (filt_out) filt_out[pix] = acc;
      if (enh) {
        const float mx = absmax[bi] + 1e-6f;
        const float v = logmag[pix] / mx + acc;
        enh[pix] = fminf(fmaxf(v, -1.f), 1.f) * mx;
      }
    }
  }
}

// Backward of the last layer.  dy0[o] = d_enh[o] * max * [ |enh[o]| < max ]  on the cropped region.
//   g_in[i][c] = sum_k dy0[i*s - 1 + k] * w[c][k];   dW[c][k] += a[i][c] * dy0[i*s - 1 + k];   db = sum dy0
__global__ __launch_bounds__(256) void last_convT_bwd_kernel(const float* __restrict__ d_enh, const float* __restrict__ enh,
                                                             const float* __restrict__ absmax, LazySrc A, LazySrc Bsrc,
                                                             const float* __restrict__ w, float* __restrict__ g_in,
                                                             float* __restrict__ dw_slab, BwdEpilogue E, LastDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG;
  float4 dwk[12];  // 9 taps, bias gradient in .x of slot 9 (channel group 0 only), producer BN sums in 10, 11
#pragma unroll
  for (int k = 0; k < 12; ++k) dwk[k] = f4(0.f);
  if (pl < PPB) {
    const int c = 4 * cg;
    const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
    const Epi4 epi = epi4_init(E, C, c);
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = make_float4(w[(c + 0) * 9 + k], w[(c + 1) * 9 + k], w[(c + 2) * 9 + k], w[(c + 3) * 9 + k]);
    int p0, p1;
    pixel_chunk(d.B * d.Hi * d.Wi, PPB, p0, p1);
    for (int pix = p0 + pl; pix < p1; pix += PPB) {
      const int ix = pix % d.Wi;
      const int r = pix / d.Wi;
      const int iy = r % d.Hi, bi = r / d.Hi;
      const float mx = absmax[bi] + 1e-6f;
      const size_t ioff = (size_t)pix * C + c;
      const float4 yraw = ld4(a.y + ioff);
      float4 av = fma4(yraw, a.scale, a.shift);
      if (a.act == ACT_RELU6) av = relu6_4(av);
      if (b.present) av = av + lazy4_at(b, ioff);
      float4 acc = f4(0.f);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int oy = iy * d.sy - 1 + ky - d.oy0;
        if (oy < 0 || oy >= d.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int ox = ix * d.sx - 1 + kx - d.ox0;
          if (ox < 0 || ox >= d.W) continue;
          const size_t o = ((size_t)bi * d.H + oy) * d.W + ox;
          const float e = enh[o];
          const float dy = (e > -mx && e < mx) ? d_enh[o] * mx : 0.f;
          acc = fma4(f4(dy), wk[ky * 3 + kx], acc);
          dwk[ky * 3 + kx] = fma4(f4(dy), av, dwk[ky * 3 + kx]);
        }
      }
      acc = epi4_apply(epi, acc, yraw, ioff, dwk[10], dwk[11]);
      st4(g_in + ioff, acc);
    }
    // bias gradient: every output position is owned by exactly one (pixel lane, cg==0) visit below
    if (cg == 0) {
      const int nout = d.B * d.H * d.W;
      float s = 0.f;
      for (int o = blockIdx.x * PPB + pl; o < nout; o += gridDim.x * PPB) {
        const int bi = o / (d.H * d.W);
        const float mx = absmax[bi] + 1e-6f;
        const float e = enh[o];
        s += (e > -mx && e < mx) ? d_enh[o] * mx : 0.f;
      }
      dwk[9].x = s;
    }
  }
  block_channel_reduce<12>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 12 * C);
}

// out[j] = sum_b slab[b][j]; optional transpose from [k][C] slab rows to the (C,k) parameter layout
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nrows, int ncols,
                                                          int row_stride, float* __restrict__ out, int tr_k, int tr_C,
                                                          int accumulate) {
  __shared__ double red[256];
  const int j = blockIdx.x * SLAB_COLS + (threadIdx.x & (SLAB_COLS - 1));
  const bool ok = j < ncols;
  const double s = slab_column_sum(slab, nrows, row_stride, j, ok, red);
  if (threadIdx.x >= SLAB_COLS || !ok) return;
  int o = j;
  if (tr_k > 0) {  // j = k*C + c  ->  c*tr_k + k
    const int k = j / tr_C, c = j - k * tr_C;
    if (k >= tr_k) return;
    o = c * tr_k + k;
  }
  out[o] = accumulate ? out[o] + (float)s : (float)s;
}

// The same fold for a table of slabs in one launch: blockIdx.y = job.  A workgroup takes FOLD_COLS = 16 columns (64
// contiguous bytes of every row) with 16 row lanes, four independent double accumulators per thread.
constexpr int FOLD_COLS = 16;
__global__ __launch_bounds__(256) void fold_jobs_kernel(const noise_fold_job* __restrict__ jobs) {
  __shared__ double red[256];
  const noise_fold_job jb = jobs[blockIdx.y];
  if ((int)blockIdx.x * FOLD_COLS >= jb.ncols) return;
  constexpr int LANES = 256 / FOLD_COLS;
  const int cl = threadIdx.x & (FOLD_COLS - 1), rl = threadIdx.x / FOLD_COLS;
  const int j = blockIdx.x * FOLD_COLS + cl;
  const bool ok = j < jb.ncols;
  const float* col = jb.slab + (ok ? j : 0);
  // a lane requests 16 rows before it adds the first one: the kernel costs dependent round trips, not bytes
  constexpr int FB = 16;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int r0 = rl; r0 < jb.rows; r0 += FB * LANES) {
    float v[FB];
#pragma unroll
    for (int u = 0; u < FB; ++u) {
      const int r = r0 + u * LANES;
      v[u] = col[(size_t)(r < jb.rows ? r : jb.rows - 1) * jb.row_stride];
    }
#pragma unroll
    for (int u = 0; u < FB; u += 4) {
      a0 += r0 + u * LANES < jb.rows ? (double)v[u] : 0.0;
      a1 += r0 + (u + 1) * LANES < jb.rows ? (double)v[u + 1] : 0.0;
      a2 += r0 + (u + 2) * LANES < jb.rows ? (double)v[u + 2] : 0.0;
      a3 += r0 + (u + 3) * LANES < jb.rows ? (double)v[u + 3] : 0.0;
    }
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x >= FOLD_COLS || !ok) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < LANES; ++k) s += red[threadIdx.x + FOLD_COLS * k];
  int o = j;
  if (jb.tr_k > 0) {
    const int k = j / jb.tr_C, c = j - k * jb.tr_C;
    if (k >= jb.tr_k) return;
    o = c * jb.tr_k + k;
  }
  jb.out[o] = (float)s;
}

int conv_grid(long npix, int C) {
  const int PPB = 256 / (C / 4);
  return noise_grid(npix, PPB * 4, 1024);
}

int check_size(const char* who, long npix, int C) {
  NOISE_REQUIRE(npix < (1L << 31) - 65536 * 64 && C > 0, "%s: %ld pixels exceed the 32-bit pixel index of this kernel", who, npix);
  return 0;
}

int check_c(const char* who, int C) {
  NOISE_REQUIRE(C >= 4 && C % 4 == 0 && C <= 1024, "%s: channel count %d must be a multiple of 4 in [4,1024]", who, C);
  return 0;
}

}  // namespace

NOISE_EXPORT int noise_conv_max_blocks(void) {
  const int cap = walk_env("NOISE_WALK_CAP", 2048);   // diagnostic override (scripts/walk_sweep.py)
  return cap > 2048 ? cap : 2048;
}
// The backward walkers hold one workgroup per CU (register-resident windows, coefficients and weight-gradient taps), so
// more than 4 workgroups per CU only lengthens the slabs the finalize kernels fold.
constexpr int BWD_MAX_BLOCKS = 1024;

// transposed = 0: Conv2d, 1: ConvTranspose2d (both k3, p1, groups=C, no bias)
NOISE_EXPORT int noise_dw3x3_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y,
                                 const float* b_coef, int b_relu6, const float* w, float* out, float* stats_slab,
                                 int* slab_rows, int transposed, int B, int Hi, int Wi, int C, int sy, int sx,
                                 void* stream) {
  if (check_c("noise_dw3x3_fwd", C)) return -1;
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_dw3x3_fwd: empty input");
  NOISE_REQUIRE((sy == 1 || sy == 2) && (sx == 1 || sx == 2), "noise_dw3x3_fwd: stride must be 1 or 2");
  // out == NULL: statistics only (stride-2 transposed layers whose output the following 1x1 layer rebuilds)
  NOISE_REQUIRE(a_y && w, "noise_dw3x3_fwd: null pointer");
  NOISE_REQUIRE(out || (transposed && sy == 2 && sx == 2 && stats_slab),
                "noise_dw3x3_fwd: out may only be NULL for a stride-2 transposed layer with a statistics slab");
  ConvDims d;
  d.B = B; d.Hi = Hi; d.Wi = Wi; d.C = C; d.sy = sy; d.sx = sx;
  d.Ho = transposed ? (Hi - 1) * sy + 1 : (Hi - 1) / sy + 1;
  d.Wo = transposed ? (Wi - 1) * sx + 1 : (Wi - 1) / sx + 1;
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  if (check_size("noise_dw3x3_fwd", (long)B * d.Ho * d.Wo + (long)B * Hi * Wi, C)) return -1;
  const int PPB = 256 / (C / 4);
  const size_t smem = (size_t)PPB * 2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (sy == sx) {
    WalkDims wd{B, Hi, Wi, d.Ho, d.Wo, C, 0, 0, 0, 0};
    int grid;
    const bool cell = transposed && sy == 2;
    wd.wown = cell ? Wi : d.Wo;
    walk_plan(wd, cell ? Hi : d.Ho, noise_conv_max_blocks(), grid);
    const bool hb = b_y != nullptr;
#define LAUNCH_FWD(...)                                                                                            \
  do {                                                                                                             \
    if (hb)                                                                                                        \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ true>), grid, smem, st, A, Bs, w, out, stats_slab, wd);          \
    else                                                                                                           \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ false>), grid, smem, st, A, Bs, w, out, stats_slab, wd);         \
  } while (0)
    if (cell && out == nullptr) {
      if (hb)
        NOISE_LAUNCH_ROUNDS((walk_cell_fwd_kernel<true, false>), grid, smem, st, A, Bs, w, out, stats_slab, wd);
      else
        NOISE_LAUNCH_ROUNDS((walk_cell_fwd_kernel<false, false>), grid, smem, st, A, Bs, w, out, stats_slab, wd);
    } else if (cell)
      LAUNCH_FWD(walk_cell_fwd_kernel<);
    else if (transposed)
      LAUNCH_FWD(walk_gather_fwd_kernel<1, true,);
    else if (sy == 1)
      LAUNCH_FWD(walk_gather_fwd_kernel<1, false,);
    else
      LAUNCH_FWD(walk_gather_fwd_kernel<2, false,);
#undef LAUNCH_FWD
    if (slab_rows) *slab_rows = grid;
    NOISE_CHECK_LAUNCH("noise_dw3x3_fwd");
    return 0;
  }
  const int grid = conv_grid((long)B * d.Ho * d.Wo, C);
  if (slab_rows) *slab_rows = grid;
  if (transposed)
    hipLaunchKernelGGL(dw_fwd_kernel<GEOM_CONVT>, dim3(grid), dim3(256), smem, (hipStream_t)stream, A, Bs, w, out, stats_slab, d);
  else
    hipLaunchKernelGGL(dw_fwd_kernel<GEOM_CONV>, dim3(grid), dim3(256), smem, (hipStream_t)stream, A, Bs, w, out, stats_slab, d);
  NOISE_CHECK_LAUNCH("noise_dw3x3_fwd");
  return 0;
}

// Backward of noise_dw3x3_fwd.  (Hi,Wi) is the forward INPUT size.  Writes g_in (B,Hi,Wi,C) and dw (C,3,3).
// Epilogue (all optional): g_in += g_other; with epi_coef (the BatchNorm coefficients of source a's producer)
// g_in is masked by that producer's ReLU6 (epi_relu6) and the producer's BatchNorm-backward sums are left in
// scratch: rows of 11*C floats, columns [9*C, 11*C) = [sum dz | sum dz*xhat]; *slab_rows receives the row count.
// scratch: at least noise_conv_max_blocks()*11*C floats.
NOISE_EXPORT int noise_dw3x3_bwd(const float* dz, const float* y_out, const float* coef3, const float* a_y,
                                 const float* a_coef, int a_relu6, const float* b_y, const float* b_coef, int b_relu6,
                                 const float* w, float* g_in, float* dw, float* scratch, const float* g_other,
                                 const float* epi_coef, int epi_relu6, int* slab_rows, int transposed, int B, int Hi,
                                 int Wi, int C, int sy, int sx, void* stream) {
  if (check_c("noise_dw3x3_bwd", C)) return -1;
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_dw3x3_bwd: empty input");
  NOISE_REQUIRE((sy == 1 || sy == 2) && (sx == 1 || sx == 2), "noise_dw3x3_bwd: stride must be 1 or 2");
  NOISE_REQUIRE(dz && a_y && w && scratch, "noise_dw3x3_bwd: null pointer");
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr, "noise_dw3x3_bwd: coef3 needs y_out");
  ConvDims d;
  d.B = B; d.Hi = Hi; d.Wi = Wi; d.C = C; d.sy = sy; d.sx = sx;
  d.Ho = transposed ? (Hi - 1) * sy + 1 : (Hi - 1) / sy + 1;
  d.Wo = transposed ? (Wi - 1) * sx + 1 : (Wi - 1) / sx + 1;
  LazyGrad G{dz, y_out, coef3};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  if (check_size("noise_dw3x3_bwd", (long)B * d.Ho * d.Wo + (long)B * Hi * Wi, C)) return -1;
  const int PPB = 256 / (C / 4);
  // the walkers keep their table of per-channel constants (BT_ROWS rows) in the reduction buffer
  const size_t smem = (size_t)(PPB * 11 > BT_ROWS ? PPB * 11 : BT_ROWS) * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  BwdEpilogue E{g_other, epi_coef, epi_relu6, nullptr};
  // One-row tensors (NOISE_DW_NO_ROW_KERNEL, diagnostic: the general kernels): a thread takes several consecutive pixels
  // and the layer's raw output is rebuilt from the input; the workgroup is as many whole pixel lanes as 512 threads and
  // 64 KB of reduction space hold.  Selected by a column stride of 2 with a different row stride, or by stride 1.
  static const bool row_off = getenv("NOISE_DW_NO_ROW_KERNEL") != nullptr;
  if (Hi == 1 && b_y == nullptr && !row_off && ((sx == 2 && sy != sx) || (sx == 1 && sy == 1))) {
    const int CG = C / 4;
    int ppb = 512 / CG;
    const int lds_cap = (int)(65536 / ((size_t)11 * C * sizeof(float)));
    if (ppb > lds_cap) ppb = lds_cap;
    if (ppb >= 1) {
      constexpr int NPX = 4, NOX = 2;   // pixels per thread: input pixels (transposed, stride 1), output pixels (plain stride 2)
      const long groups = (sx == 2 && !transposed) ? (long)B * ((d.Wo + NOX - 1) / NOX) : (long)B * ((Wi + NPX - 1) / NPX);
      int grid = noise_grid(groups, ppb * 2, 1024);
      const size_t sm = (size_t)ppb * 11 * C * sizeof(float);
#define ROW_LAUNCH(...)                                                                                                 \
  do {                                                                                                                  \
    NOISE_ALLOW_LDS((__VA_ARGS__), sm);                                                                                 \
    hipLaunchKernelGGL((__VA_ARGS__), dim3(grid), dim3(ppb * CG), sm, st, G, A, w, g_in, scratch, E, B, Wi, C, ppb);    \
  } while (0)
      if (sx == 2 && transposed) ROW_LAUNCH(dw_row_t2_bwd_kernel<NPX>);
      else if (sx == 2) ROW_LAUNCH(dw_row_c2_bwd_kernel<NOX>);
      else if (transposed) ROW_LAUNCH(dw_row_s1_bwd_kernel<true, NPX>);
      else ROW_LAUNCH(dw_row_s1_bwd_kernel<false, NPX>);
#undef ROW_LAUNCH
      if (slab_rows) *slab_rows = grid;
      NOISE_CHECK_LAUNCH("noise_dw3x3_bwd/row");
      if (dw != nullptr) {
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(noise_cdiv(9 * C, SLAB_COLS)), dim3(256), 0, st, scratch, grid, 9 * C, 11 * C, dw, 9, C, 0);
        NOISE_CHECK_LAUNCH("noise_dw3x3_bwd/reduce");
      }
      return 0;
    }
  }
  int grid;
  if (sy == sx) {
    // source of the walk = dy (the layer's output size), destination = the layer's input
    WalkDims wd{B, d.Ho, d.Wo, Hi, Wi, C, 0, 0, 0, 0};
    const bool cell = !transposed && sy == 2;
    wd.wown = cell ? d.Wo : Wi;
    walk_plan(wd, cell ? d.Ho : Hi, BWD_MAX_BLOCKS, grid);
    const bool hb = b_y != nullptr, hg = g_other != nullptr;
#define LAUNCH_BWD(...)                                                                                             \
  do {                                                                                                              \
    if (hb && hg)                                                                                                   \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ true, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);   \
    else if (hb)                                                                                                    \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ true, false>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);  \
    else if (hg)                                                                                                    \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ false, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);  \
    else                                                                                                            \
      NOISE_LAUNCH_ROUNDS((__VA_ARGS__ false, false>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd); \
  } while (0)
#define LAUNCH_BWD_RECOMP()                                                                                          \
  do {                                                                                                              \
    if (hb && hg)                                                                                                   \
      NOISE_LAUNCH_ROUNDS((walk_gather_bwd_kernel<2, false, true, true, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);   \
    else if (hb)                                                                                                    \
      NOISE_LAUNCH_ROUNDS((walk_gather_bwd_kernel<2, false, true, false, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);  \
    else if (hg)                                                                                                    \
      NOISE_LAUNCH_ROUNDS((walk_gather_bwd_kernel<2, false, false, true, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd);  \
    else                                                                                                            \
      NOISE_LAUNCH_ROUNDS((walk_gather_bwd_kernel<2, false, false, false, true>), grid, smem, st, G, A, Bs, w, g_in, scratch, E, wd); \
  } while (0)
    // NOISE_DW_NO_RECOMPUTE (diagnostic): read the transposed layers' raw output back instead of recomputing it
    static const bool recompute_off = getenv("NOISE_DW_NO_RECOMPUTE") != nullptr;
    if (cell)
      LAUNCH_BWD(walk_cell_bwd_kernel<);
    else if (!transposed)
      LAUNCH_BWD(walk_gather_bwd_kernel<1, true,);
    else if (sy == 1)
      LAUNCH_BWD(walk_gather_bwd_kernel<1, false,);
    else if (G.coef3 != nullptr && !recompute_off)
      LAUNCH_BWD_RECOMP();
    else
      LAUNCH_BWD(walk_gather_bwd_kernel<2, false,);
#undef LAUNCH_BWD
#undef LAUNCH_BWD_RECOMP
  } else {
    grid = conv_grid((long)B * Hi * Wi, C);
    // the adjoint of a CONV gather is a CONVT gather and vice versa
    if (transposed)
      hipLaunchKernelGGL(dw_bwd_kernel<GEOM_CONV>, dim3(grid), dim3(256), smem, st, G, A, Bs, w, g_in, scratch, E, d);
    else
      hipLaunchKernelGGL(dw_bwd_kernel<GEOM_CONVT>, dim3(grid), dim3(256), smem, st, G, A, Bs, w, g_in, scratch, E, d);
  }
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_dw3x3_bwd");
  if (dw != nullptr) {   // NULL: the caller folds the slab later (noise_fold_jobs)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(noise_cdiv(9 * C, SLAB_COLS)), dim3(256), 0, st, scratch, grid, 9 * C, 11 * C, dw, 9, C, 0);
    NOISE_CHECK_LAUNCH("noise_dw3x3_bwd/reduce");
  }
  return 0;
}

// Folds njobs per-workgroup slabs into their destinations in one launch (jobs: device array).  The weight-gradient
// slabs of noise_pw_bwd / noise_dw3x3_bwd called with dw = NULL are meant for this: nothing waits for a weight gradient
// before the optimizer, so one fold at the end of the backward pass replaces one small launch per layer.
NOISE_EXPORT int noise_fold_jobs(const noise_fold_job* jobs, int njobs, int max_cols, void* stream) {
  NOISE_REQUIRE(jobs && njobs > 0 && max_cols > 0, "noise_fold_jobs: bad argument");
  hipLaunchKernelGGL(fold_jobs_kernel, dim3(noise_cdiv(max_cols, FOLD_COLS), njobs), dim3(256), 0, (hipStream_t)stream, jobs);
  NOISE_CHECK_LAUNCH("noise_fold_jobs");
  return 0;
}

NOISE_EXPORT int noise_first_conv_fwd(const float* logmag, const float* absmax, const float* w, float* out,
                                      float* stats_slab, int* slab_rows, int B, int H, int W, int hp, int wp,
                                      float pad_value, int C, int sy, int sx, void* stream) {
  if (check_c("noise_first_conv_fwd", C)) return -1;
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && hp >= 0 && wp >= 0, "noise_first_conv_fwd: bad shape");
  NOISE_REQUIRE(logmag && absmax && w && out, "noise_first_conv_fwd: null pointer");
  FirstDims d{B, H, W, hp, wp, (H + hp - 1) / sy + 1, (W + wp - 1) / sx + 1, C, sy, sx, pad_value};
  if (check_size("noise_first_conv_fwd", (long)B * (H + hp) * (W + wp), C)) return -1;
  int grid = conv_grid((long)B * d.Ho * d.Wo, C);
  const int PPB = 256 / (C / 4);
  if (sy == 2 && sx == 2) {
    FirstWalkDims wd{B, H, W, hp, wp, d.Ho, d.Wo, C, pad_value, 0, 0, 0};
    first_walk_plan(wd, noise_conv_max_blocks(), grid);
    const int CG = C / 4;
    if (CG >= 8 && (CG & (CG - 1)) == 0 && CG <= 64)
      NOISE_LAUNCH_ROUNDS(first_walk_fwd_kernel<true>, grid, (size_t)PPB * 2 * C * sizeof(float),
                          (hipStream_t)stream, logmag, absmax, w, out, stats_slab, wd);
    else
      NOISE_LAUNCH_ROUNDS(first_walk_fwd_kernel<false>, grid, (size_t)PPB * 2 * C * sizeof(float),
                          (hipStream_t)stream, logmag, absmax, w, out, stats_slab, wd);
  } else {
    hipLaunchKernelGGL(first_conv_fwd_kernel, dim3(grid), dim3(256), (size_t)PPB * 2 * C * sizeof(float),
                       (hipStream_t)stream, logmag, absmax, w, out, stats_slab, d);
  }
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_first_conv_fwd");
  return 0;
}

NOISE_EXPORT int noise_first_conv_wgrad(const float* dz, const float* y_out, const float* coef3, const float* logmag,
                                        const float* absmax, const float* w, float* dw, float* scratch, int B, int H,
                                        int W, int hp, int wp, float pad_value, int C, int sy, int sx, void* stream) {
  if (check_c("noise_first_conv_wgrad", C)) return -1;
  NOISE_REQUIRE(dz && logmag && absmax && dw && scratch, "noise_first_conv_wgrad: null pointer");
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr || w != nullptr, "noise_first_conv_wgrad: coef3 needs y_out or w");
  FirstDims d{B, H, W, hp, wp, (H + hp - 1) / sy + 1, (W + wp - 1) / sx + 1, C, sy, sx, pad_value};
  LazyGrad G{dz, y_out, coef3};
  int grid = conv_grid((long)B * d.Ho * d.Wo, C);
  const int PPB = 256 / (C / 4);
  hipStream_t st = (hipStream_t)stream;
  if (sy == 2 && sx == 2) {
    FirstWalkDims wd{B, H, W, hp, wp, d.Ho, d.Wo, C, pad_value, 0, 0, 0};
    first_walk_plan(wd, BWD_MAX_BLOCKS, grid);
    const int CG = C / 4;
    const bool coop = CG >= 8 && (CG & (CG - 1)) == 0 && CG <= 64;
    const bool recomp = coef3 != nullptr && w != nullptr;   // with the weights at hand y_out is rebuilt, not read
    const size_t sm = (size_t)PPB * 9 * C * sizeof(float);
    if (coop && recomp)
      NOISE_LAUNCH_ROUNDS((first_walk_wgrad_kernel<true, true>), grid, sm, st, G, logmag, absmax, w, scratch, wd);
    else if (coop)
      NOISE_LAUNCH_ROUNDS((first_walk_wgrad_kernel<true, false>), grid, sm, st, G, logmag, absmax, w, scratch, wd);
    else if (recomp)
      NOISE_LAUNCH_ROUNDS((first_walk_wgrad_kernel<false, true>), grid, sm, st, G, logmag, absmax, w, scratch, wd);
    else
      NOISE_LAUNCH_ROUNDS((first_walk_wgrad_kernel<false, false>), grid, sm, st, G, logmag, absmax, w, scratch, wd);
  } else {
    hipLaunchKernelGGL(first_conv_wgrad_kernel, dim3(grid), dim3(256), (size_t)PPB * 9 * C * sizeof(float), st, G, logmag,
                       absmax, scratch, d);
  }
  NOISE_CHECK_LAUNCH("noise_first_conv_wgrad");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(noise_cdiv(9 * C, SLAB_COLS)), dim3(256), 0, st, scratch, grid, 9 * C, 9 * C, dw, 9, C, 0);
  NOISE_CHECK_LAUNCH("noise_first_conv_wgrad/reduce");
  return 0;
}

// enh (B,H,W) = clamp(logmag/max + filt, -1, 1) * max, filt = crop(convT(a) + bias).  Either output may be null.
NOISE_EXPORT int noise_last_convT_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y,
                                      const float* b_coef, int b_relu6, const float* w, const float* bias,
                                      const float* logmag, const float* absmax, float* enh, float* filt, int B, int Hi,
                                      int Wi, int C, int sy, int sx, int H, int W, void* stream) {
  if (check_c("noise_last_convT_fwd", C)) return -1;
  NOISE_REQUIRE(a_y && w && bias && (filt || (enh && logmag && absmax)), "noise_last_convT_fwd: null pointer");
  const int Hfull = (Hi - 1) * sy + 1, Wfull = (Wi - 1) * sx + 1;
  NOISE_REQUIRE(Hfull >= H && Wfull >= W, "noise_last_convT_fwd: output %dx%d smaller than requested crop %dx%d", Hfull, Wfull, H, W);
  LastDims d{B, Hi, Wi, C, sy, sx, H, W, Hfull - H, Wfull - W};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  if (check_size("noise_last_convT_fwd", (long)B * Hfull * Wfull, C)) return -1;
  const int CG = C / 4;
  if (sy == 2 && sx == 2 && CG >= 4 && CG <= 16 && (CG & (CG - 1)) == 0) {
    LastWalkDims wd{B, Hi, Wi, C, H, W, Hfull - H, Wfull - W, 0, 0, 0, 0, 0, 0};
    int grid;
    last_walk_plan(wd, true, 4096, grid);
    if (b_y != nullptr)
      NOISE_LAUNCH_ROUNDS(last_cell_fwd_kernel<true>, grid, 0, (hipStream_t)stream, A, Bs, w, bias, logmag,
                         absmax, enh, filt, wd);
    else
      NOISE_LAUNCH_ROUNDS(last_cell_fwd_kernel<false>, grid, 0, (hipStream_t)stream, A, Bs, w, bias, logmag,
                         absmax, enh, filt, wd);
  } else if (CG <= 16 && (CG & (CG - 1)) == 0) {
    const int grid = noise_grid((long)B * H * W, 256 / CG, 8192);
    hipLaunchKernelGGL(last_convT_fwd_split_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, A, Bs, w, bias, logmag,
                       absmax, enh, filt, d);
  } else {
    const int grid = noise_grid((long)B * H * W, 256, 4096);
    hipLaunchKernelGGL(last_convT_fwd_kernel, dim3(grid), dim3(256), (size_t)13 * C * sizeof(float), (hipStream_t)stream, A,
                       Bs, w, bias, logmag, absmax, enh, filt, d);
  }
  NOISE_CHECK_LAUNCH("noise_last_convT_fwd");
  return 0;
}

// g_in (B,Hi,Wi,C), dw (C,1,3,3), dbias (1).  Epilogue as in noise_dw3x3_bwd; scratch rows are 12*C floats with
// the producer sums in columns [10*C, 12*C).  scratch: noise_conv_max_blocks()*12*C floats.
NOISE_EXPORT int noise_last_convT_bwd(const float* d_enh, const float* enh, const float* absmax, const float* a_y,
                                      const float* a_coef, int a_relu6, const float* b_y, const float* b_coef,
                                      int b_relu6, const float* w, float* g_in, float* dw, float* dbias, float* scratch,
                                      const float* g_other, const float* epi_coef, int epi_relu6, int* slab_rows,
                                      int B, int Hi, int Wi, int C, int sy, int sx, int H, int W, void* stream) {
  if (check_c("noise_last_convT_bwd", C)) return -1;
  NOISE_REQUIRE(d_enh && enh && absmax && a_y && w && g_in && dw && dbias && scratch, "noise_last_convT_bwd: null pointer");
  const int Hfull = (Hi - 1) * sy + 1, Wfull = (Wi - 1) * sx + 1;
  NOISE_REQUIRE(Hfull >= H && Wfull >= W, "noise_last_convT_bwd: bad crop");
  LastDims d{B, Hi, Wi, C, sy, sx, H, W, Hfull - H, Wfull - W};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  int grid = conv_grid((long)B * Hi * Wi, C);
  const int PPB = 256 / (C / 4);
  hipStream_t st = (hipStream_t)stream;
  BwdEpilogue E{g_other, epi_coef, epi_relu6, nullptr};
  if (sy == 2 && sx == 2) {
    LastWalkDims wd{B, Hi, Wi, C, H, W, Hfull - H, Wfull - W, 0, 0, 0, 0, 0, 0};
    last_walk_plan(wd, false, BWD_MAX_BLOCKS, grid);
    const size_t sm = (size_t)PPB * 12 * C * sizeof(float);
    const bool hb = b_y != nullptr, hg = g_other != nullptr;
    const int CG = C / 4;
    const bool coop = CG >= 8 && CG <= 64 && (CG & (CG - 1)) == 0;
#define LAUNCH_LAST(HB, HG)                                                                                        \
  do {                                                                                                             \
    if (coop)                                                                                                      \
      NOISE_LAUNCH_ROUNDS((last_gather_bwd_kernel<HB, HG, true>), grid, sm, st, d_enh, enh, absmax, A, \
                         Bs, w, g_in, scratch, E, wd);                                                             \
    else                                                                                                           \
      NOISE_LAUNCH_ROUNDS((last_gather_bwd_kernel<HB, HG, false>), grid, sm, st, d_enh, enh, absmax, A, \
                         Bs, w, g_in, scratch, E, wd);                                                             \
  } while (0)
    if (hb && hg) LAUNCH_LAST(true, true);
    else if (hb) LAUNCH_LAST(true, false);
    else if (hg) LAUNCH_LAST(false, true);
    else LAUNCH_LAST(false, false);
#undef LAUNCH_LAST
  } else {
    hipLaunchKernelGGL(last_convT_bwd_kernel, dim3(grid), dim3(256), (size_t)PPB * 12 * C * sizeof(float), st, d_enh, enh,
                       absmax, A, Bs, w, g_in, scratch, E, d);
  }
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_last_convT_bwd");
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(noise_cdiv(9 * C, SLAB_COLS)), dim3(256), 0, st, scratch, grid, 9 * C, 12 * C, dw, 9, C, 0);
  NOISE_CHECK_LAUNCH("noise_last_convT_bwd/reduce");
  // the bias gradient sits at slab column 9*C (slot 9, channel 0)
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, st, scratch + 9 * C, grid, 1, 12 * C, dbias, 0, 0, 0);
  NOISE_CHECK_LAUNCH("noise_last_convT_bwd/bias");
  return 0;
}
