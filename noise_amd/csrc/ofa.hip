// Shape-generic NHWC fp32 kernels for the once-for-all spectral U-Net forward (train/layers.py:496-668), gfx950.
//
// That network is the spectral denoiser with every dimension made a run-time choice: the depthwise kernel is the centre
// k x k of a 7 x 7 filter passed through learned 25x25 / 9x9 matrices (layers.py:162-196), 1x1 convolutions and
// BatchNorms use the leading channels of their maximum-width parameters (layers.py:253-272, :284-308), blocks at the
// end of a stage can be skipped and stride-1 blocks with equal widths are residual (layers.py:457-458).  Channel
// counts run from 8 to 1920, so nothing here is specialised to a width: a 1x1 convolution is an fp32 MFMA product tiled
// over pixels, output channels and input-channel chunks, with the weight matrix read in place (row stride = maximum
// input width); the depthwise convolution (plain or transposed, any odd k <= 7, stride 1 or 2) is a gather per output
// pixel.  The tensors follow the lazy BatchNorm protocol of the other kernels: a convolution writes its raw output and
// per-workgroup partial sums of y and y*y (slab rows of 2C floats, folded by noise_bn_finalize_fwd into [scale|shift]),
// and the consumer applies scale, shift and ReLU6 while loading.  Block outputs are materialised by the apply kernel,
// which also adds the residual input and the (cropped) encoder skip (layers.py:662).
#include "lazy.h"

namespace {

constexpr int OFA_MAX_ROWS = 1024;   // rows of a partial slab = cap of the pixel dimension of a grid

// ---- thread layout of the channel-vector kernels ---------------------------------------------------------------------
// A block is cl channel lanes (one float4 of channels each) x 256/cl pixel lanes; blockIdx.x walks channel chunks,
// blockIdx.y starts the pixel walk.  cl is a power of two <= 64 chosen on the host from C.
struct CvLane {
  int c;        // first of the lane's 4 channels (clamped into range)
  bool ok;      // the lane's channels exist
  int pl, PL;   // pixel lane and number of pixel lanes
};

__device__ __forceinline__ CvLane cv_lane(int C, int cl) {
  CvLane r;
  const int cgl = threadIdx.x & (cl - 1);
  r.PL = 256 / cl;
  r.pl = threadIdx.x / cl;
  const int cg = blockIdx.x * cl + cgl, CG = C >> 2;
  r.ok = cg < CG;
  r.c = 4 * (r.ok ? cg : CG - 1);
  return r;
}

// per-channel partial sums of a block -> one slab row ([sum(C) | sum of squares(C)]); every thread of the block calls it
__device__ __forceinline__ void cv_stats(const CvLane& L, int C, int cl, float4 s1, float4 s2, float* __restrict__ slab) {
  __shared__ float red[256 * 8];
  float* mine = red + threadIdx.x * 8;
  st4(mine, s1);
  st4(mine + 4, s2);
  __syncthreads();
  if (L.pl == 0 && L.ok) {
    float4 a = f4(0.f), b = f4(0.f);
    for (int r = 0; r < L.PL; ++r) {
      a = a + ld4(red + (r * cl + threadIdx.x) * 8);
      b = b + ld4(red + (r * cl + threadIdx.x) * 8 + 4);
    }
    float* row = slab + (size_t)blockIdx.y * 2 * C;
    st4(row + L.c, a);
    st4(row + C + L.c, b);
  }
}

// ---- first layer: 1 -> C, 3x3, stride 2, zero padding 1, on the input padded at the top / left (layers.py:646) ---------
struct FirstDims {
  int B, H, W, hp, wp, Ho, Wo, C, cl;
};

__global__ __launch_bounds__(256) void ofa_first_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ out, float* __restrict__ slab, FirstDims d) {
  const CvLane L = cv_lane(d.C, d.cl);
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = make_float4(w[(L.c + 0) * 9 + t], w[(L.c + 1) * 9 + t], w[(L.c + 2) * 9 + t], w[(L.c + 3) * 9 + t]);
  float4 s1 = f4(0.f), s2 = f4(0.f);
  const long npix = (long)d.B * d.Ho * d.Wo;
  for (long p = (long)blockIdx.y * L.PL + L.pl; p < npix; p += (long)gridDim.y * L.PL) {
    const int wo = (int)(p % d.Wo);
    const long t = p / d.Wo;
    const int ho = (int)(t % d.Ho), b = (int)(t / d.Ho);
    float4 acc = f4(0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * ho - 1 + ky - d.hp;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * wo - 1 + kx - d.wp;
        const bool in = iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        const float v = in ? x[((size_t)b * d.H + (in ? iy : 0)) * d.W + (in ? ix : 0)] : 0.f;
        acc = fma4(f4(v), wt[ky * 3 + kx], acc);
      }
    }
    if (L.ok) {
      st4(out + (size_t)p * d.C + L.c, acc);
      s1 = s1 + acc;
      s2 = fma4(acc, acc, s2);
    }
  }
  cv_stats(L, d.C, d.cl, s1, s2, slab);
}

// ---- depthwise k x k, plain or transposed, padding k/2 (layers.py:198-215) ----------------------------------------------
struct DwDims {
  int B, Hi, Wi, Ho, Wo, C, k, s, cl;
};

template <bool TR>
__global__ __launch_bounds__(256) void ofa_dw_kernel(LazySrc src, const float* __restrict__ wt, float* __restrict__ out,
                                                     float* __restrict__ slab, DwDims d) {
  const CvLane L = cv_lane(d.C, d.cl);
  const Lazy4 a = lazy4_init(src, d.C, L.c);
  const int k = d.k, s = d.s, pad = d.k / 2, C = d.C;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  const long npix = (long)d.B * d.Ho * d.Wo;
  for (long p = (long)blockIdx.y * L.PL + L.pl; p < npix; p += (long)gridDim.y * L.PL) {
    const int wo = (int)(p % d.Wo);
    const long t = p / d.Wo;
    const int ho = (int)(t % d.Ho), b = (int)(t / d.Ho);
    // taps whose source row / column exists: plain conv iy = ho*s - pad + ky; transposed ho = iy*s - pad + ky, i.e. only
    // the taps congruent to ho + pad modulo s, walking the source backwards
    int ky_lo, ky_hi, kx_lo, kx_hi;
    if (TR) {
      ky_lo = max(0, ho + pad - s * (d.Hi - 1));
      ky_lo += (ho + pad - ky_lo) % s;
      ky_hi = min(k - 1, ho + pad);
      kx_lo = max(0, wo + pad - s * (d.Wi - 1));
      kx_lo += (wo + pad - kx_lo) % s;
      kx_hi = min(k - 1, wo + pad);
    } else {
      ky_lo = max(0, pad - ho * s);
      ky_hi = min(k - 1, d.Hi - 1 + pad - ho * s);
      kx_lo = max(0, pad - wo * s);
      kx_hi = min(k - 1, d.Wi - 1 + pad - wo * s);
    }
    const int step = TR ? s : 1;
    float4 acc = f4(0.f);
    for (int ky = ky_lo; ky <= ky_hi; ky += step) {
      const int iy = TR ? (ho + pad - ky) / s : ho * s - pad + ky;
      const float* rowp = a.y + ((size_t)b * d.Hi + iy) * d.Wi * C + L.c;
      const float* wrow = wt + (size_t)ky * k * C + L.c;
#pragma unroll 4
      for (int kx = kx_lo; kx <= kx_hi; kx += step) {
        const int ix = TR ? (wo + pad - kx) / s : wo * s - pad + kx;
        acc = fma4(lazy4_act(a, ld4(rowp + (size_t)ix * C)), ld4(wrow + (size_t)kx * C), acc);
      }
    }
    if (L.ok) {
      st4(out + (size_t)p * C + L.c, acc);
      s1 = s1 + acc;
      s2 = fma4(acc, acc, s2);
    }
  }
  cv_stats(L, C, d.cl, s1, s2, slab);
}

// Stride 1 (three of four blocks of a stage): a thread owns 4 channels of 4 consecutive output columns, so a filter row
// costs K + 3 source loads and K weight loads for 4 K multiply-adds instead of 8 K loads.  A transposed stride-1
// convolution is the plain one with the filter flipped (the filter kernel writes it that way).
template <int K>
__global__ __launch_bounds__(256) void ofa_dw1_kernel(LazySrc src, const float* __restrict__ wt, float* __restrict__ out,
                                                      float* __restrict__ slab, DwDims d) {
  const CvLane L = cv_lane(d.C, d.cl);
  const Lazy4 a = lazy4_init(src, d.C, L.c);
  constexpr int pad = K / 2;
  const int C = d.C, H = d.Hi, W = d.Wi;   // output size = input size
  const int WQ = (W + 3) >> 2;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  const long nquad = (long)d.B * H * WQ;
  for (long qd = (long)blockIdx.y * L.PL + L.pl; qd < nquad; qd += (long)gridDim.y * L.PL) {
    const int wq = (int)(qd % WQ);
    const long t = qd / WQ;
    const int ho = (int)(t % H), b = (int)(t / H), wo0 = 4 * wq;
    float4 acc[4] = {f4(0.f), f4(0.f), f4(0.f), f4(0.f)};
    const int ky_lo = max(0, pad - ho), ky_hi = min(K - 1, H - 1 + pad - ho);
    for (int ky = ky_lo; ky <= ky_hi; ++ky) {
      const float* rowp = a.y + ((size_t)b * H + ho - pad + ky) * W * C + L.c;
      float4 in[K + 3], w[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) w[kx] = ld4(wt + (size_t)(ky * K + kx) * C + L.c);
#pragma unroll
      for (int u = 0; u < K + 3; ++u) {
        const int ix = wo0 - pad + u;
        const bool v = ix >= 0 && ix < W;
        const float4 r = lazy4_act(a, ld4(rowp + (size_t)(v ? ix : 0) * C));
        in[u] = make_float4(v ? r.x : 0.f, v ? r.y : 0.f, v ? r.z : 0.f, v ? r.w : 0.f);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[j] = fma4(in[j + kx], w[kx], acc[j]);
    }
    if (L.ok) {
      float* o = out + (((size_t)b * H + ho) * W + wo0) * C + L.c;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (wo0 + j < W) {
          st4(o + (size_t)j * C, acc[j]);
          s1 = s1 + acc[j];
          s2 = fma4(acc[j], acc[j], s2);
        }
    }
  }
  cv_stats(L, C, d.cl, s1, s2, slab);
}

// Stride 2, plain or transposed: a thread owns 4 channels of 2 consecutive output columns.  Plain: a filter row costs
// K + 2 source loads for 2 K multiply-adds.  Transposed: output column 2x + par takes the taps kx = par + pad (mod 2)
// from source columns x + (par + pad - kx) / 2, so the pair shares its (pad + 1) / 2 + pad / 2 + 1 source columns and
// no tap is tested at run time.  Taps are summed in the order of ofa_dw_kernel (ky, then kx ascending), so the two
// kernels agree bit for bit.
template <int K, bool TR>
__global__ __launch_bounds__(256) void ofa_dw2_kernel(LazySrc src, const float* __restrict__ wt, float* __restrict__ out,
                                                      float* __restrict__ slab, DwDims d) {
  const CvLane L = cv_lane(d.C, d.cl);
  const Lazy4 a = lazy4_init(src, d.C, L.c);
  constexpr int pad = K / 2;
  constexpr int OMIN = -(pad / 2), NIN = TR ? (pad + 1) / 2 - OMIN + 1 : K + 2;
  const int C = d.C;
  const unsigned WP = (unsigned)(d.Wo + 1) >> 1, npair = (unsigned)d.B * d.Ho * WP;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  for (unsigned pr = blockIdx.y * L.PL + L.pl; pr < npair; pr += gridDim.y * L.PL) {
    const unsigned t = pr / WP;
    const int x = (int)(pr - t * WP), b = (int)(t / (unsigned)d.Ho), ho = (int)(t - (unsigned)b * d.Ho);
    int ky_lo, ky_hi;
    if (TR) {
      ky_lo = max(0, ho + pad - 2 * (d.Hi - 1));
      ky_lo += (ho + pad - ky_lo) & 1;
      ky_hi = min(K - 1, ho + pad);
    } else {
      ky_lo = max(0, pad - 2 * ho);
      ky_hi = min(K - 1, d.Hi - 1 + pad - 2 * ho);
    }
    float4 acc[2] = {f4(0.f), f4(0.f)};
    for (int ky = ky_lo; ky <= ky_hi; ky += TR ? 2 : 1) {
      const int iy = TR ? (ho + pad - ky) >> 1 : 2 * ho - pad + ky;
      const float* rowp = a.y + ((size_t)b * d.Hi + iy) * d.Wi * C + L.c;
      float4 in[NIN], w[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) w[kx] = ld4(wt + (size_t)(ky * K + kx) * C + L.c);
#pragma unroll
      for (int u = 0; u < NIN; ++u) {
        const int ix = TR ? x + OMIN + u : 4 * x - pad + u;
        const bool v = ix >= 0 && ix < d.Wi;
        const float4 r = lazy4_act(a, ld4(rowp + (size_t)(v ? ix : 0) * C));
        in[u] = make_float4(v ? r.x : 0.f, v ? r.y : 0.f, v ? r.z : 0.f, v ? r.w : 0.f);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          if (TR) {
            if (((kx - j - pad) & 1) == 0) acc[j] = fma4(in[(j + pad - kx) / 2 - OMIN], w[kx], acc[j]);
          } else {
            acc[j] = fma4(in[2 * j + kx], w[kx], acc[j]);
          }
        }
    }
    if (L.ok) {
      float* o = out + (((size_t)b * d.Ho + ho) * d.Wo + 2 * x) * C + L.c;
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (2 * x + j < d.Wo) {
          st4(o + (size_t)j * C, acc[j]);
          s1 = s1 + acc[j];
          s2 = fma4(acc[j], acc[j], s2);
        }
    }
  }
  cv_stats(L, C, d.cl, s1, s2, slab);
}

// ---- the active depthwise filter (layers.py:162-196), written tap-major: wt[tap][c] ---------------------------------------
// A block takes 64 channels: the filters sit in LDS, and each step (centre crop of the current filter, times the transform
// matrix: y_j = sum_i x_i M[j][i]) runs with one thread per (channel, output tap).
__global__ __launch_bounds__(256) void ofa_dw_filter_kernel(const float* __restrict__ weight, int kmax, const float* __restrict__ m1,
                                                            int k1, const float* __restrict__ m2, int k2, int k, int C,
                                                            int flip, float* __restrict__ wt) {
  __shared__ float f[64 * 49], g[64 * 49];
  const int c0 = blockIdx.x * 64, nc = min(64, C - c0), kk = kmax * kmax;
  for (int i = threadIdx.x; i < nc * kk; i += 256) f[(i / kk) * 49 + i % kk] = weight[(size_t)c0 * kk + i];
  __syncthreads();
  int cur = kmax;
  float* src = f;
  float* dst = g;
  for (int step = 0; step < 2; ++step) {
    const float* M = step == 0 ? m1 : m2;
    const int tk = step == 0 ? k1 : k2;
    if (M == nullptr || cur <= k) break;
    const int st = cur / 2 - tk / 2, n = tk * tk;
    for (int w = threadIdx.x; w < nc * n; w += 256) {
      const int c = w / n, j = w - c * n;
      float acc = 0.f;
      for (int i = 0; i < n; ++i) acc = fmaf(src[c * 49 + (st + i / tk) * cur + st + i % tk], M[j * n + i], acc);
      dst[c * 49 + j] = acc;
    }
    __syncthreads();
    float* t = src;
    src = dst;
    dst = t;
    cur = tk;
  }
  const int st = cur / 2 - k / 2, n = k * k;   // without transform matrices the active filter is the centre crop
  for (int w = threadIdx.x; w < nc * n; w += 256) {
    const int j = w / nc, c = w - j * nc;
    wt[(size_t)(flip ? n - 1 - j : j) * C + c0 + c] = src[c * 49 + (st + j / k) * cur + st + j % k];
  }
}

// ---- 1x1 convolution with the leading K x N corner of a weight matrix (layers.py:253-272) ---------------------------------
// out[p][n] = sum_k a[p][k] W[n*ldw + k].  v_mfma_f32_16x16x4_f32 with A = weights (rows = n) and B = activations
// (columns = pixels), so a lane ends up with 4 consecutive output channels of one pixel.  A workgroup owns NT*16 output
// channels (blockIdx.x) and walks 64-pixel tiles from blockIdx.y; both operands pass through LDS in chunks of 32 input
// channels (row pitch 36 floats: 16-byte aligned rows, two lanes per bank).
struct PwGDims {
  long npix;
  int K, N, ldw;
};
constexpr int PWG_KC = 32, PWG_PITCH = 36;

// GRAD: the input operand is a lazily evaluated gradient (dz, y, coef3 of lazy.h) instead of a lazily normalised
// activation, which makes the same product the input gradient of a 1x1 layer (W = the layer's weight transposed).
// slab == nullptr skips the statistics.
template <int NT, bool GRAD>
__global__ __launch_bounds__(256, (GRAD && NT >= 6) ? 2 : (NT <= 4 && !GRAD ? 4 : 3)) void ofa_pw_kernel(LazySrc src, LazyGrad grad, const float* __restrict__ W,
                                                     float* __restrict__ out, float* __restrict__ slab, PwGDims d,
                                                     LazySrc epi) {
  __shared__ float As[64 * PWG_PITCH];
  __shared__ float Ws[NT * 16 * PWG_PITCH];
  __shared__ float red[4 * 2 * NT * 16];
  __shared__ float ecf[GRAD ? 5 * NT * 16 : 4];   // backward epilogue: scale, shift, mean, invstd, mean_lo of this workgroup's channels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int K = d.K, N = d.N;
  const int n0 = blockIdx.x * NT * 16;
  const bool has_epi = GRAD && epi.y != nullptr;
  if (has_epi) {   // visible after the first barrier of the tile loop; a workgroup without tiles never reads it
    for (int i = tid; i < 5 * NT * 16; i += 256) {
      const int r = i / (NT * 16), c = i - r * (NT * 16);
      ecf[i] = n0 + c < N ? epi.coef[r * N + n0 + c] : 0.f;
    }
  }
  const float alo = src.relu6 ? 0.f : -INFINITY, ahi = src.relu6 ? 6.f : INFINITY;
  float4 s1[NT], s2[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) s1[m] = s2[m] = f4(0.f);
  const long ntiles = (d.npix + 63) / 64;
  const int nchunk = (K + PWG_KC - 1) / PWG_KC;
  // staging maps: a thread moves 2 float4 of the input chunk (64 pixels x 8 channel quads) and NT/2 of the weight chunk
  const int kq = tid & 7, r0 = tid >> 3;
  constexpr int RW = (NT + 1) / 2;   // weight rows a thread stages (32 rows apart)
  float4 ra[2], rw[RW];
  float4 rb[GRAD ? 2 : 1];   // GRAD: raw layer output
  int kcur = 0;               // first channel of the chunk held in ra / rb (the coefficient vectors are read when it is staged)
  const bool affine = GRAD && grad.coef3 != nullptr;
  const float* const ay = GRAD ? grad.dz : src.y;
  unsigned vin = 0;   // bit i: ra[i] holds an existing element; bit 8+i: rw[i] does
  auto fetch = [&](long tile, int ch) {
    const int kk = ch * PWG_KC + 4 * kq;
    const bool kin = kk < K;
    const int kc = kin ? kk : 0;
    vin = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long pix = tile * 64 + r0 + 32 * i;
      const bool in = kin && pix < d.npix;
      ra[i] = ld4(ay + (size_t)(in ? pix : 0) * K + kc);
      if (GRAD) rb[i] = affine ? ld4(grad.y + (size_t)(in ? pix : 0) * K + kc) : f4(0.f);
      vin |= in ? (1u << i) : 0u;
    }
    kcur = kc;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int nr = r0 + 32 * i;
      const bool in = kin && nr < NT * 16 && n0 + nr < N;
      rw[i] = ld4(W + (size_t)(in ? n0 + nr : 0) * d.ldw + kc);
      vin |= in ? (256u << i) : 0u;
    }
  };
  long tile = blockIdx.y;
  int ch = 0;
  if (tile < ntiles) fetch(tile, 0);
  f32x4 acc[NT];
  while (tile < ntiles) {
    // registers -> LDS (the input chunk normalised and clamped on the way), then request the next chunk so that its
    // memory round trip runs under this chunk's matrix products
    float4 rsc, rsh, rg[GRAD ? 4 : 1];
    if (GRAD) {   // dy = s*((dz - k1_hi) - k1_lo) + a2*((y - mean_hi) - mean_lo)
      rsc = affine ? ld4(grad.coef3 + kcur) : f4(1.f);
      rsh = affine ? ld4(grad.coef3 + 3 * K + kcur) : f4(0.f);
      rg[0] = affine ? ld4(grad.coef3 + K + kcur) : f4(0.f);
      rg[1] = affine ? ld4(grad.coef3 + 2 * K + kcur) : f4(0.f);
      rg[2] = affine ? ld4(grad.coef3 + 4 * K + kcur) : f4(0.f);
      rg[3] = affine ? ld4(grad.coef3 + 5 * K + kcur) : f4(0.f);
    } else {
      rsc = src.coef ? ld4(src.coef + kcur) : f4(1.f);
      rsh = src.coef ? ld4(src.coef + K + kcur) : f4(0.f);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v;
      if (GRAD) v = fma4(rsh, sub4(sub4(rb[i], rg[2]), rg[3]), rsc * sub4(sub4(ra[i], rg[0]), rg[1]));
      else v = clamp4(fma4(ra[i], rsc, rsh), alo, ahi);
      st4(As + (r0 + 32 * i) * PWG_PITCH + 4 * kq, ((vin >> i) & 1u) ? v : f4(0.f));
    }
#pragma unroll
    for (int i = 0; i < RW; ++i)
      if (NT % 2 == 0 || r0 + 32 * i < NT * 16) st4(Ws + (r0 + 32 * i) * PWG_PITCH + 4 * kq, ((vin >> (8 + i)) & 1u) ? rw[i] : f4(0.f));
    __syncthreads();
    long ntile = tile;
    int nch = ch + 1;
    if (nch == nchunk) {
      nch = 0;
      ntile = tile + gridDim.y;
    }
    if (ntile < ntiles) fetch(ntile, nch);
    if (ch == 0) {
#pragma unroll
      for (int m = 0; m < NT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float4 yr[GRAD ? NT : 1];
    if (has_epi && ch == nchunk - 1) {   // the epilogue's raw tile arrives while the last chunk is multiplied
      const long pix = tile * 64 + 16 * wave + p;
#pragma unroll
      for (int m = 0; m < (GRAD ? NT : 0); ++m) {
        const int n = n0 + 16 * m + 4 * q;
        yr[m] = ld4(epi.y + (size_t)(pix < d.npix ? pix : 0) * N + (n < N ? n : 0));
      }
    }
    {
      // Which input channel a lane feeds to which of the chunk's eight matrix products is free as long as both operands
      // agree: lane quarter q takes channels 8q .. 8q+7, one per product, so each operand row is two 16-byte LDS reads
      // (rows 36 floats apart put the 16 lanes of a quarter on disjoint banks) instead of eight 4-byte ones.
      const float* ar = As + (16 * wave + p) * PWG_PITCH + 8 * q;
      const float4 b0 = ld4(ar), b1 = ld4(ar + 4);
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const float* wr = Ws + (16 * m + p) * PWG_PITCH + 8 * q;
        const float4 w0 = ld4(wr), w1 = ld4(wr + 4);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, b0.x, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, b0.y, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, b0.z, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, b0.w, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, b1.x, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, b1.y, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, b1.z, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, b1.w, acc[m], 0, 0, 0);
      }
    }
    __syncthreads();
    if (ch == nchunk - 1) {
      const long pix = tile * 64 + 16 * wave + p;
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        const int n = n0 + 16 * m + 4 * q;
        float4 o = make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]);
        if (pix < d.npix && n < N) {
          if (has_epi) {
            // backward epilogue: `out` is the gradient of the activated output of the BatchNorm (+ReLU6) whose raw
            // input is epi.y; mask it and leave that BatchNorm's two backward sums (sum g, sum g * xhat) in the slab
            const float4 yraw = yr[GRAD ? m : 0];
            const float* ec = ecf + 16 * m + 4 * q;
            if (epi.relu6) {
              const float4 v = fma4(yraw, ld4(ec), ld4(ec + NT * 16));
              o.x = (v.x > 0.f && v.x < 6.f) ? o.x : 0.f;
              o.y = (v.y > 0.f && v.y < 6.f) ? o.y : 0.f;
              o.z = (v.z > 0.f && v.z < 6.f) ? o.z : 0.f;
              o.w = (v.w > 0.f && v.w < 6.f) ? o.w : 0.f;
            }
            const float4 xh = sub4(sub4(yraw, ld4(ec + 2 * NT * 16)), ld4(ec + 4 * NT * 16)) * ld4(ec + 3 * NT * 16);
            st4(out + (size_t)pix * N + n, o);
            s1[m] = s1[m] + o;
            s2[m] = fma4(o, xh, s2[m]);
          } else {
            st4(out + (size_t)pix * N + n, o);
            s1[m] = s1[m] + o;
            s2[m] = fma4(o, o, s2[m]);
          }
        }
      }
    }
    tile = ntile;
    ch = nch;
  }
  if (slab == nullptr) return;
  // statistics: over the 16 pixel lanes that share q, then over the 4 waves
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    float v[8] = {s1[m].x, s1[m].y, s1[m].z, s1[m].w, s2[m].x, s2[m].y, s2[m].z, s2[m].w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v[e] += __shfl_xor(v[e], o, 64);
    }
    if (p == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[(wave * 2 + 0) * NT * 16 + 16 * m + 4 * q + e] = v[e];
        red[(wave * 2 + 1) * NT * 16 + 16 * m + 4 * q + e] = v[4 + e];
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * NT * 16; i += 256) {
    const int j = i / (NT * 16), c = i - j * NT * 16;
    if (n0 + c < N) {
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) s += red[(wv * 2 + j) * NT * 16 + c];
      slab[(size_t)blockIdx.y * 2 * N + (size_t)j * N + n0 + c] = s;
    }
  }
}

// Channel tiles (of 16) per workgroup for N output channels: the instantiated count (1, 2, 4, 6, 8) that pads N the
// least, among those that split N into at most one column group more than 128-channel groups would (every column group
// stages the input tile again); ties go to the larger count.  Padding is matrix-core work: 96 channels as one group of
// 128 spends a third of the MFMAs on zeros, 16 channels as a group of 64 three quarters.  Small problems (under two
// workgroups per CU with 128-channel groups) take at most 64 channels per workgroup.
inline int pw_channel_tiles(int N, long ntiles) {
  static const int cand[5] = {1, 2, 4, 6, 8};
  const int max_groups = noise_cdiv(N, 128) + 1;
  const bool small = ntiles * noise_cdiv(N, 128) < 512;
  int best = 8, best_pad = 1 << 30;
  for (int c = 0; c < 5; ++c) {
    const int nt = cand[c], groups = noise_cdiv(N, nt * 16), padded = groups * nt * 16;
    if (small && nt > 4) continue;
    if (groups > max_groups && !(small && nt == 4)) continue;
    if (padded <= best_pad) {
      best_pad = padded;
      best = nt;
    }
  }
  return best;
}

// Pixel rows of a gx x gy grid cut down to whole rounds of the GPU: workgroups walk their tiles with a fixed stride, so
// all of them carry the same work and a last, partly filled round costs a full one (1024 workgroups over 768 resident
// slots take two rounds at two thirds of the machine).  per_cu caches the kernel's resident workgroups per CU.
template <typename KernelT>
inline int whole_round_rows(KernelT kernel, int& per_cu, int gx, int gy) {
  if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1))
    per_cu = -1;
  if (per_cu < 1) return gy;
  const long round = (long)noise_num_cus() * per_cu, total = (long)gx * gy;
  if (total <= round) return gy;
  const long rows = total / round * round / gx;
  return rows < 1 ? 1 : (int)rows;
}

// ---- block output: BatchNorm (+ReLU6) of a raw tensor, plus the residual input, plus the cropped skip -------------------
// out[b][h][w][c] = act(scale*y[b][h+oy][w+ox][c] + shift) + res[b][h][w][c] + skip[b][h+sy][w+sx][c]
// (layers.py:457-458 residual, :662 skip cropped from the end, :666 output crop).
struct ApplyDims {
  int B, Ho, Wo, C, Hy, Wy, Hs, Ws, cl;
};

__global__ __launch_bounds__(256) void ofa_apply_kernel(LazySrc y, const float* __restrict__ res, const float* __restrict__ skip,
                                                        float* __restrict__ out, ApplyDims d) {
  const CvLane L = cv_lane(d.C, d.cl);
  if (!L.ok) return;
  const Lazy4 a = lazy4_init(y, d.C, L.c);
  const int oy = d.Hy - d.Ho, ox = d.Wy - d.Wo, sy = d.Hs - d.Ho, sx = d.Ws - d.Wo, C = d.C;
  const long npix = (long)d.B * d.Ho * d.Wo;
  for (long p = (long)blockIdx.y * L.PL + L.pl; p < npix; p += (long)gridDim.y * L.PL) {
    const int wo = (int)(p % d.Wo);
    const long t = p / d.Wo;
    const int ho = (int)(t % d.Ho), b = (int)(t / d.Ho);
    float4 v = lazy4_at(a, (((size_t)b * d.Hy + ho + oy) * d.Wy + wo + ox) * C + L.c);
    if (res != nullptr) v = v + ld4(res + (size_t)p * C + L.c);
    if (skip != nullptr) v = v + ld4(skip + (((size_t)b * d.Hs + ho + sy) * d.Ws + wo + sx) * C + L.c);
    st4(out + (size_t)p * C + L.c, v);
  }
}

// ---- last layer: C -> 1 transposed 3x3, stride 2, padding 1 (layers.py:556-564 through :253-272) --------------------------
// gl lanes (a power of two >= C/4) share an output pixel and split the channels; the raw single-channel output and its
// partial sums (slab rows of 2 floats) go to the one-channel BatchNorm.
struct LastDims {
  int B, Hi, Wi, Ho, Wo, C, gl;
};

__global__ __launch_bounds__(256) void ofa_last_kernel(LazySrc src, const float* __restrict__ w, float* __restrict__ out,
                                                       float* __restrict__ slab, LastDims d) {
  __shared__ float red[8];
  const int gl = d.gl, cgl = threadIdx.x & (gl - 1), PL = 256 / gl, pl = threadIdx.x / gl;
  const int CG = d.C >> 2;
  const bool ok = cgl < CG;
  const int c = 4 * (ok ? cgl : CG - 1), C = d.C;
  const Lazy4 a = lazy4_init(src, C, c);
  float4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
  float s1 = 0.f, s2 = 0.f;
  const long npix = (long)d.B * d.Ho * d.Wo;
  const long nround = (npix + PL - 1) / PL;   // every lane of a group takes part in the shuffles of every round
  for (long r = blockIdx.x; r < nround; r += gridDim.x) {
    const long p = r * PL + pl;
    const bool pv = p < npix;
    const long pc = pv ? p : 0;
    const int wo = (int)(pc % d.Wo);
    const long t = pc / d.Wo;
    const int ho = (int)(t % d.Ho), b = (int)(t / d.Ho);
    float4 acc = f4(0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int u = ho + 1 - ky, iy = u / 2;
      const bool vy = u >= 0 && (u & 1) == 0 && iy < d.Hi;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int v = wo + 1 - kx, ix = v / 2;
        const bool in = vy && v >= 0 && (v & 1) == 0 && ix < d.Wi && ok && pv;
        if (in) acc = fma4(lazy4_at(a, (((size_t)b * d.Hi + iy) * d.Wi + ix) * C + c), wt[ky * 3 + kx], acc);
      }
    }
    float sum = (acc.x + acc.y) + (acc.z + acc.w);
    for (int o = gl >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (cgl == 0 && pv) {
      out[p] = sum;
      s1 += sum;
      s2 = fmaf(sum, sum, s2);
    }
  }
  s1 = wave_reduce_sum(s1);
  s2 = wave_reduce_sum(s2);
  if ((threadIdx.x & 63) == 0) {
    red[(threadIdx.x >> 6) * 2] = s1;
    red[(threadIdx.x >> 6) * 2 + 1] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 2) slab[(size_t)blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x] + red[2 + threadIdx.x]) + (red[4 + threadIdx.x] + red[6 + threadIdx.x]);
}

// ---- network output: one-channel BatchNorm + ReLU6, cropped from the end to the input size (layers.py:666) ---------------
__global__ void ofa_final_kernel(const float* __restrict__ y, const float* __restrict__ coef, float* __restrict__ out, int B,
                                 int Hy, int Wy, int H, int W, int relu6) {
  const long n = (long)B * H * W;
  const float sc = coef[0], sh = coef[1];
  const float lo = relu6 ? 0.f : -INFINITY, hi = relu6 ? 6.f : INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int wv = (int)(i % W);
    const long t = i / W;
    const int h = (int)(t % H), b = (int)(t / H);
    const float v = fmaf(y[((size_t)b * Hy + h + (Hy - H)) * Wy + wv + (Wy - W)], sc, sh);
    out[i] = __builtin_amdgcn_fmed3f(v, lo, hi);
  }
}

int channel_lanes(int C) {
  const int CG = C / 4;
  int cl = 1;
  while (cl < CG && cl < 64) cl <<= 1;
  return cl;
}

int pixel_rows(long npix, int per_block) {
  long r = (npix + per_block - 1) / per_block;
  if (r > OFA_MAX_ROWS) r = OFA_MAX_ROWS;
  if (r < 1) r = 1;
  return (int)r;
}

}  // namespace

NOISE_EXPORT int noise_ofa_max_rows(void) { return OFA_MAX_ROWS; }

NOISE_EXPORT int noise_ofa_first_conv(const float* x, const float* w, float* out, float* slab, int* rows, int B, int H, int W,
                                      int hp, int wp, int C, void* stream) {
  NOISE_REQUIRE(x && w && out && slab && rows, "noise_ofa_first_conv: null pointer");
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && hp >= 0 && wp >= 0 && C >= 4 && C % 4 == 0, "noise_ofa_first_conv: bad shape");
  FirstDims d{B, H, W, hp, wp, (H + hp - 1) / 2 + 1, (W + wp - 1) / 2 + 1, C, channel_lanes(C)};
  const int gy = pixel_rows((long)B * d.Ho * d.Wo, 256 / d.cl * 4);
  hipLaunchKernelGGL(ofa_first_kernel, dim3(noise_cdiv(C / 4, d.cl), gy), dim3(256), 0, (hipStream_t)stream, x, w, out, slab, d);
  NOISE_CHECK_LAUNCH("noise_ofa_first_conv");
  *rows = gy;
  return 0;
}

NOISE_EXPORT int noise_ofa_dw_filter(const float* weight, int kmax, const float* m1, int k1, const float* m2, int k2, int k,
                                     int C, int flip, float* wt, void* stream) {
  NOISE_REQUIRE(weight && wt, "noise_ofa_dw_filter: null pointer");
  NOISE_REQUIRE(kmax <= 7 && (kmax & 1) && k >= 1 && (k & 1) && k <= kmax && C > 0, "noise_ofa_dw_filter: kernel sizes must be odd "
                "and at most 7 (got max %d, active %d)", kmax, k);
  NOISE_REQUIRE((!m1 || ((k1 & 1) && k1 >= 1 && k1 < kmax)) && (!m2 || (m1 && (k2 & 1) && k2 >= 1 && k2 < k1)),
                "noise_ofa_dw_filter: transform steps must shrink from %d (got %d, %d)", kmax, k1, k2);
  hipLaunchKernelGGL(ofa_dw_filter_kernel, dim3(noise_cdiv(C, 64)), dim3(256), 0, (hipStream_t)stream, weight, kmax, m1, k1, m2,
                     k2, k, C, flip, wt);
  NOISE_CHECK_LAUNCH("noise_ofa_dw_filter");
  return 0;
}

NOISE_EXPORT int noise_ofa_dw(const float* src_y, const float* src_coef, int src_relu6, const float* wt, float* out, float* slab,
                              int* rows, int B, int Hi, int Wi, int C, int k, int stride, int transposed, void* stream) {
  NOISE_REQUIRE(src_y && wt && out && slab && rows, "noise_ofa_dw: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C >= 4 && C % 4 == 0, "noise_ofa_dw: bad shape (C must be a multiple of 4)");
  NOISE_REQUIRE((k & 1) && k >= 1 && k <= 7 && (stride == 1 || stride == 2), "noise_ofa_dw: k must be odd and <= 7, stride 1 or 2");
  const int pad = k / 2;
  const int Ho = transposed ? (Hi - 1) * stride - 2 * pad + k : (Hi + 2 * pad - k) / stride + 1;
  const int Wo = transposed ? (Wi - 1) * stride - 2 * pad + k : (Wi + 2 * pad - k) / stride + 1;
  DwDims d{B, Hi, Wi, Ho, Wo, C, k, stride, channel_lanes(C)};
  const int gy = pixel_rows((long)B * Ho * Wo, 256 / d.cl * 2);
  const LazySrc src{src_y, src_coef, src_relu6};
  const dim3 grid(noise_cdiv(C / 4, d.cl), gy);
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1 && k >= 3) {
    // the filter must have been written flipped (noise_ofa_dw_filter, flip = 1) when `transposed`
    const dim3 g1(grid.x, pixel_rows((long)B * Ho * ((Wo + 3) / 4), 256 / d.cl));
    if (k == 3) hipLaunchKernelGGL(ofa_dw1_kernel<3>, g1, dim3(256), 0, st, src, wt, out, slab, d);
    else if (k == 5) hipLaunchKernelGGL(ofa_dw1_kernel<5>, g1, dim3(256), 0, st, src, wt, out, slab, d);
    else hipLaunchKernelGGL(ofa_dw1_kernel<7>, g1, dim3(256), 0, st, src, wt, out, slab, d);
    NOISE_CHECK_LAUNCH("noise_ofa_dw");
    *rows = (int)g1.y;
    return 0;
  }
  if (stride == 2 && k >= 3 && (long)B * Ho * ((Wo + 1) / 2) < (1L << 31)) {
    const dim3 g2(grid.x, pixel_rows((long)B * Ho * ((Wo + 1) / 2), 256 / d.cl));
#define DW2_CASE(KK)                                                                                                    \
  if (transposed) hipLaunchKernelGGL((ofa_dw2_kernel<KK, true>), g2, dim3(256), 0, st, src, wt, out, slab, d);             \
  else hipLaunchKernelGGL((ofa_dw2_kernel<KK, false>), g2, dim3(256), 0, st, src, wt, out, slab, d)
    if (k == 3) { DW2_CASE(3); }
    else if (k == 5) { DW2_CASE(5); }
    else { DW2_CASE(7); }
#undef DW2_CASE
    NOISE_CHECK_LAUNCH("noise_ofa_dw");
    *rows = (int)g2.y;
    return 0;
  }
  if (transposed)
    hipLaunchKernelGGL(ofa_dw_kernel<true>, grid, dim3(256), 0, st, src, wt, out, slab, d);
  else
    hipLaunchKernelGGL(ofa_dw_kernel<false>, grid, dim3(256), 0, st, src, wt, out, slab, d);
  NOISE_CHECK_LAUNCH("noise_ofa_dw");
  *rows = gy;
  return 0;
}

NOISE_EXPORT int noise_ofa_pw(const float* src_y, const float* src_coef, int src_relu6, const float* W, int ldw, float* out,
                              float* slab, int* rows, long npix, int K, int N, void* stream) {
  NOISE_REQUIRE(src_y && W && out && slab && rows, "noise_ofa_pw: null pointer");
  NOISE_REQUIRE(npix > 0 && K >= 4 && K % 4 == 0 && N >= 4 && N % 4 == 0 && ldw >= K && ldw % 4 == 0,
                "noise_ofa_pw: K, N and the weight row stride must be multiples of 4 (K %d, N %d, stride %d)", K, N, ldw);
  const LazySrc src{src_y, src_coef, src_relu6};
  const PwGDims d{npix, K, N, ldw};
  const long ntiles = (npix + 63) / 64;
  const int nt = pw_channel_tiles(N, ntiles);
  const int gx = noise_cdiv(N, nt * 16);
  int gy = pixel_rows(ntiles, 1);
  const int want = noise_cdiv(2048, gx);   // few enough pixel rows that gx x rows stays near two rounds of the GPU
  if (gy > want) gy = want;
  const LazyGrad none{nullptr, nullptr, nullptr};
#define PW_CASE(T)                                                                                                      \
  case T: {                                                                                                             \
    static int per_cu = 0;                                                                                              \
    gy = whole_round_rows(ofa_pw_kernel<T, false>, per_cu, gx, gy);                                                     \
    hipLaunchKernelGGL((ofa_pw_kernel<T, false>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, src, none, W, out, slab, d, \
                       src);                                                                                            \
    break;                                                                                                              \
  }
  switch (nt) { PW_CASE(1) PW_CASE(2) PW_CASE(4) PW_CASE(6) default: PW_CASE(8) }
#undef PW_CASE
  *rows = gy;
  NOISE_CHECK_LAUNCH("noise_ofa_pw");
  return 0;
}

// Input gradient of a 1x1 layer: g_in [npix][Cin] = dy [npix][Cout] x Wt[:Cin, :Cout]^T with Wt = the layer's weight
// transposed (row stride ldw) and dy the lazily evaluated gradient (dz, y_out, coef3); coef3 == NULL: dy = dz.
// Optional epilogue (epi_y != NULL): the layer's input is act(BatchNorm(epi_y)) with coefficients epi_coef
// ([scale|shift|mean|invstd|mean_lo], Cin each); g_in is then masked by that ReLU6 (epi_relu6) and slab receives *rows rows
// of [sum g | sum g * xhat] (2 Cin floats) for noise_bn_bwd_finalize.  slab: noise_ofa_max_rows() x 2 Cin floats.
NOISE_EXPORT int noise_gen_pw_dgrad(const float* dz, const float* y_out, const float* coef3, const float* Wt, int ldw,
                                    float* g_in, const float* epi_y, const float* epi_coef, int epi_relu6, float* slab,
                                    int* rows, long npix, int Cout, int Cin, void* stream) {
  NOISE_REQUIRE(dz && Wt && g_in && (coef3 == nullptr || y_out != nullptr), "noise_gen_pw_dgrad: null pointer");
  NOISE_REQUIRE(epi_y == nullptr || (epi_coef && slab && rows), "noise_gen_pw_dgrad: the epilogue needs coefficients and a slab");
  NOISE_REQUIRE(npix > 0 && Cout >= 4 && Cout % 4 == 0 && Cin >= 4 && Cin % 4 == 0 && ldw >= Cout && ldw % 4 == 0,
                "noise_gen_pw_dgrad: widths and the weight row stride must be multiples of 4 (Cout %d, Cin %d, stride %d)", Cout,
                Cin, ldw);
  const LazySrc none{nullptr, nullptr, 0};
  const LazySrc epi{epi_y, epi_coef, epi_relu6};
  const LazyGrad grad{dz, y_out, coef3};
  const PwGDims d{npix, Cout, Cin, ldw};
  const long ntiles = (npix + 63) / 64;
  const int nt = pw_channel_tiles(Cin, ntiles);
  const int gx = noise_cdiv(Cin, nt * 16);
  int gy = pixel_rows(ntiles, 1);
  const int want = noise_cdiv(2048, gx);
  if (gy > want) gy = want;
  float* st = epi_y ? slab : nullptr;
#define PW_CASE(T)                                                                                                      \
  case T: {                                                                                                             \
    static int per_cu = 0;                                                                                              \
    gy = whole_round_rows(ofa_pw_kernel<T, true>, per_cu, gx, gy);                                                      \
    hipLaunchKernelGGL((ofa_pw_kernel<T, true>), dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, none, grad, Wt, g_in, st, d, \
                       epi);                                                                                            \
    break;                                                                                                              \
  }
  switch (nt) { PW_CASE(1) PW_CASE(2) PW_CASE(4) PW_CASE(6) default: PW_CASE(8) }
#undef PW_CASE
  if (rows) *rows = gy;
  NOISE_CHECK_LAUNCH("noise_gen_pw_dgrad");
  return 0;
}

NOISE_EXPORT int noise_ofa_apply(const float* y, const float* coef, int relu6, int Hy, int Wy, const float* res,
                                 const float* skip, int Hs, int Ws, float* out, int B, int Ho, int Wo, int C, void* stream) {
  NOISE_REQUIRE(y && out, "noise_ofa_apply: null pointer");
  NOISE_REQUIRE(B > 0 && Ho > 0 && Wo > 0 && C >= 4 && C % 4 == 0, "noise_ofa_apply: bad shape (C must be a multiple of 4)");
  NOISE_REQUIRE(Hy >= Ho && Wy >= Wo, "noise_ofa_apply: the source (%d x %d) is smaller than the output (%d x %d)", Hy, Wy, Ho, Wo);
  NOISE_REQUIRE(!skip || (Hs >= Ho && Ws >= Wo), "noise_ofa_apply: the skip tensor (%d x %d) is smaller than the decoder tensor "
                "(%d x %d)", Hs, Ws, Ho, Wo);
  ApplyDims d{B, Ho, Wo, C, Hy, Wy, skip ? Hs : Ho, skip ? Ws : Wo, channel_lanes(C)};
  const int gy = noise_grid((long)B * Ho * Wo, 256 / d.cl * 4, 4096);
  const LazySrc src{y, coef, relu6};
  hipLaunchKernelGGL(ofa_apply_kernel, dim3(noise_cdiv(C / 4, d.cl), gy), dim3(256), 0, (hipStream_t)stream, src, res, skip, out, d);
  NOISE_CHECK_LAUNCH("noise_ofa_apply");
  return 0;
}

NOISE_EXPORT int noise_ofa_last_convT(const float* src_y, const float* src_coef, int src_relu6, const float* w, float* out,
                                      float* slab, int* rows, int B, int Hi, int Wi, int C, void* stream) {
  NOISE_REQUIRE(src_y && w && out && slab && rows, "noise_ofa_last_convT: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C >= 4 && C % 4 == 0 && C <= 256, "noise_ofa_last_convT: bad shape (C a multiple of 4, "
                "at most 256)");
  LastDims d{B, Hi, Wi, 2 * Hi - 1, 2 * Wi - 1, C, channel_lanes(C)};
  const int grid = pixel_rows((long)B * d.Ho * d.Wo, 256 / d.gl * 4);
  const LazySrc src{src_y, src_coef, src_relu6};
  hipLaunchKernelGGL(ofa_last_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, w, out, slab, d);
  NOISE_CHECK_LAUNCH("noise_ofa_last_convT");
  *rows = grid;
  return 0;
}

NOISE_EXPORT int noise_ofa_final(const float* y, const float* coef, int relu6, float* out, int B, int Hy, int Wy, int H, int W,
                                 void* stream) {
  NOISE_REQUIRE(y && coef && out, "noise_ofa_final: null pointer");
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && Hy >= H && Wy >= W, "noise_ofa_final: the network output (%d x %d) is smaller than the "
                "input (%d x %d)", Hy, Wy, H, W);
  hipLaunchKernelGGL(ofa_final_kernel, dim3(noise_grid((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, y, coef, out,
                     B, Hy, Wy, H, W, relu6);
  NOISE_CHECK_LAUNCH("noise_ofa_final");
  return 0;
}
