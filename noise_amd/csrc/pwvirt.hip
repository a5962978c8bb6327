// Pointwise (1x1) convolution whose input is never materialised: the projection of a transposed inverted-residual block
// (train/models.py:299-345, :385-396) reads the *input* of the block's stride-2 depthwise ConvTranspose2d and rebuilds
// the depthwise output on the fly, so the 4x larger full-resolution hidden tensor is neither written nor read.
//
//   a_in[i]   = act_in(scale_in * y_in[i] + shift_in)                      (lazy BatchNorm + ReLU6 of the expand layer)
//   y_dw[o]   = sum_taps wdw[k] * a_in[i(o, k)]                            (3x3, stride 2, padding 1: output 2n - 1)
//   a_dw[o]   = act_dw(scale_dw * y_dw[o] + shift_dw)                      (lazy BatchNorm + ReLU6 of the depthwise layer)
//   out[o][co] = sum_ci a_dw[o][ci] * W[co][ci]
//
// Output pixel (2m + py, 2x + px) takes source cells (m, x), (m, x + px), (m + py, x), (m + py, x + px) with the taps
// and in the summation order of the stand-alone depthwise kernel (walk_cell_fwd_kernel), so both agree bit for bit.
// MFMA arrangement, weight fragments, prefetch and statistics epilogue are those of pw_gemm_kernel (pw.hip).
#include <type_traits>

#include "lazy.h"

extern "C" int noise_pw_bwd_max_rows(int Cout, int Cin);   // pw.hip

namespace {

struct VirtDims {
  long npix;            // B * Ho * Wo output pixels
  int K, N;             // hidden channels (multiple of 8), output channels (multiple of 4)
  int Hi, Wi, Ho, Wo;   // source and output image sizes, Ho = 2 Hi - 1, Wo = 2 Wi - 1
  long ncells;          // B * Hi * Wi source pixels
};

// Tiles of 16 consecutive source cells.  The position of a tile's first cell is kept as (image, row, column) and
// advanced by the grid stride without a division (uniform values).  A tile past the end keeps the position of the last
// one it passed: its loads stay in bounds and its cells do not "exist".
struct TilePos {
  long tile;
  int b, m, x;
};
struct CellTiles {
  long ncells, ntiles;
  int Hi, Wi;
  int sb, sm, sx;   // 16 * gridDim.x cells as (images, rows, columns)
  long tstep;        // tiles between two consecutive tiles of one walker
  __device__ __forceinline__ CellTiles(const VirtDims& d, int tiles_per_block = 1) {
    ncells = d.ncells;
    ntiles = (ncells + 15) >> 4;
    Hi = d.Hi;
    Wi = d.Wi;
    tstep = (long)tiles_per_block * gridDim.x;
    const unsigned per_src = (unsigned)Hi * (unsigned)Wi, step = 16u * (unsigned)tstep;
    sb = (int)(step / per_src);
    const unsigned rem = step - (unsigned)sb * per_src;
    sm = (int)(rem / (unsigned)Wi);
    sx = (int)(rem - (unsigned)sm * (unsigned)Wi);
  }
  __device__ __forceinline__ TilePos pos(long tile) const {
    const long tl = tile < ntiles ? tile : ntiles - 1;
    const unsigned c0 = (unsigned)(tl * 16), per_src = (unsigned)Hi * (unsigned)Wi;
    TilePos t;
    t.tile = tile;
    t.b = (int)(c0 / per_src);
    const unsigned rem = c0 - (unsigned)t.b * per_src;
    t.m = (int)(rem / (unsigned)Wi);
    t.x = (int)(rem - (unsigned)t.m * (unsigned)Wi);
    return t;
  }
  __device__ __forceinline__ void advance(TilePos& t) const {
    t.tile += tstep;
    if (t.tile >= ntiles) return;
    t.x += sx;
    if (t.x >= Wi) {
      t.x -= Wi;
      ++t.m;
    }
    t.m += sm;
    if (t.m >= Hi) {
      t.m -= Hi;
      ++t.b;
    }
    t.b += sb;
  }
  // cell i of a tile (clamped to the last cell): image, row, column; returns whether the cell exists
  __device__ __forceinline__ bool cell(const TilePos& t, int i, int& b, int& m, int& x) const {
    const long tl = t.tile < ntiles ? t.tile : ntiles - 1;
    const bool exists = t.tile < ntiles && tl * 16 + i < ncells;
    const int last = (int)(ncells - 1 - tl * 16);
    b = t.b;
    m = t.m;
    x = t.x + (i < last ? i : last);
    while (x >= Wi) {
      x -= Wi;
      ++m;
    }
    while (m >= Hi) {
      m -= Hi;
      ++b;
    }
    return exists;
  }
};

// Forward.  Each wave of a workgroup walks its own tiles of 16 consecutive source cells; a tile produces 64 output
// pixels, the four parity classes (py, px) of every cell.  Lane (p, q) builds the MFMA operands of the four output pixels
// (2m + py, 2x + px) of cell p directly in registers (channels 16 jj + 4 q .. + 3), so nothing is staged through LDS and
// the 2x2 source window, its activation, the lazy coefficients and the weight fragments serve four pixels.  (One wave
// per parity class, the first form of this kernel, read the coefficient and tap tables from LDS once per output pixel
// and was bound by the LDS pipe: 0.27 ms for the 72 -> 16 layer of 1280 segments.)  The K channels are processed in KH
// chunks of KCH 16-channel tiles; the registers of a consumed channel tile are refilled at once with the raw source cells
// of the next chunk or tile.
template <int KCH, int KH, int MT, int WGS>
__global__ __launch_bounds__(256, WGS) void pw_virt_fwd_kernel(LazySrc In, const float* __restrict__ wdw, LazySrc Dw,
                                                             const float* __restrict__ W, float* __restrict__ out,
                                                             float* __restrict__ slab, VirtDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int K = d.K, N = d.N;
  constexpr int KC = KCH * KH, KP = KC * 16;
  float* wfrag0 = lds;                    // [MT][KC][64][4]
  float* cf0 = wfrag0 + MT * KC * 256;     // [4][KP]: scale_in, shift_in, scale_dw, shift_dw
  float* wt0 = cf0 + 4 * KP;               // [9][KP]: the taps, tap-major
  if ((N & 15) || (K & 15)) {
    for (int idx = tid; idx < MT * KC * 256; idx += 256) wfrag0[idx] = 0.f;
    __syncthreads();
  }
  {
    const int K4 = K >> 2;
#pragma unroll 4
    for (int e = tid; e < N * K4; e += 256) {
      const int co = e / K4, ci = 4 * (e - co * K4);
      const float4 v = ld4(W + (size_t)co * K + ci);
      st4(wfrag0 + ((((co >> 4) * KC + (ci >> 4)) * 64) + (co & 15) + 16 * ((ci & 15) >> 2)) * 4, v);
    }
  }
  for (int k = tid; k < KP; k += 256) {
    const bool in = k < K;
    cf0[k] = (in && In.coef) ? In.coef[k] : 1.f;
    cf0[KP + k] = (in && In.coef) ? In.coef[K + k] : 0.f;
    cf0[2 * KP + k] = (in && Dw.coef) ? Dw.coef[k] : 1.f;
    cf0[3 * KP + k] = (in && Dw.coef) ? Dw.coef[K + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wt0[t * KP + k] = in ? wdw[(size_t)k * 9 + t] : 0.f;    // depthwise weight (C, 1, 3, 3)
  }
  __syncthreads();

  float4 s1[MT], s2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) s1[m] = s2[m] = f4(0.f);
  const float ilo = In.relu6 ? 0.f : -INFINITY, ihi = In.relu6 ? 6.f : INFINITY;
  const float dlo = Dw.relu6 ? 0.f : -INFINITY, dhi = Dw.relu6 ? 6.f : INFINITY;
  const CellTiles ct(d, 4);
  const long iters = (ct.ntiles + 4L * gridDim.x - 1) / (4L * gridDim.x);

  // Source cells of a parity class in the order the stand-alone kernel sums them, with their taps:
  //   (0,0): (m,x) w4      (0,1): (m,x) w5, (m,x+1) w3      (1,0): (m,x) w7, (m+1,x) w1
  //   (1,1): (m,x) w8, (m,x+1) w6, (m+1,x) w2, (m+1,x+1) w0
  // A wave owns a whole tile: lane (p, q) loads the 2x2 source window of cell p once (channels 16 jj + 4 q .. + 3),
  // activates it once and builds its four output pixels from it, so the window, the lazy coefficients and the weight
  // fragments are fetched once per four output pixels.  The channel tiles are walked one at a time; the registers of a
  // consumed channel tile are refilled at once with the same tile of the next chunk (or of the wave's next tile).
  float4 r[KCH][4];
  auto request = [&](const TilePos& t, int h, int jj) {
    int b, m, x;
    ct.cell(t, p, b, m, x);
    const float* src = In.y + (((size_t)b * d.Hi + m) * d.Wi + x) * K;
    const size_t no1 = (x + 1 < d.Wi) ? (size_t)K : 0, no2 = (m + 1 < d.Hi) ? (size_t)d.Wi * K : 0;
    const int ci = 16 * (h * KCH + jj) + 4 * q, cc = ci < K ? ci : 0;
    r[jj][0] = ld4(src + cc);
    r[jj][1] = ld4(src + no1 + cc);
    r[jj][2] = ld4(src + no2 + cc);
    r[jj][3] = ld4(src + no1 + no2 + cc);
  };
  // with few channel tiles the four lazy coefficient vectors of a lane's channels stay in registers
  constexpr bool CF_REGS = KC <= 5 && WGS <= 2;
  float4 c_sc[CF_REGS ? KC : 1], c_sh[CF_REGS ? KC : 1], c_dsc[CF_REGS ? KC : 1], c_dsh[CF_REGS ? KC : 1];
  if (CF_REGS) {
#pragma unroll
    for (int j = 0; j < (CF_REGS ? KC : 0); ++j) {
      const int ci = 16 * j + 4 * q;
      c_sc[j] = ld4(cf0 + ci);
      c_sh[j] = ld4(cf0 + KP + ci);
      c_dsc[j] = ld4(cf0 + 2 * KP + ci);
      c_dsh[j] = ld4(cf0 + 3 * KP + ci);
    }
  }
  TilePos cur = ct.pos((long)blockIdx.x * 4 + wave), nxt = cur;
#pragma unroll
  for (int jj = 0; jj < KCH; ++jj) request(cur, 0, jj);
  for (long it = 0; it < iters; ++it) {
    int b, m, x;
    const bool ex = ct.cell(cur, p, b, m, x);
    const bool vy1 = m + 1 < d.Hi, vx1 = x + 1 < d.Wi;
    const bool valid[4] = {ex, ex && vx1, ex && vy1, ex && vy1 && vx1};
    f32x4 acc[4][MT];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[cl][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    ct.advance(nxt);
    // The tables are read from LDS at every use: a lane's channels are the same for every tile, so the compiler would
    // otherwise keep all of them (15 vectors per channel tile) in registers across the tile loop and spill.  An offset it
    // cannot see through keeps the reads where they are.
    int lofs = 0;
    asm volatile("" : "+v"(lofs));
    const float* cf = cf0 + lofs;
    const float* wt = wt0 + lofs;
    const float* wfrag = wfrag0 + lofs;
#pragma unroll
    for (int h = 0; h < KH; ++h) {
#pragma unroll
      for (int jj = 0; jj < KCH; ++jj) {
        const int ci = 16 * (h * KCH + jj) + 4 * q;
        const float4 sc = CF_REGS ? c_sc[CF_REGS ? h * KCH + jj : 0] : ld4(cf + ci);
        const float4 sh = CF_REGS ? c_sh[CF_REGS ? h * KCH + jj : 0] : ld4(cf + KP + ci);
        const float4 s00 = clamp4(fma4(r[jj][0], sc, sh), ilo, ihi), s01 = clamp4(fma4(r[jj][1], sc, sh), ilo, ihi);
        const float4 s10 = clamp4(fma4(r[jj][2], sc, sh), ilo, ihi), s11 = clamp4(fma4(r[jj][3], sc, sh), ilo, ihi);
        if (h + 1 < KH) request(cur, h + 1, jj);
        else request(nxt, 0, jj);
        float4 y[4];
        y[0] = s00 * ld4(wt + 4 * KP + ci);
        y[1] = fma4(s01, ld4(wt + 3 * KP + ci), s00 * ld4(wt + 5 * KP + ci));
        y[2] = fma4(s10, ld4(wt + 1 * KP + ci), s00 * ld4(wt + 7 * KP + ci));
        y[3] = s00 * ld4(wt + 8 * KP + ci);
        y[3] = fma4(s01, ld4(wt + 6 * KP + ci), y[3]);
        y[3] = fma4(s10, ld4(wt + 2 * KP + ci), y[3]);
        y[3] = fma4(s11, ld4(wt + ci), y[3]);
        const float4 dsc = CF_REGS ? c_dsc[CF_REGS ? h * KCH + jj : 0] : ld4(cf + 2 * KP + ci);
        const float4 dsh = CF_REGS ? c_dsh[CF_REGS ? h * KCH + jj : 0] : ld4(cf + 3 * KP + ci);
        float4 in[4];
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
          const float4 a = clamp4(fma4(y[cl], dsc, dsh), dlo, dhi);
          const bool okc = valid[cl] && ci < K;
          in[cl] = make_float4(okc ? a.x : 0.f, okc ? a.y : 0.f, okc ? a.z : 0.f, okc ? a.w : 0.f);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4 wa = ld4(wfrag + ((mt * KC + h * KCH + jj) * 64 + lane) * 4);
#pragma unroll
          for (int cl = 0; cl < 4; ++cl) {
            acc[cl][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, in[cl].x, acc[cl][mt], 0, 0, 0);
            acc[cl][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, in[cl].y, acc[cl][mt], 0, 0, 0);
            acc[cl][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, in[cl].z, acc[cl][mt], 0, 0, 0);
            acc[cl][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, in[cl].w, acc[cl][mt], 0, 0, 0);
          }
        }
      }
    }
    cur = nxt;
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) {
      const size_t pix = ((size_t)b * d.Ho + 2 * m + (cl >> 1)) * d.Wo + 2 * x + (cl & 1);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int co = 16 * mt + 4 * q;
        const float4 o = make_float4(acc[cl][mt][0], acc[cl][mt][1], acc[cl][mt][2], acc[cl][mt][3]);
        if (valid[cl] && co < N) {
          st4(out + pix * N + co, o);
          s1[mt] = s1[mt] + o;
          s2[mt] = fma4(o, o, s2[mt]);
        }
      }
    }
  }

  if (slab != nullptr) {
    __syncthreads();
    float* red = lds;  // reuse: [4 waves][2][MT*16]
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float v[8] = {s1[m].x, s1[m].y, s1[m].z, s1[m].w, s2[m].x, s2[m].y, s2[m].z, s2[m].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v[e] += __shfl_xor(v[e], o, 64);
      }
      if (p == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * MT * 16 + 16 * m + 4 * q + e] = v[e];
          red[(wave * 2 + 1) * MT * 16 + 16 * m + 4 * q + e] = v[4 + e];
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < 2 * N; i += 256) {
      const int j = i / N, c = i - j * N;
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) sum += red[(wv * 2 + j) * MT * 16 + c];
      slab[(size_t)blockIdx.x * 2 * N + i] = sum;
    }
  }
}

template <int KCH, int KH, int MT, int WGS>
int launch_virt(LazySrc In, const float* wdw, LazySrc Dw, const float* W, float* out, float* slab, VirtDims d, int& grid,
                hipStream_t st) {
  constexpr int KC = KCH * KH;
  const size_t smem = (size_t)(MT * KC * 256 + 13 * KC * 16) * sizeof(float);
  auto kfn = pw_virt_fwd_kernel<KCH, KH, MT, WGS>;
  NOISE_ALLOW_LDS(kfn, smem);
  grid = noise_whole_rounds(grid, kfn, 256, smem);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), smem, st, In, wdw, Dw, W, out, slab, d);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Backward of the same projection (fused input gradient + weight gradient + the depthwise BatchNorm's backward sums,
// see pw_bwd_kernel in pw.hip), again without the depthwise output in memory.  A tile is 16 consecutive source cells
// (b, m, x) = 64 output pixels (2m + py, 2x + px): wave w owns parity class (py, px) = (w >> 1, w & 1), so one source
// window (m..m+1, x..x+1) serves four output pixels.  Per tile:
//   1. dy of the 64 pixels (lazy BatchNorm backward of the projection's own BatchNorm) -> LDS;
//      the raw depthwise output and its activation, rebuilt from the source window -> LDS (two tiles);
//   2. input gradient g = dy W per wave (MFMA), ReLU6 mask and the producer's sums from the raw tile, store;
//      weight gradient dW += dy^T a over the 64 pixels from the activated tile.
// The next tile's global loads are requested right after the staging barrier and stay in registers through the MFMAs.
constexpr int virt_bwd_pitch(int tiles) { return tiles * 16 + 4; }
// LEAN (the 72 -> 16 instance, more than half of the family's time): three workgroups per CU instead of two.  The
// activated tile is not kept in LDS (it is one multiply-add and a clamp away from the raw tile) and the per-lane
// coefficient vectors of the epilogue stay in LDS instead of registers: the kernel lives on memory parallelism, and a
// third resident workgroup is worth more than the LDS reads it costs.
constexpr bool virt_bwd_lean(int KC, int MTI) { return KC == 1 && MTI == 5; }
constexpr int virt_bwd_lds_floats(int KC, int MTI) {
  return MTI * KC * 256 + 64 * virt_bwd_pitch(KC) + (virt_bwd_lean(KC, MTI) ? 1 : 2) * 64 * virt_bwd_pitch(MTI) + 17 * MTI * 16 +
         (virt_bwd_lean(KC, MTI) ? 6 * KC * 16 : 0);
}
constexpr int virt_bwd_min_blocks(int KC, int MTI) {
  return virt_bwd_lean(KC, MTI) ? 3 : (virt_bwd_lds_floats(KC, MTI) * 8 <= 160 * 1024 ? 2 : 1);
}

template <int KC, int MTI>
__global__ __launch_bounds__(256, virt_bwd_min_blocks(KC, MTI)) void pw_virt_bwd_kernel(
    LazyGrad G, LazySrc In, const float* __restrict__ wdw, const float* __restrict__ dwcoef, int dw_relu6,
    const float* __restrict__ W, float* __restrict__ g_in, float* __restrict__ stats_slab, float* __restrict__ dw_slab,
    VirtDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int K = d.K, N = d.N;
  constexpr int NP = KC * 16, KP = MTI * 16;
  constexpr int pn = virt_bwd_pitch(KC), pk = virt_bwd_pitch(MTI);
  float* wfrag = lds;                       // [MTI][KC][64][4] fragment image of W for the input gradient
  float* dys = wfrag + MTI * KC * 256;      // [64][pn] dy, row 16 * parity + cell
  constexpr bool LEAN = virt_bwd_lean(KC, MTI);
  float* as = dys + 64 * pn;                // [64][pk] activated depthwise output (not kept when LEAN)
  float* raws = LEAN ? as : as + 64 * pk;   // [64][pk] raw depthwise output
  float* cf = raws + 64 * pk;               // [7][KP]: source scale, shift; depthwise scale, shift, mean, invstd, mean_lo
  float* wt = cf + 7 * KP;                  // [10][KP]: the 9 taps, row 9 unused
  float* gcf = wt + 10 * KP;                // [6][NP] (LEAN): the lazy-gradient coefficients of the projection's BatchNorm
  if ((N & 15) || (K & 15)) {
    for (int idx = tid; idx < MTI * KC * 256; idx += 256) wfrag[idx] = 0.f;
    __syncthreads();
  }
  {
    const int K4 = K >> 2;
#pragma unroll 4
    for (int e = tid; e < N * K4; e += 256) {
      const int co = e / K4, ci = 4 * (e - co * K4);
      const float4 v = ld4(W + (size_t)co * K + ci);
      float* dst = wfrag + ((((ci >> 4) * KC + (co >> 4)) * 64) + (ci & 15) + 16 * ((co & 15) >> 2)) * 4 + (co & 3);
      dst[0] = v.x;
      dst[4] = v.y;
      dst[8] = v.z;
      dst[12] = v.w;
    }
  }
  for (int k = tid; k < KP; k += 256) {
    const bool in = k < K;
    cf[k] = (in && In.coef) ? In.coef[k] : 1.f;
    cf[KP + k] = (in && In.coef) ? In.coef[K + k] : 0.f;
    cf[2 * KP + k] = in ? dwcoef[k] : 1.f;
    cf[3 * KP + k] = in ? dwcoef[K + k] : 0.f;
    cf[4 * KP + k] = in ? dwcoef[2 * K + k] : 0.f;
    cf[5 * KP + k] = in ? dwcoef[3 * K + k] : 1.f;
    cf[6 * KP + k] = in ? dwcoef[4 * K + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t * KP + k] = in ? wdw[(size_t)k * 9 + t] : 0.f;
  }

  constexpr int T = KC * MTI, TPW = (T + 3) / 4;
  f32x4 acc[TPW];
  int dyo[TPW], avo[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int t = wave + 4 * j < T ? wave + 4 * j : wave % T, m = t / MTI, n = t - m * MTI;
    dyo[j] = 16 * m + p;
    avo[j] = 16 * n + p;
  }
  float4 s1[MTI], s2[MTI];
#pragma unroll
  for (int m = 0; m < MTI; ++m) s1[m] = s2[m] = f4(0.f);

  // dy staging map: channel group fastest, then the cell, then the parity class (KC classes per pass)
  constexpr int CGN = NP / 4, JP = 16 / CGN, PASS_N = 4 / JP;
  const int cn = (tid % CGN) * 4, in_cell = (tid / CGN) & 15, jn0 = tid / (CGN * 16);
  const bool on = cn < N;
  float4 gs = f4(1.f), gk1h = f4(0.f), gk1l = f4(0.f), ga2 = f4(0.f), gmh = f4(0.f), gml = f4(0.f);
  if (LEAN) {      // the six vectors go to LDS (read back per tile) instead of 24 registers
    for (int k = tid; k < 6 * NP; k += 256) {
      const int r = k / NP, c = k - r * NP;
      gcf[k] = (c < N && G.coef3) ? G.coef3[r * N + c] : (r == 0 ? 1.f : 0.f);
    }
  } else if (on && G.coef3) {
    gs = ld4(G.coef3 + cn);
    gk1h = ld4(G.coef3 + N + cn);
    gk1l = ld4(G.coef3 + 2 * N + cn);
    ga2 = ld4(G.coef3 + 3 * N + cn);
    gmh = ld4(G.coef3 + 4 * N + cn);
    gml = ld4(G.coef3 + 5 * N + cn);
  }
  // source staging map: item = (cell, channel group), channel group fastest
  constexpr int CGK = KP / 4, ITEMS = 16 * CGK, PASS_V = (ITEMS + 255) / 256;
  const float ilo = In.relu6 ? 0.f : -INFINITY, ihi = In.relu6 ? 6.f : INFINITY;
  const float dlo = dw_relu6 ? 0.f : -INFINITY, dhi = dw_relu6 ? 6.f : INFINITY;

  const long ntiles = (d.ncells + 15) >> 4;
  const long iters = (ntiles + gridDim.x - 1) / gridDim.x;
  const CellTiles ct(d);

  float4 rz[PASS_N], ry[PASS_N], r[PASS_V][4];
  int nflag = 0;
  auto issue_tile = [&](const TilePos& tile) {
    {
      int b, m, x;
      const bool ex = ct.cell(tile, in_cell, b, m, x);
      const bool vy1 = m + 1 < d.Hi, vx1 = x + 1 < d.Wi;
      nflag = 0;
#pragma unroll
      for (int u = 0; u < PASS_N; ++u) {
        const int j = jn0 + JP * u, py = j >> 1, px = j & 1;
        const bool ok = ex && (!py || vy1) && (!px || vx1);
        const int oy = 2 * m + ((py && vy1) ? 1 : 0), ox = 2 * x + ((px && vx1) ? 1 : 0);
        const size_t off = (((size_t)b * d.Ho + oy) * d.Wo + ox) * N + (on ? cn : 0);
        rz[u] = ld4(G.dz + off);
        ry[u] = G.coef3 ? ld4(G.y + off) : f4(0.f);
        nflag |= ok ? (1 << u) : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < PASS_V; ++u) {
      const int idx = tid + 256 * u;
      const int ci = idx < ITEMS ? idx / CGK : 15, cg = idx < ITEMS ? idx - ci * CGK : 0;
      int b, m, x;
      ct.cell(tile, ci, b, m, x);
      const bool vy1 = m + 1 < d.Hi, vx1 = x + 1 < d.Wi;
      const int ck = 4 * cg;
      const float* src = In.y + (((size_t)b * d.Hi + m) * d.Wi + x) * K + (ck < K ? ck : 0);
      const size_t o1 = vx1 ? (size_t)K : 0, o2 = vy1 ? (size_t)d.Wi * K : 0;
      r[u][0] = ld4(src);
      r[u][1] = ld4(src + o1);
      r[u][2] = ld4(src + o2);
      r[u][3] = ld4(src + o1 + o2);
    }
  };
  // The input-gradient epilogue of a lane always works on the same channels (16 m2 + 4 q .. + 3): with few channel tiles
  // its three coefficient vectors per tile stay in registers instead of being read from LDS for every pixel (the LDS
  // pipe, not memory, bounds this kernel).
  constexpr bool EPI_REGS = MTI <= 5 && !LEAN;
  float4 e_mean[EPI_REGS ? MTI : 1], e_mlo[EPI_REGS ? MTI : 1], e_istd[EPI_REGS ? MTI : 1];
  __syncthreads();   // the tables are written
  float wsc[TPW], wsh[TPW];      // LEAN: depthwise BatchNorm scale / shift of the channel this lane feeds to each dW tile
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    wsc[j] = LEAN ? cf[2 * KP + avo[j]] : 1.f;
    wsh[j] = LEAN ? cf[3 * KP + avo[j]] : 0.f;
  }
  constexpr bool WF_REGS = EPI_REGS && KC == 1;   // the lane's weight fragments of the input gradient as well
  float4 wf[WF_REGS ? MTI : 1];
  if (EPI_REGS) {
#pragma unroll
    for (int m2 = 0; m2 < (EPI_REGS ? MTI : 0); ++m2) {
      const int ci = 16 * m2 + 4 * q;
      e_mean[m2] = ld4(cf + 4 * KP + ci);
      e_istd[m2] = ld4(cf + 5 * KP + ci);
      e_mlo[m2] = ld4(cf + 6 * KP + ci);
      if (WF_REGS) wf[WF_REGS ? m2 : 0] = ld4(wfrag + (m2 * KC * 64 + lane) * 4);
    }
  }
  TilePos cur = ct.pos(blockIdx.x), nxt = cur;
  issue_tile(cur);
  for (long it = 0; it < iters; ++it) {
    __syncthreads();   // the previous tile's MFMAs have read the staged tiles
#pragma unroll
    for (int u = 0; u < PASS_N; ++u) {
      const int pp = 16 * (jn0 + JP * u) + in_cell;
      float4 v = rz[u];
      if (LEAN) {
        if (G.coef3)
          v = fma4(ld4(gcf + 3 * NP + cn), sub4(sub4(ry[u], ld4(gcf + 4 * NP + cn)), ld4(gcf + 5 * NP + cn)),
                   ld4(gcf + cn) * sub4(sub4(v, ld4(gcf + NP + cn)), ld4(gcf + 2 * NP + cn)));
      } else if (G.coef3) v = fma4(ga2, sub4(sub4(ry[u], gmh), gml), gs * sub4(sub4(v, gk1h), gk1l));
      st4(dys + pp * pn + cn, (on && ((nflag >> u) & 1)) ? v : f4(0.f));
    }
#pragma unroll
    for (int u = 0; u < PASS_V; ++u) {
      const int idx = tid + 256 * u;
      if (idx < ITEMS) {
        const int ci = idx / CGK, ck = 4 * (idx - ci * CGK);
        const float4 sc = ld4(cf + ck), sh = ld4(cf + KP + ck);
        const float4 s00 = clamp4(fma4(r[u][0], sc, sh), ilo, ihi), s01 = clamp4(fma4(r[u][1], sc, sh), ilo, ihi);
        const float4 s10 = clamp4(fma4(r[u][2], sc, sh), ilo, ihi), s11 = clamp4(fma4(r[u][3], sc, sh), ilo, ihi);
        float4 y[4];
        y[0] = s00 * ld4(wt + 4 * KP + ck);
        y[1] = fma4(s01, ld4(wt + 3 * KP + ck), s00 * ld4(wt + 5 * KP + ck));
        y[2] = fma4(s10, ld4(wt + 1 * KP + ck), s00 * ld4(wt + 7 * KP + ck));
        y[3] = s00 * ld4(wt + 8 * KP + ck);
        y[3] = fma4(s01, ld4(wt + 6 * KP + ck), y[3]);
        y[3] = fma4(s10, ld4(wt + 2 * KP + ck), y[3]);
        y[3] = fma4(s11, ld4(wt + ck), y[3]);
        const float4 dsc = ld4(cf + 2 * KP + ck), dsh = ld4(cf + 3 * KP + ck);
        // rows of pixels that do not exist keep whatever the clamped loads gave (finite): their dy rows are zero, so
        // the weight gradient ignores them, and the input-gradient epilogue skips them
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st4(raws + (16 * j + ci) * pk + ck, y[j]);
          if (!LEAN) st4(as + (16 * j + ci) * pk + ck, clamp4(fma4(y[j], dsc, dsh), dlo, dhi));
        }
      }
    }
    __syncthreads();
    ct.advance(nxt);
    issue_tile(nxt);
    // ---- input gradient of this wave's parity class
    {
      int b, m, x;
      const bool ex = ct.cell(cur, p, b, m, x);
      cur = nxt;
      const int py = wave >> 1, px = wave & 1;
      const bool valid = ex && (!py || m + 1 < d.Hi) && (!px || x + 1 < d.Wi);
      const size_t prow = (((size_t)b * d.Ho + 2 * m + py) * d.Wo + 2 * x + px) * K;
      float4 in[KC];
#pragma unroll
      for (int jj = 0; jj < KC; ++jj) in[jj] = ld4(dys + (16 * wave + p) * pn + 16 * jj + 4 * q);
#pragma unroll
      for (int m2 = 0; m2 < MTI; ++m2) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < KC; ++jj) {
          const float4 wa = WF_REGS ? wf[WF_REGS ? m2 : 0] : ld4(wfrag + ((m2 * KC + jj) * 64 + lane) * 4);
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, in[jj].x, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, in[jj].y, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, in[jj].z, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, in[jj].w, o, 0, 0, 0);
        }
        const int ci = 16 * m2 + 4 * q;
        if (valid && ci < K) {
          float4 g = make_float4(o[0], o[1], o[2], o[3]);
          const float4 yraw = ld4(raws + (16 * wave + p) * pk + ci);
          if (dw_relu6) {   // the clamped activation is strictly inside (0, 6) exactly where the ReLU6 passes gradient
            const float4 v = LEAN ? clamp4(fma4(yraw, ld4(cf + 2 * KP + ci), ld4(cf + 3 * KP + ci)), dlo, dhi)
                                  : ld4(as + (16 * wave + p) * pk + ci);
            g.x = (v.x > 0.f && v.x < 6.f) ? g.x : 0.f;
            g.y = (v.y > 0.f && v.y < 6.f) ? g.y : 0.f;
            g.z = (v.z > 0.f && v.z < 6.f) ? g.z : 0.f;
            g.w = (v.w > 0.f && v.w < 6.f) ? g.w : 0.f;
          }
          const float4 xh = EPI_REGS ? sub4(sub4(yraw, e_mean[EPI_REGS ? m2 : 0]), e_mlo[EPI_REGS ? m2 : 0]) * e_istd[EPI_REGS ? m2 : 0]
                                     : sub4(sub4(yraw, ld4(cf + 4 * KP + ci)), ld4(cf + 6 * KP + ci)) * ld4(cf + 5 * KP + ci);
          s1[m2] = s1[m2] + g;
          s2[m2] = fma4(g, xh, s2[m2]);
          st4(g_in + prow + ci, g);
        }
      }
    }
    // ---- weight gradient: 16 k-steps of 4 pixels over this wave's tiles of dW
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
      const int pp = 4 * s + q;
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const float av = LEAN ? __builtin_amdgcn_fmed3f(fmaf(raws[pp * pk + avo[j]], wsc[j], wsh[j]), dlo, dhi) : as[pp * pk + avo[j]];
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dys[pp * pn + dyo[j]], av, acc[j], 0, 0, 0);
      }
    }
  }

  // ---- the depthwise BatchNorm's backward sums: reduce over the 16 pixel lanes, then over the waves
  __syncthreads();
  {
    float* red = lds;   // [4 waves][2][KP]
#pragma unroll
    for (int m = 0; m < MTI; ++m) {
      float v[8] = {s1[m].x, s1[m].y, s1[m].z, s1[m].w, s2[m].x, s2[m].y, s2[m].z, s2[m].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v[e] += __shfl_xor(v[e], o, 64);
      }
      if (p == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * KP + 16 * m + 4 * q + e] = v[e];
          red[(wave * 2 + 1) * KP + 16 * m + 4 * q + e] = v[4 + e];
        }
      }
    }
    __syncthreads();
    for (int i2 = tid; i2 < 2 * K; i2 += 256) {
      const int j = i2 / K, c = i2 - j * K;
      float sum = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) sum += red[(wv * 2 + j) * KP + c];
      stats_slab[(size_t)blockIdx.x * 2 * K + i2] = sum;
    }
    __syncthreads();
  }
  // ---- dW: every tile has one owning wave; the block writes its slab row with coalesced stores
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = wave + 4 * j;
    if (t < T) {
      const int m = t / MTI, n = t - m * MTI;
      const int ci = 16 * n + p;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = 16 * m + 4 * q + e;
        if (co < N && ci < K) lds[co * K + ci] = acc[j][e];
      }
    }
  }
  __syncthreads();
  for (int j = tid; j < N * K; j += 256) dw_slab[(size_t)blockIdx.x * N * K + j] = lds[j];
}

template <int KC, int MTI>
int launch_virt_bwd(LazyGrad G, LazySrc In, const float* wdw, const float* dwcoef, int dw_relu6, const float* W,
                    float* g_in, float* stats_slab, float* dw_slab, VirtDims d, int& grid, hipStream_t st) {
  const size_t smem = (size_t)virt_bwd_lds_floats(KC, MTI) * sizeof(float);
  auto kfn = pw_virt_bwd_kernel<KC, MTI>;
  NOISE_ALLOW_LDS(kfn, smem);
  grid = noise_whole_rounds(grid, kfn, 256, smem);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), smem, st, G, In, wdw, dwcoef, dw_relu6, W, g_in, stats_slab, dw_slab, d);
  return 0;
}

}  // namespace

NOISE_EXPORT int noise_pw_fwd_virt_supported(int K, int N) {
  const int KC = (K + 15) / 16, MT = (N + 15) / 16;
  return (KC == 5 && MT == 1) || (KC == 6 && MT == 2) || (KC == 12 && MT == 2);
}

// out (B, 2Hi-1, 2Wi-1, N) raw = act_dw(bn_dw(dwT3x3s2(act_in(bn_in(in))))) x W^T with in (B, Hi, Wi, K), wdw (K, 1, 3, 3),
// W (N, K); in_coef / dw_coef: [scale(K) | shift(K) | ...] of the two BatchNorms (NULL: identity).  stats_slab
// (optional): per-workgroup [sum(N) | sumsq(N)] rows of the output; *slab_rows receives the row count (<= 1024).
NOISE_EXPORT int noise_pw_fwd_virt(const float* in_y, const float* in_coef, int in_relu6, const float* wdw,
                                   const float* dw_coef, int dw_relu6, const float* w, float* out, float* stats_slab,
                                   int* slab_rows, int B, int Hi, int Wi, int K, int N, void* stream) {
  NOISE_REQUIRE(in_y && wdw && w && out, "noise_pw_fwd_virt: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_pw_fwd_virt: empty input");
  NOISE_REQUIRE(K % 8 == 0 && N % 4 == 0 && noise_pw_fwd_virt_supported(K, N),
                "noise_pw_fwd_virt: no kernel instance for K=%d N=%d (served: 72->16, 96->24, 192->32)", K, N);
  const int Ho = 2 * Hi - 1, Wo = 2 * Wi - 1;
  const long npix = (long)B * Ho * Wo;
  NOISE_REQUIRE(npix < (1L << 31) && (long)B * Hi * Wi * K < (1L << 40), "noise_pw_fwd_virt: tensor too large");
  VirtDims d{npix, K, N, Hi, Wi, Ho, Wo, (long)B * Hi * Wi};
  LazySrc In{in_y, in_coef, in_relu6}, Dw{nullptr, dw_coef, dw_relu6};
  int grid = noise_grid((long)B * Hi * Wi, 64, 1024);   // four tiles of 16 cells per workgroup and iteration
  const int KC = (K + 15) / 16, MT = (N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  // Three workgroups per CU where the registers allow it (a chunk of three channel tiles in flight instead of six): the
  // two small layers are a handful of tiles per wave and live on resident waves, 0.103 -> 0.074 ms and 0.105 -> 0.091 ms
  // for 1280 segments; the 72 -> 16 layer keeps its lazy coefficients in registers at two (0.186 ms, 0.197 ms at three).
  if (KC == 5 && MT == 1) launch_virt<5, 1, 1, 2>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  else if (KC == 6 && MT == 2) launch_virt<3, 2, 2, 3>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  else launch_virt<3, 4, 2, 3>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_pw_fwd_virt");
  return 0;
}

// Backward of noise_pw_fwd_virt: g_in (B, 2Hi-1, 2Wi-1, K) = dL/d(activated depthwise output), masked by the depthwise
// ReLU6, with the depthwise BatchNorm's backward sums [sum g | sum g*xhat] in stats_slab (*slab_rows rows of 2K floats),
// and the per-workgroup weight-gradient rows (N*K floats each) in scratch.  dz / y_out / coef3: lazy gradient of the
// projection's output (coef3 NULL: dz is dL/dy).  dw_coef: [scale | shift | mean | invstd | mean_lo] of the depthwise
// BatchNorm.  scratch and stats_slab: noise_pw_bwd_max_rows(N, K) rows.
NOISE_EXPORT int noise_pw_bwd_virt(const float* dz, const float* y_out, const float* coef3, const float* in_y,
                                   const float* in_coef, int in_relu6, const float* wdw, const float* dw_coef,
                                   int dw_relu6, const float* w, float* g_in, float* scratch, float* stats_slab,
                                   int* slab_rows, int B, int Hi, int Wi, int K, int N, void* stream) {
  NOISE_REQUIRE(dz && in_y && wdw && dw_coef && w && g_in && scratch && stats_slab, "noise_pw_bwd_virt: null pointer");
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr, "noise_pw_bwd_virt: coef3 needs y_out");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_pw_bwd_virt: empty input");
  NOISE_REQUIRE(K % 8 == 0 && N % 4 == 0 && noise_pw_fwd_virt_supported(K, N),
                "noise_pw_bwd_virt: no kernel instance for K=%d N=%d (served: 72->16, 96->24, 192->32)", K, N);
  const int Ho = 2 * Hi - 1, Wo = 2 * Wi - 1;
  const long npix = (long)B * Ho * Wo, ncells = (long)B * Hi * Wi;
  NOISE_REQUIRE(npix < (1L << 31) && ncells * K < (1L << 40), "noise_pw_bwd_virt: tensor too large");
  VirtDims d{npix, K, N, Hi, Wi, Ho, Wo, ncells};
  LazyGrad G{dz, y_out, coef3};
  LazySrc In{in_y, in_coef, in_relu6};
  int grid = noise_grid(ncells, 16, noise_pw_bwd_max_rows(N, K));
  const int KC = (N + 15) / 16, MTI = (K + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (KC == 1 && MTI == 5) launch_virt_bwd<1, 5>(G, In, wdw, dw_coef, dw_relu6, w, g_in, stats_slab, scratch, d, grid, st);
  else if (KC == 2 && MTI == 6) launch_virt_bwd<2, 6>(G, In, wdw, dw_coef, dw_relu6, w, g_in, stats_slab, scratch, d, grid, st);
  else launch_virt_bwd<2, 12>(G, In, wdw, dw_coef, dw_relu6, w, g_in, stats_slab, scratch, d, grid, st);
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_pw_bwd_virt");
  return 0;
}
