// Pointwise (1x1) convolutions of the inverted-residual blocks as fp32-MFMA tile products, gfx950.
//
// Replaces train/models.py:371-381 (expand 1x1 + BN + ReLU6) and :395-396 (linear 1x1 + BN) in
// forward, and their input/weight gradients in backward.  Activations are NHWC fp32, so a layer is
//     out[p][co] = sum_ci a[p][ci] * W[co][ci]           p = flattened (b,y,x)
// with a[] the lazily normalised input (one or two sources: decoder inputs are x + skip).
//
// v_mfma_f32_16x16x4_f32 with A = weights (rows = co), B = activations (cols = 16 pixels): a lane then
// owns 4 consecutive output channels of one pixel, i.e. one 16-byte NHWC store.  The k index of the
// four MFMAs of a 16-channel chunk is permuted so that both operands are fetched as float4
// (lane (p,q) holds channels 16j+4q..+3 and feeds element r to MFMA r).  Weights are staged once per
// workgroup in LDS in exactly that fragment order (conflict-free ds_read_b128).
#include "lazy.h"

namespace {

// 16 zero bytes that an absent second source points at: its loads stay unconditional (a branch around a load makes
// the compiler drain every outstanding load behind it) and cost one broadcast cache hit.
__device__ float4 g_zero4;

struct PwDims {
  long npix;
  int K, N, transW;  // N: output channels = output row stride (multiple of 4)
};

// GRAD = false: input = LazySrc A (+B);   GRAD = true: input = LazyGrad G (dy rebuilt on the fly)
template <bool GRAD, int KC, int MT>
__global__ __launch_bounds__(256) void pw_gemm_kernel(LazySrc A, LazySrc Bs, LazyGrad G, const float* __restrict__ W,
                                                      const float* __restrict__ add, float* __restrict__ out,
                                                      float* __restrict__ slab, PwDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int K = d.K, N = d.N;
  float* wfrag = lds;                    // [MT][KC][64][4]
  float* cf = lds + MT * KC * 256;       // [4][KC*16] (forward) or [6][KC*16] (gradient)
  constexpr int KP = KC * 16;
  if (d.transW) {
    for (int idx = tid; idx < MT * KC * 256; idx += 256) {
      const int r = idx & 3, l = (idx >> 2) & 63, t = idx >> 8;
      const int jj = t % KC, m = t / KC;
      const int co = 16 * m + (l & 15), ci = 16 * jj + 4 * (l >> 4) + r;
      wfrag[idx] = (co < N && ci < K) ? W[(size_t)ci * N + co] : 0.f;
    }
  } else {
    // W is read once, row by row, with coalesced 16-byte loads; four consecutive input channels of one output channel
    // are one float4 of the fragment image ([m][jj][lane = co%16 + 16*((ci%16)/4)][ci%4])
    if ((N & 15) || (K & 15)) {
      for (int idx = tid; idx < MT * KC * 256; idx += 256) wfrag[idx] = 0.f;
      __syncthreads();
    }
    const int K4 = K >> 2;
#pragma unroll 4
    for (int e = tid; e < N * K4; e += 256) {
      const int co = e / K4, ci = 4 * (e - co * K4);
      const float4 v = ld4(W + (size_t)co * K + ci);
      st4(wfrag + ((((co >> 4) * KC + (ci >> 4)) * 64) + (co & 15) + 16 * ((ci & 15) >> 2)) * 4, v);
    }
  }
  for (int k = tid; k < KP; k += 256) {
    const bool in = k < K;
    if (GRAD) {
      cf[k] = (in && G.coef3) ? G.coef3[k] : 1.f;
#pragma unroll
      for (int j = 1; j < 6; ++j) cf[j * KP + k] = (in && G.coef3) ? G.coef3[j * K + k] : 0.f;
    } else {
      cf[k] = (in && A.coef) ? A.coef[k] : 1.f;
      cf[KP + k] = (in && A.coef) ? A.coef[K + k] : 0.f;
      cf[2 * KP + k] = (in && Bs.y && Bs.coef) ? Bs.coef[k] : 1.f;
      cf[3 * KP + k] = (in && Bs.y && Bs.coef) ? Bs.coef[K + k] : 0.f;
    }
  }
  __syncthreads();

  float4 s1[MT], s2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) s1[m] = s2[m] = f4(0.f);

  // Branch-free operand fetch: clamped (always valid) addresses, an absent second source reads the zero dummy with
  // element stride 0 and scale 0, an absent activation clamps to (-inf, +inf).  The raw operands of the NEXT tile
  // are requested before the MFMAs of the current one.
  const float* zero = reinterpret_cast<const float*>(&g_zero4);
  const float* src0 = GRAD ? G.dz : A.y;
  const float* src1 = GRAD ? (G.coef3 ? G.y : G.dz) : (Bs.y ? Bs.y : zero);
  const size_t mul1 = (GRAD || Bs.y) ? 1 : 0;
  const bool has_b = !GRAD && Bs.y != nullptr;
  const float alo = (!GRAD && A.relu6) ? 0.f : -INFINITY, ahi = (!GRAD && A.relu6) ? 6.f : INFINITY;
  const float blo = (has_b && Bs.relu6) ? 0.f : -INFINITY, bhi = (has_b && Bs.relu6) ? 6.f : INFINITY;
  const float bgate = has_b ? 1.f : 0.f;
  const long ntiles = (d.npix + 15) >> 4;
  float4 r0[KC], r1[KC];
  auto issue = [&](long tile) {
    long pix = tile * 16 + p;
    pix = pix < d.npix ? pix : d.npix - 1;
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
      const int ci = 16 * jj + 4 * q;
      const size_t off = (size_t)pix * K + (ci < K ? ci : 0);
      r0[jj] = ld4(src0 + off);
      r1[jj] = ld4(src1 + off * mul1);
    }
  };
  const long tile0 = (long)blockIdx.x * 4 + wave, tstep = (long)gridDim.x * 4;
  if (tile0 < ntiles) issue(tile0);
  for (long tile = tile0; tile < ntiles; tile += tstep) {
    const long pix = tile * 16 + p;
    const bool valid = pix < d.npix;
    float4 in[KC];
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
      const int ci = 16 * jj + 4 * q;
      float4 v;
      if (GRAD) {
        const float4 t1 = sub4(sub4(r0[jj], ld4(cf + KP + ci)), ld4(cf + 2 * KP + ci));
        const float4 t2 = sub4(sub4(r1[jj], ld4(cf + 4 * KP + ci)), ld4(cf + 5 * KP + ci));
        v = fma4(ld4(cf + 3 * KP + ci), t2, ld4(cf + ci) * t1);
      } else {
        v = clamp4(fma4(r0[jj], ld4(cf + ci), ld4(cf + KP + ci)), alo, ahi);
        const float4 u = clamp4(fma4(r1[jj], ld4(cf + 2 * KP + ci), ld4(cf + 3 * KP + ci)), blo, bhi);
        v = fma4(u, f4(bgate), v);
      }
      const bool okc = valid && ci < K;
      in[jj] = make_float4(okc ? v.x : 0.f, okc ? v.y : 0.f, okc ? v.z : 0.f, okc ? v.w : 0.f);
    }
    issue(tile + tstep < ntiles ? tile + tstep : tile);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < KC; ++jj) {
        const float4 wa = ld4(wfrag + ((m * KC + jj) * 64 + lane) * 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, in[jj].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, in[jj].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, in[jj].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, in[jj].w, acc, 0, 0, 0);
      }
      const int co = 16 * m + 4 * q;
      float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (valid && co < N) {
        const size_t ooff = (size_t)pix * N + co;
        if (add) o = o + ld4(add + ooff);
        st4(out + ooff, o);
        s1[m] = s1[m] + o;
        s2[m] = fma4(o, o, s2[m]);
      }
    }
  }

  if (slab != nullptr) {
    // reduce over the 16 pixel lanes that share q, then over the 4 waves through LDS
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
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) s += red[(wv * 2 + j) * MT * 16 + c];
      slab[(size_t)blockIdx.x * 2 * N + i] = s;
    }
  }
}

// dW[co][ci] = sum_p dy[p][co] * a[p][ci]:   A[i=co][k=pixel], B[k=pixel][j=ci]; 4 pixels per MFMA.
// A workgroup stages 64 pixels of dy and a in LDS with coalesced float4 loads (the lazy transforms are
// applied once, on the way in); each wave then multiplies its 16 pixels against every (co,ci) tile.
constexpr int WG_TILE = 64;

template <int MT, int NT>
__global__ __launch_bounds__(256) void pw_wgrad_kernel(LazyGrad G, LazySrc A, LazySrc Bs, float* __restrict__ slab,
                                                       long npix, int N, int K) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int pn = N + 4, pk = K + 4;          // LDS row pitches (keep 16-byte alignment)
  float* dys = lds;                          // [64][pn]
  float* as = lds + WG_TILE * pn;            // [64][pk]
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int n4 = N >> 2, k4 = K >> 2;
  const long ntiles = (npix + WG_TILE - 1) / WG_TILE;
  const long iters = (ntiles + gridDim.x - 1) / gridDim.x;      // uniform trip count: the loop has barriers
  for (long it = 0; it < iters; ++it) {
    const long tile = (long)blockIdx.x + it * gridDim.x;
    const long pix0 = tile * WG_TILE;
    __syncthreads();
    for (int e = tid; e < WG_TILE * n4; e += 256) {
      const int p = e / n4, c = (e - p * n4) * 4;
      float4 v = f4(0.f);
      if (tile < ntiles && pix0 + p < npix) {
        const size_t off = (size_t)(pix0 + p) * N + c;
        v = ld4(G.dz + off);
        if (G.coef3) {
          const float4 t1 = sub4(sub4(v, ld4(G.coef3 + N + c)), ld4(G.coef3 + 2 * N + c));
          const float4 t2 = sub4(sub4(ld4(G.y + off), ld4(G.coef3 + 4 * N + c)), ld4(G.coef3 + 5 * N + c));
          v = fma4(ld4(G.coef3 + 3 * N + c), t2, ld4(G.coef3 + c) * t1);
        }
      }
      st4(dys + p * pn + c, v);
    }
    for (int e = tid; e < WG_TILE * k4; e += 256) {
      const int p = e / k4, c = (e - p * k4) * 4;
      float4 v = f4(0.f);
      if (tile < ntiles && pix0 + p < npix) {
        const size_t off = (size_t)(pix0 + p) * K + c;
        v = ld4(A.y + off);
        if (A.coef) v = fma4(v, ld4(A.coef + c), ld4(A.coef + K + c));
        if (A.relu6) v = relu6_4(v);
        if (Bs.y) {
          float4 u = ld4(Bs.y + off);
          if (Bs.coef) u = fma4(u, ld4(Bs.coef + c), ld4(Bs.coef + K + c));
          if (Bs.relu6) u = relu6_4(u);
          v = v + u;
        }
      }
      st4(as + p * pk + c, v);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = 16 * wave + 4 * s + kq;
      float dy[MT], av[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) dy[m] = (16 * m + i < N) ? dys[p * pn + 16 * m + i] : 0.f;
#pragma unroll
      for (int n = 0; n < NT; ++n) av[n] = (16 * n + i < K) ? as[p * pk + 16 * n + i] : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dy[m], av[n], acc[m][n], 0, 0, 0);
    }
  }

  // D layout: col = ci0 + (lane&15), rows = co0 + 4*(lane>>4) + reg.  Fold the four waves through LDS.
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const int ci = 16 * n + i;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int co = 16 * m + 4 * kq + e;
            if (co < N && ci < K) {
              float* dst = lds + co * K + ci;
              *dst = (wv == 0) ? acc[m][n][e] : *dst + acc[m][n][e];
            }
          }
        }
    }
  }
  __syncthreads();
  for (int j = tid; j < N * K; j += 256) slab[(size_t)blockIdx.x * N * K + j] = lds[j];
}

// Fused backward of a pointwise layer: one pass over (dz, y_out) and the layer input computes
//   g_in[p][ci] = sum_co dy[p][co] W[co][ci]     (then the BwdEpilogue: + g_other, producer ReLU6 mask, producer BN sums)
//   dW[co][ci] += sum_p dy[p][co] a[p][ci]
// KC = ceil(N/16) (N = Cout), MTI = ceil(K/16) (K = Cin).  64 pixels per workgroup iteration are staged in LDS
// (dy with the lazy BatchNorm backward applied, a with the lazy activation applied); each wave owns 16 of them.
// LDS of one workgroup (floats) and the number of workgroups a CU can hold by that measure (at most 3-4): the register
// budget is pinned to the same occupancy, otherwise the scheduler hoists every staged load and halves it.
#ifdef PW_STAMPS
__device__ unsigned long long g_pw_stamps[16];
#define PW_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_pw_stamps[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define PW_T(i)
#endif
#ifndef PW_BWD_PF_MAX
#define PW_BWD_PF_MAX 40
#endif
// Row pitch (floats) of a staged tile with `tiles` 16-channel tiles: 16-byte aligned rows, 4 banks apart.  (A pitch
// congruent to 16 modulo 32 makes the weight-gradient operand reads conflict-free but the 128-bit row reads of the
// input-gradient pass 8-way conflicted: measured 25 % slower on the 96 -> 32 layers.)
constexpr int pw_bwd_pitch(int tiles) { return tiles * 16 + 4; }
constexpr int pw_bwd_lds_floats(int KC, int MTI) {
  return MTI * KC * 256 + WG_TILE * pw_bwd_pitch(KC) + WG_TILE * pw_bwd_pitch(MTI) + 5 * MTI * 16;
}
constexpr int pw_bwd_min_blocks(int KC, int MTI) {
  const int b = 160 * 1024 / (pw_bwd_lds_floats(KC, MTI) * 4);
  const int cap = KC * MTI <= 3 ? 4 : (MTI >= 4 ? 2 : 3);   // register budget: 128 / 256 / 168 per lane
  return b > cap ? cap : (b < 1 ? 1 : b);
}

// rows of the 64-pixel tile a thread stages when a row has `tiles` 16-channel tiles (see the staging map below)
constexpr int pw_bwd_passes(int tiles) { return (WG_TILE + 256 / (tiles * 4) - 1) / (256 / (tiles * 4)); }

// PF: the next tile's global loads are requested while the current tile is multiplied and stay in registers until it
// is staged, so the memory round trip of a tile hides behind MFMAs instead of relying on other resident workgroups.
// Used where a CU holds at most two workgroups anyway (wide layers: LDS or the register cap), which leaves 256
// registers per lane; the narrower layers run three or four workgroups per CU and spill with the extra live
// registers, and with 96 input channels (6 tiles) the prefetch and the epilogue operands do not fit together
// (measured 0.25 -> 0.33 ms on the 96 -> 24 layer with either of them spilled or split into rounds).
// WGF: the weight-gradient MFMAs run between the request for the input-gradient epilogue's operands and their first
// use.  Kept where the extra live registers fit (measured per instantiation: 72 -> 16 layer 0.61 -> 0.58 ms, 32 -> 16
// layer 0.28 -> 0.26 ms); the instantiations that spill with it lose 10-30 %.
constexpr bool pw_bwd_wgrad_first(int KC, int MTI, bool has_b) {
  return !has_b && ((KC == 1 && (MTI == 2 || MTI == 5)) || (KC == 2 && (MTI == 3 || MTI == 6)));
}
constexpr bool pw_bwd_prefetch(int KC, int MTI, bool has_b) {
  return pw_bwd_min_blocks(KC, MTI) <= 2 && MTI != 6 &&
         2 * pw_bwd_passes(KC) + (has_b ? 2 : 1) * pw_bwd_passes(MTI) <= PW_BWD_PF_MAX;
}

// RY (expanding layers with at most 16 input channels, BatchNorm behind the layer; with more input channels the extra
// MFMAs and barrier cost more than the read they save): the raw layer output y that the lazy
// gradient needs (dy = s*(dz - k1) + a2*(y - mean)) is recomputed from the staged input tile, y = a W^T (KC*MTI*4
// MFMAs per wave and tile), instead of being read back: the widest tensor of the layer is read once (dz), not twice.
constexpr bool pw_bwd_recompute(int KC, int MTI) { return MTI == 1 && KC >= 2; }

template <int KC, int MTI, bool HAS_B, bool RY>
__global__ __launch_bounds__(256, pw_bwd_min_blocks(KC, MTI)) void pw_bwd_kernel(LazyGrad G, LazySrc A, LazySrc Bs, const float* __restrict__ W,
                                                     float* __restrict__ g_in, BwdEpilogue E, float* __restrict__ dw_slab,
                                                     long npix, int N, int K) {
  extern __shared__ float lds[];
#ifdef PW_STAMPS
  unsigned long long last_ = clock64();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;       // dgrad view: pixel lane / k-quarter;  wgrad view: i = p, kq = q
  constexpr int NP = KC * 16, KP = MTI * 16;
  constexpr int pn = pw_bwd_pitch(KC), pk = pw_bwd_pitch(MTI);
  float* wfrag = lds;                            // [MTI][KC][64][4]
  float* dys = wfrag + MTI * KC * 256;           // [64][pn]
  float* as = dys + WG_TILE * pn;                // [64][pk]
  float* ecf = as + WG_TILE * pk;                // [5][KP] producer BN coefficients for the epilogue
  float* wfy = ecf + 5 * KP;                     // RY: [KC][MTI][64][4] forward fragment image of W
  float* gcf = wfy + KC * MTI * 256;             // RY: [6][NP] lazy-gradient coefficients (0 for padded channels)
  // W is read once, row by row, with coalesced 16-byte loads and scattered into the fragment image
  // ([m][jj][lane = ci%16 + 16*((co%16)/4)][co%4])
  if ((N & 15) || (K & 15)) {
    for (int idx = tid; idx < MTI * KC * 256; idx += 256) {
      wfrag[idx] = 0.f;
      if (RY) wfy[idx] = 0.f;
    }
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
      if (RY) st4(wfy + ((((co >> 4) * MTI + (ci >> 4)) * 64) + (co & 15) + 16 * ((ci & 15) >> 2)) * 4, v);
    }
  }
  if (RY) {
    for (int k = tid; k < 6 * NP; k += 256) {
      const int j = k / NP, c = k - j * NP;
      gcf[k] = c < N ? G.coef3[j * N + c] : 0.f;
    }
  }
  for (int k = tid; k < KP; k += 256) {
    const bool in = k < K && E.coef != nullptr;
    ecf[k] = in ? E.coef[k] : 1.f;
    ecf[KP + k] = in ? E.coef[K + k] : 0.f;
    ecf[2 * KP + k] = in ? E.coef[2 * K + k] : 0.f;
    ecf[3 * KP + k] = in ? E.coef[3 * K + k] : 1.f;
    ecf[4 * KP + k] = in ? E.coef[4 * K + k] : 0.f;
  }
  // weight gradient: the KC x MTI tiles of dW are dealt round-robin to the four waves (tile t = m*MTI + n belongs to
  // wave t % 4); a wave runs all 64 pixels of a staged tile through its own dW tiles, so no cross-wave sum is needed
  constexpr int T = KC * MTI, TPW = (T + 3) / 4;
  f32x4 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int dyo[TPW], avo[TPW];   // LDS offsets of this wave's tiles: dys column 16*m + p, as column 16*n + p
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    // a wave without a j-th tile recomputes its first one into an accumulator that is never written back: the
    // k-loop below then has no branches and its LDS reads can run ahead of the MFMAs
    const int t = wave + 4 * j < T ? wave + 4 * j : wave % T, m = t / MTI, n = t - m * MTI;
    dyo[j] = 16 * m + p;
    avo[j] = 16 * n + p;
  }
  float4 s1[MTI], s2[MTI];
#pragma unroll
  for (int m = 0; m < MTI; ++m) s1[m] = s2[m] = f4(0.f);

  // Staging map: a thread keeps one group of 4 channels for every row it stages (rows rn, rn + RN, ... of the
  // 64-pixel tile), so the per-channel coefficients of the lazy transforms are loaded once, into registers.
  constexpr int CGN = NP / 4, RN = 256 / CGN, PASS_N = pw_bwd_passes(KC);
  constexpr int CGK = KP / 4, RK = 256 / CGK, PASS_K = pw_bwd_passes(MTI);
  const int rn = tid / CGN, cn = (tid - rn * CGN) * 4;
  const int rk = tid / CGK, ck = (tid - rk * CGK) * 4;
  const bool on = rn < RN && cn < N, ok = rk < RK && ck < K;
  float4 gs = f4(1.f), gk1h = f4(0.f), gk1l = f4(0.f), ga2 = f4(0.f), gmh = f4(0.f), gml = f4(0.f);
  if (on && G.coef3) {
    gs = ld4(G.coef3 + cn);
    gk1h = ld4(G.coef3 + N + cn);
    gk1l = ld4(G.coef3 + 2 * N + cn);
    ga2 = ld4(G.coef3 + 3 * N + cn);
    gmh = ld4(G.coef3 + 4 * N + cn);
    gml = ld4(G.coef3 + 5 * N + cn);
  }
  float4 asc = f4(1.f), ash = f4(0.f), bsc = f4(1.f), bsh = f4(0.f);
  if (ok && A.coef) {
    asc = ld4(A.coef + ck);
    ash = ld4(A.coef + K + ck);
  }
  if (HAS_B && ok && Bs.coef) {
    bsc = ld4(Bs.coef + ck);
    bsh = ld4(Bs.coef + K + ck);
  }

  // an absent second gradient reads the zero dummy with element stride 0: its loads need no branch
  const float* gop = E.g_other ? E.g_other : reinterpret_cast<const float*>(&g_zero4);
  const size_t gmul = E.g_other ? 1 : 0;
  const long ntiles = (npix + WG_TILE - 1) / WG_TILE;
  const long iters = (ntiles + gridDim.x - 1) / gridDim.x;
  constexpr bool PF = pw_bwd_prefetch(KC, MTI, HAS_B);
  constexpr bool WGF = pw_bwd_wgrad_first(KC, MTI, HAS_B);
  constexpr bool ONE_ROUND = PF || 2 * PASS_N + (HAS_B ? 2 : 1) * PASS_K <= 12;
  // input-channel tiles per epilogue round
  constexpr int MC = MTI <= 6 ? MTI : (MTI % 6 == 0 ? 6 : 4);
  float4 rz[PASS_N], ry[PASS_N], ra[PASS_K], rb[HAS_B ? PASS_K : 1];   // a thread's share of one staged tile
  const int cnc1 = on ? cn : 0, ckc1 = ok ? ck : 0;
  const int rnc1 = rn < RN ? rn : 0, rkc1 = rk < RK ? rk : 0;
  auto issue_tile = [&](long tile_) {
    const long p0 = tile_ * WG_TILE;
#pragma unroll
    for (int u = 0; u < PASS_N; ++u) {
      long pix = p0 + rnc1 + u * RN;
      pix = pix < npix ? pix : npix - 1;
      const size_t off = (size_t)pix * N + cnc1;
      rz[u] = ld4(G.dz + off);
      ry[u] = (G.coef3 && !RY) ? ld4(G.y + off) : f4(0.f);
    }
#pragma unroll
    for (int u = 0; u < PASS_K; ++u) {
      long pix = p0 + rkc1 + u * RK;
      pix = pix < npix ? pix : npix - 1;
      const size_t off = (size_t)pix * K + ckc1;
      ra[u] = ld4(A.y + off);
      if (HAS_B) rb[u] = ld4(Bs.y + off);
    }
  };
  if (PF) issue_tile(blockIdx.x);
  PW_T(0);
  for (long it = 0; it < iters; ++it) {
    const long tile = (long)blockIdx.x + it * gridDim.x;
    const long pix0 = tile * WG_TILE;
    const long left = tile < ntiles ? npix - pix0 : 0;   // valid rows of this tile
    __syncthreads();
    PW_T(1);
    // Every load is issued from a clamped (always valid) address, then transformed and stored.  When a thread's share
    // of the tile is small enough to sit in registers (ONE_ROUND) all of it is requested before the first use, so a
    // tile costs one memory round trip; otherwise in batches of up to 4 rows.  With PF the registers were filled
    // while the previous tile was being multiplied (the request for the next tile follows the barrier below).
    if (ONE_ROUND) {
      if (!PF) issue_tile(tile);
      if (rn < RN) {
#pragma unroll
        for (int u = 0; u < PASS_N; ++u) {
          const int pp = rn + u * RN;
          if (pp < WG_TILE) {
            float4 v = rz[u];
            if (G.coef3 && !RY) v = fma4(ga2, sub4(sub4(ry[u], gmh), gml), gs * sub4(sub4(v, gk1h), gk1l));
            st4(dys + pp * pn + cn, (on && pp < left) ? v : f4(0.f));
          }
        }
      }
      if (rk < RK) {
#pragma unroll
        for (int u = 0; u < PASS_K; ++u) {
          const int pp = rk + u * RK;
          if (pp < WG_TILE) {
            float4 v = fma4(ra[u], asc, ash);
            if (A.relu6) v = relu6_4(v);
            if (HAS_B) {
              float4 w2 = fma4(rb[u], bsc, bsh);
              if (Bs.relu6) w2 = relu6_4(w2);
              v = v + w2;
            }
            st4(as + pp * pk + ck, (ok && pp < left) ? v : f4(0.f));
          }
        }
      }
    } else {
    if (rn < RN) {
      const int cnc = on ? cn : 0;
#pragma unroll
      for (int ps0 = 0; ps0 < PASS_N; ps0 += 4) {
        float4 rz[4], ry[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ps0 + u < PASS_N) {
            long pix = pix0 + rn + (ps0 + u) * RN;
            pix = pix < npix ? pix : npix - 1;
            const size_t off = (size_t)pix * N + cnc;
            rz[u] = ld4(G.dz + off);
            ry[u] = (G.coef3 && !RY) ? ld4(G.y + off) : f4(0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = rn + (ps0 + u) * RN;
          if (ps0 + u < PASS_N && pp < WG_TILE) {
            float4 v = rz[u];
            if (G.coef3 && !RY) v = fma4(ga2, sub4(sub4(ry[u], gmh), gml), gs * sub4(sub4(v, gk1h), gk1l));
            st4(dys + pp * pn + cn, (on && pp < left) ? v : f4(0.f));
          }
        }
      }
    }
    if (rk < RK) {
      const int ckc = ok ? ck : 0;
#pragma unroll
      for (int ps0 = 0; ps0 < PASS_K; ps0 += 4) {
        float4 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (ps0 + u < PASS_K) {
            long pix = pix0 + rk + (ps0 + u) * RK;
            pix = pix < npix ? pix : npix - 1;
            const size_t off = (size_t)pix * K + ckc;
            ra[u] = ld4(A.y + off);
            rb[u] = Bs.y ? ld4(Bs.y + off) : f4(0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = rk + (ps0 + u) * RK;
          if (ps0 + u < PASS_K && pp < WG_TILE) {
            float4 v = fma4(ra[u], asc, ash);
            if (A.relu6) v = relu6_4(v);
            if (Bs.y) {
              float4 w2 = fma4(rb[u], bsc, bsh);
              if (Bs.relu6) w2 = relu6_4(w2);
              v = v + w2;
            }
            st4(as + pp * pk + ck, (ok && pp < left) ? v : f4(0.f));
          }
        }
      }
    }
    }
    PW_T(2);
    __syncthreads();
    PW_T(3);
    float4 in[KC];   // dy of this wave's 16 pixels: lane (p, q) holds channels 16jj + 4q .. + 3 of pixel p
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) in[jj] = ld4(dys + (16 * wave + p) * pn + 16 * jj + 4 * q);
    if (RY) {
      // y = a W^T for these pixels (D[i = co][j = pixel]: the lane receives exactly the channels it holds of dz),
      // then the lazy BatchNorm backward; the finished dy goes back to the staged tile for the weight gradient
      const bool live = tile < ntiles && pix0 + 16 * wave + p < npix;
      float4 av[MTI];
#pragma unroll
      for (int m = 0; m < MTI; ++m) av[m] = ld4(as + (16 * wave + p) * pk + 16 * m + 4 * q);
#pragma unroll
      for (int jj = 0; jj < KC; ++jj) {
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MTI; ++m) {
          const float4 wa = ld4(wfy + ((jj * MTI + m) * 64 + lane) * 4);
          y = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, av[m].x, y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, av[m].y, y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, av[m].z, y, 0, 0, 0);
          y = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, av[m].w, y, 0, 0, 0);
        }
        const int co = 16 * jj + 4 * q;
        const float4 yv = make_float4(y[0], y[1], y[2], y[3]);
        const float4 t1 = sub4(sub4(in[jj], ld4(gcf + NP + co)), ld4(gcf + 2 * NP + co));
        const float4 t2 = sub4(sub4(yv, ld4(gcf + 4 * NP + co)), ld4(gcf + 5 * NP + co));
        const float4 v = fma4(ld4(gcf + 3 * NP + co), t2, ld4(gcf + co) * t1);
        in[jj] = live ? v : f4(0.f);
        st4(dys + (16 * wave + p) * pn + co, in[jj]);
      }
    }
    if (RY) __syncthreads();   // every wave has put its finished dy rows back
    // ---- wgrad: 16 k-steps of 4 pixels over this wave's tiles.  It only reads the staged tiles, so it runs between
    // the request for the input-gradient epilogue's operands and their first use: the MFMAs cover that round trip.
    auto wgrad_pass = [&]() {
#pragma unroll 2
      for (int s = 0; s < 16; ++s) {
        const int pp = 4 * s + q;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(dys[pp * pn + dyo[j]], as[pp * pk + avo[j]], acc[j], 0, 0, 0);
      }
    };
    // ---- dgrad for this wave's 16 pixels
    if (g_in != nullptr) {
      const long pix = pix0 + 16 * wave + p;
      const bool valid = tile < ntiles && pix < npix;
      const size_t prow = (size_t)(pix < npix ? pix : npix - 1) * K;
      // What the epilogue needs from memory is requested before the MFMA chains, from clamped (always valid)
      // addresses with no branch around the loads: all input-channel tiles at once when the registers allow it,
      // otherwise MC of them at a time (a wide layer then pays MTI / MC memory round trips per tile, not MTI).
      // Memory operations complete in order: the next tile (PF) is requested AFTER the last loads this tile's
      // epilogue waits for, so that wait does not include it, and it stays in flight through the MFMA chains.
#pragma unroll
      for (int m0 = 0; m0 < MTI; m0 += MC) {
        float4 yr[MC], go[MC];
#pragma unroll
        for (int u = 0; u < MC; ++u) {
          const int ci = 16 * (m0 + u) + 4 * q, cic = (m0 + u < MTI && ci < K) ? ci : 0;
#ifdef PW_DBG_NOLOAD
          yr[u] = f4(1.f); go[u] = f4(0.f);
#else
          yr[u] = ld4(A.y + prow + cic);
          go[u] = ld4(gop + (prow + cic) * gmul);
#endif
        }
        if (PF && m0 + MC >= MTI) issue_tile(tile + gridDim.x);
        if (WGF && m0 == 0) wgrad_pass();
#pragma unroll
        for (int u = 0; u < MC; ++u) {
          const int m = m0 + u;
          if (m >= MTI) continue;
          f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jj = 0; jj < KC; ++jj) {
            const float4 wa = ld4(wfrag + ((m * KC + jj) * 64 + lane) * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.x, in[jj].x, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.y, in[jj].y, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.z, in[jj].z, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(wa.w, in[jj].w, o, 0, 0, 0);
          }
          const int ci = 16 * m + 4 * q;
          if (valid && ci < K) {
            const size_t off = prow + ci;
            float4 g = make_float4(o[0], o[1], o[2], o[3]) + go[u];
            if (E.coef) {
              const float4 yraw = yr[u];
              if (E.relu6) {
                const float4 v = fma4(yraw, ld4(ecf + ci), ld4(ecf + KP + ci));
                g.x = (v.x > 0.f && v.x < 6.f) ? g.x : 0.f;
                g.y = (v.y > 0.f && v.y < 6.f) ? g.y : 0.f;
                g.z = (v.z > 0.f && v.z < 6.f) ? g.z : 0.f;
                g.w = (v.w > 0.f && v.w < 6.f) ? g.w : 0.f;
              }
              const float4 xh = sub4(sub4(yraw, ld4(ecf + 2 * KP + ci)), ld4(ecf + 4 * KP + ci)) * ld4(ecf + 3 * KP + ci);
              s1[m] = s1[m] + g;
              s2[m] = fma4(g, xh, s2[m]);
            }
#ifdef PW_DBG_NOSTORE
            if (g.x == 123.456f) st4(g_in + off, g);
#else
            st4(g_in + off, g);
#endif
          }
        }
      }
    }
    if (PF && g_in == nullptr) issue_tile(tile + gridDim.x);
    PW_T(4);
    if (!WGF || g_in == nullptr) wgrad_pass();
    PW_T(5);
  }

  // ---- producer BN sums: reduce over the 16 pixel lanes, then over the waves
  __syncthreads();
  if (E.stats_slab != nullptr) {
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
      E.stats_slab[(size_t)blockIdx.x * 2 * K + i2] = sum;
    }
    __syncthreads();
  }
  // ---- dW: D layout col = ci0 + (lane&15), rows = co0 + 4*(lane>>4) + reg; every tile has one owner, so the waves
  // drop their tiles into LDS side by side and the block writes the slab row with coalesced stores
  __syncthreads();
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
  PW_T(6);
}

__global__ __launch_bounds__(256) void pw_slab_reduce_kernel(const float* __restrict__ slab, int nrows, int ncols,
                                                             float* __restrict__ out) {
  __shared__ double red[256];
  const int j = blockIdx.x * SLAB_COLS + (threadIdx.x & (SLAB_COLS - 1));
  const bool ok = j < ncols;
  const double s = slab_column_sum(slab, nrows, ncols, j, ok, red);
  if (threadIdx.x < SLAB_COLS && ok) out[j] = (float)s;
}

template <bool GRAD, int KC, int MT>
void launch_pw(LazySrc A, LazySrc Bs, LazyGrad G, const float* W, const float* add, float* out, float* slab, PwDims d,
               int& grid, hipStream_t st) {
  size_t smem = (size_t)(MT * KC * 256 + 6 * KC * 16) * sizeof(float);
  const size_t red = (size_t)8 * MT * 16 * sizeof(float);
  if (red > smem) smem = red;
  NOISE_LAUNCH_ROUNDS((pw_gemm_kernel<GRAD, KC, MT>), grid, smem, st, A, Bs, G, W, add, out, slab, d);
}

#define PW_CASES(X)                                                                                     \
  X(1, 1) X(1, 2) X(1, 3) X(1, 4) X(1, 5) X(1, 6) X(2, 1) X(2, 2) X(2, 3) X(2, 4) X(2, 5) X(2, 6) X(2, 12)  \
  X(3, 1) X(3, 2) X(3, 3) X(3, 4) X(4, 1) X(4, 2) X(4, 3) X(4, 4) X(4, 6) X(4, 12) X(5, 1) X(5, 2) X(5, 5) \
  X(6, 1) X(6, 2) X(6, 4) X(6, 6) X(12, 2) X(12, 4) X(12, 12)

template <bool GRAD>
int dispatch_pw(int KC, int MT, LazySrc A, LazySrc Bs, LazyGrad G, const float* W, const float* add, float* out,
                float* slab, PwDims d, int& grid, hipStream_t st) {
#define X(kc, mt)                                                          \
  if (KC == kc && MT == mt) {                                              \
    launch_pw<GRAD, kc, mt>(A, Bs, G, W, add, out, slab, d, grid, st);     \
    return 0;                                                              \
  }
  PW_CASES(X)
#undef X
  return 1;
}

int pw_grid(long npix) { return noise_grid(npix, 16 * 4 * 4, 1024); }

}  // namespace

// out (npix, N) raw = a (npix, K) x W^T, W (N, K).
// stats_slab (optional): per-workgroup [sum(N) | sumsq(N)] rows; *slab_rows receives the row count.
NOISE_EXPORT int noise_pw_fwd(const float* a_y, const float* a_coef, int a_relu6, const float* b_y, const float* b_coef,
                              int b_relu6, const float* w, float* out, float* stats_slab, int* slab_rows, long npix,
                              int K, int N, void* stream) {
  NOISE_REQUIRE(a_y && w && out && npix > 0, "noise_pw_fwd: bad argument");
  NOISE_REQUIRE(K >= 8 && K % 8 == 0 && N >= 4 && N % 4 == 0, "noise_pw_fwd: K=%d must be a multiple of 8, N=%d of 4", K, N);
  const int KC = (K + 15) / 16, MT = (N + 15) / 16;
  PwDims d{npix, K, N, 0};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  LazyGrad G{nullptr, nullptr, nullptr};
  int grid = pw_grid(npix);
  NOISE_REQUIRE(dispatch_pw<false>(KC, MT, A, Bs, G, w, nullptr, out, stats_slab, d, grid, (hipStream_t)stream) == 0,
                "noise_pw_fwd: no kernel instance for K=%d N=%d", K, N);
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_pw_fwd");
  return 0;
}

// g_in (npix, Cin) = dy (npix, Cout) x W, W (Cout, Cin); dy = s*dz + alpha*y + beta (coef3 may be null).
// `add` (optional, same shape as g_in) is summed into the result: the second gradient of a skip tensor.
NOISE_EXPORT int noise_pw_dgrad(const float* dz, const float* y_out, const float* coef3, const float* w, const float* add,
                                float* g_in, long npix, int Cout, int Cin, void* stream) {
  NOISE_REQUIRE(dz && w && g_in && npix > 0, "noise_pw_dgrad: bad argument");
  NOISE_REQUIRE(Cout % 8 == 0 && Cin % 4 == 0, "noise_pw_dgrad: Cout=%d must be a multiple of 8 and Cin=%d of 4", Cout, Cin);
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr, "noise_pw_dgrad: coef3 needs y_out");
  const int KC = (Cout + 15) / 16, MT = (Cin + 15) / 16;
  PwDims d{npix, Cout, Cin, 1};
  LazySrc A{nullptr, nullptr, 0};
  LazyGrad G{dz, y_out, coef3};
  int grid = pw_grid(npix);
  NOISE_REQUIRE(dispatch_pw<true>(KC, MT, A, A, G, w, add, g_in, nullptr, d, grid, (hipStream_t)stream) == 0,
                "noise_pw_dgrad: no kernel instance for Cout=%d Cin=%d", Cout, Cin);
  NOISE_CHECK_LAUNCH("noise_pw_dgrad");
  return 0;
}

NOISE_EXPORT int noise_pw_wgrad_max_blocks(void) { return 256; }

// dw (Cout, Cin).  scratch: noise_pw_wgrad_max_blocks()*Cout*Cin floats.
NOISE_EXPORT int noise_pw_wgrad(const float* dz, const float* y_out, const float* coef3, const float* a_y,
                                const float* a_coef, int a_relu6, const float* b_y, const float* b_coef, int b_relu6,
                                float* dw, float* scratch, long npix, int Cout, int Cin, void* stream) {
  NOISE_REQUIRE(dz && a_y && dw && scratch && npix > 0, "noise_pw_wgrad: bad argument");
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr, "noise_pw_wgrad: coef3 needs y_out");
  NOISE_REQUIRE((size_t)Cout * Cin * 4 <= 64 * 1024, "noise_pw_wgrad: %dx%d weight too large for the LDS fold", Cout, Cin);
  const int MT = (Cout + 15) / 16, NT = (Cin + 15) / 16;
  LazyGrad G{dz, y_out, coef3};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  NOISE_REQUIRE(Cout % 4 == 0 && Cin % 4 == 0, "noise_pw_wgrad: channel counts must be multiples of 4");
  const int grid = noise_grid(npix, WG_TILE, 256);
  size_t smem = (size_t)WG_TILE * (Cout + Cin + 8) * sizeof(float);
  if ((size_t)Cout * Cin * sizeof(float) > smem) smem = (size_t)Cout * Cin * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  bool done = false;
#define X(nt, mt)                                                                                                      \
  if constexpr (mt * nt <= 48) {                                                                                       \
    if (!done && MT == mt && NT == nt) {                                                                               \
      NOISE_ALLOW_LDS((pw_wgrad_kernel<mt, nt>), smem);                                                                \
      hipLaunchKernelGGL((pw_wgrad_kernel<mt, nt>), dim3(grid), dim3(256), smem, st, G, A, Bs, scratch, npix, Cout, Cin); \
      done = true;                                                                                                     \
    }                                                                                                                  \
  }
  PW_CASES(X)
#undef X
  NOISE_REQUIRE(done, "noise_pw_wgrad: no kernel instance for Cout=%d Cin=%d", Cout, Cin);
  NOISE_CHECK_LAUNCH("noise_pw_wgrad");
  hipLaunchKernelGGL(pw_slab_reduce_kernel, dim3(noise_cdiv(Cout * Cin, SLAB_COLS)), dim3(256), 0, st, scratch, grid, Cout * Cin, dw);
  NOISE_CHECK_LAUNCH("noise_pw_wgrad/reduce");
  return 0;
}

#ifdef PW_STAMPS
NOISE_EXPORT int noise_debug_pw_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pw_stamps), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pw_stamps), 16 * sizeof(unsigned long long));
}
#endif

// Upper bound of workgroups (= slab rows) noise_pw_bwd uses for a Cout x Cin layer: enough to keep several
// workgroups per CU in flight, while the dW slab stays below 16 MiB.
NOISE_EXPORT int noise_pw_bwd_max_rows(int Cout, int Cin) {
  long cap = (16L << 20) / ((long)Cout * Cin * 4);
  if (cap < 256) cap = 256;
  if (cap > 1024) cap = 1024;
  return (int)cap;
}

// 1 when noise_pw_bwd reads y_out for a Cout x Cin layer with coef3, 0 when it recomputes it from the layer input.
NOISE_EXPORT int noise_pw_bwd_reads_y(int Cout, int Cin) {
  return pw_bwd_recompute((Cout + 15) / 16, (Cin + 15) / 16) && getenv("NOISE_PW_NO_RECOMPUTE") == nullptr ? 0 : 1;
}

// Fused backward of a pointwise layer (see pw_bwd_kernel): g_in (npix, Cin) and dw (Cout, Cin) from one pass.
// Epilogue (optional): g_in += g_other; with epi_coef (BatchNorm coefficients of source a's producer) g_in is
// masked by the producer's ReLU6 (epi_relu6) and stats_slab receives *slab_rows rows of [sum dz | sum dz*xhat]
// (2*Cin floats each).  scratch: noise_pw_bwd_max_rows(Cout, Cin)*Cout*Cin floats; stats_slab: that many rows.
NOISE_EXPORT int noise_pw_bwd(const float* dz, const float* y_out, const float* coef3, const float* a_y,
                              const float* a_coef, int a_relu6, const float* b_y, const float* b_coef, int b_relu6,
                              const float* w, float* g_in, float* dw, float* scratch, const float* g_other,
                              const float* epi_coef, int epi_relu6, float* stats_slab, int* slab_rows, long npix,
                              int Cout, int Cin, void* stream) {
  NOISE_REQUIRE(dz && a_y && w && g_in && scratch && npix > 0, "noise_pw_bwd: bad argument");
  NOISE_REQUIRE(coef3 == nullptr || y_out != nullptr, "noise_pw_bwd: coef3 needs y_out");
  NOISE_REQUIRE(epi_coef == nullptr || stats_slab != nullptr, "noise_pw_bwd: epi_coef needs stats_slab");
  NOISE_REQUIRE(Cout % 4 == 0 && Cin % 4 == 0, "noise_pw_bwd: channel counts must be multiples of 4");
  const int KC = (Cout + 15) / 16, MTI = (Cin + 15) / 16;
  LazyGrad G{dz, y_out, coef3};
  LazySrc A{a_y, a_coef, a_relu6}, Bs{b_y, b_coef, b_relu6};
  BwdEpilogue E{g_other, epi_coef, epi_relu6, epi_coef ? stats_slab : nullptr};
  int grid = noise_grid(npix, WG_TILE, noise_pw_bwd_max_rows(Cout, Cin));
  size_t smem = (size_t)pw_bwd_lds_floats(KC, MTI) * sizeof(float);
  if ((size_t)Cout * Cin * sizeof(float) > smem) smem = (size_t)Cout * Cin * sizeof(float);
  NOISE_REQUIRE(smem <= 160 * 1024, "noise_pw_bwd: %dx%d layer needs %zu bytes of LDS", Cout, Cin, smem);
  hipStream_t st = (hipStream_t)stream;
  bool done = false;
  // NOISE_PW_NO_RECOMPUTE (diagnostic): always read the layer's raw output back
  static const bool recompute_off = getenv("NOISE_PW_NO_RECOMPUTE") != nullptr;
  const bool recompute = coef3 != nullptr && !recompute_off;
#define X(nt, mt)                                                                                                       \
  if constexpr (mt * nt <= 48) {                                                                                        \
    if (!done && KC == mt && MTI == nt) {                                                                               \
      if constexpr (pw_bwd_recompute(mt, nt)) {                                                                         \
        if (recompute) {                                                                                                \
          const size_t sm2 = smem + (size_t)(mt * nt * 256 + 6 * mt * 16) * sizeof(float);                              \
          if (b_y != nullptr) {                                                                                         \
            NOISE_ALLOW_LDS((pw_bwd_kernel<mt, nt, true, true>), sm2);                                                  \
            NOISE_LAUNCH_ROUNDS((pw_bwd_kernel<mt, nt, true, true>), grid, sm2, st, G, A, Bs, w, g_in, E, scratch,      \
                                npix, Cout, Cin);                                                                       \
          } else {                                                                                                      \
            NOISE_ALLOW_LDS((pw_bwd_kernel<mt, nt, false, true>), sm2);                                                 \
            NOISE_LAUNCH_ROUNDS((pw_bwd_kernel<mt, nt, false, true>), grid, sm2, st, G, A, Bs, w, g_in, E, scratch,     \
                                npix, Cout, Cin);                                                                       \
          }                                                                                                             \
          done = true;                                                                                                  \
        }                                                                                                               \
      }                                                                                                                 \
      if (!done && b_y != nullptr) {                                                                                    \
        NOISE_ALLOW_LDS((pw_bwd_kernel<mt, nt, true, false>), smem);                                                    \
        NOISE_LAUNCH_ROUNDS((pw_bwd_kernel<mt, nt, true, false>), grid, smem, st, G, A, Bs, w, g_in, E, scratch, npix,  \
                            Cout, Cin);                                                                                 \
      } else if (!done) {                                                                                               \
        NOISE_ALLOW_LDS((pw_bwd_kernel<mt, nt, false, false>), smem);                                                   \
        NOISE_LAUNCH_ROUNDS((pw_bwd_kernel<mt, nt, false, false>), grid, smem, st, G, A, Bs, w, g_in, E, scratch, npix, \
                            Cout, Cin);                                                                                 \
      }                                                                                                                 \
      done = true;                                                                                                      \
    }                                                                                                                   \
  }
  PW_CASES(X)
#undef X
  NOISE_REQUIRE(done, "noise_pw_bwd: no kernel instance for Cout=%d Cin=%d", Cout, Cin);
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_pw_bwd");
  if (dw != nullptr) {   // NULL: the caller folds the slab later (noise_fold_jobs)
    hipLaunchKernelGGL(pw_slab_reduce_kernel, dim3(noise_cdiv(Cout * Cin, SLAB_COLS)), dim3(256), 0, st, scratch, grid, Cout * Cin, dw);
    NOISE_CHECK_LAUNCH("noise_pw_bwd/reduce");
  }
  return 0;
}
