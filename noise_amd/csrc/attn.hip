// Bottleneck attention block of the denoiser (one post-LN Transformer encoder layer), gfx950.
//
// Replaces train/models.py:207-225 (TransformerEncoder.forward: LayerNorm, layer stack),
// :60-89 (MultiHeadAttention), :27-37 (ScaledDotProductAttention incl. its dropout on the attention
// weights, p = 0.1 in train mode), :102-112 (PositionwiseFeedForward) and :602 (VAD head on token 0).
//
// Tokens are the NHWC pixels of the bottleneck tensor, so token-wise steps work on flat tiles of 64
// tokens (d_model = 64 lives in registers / LDS; these layers are <3 % of the step and VALU-sized),
// and the attention core runs one workgroup per (clip, head) with K and V resident in LDS.
#include "lazy.h"

namespace {

constexpr int D = 64;     // d_model
constexpr int TT = 64;    // tokens per tile
constexpr int XS = D + 4; // LDS row pitch of a token tile
constexpr int WP = D + 4; // LDS row pitch of a weight matrix: the 4 lanes of a token read 4 consecutive rows at once

struct AttnDims {
  long ntok;    // B * n
  int n;        // tokens per clip
  int heads, dk, hd;  // hd = heads*dk (<= 64)
};

// exp as one multiply and v_exp_f32 (the library expf is about twenty instructions; the softmax weights of every kernel
// here, forward and backward, go through this one function, so they stay consistent with each other)
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  return v;
}

// LayerNorm of one token row held in LDS (row of D floats), 4 lanes per token.  Writes xhat in place,
// returns rstd.  Two-pass variance like torch.
__device__ __forceinline__ float ln_row(float* row, int part, float eps) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < D / 4; ++k) s += row[part * (D / 4) + k];
  const float mean = quad_sum(s) * (1.f / D);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < D / 4; ++k) {
    const float c = row[part * (D / 4) + k] - mean;
    v += c * c;
  }
  const float rstd = rsqrtf(quad_sum(v) * (1.f / D) + eps);
#pragma unroll
  for (int k = 0; k < D / 4; ++k) row[part * (D / 4) + k] = (row[part * (D / 4) + k] - mean) * rstd;
  return rstd;
}

// ---- fp32 MFMA (v_mfma_f32_16x16x4_f32) on LDS token tiles.  A tile holds TT = 64 tokens; wave w owns tokens
// 16w..16w+15, the same tokens its threads own in the row-wise (4 lanes per token) steps.  Lane (p = lane & 15,
// q = lane >> 4) feeds A[i = p][k = q] and B[k = q][j = p] and receives D[i = 4q + r][j = p], r = 0..3.

// ot[t][o] = (sum_k xt[t][k] * Wl[o][k] + bias[o] + resid[t][o]), ReLU optional, for o < O and this wave's tokens.
// K = 4*KSTEPS columns of xt must hold finite numbers; Wl is [O][wpitch]; MT = ceil(O / 16).
template <int KSTEPS, int MT>
__device__ __forceinline__ void tile_linear_mfma(const float* xt, int xpitch, const float* Wl, int wpitch, int O, float* ot,
                                                 int opitch, const float* bias, bool relu, const float* resid,
                                                 int rpitch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  f32x4 acc[MT];
  int wrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    wrow[m] = (16 * m + p < O ? 16 * m + p : O - 1) * wpitch + q;   // rows past O repeat the last one (discarded)
  }
  const float* xr = xt + (16 * wave + p) * xpitch + q;
#pragma unroll 4
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const float b = xr[4 * ks];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[wrow[m] + 4 * ks], b, acc[m], 0, 0, 0);
  }
  const int tok = 16 * wave + p;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * m + 4 * q + r;
      if (o < O) {
        float v = acc[m][r] + (bias ? bias[o] : 0.f);
        if (resid) v += resid[tok * rpitch + o];
        ot[tok * opitch + o] = relu ? fmaxf(v, 0.f) : v;
      }
    }
}

// og[(tok0 + t) * O + o] = sum_k xt[t][k] * Wl[o][k] straight from the accumulators (a lane holds 4 consecutive outputs
// of one token = one 16-byte store; O is a multiple of 4), for this wave's tokens below ntok.
template <int KSTEPS, int MT>
__device__ __forceinline__ void tile_linear_mfma_global(const float* xt, int xpitch, const float* Wl, int wpitch, int O,
                                                        float* __restrict__ og, long tok0, long ntok) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  f32x4 acc[MT];
  int wrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    wrow[m] = (16 * m + p < O ? 16 * m + p : O - 1) * wpitch + q;
  }
  const float* xr = xt + (16 * wave + p) * xpitch + q;
#pragma unroll 4
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const float b = xr[4 * ks];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[wrow[m] + 4 * ks], b, acc[m], 0, 0, 0);
  }
  const long tok = tok0 + 16 * wave + p;
  if (tok < ntok) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int o = 16 * m + 4 * q;
      if (o < O) st4(og + tok * O + o, make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
    }
  }
}

// acc[n] += sum_t dy[t][16m + i] * x[t][16n + j] over the 64 tokens of a tile (n < NT): the weight-gradient tiles
// (m, 0..NT-1) of dW = dy^T x.
template <int NT>
__device__ __forceinline__ void tile_outer_mfma(f32x4 (&acc)[NT], const float* dy, int dpitch, int m, const float* x,
                                                int xpitch) {
  const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4;
  const float* dr = dy + q * dpitch + 16 * m + p;
  const float* xr = x + q * xpitch + p;
#pragma unroll 4
  for (int s = 0; s < TT / 4; ++s) {
    const float a = dr[4 * s * dpitch];
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[4 * s * xpitch + 16 * n], acc[n], 0, 0, 0);
  }
}

// dst[(16m + i) * ldk + 16n + j] = acc[n] tile, rows < O and columns < K only
template <int NT>
__device__ __forceinline__ void store_outer_mfma(float* dst, int ldk, const f32x4 (&acc)[NT], int m, int O, int K) {
  const int lane = threadIdx.x & 63, p = lane & 15, q = lane >> 4;
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * m + 4 * q + r, k = 16 * n + p;
      if (o < O && k < K) dst[o * ldk + k] = acc[n][r];
    }
}

// Token tiles move as 16-byte pieces (widths and pitches are multiples of 4 floats), from clamped (always valid)
// addresses with a select afterwards, so the loads of a tile are independent and issued together.
__device__ __forceinline__ void tile_store(float* g, const float* tile, int pitch, long tok0, long ntok, int width) {
  // coalesced copy of a [TT][width] LDS tile (row pitch `pitch`) to global rows of `width` floats
  const int w4 = width >> 2;
#pragma unroll 4
  for (int i = threadIdx.x; i < TT * w4; i += 256) {
    const int t = i / w4, c = 4 * (i - t * w4);
    if (tok0 + t < ntok) st4(g + (tok0 + t) * width + c, ld4(tile + t * pitch + c));
  }
}

__device__ __forceinline__ void tile_load(float* tile, int pitch, const float* g, long tok0, long ntok, int width) {
  const int w4 = width >> 2;
#pragma unroll 4
  for (int i = threadIdx.x; i < TT * w4; i += 256) {
    const int t = i / w4, c = 4 * (i - t * w4);
    const bool in = tok0 + t < ntok;
    const float4 v = ld4(g + (in ? tok0 + t : ntok - 1) * width + c);
    st4(tile + t * pitch + c, in ? v : f4(0.f));
  }
}

// like tile_load, and columns [width, padded) are zeroed (they feed MFMA k-steps)
__device__ __forceinline__ void tile_load_padded(float* tile, int pitch, const float* g, long tok0, long ntok, int width,
                                                 int padded) {
  const int p4 = padded >> 2;
#pragma unroll 4
  for (int i = threadIdx.x; i < TT * p4; i += 256) {
    const int t = i / p4, c = 4 * (i - t * p4);
    const bool in = tok0 + t < ntok && c < width;
    const float4 v = ld4(g + (tok0 + t < ntok ? tok0 + t : ntok - 1) * width + (c < width ? c : 0));
    st4(tile + t * pitch + c, in ? v : f4(0.f));
  }
}

// ------------------------------------------------------------------------------------------------
// A: x0 (lazy) -> xhat0, rstd0 -> x1 = g0*xhat0 + b0 -> qkv = x1 [Wq;Wk;Wv]^T
__global__ __launch_bounds__(256) void attn_pre_fwd_kernel(LazySrc X, const float* __restrict__ ln_w,
                                                           const float* __restrict__ ln_b, const float* __restrict__ wq,
                                                           const float* __restrict__ wk, const float* __restrict__ wv,
                                                           float* __restrict__ xhat0, float* __restrict__ rstd0,
                                                           float* __restrict__ qkv, AttnDims d) {
  extern __shared__ float lds[];
  const int O = 3 * d.hd;
  float* Wl = lds;                  // [O][WP]
  float* xs = Wl + O * WP;          // [TT][XS]
  float* lw = xs + TT * XS;         // [2][D]
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3;
  for (int i = tid; i < O * D; i += 256) {
    const int o = i / D, k = i - o * D;
    const int which = o / d.hd, oo = o - which * d.hd;
    const float* w = which == 0 ? wq : (which == 1 ? wk : wv);
    Wl[o * WP + k] = w[oo * D + k];
  }
  for (int i = tid; i < D; i += 256) {
    lw[i] = ln_w[i];
    lw[D + i] = ln_b[i];
  }
  // the lazy transform of the bottleneck tensor: scale / shift of this thread's 4 columns (identity when absent) and
  // the ReLU6 clamp as bounds, so the tile load has no branch around it
  const int c4 = 4 * (tid & 15);
  float4 xsc = f4(1.f), xsh = f4(0.f);
  if (X.coef) {
    xsc = ld4(X.coef + c4);
    xsh = ld4(X.coef + D + c4);
  }
  const float xlo = X.relu6 ? 0.f : -INFINITY, xhi = X.relu6 ? 6.f : INFINITY;
  for (long tok0 = (long)blockIdx.x * TT; tok0 < d.ntok; tok0 += (long)gridDim.x * TT) {
    __syncthreads();
    {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {       // row tt = tid/16 + 16u, columns c4..c4+3: 16-byte pieces, clamped addresses
        const long tok = tok0 + (tid >> 4) + 16 * u;
        v[u] = ld4(X.y + (tok < d.ntok ? tok : d.ntok - 1) * D + c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tt = (tid >> 4) + 16 * u;
        const float4 a = clamp4(fma4(v[u], xsc, xsh), xlo, xhi);
        st4(xs + tt * XS + c4, tok0 + tt < d.ntok ? a : f4(0.f));
      }
    }
    __syncthreads();
    float* row = xs + t * XS;
    const float rstd = ln_row(row, part, 1e-6f);
    const bool valid = tok0 + t < d.ntok;
    if (valid) {
      if (part == 0) rstd0[tok0 + t] = rstd;
#pragma unroll
      for (int k = 0; k < D / 16; ++k) st4(xhat0 + (tok0 + t) * D + part * (D / 4) + 4 * k, ld4(row + part * (D / 4) + 4 * k));
    }
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      row[c] = fmaf(row[c], lw[c], lw[D + c]);
    }
    __syncthreads();  // also orders the 4 lanes of a token (same wave) and the weight staging
    tile_linear_mfma_global<D / 4, 12>(xs, XS, Wl, WP, O, qkv, tok0, d.ntok);
  }
}

// ------------------------------------------------------------------------------------------------
// B: attention core, one workgroup per (clip, head); one query per thread, K/V/Q of the head in LDS.
__device__ __forceinline__ bool keep_elem(const unsigned char* keep, unsigned long long seed, float p_drop, long idx) {
  if (keep) return keep[idx] != 0;
  if (p_drop <= 0.f) return true;
  // counter-based hash of (seed, element index): two 32-bit multiply-xorshift rounds (murmur3 finaliser) -> uniform
  // in [0,1).  32-bit arithmetic keeps it to a dozen VALU instructions per attention weight.
  unsigned x = (unsigned)idx ^ (unsigned)seed;
  x *= 0x9E3779B1u;
  x ^= (unsigned)(seed >> 32) + (unsigned)((unsigned long long)idx >> 32) * 0x85EBCA6Bu;
  x ^= x >> 16;
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  x ^= x >> 16;
  return (float)(x >> 8) * (1.f / 16777216.f) >= p_drop;
}

constexpr int DKMAX = 16;
constexpr int ATTN_CHUNK_FLOATS = 6144;   // per staged array: two of them + three per-token vectors stay under 64 KiB up to n ~ 1300

// Clips of any length: the rows the inner loops walk (keys and values for a query thread, queries and output
// gradients for a key thread) pass through LDS in chunks of `kc` rows, padded to a multiple of 4 floats so a row is
// read with 128-bit broadcasts.  The order of every sum is the plain ascending one whatever the chunk size.
__device__ __forceinline__ int attn_chunk_rows(int n, int dk) {
  const int dkp = (dk + 3) & ~3, kc = ATTN_CHUNK_FLOATS / dkp;
  return n < kc ? n : kc;
}

// dst[j][c] = src[(row0 + j) * ld + col + c] / div for j < cnt, c < dk; zero for the padding columns.  Barriers on
// both sides: every thread of the workgroup must call it.
__device__ __forceinline__ void attn_stage(float* dst, const float* __restrict__ src, long row0, int cnt, int ld, int col,
                                           int dk, int dkp, float div) {
  __syncthreads();
  for (int idx = threadIdx.x; idx < cnt * dkp; idx += 256) {
    const int j = idx / dkp, c = idx - j * dkp;
    dst[idx] = c < dk ? src[(row0 + j) * ld + col + c] / div : 0.f;
  }
  __syncthreads();
}

__device__ __forceinline__ float attn_dot(const float (&a)[DKMAX], const float* row, int dk4) {
  float s = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < DKMAX / 4; ++c4)
    if (c4 < dk4) {
      const float4 r = *reinterpret_cast<const float4*>(row + 4 * c4);
      s = fmaf(a[4 * c4 + 3], r.w, fmaf(a[4 * c4 + 2], r.z, fmaf(a[4 * c4 + 1], r.y, fmaf(a[4 * c4], r.x, s))));
    }
  return s;
}

__device__ __forceinline__ void attn_axpy(float (&acc)[DKMAX], float a, const float* row, int dk4) {
#pragma unroll
  for (int c4 = 0; c4 < DKMAX / 4; ++c4)
    if (c4 < dk4) {
      const float4 r = *reinterpret_cast<const float4*>(row + 4 * c4);
      acc[4 * c4] = fmaf(a, r.x, acc[4 * c4]);
      acc[4 * c4 + 1] = fmaf(a, r.y, acc[4 * c4 + 1]);
      acc[4 * c4 + 2] = fmaf(a, r.z, acc[4 * c4 + 2]);
      acc[4 * c4 + 3] = fmaf(a, r.w, acc[4 * c4 + 3]);
    }
}

__global__ __launch_bounds__(256) void attn_core_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                            float* __restrict__ ml, const unsigned char* __restrict__ keep,
                                                            unsigned long long seed, float p_drop, AttnDims d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int dk4 = (dk + 3) >> 2, dkp = 4 * dk4, kc = attn_chunk_rows(n, dk);
  const bool one_chunk = kc >= n;
  const int b = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
  float* ks = lds;              // [kc][dkp]
  float* vs = ks + kc * dkp;    // [kc][dkp]
  const long base = (long)b * n;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  if (one_chunk) {
    attn_stage(ks, qkv, base, n, O, d.hd + h * dk, dk, dkp, 1.f);
    attn_stage(vs, qkv, base, n, O, 2 * d.hd + h * dk, dk, dkp, 1.f);
  }
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool act = i < n;
    float q[DKMAX];
#pragma unroll
    for (int c = 0; c < DKMAX; ++c) q[c] = (act && c < dk) ? qkv[(base + i) * O + h * dk + c] / temp : 0.f;
    float m = -INFINITY;
    for (int c0 = 0; c0 < n; c0 += kc) {
      const int cnt = min(kc, n - c0);
      if (!one_chunk) attn_stage(ks, qkv, base + c0, cnt, O, d.hd + h * dk, dk, dkp, 1.f);
      if (act)
        for (int j = 0; j < cnt; ++j) m = fmaxf(m, attn_dot(q, ks + j * dkp, dk4));
    }
    float l = 0.f, acc[DKMAX];
#pragma unroll
    for (int c = 0; c < DKMAX; ++c) acc[c] = 0.f;
    const long mrow = (((long)b * d.heads + h) * n + i) * n;
    for (int c0 = 0; c0 < n; c0 += kc) {
      const int cnt = min(kc, n - c0);
      if (!one_chunk) {
        attn_stage(ks, qkv, base + c0, cnt, O, d.hd + h * dk, dk, dkp, 1.f);
        attn_stage(vs, qkv, base + c0, cnt, O, 2 * d.hd + h * dk, dk, dkp, 1.f);
      }
      if (act)
        for (int j = 0; j < cnt; ++j) {
          const float e = exp_hw(attn_dot(q, ks + j * dkp, dk4) - m);
          l += e;
          if (keep_elem(keep, seed, p_drop, mrow + c0 + j)) attn_axpy(acc, e, vs + j * dkp, dk4);
        }
    }
    if (act) {
      const float sc = inv_keep / l;
#pragma unroll
      for (int c = 0; c < DKMAX; ++c)
        if (c < dk) o[(base + i) * d.hd + h * dk + c] = acc[c] * sc;
      if (ml) {
        ml[(((long)b * d.heads + h) * n + i) * 2] = m;
        ml[(((long)b * d.heads + h) * n + i) * 2 + 1] = l;
      }
    }
  }
}

// backward of the core: d_qkv from d_o.  Phase 1 (thread = query): D_i and dq_i.  Phase 2 (thread = key): dk_j, dv_j.
__global__ __launch_bounds__(256) void attn_core_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                            const float* __restrict__ ml, const unsigned char* __restrict__ keep,
                                                            unsigned long long seed, float p_drop, float* __restrict__ d_qkv,
                                                            AttnDims d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int dk4 = (dk + 3) >> 2, dkp = 4 * dk4, kc = attn_chunk_rows(n, dk);
  const bool one_chunk = kc >= n;
  const int b = blockIdx.x / d.heads, h = blockIdx.x % d.heads;
  float* sa = lds;              // [kc][dkp]  keys, then queries (divided by the temperature)
  float* sb = sa + kc * dkp;    // [kc][dkp]  values, then output gradients
  float* mm = sb + kc * dkp;    // [n] max
  float* ll = mm + n;           // [n] 1/l
  float* Dd = ll + n;           // [n] D_i
  const long base = (long)b * n;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  for (int i = threadIdx.x; i < n; i += 256) {
    mm[i] = ml[(((long)b * d.heads + h) * n + i) * 2];
    ll[i] = 1.f / ml[(((long)b * d.heads + h) * n + i) * 2 + 1];
  }
  const int kcol = d.hd + h * dk, vcol = 2 * d.hd + h * dk;
  if (one_chunk) {
    attn_stage(sa, qkv, base, n, O, kcol, dk, dkp, 1.f);
    attn_stage(sb, qkv, base, n, O, vcol, dk, dkp, 1.f);
  }
  for (int i0 = 0; i0 < n; i0 += 256) {
    const int i = i0 + threadIdx.x;
    const bool act = i < n;
    const int ic = act ? i : n - 1;
    float q[DKMAX], go[DKMAX];
#pragma unroll
    for (int c = 0; c < DKMAX; ++c) {
      q[c] = (act && c < dk) ? qkv[(base + i) * O + h * dk + c] / temp : 0.f;
      go[c] = (act && c < dk) ? d_o[(base + i) * d.hd + h * dk + c] : 0.f;
    }
    const long mrow = (((long)b * d.heads + h) * n + ic) * n;
    float Di = 0.f;
    for (int c0 = 0; c0 < n; c0 += kc) {
      const int cnt = min(kc, n - c0);
      if (!one_chunk) {
        attn_stage(sa, qkv, base + c0, cnt, O, kcol, dk, dkp, 1.f);
        attn_stage(sb, qkv, base + c0, cnt, O, vcol, dk, dkp, 1.f);
      }
      if (act) {
        const float mi = mm[i], li = ll[i];
        for (int j = 0; j < cnt; ++j) {
          const float sc = attn_dot(q, sa + j * dkp, dk4);
          float dp = attn_dot(go, sb + j * dkp, dk4);
          if (!keep_elem(keep, seed, p_drop, mrow + c0 + j)) dp = 0.f;
          Di = fmaf(exp_hw(sc - mi) * li, dp * inv_keep, Di);
        }
      }
    }
    float dq[DKMAX];
#pragma unroll
    for (int c = 0; c < DKMAX; ++c) dq[c] = 0.f;
    for (int c0 = 0; c0 < n; c0 += kc) {
      const int cnt = min(kc, n - c0);
      if (!one_chunk) {
        attn_stage(sa, qkv, base + c0, cnt, O, kcol, dk, dkp, 1.f);
        attn_stage(sb, qkv, base + c0, cnt, O, vcol, dk, dkp, 1.f);
      }
      if (act) {
        const float mi = mm[i], li = ll[i];
        for (int j = 0; j < cnt; ++j) {
          const float sc = attn_dot(q, sa + j * dkp, dk4);
          float dp = attn_dot(go, sb + j * dkp, dk4);
          if (!keep_elem(keep, seed, p_drop, mrow + c0 + j)) dp = 0.f;
          const float p = exp_hw(sc - mi) * li;
          attn_axpy(dq, p * (dp * inv_keep - Di), sa + j * dkp, dk4);
        }
      }
    }
    if (act) {
      Dd[i] = Di;
#pragma unroll
      for (int c = 0; c < DKMAX; ++c)
        if (c < dk) d_qkv[(base + i) * O + h * dk + c] = dq[c] / temp;
    }
  }
  if (one_chunk) {
    attn_stage(sa, qkv, base, n, O, h * dk, dk, dkp, temp);
    attn_stage(sb, d_o, base, n, d.hd, h * dk, dk, dkp, 1.f);
  } else {
    __syncthreads();      // Dd complete before any key thread reads it
  }
  for (int j0 = 0; j0 < n; j0 += 256) {
    const int j = j0 + threadIdx.x;
    const bool act = j < n;
    float kr[DKMAX], vr[DKMAX], dkk[DKMAX], dvv[DKMAX];
#pragma unroll
    for (int c = 0; c < DKMAX; ++c) {
      kr[c] = (act && c < dk) ? qkv[(base + j) * O + kcol + c] : 0.f;
      vr[c] = (act && c < dk) ? qkv[(base + j) * O + vcol + c] : 0.f;
      dkk[c] = dvv[c] = 0.f;
    }
    for (int c0 = 0; c0 < n; c0 += kc) {
      const int cnt = min(kc, n - c0);
      if (!one_chunk) {
        attn_stage(sa, qkv, base + c0, cnt, O, h * dk, dk, dkp, temp);
        attn_stage(sb, d_o, base + c0, cnt, d.hd, h * dk, dk, dkp, 1.f);
      }
      if (act)
        for (int ii = 0; ii < cnt; ++ii) {
          const int i = c0 + ii;
          const float sc = attn_dot(kr, sa + ii * dkp, dk4);
          const float dp = attn_dot(vr, sb + ii * dkp, dk4);
          const bool kp = keep_elem(keep, seed, p_drop, (((long)b * d.heads + h) * n + i) * n + j);
          const float p = exp_hw(sc - mm[i]) * ll[i];
          attn_axpy(dkk, p * ((kp ? dp * inv_keep : 0.f) - Dd[i]), sa + ii * dkp, dk4);
          attn_axpy(dvv, kp ? p * inv_keep : 0.f, sb + ii * dkp, dk4);
        }
    }
    if (act) {
#pragma unroll
      for (int c = 0; c < DKMAX; ++c)
        if (c < dk) {
          d_qkv[(base + j) * O + kcol + c] = dkk[c];
          d_qkv[(base + j) * O + vcol + c] = dvv[c];
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// B': the same core for clips of at most 64 tokens (the training segments: 55): one WAVE per (clip, head), lane =
// query (phase 1) or key (phase 2).  A lane keeps its whole row of attention weights in registers, so the
// scores are computed once, K / V / Q / dO rows are read from LDS as float4 broadcasts (rows padded to DK4*4
// floats), and the backward pass hands p and ds from the query lanes to the key lanes through one n x n LDS
// matrix instead of recomputing them.  Summation orders are those of the kernels above.
constexpr int AW = 64;

template <int DK4>
__device__ __forceinline__ float dot_row(const float (&a)[DK4 * 4], const float* row) {
  float s = 0.f;
#pragma unroll
  for (int c4 = 0; c4 < DK4; ++c4) {
    const float4 k = ld4(row + 4 * c4);
    s = fmaf(a[4 * c4 + 0], k.x, s);
    s = fmaf(a[4 * c4 + 1], k.y, s);
    s = fmaf(a[4 * c4 + 2], k.z, s);
    s = fmaf(a[4 * c4 + 3], k.w, s);
  }
  return s;
}

template <int DK4>
__device__ __forceinline__ void axpy_row(float a, const float* row, float (&acc)[DK4 * 4]) {
#pragma unroll
  for (int c4 = 0; c4 < DK4; ++c4) {
    const float4 v = ld4(row + 4 * c4);
    acc[4 * c4 + 0] = fmaf(a, v.x, acc[4 * c4 + 0]);
    acc[4 * c4 + 1] = fmaf(a, v.y, acc[4 * c4 + 1]);
    acc[4 * c4 + 2] = fmaf(a, v.z, acc[4 * c4 + 2]);
    acc[4 * c4 + 3] = fmaf(a, v.w, acc[4 * c4 + 3]);
  }
}

// stage rows [n][dk] of one head (column offset col0 of a row-major matrix with row stride ld) as [n][DK4*4]
template <int DK4>
__device__ __forceinline__ void stage_head(float* dst, const float* __restrict__ src, long base, int ld, int col0, int n,
                                           int dk, float scale_div) {
  constexpr int DKP = DK4 * 4;
  if (dk == DKP && ((ld | col0) & 3) == 0) {
    // whole 16-byte pieces: all of a lane's loads are independent and issued together
#pragma unroll 4
    for (int i = threadIdx.x; i < n * DK4; i += AW) {
      const int j = i / DK4, c4 = i - j * DK4;
      float4 v = ld4(src + (base + j) * ld + col0 + 4 * c4);
      if (scale_div != 0.f) v = make_float4(v.x / scale_div, v.y / scale_div, v.z / scale_div, v.w / scale_div);
      st4(dst + 4 * i, v);
    }
    return;
  }
  // any d_k (the reference's 64 // 6 = 10): scalar loads from clamped addresses, padding selected afterwards, so the
  // loads of several trips are in flight together
#pragma unroll 4
  for (int i = threadIdx.x; i < n * DKP; i += AW) {
    const int j = i / DKP, c = i - j * DKP;
    float v = src[(base + j) * ld + col0 + (c < dk ? c : dk - 1)];
    if (scale_div != 0.f) v = v / scale_div;
    dst[i] = c < dk ? v : 0.f;
  }
}

// a lane's row of one head to / from global memory (columns >= dk are padding)
template <int DK4>
__device__ __forceinline__ void store_head_row(float* dst, const float (&v)[DK4 * 4], int dk, float div) {
  if (dk == DK4 * 4 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
#pragma unroll
    for (int c4 = 0; c4 < DK4; ++c4)
      st4(dst + 4 * c4, div != 0.f ? make_float4(v[4 * c4] / div, v[4 * c4 + 1] / div, v[4 * c4 + 2] / div, v[4 * c4 + 3] / div)
                                    : make_float4(v[4 * c4], v[4 * c4 + 1], v[4 * c4 + 2], v[4 * c4 + 3]));
    return;
  }
#pragma unroll
  for (int c = 0; c < DK4 * 4; ++c)
    if (c < dk) dst[c] = div != 0.f ? v[c] / div : v[c];
}

// DROP: 0 = no dropout, 1 = explicit keep mask, 2 = counter-based RNG.  A compile-time mode keeps the per-weight code
// of the wave kernels free of the run-time tests in keep_elem.
template <int DROP>
__device__ __forceinline__ bool keep_at(const unsigned char* keep, unsigned long long seed, float p_drop, long idx) {
  if (DROP == 0) return true;
  if (DROP == 1) return keep[idx] != 0;
  return keep_elem(nullptr, seed, p_drop, idx);
}

template <int DK4, int DROP>
__global__ __launch_bounds__(AW) void attn_core_fwd_wave_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                float* __restrict__ ml,
                                                                const unsigned char* __restrict__ keep,
                                                                unsigned long long seed, float p_drop, AttnDims d) {
  extern __shared__ float lds[];
  constexpr int DKP = DK4 * 4;
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  // the heads of a clip read the same token rows (40-byte pieces of 720-byte rows): they are placed on one XCD
  const int vblk = xcd_chunked_block();
  if (vblk >= (int)(d.ntok / n) * d.heads) return;
  const int b = vblk / d.heads, h = vblk % d.heads;
  float* ks = lds;             // [n][DKP]
  float* vs = ks + n * DKP;    // [n][DKP]
  const long base = (long)b * n;
  stage_head<DK4>(ks, qkv, base, O, d.hd + h * dk, n, dk, 0.f);
  stage_head<DK4>(vs, qkv, base, O, 2 * d.hd + h * dk, n, dk, 0.f);
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= n) return;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  float q[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) {
    const float v = qkv[(base + i) * O + h * dk + (c < dk ? c : dk - 1)] / temp;   // clamped, then selected: no branch
    q[c] = c < dk ? v : 0.f;
  }
  float s[AW];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < AW; ++j) {
    if (j < n) {
      s[j] = dot_row<DK4>(q, ks + j * DKP);
      m = fmaxf(m, s[j]);
    }
  }
  float l = 0.f, acc[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) acc[c] = 0.f;
  const long mrow = (((long)b * d.heads + h) * n + i) * n;
#pragma unroll
  for (int j = 0; j < AW; ++j) {
    if (j < n) {
      const float e = exp_hw(s[j] - m);
      l += e;
      if (keep_at<DROP>(keep, seed, p_drop, mrow + j)) axpy_row<DK4>(e, vs + j * DKP, acc);
    }
  }
  const float sc = inv_keep / l;
#pragma unroll
  for (int c = 0; c < DKP; ++c)
    if (c < dk) o[(base + i) * d.hd + h * dk + c] = acc[c] * sc;
  if (ml) {
    ml[(((long)b * d.heads + h) * n + i) * 2] = m;
    ml[(((long)b * d.heads + h) * n + i) * 2 + 1] = l;
  }
}

template <int DK4, int DROP>
__global__ __launch_bounds__(AW) void attn_core_bwd_wave_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                const float* __restrict__ ml,
                                                                const unsigned char* __restrict__ keep,
                                                                unsigned long long seed, float p_drop,
                                                                float* __restrict__ d_qkv, AttnDims d) {
  extern __shared__ float lds[];
  constexpr int DKP = DK4 * 4;
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int vblk = xcd_chunked_block();      // see attn_core_fwd_wave_kernel
  if (vblk >= (int)(d.ntok / n) * d.heads) return;
  const int b = vblk / d.heads, h = vblk % d.heads;
  const int mp = n | 1;            // odd pitch: row-wise writes and column-wise reads are both conflict-free
  float* qs = lds;                 // [n][DKP], already divided by the temperature
  float* ks = qs + n * DKP;
  float* vs = ks + n * DKP;
  float* dos = vs + n * DKP;
  float* M = dos + n * DKP;        // [n][mp]: first the kept, rescaled weights (for dV), then ds (for dK)
  const long base = (long)b * n;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  stage_head<DK4>(qs, qkv, base, O, h * dk, n, dk, temp);
  stage_head<DK4>(ks, qkv, base, O, d.hd + h * dk, n, dk, 0.f);
  stage_head<DK4>(vs, qkv, base, O, 2 * d.hd + h * dk, n, dk, 0.f);
  stage_head<DK4>(dos, d_o, base, d.hd, h * dk, n, dk, 0.f);
  __syncthreads();
  const int i = threadIdx.x;
  const bool act = i < n;
  const int ic = act ? i : 0;
  const long mlrow = (((long)b * d.heads + h) * n + ic);
  const float mi = ml[mlrow * 2], li = 1.f / ml[mlrow * 2 + 1];
  const long mrow = mlrow * n;
  float q[DKP], dout[DKP];
#pragma unroll
  for (int c4 = 0; c4 < DK4; ++c4) {
    const float4 a = ld4(qs + ic * DKP + 4 * c4), g = ld4(dos + ic * DKP + 4 * c4);
    q[4 * c4] = a.x; q[4 * c4 + 1] = a.y; q[4 * c4 + 2] = a.z; q[4 * c4 + 3] = a.w;
    dout[4 * c4] = g.x; dout[4 * c4 + 1] = g.y; dout[4 * c4 + 2] = g.z; dout[4 * c4 + 3] = g.w;
  }
  // ---- phase 1 (lane = query): p_ij, D_i; kept weights to M
  float pr[AW];
  unsigned long long kept = 0ull;
  float Di = 0.f;
#pragma unroll
  for (int j = 0; j < AW; ++j) {
    if (j < n) {
      const float sj = dot_row<DK4>(q, ks + j * DKP);
      const bool kp = keep_at<DROP>(keep, seed, p_drop, mrow + j);
      const float dp = kp ? dot_row<DK4>(dout, vs + j * DKP) : 0.f;
      pr[j] = exp_hw(sj - mi) * li;
      Di = fmaf(pr[j], dp * inv_keep, Di);
      kept |= kp ? (1ull << j) : 0ull;
      if (act) M[i * mp + j] = kp ? pr[j] * inv_keep : 0.f;
    }
  }
  __syncthreads();
  // ---- phase 2a (lane = key): dV_j = sum_i pk_ij dO_i
  float dvv[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) dvv[c] = 0.f;
  for (int r = 0; r < n; ++r) axpy_row<DK4>(M[r * mp + ic], dos + r * DKP, dvv);
  __syncthreads();
  // ---- phase 1b (lane = query): ds_ij, dQ_i; ds to M
  float dq[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) dq[c] = 0.f;
#pragma unroll
  for (int j = 0; j < AW; ++j) {
    if (j < n) {
      const float dp = ((kept >> j) & 1ull) ? dot_row<DK4>(dout, vs + j * DKP) : 0.f;
      const float ds = pr[j] * (dp * inv_keep - Di);
      axpy_row<DK4>(ds, ks + j * DKP, dq);
      if (act) M[i * mp + j] = ds;
    }
  }
  __syncthreads();
  // ---- phase 2b (lane = key): dK_j = sum_i ds_ij q_i
  float dkk[DKP];
#pragma unroll
  for (int c = 0; c < DKP; ++c) dkk[c] = 0.f;
  for (int r = 0; r < n; ++r) axpy_row<DK4>(M[r * mp + ic], qs + r * DKP, dkk);
  if (act) {
    float* row = d_qkv + (base + i) * O + h * dk;
    store_head_row<DK4>(row, dq, dk, temp);
    store_head_row<DK4>(row + d.hd, dkk, dk, 0.f);
    store_head_row<DK4>(row + 2 * d.hd, dvv, dk, 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of the core for segments (n <= 64 tokens) on the matrix cores: one workgroup of four waves per (clip, head),
// q (already divided by the temperature), k, v and dO of the head in LDS (rows of 36 floats: the 64 lanes of every
// operand read fall on 64 banks).  The score tiles stay in the registers the fp32 MFMA leaves them in, which is the
// operand layout of the next product:
//   pass Q  (a wave owns a tile of 16 queries): S^T = K Q^T and dP^T = V dO^T give a lane 4 keys of one query; with
//           w = p * dp (dropout applied) the row term D_i = sum_j w_ij needs all keys, so the pass accumulates
//           A1 = K^T W^T and A2 = K^T P^T and forms dQ^T = A1 - D A2 at the end;
//   pass KV (a wave owns a tile of 16 keys): S = Q K^T and dP = dO V^T give a lane 4 queries of one key;
//           dV^T += dO^T Pk and dK^T += Q^T dS with k = the lane's own queries.
// The softmax statistics are the forward's (ml), the keep decisions come from the same keep_at as everywhere else.
// (The thread-per-query form spends its time on the LDS pipe: ~1100 LDS instructions per wave; 0.233 ms per step.)
constexpr int ASG_P = 36;

__device__ __forceinline__ float sum_over_quarters(float v) {   // over lanes p, p + 16, p + 32, p + 48
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float max_over_quarters(float v) {   // over lanes p, p + 16, p + 32, p + 48
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Forward of the core for segments, same arrangement: a wave owns 16 queries; S^T = K Q^T for all (at most four) key
// tiles, the row maximum and sum as in the thread-per-query kernel (the sum runs over all keys, dropout only removes
// terms of the numerator), then O^T += V^T P^T with k = the lane's own keys.
template <int DK4, int DROP>
__global__ __launch_bounds__(256) void attn_core_fwd_seg_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                     float* __restrict__ ml,
                                                                     const unsigned char* __restrict__ keep,
                                                                     unsigned long long seed, float p_drop, AttnDims d) {
  __shared__ float ks[64 * ASG_P], vs[64 * ASG_P];
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int vblk = xcd_chunked_block();
  if (vblk >= (int)(d.ntok / n) * d.heads) return;
  const int b = vblk / d.heads, h = vblk % d.heads;
  const long base = (long)b * n;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  const long bhn = ((long)b * d.heads + h) * n;
  for (int e = threadIdx.x; e < 64 * 16; e += 256) {
    const int j = e >> 4, c = e & 15;
    const bool in = j < n && c < dk;
    const float* row = qkv + (base + (j < n ? j : n - 1)) * O + h * dk + (c < dk ? c : 0);
    const float kv = row[d.hd], vv = row[2 * d.hd];
    ks[j * ASG_P + c] = in ? kv : 0.f;
    vs[j * ASG_P + c] = in ? vv : 0.f;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, qd = lane >> 4;
  const int q0 = 16 * wave, qi = q0 + p, qc = qi < n ? qi : n - 1;
  float qreg[DK4];
#pragma unroll
  for (int kk = 0; kk < DK4; ++kk) {
    const int c = 4 * kk + qd;
    const float v = qkv[(base + qc) * O + h * dk + (c < dk ? c : 0)] / temp;
    qreg[kk] = c < dk ? v : 0.f;
  }
  __syncthreads();
  if (q0 >= n) return;
  const int ntile = (n + 15) >> 4;
  f32x4 st[4];
  float m = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    st[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt < ntile) {
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk)
        st[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ks[(16 * kt + p) * ASG_P + 4 * kk + qd], qreg[kk], st[kt], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * qd + r < n) m = fmaxf(m, st[kt][r]);
    }
  }
  m = max_over_quarters(m);
  const long mrow = (bhn + qc) * n;
  float l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    if (kt < ntile) {
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * qd + r;
        const bool ok = key < n;
        const float e = ok ? exp_hw(st[kt][r] - m) : 0.f;
        l += e;
        pr[r] = (ok && keep_at<DROP>(keep, seed, p_drop, mrow + (ok ? key : 0))) ? e : 0.f;
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[(16 * kt + 4 * qd + s2) * ASG_P + p], pr[s2], acc, 0, 0, 0);
    }
  }
  l = sum_over_quarters(l);
  if (qi < n) {
    const float sc = inv_keep / l;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * qd + r;
      if (c < dk) o[(base + qi) * d.hd + h * dk + c] = acc[r] * sc;
    }
    if (ml && qd == 0) {
      ml[(bhn + qi) * 2] = m;
      ml[(bhn + qi) * 2 + 1] = l;
    }
  }
}

template <int DK4, int DROP>
__global__ __launch_bounds__(256) void attn_core_bwd_seg_mfma_kernel(const float* __restrict__ qkv, const float* __restrict__ d_o,
                                                                     const float* __restrict__ ml,
                                                                     const unsigned char* __restrict__ keep,
                                                                     unsigned long long seed, float p_drop,
                                                                     float* __restrict__ d_qkv, AttnDims d) {
  __shared__ float qs[64 * ASG_P], ks[64 * ASG_P], vs[64 * ASG_P], gs[64 * ASG_P];
  __shared__ float sm[64], srl[64], sD[64];
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int vblk = xcd_chunked_block();      // see attn_core_fwd_wave_kernel
  if (vblk >= (int)(d.ntok / n) * d.heads) return;
  const int b = vblk / d.heads, h = vblk % d.heads;
  const long base = (long)b * n;
  const float temp = sqrtf((float)dk);
  const float inv_keep = 1.f / (1.f - p_drop);
  const long bhn = ((long)b * d.heads + h) * n;
  for (int e = threadIdx.x; e < 64 * 16; e += 256) {   // 16 columns: every k-step and the one 16-wide output tile
    const int j = e >> 4, c = e & 15;
    const bool in = j < n && c < dk;
    const float* row = qkv + (base + (j < n ? j : n - 1)) * O + h * dk + (c < dk ? c : 0);
    const float qv = row[0], kv = row[d.hd], vv = row[2 * d.hd];
    const float gv = d_o[(base + (j < n ? j : n - 1)) * d.hd + h * dk + (c < dk ? c : 0)];
    qs[j * ASG_P + c] = in ? qv / temp : 0.f;
    ks[j * ASG_P + c] = in ? kv : 0.f;
    vs[j * ASG_P + c] = in ? vv : 0.f;
    gs[j * ASG_P + c] = in ? gv : 0.f;
  }
  if (threadIdx.x < 64) {
    const int j = threadIdx.x < n ? threadIdx.x : n - 1;
    sm[threadIdx.x] = ml[(bhn + j) * 2];
    srl[threadIdx.x] = 1.f / ml[(bhn + j) * 2 + 1];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, qd = lane >> 4;
  const int ntile = (n + 15) >> 4;
  // ---- pass Q
  {
    const int q0 = 16 * wave;
    if (q0 < n) {
      const int qi = q0 + p, qc = qi < n ? qi : n - 1;
      float qreg[DK4], greg[DK4];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) {
        qreg[kk] = qs[(q0 + p) * ASG_P + 4 * kk + qd];
        greg[kk] = gs[(q0 + p) * ASG_P + 4 * kk + qd];
      }
      const float mi = sm[qc], li = srl[qc];
      const long mrow = (bhn + qc) * n;
      f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
      float dpart = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (kt >= ntile) break;   // at most four tiles: unrolled so that a tile's products start under the previous tile's arithmetic
        const int key0 = 16 * kt;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(ks[(key0 + p) * ASG_P + 4 * kk + qd], qreg[kk], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[(key0 + p) * ASG_P + 4 * kk + qd], greg[kk], dp, 0, 0, 0);
        }
        float w1[4], w2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = key0 + 4 * qd + r;
          const bool ok = key < n;
          const float pj = ok ? exp_hw(st[r] - mi) * li : 0.f;
          const bool kept = ok && keep_at<DROP>(keep, seed, p_drop, mrow + (ok ? key : 0));
          w1[r] = kept ? pj * (dp[r] * inv_keep) : 0.f;
          w2[r] = pj;
          dpart += w1[r];
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          const float kb = ks[(key0 + 4 * qd + s2) * ASG_P + p];
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb, w1[s2], a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(kb, w2[s2], a2, 0, 0, 0);
        }
      }
      const float Di = sum_over_quarters(dpart);
      if (qi < n) {
        float* row = d_qkv + (base + qi) * O + h * dk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * qd + r;
          if (c < dk) row[c] = (a1[r] - Di * a2[r]) / temp;
        }
        if (qd == 0) sD[qi] = Di;
      }
    }
  }
  __syncthreads();
  // ---- pass KV
  {
    const int key0 = 16 * wave;
    if (key0 < n) {
      const int kj = key0 + p;
      const bool kok = kj < n;
      float kreg[DK4], vreg[DK4];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) {
        kreg[kk] = ks[(key0 + p) * ASG_P + 4 * kk + qd];
        vreg[kk] = vs[(key0 + p) * ASG_P + 4 * kk + qd];
      }
      f32x4 ak = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        if (qt >= ntile) break;
        const int q0 = 16 * qt;
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(q0 + p) * ASG_P + 4 * kk + qd], kreg[kk], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(q0 + p) * ASG_P + 4 * kk + qd], vreg[kk], dp, 0, 0, 0);
        }
        float pm[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = q0 + 4 * qd + r;
          const bool ok = qq < n && kok;
          const int qc = qq < n ? qq : n - 1;
          const float pj = ok ? exp_hw(st[r] - sm[qc]) * srl[qc] : 0.f;
          const bool kept = ok && keep_at<DROP>(keep, seed, p_drop, (bhn + qc) * n + (kok ? kj : 0));
          pm[r] = kept ? pj * inv_keep : 0.f;
          const float dpk = kept ? dp[r] * inv_keep : 0.f;
          ds[r] = ok ? pj * (dpk - sD[qc]) : 0.f;
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
          av = __builtin_amdgcn_mfma_f32_16x16x4f32(gs[(q0 + 4 * qd + s2) * ASG_P + p], pm[s2], av, 0, 0, 0);
          ak = __builtin_amdgcn_mfma_f32_16x16x4f32(qs[(q0 + 4 * qd + s2) * ASG_P + p], ds[s2], ak, 0, 0, 0);
        }
      }
      if (kok) {
        float* row = d_qkv + (base + kj) * O + h * dk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 4 * qd + r;
          if (c < dk) {
            row[d.hd + c] = ak[r];
            row[2 * d.hd + c] = av[r];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// B'': the core without dropout (inference, validation) for clips of any length: fp32 MFMA tiles with an online
// softmax, so the n x n scores are never materialised and never recomputed.  K and V of one (clip, head) live in
// LDS; a wave owns 16 queries and walks the keys 16 at a time:
//     S^T tile [key][query] = K Q^T              (DK4 MFMAs; a lane gets 4 keys of one query)
//     running max / rescale / p = exp(s - m)     (row maximum across the 4 lane groups by two shuffles)
//     O^T [d][query] += V^T P^T                  (4 MFMAs; MFMA k-step s takes key 4*q + s, i.e. the lane's own
//                                                 register s, so P goes from one MFMA to the next without moving)
template <int DK4>
__global__ __launch_bounds__(1024) void attn_core_fwd_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ o,
                                                                  float* __restrict__ ml, AttnDims d, int nsplit) {
  extern __shared__ float lds[];
  constexpr int DKP = DK4 * 4;
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  const int ntile = (n + 15) >> 4, n16 = ntile * 16;
  const int split = blockIdx.x % nsplit, bh = blockIdx.x / nsplit;
  const int b = bh / d.heads, h = bh % d.heads;
  float* ks = lds;                   // [n16][DKP], rows >= n zero
  float* vs = ks + n16 * DKP;        // [n16 + 1][DKP] (one spare row: lanes d >= DKP read past a row)
  const long base = (long)b * n;
  const int nthr = blockDim.x, nw = nthr >> 6;
#pragma unroll 4
  for (int i = threadIdx.x; i < (n16 + 1) * DKP; i += nthr) {   // clamped addresses, padding selected afterwards
    const int j = i / DKP, c = i - j * DKP;
    const bool in = j < n && c < dk;
    const float* row = qkv + (base + (j < n ? j : n - 1)) * O + h * dk + (c < dk ? c : dk - 1);
    const float kv = row[d.hd], vv = row[2 * d.hd];
    if (j < n16) ks[i] = in ? kv : 0.f;
    vs[i] = in ? vv : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const float temp = sqrtf((float)dk);
  for (int qt = split * nw + wave; qt < ntile; qt += nw * nsplit) {
    const int qi = qt * 16 + p;
    const bool valid = qi < n;
    const float* qrow = qkv + (base + (valid ? qi : n - 1)) * O + h * dk;
    float qreg[DK4];
#pragma unroll
    for (int kk = 0; kk < DK4; ++kk) qreg[kk] = (4 * kk + q < dk) ? qrow[4 * kk + q] / temp : 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int kb = 0; kb < ntile; ++kb) {
      const int key0 = kb * 16;
      f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk)
        st = __builtin_amdgcn_mfma_f32_16x16x4f32(ks[(key0 + p) * DKP + 4 * kk + q], qreg[kk], st, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (key0 + 4 * q + r >= n) st[r] = -INFINITY;
      float bm = fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3]));
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float mn = fmaxf(m, bm);          // finite: every block holds at least one real key
      const float sc = exp_hw(m - mn);
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = exp_hw(st[r] - mn);
      l = l * sc + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= sc;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vs[(key0 + 4 * q + s2) * DKP + p], pr[s2], acc, 0, 0, 0);
      m = mn;
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (valid) {
      const float inv = 1.f / l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * q + r;
        if (c < dk) o[(base + qi) * d.hd + h * dk + c] = acc[r] * inv;
      }
      if (ml && q == 0) {
        ml[(((long)b * d.heads + h) * n + qi) * 2] = m;
        ml[(((long)b * d.heads + h) * n + qi) * 2 + 1] = l;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// C: o, x1 -> y1 = o Wfc^T + x1 -> xhat1 -> x2 -> hdn = relu(x2 W1^T + b1) -> y2 = hdn W2^T + b2 + x2 -> xhat2 -> x3
struct PostParams {
  const float *fc, *ln1_w, *ln1_b, *w1, *b1, *w2, *b2, *ln2_w, *ln2_b, *vad_w, *vad_b;
};

// The forward of everything behind the attention core runs as two kernels so that each keeps few enough weights in
// LDS for two or three workgroups per CU (one kernel with all three matrices and three token tiles needs 107 KB: one
// workgroup per CU, every global load and barrier of the chain exposed).  tile_linear_mfma may write its result over
// its residual operand: a lane reads the residual element it is about to overwrite, and a wave only touches the rows
// of its own 16 tokens.

// 16-byte store of a thread's 16 columns of token row `row` (LDS) to g[(tok) * D + ...]
__device__ __forceinline__ void row_store16(float* g, long tok, const float* row, int part) {
#pragma unroll
  for (int k = 0; k < D / 16; ++k) st4(g + tok * D + part * (D / 4) + 4 * k, ld4(row + part * (D / 4) + 4 * k));
}

// C1: y1 = o Wfc^T + x1 (x1 = g0*xhat0 + b0) -> xhat1, rstd1
__global__ __launch_bounds__(256) void attn_post_fc_kernel(const float* __restrict__ o, const float* __restrict__ xhat0,
                                                           const float* __restrict__ ln0_w, const float* __restrict__ ln0_b,
                                                           const float* __restrict__ fc, float* __restrict__ xhat1,
                                                           float* __restrict__ rstd1, AttnDims d) {
  extern __shared__ float lds[];
  const int HD = d.hd;
  float* Wfc = lds;                 // [D][WP], columns >= HD zero
  float* vec = Wfc + D * WP;        // ln0_w, ln0_b
  float* ta = vec + 2 * D;          // [TT][XS]  o, columns >= HD zero
  float* tb = ta + TT * XS;         // [TT][XS]  xhat0 -> x1 -> y1 -> xhat1
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3;
  for (int i = tid; i < D * D; i += 256) {
    const int oo = i / D, k = i - oo * D;
    Wfc[oo * WP + k] = k < HD ? fc[oo * HD + k] : 0.f;
  }
  for (int i = tid; i < D; i += 256) {
    vec[i] = ln0_w[i];
    vec[D + i] = ln0_b[i];
  }
  for (long tok0 = (long)blockIdx.x * TT; tok0 < d.ntok; tok0 += (long)gridDim.x * TT) {
    __syncthreads();
    tile_load_padded(ta, XS, o, tok0, d.ntok, HD, D);
    tile_load(tb, XS, xhat0, tok0, d.ntok, D);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      tb[t * XS + c] = fmaf(tb[t * XS + c], vec[c], vec[D + c]);
    }
    __syncthreads();
    tile_linear_mfma<D / 4, 4>(ta, XS, Wfc, WP, D, tb, XS, nullptr, false, tb, XS);   // tb = y1 (in place over x1)
    __syncthreads();
    const float r1 = ln_row(tb + t * XS, part, 1e-6f);   // tb = xhat1
    if (tok0 + t < d.ntok) {
      if (part == 0) rstd1[tok0 + t] = r1;
      row_store16(xhat1, tok0 + t, tb + t * XS, part);
    }
  }
}

// C2: x2 = g1*xhat1 + b1 -> hdn = relu(x2 W1^T + b1) -> y2 = hdn W2^T + b2 + x2 -> xhat2, rstd2, x3, VAD logit
__global__ __launch_bounds__(256) void attn_post_ffn_kernel(const float* __restrict__ xhat1, PostParams P,
                                                            float* __restrict__ hdn, float* __restrict__ xhat2,
                                                            float* __restrict__ rstd2, float* __restrict__ x3,
                                                            float* __restrict__ vad, AttnDims d) {
  extern __shared__ float lds[];
  float* W1 = lds;                  // [D][WP]
  float* W2 = W1 + D * WP;          // [D][WP]
  float* vec = W2 + D * WP;         // ln1_w, ln1_b, b1, b2, ln2_w, ln2_b, vad_w  (7*D)
  float* ta = vec + 7 * D;          // [TT][XS]  hdn
  float* tb = ta + TT * XS;         // [TT][XS]  xhat1 -> x2 -> y2 -> xhat2 -> x3
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3;
  for (int i = tid; i < D * D; i += 256) {
    const int oo = i / D, k = i - oo * D;
    W1[oo * WP + k] = P.w1[i];
    W2[oo * WP + k] = P.w2[i];
  }
  for (int i = tid; i < D; i += 256) {
    vec[i] = P.ln1_w[i]; vec[D + i] = P.ln1_b[i]; vec[2 * D + i] = P.b1[i]; vec[3 * D + i] = P.b2[i];
    vec[4 * D + i] = P.ln2_w[i]; vec[5 * D + i] = P.ln2_b[i]; vec[6 * D + i] = P.vad_w[i];
  }
  for (long tok0 = (long)blockIdx.x * TT; tok0 < d.ntok; tok0 += (long)gridDim.x * TT) {
    __syncthreads();
    tile_load(tb, XS, xhat1, tok0, d.ntok, D);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      tb[t * XS + c] = fmaf(tb[t * XS + c], vec[c], vec[D + c]);   // tb = x2
    }
    __syncthreads();
    tile_linear_mfma<D / 4, 4>(tb, XS, W1, WP, D, ta, XS, vec + 2 * D, true, nullptr, 0);   // ta = hdn
    __syncthreads();
    tile_store(hdn, ta, XS, tok0, d.ntok, D);
    tile_linear_mfma<D / 4, 4>(ta, XS, W2, WP, D, tb, XS, vec + 3 * D, false, tb, XS);      // tb = y2 (in place over x2)
    __syncthreads();
    const bool valid = tok0 + t < d.ntok;
    const float r2 = ln_row(tb + t * XS, part, 1e-6f);   // tb = xhat2
    if (valid) {
      if (part == 0) rstd2[tok0 + t] = r2;
      row_store16(xhat2, tok0 + t, tb + t * XS, part);
    }
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      tb[t * XS + c] = fmaf(tb[t * XS + c], vec[4 * D + c], vec[5 * D + c]);   // tb = x3
    }
    if (valid) row_store16(x3, tok0 + t, tb + t * XS, part);
    __syncthreads();
    // VAD head on the first token of each clip
    if (valid && part == 0 && (tok0 + t) % d.n == 0) {
      float acc = P.vad_b[0];
      for (int c = 0; c < D; ++c) acc = fmaf(tb[t * XS + c], vec[6 * D + c], acc);
      vad[(tok0 + t) / d.n] = acc;
    }
  }
}

// LayerNorm backward on a tile: g (LDS, in/out: dL/d(ln out) -> dL/d(ln in)), xhat tile (LDS), gamma (LDS).
// Also accumulates column sums for dgamma / dbeta into per-thread registers: thread (wave w, lane c) holds the part of
// column c that comes from rows 16w..16w+15 of every tile (sum the four waves with cols_over_waves at the end).
__device__ __forceinline__ void ln_bwd_tile(float* g, const float* xh, const float* gamma, const float* rstd_g, long tok0,
                                            long ntok, float& dgam, float& dbet) {
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3;
  {   // column sums: wave w takes rows 16w..16w+15 of column (tid & 63); the waves' partials meet at the kernel's end
    const int col = tid & (D - 1), r0 = (tid >> 6) * (TT / 4);
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int tt = 0; tt < TT / 4; ++tt) {
      const float gv = g[(r0 + tt) * XS + col];
      sg = fmaf(gv, xh[(r0 + tt) * XS + col], sg);
      sb += gv;
    }
    dgam += sg;
    dbet += sb;
  }
  __syncthreads();
  float a = 0.f, bsum = 0.f;
#pragma unroll
  for (int k = 0; k < D / 4; ++k) {
    const int c = part * (D / 4) + k;
    const float gg = g[t * XS + c] * gamma[c];
    a += gg;
    bsum = fmaf(gg, xh[t * XS + c], bsum);
  }
  a = quad_sum(a) * (1.f / D);
  bsum = quad_sum(bsum) * (1.f / D);
  const float rs = (tok0 + t < ntok) ? rstd_g[tok0 + t] : 0.f;
#pragma unroll
  for (int k = 0; k < D / 4; ++k) {
    const int c = part * (D / 4) + k;
    g[t * XS + c] = rs * (g[t * XS + c] * gamma[c] - a - xh[t * XS + c] * bsum);
  }
}

// Sum over the four waves of a per-(wave, column) partial; valid in threads tid < D.  red: 4*D floats of LDS.
__device__ __forceinline__ float cols_over_waves(float v, float* red) {
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  return tid < D ? (red[tid] + red[D + tid]) + (red[2 * D + tid] + red[3 * D + tid]) : 0.f;
}

// slab row layout of attn_post_bwd (floats): dW2[D*D] dW1[D*D] dWfc[D*HD] then 7 vectors of D:
// dln2_w dln2_b db2 db1 dln1_w dln1_b dvad_w, then dvad_b (1)
__global__ __launch_bounds__(256) void attn_post_bwd_kernel(const float* __restrict__ g3, const float* __restrict__ d_vad,
                                                            const float* __restrict__ o, const float* __restrict__ xhat1,
                                                            const float* __restrict__ rstd1, const float* __restrict__ hdn,
                                                            const float* __restrict__ xhat2, const float* __restrict__ rstd2,
                                                            const float* __restrict__ x3, PostParams P,
                                                            float* __restrict__ d_o, float* __restrict__ d_x1,
                                                            float* __restrict__ slab, AttnDims d) {
  extern __shared__ float lds[];
  const int HD = d.hd;
  float* WfcT = lds;                // [HD][WP]  (transposed: d_o = dy1 Wfc)
  float* W1T = WfcT + D * WP;       // [D][WP]   W1T[k][o] = W1[o][k]
  float* W2T = W1T + D * WP;
  float* vec = W2T + D * WP;        // ln1_w, ln2_w, vad_w, ln1_b  (4*D)
  float* ta = vec + 4 * D;
  float* tb = ta + TT * XS;
  float* tc = tb + TT * XS;
  float* td = tc + TT * XS;
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3;
  for (int i = tid; i < D * D; i += 256) {     // rows k >= HD of the transposed fc weight are never read
    const int oo = i / D, k = i - oo * D;
    if (k < HD) WfcT[k * WP + oo] = P.fc[oo * HD + k];
  }
  for (int i = tid; i < D * D; i += 256) {
    const int oo = i / D, k = i - oo * D;
    W1T[k * WP + oo] = P.w1[i];
    W2T[k * WP + oo] = P.w2[i];
  }
  for (int i = tid; i < D; i += 256) {
    vec[i] = P.ln1_w[i]; vec[D + i] = P.ln2_w[i]; vec[2 * D + i] = P.vad_w[i]; vec[3 * D + i] = P.ln1_b[i];
  }
  // weight gradients: wave w accumulates the 16 output rows 16w..16w+15 of each matrix (4 tiles of 16 columns)
  const int wave = tid >> 6;
  f32x4 aW2[4], aW1[4], aFc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) aW2[e] = aW1[e] = aFc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dg2 = 0.f, db2ln = 0.f, dbias2 = 0.f, dbias1 = 0.f, dg1 = 0.f, db1ln = 0.f, dvw = 0.f, dvb = 0.f;

  for (long tok0 = (long)blockIdx.x * TT; tok0 < d.ntok; tok0 += (long)gridDim.x * TT) {
    __syncthreads();
    tile_load(ta, XS, g3, tok0, d.ntok, D);       // ta = dL/dx3
    tile_load(tb, XS, xhat2, tok0, d.ntok, D);    // tb = xhat2
    tile_load(tc, XS, hdn, tok0, d.ntok, D);      // tc = hdn
    __syncthreads();
    // VAD head gradient enters at token 0 of each clip
    if (tok0 + t < d.ntok && (tok0 + t) % d.n == 0) {
      const float dv = d_vad[(tok0 + t) / d.n];
#pragma unroll
      for (int k = 0; k < D / 4; ++k) {
        const int c = part * (D / 4) + k;
        ta[t * XS + c] = fmaf(dv, vec[2 * D + c], ta[t * XS + c]);
      }
    }
    {   // dvad_w[c] += sum over clips dv * x3[token0][c]; rows split over the waves like the other column sums
      const int col = tid & (D - 1), r0 = (tid >> 6) * (TT / 4);
      for (int tt = r0; tt < r0 + TT / 4; ++tt)
        if (tok0 + tt < d.ntok && (tok0 + tt) % d.n == 0) {
          const float dv = d_vad[(tok0 + tt) / d.n];
          dvw = fmaf(dv, x3[(tok0 + tt) * D + col], dvw);
          if (col == 0) dvb += dv;
        }
    }
    __syncthreads();
    ln_bwd_tile(ta, tb, vec + D, rstd2, tok0, d.ntok, dg2, db2ln);   // ta = dy2
    __syncthreads();
    {
      const int col = tid & (D - 1), r0 = (tid >> 6) * (TT / 4);
      float s = 0.f;
#pragma unroll
      for (int tt = 0; tt < TT / 4; ++tt) s += ta[(r0 + tt) * XS + col];
      dbias2 += s;
    }
    tile_outer_mfma<4>(aW2, ta, XS, wave, tc, XS);                     // dW2 += dy2^T hdn
    // td = d_pre = (dy2 W2) * [hdn > 0]
    tile_linear_mfma<D / 4, 4>(ta, XS, W2T, WP, D, td, XS, nullptr, false, nullptr, 0);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      if (!(tc[t * XS + c] > 0.f)) td[t * XS + c] = 0.f;
    }
    __syncthreads();
    tile_load(tb, XS, xhat1, tok0, d.ntok, D);    // tb = xhat1
    __syncthreads();
    {
      const int col = tid & (D - 1), r0 = (tid >> 6) * (TT / 4);
      float s = 0.f;
#pragma unroll
      for (int tt = 0; tt < TT / 4; ++tt) s += td[(r0 + tt) * XS + col];
      dbias1 += s;
    }
    // tc = x2 = ln1_w*xhat1 + ln1_b
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      tc[t * XS + c] = fmaf(tb[t * XS + c], vec[c], vec[3 * D + c]);
    }
    __syncthreads();
    tile_outer_mfma<4>(aW1, td, XS, wave, tc, XS);                     // dW1 += d_pre^T x2
    __syncthreads();
    // tc = d_x2 = dy2 + d_pre W1
    tile_linear_mfma<D / 4, 4>(td, XS, W1T, WP, D, tc, XS, nullptr, false, ta, XS);
    __syncthreads();
    ln_bwd_tile(tc, tb, vec, rstd1, tok0, d.ntok, dg1, db1ln);        // tc = dy1
    __syncthreads();
    tile_store(d_x1, tc, XS, tok0, d.ntok, D);                          // residual branch of x1
    tile_load_padded(ta, XS, o, tok0, d.ntok, HD, D);                   // ta = o, columns >= HD zero
    __syncthreads();
    tile_outer_mfma<4>(aFc, tc, XS, wave, ta, XS);                      // dWfc += dy1^T o
    // td = d_o = dy1 Wfc  (HD outputs)
    tile_linear_mfma<D / 4, 4>(tc, XS, WfcT, WP, HD, td, XS, nullptr, false, nullptr, 0);
    __syncthreads();
    tile_store(d_o, td, XS, tok0, d.ntok, HD);
  }
  float* row = slab + (size_t)blockIdx.x * (2 * D * D + D * HD + 7 * D + 1);
  store_outer_mfma<4>(row, D, aW2, wave, D, D);
  store_outer_mfma<4>(row + D * D, D, aW1, wave, D, D);
  store_outer_mfma<4>(row + 2 * D * D, HD, aFc, wave, D, HD);
  dg2 = cols_over_waves(dg2, ta);
  db2ln = cols_over_waves(db2ln, ta);
  dbias2 = cols_over_waves(dbias2, ta);
  dbias1 = cols_over_waves(dbias1, ta);
  dg1 = cols_over_waves(dg1, ta);
  db1ln = cols_over_waves(db1ln, ta);
  dvw = cols_over_waves(dvw, ta);
  dvb = cols_over_waves(dvb, ta);
  if (tid < D) {
    float* v = row + 2 * D * D + D * HD;
    v[tid] = dg2; v[D + tid] = db2ln; v[2 * D + tid] = dbias2; v[3 * D + tid] = dbias1;
    v[4 * D + tid] = dg1; v[5 * D + tid] = db1ln; v[6 * D + tid] = dvw;
    if (tid == 0) v[7 * D] = dvb;
  }
}

// A': d_qkv, d_x1 (partial) -> d_x1 += d_qkv Wqkv; dWqkv += d_qkv^T x1; LN0 backward -> d_x0
// slab row: dWqkv[O*D] dln0_w[D] dln0_b[D]
__global__ __launch_bounds__(256) void attn_pre_bwd_kernel(const float* __restrict__ d_qkv, const float* __restrict__ d_x1,
                                                           const float* __restrict__ xhat0, const float* __restrict__ rstd0,
                                                           const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                                           const float* __restrict__ wq, const float* __restrict__ wk,
                                                           const float* __restrict__ wv, float* __restrict__ d_x0,
                                                           float* __restrict__ slab, AttnDims d) {
  extern __shared__ float lds[];
  const int O = 3 * d.hd;
  constexpr int OP = 3 * D + 4;     // pitch of the O-wide tiles (O <= 192), = 4 mod 64 banks
  float* WT = lds;                  // [D][OP]: WT[k][o] = Wqkv[o][k], columns >= O zero
  float* dq = WT + D * OP;          // [TT][OP], columns >= O zero
  float* ta = dq + TT * OP;         // [TT][XS]
  float* tb = ta + TT * XS;
  float* tc = tb + TT * XS;
  float* lw = tc + TT * XS;         // [2][D]
  const int tid = threadIdx.x, t = tid >> 2, part = tid & 3, wave = tid >> 6;
  for (int i = tid; i < D * 3 * D; i += 256) {
    const int k = i / (3 * D), o = i - k * (3 * D);
    float v = 0.f;
    if (o < O) {
      const int which = o / d.hd, oo = o - which * d.hd;
      const float* w = which == 0 ? wq : (which == 1 ? wk : wv);
      v = w[oo * D + k];
    }
    WT[k * OP + o] = v;
  }
  for (int i = tid; i < D; i += 256) {
    lw[i] = ln_w[i];
    lw[D + i] = ln_b[i];
  }
  // dWqkv (O x D): wave w accumulates the row tiles w, w+4, w+8 (16 output features each, 4 tiles of 16 columns)
  f32x4 aW[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int e = 0; e < 4; ++e) aW[r][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dg0 = 0.f, db0 = 0.f;

  for (long tok0 = (long)blockIdx.x * TT; tok0 < d.ntok; tok0 += (long)gridDim.x * TT) {
    __syncthreads();
    tile_load_padded(dq, OP, d_qkv, tok0, d.ntok, O, 3 * D);
    tile_load(ta, XS, d_x1, tok0, d.ntok, D);     // ta = d_x1 (residual part)
    tile_load(tb, XS, xhat0, tok0, d.ntok, D);    // tb = xhat0
    __syncthreads();
    // tc = x1 = ln_w*xhat0 + ln_b
#pragma unroll
    for (int k = 0; k < D / 4; ++k) {
      const int c = part * (D / 4) + k;
      tc[t * XS + c] = fmaf(tb[t * XS + c], lw[c], lw[D + c]);
    }
    // ta += d_qkv Wqkv
    tile_linear_mfma<3 * D / 4, 4>(dq, OP, WT, OP, D, ta, XS, nullptr, false, ta, XS);
    __syncthreads();
    // dWqkv += d_qkv^T x1   (rows of padded tokens are zero in dq)
#pragma unroll
    for (int r = 0; r < 3; ++r) tile_outer_mfma<4>(aW[r], dq, OP, wave + 4 * r, tc, XS);
    ln_bwd_tile(ta, tb, lw, rstd0, tok0, d.ntok, dg0, db0);
    __syncthreads();
    tile_store(d_x0, ta, XS, tok0, d.ntok, D);
  }
  float* row = slab + (size_t)blockIdx.x * (O * D + 2 * D);
#pragma unroll
  for (int r = 0; r < 3; ++r) store_outer_mfma<4>(row, D, aW[r], wave + 4 * r, O, D);
  dg0 = cols_over_waves(dg0, ta);
  db0 = cols_over_waves(db0, ta);
  if (tid < D) {
    row[O * D + tid] = dg0;
    row[O * D + D + tid] = db0;
  }
}

__global__ __launch_bounds__(256) void attn_slab_reduce_kernel(const float* __restrict__ slab, int nrows, int ncols,
                                                             float* __restrict__ out) {
  __shared__ double red[256];
  const int j = blockIdx.x * SLAB_COLS + (threadIdx.x & (SLAB_COLS - 1));
  const bool ok = j < ncols;
  const double s = slab_column_sum(slab, nrows, ncols, j, ok, red);
  if (threadIdx.x < SLAB_COLS && ok) out[j] = (float)s;
}

int check_attn(const char* who, long ntok, int n, int d_model, int heads, int dk) {
  NOISE_REQUIRE(d_model == D, "%s: d_model=%d unsupported (this build handles %d)", who, d_model, D);
  NOISE_REQUIRE(heads > 0 && dk > 0 && dk <= DKMAX && heads * dk <= D && (heads * dk) % 4 == 0,
                "%s: heads=%d d_k=%d unsupported", who, heads, dk);
  NOISE_REQUIRE(n > 0 && ntok > 0 && ntok % n == 0, "%s: %ld tokens is not a multiple of n=%d", who, ntok, n);
  return 0;
}

constexpr int ATTN_MAX_BLOCKS = 256;

int attn_chunk_rows_host(int n, int dk) {
  const int kc = ATTN_CHUNK_FLOATS / ((dk + 3) / 4 * 4);
  return n < kc ? n : kc;
}

}  // namespace

NOISE_EXPORT int noise_attn_max_blocks(void) { return ATTN_MAX_BLOCKS; }

NOISE_EXPORT int noise_attn_pre_fwd(const float* x_y, const float* x_coef, int x_relu6, const float* ln_w,
                                    const float* ln_b, const float* wq, const float* wk, const float* wv, float* xhat0,
                                    float* rstd0, float* qkv, long ntok, int n, int d_model, int heads, int dk,
                                    void* stream) {
  if (check_attn("noise_attn_pre_fwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(x_y && ln_w && ln_b && wq && wk && wv && xhat0 && rstd0 && qkv, "noise_attn_pre_fwd: null pointer");
  AttnDims d{ntok, n, heads, dk, heads * dk};
  const int O = 3 * d.hd;
  const size_t smem = (size_t)(O * WP + TT * XS + 2 * D) * sizeof(float);
  LazySrc X{x_y, x_coef, x_relu6};
  NOISE_ALLOW_LDS(attn_pre_fwd_kernel, smem);
  // no per-workgroup slab in the forward direction: two workgroups per CU (67 KB of LDS each)
  hipLaunchKernelGGL(attn_pre_fwd_kernel, dim3(noise_grid(ntok, TT, 2 * ATTN_MAX_BLOCKS)), dim3(256), smem, (hipStream_t)stream, X, ln_w,
                     ln_b, wq, wk, wv, xhat0, rstd0, qkv, d);
  NOISE_CHECK_LAUNCH("noise_attn_pre_fwd");
  return 0;
}

// keep (optional): uint8 [B][heads][n][n] keep-mask; otherwise a counter-based hash of (seed, index) with
// drop probability p_drop (0 disables dropout).  ml (optional unless training): [B][heads][n][2] = (max, sum).
NOISE_EXPORT int noise_attn_core_fwd(const float* qkv, float* o, float* ml, const unsigned char* keep,
                                     unsigned long long seed, float p_drop, long ntok, int n, int d_model, int heads,
                                     int dk, void* stream) {
  if (check_attn("noise_attn_core_fwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(qkv && o, "noise_attn_core_fwd: null pointer");
  NOISE_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "noise_attn_core_fwd: p_drop=%f out of range", p_drop);
  AttnDims d{ntok, n, heads, dk, heads * dk};
  // segments: fp32 MFMA tiles, four waves per (clip, head) (NOISE_ATTN_FWD_WAVE, diagnostic: the thread-per-query form)
  static const bool wave_form = getenv("NOISE_ATTN_FWD_WAVE") != nullptr;
  if (n <= AW && !wave_form) {
    const int dk4 = (dk + 3) / 4;
    const dim3 grid((unsigned)noise_pad8((ntok / n) * heads));
    const int drop = keep ? 1 : (p_drop > 0.f ? 2 : 0);
#define X(K4)                                                                                                     \
  if (dk4 == K4) {                                                                                                \
    if (drop == 0)                                                                                                \
      hipLaunchKernelGGL((attn_core_fwd_seg_mfma_kernel<K4, 0>), grid, dim3(256), 0, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
    else if (drop == 1)                                                                                           \
      hipLaunchKernelGGL((attn_core_fwd_seg_mfma_kernel<K4, 1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
    else                                                                                                          \
      hipLaunchKernelGGL((attn_core_fwd_seg_mfma_kernel<K4, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
  }
    X(1) X(2) X(3) X(4)
#undef X
    NOISE_CHECK_LAUNCH("noise_attn_core_fwd/mfma");
    return 0;
  }
  if (n <= AW) {   // one wave per (clip, head)
    const int dk4 = (dk + 3) / 4;
    const size_t sm = (size_t)2 * n * dk4 * 4 * sizeof(float);
    const dim3 grid((unsigned)noise_pad8((ntok / n) * heads));
    const int drop = keep ? 1 : (p_drop > 0.f ? 2 : 0);
#define X(K4)                                                                                                     \
  if (dk4 == K4) {                                                                                                \
    if (drop == 0)                                                                                                \
      hipLaunchKernelGGL((attn_core_fwd_wave_kernel<K4, 0>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
    else if (drop == 1)                                                                                           \
      hipLaunchKernelGGL((attn_core_fwd_wave_kernel<K4, 1>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
    else                                                                                                          \
      hipLaunchKernelGGL((attn_core_fwd_wave_kernel<K4, 2>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, o, ml, keep, \
                         seed, p_drop, d);                                                                        \
  }
    X(1) X(2) X(3) X(4)
#undef X
    NOISE_CHECK_LAUNCH("noise_attn_core_fwd/wave");
    return 0;
  }
  const size_t sm_mfma = (size_t)(2 * ((n + 15) / 16 * 16) + 1) * ((dk + 3) / 4) * 4 * sizeof(float);
  if (keep == nullptr && p_drop == 0.f && sm_mfma <= 158 * 1024) {   // no dropout: MFMA tiles with an online softmax
    const int dk4 = (dk + 3) / 4, n16 = (n + 15) / 16 * 16;
    const size_t sm = sm_mfma;
    const long bh = (ntok / n) * heads;
    // K and V in LDS bound the workgroups a CU holds; past 32 KB the waves have to come from inside the workgroup
    const int threads = sm > 32 * 1024 ? 1024 : 256, nw = threads / 64;
    int nsplit = (int)(1024 / bh);
    const int max_split = (n16 / 16 + nw - 1) / nw;
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    const dim3 grid((unsigned)(bh * nsplit));
#define X(K4)                                                                                                      \
  if (dk4 == K4) {                                                                                                 \
    NOISE_ALLOW_LDS(attn_core_fwd_mfma_kernel<K4>, sm);                                                            \
    hipLaunchKernelGGL(attn_core_fwd_mfma_kernel<K4>, grid, dim3(threads), sm, (hipStream_t)stream, qkv, o, ml, d, nsplit); \
  }
    X(1) X(2) X(3) X(4)
#undef X
    NOISE_CHECK_LAUNCH("noise_attn_core_fwd/mfma");
    return 0;
  }
  const size_t smem = (size_t)2 * attn_chunk_rows_host(n, dk) * ((dk + 3) / 4 * 4) * sizeof(float);
  NOISE_ALLOW_LDS(attn_core_fwd_kernel, smem);
  hipLaunchKernelGGL(attn_core_fwd_kernel, dim3((unsigned)(ntok / n) * heads), dim3(256), smem, (hipStream_t)stream, qkv, o,
                     ml, keep, seed, p_drop, d);
  NOISE_CHECK_LAUNCH("noise_attn_core_fwd");
  return 0;
}

NOISE_EXPORT int noise_attn_core_bwd(const float* qkv, const float* d_o, const float* ml, const unsigned char* keep,
                                     unsigned long long seed, float p_drop, float* d_qkv, long ntok, int n, int d_model,
                                     int heads, int dk, void* stream) {
  if (check_attn("noise_attn_core_bwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(qkv && d_o && ml && d_qkv, "noise_attn_core_bwd: null pointer");
  AttnDims d{ntok, n, heads, dk, heads * dk};
  // segments: fp32 MFMA tiles, four waves per (clip, head) (NOISE_ATTN_BWD_WAVE, diagnostic: the thread-per-query form)
  static const bool wave_form = getenv("NOISE_ATTN_BWD_WAVE") != nullptr;
  if (n <= AW && !wave_form) {
    const int dk4 = (dk + 3) / 4;
    const dim3 grid((unsigned)noise_pad8((ntok / n) * heads));
    const int drop = keep ? 1 : (p_drop > 0.f ? 2 : 0);
#define X(K4)                                                                                                        \
  if (dk4 == K4) {                                                                                                   \
    if (drop == 0)                                                                                                   \
      hipLaunchKernelGGL((attn_core_bwd_seg_mfma_kernel<K4, 0>), grid, dim3(256), 0, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
    else if (drop == 1)                                                                                              \
      hipLaunchKernelGGL((attn_core_bwd_seg_mfma_kernel<K4, 1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
    else                                                                                                             \
      hipLaunchKernelGGL((attn_core_bwd_seg_mfma_kernel<K4, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
  }
    X(1) X(2) X(3) X(4)
#undef X
    NOISE_CHECK_LAUNCH("noise_attn_core_bwd/mfma");
    return 0;
  }
  if (n <= AW) {   // one wave per (clip, head)
    const int dk4 = (dk + 3) / 4;
    const size_t sm = (size_t)(4 * n * dk4 * 4 + n * (n | 1)) * sizeof(float);
    const dim3 grid((unsigned)noise_pad8((ntok / n) * heads));
    const int drop = keep ? 1 : (p_drop > 0.f ? 2 : 0);
#define X(K4)                                                                                                        \
  if (dk4 == K4) {                                                                                                   \
    if (drop == 0)                                                                                                   \
      hipLaunchKernelGGL((attn_core_bwd_wave_kernel<K4, 0>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
    else if (drop == 1)                                                                                              \
      hipLaunchKernelGGL((attn_core_bwd_wave_kernel<K4, 1>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
    else                                                                                                             \
      hipLaunchKernelGGL((attn_core_bwd_wave_kernel<K4, 2>), grid, dim3(AW), sm, (hipStream_t)stream, qkv, d_o, ml, keep, \
                         seed, p_drop, d_qkv, d);                                                                    \
  }
    X(1) X(2) X(3) X(4)
#undef X
    NOISE_CHECK_LAUNCH("noise_attn_core_bwd/wave");
    return 0;
  }
  const size_t smem = ((size_t)2 * attn_chunk_rows_host(n, dk) * ((dk + 3) / 4 * 4) + (size_t)3 * n) * sizeof(float);
  NOISE_REQUIRE(smem <= 150 * 1024, "noise_attn_core_bwd: n=%d tokens per clip exceed the LDS limit (3 floats per token)", n);
  NOISE_ALLOW_LDS(attn_core_bwd_kernel, smem);
  hipLaunchKernelGGL(attn_core_bwd_kernel, dim3((unsigned)(ntok / n) * heads), dim3(256), smem, (hipStream_t)stream, qkv,
                     d_o, ml, keep, seed, p_drop, d_qkv, d);
  NOISE_CHECK_LAUNCH("noise_attn_core_bwd");
  return 0;
}

NOISE_EXPORT int noise_attn_post_fwd(const float* o, const float* xhat0, const float* ln0_w, const float* ln0_b,
                                     const float* fc, const float* ln1_w, const float* ln1_b, const float* w1,
                                     const float* b1, const float* w2, const float* b2, const float* ln2_w,
                                     const float* ln2_b, const float* vad_w, const float* vad_b, float* xhat1,
                                     float* rstd1, float* hdn, float* xhat2, float* rstd2, float* x3, float* vad,
                                     long ntok, int n, int d_model, int heads, int dk, void* stream) {
  if (check_attn("noise_attn_post_fwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(o && xhat0 && ln0_w && ln0_b && fc && ln1_w && ln1_b && w1 && b1 && w2 && b2 && ln2_w && ln2_b && vad_w &&
                    vad_b && xhat1 && rstd1 && hdn && xhat2 && rstd2 && x3 && vad,
                "noise_attn_post_fwd: null pointer");
  AttnDims d{ntok, n, heads, dk, heads * dk};
  PostParams P{fc, ln1_w, ln1_b, w1, b1, w2, b2, ln2_w, ln2_b, vad_w, vad_b};
  // no per-workgroup slab in the forward direction: the grids follow what a CU holds (3 and 2 workgroups)
  const size_t sm1 = (size_t)(D * WP + 2 * D + 2 * TT * XS) * sizeof(float);
  const size_t sm2 = (size_t)(2 * D * WP + 7 * D + 2 * TT * XS) * sizeof(float);
  NOISE_ALLOW_LDS(attn_post_fc_kernel, sm1);
  NOISE_ALLOW_LDS(attn_post_ffn_kernel, sm2);
  hipLaunchKernelGGL(attn_post_fc_kernel, dim3(noise_grid(ntok, TT, 3 * ATTN_MAX_BLOCKS)), dim3(256), sm1, (hipStream_t)stream, o,
                     xhat0, ln0_w, ln0_b, fc, xhat1, rstd1, d);
  hipLaunchKernelGGL(attn_post_ffn_kernel, dim3(noise_grid(ntok, TT, 2 * ATTN_MAX_BLOCKS)), dim3(256), sm2, (hipStream_t)stream,
                     xhat1, P, hdn, xhat2, rstd2, x3, vad, d);
  NOISE_CHECK_LAUNCH("noise_attn_post_fwd");
  return 0;
}

// grads: flat output [dW2 D*D | dW1 D*D | dWfc D*hd | dln2_w | dln2_b | db2 | db1 | dln1_w | dln1_b | dvad_w | dvad_b(1)]
// scratch: noise_attn_max_blocks() rows of that size.
NOISE_EXPORT int noise_attn_post_bwd(const float* g3, const float* d_vad, const float* o, const float* xhat1,
                                     const float* rstd1, const float* hdn, const float* xhat2, const float* rstd2,
                                     const float* x3, const float* fc, const float* ln1_w, const float* ln1_b,
                                     const float* w1, const float* w2, const float* ln2_w, const float* vad_w, float* d_o,
                                     float* d_x1, float* grads, float* scratch, long ntok, int n, int d_model, int heads,
                                     int dk, void* stream) {
  if (check_attn("noise_attn_post_bwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(g3 && d_vad && o && xhat1 && rstd1 && hdn && xhat2 && rstd2 && x3 && fc && ln1_w && ln1_b && w1 && w2 &&
                    ln2_w && vad_w && d_o && d_x1 && scratch,
                "noise_attn_post_bwd: null pointer");
  AttnDims d{ntok, n, heads, dk, heads * dk};
  PostParams P{fc, ln1_w, ln1_b, w1, nullptr, w2, nullptr, ln2_w, nullptr, vad_w, nullptr};
  const int ncols = 2 * D * D + D * d.hd + 7 * D + 1;
  const int grid = noise_grid(ntok, TT, ATTN_MAX_BLOCKS);
  const size_t smem = (size_t)(3 * D * WP + 4 * D + 4 * TT * XS) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  NOISE_ALLOW_LDS(attn_post_bwd_kernel, smem);
  hipLaunchKernelGGL(attn_post_bwd_kernel, dim3(grid), dim3(256), smem, st, g3, d_vad, o, xhat1, rstd1, hdn, xhat2, rstd2, x3,
                     P, d_o, d_x1, scratch, d);
  NOISE_CHECK_LAUNCH("noise_attn_post_bwd");
  if (grads != nullptr) {   // NULL: the caller folds the slab rows (min(ceil(ntok/64), noise_attn_max_blocks())) itself
    hipLaunchKernelGGL(attn_slab_reduce_kernel, dim3(noise_cdiv(ncols, SLAB_COLS)), dim3(256), 0, st, scratch, grid, ncols, grads);
    NOISE_CHECK_LAUNCH("noise_attn_post_bwd/reduce");
  }
  return 0;
}

// grads: flat output [dWq hd*D | dWk hd*D | dWv hd*D | dln0_w D | dln0_b D]
NOISE_EXPORT int noise_attn_pre_bwd(const float* d_qkv, const float* d_x1, const float* xhat0, const float* rstd0,
                                    const float* ln_w, const float* ln_b, const float* wq, const float* wk,
                                    const float* wv, float* d_x0, float* grads, float* scratch, long ntok, int n,
                                    int d_model, int heads, int dk, void* stream) {
  if (check_attn("noise_attn_pre_bwd", ntok, n, d_model, heads, dk)) return -1;
  NOISE_REQUIRE(d_qkv && d_x1 && xhat0 && rstd0 && ln_w && ln_b && wq && wk && wv && d_x0 && scratch,
                "noise_attn_pre_bwd: null pointer");
  AttnDims d{ntok, n, heads, dk, heads * dk};
  const int O = 3 * d.hd;
  NOISE_REQUIRE((O / 4) * (D / 4) <= 4 * 256, "noise_attn_pre_bwd: projection too wide");
  const int ncols = O * D + 2 * D;
  const int grid = noise_grid(ntok, TT, ATTN_MAX_BLOCKS);
  const size_t smem = (size_t)((D + TT) * (3 * D + 4) + 3 * TT * XS + 2 * D) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  NOISE_ALLOW_LDS(attn_pre_bwd_kernel, smem);
  hipLaunchKernelGGL(attn_pre_bwd_kernel, dim3(grid), dim3(256), smem, st, d_qkv, d_x1, xhat0, rstd0, ln_w, ln_b, wq, wk, wv,
                     d_x0, scratch, d);
  NOISE_CHECK_LAUNCH("noise_attn_pre_bwd");
  if (grads != nullptr) {   // NULL: the caller folds the slab rows (min(ceil(ntok/64), noise_attn_max_blocks())) itself
    hipLaunchKernelGGL(attn_slab_reduce_kernel, dim3(noise_cdiv(ncols, SLAB_COLS)), dim3(256), 0, st, scratch, grid, ncols, grads);
    NOISE_CHECK_LAUNCH("noise_attn_pre_bwd/reduce");
  }
  return 0;
}
