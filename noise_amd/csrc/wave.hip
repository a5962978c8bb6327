// Shape-generic kernels for configurations outside the tuned schedule: the token-level pieces of the transformer
// bottleneck (train/models.py:14-225) for any model width / head size / sequence length, and the waveform-domain wrapper
// of Base.forward (train/models.py:576-581, :614-626).  Together with the generic convolution kernels of ofa.hip they
// run the forward pass of the waveform-domain U-Net (train/configs/mobilenetv2.yaml).
#include "common.h"

namespace {

// ---- per-clip normaliser (models.py:614-622): mx[b] = max|x[b]| + 1e-6, out = x / mx -----------------------------------
__global__ __launch_bounds__(256) void wave_normalize_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             float* __restrict__ mx, long L) {
  __shared__ float red[4];
  const float* xb = x + (size_t)blockIdx.x * L;
  float m = 0.f;
  for (long i = threadIdx.x; i < L; i += 256) m = fmaxf(m, fabsf(xb[i]));
  m = wave_reduce_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) + 1e-6f;
  if (threadIdx.x == 0) mx[blockIdx.x] = m;
  float* ob = out + (size_t)blockIdx.x * L;
  for (long i = threadIdx.x; i < L; i += 256) ob[i] = xb[i] / m;
}

// ---- out = (y[crop from the end] + bias) * sigmoid(1e4 * vad) * mx   (models.py:578-580, :610, :624-626) ---------------
__global__ __launch_bounds__(256) void wave_final_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                         const float* __restrict__ vad, const float* __restrict__ mx,
                                                         float* __restrict__ out, int B, int Hy, int Wy, int H, int W) {
  const long n = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int w = (int)(i % W);
    const long t = i / W;
    const int h = (int)(t % H), b = (int)(t / H);
    float v = y[((size_t)b * Hy + (Hy - H + h)) * Wy + (Wy - W + w)] + (bias ? bias[0] : 0.f);
    if (vad) v *= 1.f / (1.f + expf(-1e4f * vad[b]));
    if (mx) v *= mx[b];
    out[i] = v;
  }
}

// ---- token epilogue: out[t] = LayerNorm(act(y[t] + bias) + res[t]) (models.py:88-92, :109-118); one wave per token ------
constexpr int TOK_PER = 16;   // features per lane: D <= 1024

__global__ __launch_bounds__(256) void tok_epilogue_kernel(const float* __restrict__ y, int ldy, const float* __restrict__ bias,
                                                           int relu, const float* __restrict__ res, int ldr,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ out, int ldo, float* __restrict__ pre,
                                                           long M, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long t = (long)blockIdx.x * 4 + wave; t < M; t += (long)gridDim.x * 4) {
    float v[TOK_PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < TOK_PER; ++j) {
      const int i = lane + 64 * j;
      float x = 0.f;
      if (i < D) {
        x = y[(size_t)t * ldy + i];
        if (bias) x += bias[i];
        if (relu) x = fmaxf(x, 0.f);
        if (res) x += res[(size_t)t * ldr + i];
      }
      v[j] = x;
      s += x;
      if (pre != nullptr && i < D) pre[(size_t)t * D + i] = x;   // the LayerNorm's input, kept for the backward pass
    }
    if (gamma != nullptr) {
      const float mean = wave_reduce_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < TOK_PER; ++j) {
        const float dlt = v[j] - mean;
        q += (lane + 64 * j < D) ? dlt * dlt : 0.f;
      }
      const float rstd = 1.f / sqrtf(wave_reduce_sum(q) / (float)D + eps);
#pragma unroll
      for (int j = 0; j < TOK_PER; ++j) {
        const int i = lane + 64 * j;
        if (i < D) out[(size_t)t * ldo + i] = (v[j] - mean) * rstd * gamma[i] + beta[i];
      }
    } else {
#pragma unroll
      for (int j = 0; j < TOK_PER; ++j) {
        const int i = lane + 64 * j;
        if (i < D) out[(size_t)t * ldo + i] = v[j];
      }
    }
  }
}

// ---- scaled dot-product attention (models.py:14-36) for one (clip, head, block of 256 queries) ----------------------------
// A thread owns a query; the keys and values of the head pass through LDS in blocks of 128, the softmax runs online (a
// running maximum, the sums rescaled when it moves).  An optional keep mask (0/1 per query-key pair) is the attention
// dropout: masked weights drop out of the numerator only, the kept ones are scaled by keep_scale = 1 / (1 - p).
constexpr int ATT_KB = 128;

// 16 consecutive keep bytes of a query's row (any alignment) as four words; only the first `count` are read from memory.
// A thread walks its own row, so byte-sized loads would touch a different cache line in every lane for every key; the
// 16-byte form takes a line in eight requests instead of 128.
__device__ __forceinline__ void keep_bytes16(const unsigned char* p, int count, unsigned (&w)[4]) {
  if (count >= 16) {
    __builtin_memcpy(w, p, 16);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned v = 0;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb)
        if (4 * e + bb < count) v |= (unsigned)p[4 * e + bb] << (8 * bb);
      w[e] = v;
    }
  }
}

template <int DKP>
__global__ __launch_bounds__(256) void tok_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, int ld,
                                                            const unsigned char* __restrict__ keep, float keep_scale,
                                                            float* __restrict__ out, int ldo, int n, int H, int dk,
                                                            float inv_temp) {
  __shared__ float Ks[ATT_KB * DKP];
  __shared__ float Vs[ATT_KB * DKP];
  const int h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const size_t row0 = (size_t)b * n;
  float qr[DKP], acc[DKP];
#pragma unroll
  for (int d = 0; d < DKP; ++d) {
    qr[d] = (live && d < dk) ? q[(row0 + i) * ld + h * dk + d] * inv_temp : 0.f;
    acc[d] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  const unsigned char* kp = keep ? keep + (((size_t)b * H + h) * n + (live ? i : 0)) * n : nullptr;
  for (int j0 = 0; j0 < n; j0 += ATT_KB) {
    __syncthreads();
    for (int e = threadIdx.x; e < ATT_KB * DKP; e += blockDim.x) {
      const int j = e / DKP, d = e - j * DKP;
      const bool in = j0 + j < n && d < dk;
      Ks[e] = in ? k[(row0 + j0 + j) * ld + h * dk + d] : 0.f;
      Vs[e] = in ? v[(row0 + j0 + j) * ld + h * dk + d] : 0.f;
    }
    __syncthreads();
    const int jn = n - j0 < ATT_KB ? n - j0 : ATT_KB;
    for (int jb = 0; jb < jn; jb += 16) {
      const int kn = jn - jb < 16 ? jn - jb : 16;
      unsigned kw[4];
      if (kp) keep_bytes16(kp + j0 + jb, kn, kw);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        if (jj < kn) {   // the same for every lane
          const int j = jb + jj;
          float s = 0.f;
#pragma unroll
          for (int d = 0; d < DKP; ++d) s = fmaf(qr[d], Ks[j * DKP + d], s);
          if (s > m) {
            const float c = expf(m - s);   // 0 on the first key (m = -inf)
            l *= c;
#pragma unroll
            for (int d = 0; d < DKP; ++d) acc[d] *= c;
            m = s;
          }
          const float p = expf(s - m);
          l += p;
          const float pk = kp ? (((kw[jj >> 2] >> (8 * (jj & 3))) & 0xffu) ? p * keep_scale : 0.f) : p;
#pragma unroll
          for (int d = 0; d < DKP; ++d) acc[d] = fmaf(pk, Vs[j * DKP + d], acc[d]);
        }
      }
    }
  }
  if (live) {
    const float r = 1.f / l;
#pragma unroll
    for (int d = 0; d < DKP; ++d)
      if (d < dk) out[(row0 + i) * ldo + h * dk + d] = acc[d] * r;
  }
}

// The same attention on the matrix cores (head size up to 32): fp32 v_mfma_f32_16x16x4_f32 tiles with the online softmax,
// so the n x n scores are neither stored nor recomputed.  A workgroup of eight waves owns 256 queries of one (clip, head),
// a wave two tiles of 16; the keys and values pass through LDS 64 at a time.  Per 16 keys and query tile:
//     S^T tile [key][query] = K Q^T         (DK4 products; a lane ends up with 4 consecutive keys of one query)
//     running max / rescale / p = exp(s-m)  (the maximum over a query's 16 keys: in-lane, then two shuffles; the scores
//                                            carry a factor log2(e) from the query operand, so exp is one v_exp_f32:
//                                            with the library expf the kernel spent more time there than in the products)
//     O^T [d][query] += V^T P^T             (4 products per 16 output columns; k-step s takes key 4q + s, the lane's own
//                                            register s, so P feeds the next product without moving)
// The keep mask of a lane's 4 keys is one 4-byte load.  Row pitches (DKP + 4 and 16 DT + 4 floats) put the 64 lanes of
// every operand read on 64 different banks.
constexpr int ATM_KB = 64, ATM_QT = 2;

// maximum over the lanes p, p + 16, p + 32, p + 48 (a query's four key groups): the two row-swap instructions of gfx950
// instead of two round trips through the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32 and a wait)
__device__ __forceinline__ float max_over_quarters(float v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int DK4, int DT>
__global__ __launch_bounds__(512) void tok_attention_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, int ld,
                                                                 const unsigned char* __restrict__ keep, float keep_scale,
                                                                 float* __restrict__ out, int ldo, int n, int H, int dk,
                                                                 float inv_temp) {
  constexpr int DKP = 4 * DK4, KP = DKP + 4, VP = 16 * DT + 4;
  __shared__ float ks[ATM_KB * KP];
  __shared__ float vs[ATM_KB * VP];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, qd = lane >> 4;
  const size_t row0 = (size_t)b * n;
  int qi[ATM_QT];
  float qreg[ATM_QT][DK4];
  f32x4 acc[ATM_QT][DT];
  float m[ATM_QT], l[ATM_QT];
  const unsigned char* kp[ATM_QT];
#pragma unroll
  for (int t = 0; t < ATM_QT; ++t) {
    qi[t] = blockIdx.x * (16 * ATM_QT * 8) + (wave * ATM_QT + t) * 16 + p;
    const int qc = qi[t] < n ? qi[t] : n - 1;
    const float* qrow = q + (row0 + qc) * ld + h * dk;
#pragma unroll
    for (int kk = 0; kk < DK4; ++kk) qreg[t][kk] = (4 * kk + qd < dk) ? qrow[4 * kk + qd] * (inv_temp * 1.44269504088896341f) : 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[t] = -INFINITY;
    l[t] = 0.f;
    kp[t] = keep ? keep + (((size_t)b * H + h) * n + qc) * n : nullptr;
  }
  // a block's keys and values are requested into registers while the previous block is multiplied
  constexpr int NV = ATM_KB * DKP / 512;
  float rk[NV], rv[NV];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + 512 * i, j = e / DKP, d = e - j * DKP;
      const bool in = j0 + j < n && d < dk;
      const size_t at = (row0 + (j0 + j < n ? j0 + j : n - 1)) * ld + h * dk + (d < dk ? d : 0);
      const float kv = k[at], vv = v[at];
      rk[i] = in ? kv : 0.f;
      rv[i] = in ? vv : 0.f;
    }
  };
  if (DKP < 16 * DT) {   // value columns the staging never reaches stay zero
    for (int e = threadIdx.x; e < ATM_KB * (16 * DT - DKP); e += 512) {
      const int j = e / (16 * DT - DKP), d = DKP + e - j * (16 * DT - DKP);
      vs[j * VP + d] = 0.f;
    }
  }
  fetch(0);
  for (int j0 = 0; j0 < n; j0 += ATM_KB) {
    __syncthreads();   // the previous block's products have read the tiles
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = threadIdx.x + 512 * i, j = e / DKP, d = e - j * DKP;
      ks[j * KP + d] = rk[i];
      if (d < 16 * DT) vs[j * VP + d] = rv[i];
    }
    __syncthreads();
    if (j0 + ATM_KB < n) fetch(j0 + ATM_KB);
    const int nkt = (min(n - j0, ATM_KB) + 15) >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
      const int key0 = 16 * kt;
      float ka[DK4];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) ka[kk] = ks[(key0 + p) * KP + 4 * kk + qd];
      float va[4][DT];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) va[s2][dt] = vs[(key0 + 4 * qd + s2) * VP + 16 * dt + p];
      const int kleft = n - (j0 + key0 + 4 * qd);   // keys of this lane that exist (may be <= 0 or >= 4)
#pragma unroll
      for (int t = 0; t < ATM_QT; ++t) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kk], qreg[t][kk], st, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r >= kleft) st[r] = -INFINITY;
        const float bm = max_over_quarters(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
        const float mn = fmaxf(m[t], bm);        // finite: every tile visited holds at least one existing key
        const float sc = __builtin_amdgcn_exp2f(m[t] - mn);
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f(st[r] - mn);
        l[t] = l[t] * sc + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
        if (kp[t]) {
          unsigned kw = 0;
          const unsigned char* kb = kp[t] + j0 + key0 + 4 * qd;
          if (kleft >= 4) __builtin_memcpy(&kw, kb, 4);
          else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
              if (r < kleft) kw |= (unsigned)kb[r] << (8 * r);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) pr[r] = ((kw >> (8 * r)) & 0xffu) ? pr[r] * keep_scale : 0.f;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][dt][r] *= sc;
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[s2][dt], pr[s2], acc[t][dt], 0, 0, 0);
        }
        m[t] = mn;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < ATM_QT; ++t) {
    float lt = l[t];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    if (qi[t] < n) {
      const float inv = 1.f / lt;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * dt + 4 * qd + r;
          if (c < dk) out[(row0 + qi[t]) * ldo + h * dk + c] = acc[t][dt][r] * inv;
        }
    }
  }
}

// ---- voice-activity logit of a clip from its first token (models.py:601): one wave per clip --------------------------------
__global__ __launch_bounds__(64) void tok_vad_kernel(const float* __restrict__ x, int ld, long tokens_per_clip,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ vad, int D) {
  const float* row = x + (size_t)blockIdx.x * tokens_per_clip * ld;
  float s = 0.f;
  for (int i = threadIdx.x; i < D; i += 64) s = fmaf(row[i], w[i], s);
  s = wave_reduce_sum(s);
  if (threadIdx.x == 0) vad[blockIdx.x] = s + bias[0];
}

}  // namespace

NOISE_EXPORT int noise_wave_normalize(const float* x, float* out, float* mx, int B, long L, void* stream) {
  NOISE_REQUIRE(x && out && mx, "noise_wave_normalize: null pointer");
  NOISE_REQUIRE(B > 0 && L > 0, "noise_wave_normalize: empty input");
  hipLaunchKernelGGL(wave_normalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, mx, L);
  NOISE_CHECK_LAUNCH("noise_wave_normalize");
  return 0;
}

NOISE_EXPORT int noise_wave_final(const float* y, const float* bias, const float* vad, const float* mx, float* out, int B,
                                  int Hy, int Wy, int H, int W, void* stream) {
  NOISE_REQUIRE(y && out, "noise_wave_final: null pointer");
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && Hy >= H && Wy >= W, "noise_wave_final: the network output (%d x %d) is smaller than "
                "the input (%d x %d)", Hy, Wy, H, W);
  hipLaunchKernelGGL(wave_final_kernel, dim3(noise_grid((long)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, y, bias, vad,
                     mx, out, B, Hy, Wy, H, W);
  NOISE_CHECK_LAUNCH("noise_wave_final");
  return 0;
}

NOISE_EXPORT int noise_tok_epilogue(const float* y, int ldy, const float* bias, int relu, const float* res, int ldr,
                                    const float* gamma, const float* beta, float eps, float* out, int ldo, float* pre, long M,
                                    int D, void* stream) {
  NOISE_REQUIRE(y && out, "noise_tok_epilogue: null pointer");
  NOISE_REQUIRE(M > 0 && D > 0 && D <= 64 * TOK_PER, "noise_tok_epilogue: 1 <= D <= %d (got %d)", 64 * TOK_PER, D);
  NOISE_REQUIRE(ldy >= D && ldo >= D && (!res || ldr >= D), "noise_tok_epilogue: row strides shorter than D");
  NOISE_REQUIRE((gamma == nullptr) == (beta == nullptr), "noise_tok_epilogue: LayerNorm needs both weight and bias");
  hipLaunchKernelGGL(tok_epilogue_kernel, dim3(noise_grid(M, 4, 4096)), dim3(256), 0, (hipStream_t)stream, y, ldy, bias, relu,
                     res, ldr, gamma, beta, eps, out, ldo, pre, M, D);
  NOISE_CHECK_LAUNCH("noise_tok_epilogue");
  return 0;
}

NOISE_EXPORT int noise_tok_attention(const float* q, const float* k, const float* v, int ld, const unsigned char* keep,
                                     float keep_scale, float* out, int ldo, int B, int n, int H, int dk, float temperature,
                                     void* stream) {
  NOISE_REQUIRE(q && k && v && out, "noise_tok_attention: null pointer");
  NOISE_REQUIRE(B > 0 && n > 0 && H > 0 && dk > 0 && dk <= 64, "noise_tok_attention: head size must be in 1..64 (got %d)", dk);
  NOISE_REQUIRE(B <= 65535 && H <= 65535, "noise_tok_attention: more than 65535 clips or heads in one call");
  NOISE_REQUIRE(ld >= H * dk && ldo >= H * dk && temperature > 0.f, "noise_tok_attention: bad strides or temperature");
  // short clips: workgroups of 64 or 128 queries, so that few lanes are left without a query
  const int nt = n <= 64 ? 64 : (n <= 128 ? 128 : 256);
  const dim3 grid(noise_cdiv(n, nt), H, B);
  hipStream_t st = (hipStream_t)stream;
  const float it = 1.f / temperature;
#define ATT_CASE(P)                                                                                                     \
  hipLaunchKernelGGL(tok_attention_kernel<P>, grid, dim3(nt), 0, st, q, k, v, ld, keep, keep_scale, out, ldo, n, H, dk, it)
  // heads of at most 32 run on the matrix cores (NOISE_TOK_ATT_VALU, diagnostic: the thread-per-query kernel)
  static const bool valu_only = getenv("NOISE_TOK_ATT_VALU") != nullptr;
  if (dk <= 32 && !valu_only) {
    const dim3 gm(noise_cdiv(n, 16 * ATM_QT * 8), H, B);
#define ATM_CASE(K4, T)                                                                                                 \
  hipLaunchKernelGGL((tok_attention_mfma_kernel<K4, T>), gm, dim3(512), 0, st, q, k, v, ld, keep, keep_scale, out, ldo, n, \
                     H, dk, it)
    if (dk <= 8) ATM_CASE(2, 1);
    else if (dk <= 16) ATM_CASE(4, 1);
    else if (dk <= 24) ATM_CASE(6, 2);
    else ATM_CASE(8, 2);
#undef ATM_CASE
    NOISE_CHECK_LAUNCH("noise_tok_attention");
    return 0;
  }
  if (dk <= 8) ATT_CASE(8);
  else if (dk <= 16) ATT_CASE(16);
  else if (dk <= 24) ATT_CASE(24);
  else if (dk <= 32) ATT_CASE(32);
  else ATT_CASE(64);
#undef ATT_CASE
  NOISE_CHECK_LAUNCH("noise_tok_attention");
  return 0;
}

NOISE_EXPORT int noise_tok_vad(const float* x, int ld, long tokens_per_clip, const float* w, const float* bias, float* vad, int B,
                               int D, void* stream) {
  NOISE_REQUIRE(x && w && bias && vad, "noise_tok_vad: null pointer");
  NOISE_REQUIRE(B > 0 && D > 0 && ld >= D && tokens_per_clip > 0, "noise_tok_vad: bad shape");
  hipLaunchKernelGGL(tok_vad_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, x, ld, tokens_per_clip, w, bias, vad, D);
  NOISE_CHECK_LAUNCH("noise_tok_vad");
  return 0;
}
