// Shape-generic backward kernels: with noise_gen_pw_dgrad (ofa.hip), noise_dw3x3_bwd / noise_first_conv_wgrad (conv.hip)
// and noise_bn_bwd (bn.hip) they differentiate the generic forward of generic.py, i.e. the training step of the
// waveform-domain U-Net (train/models.py:641-685 on the layer stack of :1096-1218).  Activations are NHWC fp32, token
// matrices row-major [tokens][features]; gradients of a conv output arrive in the lazy form of lazy.h (dz, y, coef3).
// Every reduction over pixels or tokens goes through per-workgroup partial rows (`slab`) folded in a fixed order by
// noise_gen_slab_reduce, so results do not depend on scheduling.
#include "lazy.h"

namespace {

// out[c] = sum_r slab[r][c].  A workgroup folds 64 columns: four row lanes take every fourth row each and are combined in a
// fixed order, so the result does not depend on scheduling.
__global__ __launch_bounds__(256) void gen_slab_reduce_kernel(const float* __restrict__ slab, int rows, long cols,
                                                              float* __restrict__ out) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const long c = (long)blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int r = rl;
    for (; r + 4 < rows; r += 8) {   // two independent chains per thread
      s0 += slab[(size_t)r * cols + c];
      s1 += slab[(size_t)(r + 4) * cols + c];
    }
    if (r < rows) s0 += slab[(size_t)r * cols + c];
  }
  part[rl][cl] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c < cols) out[c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// ---- weight gradient of a 1x1 layer: dW[co][ci] = sum_p dy[p][co] a[p][ci] ------------------------------------------------
// A workgroup owns a block of dW and walks pixel tiles from blockIdx.z: both operands are staged in LDS with their lazy
// transforms applied and multiplied by v_mfma_f32_16x16x4_f32 with k = pixels.  One operand (the "wave side", 64
// channels per workgroup, blockIdx.x) gives every wave one 16-channel tile; the other (the "loop side", 16 CT channels,
// blockIdx.y) is walked by every wave, so a staged wave-side value feeds CT matrix products.  SWAP = false: the wave side
// is dy (output channels), the loop side the layer input; SWAP = true the other way round, which is how a layer with few
// output channels keeps all four waves busy.  Wide loop sides (CT >= 8) take 32-pixel tiles, the others 64.
// Row pitches are 16 (mod 64) floats: the four pixel rows a matrix product reads fall into four disjoint bank groups.
// Partial blocks go to slab[blockIdx.z][Cout][Cin].
constexpr int WGR_P = 80;

template <int CT, bool SWAP>
__global__ __launch_bounds__(256) void gen_pw_wgrad_kernel(LazyGrad G, LazySrc A, float* __restrict__ slab, long npix, int N,
                                                           int K) {
  constexpr int PT = CT >= 8 ? 32 : 64;                 // pixels per tile
  constexpr int YP = CT == 1 ? 16 : 16 * CT + 16;       // loop-side row pitch
  constexpr int YQ = 4 * CT, YROWS = 256 / YQ;          // loop-side channel quads; rows one pass of the workgroup covers
  constexpr int YI = PT / YROWS, XI = PT / 16;          // float4 a thread stages per tile, loop side / wave side
  __shared__ float xs[PT * WGR_P];
  __shared__ float ys[PT * YP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int cx0 = blockIdx.x * 64, cy0 = blockIdx.y * 16 * CT;
  const int xq = tid & 15, xr0 = tid >> 4, yq = tid % YQ, yr0 = tid / YQ;
  // the gradient operand and the input operand, whichever side each is on
  constexpr int GI = SWAP ? YI : XI, GSTEP = SWAP ? YROWS : 16, GP = SWAP ? YP : WGR_P;
  constexpr int AI = SWAP ? XI : YI, ASTEP = SWAP ? 16 : YROWS, AP = SWAP ? WGR_P : YP;
  float* const gbuf = SWAP ? ys : xs;
  float* const abuf = SWAP ? xs : ys;
  const int gq = SWAP ? yq : xq, gr0 = SWAP ? yr0 : xr0, aq = SWAP ? xq : yq, ar0 = SWAP ? xr0 : yr0;
  const int co = (SWAP ? cy0 : cx0) + 4 * gq, ci = (SWAP ? cx0 : cy0) + 4 * aq;
  const bool von = co < N, vok = ci < K;
  const int coc = von ? co : 0, cic = vok ? ci : 0;
  const bool affine = G.coef3 != nullptr;
  float4 gs = f4(1.f), k1h = f4(0.f), k1l = f4(0.f), a2 = f4(0.f), mh = f4(0.f), ml = f4(0.f);
  if (affine) {
    gs = ld4(G.coef3 + coc);
    k1h = ld4(G.coef3 + N + coc);
    k1l = ld4(G.coef3 + 2 * N + coc);
    a2 = ld4(G.coef3 + 3 * N + coc);
    mh = ld4(G.coef3 + 4 * N + coc);
    ml = ld4(G.coef3 + 5 * N + coc);
  }
  const float4 asc = A.coef ? ld4(A.coef + cic) : f4(1.f), ash = A.coef ? ld4(A.coef + K + cic) : f4(0.f);
  const float alo = A.relu6 ? 0.f : -INFINITY, ahi = A.relu6 ? 6.f : INFINITY;
  // a wave whose 16 wave-side channels do not exist has nothing to multiply (it still stages and waits with the others)
  const bool wave_on = cx0 + 16 * wave < (SWAP ? K : N);
  f32x4 acc[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long ntiles = (npix + PT - 1) / PT;
  float4 rz[GI], ry[GI], ra[AI];
  auto fetch = [&](long tile) {
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const long pix = tile * PT + gr0 + GSTEP * i;
      const size_t pc = (size_t)(pix < npix ? pix : npix - 1);
      rz[i] = ld4(G.dz + pc * N + coc);
      ry[i] = affine ? ld4(G.y + pc * N + coc) : f4(0.f);
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const long pix = tile * PT + ar0 + ASTEP * i;
      const size_t pc = (size_t)(pix < npix ? pix : npix - 1);
      ra[i] = ld4(A.y + pc * K + cic);
    }
  };
  if ((long)blockIdx.z < ntiles) fetch(blockIdx.z);
  for (long tile = blockIdx.z; tile < ntiles; tile += gridDim.z) {
    __syncthreads();   // the previous tile's products have read the staged operands
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const bool in = tile * PT + gr0 + GSTEP * i < npix;
      const float4 dy = affine ? fma4(a2, sub4(sub4(ry[i], mh), ml), gs * sub4(sub4(rz[i], k1h), k1l)) : rz[i];
      st4(gbuf + (gr0 + GSTEP * i) * GP + 4 * gq, (in && von) ? dy : f4(0.f));
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const bool in = tile * PT + ar0 + ASTEP * i < npix;
      st4(abuf + (ar0 + ASTEP * i) * AP + 4 * aq, (in && vok) ? clamp4(fma4(ra[i], asc, ash), alo, ahi) : f4(0.f));
    }
    __syncthreads();
    // the next tile's memory round trip runs under this tile's matrix products
    if (tile + gridDim.z < ntiles) fetch(tile + gridDim.z);
    if (wave_on) {
#pragma unroll 2
      for (int s = 0; s < PT / 4; ++s) {
        const int pp = 4 * s + q;
        const float xv = xs[pp * WGR_P + 16 * wave + p];
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, ys[pp * YP + 16 * j + p], acc[j], 0, 0, 0);
      }
    }
  }
  float* blk = slab + (size_t)blockIdx.z * N * K;
  if (!wave_on) return;
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    if (SWAP) {   // acc[j][e]: input channel cx0 + 16 wave + 4 q + e, output channel cy0 + 16 j + p
      const int o = cy0 + 16 * j + p, i = cx0 + 16 * wave + 4 * q;
      if (o < N && i < K) st4(blk + (size_t)o * K + i, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = cx0 + 16 * wave + 4 * q + e, i = cy0 + 16 * j + p;
        if (o < N && i < K) blk[(size_t)o * K + i] = acc[j][e];
      }
    }
  }
}

// Blocking of the weight gradient for a Cout x Cin layer: which operand the waves split, and how many 16-channel tiles
// of the other one a workgroup walks.  Every (side, tile count) pair is priced from the matrix-core time of its padded
// block grid (tm) and the time to stage each operand once per block of the other side (tb, mostly L2 hits: the blocks
// of one pixel range run together), and the cheapest is taken.  The two overlap fully at four waves per SIMD (one or two
// tiles) and for about a fifth at three (four or eight tiles); sixteen tiles leave two waves per SIMD and no overlap.
// Constants fitted to a sweep of all ten blockings over the layer shapes of the waveform network
// (scripts/dbg/wgrad_sweep.py).
struct WgradPlan {
  bool swap;
  int ct, pt, nx, ny;
};
inline WgradPlan wgrad_plan(int Cout, int Cin) {
  WgradPlan best{false, 4, 64, noise_cdiv(Cout, 64), noise_cdiv(Cin, 64)};
  double best_t = 1e30;
  static const char* force = getenv("NOISE_WGRAD_PLAN");   // diagnostic: "<swap><ct>", e.g. "08" or "116"
  if (force && force[0]) {
    const bool sw = force[0] == '1';
    const int ct = atoi(force + 1);
    const int Cw = sw ? Cin : Cout, Cl = sw ? Cout : Cin;
    return WgradPlan{sw, ct, ct >= 8 ? 32 : 64, noise_cdiv(Cw, 64), noise_cdiv(Cl, 16 * ct)};
  }
  for (int sw = 0; sw < 2; ++sw) {
    const int Cw = sw ? Cin : Cout, Cl = sw ? Cout : Cin;
    const double gw = sw ? 1.0 : 2.0, gl = sw ? 2.0 : 1.0;   // the gradient operand is two tensors (dz and y)
    for (int ct = 1; ct <= 16; ct *= 2) {
      const int nx = noise_cdiv(Cw, 64), ny = noise_cdiv(Cl, 16 * ct);
      const double flops = 2.0 * nx * 64.0 * ny * 16.0 * ct, bytes = 4.0 * (ny * Cw * gw + nx * Cl * gl);
      const double tm = flops / 124e12 * (ct == 16 ? 1.15 : 1.0), tb = bytes / 6.6e12;
      const double ov = ct <= 2 ? 0.0 : (ct <= 8 ? 0.8 : 1.0);
      const double t = (tm > tb ? tm : tb) + ov * (tm < tb ? tm : tb) + 1e-3 * tb;
      if (t < best_t) {
        best_t = t;
        best = WgradPlan{sw != 0, ct, ct >= 8 ? 32 : 64, nx, ny};
      }
    }
  }
  return best;
}

// ---- column sums of a row-major matrix (bias gradients): slab[blockIdx.x][N] ----------------------------------------------
__global__ __launch_bounds__(256) void gen_colsum_kernel(const float* __restrict__ x, int ld, long M, int N,
                                                         float* __restrict__ slab) {
  __shared__ float red[256];
  const long chunk = (M + gridDim.x - 1) / gridDim.x;
  const long m0 = (long)blockIdx.x * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
  if (N >= 64) {   // a thread owns columns tid, tid + 256, ...
    // the 256 threads split into column lanes and row lanes (N = 128: two row lanes), eight loads in flight per thread;
    // the row lanes' partials meet in LDS in a fixed order
    const int CL = N >= 256 ? 256 : (N >= 128 ? 128 : 64), RL = 256 / CL;
    const int cl = threadIdx.x % CL, rl = threadIdx.x / CL;
    for (int c0 = 0; c0 < N; c0 += CL) {
      const int c = c0 + cl;
      float s = 0.f;
      if (c < N) {
        long m = m0 + rl;
        float t[8];
        for (; m + 7 * RL < m1; m += 8 * RL) {
#pragma unroll
          for (int u = 0; u < 8; ++u) t[u] = x[(size_t)(m + u * RL) * ld + c];
          s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        }
        for (; m < m1; m += RL) s += x[(size_t)m * ld + c];
      }
      __syncthreads();
      red[threadIdx.x] = s;
      __syncthreads();
      if (rl == 0 && c < N) {
        float sum = red[cl];
        for (int r = 1; r < RL; ++r) sum += red[r * CL + cl];
        slab[(size_t)blockIdx.x * N + c] = sum;
      }
    }
  } else {         // narrow matrices: the threads split the rows, then fold in a fixed order
    for (int c = 0; c < N; ++c) {
      float s = 0.f;
      for (long m = m0 + threadIdx.x; m < m1; m += 256) s += x[(size_t)m * ld + c];
      red[threadIdx.x] = s;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
      }
      if (threadIdx.x == 0) slab[(size_t)blockIdx.x * N + c] = red[0];
      __syncthreads();
    }
  }
}

// ---- elementwise pieces ---------------------------------------------------------------------------------------------------------
// out = h > 0 ? g : 0   (backward of the feed-forward ReLU, models.py:112)
__global__ __launch_bounds__(256) void gen_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h,
                                                           float* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 gv = ld4(g + 4 * i), hv = ld4(h + 4 * i);
    st4(out + 4 * i, make_float4(hv.x > 0.f ? gv.x : 0.f, hv.y > 0.f ? gv.y : 0.f, hv.z > 0.f ? gv.z : 0.f, hv.w > 0.f ? gv.w : 0.f));
  }
}

// dst[b][Hd - Hs + h][Wd - Ws + w][:] += src[b][h][w][:]   (gradient of `x + skip[..., -h:, -w:]`, models.py:608, and plain sums)
__global__ __launch_bounds__(256) void gen_add_cropped_kernel(float* __restrict__ dst, const float* __restrict__ src, int B,
                                                              int Hd, int Wd, int Hs, int Ws, int C4) {
  const long n = (long)B * Hs * Ws * C4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % Ws);
    t /= Ws;
    const int h = (int)(t % Hs), b = (int)(t / Hs);
    float* d = dst + ((((size_t)b * Hd + (Hd - Hs + h)) * Wd + (Wd - Ws + w)) * C4 + c) * 4;
    st4(d, ld4(d) + ld4(src + 4 * i));
  }
}

// ---- LayerNorm backward (models.py:90, :116, :214): one wave per token ----------------------------------------------------
// y = (r - mean) * rstd * gamma + beta;  dr = rstd * (g' - mean(g') - xhat * mean(g' xhat)), g' = g * gamma  (+ add);
// per-workgroup partial sums of g * xhat and g go to slab[blockIdx.x][2D].
constexpr int LNB_PER = 16;

__global__ __launch_bounds__(256) void gen_ln_bwd_kernel(const float* __restrict__ g, const float* __restrict__ r,
                                                         const float* __restrict__ gamma, float eps,
                                                         const float* __restrict__ add, float* __restrict__ out,
                                                         float* __restrict__ slab, long M, int D) {
  __shared__ float red[4 * 2 * 64 * LNB_PER];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[LNB_PER], db[LNB_PER], gm[LNB_PER];
#pragma unroll
  for (int j = 0; j < LNB_PER; ++j) {
    dg[j] = db[j] = 0.f;
    gm[j] = lane + 64 * j < D ? gamma[lane + 64 * j] : 0.f;
  }
  for (long t = (long)blockIdx.x * 4 + wave; t < M; t += (long)gridDim.x * 4) {
    float v[LNB_PER], gv[LNB_PER];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_PER; ++j) {
      const int i = lane + 64 * j;
      v[j] = i < D ? r[(size_t)t * D + i] : 0.f;
      gv[j] = i < D ? g[(size_t)t * D + i] : 0.f;
      s += v[j];
    }
    const float mean = wave_reduce_sum(s) / (float)D;
    float qq = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_PER; ++j) {
      const float dlt = v[j] - mean;
      qq += (lane + 64 * j < D) ? dlt * dlt : 0.f;
    }
    const float rstd = 1.f / sqrtf(wave_reduce_sum(qq) / (float)D + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LNB_PER; ++j) {
      const bool in = lane + 64 * j < D;
      const float xh = in ? (v[j] - mean) * rstd : 0.f;
      const float gp = gv[j] * gm[j];
      s1 += gp;
      s2 += gp * xh;
      dg[j] += gv[j] * xh;
      db[j] += gv[j];
      v[j] = xh;
      gv[j] = gp;
    }
    const float m1 = wave_reduce_sum(s1) / (float)D, m2 = wave_reduce_sum(s2) / (float)D;
#pragma unroll
    for (int j = 0; j < LNB_PER; ++j) {
      const int i = lane + 64 * j;
      if (i < D) {
        float o = rstd * (gv[j] - m1 - v[j] * m2);
        if (add) o += add[(size_t)t * D + i];
        out[(size_t)t * D + i] = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < LNB_PER; ++j) {
    red[((wave * 2 + 0) * LNB_PER + j) * 64 + lane] = dg[j];
    red[((wave * 2 + 1) * LNB_PER + j) * 64 + lane] = db[j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 256) {
    const int k = i / D, c = i - k * D, j = c >> 6, l = c & 63;
    float sum = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) sum += red[((wv * 2 + k) * LNB_PER + j) * 64 + l];
    slab[(size_t)blockIdx.x * 2 * D + i] = sum;
  }
}

// ---- attention backward (models.py:26-36): two passes, both recompute the scores ----------------------------------------------
// Pass Q (a thread owns a query): softmax statistics (max of the scores in units of log2, sum) and D_i = dO_i . O_i, then dq_i = sum_j ds_ij k_j / T with
// ds_ij = p_ij (dp_ij - D_i), dp_ij = keep_ij * keep_scale * (dO_i . v_j).  The three statistics are kept for pass KV (a
// thread owns a key): dv_j = sum_i p_ij keep_ij keep_scale dO_i, dk_j = sum_i ds_ij q_i / T.  No atomics.
constexpr int ATB_KB = 128;
constexpr float LOG2E = 1.44269504088896341f;

// 16 consecutive keep bytes of a query's row (any alignment) as four words; only the first `count` are read from memory
// (byte-sized loads along a thread's own row touch a different cache line in every lane for every key).
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
__global__ __launch_bounds__(256) void gen_attn_bwd_q_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int ld,
                                                             const unsigned char* __restrict__ keep, float keep_scale,
                                                             const float* __restrict__ o, const float* __restrict__ go, int ldo,
                                                             float* __restrict__ dq, float* __restrict__ stats, int n, int H,
                                                             int dk, float inv_temp) {
  __shared__ float Ks[ATB_KB * DKP];
  __shared__ float Vs[ATB_KB * DKP];
  const int h = blockIdx.y, b = blockIdx.z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  const size_t row0 = (size_t)b * n, ic = live ? i : 0;
  float qr[DKP], gr[DKP], acc[DKP];
  float Di = 0.f;
#pragma unroll
  for (int d = 0; d < DKP; ++d) {
    const bool in = live && d < dk;
    qr[d] = in ? q[(row0 + ic) * ld + h * dk + d] * (inv_temp * LOG2E) : 0.f;   // scores in units of log2: exp is v_exp_f32
    gr[d] = in ? go[(row0 + ic) * ldo + h * dk + d] : 0.f;
    Di = fmaf(gr[d], in ? o[(row0 + ic) * ldo + h * dk + d] : 0.f, Di);
    acc[d] = 0.f;
  }
  const unsigned char* kp = keep ? keep + (((size_t)b * H + h) * n + ic) * n : nullptr;
  float m = -INFINITY, l = 0.f;
  for (int pass = 0; pass < 2; ++pass) {
    const float rl = pass ? 1.f / l : 0.f;
    for (int j0 = 0; j0 < n; j0 += ATB_KB) {
      __syncthreads();
      for (int e = threadIdx.x; e < ATB_KB * DKP; e += blockDim.x) {
        const int j = e / DKP, d = e - j * DKP;
        const bool in = j0 + j < n && d < dk;
        Ks[e] = in ? k[(row0 + j0 + j) * ld + h * dk + d] : 0.f;
        if (pass) Vs[e] = in ? v[(row0 + j0 + j) * ld + h * dk + d] : 0.f;
      }
      __syncthreads();
      const int jn = n - j0 < ATB_KB ? n - j0 : ATB_KB;
      if (pass == 0) {
        for (int j = 0; j < jn; ++j) {
          float s = 0.f;
#pragma unroll
          for (int d = 0; d < DKP; ++d) s = fmaf(qr[d], Ks[j * DKP + d], s);
          if (s > m) {
            l *= __builtin_amdgcn_exp2f(m - s);
            m = s;
          }
          l += __builtin_amdgcn_exp2f(s - m);
        }
      } else {
        for (int jb = 0; jb < jn; jb += 16) {
          const int kn = jn - jb < 16 ? jn - jb : 16;
          unsigned kw[4];
          if (kp) keep_bytes16(kp + j0 + jb, kn, kw);
#pragma unroll 1
          for (int wi = 0; wi < 4; ++wi) {
            const unsigned w = wi == 0 ? kw[0] : (wi == 1 ? kw[1] : (wi == 2 ? kw[2] : kw[3]));
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            if (4 * wi + jj < kn) {   // the same for every lane
              const int j = jb + 4 * wi + jj;
              float s = 0.f;
#pragma unroll
              for (int d = 0; d < DKP; ++d) s = fmaf(qr[d], Ks[j * DKP + d], s);
              const float pj = __builtin_amdgcn_exp2f(s - m) * rl;
              float dp = 0.f;
#pragma unroll
              for (int d = 0; d < DKP; ++d) dp = fmaf(gr[d], Vs[j * DKP + d], dp);
              if (kp) dp = ((w >> (8 * jj)) & 0xffu) ? dp * keep_scale : 0.f;
              const float ds = pj * (dp - Di);
#pragma unroll
              for (int d = 0; d < DKP; ++d) acc[d] = fmaf(ds, Ks[j * DKP + d], acc[d]);
            }
          }
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int d = 0; d < DKP; ++d)
      if (d < dk) dq[(row0 + i) * ld + h * dk + d] = acc[d] * inv_temp;
    float* st = stats + (((size_t)b * H + h) * n + i) * 3;
    st[0] = m;
    st[1] = 1.f / l;
    st[2] = Di;
  }
}

template <int DKP, int QB>
__global__ __launch_bounds__(256) void gen_attn_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, int ld,
                                                              const unsigned char* __restrict__ keep, float keep_scale,
                                                              const float* __restrict__ go, int ldo,
                                                              const float* __restrict__ stats, float* __restrict__ dkk,
                                                              float* __restrict__ dv, int n, int H, int dk, float inv_temp) {
  __shared__ float Qs[QB * DKP];
  __shared__ float Gs[QB * DKP];
  __shared__ float St[QB * 3];
  const int h = blockIdx.y, b = blockIdx.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = j < n;
  const size_t row0 = (size_t)b * n, jc = live ? j : 0;
  float kr[DKP], vr[DKP], ak[DKP], av[DKP];
#pragma unroll
  for (int d = 0; d < DKP; ++d) {
    const bool in = live && d < dk;
    kr[d] = in ? k[(row0 + jc) * ld + h * dk + d] : 0.f;
    vr[d] = in ? v[(row0 + jc) * ld + h * dk + d] : 0.f;
    ak[d] = av[d] = 0.f;
  }
  const unsigned char* kp = keep ? keep + ((size_t)b * H + h) * n * n + jc : nullptr;
  for (int i0 = 0; i0 < n; i0 += QB) {
    __syncthreads();
    for (int e = threadIdx.x; e < QB * DKP; e += blockDim.x) {
      const int i = e / DKP, d = e - i * DKP;
      const bool in = i0 + i < n && d < dk;
      Qs[e] = in ? q[(row0 + i0 + i) * ld + h * dk + d] * inv_temp : 0.f;
      Gs[e] = in ? go[(row0 + i0 + i) * ldo + h * dk + d] : 0.f;
    }
    for (int e = threadIdx.x; e < QB * 3; e += blockDim.x)
      St[e] = i0 + e / 3 < n ? stats[(((size_t)b * H + h) * n + i0) * 3 + e] : 0.f;
    __syncthreads();
    const int in_ = n - i0 < QB ? n - i0 : QB;
    for (int ib = 0; ib < in_; ib += 4) {
      const int kn = in_ - ib < 4 ? in_ - ib : 4;
      // the keep bytes of 4 queries are requested together (a column of the mask: consecutive lanes, consecutive bytes)
      unsigned char kb[4];
      if (kp) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) kb[ii] = ii < kn ? kp[(size_t)(i0 + ib + ii) * n] : 0;
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        if (ii < kn) {   // the same for every lane
          const int i = ib + ii;
          float s = 0.f, dp = 0.f;
#pragma unroll
          for (int d = 0; d < DKP; ++d) {
            s = fmaf(Qs[i * DKP + d], kr[d], s);
            dp = fmaf(Gs[i * DKP + d], vr[d], dp);
          }
          const float pj = __builtin_amdgcn_exp2f(s * LOG2E - St[3 * i]) * St[3 * i + 1];   // the stored maximum is in units of log2
          float pm = pj;
          if (kp) {
            const bool kept = kb[ii] != 0;
            pm = kept ? pj * keep_scale : 0.f;
            dp = kept ? dp * keep_scale : 0.f;
          }
          const float ds = pj * (dp - St[3 * i + 2]);
#pragma unroll
          for (int d = 0; d < DKP; ++d) {
            av[d] = fmaf(pm, Gs[i * DKP + d], av[d]);
            ak[d] = fmaf(ds, Qs[i * DKP + d], ak[d]);
          }
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int d = 0; d < DKP; ++d) {
      if (d < dk) {
        dkk[(row0 + j) * ld + h * dk + d] = ak[d];
        dv[(row0 + j) * ld + h * dk + d] = av[d];
      }
    }
  }
}

// ---- the same backward on the matrix cores (head size up to 32) -----------------------------------------------------------------
// Both passes keep the score tiles in registers in the layout the fp32 MFMA leaves them in, which is the operand layout of
// the next product (see tok_attention_mfma_kernel in wave.hip):
//   pass Q  (a wave owns query tiles, keys stream through LDS): S^T = K Q^T and dP^T = V dO^T give a lane 4 keys of one
//           query; ds = p (dp - D) feeds dQ^T += K^T dS^T with k = the lane's own keys.  A first sweep over the keys finds
//           the softmax statistics (kept for pass KV with D_i = dO_i . O_i).
//   pass KV (a wave owns key tiles, queries stream through LDS): S = Q K^T and dP = dO V^T give a lane 4 queries of one
//           key; dV^T += dO^T P and dK^T += Q^T dS with k = the lane's own queries.
// Scores carry log2(e), so exp is one v_exp_f32.  Rows of 36 floats put the 64 lanes of every operand read on 64 banks.
constexpr int ABM_P = 36, ABM_BLK = 64, ABM_T = 2;

__device__ __forceinline__ float max_over_quarters(float v) {   // over lanes p, p + 16, p + 32, p + 48
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_over_quarters(float v) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned u = __float_as_uint(v);
  u32x2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(r[0]) + __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// rows j0 .. j0 + ABM_BLK - 1 of one head (dk columns from col0) on their way into a [ABM_BLK][ABM_P] tile, in two steps
// so that a block's memory round trip runs under the previous block's products: abm_fetch requests them into registers
// (columns dk .. CW - 1 and rows past n: zero), abm_put writes the registers to LDS (times scale).
template <int CW>
struct AbmRegs {
  float v[ABM_BLK * CW / 512];
};
template <int CW>
__device__ __forceinline__ void abm_fetch(AbmRegs<CW>& r, const float* __restrict__ src, size_t row0, int ld, int col0, int j0,
                                          int n, int dk) {
#pragma unroll
  for (int i = 0; i < ABM_BLK * CW / 512; ++i) {
    const int e = threadIdx.x + 512 * i, j = e / CW, d = e - j * CW;
    const bool in = j0 + j < n && d < dk;
    const float v = src[(row0 + (j0 + j < n ? j0 + j : n - 1)) * ld + col0 + (d < dk ? d : 0)];
    r.v[i] = in ? v : 0.f;
  }
}
template <int CW>
__device__ __forceinline__ void abm_put(const AbmRegs<CW>& r, float* dst, float scale) {
#pragma unroll
  for (int i = 0; i < ABM_BLK * CW / 512; ++i) {
    const int e = threadIdx.x + 512 * i, j = e / CW, d = e - j * CW;
    dst[j * ABM_P + d] = r.v[i] * scale;
  }
}

template <int DK4, int DT>
__global__ __launch_bounds__(512) void gen_attn_bwd_q_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ v, int ld,
                                                                  const unsigned char* __restrict__ keep, float keep_scale,
                                                                  const float* __restrict__ o, const float* __restrict__ go,
                                                                  int ldo, float* __restrict__ dq, float* __restrict__ stats,
                                                                  int n, int H, int dk, float inv_temp) {
  constexpr int CW = DK4 * 4 > 16 * DT ? DK4 * 4 : 16 * DT;
  __shared__ float ks[ABM_BLK * ABM_P];
  __shared__ float vs[ABM_BLK * ABM_P];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, qd = lane >> 4;
  const size_t row0 = (size_t)b * n;
  const float kscale = keep ? keep_scale : 1.f;   // without a mask nothing is rescaled
  int qi[ABM_T];
  float qreg[ABM_T][DK4], greg[ABM_T][DK4], m[ABM_T], l[ABM_T], Di[ABM_T];
  const unsigned char* kp[ABM_T];
#pragma unroll
  for (int t = 0; t < ABM_T; ++t) {
    qi[t] = blockIdx.x * (16 * ABM_T * 8) + (wave * ABM_T + t) * 16 + p;
    const int qc = qi[t] < n ? qi[t] : n - 1;
    const float* qrow = q + (row0 + qc) * ld + h * dk;
    const float* grow = go + (row0 + qc) * ldo + h * dk;
    const float* orow = o + (row0 + qc) * ldo + h * dk;
    float dpart = 0.f;
#pragma unroll
    for (int kk = 0; kk < DK4; ++kk) {
      const int d = 4 * kk + qd;
      const bool in = d < dk;
      qreg[t][kk] = in ? qrow[d] * (inv_temp * LOG2E) : 0.f;
      greg[t][kk] = in ? grow[d] : 0.f;
      dpart = fmaf(greg[t][kk], in ? orow[d] : 0.f, dpart);
    }
    Di[t] = sum_over_quarters(dpart);
    m[t] = -INFINITY;
    l[t] = 0.f;
    kp[t] = keep ? keep + (((size_t)b * H + h) * n + qc) * n : nullptr;
  }
  AbmRegs<CW> rk, rv;
  // ---- sweep 1: softmax statistics
  abm_fetch<CW>(rk, k, row0, ld, h * dk, 0, n, dk);
  for (int j0 = 0; j0 < n; j0 += ABM_BLK) {
    __syncthreads();   // the previous block's products have read the tile
    abm_put<CW>(rk, ks, 1.f);
    __syncthreads();
    if (j0 + ABM_BLK < n) abm_fetch<CW>(rk, k, row0, ld, h * dk, j0 + ABM_BLK, n, dk);
    const int nkt = (min(n - j0, ABM_BLK) + 15) >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
      const int key0 = 16 * kt;
      float ka[DK4];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) ka[kk] = ks[(key0 + p) * ABM_P + 4 * kk + qd];
      const int kleft = n - (j0 + key0 + 4 * qd);
#pragma unroll
      for (int t = 0; t < ABM_T; ++t) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kk], qreg[t][kk], st, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r >= kleft) st[r] = -INFINITY;
        const float bm = max_over_quarters(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
        const float mn = fmaxf(m[t], bm);   // finite: the tile holds at least one existing key
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += __builtin_amdgcn_exp2f(st[r] - mn);
        l[t] = l[t] * __builtin_amdgcn_exp2f(m[t] - mn) + sum;
        m[t] = mn;
      }
    }
  }
  float rl[ABM_T];
  f32x4 acc[ABM_T][DT];
#pragma unroll
  for (int t = 0; t < ABM_T; ++t) {
    rl[t] = 1.f / sum_over_quarters(l[t]);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- sweep 2: ds and dQ
  abm_fetch<CW>(rk, k, row0, ld, h * dk, 0, n, dk);
  abm_fetch<CW>(rv, v, row0, ld, h * dk, 0, n, dk);
  for (int j0 = 0; j0 < n; j0 += ABM_BLK) {
    __syncthreads();
    abm_put<CW>(rk, ks, 1.f);
    abm_put<CW>(rv, vs, 1.f);
    __syncthreads();
    if (j0 + ABM_BLK < n) {
      abm_fetch<CW>(rk, k, row0, ld, h * dk, j0 + ABM_BLK, n, dk);
      abm_fetch<CW>(rv, v, row0, ld, h * dk, j0 + ABM_BLK, n, dk);
    }
    const int nkt = (min(n - j0, ABM_BLK) + 15) >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
      const int key0 = 16 * kt;
      float ka[DK4], va[DK4], kb[4][DT];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) {
        ka[kk] = ks[(key0 + p) * ABM_P + 4 * kk + qd];
        va[kk] = vs[(key0 + p) * ABM_P + 4 * kk + qd];
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) kb[s2][dt] = ks[(key0 + 4 * qd + s2) * ABM_P + 16 * dt + p];
      const int kleft = n - (j0 + key0 + 4 * qd);
#pragma unroll
      for (int t = 0; t < ABM_T; ++t) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[kk], qreg[t][kk], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(va[kk], greg[t][kk], dp, 0, 0, 0);
        }
        unsigned kw = 0x01010101u;
        if (kp[t]) {
          const unsigned char* kbp = kp[t] + j0 + key0 + 4 * qd;
          kw = 0;
          if (kleft >= 4) __builtin_memcpy(&kw, kbp, 4);
          else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
              if (r < kleft) kw |= (unsigned)kbp[r] << (8 * r);
          }
        }
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pj = r < kleft ? __builtin_amdgcn_exp2f(st[r] - m[t]) * rl[t] : 0.f;
          const float dpk = ((kw >> (8 * r)) & 0xffu) ? dp[r] * kscale : 0.f;
          ds[r] = pj * (dpk - Di[t]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kb[s2][dt], ds[s2], acc[t][dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < ABM_T; ++t) {
    if (qi[t] < n) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * dt + 4 * qd + r;
          if (c < dk) dq[(row0 + qi[t]) * ld + h * dk + c] = acc[t][dt][r] * inv_temp;
        }
      if (qd == 0) {
        float* st = stats + (((size_t)b * H + h) * n + qi[t]) * 3;
        st[0] = m[t];
        st[1] = rl[t];
        st[2] = Di[t];
      }
    }
  }
}

template <int DK4, int DT>
__global__ __launch_bounds__(512) void gen_attn_bwd_kv_mfma_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                   const float* __restrict__ v, int ld,
                                                                   const unsigned char* __restrict__ keep, float keep_scale,
                                                                   const float* __restrict__ go, int ldo,
                                                                   const float* __restrict__ stats, float* __restrict__ dkk,
                                                                   float* __restrict__ dv, int n, int H, int dk,
                                                                   float inv_temp) {
  constexpr int CW = DK4 * 4 > 16 * DT ? DK4 * 4 : 16 * DT;
  __shared__ float qs[ABM_BLK * ABM_P];
  __shared__ float gs[ABM_BLK * ABM_P];
  __shared__ float sts[ABM_BLK * 3];
  const int h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, qd = lane >> 4;
  const size_t row0 = (size_t)b * n;
  const float kscale = keep ? keep_scale : 1.f;   // without a mask nothing is rescaled
  int kj[ABM_T];
  float kreg[ABM_T][DK4], vreg[ABM_T][DK4];
  f32x4 accK[ABM_T][DT], accV[ABM_T][DT];
  const unsigned char* kp[ABM_T];
#pragma unroll
  for (int t = 0; t < ABM_T; ++t) {
    kj[t] = blockIdx.x * (16 * ABM_T * 8) + (wave * ABM_T + t) * 16 + p;
    const int jc = kj[t] < n ? kj[t] : n - 1;
    const float* krow = k + (row0 + jc) * ld + h * dk;
    const float* vrow = v + (row0 + jc) * ld + h * dk;
#pragma unroll
    for (int kk = 0; kk < DK4; ++kk) {
      const int d = 4 * kk + qd;
      kreg[t][kk] = d < dk ? krow[d] : 0.f;
      vreg[t][kk] = d < dk ? vrow[d] : 0.f;
    }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) accK[t][dt] = accV[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    kp[t] = keep ? keep + ((size_t)b * H + h) * n * n + jc : nullptr;
  }
  AbmRegs<CW> rq, rg;
  float rs = 0.f;
  const float* const srow = stats + ((size_t)b * H + h) * n * 3;
  abm_fetch<CW>(rq, q, row0, ld, h * dk, 0, n, dk);
  abm_fetch<CW>(rg, go, row0, ldo, h * dk, 0, n, dk);
  if (threadIdx.x < ABM_BLK * 3) rs = (int)threadIdx.x / 3 < n ? srow[threadIdx.x] : 0.f;
  for (int i0 = 0; i0 < n; i0 += ABM_BLK) {
    __syncthreads();   // the previous block's products have read the tiles
    abm_put<CW>(rq, qs, inv_temp);
    abm_put<CW>(rg, gs, 1.f);
    if (threadIdx.x < ABM_BLK * 3) sts[threadIdx.x] = rs;
    __syncthreads();
    if (i0 + ABM_BLK < n) {
      const int i1 = i0 + ABM_BLK;
      abm_fetch<CW>(rq, q, row0, ld, h * dk, i1, n, dk);
      abm_fetch<CW>(rg, go, row0, ldo, h * dk, i1, n, dk);
      if (threadIdx.x < ABM_BLK * 3) rs = i1 + (int)threadIdx.x / 3 < n ? srow[(size_t)i1 * 3 + threadIdx.x] : 0.f;
    }
    const int nqt = (min(n - i0, ABM_BLK) + 15) >> 4;
    for (int qt = 0; qt < nqt; ++qt) {
      const int q0 = 16 * qt;
      float qa[DK4], ga[DK4], qb[4][DT], gb[4][DT], sm[4], srl[4], sD[4];
#pragma unroll
      for (int kk = 0; kk < DK4; ++kk) {
        qa[kk] = qs[(q0 + p) * ABM_P + 4 * kk + qd];
        ga[kk] = gs[(q0 + p) * ABM_P + 4 * kk + qd];
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          qb[s2][dt] = qs[(q0 + 4 * qd + s2) * ABM_P + 16 * dt + p];
          gb[s2][dt] = gs[(q0 + 4 * qd + s2) * ABM_P + 16 * dt + p];
        }
        sm[s2] = sts[(q0 + 4 * qd + s2) * 3];
        srl[s2] = sts[(q0 + 4 * qd + s2) * 3 + 1];
        sD[s2] = sts[(q0 + 4 * qd + s2) * 3 + 2];
      }
      const int qleft = n - (i0 + q0 + 4 * qd);   // queries of this lane that exist
      // the keep bytes of the lane's (query, key) pairs are requested before the products that need them
      unsigned char kbyte[ABM_T][4];
      if (keep) {
#pragma unroll
        for (int t = 0; t < ABM_T; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) kbyte[t][r] = r < qleft ? kp[t][(size_t)(i0 + q0 + 4 * qd + r) * n] : 0;
      }
#pragma unroll
      for (int t = 0; t < ABM_T; ++t) {
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < DK4; ++kk) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kk], kreg[t][kk], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[kk], vreg[t][kk], dp, 0, 0, 0);
        }
        float pm[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pj = r < qleft ? __builtin_amdgcn_exp2f(st[r] * LOG2E - sm[r]) * srl[r] : 0.f;
          const bool kept = keep ? kbyte[t][r] != 0 : true;
          pm[r] = kept ? pj * kscale : 0.f;
          const float dpk = kept ? dp[r] * kscale : 0.f;
          ds[r] = pj * (dpk - sD[r]);
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            accV[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(gb[s2][dt], pm[s2], accV[t][dt], 0, 0, 0);
            accK[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qb[s2][dt], ds[s2], accK[t][dt], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < ABM_T; ++t) {
    if (kj[t] < n) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * dt + 4 * qd + r;
          if (c < dk) {
            dkk[(row0 + kj[t]) * ld + h * dk + c] = accK[t][dt][r];
            dv[(row0 + kj[t]) * ld + h * dk + c] = accV[t][dt][r];
          }
        }
    }
  }
}

// ---- VAD head backward (models.py:601): gx[b][token 0][:] += dvad[b] * w; slab-free: one workgroup folds dw and db --------
__global__ __launch_bounds__(256) void gen_vad_bwd_kernel(const float* __restrict__ x, float* __restrict__ gx, int ld,
                                                          long tokens_per_clip, const float* __restrict__ w,
                                                          const float* __restrict__ dvad, float* __restrict__ dw,
                                                          float* __restrict__ db, int B, int D) {
  __shared__ float red[256];
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const size_t row = (size_t)b * tokens_per_clip * ld;
      s = fmaf(dvad[b], x[row + d], s);
      gx[row + d] += dvad[b] * w[d];
    }
    dw[d] = s;
  }
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += dvad[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[0] = red[0];
}

// ---- last layer backward: C -> 1 transposed 3x3, stride 2, padding 1 (models.py:1164-1171) ----------------------------------
// out[b][Y][X] = bias + sum_c sum_k a[b][(Y+1-ky)/2][(X+1-kx)/2][c] w[c][k]; given d = dL/dout on the full (2Hi-1) x (2Wi-1)
// output: g[b][iy][ix][c] = sum_k d[b][2iy-1+ky][2ix-1+kx] w[c][k], dW[c][k] = sum a d.  A thread owns 4 channels of an
// input pixel; per-workgroup partial dW rows ([9][C]) go to the slab.
__global__ __launch_bounds__(256) void gen_last_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                           const float* __restrict__ w, float* __restrict__ g,
                                                           float* __restrict__ slab, int B, int Hi, int Wi, int C, int cl) {
  extern __shared__ float red[];   // [256][36]
  const int CG = C >> 2, PL = 256 / cl;
  const int cgl = threadIdx.x & (cl - 1), pl = threadIdx.x / cl;
  const bool ok = cgl < CG;
  const int c = 4 * (ok ? cgl : CG - 1);
  const int Hy = 2 * Hi - 1, Wy = 2 * Wi - 1;
  float4 wt[9], dwk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    wt[t] = make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t], w[(c + 3) * 9 + t]);
    dwk[t] = f4(0.f);
  }
  const long npix = (long)B * Hi * Wi;
  for (long px = (long)blockIdx.x * PL + pl; px < npix; px += (long)gridDim.x * PL) {
    const int ix = (int)(px % Wi);
    const long t = px / Wi;
    const int iy = (int)(t % Hi), b = (int)(t / Hi);
    const float4 av = ok ? ld4(a + (size_t)px * C + c) : f4(0.f);
    float4 acc = f4(0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int Y = 2 * iy - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int X = 2 * ix - 1 + kx;
        const bool in = Y >= 0 && Y < Hy && X >= 0 && X < Wy;
        const float d = in ? dout[((size_t)b * Hy + (in ? Y : 0)) * Wy + (in ? X : 0)] : 0.f;
        acc = fma4(f4(d), wt[ky * 3 + kx], acc);
        dwk[ky * 3 + kx] = fma4(f4(d), av, dwk[ky * 3 + kx]);
      }
    }
    if (ok) st4(g + (size_t)px * C + c, acc);
  }
  float* mine = red + threadIdx.x * 36;
#pragma unroll
  for (int t = 0; t < 9; ++t) st4(mine + 4 * t, dwk[t]);
  __syncthreads();
  if (pl == 0 && ok) {
    for (int t = 0; t < 9; ++t) {
      float4 s = f4(0.f);
      for (int r = 0; r < PL; ++r) s = s + ld4(red + (r * cl + threadIdx.x) * 36 + 4 * t);
      // slab row layout [C][9] like the weight tensor
      float* row = slab + (size_t)blockIdx.x * 9 * C;
      row[(c + 0) * 9 + t] = s.x;
      row[(c + 1) * 9 + t] = s.y;
      row[(c + 2) * 9 + t] = s.z;
      row[(c + 3) * 9 + t] = s.w;
    }
  }
}

inline int pow2_lanes(int C) {
  int cl = 1;
  while (cl < C / 4 && cl < 64) cl <<= 1;
  return cl;
}

}  // namespace

NOISE_EXPORT int noise_gen_slab_reduce(const float* slab, int rows, long cols, float* out, void* stream) {
  NOISE_REQUIRE(slab && out && rows > 0 && cols > 0, "noise_gen_slab_reduce: bad argument");
  hipLaunchKernelGGL(gen_slab_reduce_kernel, dim3(noise_cdiv(cols, 64)), dim3(256), 0, (hipStream_t)stream, slab, rows, cols, out);
  NOISE_CHECK_LAUNCH("noise_gen_slab_reduce");
  return 0;
}

// rows of the partial slab noise_gen_pw_wgrad writes for npix pixels (each Cout*Cin floats)
// resident workgroups per CU of the instance a plan selects (cached)
static int wgrad_per_cu(const WgradPlan& pl) {
  static int cache[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  const int ci = pl.ct == 1 ? 0 : (pl.ct == 2 ? 1 : (pl.ct == 4 ? 2 : (pl.ct == 8 ? 3 : 4)));
  int& v = cache[pl.swap ? 1 : 0][ci];
  if (v == 0) {
    int n = 0;
    hipError_t e = hipErrorUnknown;
#define WG_OCC(T)                                                                                                       \
  case T:                                                                                                               \
    e = pl.swap ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gen_pw_wgrad_kernel<T, true>, 256, 0)               \
                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gen_pw_wgrad_kernel<T, false>, 256, 0);             \
    break;
    switch (pl.ct) { WG_OCC(1) WG_OCC(2) WG_OCC(4) WG_OCC(8) default: WG_OCC(16) }
#undef WG_OCC
    v = (e == hipSuccess && n >= 1) ? n : 3;
  }
  return v;
}

NOISE_EXPORT int noise_gen_pw_wgrad_rows(long npix, int Cout, int Cin) {
  const WgradPlan pl = wgrad_plan(Cout, Cin);
  const long ntiles = (npix + pl.pt - 1) / pl.pt;
  // one round of resident workgroups over the whole grid: more rows only lengthen the fold, and a partly filled second
  // round would cost a full one (every workgroup carries the same share of the tiles)
  const long blocks = (long)pl.nx * pl.ny, round = (long)noise_num_cus() * wgrad_per_cu(pl);
  long rows = round / blocks;
  const long cap = (64L << 20) / ((long)Cout * Cin * 4);   // at most 64 MiB of partials
  if (rows > cap) rows = cap;
  if (rows > ntiles) rows = ntiles;
  return rows < 1 ? 1 : (int)rows;
}

NOISE_EXPORT int noise_gen_pw_wgrad(const float* dz, const float* y_out, const float* coef3, const float* a_y,
                                    const float* a_coef, int a_relu6, float* dw, float* slab, long npix, int Cout, int Cin,
                                    void* stream) {
  NOISE_REQUIRE(dz && a_y && dw && slab && (coef3 == nullptr || y_out != nullptr), "noise_gen_pw_wgrad: null pointer");
  NOISE_REQUIRE(npix > 0 && Cout >= 4 && Cout % 4 == 0 && Cin >= 4 && Cin % 4 == 0, "noise_gen_pw_wgrad: widths must be "
                "multiples of 4 (Cout %d, Cin %d)", Cout, Cin);
  const int rows = noise_gen_pw_wgrad_rows(npix, Cout, Cin);
  const WgradPlan pl = wgrad_plan(Cout, Cin);
  const dim3 grid(pl.nx, pl.ny, rows);
  NOISE_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "noise_gen_pw_wgrad: layer too wide");
  const LazyGrad G{dz, y_out, coef3};
  const LazySrc A{a_y, a_coef, a_relu6};
  hipStream_t st = (hipStream_t)stream;
#define WG_CASE(T)                                                                                                      \
  case T:                                                                                                               \
    if (pl.swap) hipLaunchKernelGGL((gen_pw_wgrad_kernel<T, true>), grid, dim3(256), 0, st, G, A, slab, npix, Cout, Cin);  \
    else hipLaunchKernelGGL((gen_pw_wgrad_kernel<T, false>), grid, dim3(256), 0, st, G, A, slab, npix, Cout, Cin);        \
    break;
  switch (pl.ct) { WG_CASE(1) WG_CASE(2) WG_CASE(4) WG_CASE(8) default: WG_CASE(16) }
#undef WG_CASE
  NOISE_CHECK_LAUNCH("noise_gen_pw_wgrad");
  return noise_gen_slab_reduce(slab, rows, (long)Cout * Cin, dw, stream);
}

NOISE_EXPORT int noise_gen_colsum(const float* x, int ld, long M, int N, float* out, float* slab, void* stream) {
  NOISE_REQUIRE(x && out && slab && M > 0 && N > 0 && ld >= N, "noise_gen_colsum: bad argument");
  const int rows = (int)(M < 256 ? M : 256);   // slab: 256 x N floats
  hipLaunchKernelGGL(gen_colsum_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, M, N, slab);
  NOISE_CHECK_LAUNCH("noise_gen_colsum");
  return noise_gen_slab_reduce(slab, rows, N, out, stream);
}

NOISE_EXPORT int noise_gen_relu_bwd(const float* g, const float* h, float* out, long n, void* stream) {
  NOISE_REQUIRE(g && h && out && n > 0 && n % 4 == 0, "noise_gen_relu_bwd: n must be a positive multiple of 4");
  hipLaunchKernelGGL(gen_relu_bwd_kernel, dim3(noise_grid(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g, h, out, n / 4);
  NOISE_CHECK_LAUNCH("noise_gen_relu_bwd");
  return 0;
}

NOISE_EXPORT int noise_gen_add_cropped(float* dst, const float* src, int B, int Hd, int Wd, int Hs, int Ws, int C, void* stream) {
  NOISE_REQUIRE(dst && src, "noise_gen_add_cropped: null pointer");
  NOISE_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd >= Hs && Wd >= Ws && C >= 4 && C % 4 == 0, "noise_gen_add_cropped: bad shape");
  hipLaunchKernelGGL(gen_add_cropped_kernel, dim3(noise_grid((long)B * Hs * Ws * (C / 4), 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, dst, src, B, Hd, Wd, Hs, Ws, C / 4);
  NOISE_CHECK_LAUNCH("noise_gen_add_cropped");
  return 0;
}

// slab: noise_gen_ln_bwd_rows(M) x 2D floats; dgamma, dbeta: D floats
NOISE_EXPORT int noise_gen_ln_bwd_rows(long M) { return noise_grid(M, 4 * 16, 1024); }

NOISE_EXPORT int noise_gen_ln_bwd(const float* g, const float* r, const float* gamma, float eps, const float* add, float* out,
                                  float* dgamma, float* dbeta, float* slab, long M, int D, void* stream) {
  NOISE_REQUIRE(g && r && gamma && out && dgamma && dbeta && slab, "noise_gen_ln_bwd: null pointer");
  NOISE_REQUIRE(M > 0 && D > 0 && D <= 64 * LNB_PER, "noise_gen_ln_bwd: 1 <= D <= %d (got %d)", 64 * LNB_PER, D);
  // a slab row is [sum g*xhat (D) | sum g (D)] and is folded in one go
  NOISE_REQUIRE(dbeta == dgamma + D, "noise_gen_ln_bwd: dbeta must follow dgamma in memory (dgamma + D)");
  const int rows = noise_gen_ln_bwd_rows(M);
  hipLaunchKernelGGL(gen_ln_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g, r, gamma, eps, add, out, slab, M, D);
  NOISE_CHECK_LAUNCH("noise_gen_ln_bwd");
  return noise_gen_slab_reduce(slab, rows, 2L * D, dgamma, stream);
}

// q, k, v, dq, dk, dv: [B*n][ld] with head h in columns [h*dk, (h+1)*dk); o, go: [B*n][ldo]; stats: B*H*n*3 floats
NOISE_EXPORT int noise_gen_attn_bwd(const float* q, const float* k, const float* v, int ld, const unsigned char* keep,
                                    float keep_scale, const float* o, const float* go, int ldo, float* dq, float* dk_, float* dv,
                                    float* stats, int B, int n, int H, int dk, float temperature, void* stream) {
  NOISE_REQUIRE(q && k && v && o && go && dq && dk_ && dv && stats, "noise_gen_attn_bwd: null pointer");
  NOISE_REQUIRE(B > 0 && n > 0 && H > 0 && dk > 0 && dk <= 64, "noise_gen_attn_bwd: head size must be in 1..64 (got %d)", dk);
  NOISE_REQUIRE(B <= 65535 && H <= 65535 && ld >= H * dk && ldo >= H * dk && temperature > 0.f, "noise_gen_attn_bwd: bad shape");
  hipStream_t st0 = (hipStream_t)stream;
  // heads of at most 32 run on the matrix cores (NOISE_TOK_ATT_VALU, diagnostic: the thread-per-query / per-key kernels)
  static const bool valu_only = getenv("NOISE_TOK_ATT_VALU") != nullptr;
  if (dk <= 32 && !valu_only) {
    const dim3 gm(noise_cdiv(n, 16 * ABM_T * 8), H, B);
    const float itm = 1.f / temperature;
#define ABM_CASE(K4, T)                                                                                                 \
  do {                                                                                                                  \
    hipLaunchKernelGGL((gen_attn_bwd_q_mfma_kernel<K4, T>), gm, dim3(512), 0, st0, q, k, v, ld, keep, keep_scale, o, go, ldo, \
                       dq, stats, n, H, dk, itm);                                                                       \
    hipLaunchKernelGGL((gen_attn_bwd_kv_mfma_kernel<K4, T>), gm, dim3(512), 0, st0, q, k, v, ld, keep, keep_scale, go, ldo,   \
                       stats, dk_, dv, n, H, dk, itm);                                                                  \
  } while (0)
    if (dk <= 8) ABM_CASE(2, 1);
    else if (dk <= 16) ABM_CASE(4, 1);
    else if (dk <= 24) ABM_CASE(6, 2);
    else ABM_CASE(8, 2);
#undef ABM_CASE
    NOISE_CHECK_LAUNCH("noise_gen_attn_bwd");
    return 0;
  }
  // short clips: workgroups of 64 or 128 queries / keys, so that few lanes are left without one
  const int nt = n <= 64 ? 64 : (n <= 128 ? 128 : 256);
  const dim3 grid(noise_cdiv(n, nt), H, B);
  hipStream_t st = (hipStream_t)stream;
  const float it = 1.f / temperature;
#define ATB_CASE(P, QB)                                                                                                   \
  do {                                                                                                                    \
    hipLaunchKernelGGL(gen_attn_bwd_q_kernel<P>, grid, dim3(nt), 0, st,  q, k, v, ld, keep, keep_scale, o, go, ldo, dq,    \
                       stats, n, H, dk, it);                                                                              \
    hipLaunchKernelGGL((gen_attn_bwd_kv_kernel<P, QB>), grid, dim3(nt), 0, st,  q, k, v, ld, keep, keep_scale, go, ldo,    \
                       stats, dk_, dv, n, H, dk, it);                                                                     \
  } while (0)
  if (dk <= 8) ATB_CASE(8, 128);
  else if (dk <= 16) ATB_CASE(16, 128);
  else if (dk <= 24) ATB_CASE(24, 128);
  else if (dk <= 32) ATB_CASE(32, 128);
  else ATB_CASE(64, 64);
#undef ATB_CASE
  NOISE_CHECK_LAUNCH("noise_gen_attn_bwd");
  return 0;
}

NOISE_EXPORT int noise_gen_vad_bwd(const float* x, float* gx, int ld, long tokens_per_clip, const float* w, const float* dvad,
                                   float* dw, float* db, int B, int D, void* stream) {
  NOISE_REQUIRE(x && gx && w && dvad && dw && db && B > 0 && D > 0 && ld >= D, "noise_gen_vad_bwd: bad argument");
  hipLaunchKernelGGL(gen_vad_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, gx, ld, tokens_per_clip, w, dvad, dw, db,
                     B, D);
  NOISE_CHECK_LAUNCH("noise_gen_vad_bwd");
  return 0;
}

// dout [B][2Hi-1][2Wi-1] (zero outside the cropped region), a [B][Hi][Wi][C] plain, w (C,1,3,3); g like a; dw (C,1,3,3);
// slab: noise_gen_last_bwd_rows() x 9C floats
NOISE_EXPORT int noise_gen_last_bwd_rows(void) { return 1024; }

NOISE_EXPORT int noise_gen_last_convT_bwd(const float* dout, const float* a, const float* w, float* g, float* dw, float* slab,
                                          int B, int Hi, int Wi, int C, void* stream) {
  NOISE_REQUIRE(dout && a && w && g && dw && slab, "noise_gen_last_convT_bwd: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0 && C >= 4 && C % 4 == 0 && C <= 256, "noise_gen_last_convT_bwd: bad shape (C a multiple "
                "of 4, at most 256)");
  const int cl = pow2_lanes(C);
  const int rows = noise_grid((long)B * Hi * Wi, 256 / cl * 4, 1024);
  hipLaunchKernelGGL(gen_last_bwd_kernel, dim3(rows), dim3(256), 256 * 36 * sizeof(float), (hipStream_t)stream, dout, a, w, g,
                     slab, B, Hi, Wi, C, cl);
  NOISE_CHECK_LAUNCH("noise_gen_last_convT_bwd");
  return noise_gen_slab_reduce(slab, rows, 9L * C, dw, stream);
}
