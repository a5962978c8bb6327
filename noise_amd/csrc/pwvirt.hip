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
#include "lazy.h"

namespace {

struct VirtDims {
  long npix;            // B * Ho * Wo output pixels
  int K, N;             // hidden channels (multiple of 8), output channels (multiple of 4)
  int Hi, Wi, Ho, Wo;   // source and output image sizes, Ho = 2 Hi - 1, Wo = 2 Wi - 1
};

// tap index of the stand-alone kernel for each (parity, source) pair; -1: the source does not contribute
__device__ __forceinline__ int virt_tap(int py, int px, int src) {
  // src 0 = (m, x), 1 = (m, x+1), 2 = (m+1, x), 3 = (m+1, x+1)
  if (src == 0) return py ? (px ? 8 : 7) : (px ? 5 : 4);
  if (src == 1) return px ? (py ? 6 : 3) : -1;
  if (src == 2) return py ? (px ? 2 : 1) : -1;
  return (px && py) ? 0 : -1;
}

template <int KC, int MT>
__global__ __launch_bounds__(256) void pw_virt_fwd_kernel(LazySrc In, const float* __restrict__ wdw, LazySrc Dw,
                                                          const float* __restrict__ W, float* __restrict__ out,
                                                          float* __restrict__ slab, VirtDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, q = lane >> 4;
  const int K = d.K, N = d.N;
  constexpr int KP = KC * 16;
  float* wfrag = lds;                    // [MT][KC][64][4]
  float* cf = wfrag + MT * KC * 256;     // [4][KP]: scale_in, shift_in, scale_dw, shift_dw
  float* wt = cf + 4 * KP;               // [10][KP]: the 9 taps (tap-major), row 9 = zeros (absent source)
  if ((N & 15) || (K & 15)) {
    for (int idx = tid; idx < MT * KC * 256; idx += 256) wfrag[idx] = 0.f;
    __syncthreads();
  }
  {
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
    cf[k] = (in && In.coef) ? In.coef[k] : 1.f;
    cf[KP + k] = (in && In.coef) ? In.coef[K + k] : 0.f;
    cf[2 * KP + k] = (in && Dw.coef) ? Dw.coef[k] : 1.f;
    cf[3 * KP + k] = (in && Dw.coef) ? Dw.coef[K + k] : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t * KP + k] = in ? wdw[(size_t)k * 9 + t] : 0.f;    // depthwise weight (C, 1, 3, 3)
    wt[9 * KP + k] = 0.f;
  }
  __syncthreads();

  float4 s1[MT], s2[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) s1[m] = s2[m] = f4(0.f);
  const float ilo = In.relu6 ? 0.f : -INFINITY, ihi = In.relu6 ? 6.f : INFINITY;
  const float dlo = Dw.relu6 ? 0.f : -INFINITY, dhi = Dw.relu6 ? 6.f : INFINITY;
  const long ntiles = (d.npix + 15) >> 4;
  const unsigned per_img = (unsigned)d.Ho * (unsigned)d.Wo;

  // raw source cells of the NEXT tile are requested before the MFMAs of the current one (clamped, always valid
  // addresses; an absent source repeats cell (m, x) and meets a zero tap)
  float4 r[KC][4];
  int npy = 0, npx = 0;
  auto issue = [&](long tile) {
    long pix = tile * 16 + p;
    pix = pix < d.npix ? pix : d.npix - 1;
    const unsigned b = (unsigned)pix / per_img, rem = (unsigned)pix - b * per_img;
    const unsigned oy = rem / (unsigned)d.Wo, ox = rem - oy * (unsigned)d.Wo;
    npy = oy & 1;
    npx = ox & 1;
    const size_t cell = (((size_t)b * d.Hi + (oy >> 1)) * d.Wi + (ox >> 1)) * K;
    const size_t o1 = npx ? (size_t)K : 0, o2 = npy ? (size_t)d.Wi * K : 0;
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
      const int ci = 16 * jj + 4 * q;
      const float* src = In.y + cell + (ci < K ? ci : 0);
      r[jj][0] = ld4(src);
      r[jj][1] = ld4(src + o1);
      r[jj][2] = ld4(src + o2);
      r[jj][3] = ld4(src + o1 + o2);
    }
  };
  const long tile0 = (long)blockIdx.x * 4 + wave, tstep = (long)gridDim.x * 4;
  if (tile0 < ntiles) issue(tile0);
  for (long tile = tile0; tile < ntiles; tile += tstep) {
    const long pix = tile * 16 + p;
    const bool valid = pix < d.npix;
    const int py = npy, px = npx;
    int tap[4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const int t = virt_tap(py, px, sidx);
      tap[sidx] = (t < 0 ? 9 : t) * KP;
    }
    float4 in[KC];
#pragma unroll
    for (int jj = 0; jj < KC; ++jj) {
      const int ci = 16 * jj + 4 * q;
      const float4 sc = ld4(cf + ci), sh = ld4(cf + KP + ci);
      float4 y = clamp4(fma4(r[jj][0], sc, sh), ilo, ihi) * ld4(wt + tap[0] + ci);
      y = fma4(clamp4(fma4(r[jj][1], sc, sh), ilo, ihi), ld4(wt + tap[1] + ci), y);
      y = fma4(clamp4(fma4(r[jj][2], sc, sh), ilo, ihi), ld4(wt + tap[2] + ci), y);
      y = fma4(clamp4(fma4(r[jj][3], sc, sh), ilo, ihi), ld4(wt + tap[3] + ci), y);
      const float4 v = clamp4(fma4(y, ld4(cf + 2 * KP + ci), ld4(cf + 3 * KP + ci)), dlo, dhi);
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
      const float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (valid && co < N) {
        st4(out + (size_t)pix * N + co, o);
        s1[m] = s1[m] + o;
        s2[m] = fma4(o, o, s2[m]);
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
      float s = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) s += red[(wv * 2 + j) * MT * 16 + c];
      slab[(size_t)blockIdx.x * 2 * N + i] = s;
    }
  }
}

template <int KC, int MT>
int launch_virt(LazySrc In, const float* wdw, LazySrc Dw, const float* W, float* out, float* slab, VirtDims d, int& grid,
                hipStream_t st) {
  const size_t smem = (size_t)(MT * KC * 256 + 14 * KC * 16) * sizeof(float);
  auto kfn = pw_virt_fwd_kernel<KC, MT>;
  NOISE_ALLOW_LDS(kfn, smem);
  grid = noise_whole_rounds(grid, kfn, 256, smem);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), smem, st, In, wdw, Dw, W, out, slab, d);
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
  VirtDims d{npix, K, N, Hi, Wi, Ho, Wo};
  LazySrc In{in_y, in_coef, in_relu6}, Dw{nullptr, dw_coef, dw_relu6};
  int grid = noise_grid(npix, 16 * 4 * 4, 1024);
  const int KC = (K + 15) / 16, MT = (N + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  if (KC == 5 && MT == 1) launch_virt<5, 1>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  else if (KC == 6 && MT == 2) launch_virt<6, 2>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  else launch_virt<12, 2>(In, wdw, Dw, w, out, stats_slab, d, grid, st);
  if (slab_rows) *slab_rows = grid;
  NOISE_CHECK_LAUNCH("noise_pw_fwd_virt");
  return 0;
}
