// Multi-resolution STFT loss of the waveform models and its gradient, gfx950.
//
// One resolution of train/models.py:687-720: mag = |torch.stft(x, n_fft, hop, win_length, hann(win_length))| clamped
// from below at sqrt(1e-7), loss = mean |log mag(clean) - log mag(enhanced)| over [B][T][F].  torch.stft as called
// there: frames centred on multiples of `hop` over the reflect-padded signal (n_fft/2 each side), T = 1 + L / hop,
// the window zero-padded to n_fft on both sides, one-sided spectrum (F = n_fft/2 + 1).  The waveform loss next to it
// (smooth-L1 with beta, models.py:657) is served by noise_smooth_l1.
//
// A workgroup takes FR frames of one clip, of BOTH signals (2 FR rows of one LDS buffer), with the two-pass register
// FFT of fft_common.h: a frame of N real samples is one complex N/2-point transform plus a split step.  The split
// step produces |X|, |Y| of a bin pair (k, N/2 - k), adds |log|Y| - log|X|| to the loss partial and overwrites the
// enhanced row with the spectrum of the gradient, G_k = dL/d|X_k| * X_k / |X_k|, arranged so that one more forward
// transform of the same rows yields the per-frame gradient sum_k Re(G_k e^{+2 pi i k n / N}) (inverse real transform
// through conjugation).  The frame gradients (window applied, only the `win_length` samples where the window is not
// zero) go to a workspace [B][T][win_length]; a second kernel gathers, for every sample, the frames that cover it and
// folds the reflect padding back, in a fixed order: no atomics, bit-reproducible.
#include "common.h"
#include "fft_common.h"

namespace {

constexpr float MR_CLAMP = 3.16227766016837933e-4f;   // sqrt(1e-7), models.py:698

struct MrDims {
  int B, L, T, hop, wl, off;   // off = (n_fft - win_length) / 2: first sample of a frame the window covers
  float coef;                  // gradient multiplier / (B T F)
  int want_grad;
};

template <int N2, int R1, int R2, int FR>
struct MrCfg {
  static constexpr int TPF = R1 > R2 ? R1 : R2;   // threads per row: pass A uses R1 of them, pass B R2
  static constexpr int ROWS = 2 * FR;             // enhanced frames, then clean frames
  static constexpr int NT = ROWS * TPF;
  static constexpr int ZP = N2 + 1;
  static size_t smem(int hop) {
    (void)hop;   // the staged samples live inside the complex buffer
    return (size_t)ROWS * ZP * 8 + (size_t)(N2 + N2 + 1) * 8 + (size_t)(2 * N2) * 4;
  }
};

// One-sided bins k and N2 - k of a real N-point transform from the complex N2-point transform of the packed frame.
__device__ __forceinline__ void split_pair(const c32* zr, int k, int N2, c32 w, c32& xa, c32& xb) {
  const c32 zk = zr[k], zn = zr[k == 0 ? 0 : N2 - k];
  if (k == 0) {          // DC and Nyquist: both real
    xa = {zk.x + zk.y, 0.f};
    xb = {zk.x - zk.y, 0.f};
    return;
  }
  const c32 e = {0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
  const c32 o = {0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
  const c32 wo = cmul(w, o);
  xa = cadd(e, wo);
  xb = cconj(csub(e, wo));
}

// |log max(|Y|, c) - log max(|X|, c)| and the gradient bin a * X / |X| with a = dL/d|X| (zero on the clamp).
__device__ __forceinline__ float bin_term(c32 x, c32 y, float coef, c32& g) {
  const float ax = sqrtf(x.x * x.x + x.y * x.y), ay = sqrtf(y.x * y.x + y.y * y.y);
  const float mx = fmaxf(ax, MR_CLAMP), my = fmaxf(ay, MR_CLAMP);
  const float diff = logf(my) - logf(mx);
  float a = 0.f;
  if (ax > MR_CLAMP && diff != 0.f) a = (diff > 0.f ? -coef : coef) / (mx * ax);
  g = {a * x.x, a * x.y};
  return fabsf(diff);
}

template <int N2, int R1, int R2, int FR>
__global__ __launch_bounds__((MrCfg<N2, R1, R2, FR>::NT)) void mrstft_frames_kernel(
    const float* __restrict__ xg, const float* __restrict__ yg, const float* __restrict__ win, const c32* __restrict__ tw1g,
    const c32* __restrict__ tw2g, float* __restrict__ frame_grad, double* __restrict__ partial, MrDims d) {
  using Cfg = MrCfg<N2, R1, R2, FR>;
  constexpr int N = 2 * N2, NT = Cfg::NT, ZP = Cfg::ZP, TPF = Cfg::TPF, ROWS = Cfg::ROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  c32* zbuf = reinterpret_cast<c32*>(lds_raw);              // [ROWS][ZP]
  c32* tw1 = zbuf + ROWS * ZP;                              // [N2]      e^{-2 pi i j / N2}
  c32* tw2 = tw1 + N2;                                      // [N2 + 1]  e^{-2 pi i k / N}
  float* wn = reinterpret_cast<float*>(tw2 + N2 + 1);       // [N]
  // the staged samples [2][(FR-1) hop + N] share the complex buffer: pass A reads its samples into registers, a barrier,
  // and only then writes over them (one buffer less is a second resident workgroup at n_fft 512 and 1024)
  float* xs = reinterpret_cast<float*>(lds_raw);
  static_assert((size_t)2 * FR * N * 4 <= (size_t)ROWS * ZP * 8, "the staged samples must fit in the complex buffer");
  __shared__ double red[NT / 64];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * FR;
  const int hop = d.hop;
  const int nS = (FR - 1) * hop + N;
  // ---- prologue: the tables and the first samples are requested before the first LDS store, the samples 8 per
  // thread and round (one memory round trip per round instead of one per element)
  {
    constexpr int TWN = (N2 + NT) / NT, WNN = (N + NT - 1) / NT;
    c32 t1r[TWN], t2r[TWN];
    float wr[WNN];
#pragma unroll
    for (int u = 0; u < TWN; ++u) {
      const int k = tid + u * NT;
      t1r[u] = tw1g[k < N2 ? k : N2 - 1];
      t2r[u] = tw2g[k <= N2 ? k : N2];
    }
#pragma unroll
    for (int u = 0; u < WNN; ++u) wr[u] = win[min(tid + u * NT, N - 1)];
    bool first = true;
    for (int base = 0; base < 2 * nS; base += 8 * NT) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(base + tid + u * NT, 2 * nS - 1);
        const int which = i >= nS, ii = which ? i - nS : i;
        int sidx = t0 * hop + ii - N2;            // sample index of padded position t0*hop + ii
        if (sidx < 0) sidx = -sidx;
        if (sidx >= d.L) sidx = 2 * (d.L - 1) - sidx;
        sidx = min(max(sidx, 0), d.L - 1);        // only reached by frames past T (never used)
        xv[u] = (which ? yg : xg)[(size_t)b * d.L + sidx];
      }
      if (first) {
        first = false;
#pragma unroll
        for (int u = 0; u < TWN; ++u) {
          const int k = tid + u * NT;
          if (k < N2) tw1[k] = t1r[u];
          if (k <= N2) tw2[k] = t2r[u];
        }
#pragma unroll
        for (int u = 0; u < WNN; ++u)
          if (tid + u * NT < N) wn[tid + u * NT] = wr[u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + tid + u * NT;
        if (i < 2 * nS) xs[i] = xv[u];
      }
    }
  }
  __syncthreads();
  const int row = tid / TPF, j = tid % TPF;
  c32* zrow = zbuf + row * ZP;
  // ---- forward pass A (both signals)
  c32 v[R2];
  if (j < R1) {
    const float* x = xs + (row >= FR ? nS : 0) + (row % FR) * hop;
#pragma unroll
    for (int m2 = 0; m2 < R2; ++m2) {
      const int n = 2 * (j + R1 * m2);
      v[m2] = {x[n] * wn[n], x[n + 1] * wn[n + 1]};
    }
  }
  __syncthreads();   // every frame has its samples: the complex buffer may overwrite them
  if (j < R1) {
    Dft<R2>::run(v);
    zrow[j * R2] = v[0];
#pragma unroll
    for (int k2 = 1; k2 < R2; ++k2) zrow[j * R2 + k2] = cmul(v[k2], tw1[j * k2]);
  }
  __syncthreads();
  fft_pass_b<R1, R2>(zrow, j);
  __syncthreads();
  // ---- split step, loss terms, gradient spectrum (thread = frame, bin group)
  float lsum = 0.f;
  {
    const int f2 = tid % FR, kg = tid / FR;
    constexpr int KG = NT / FR;
    const bool tin = t0 + f2 < d.T;
    c32* zx = zbuf + f2 * ZP;
    const c32* zy = zbuf + (FR + f2) * ZP;
    for (int k = kg; k <= N2 / 2; k += KG) {
      const int kb = N2 - k;
      c32 xa, xb, ya, yb, ga, gb;
      split_pair(zx, k, N2, tw2[k], xa, xb);
      split_pair(zy, k, N2, tw2[k], ya, yb);
      float term = bin_term(xa, ya, d.coef, ga);
      if (kb != k) term += bin_term(xb, yb, d.coef, gb);
      else gb = ga;
      if (tin) lsum += term;
      else ga = gb = {0.f, 0.f};
      // conj(Z) of the half-spectrum H with H_0 = 2 Re G_0, H_{N2} = 2 Re G_{N2}, H_k = G_k: the forward transform of
      // these rows is N2 conj(z) with z[m] = r[2m] + i r[2m+1] and (N/2) r[n] = sum_k Re(G_k e^{+2 pi i k n / N})
      if (k == 0) {
        zx[0] = {ga.x + gb.x, -(ga.x - gb.x)};
      } else {
        const c32 e = {0.5f * (ga.x + gb.x), 0.5f * (ga.y - gb.y)};
        const c32 df = {0.5f * (ga.x - gb.x), 0.5f * (ga.y + gb.y)};
        const c32 o = cmul(df, cconj(tw2[k]));
        zx[k] = {e.x - o.y, -(e.y + o.x)};
        zx[kb] = {e.x + o.y, -(-e.y + o.x)};
      }
    }
  }
  {
    double ls = (double)wave_reduce_sum(lsum);
    if ((tid & 63) == 0) red[tid >> 6] = ls;
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
  if (!d.want_grad) return;
  // ---- transform of the gradient rows (enhanced rows only); pass A reads its inputs before any row is rewritten
  {
    c32 v[R2];
    const bool act = row < FR && j < R1;
    if (act) {
#pragma unroll
      for (int m2 = 0; m2 < R2; ++m2) v[m2] = zrow[j + R1 * m2];
    }
    __syncthreads();
    if (act) {
      Dft<R2>::run(v);
      zrow[j * R2] = v[0];
#pragma unroll
      for (int k2 = 1; k2 < R2; ++k2) zrow[j * R2 + k2] = cmul(v[k2], tw1[j * k2]);
    }
    __syncthreads();
    if (row < FR) fft_pass_b<R1, R2>(zrow, j);
    __syncthreads();
  }
  // frame gradient at sample n: w[n] * (even n: Re, odd n: -Im of element n/2)
  const float* fl = reinterpret_cast<const float*>(zbuf);
  const int wl = d.wl;
  for (int i = tid; i < FR * wl; i += NT) {
    const int f = i / wl, jj = i - f * wl;
    const int t = t0 + f;
    if (t >= d.T) break;
    const int n = d.off + jj;
    const float v = fl[f * (2 * ZP) + n];
    frame_grad[((size_t)b * d.T + t) * wl + jj] = ((n & 1) ? -v : v) * wn[n];
  }
}

// dx[b][s] (+)= sum over the padded positions that mirror sample s and the frames that cover each of them.
__global__ __launch_bounds__(256) void mrstft_gather_kernel(const float* __restrict__ frame_grad, float* __restrict__ dx,
                                                            MrDims d, int pad, int accumulate) {
  const long total = (long)d.B * d.L;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / d.L), s = (int)(i - (long)b * d.L);
    const float* fg = frame_grad + (size_t)b * d.T * d.wl;
    int pos[3];
    int np = 0;
    pos[np++] = s + pad;
    if (s >= 1 && s <= pad) pos[np++] = pad - s;
    if (s >= d.L - 1 - pad && s <= d.L - 2) pos[np++] = pad + 2 * (d.L - 1) - s;
    float acc = 0.f;
    for (int q = 0; q < np; ++q) {
      const int r = pos[q] - d.off;                    // position relative to the window's first sample of frame 0
      if (r < 0) continue;
      int thi = r / d.hop;
      if (thi > d.T - 1) thi = d.T - 1;
      int tlo = r - d.wl + 1 <= 0 ? 0 : (r - d.wl + d.hop) / d.hop;     // ceil((r - wl + 1) / hop)
      for (int t = tlo; t <= thi; ++t) acc += fg[(size_t)t * d.wl + (r - t * d.hop)];
    }
    dx[i] = (accumulate ? dx[i] : 0.f) + acc;
  }
}

// out[0] (+)= mul * sum of n partials, summed in a fixed order by one wave
__global__ __launch_bounds__(64) void sum_partials_kernel(const double* __restrict__ partial, int n, double mul,
                                                          double* __restrict__ out, int accumulate) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
  s = wave_reduce_sum(s);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0) + mul * s;
}

constexpr int SL1_BLOCKS = 1024;

__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ x, const float* __restrict__ y, long n,
                                                        float beta, float gscale, float* __restrict__ dx, int accumulate,
                                                        double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  const float inv_beta = 1.f / beta;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float dd = x[i] - y[i], ad = fabsf(dd);
    const bool quad = ad < beta;
    s += (double)(quad ? 0.5f * dd * dd * inv_beta : ad - 0.5f * beta);
    if (dx != nullptr) {
      const float g = quad ? dd * inv_beta : (dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f));
      dx[i] = (accumulate ? dx[i] : 0.f) + gscale * g;
    }
  }
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int N2, int R1, int R2, int FR>
int launch_frames(const float* x, const float* y, const float* win, const float* tw1, const float* tw2, float* frame_grad,
                  double* partial, MrDims d, hipStream_t st) {
  using Cfg = MrCfg<N2, R1, R2, FR>;
  const size_t smem = Cfg::smem(d.hop);
  NOISE_REQUIRE(smem <= 158 * 1024, "noise_mrstft_term: hop=%d needs %zu bytes of LDS", d.hop, smem);
  auto kfn = mrstft_frames_kernel<N2, R1, R2, FR>;
  NOISE_ALLOW_LDS(kfn, smem);
  hipLaunchKernelGGL(kfn, dim3(noise_cdiv(d.T, FR), d.B), dim3(Cfg::NT), smem, st, x, y, win, (const c32*)tw1, (const c32*)tw2,
                     frame_grad, partial, d);
  return 0;
}

int frames_per_group(int n_fft) { return n_fft == 512 ? 16 : (n_fft == 1024 ? 8 : 4); }

}  // namespace

NOISE_EXPORT int noise_mrstft_supported(int n_fft) { return n_fft == 512 || n_fft == 1024 || n_fft == 2048; }

// Number of doubles of `partial` one call needs.
NOISE_EXPORT int noise_mrstft_partials(int B, int L, int n_fft, int hop) {
  if (!noise_mrstft_supported(n_fft) || hop <= 0) return 0;
  return B * noise_cdiv(L / hop + 1, frames_per_group(n_fft));
}

NOISE_EXPORT int noise_mrstft_term(const float* enhanced, const float* clean, const float* window, const float* tw1,
                                   const float* tw2, float grad_scale, float* frame_grad, double* partial, double* loss,
                                   int accumulate_loss, float* d_enhanced, int accumulate_grad, int B, int L, int n_fft,
                                   int hop, int win_length, void* stream) {
  NOISE_REQUIRE(enhanced && clean && window && tw1 && tw2 && partial && loss, "noise_mrstft_term: null pointer");
  NOISE_REQUIRE(noise_mrstft_supported(n_fft), "noise_mrstft_term: n_fft=%d is not served (512, 1024, 2048)", n_fft);
  NOISE_REQUIRE(B > 0 && B <= 65535, "noise_mrstft_term: B=%d outside 1..65535", B);
  NOISE_REQUIRE(hop > 0 && win_length > 0 && win_length <= n_fft, "noise_mrstft_term: hop=%d win_length=%d n_fft=%d", hop,
                win_length, n_fft);
  NOISE_REQUIRE(L > n_fft / 2, "noise_mrstft_term: L=%d too short for reflect padding of %d", L, n_fft / 2);
  NOISE_REQUIRE((d_enhanced == nullptr) == (frame_grad == nullptr),
                "noise_mrstft_term: the gradient needs both d_enhanced and the frame_grad workspace");
  hipStream_t st = (hipStream_t)stream;
  const int T = L / hop + 1, F = n_fft / 2 + 1;
  const double count = (double)B * T * F;
  MrDims d{B, L, T, hop, win_length, (n_fft - win_length) / 2, (float)((double)grad_scale / count), d_enhanced != nullptr};
  int rc = 0;
  if (n_fft == 512) rc = launch_frames<256, 16, 16, 16>(enhanced, clean, window, tw1, tw2, frame_grad, partial, d, st);
  else if (n_fft == 1024) rc = launch_frames<512, 16, 32, 8>(enhanced, clean, window, tw1, tw2, frame_grad, partial, d, st);
  else rc = launch_frames<1024, 32, 32, 4>(enhanced, clean, window, tw1, tw2, frame_grad, partial, d, st);
  if (rc) return rc;
  NOISE_CHECK_LAUNCH("noise_mrstft_term (frames)");
  const int np = noise_mrstft_partials(B, L, n_fft, hop);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, partial, np, 1.0 / count, loss, accumulate_loss);
  if (d_enhanced != nullptr)
    hipLaunchKernelGGL(mrstft_gather_kernel, dim3(noise_grid((long)B * L, 256, 8192)), dim3(256), 0, st, frame_grad, d_enhanced, d,
                       n_fft / 2, accumulate_grad);
  NOISE_CHECK_LAUNCH("noise_mrstft_term");
  return 0;
}

NOISE_EXPORT int noise_smooth_l1_blocks(void) { return SL1_BLOCKS; }

// loss[0] (+)= mean smooth_l1(x - y; beta); d_x (+)= grad_scale / n * d smooth_l1 (d_x may be null).
NOISE_EXPORT int noise_smooth_l1(const float* x, const float* y, long n, float beta, float grad_scale, double* partial,
                                 double* loss, int accumulate_loss, float* d_x, int accumulate_grad, void* stream) {
  NOISE_REQUIRE(x && y && partial && loss, "noise_smooth_l1: null pointer");
  NOISE_REQUIRE(n > 0 && beta > 0.f, "noise_smooth_l1: n=%ld beta=%g", n, (double)beta);
  hipStream_t st = (hipStream_t)stream;
  const int grid = noise_grid(n, 256, SL1_BLOCKS);
  hipLaunchKernelGGL(smooth_l1_kernel, dim3(grid), dim3(256), 0, st, x, y, n, beta, (float)((double)grad_scale / (double)n), d_x,
                     accumulate_grad, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, partial, grid, 1.0 / (double)n, loss, accumulate_loss);
  NOISE_CHECK_LAUNCH("noise_smooth_l1");
  return 0;
}
