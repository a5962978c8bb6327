// On-GPU waveform augmentation of the VAD training data module, gfx950.
//
// The chain train/data.py:439-461, parameters drawn once per pair and frozen for the second pass:
//   input : Gain -> Shift -> AddBackgroundNoise -> ImpulseResponse (p = 0.5) -> Cut -> PolarityInversion -> Clip
//   target: VADNoiseDataset (train/data.py:541-562): both passes start from the CLEAN clip and the frozen pass keeps
//           Gain, Shift, Cut and Clip -> target = Clip(Cut(Shift(Gain(clean)))), same gain as the input;
//           NoiseDataset (train/data.py:410-427): separate noisy / clean files, the frozen pass keeps Shift and Cut only.
// ImpulseResponse.apply (train/data.py:186-209) and process/process.py:69-75 (add_pyreverb) are a full FFT
// convolution truncated to the input length; the former then matches the RMS of its input and clips.
//
// Kernels:
//   aug_stats_kernel    sums of squares of the gained, shifted clip and of the noise segment      (per clip and chunk)
//   aug_mix_kernel      y = s + noise * factor; clips without an impulse response leave through Cut / polarity / Clip
//                       here, the others write y whole (the convolution needs every sample) with its sum of squares
//   fir_spectra_kernel  spectra of 2048-sample segments at hop 1024 (signal: [previous block | block], filter:
//                       [partition | zeros]) with the two-pass register FFT; real data as one 1024-point complex transform
//   fir_apply_kernel    uniformly partitioned overlap-save: Y_k = sum_p X_{k-p} H_p, inverse transform, last 1024 samples;
//                       a thread owns a bin pair for 8 consecutive output blocks and slides a register window over X, so
//                       every spectrum block is read once per workgroup
//   aug_finish_kernel   RMS match, clip, Cut, polarity, Clip of the convolved clips
// All sums are per-workgroup partials in double, added in a fixed order: results are reproducible run to run.
#include "common.h"
#include "fft_common.h"
#include "noise_hip.h"

namespace {

constexpr int FB = 1024;        // new samples per block = filter partition length
constexpr int FN2 = 1024;       // complex transform length of a 2048-sample real segment
constexpr int FROWS = 8;        // segments / output blocks per workgroup
constexpr int FZP = FN2 + 1;    // LDS row pitch (complex elements)
constexpr int FNT = FROWS * 32;
constexpr int AUG_CH = 8192;    // samples per workgroup of the elementwise kernels

// LDS holds the row buffer only: 8 rows x 1025 complex = 65.6 KB, two workgroups per CU.  The two twiddle tables (16 KB)
// are read from global memory; they stay cache resident.
constexpr size_t fir_smem() { return (size_t)FROWS * FZP * 8; }

// Forward 1024-point transform of the rows of zbuf whose pass-A inputs the caller holds in v (thread = (row, j)).
__device__ __forceinline__ void fir_fft_rows(c32 (&v)[32], c32* zrow, const c32* __restrict__ tw1, int j) {
  Dft<32>::run(v);
  zrow[j * 32] = v[0];
#pragma unroll
  for (int k2 = 1; k2 < 32; ++k2) zrow[j * 32 + k2] = cmul(v[k2], tw1[j * k2]);
  __syncthreads();
  fft_pass_b<32, 32>(zrow, j);
  __syncthreads();
}

struct SpecDims {
  long L, x_stride;
  int nblk, filter;
};

// spec[row][k][f], f < 1024: one-sided spectrum of segment k, bin 0 holding (DC, Nyquist).
__global__ __launch_bounds__(FNT, 2) void fir_spectra_kernel(const float* __restrict__ xg, const c32* __restrict__ tw1g,
                                                          const c32* __restrict__ tw2g, c32* __restrict__ spec, SpecDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  c32* zbuf = reinterpret_cast<c32*>(lds_raw);
  const c32* tw1 = tw1g;
  const c32* tw2 = tw2g;
  const int tid = threadIdx.x, row = tid >> 5, j = tid & 31;
  const int k0 = blockIdx.x * FROWS, k = k0 + row;
  const float* x = xg + (size_t)blockIdx.y * d.x_stride;
  // first sample of the segment and the number of samples of it that come from the signal
  const long first = d.filter ? (long)k * FB : (long)(k - 1) * FB;
  const int span = d.filter ? FB : 2 * FB;
  c32 v[32];
#pragma unroll
  for (int m2 = 0; m2 < 32; ++m2) {
    const int n = 2 * (j + 32 * m2);
    const long g0 = first + n, g1 = g0 + 1;
    const bool ok0 = n < span && g0 >= 0 && g0 < d.L, ok1 = n + 1 < span && g1 >= 0 && g1 < d.L;
    const float a = x[min(max(g0, 0L), d.L - 1)], b = x[min(max(g1, 0L), d.L - 1)];
    v[m2] = {ok0 ? a : 0.f, ok1 ? b : 0.f};
  }
  fir_fft_rows(v, zbuf + row * FZP, tw1, j);
  for (int r = 0; r < FROWS; ++r) {
    if (k0 + r >= d.nblk) break;
    const c32* zr = zbuf + r * FZP;
    c32* out = spec + ((size_t)blockIdx.y * d.nblk + (k0 + r)) * FN2;
    for (int kk = tid; kk <= FN2 / 2; kk += FNT) {
      c32 xa, xb;
      {
        const c32 zk = zr[kk], zn = zr[kk == 0 ? 0 : FN2 - kk];
        if (kk == 0) {
          xa = {zk.x + zk.y, zk.x - zk.y};   // (DC, Nyquist), both real
          xb = xa;
        } else {
          const c32 e = {0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
          const c32 o = {0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
          const c32 wo = cmul(tw2[kk], o);
          xa = cadd(e, wo);
          xb = cconj(csub(e, wo));
        }
      }
      out[kk] = xa;
      if (kk != 0 && kk != FN2 / 2) out[FN2 - kk] = xb;
    }
  }
}

// Product of two packed one-sided bins: bin 0 carries two real values that multiply separately.
__device__ __forceinline__ c32 bin_mul(c32 x, c32 h, bool packed_reals) {
  const c32 m = cmul(x, h);
  return packed_reals ? c32{x.x * h.x, x.y * h.y} : m;
}

struct ApplyDims {
  long L, y_stride;
  int nblk, n_ir, n_rows, ntiles;
};

__global__ __launch_bounds__(FNT, 2) void fir_apply_kernel(const c32* __restrict__ xspec, const int* __restrict__ ir_of_row,
                                                        const c32* __restrict__ hspec, const long* __restrict__ ir_first_block,
                                                        const int* __restrict__ ir_nblk, const c32* __restrict__ tw1g,
                                                        const c32* __restrict__ tw2g, float* __restrict__ yg,
                                                        double* __restrict__ partial, ApplyDims d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  c32* zbuf = reinterpret_cast<c32*>(lds_raw);
  const c32* tw1 = tw1g;
  const c32* tw2 = tw2g;
  __shared__ double red[FNT / 64];
  const int tid = threadIdx.x;
  // The workgroups of a clip read overlapping windows of its spectra (every block 2-3 times with a 1 s filter):
  // consecutive virtual indices run on one XCD, so those re-reads meet in one L2 instead of going to HBM from eight.
  const int vblk = xcd_chunked_block();
  if (vblk >= d.n_rows * d.ntiles) return;
  const int b = vblk / d.ntiles, tile = vblk - b * d.ntiles, k0 = tile * FROWS;
  const int ir = min(max(ir_of_row[b], 0), d.n_ir - 1);
  const int P = ir_nblk[ir];
  const c32* H = hspec + (size_t)ir_first_block[ir] * FN2;
  const c32* X = xspec + (size_t)b * d.nblk * FN2;
  // 513 bin pairs (k, 1024 - k) over 256 threads in two rounds: the thread that takes the packed bin 0 (DC and Nyquist,
  // two real products) carries the self-paired bin 512 in its second accumulator set instead of a copy of bin 0
  for (int k = tid; k < FN2 / 2; k += FNT) {
    const int kb = k == 0 ? FN2 / 2 : FN2 - k;
    const bool packed = k == 0;
    c32 ya[FROWS], yb[FROWS], xa[FROWS], xb[FROWS];
#pragma unroll
    for (int r = 0; r < FROWS; ++r) {
      ya[r] = yb[r] = {0.f, 0.f};
      const int idx = k0 + r;
      const bool ok = idx < d.nblk;
      const size_t o = (size_t)min(idx, d.nblk - 1) * FN2;
      const c32 a = X[o + k], c = X[o + kb];
      xa[r] = ok ? a : c32{0.f, 0.f};
      xb[r] = ok ? c : c32{0.f, 0.f};
    }
    for (int p8 = 0; p8 < P; p8 += FROWS) {
      // the 8 filter partitions of this round and the 8 blocks that enter the window during it, requested together
      c32 ha[FROWS], hb[FROWS], na[FROWS], nb[FROWS];
#pragma unroll
      for (int u = 0; u < FROWS; ++u) {
        const int p = p8 + u;
        const bool okp = p < P;
        const size_t ho = (size_t)min(p, P - 1) * FN2;
        const c32 h0 = H[ho + k], h1 = H[ho + kb];
        ha[u] = okp ? h0 : c32{0.f, 0.f};
        hb[u] = okp ? h1 : c32{0.f, 0.f};
        const int idx = k0 - p - 1;
        const bool okx = idx >= 0 && idx < d.nblk;
        const size_t xo = (size_t)min(max(idx, 0), d.nblk - 1) * FN2;
        const c32 x0 = X[xo + k], x1 = X[xo + kb];
        na[u] = okx ? x0 : c32{0.f, 0.f};
        nb[u] = okx ? x1 : c32{0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < FROWS; ++u) {
        // partition p = p8 + u: output block k0 + r takes block k0 + r - p, which sits in slot (r - u) mod 8
#pragma unroll
        for (int r = 0; r < FROWS; ++r) {
          const int s = (r - u) & (FROWS - 1);
          ya[r] = cadd(ya[r], bin_mul(xa[s], ha[u], packed));
          yb[r] = cadd(yb[r], cmul(xb[s], hb[u]));
        }
        xa[(FROWS - 1 - u) & (FROWS - 1)] = na[u];
        xb[(FROWS - 1 - u) & (FROWS - 1)] = nb[u];
      }
    }
    // conj(Z) of the packed real sequence whose one-sided spectrum is Y: a forward transform of these rows gives
    // 1024 conj(z), z[m] = y[2m] + i y[2m+1]
    const c32 w = tw2[k];
#pragma unroll
    for (int r = 0; r < FROWS; ++r) {
      c32* zr = zbuf + r * FZP;
      if (packed) {
        zr[0] = {0.5f * (ya[r].x + ya[r].y), -0.5f * (ya[r].x - ya[r].y)};
        zr[FN2 / 2] = yb[r];   // Z[512] = conj(Y[512]), stored conjugated
      } else {
        const c32 e = {0.5f * (ya[r].x + yb[r].x), 0.5f * (ya[r].y - yb[r].y)};
        const c32 df = {0.5f * (ya[r].x - yb[r].x), 0.5f * (ya[r].y + yb[r].y)};
        const c32 o = cmul(df, cconj(w));
        zr[k] = {e.x - o.y, -(e.y + o.x)};
        zr[kb] = {e.x + o.y, e.y - o.x};
      }
    }
  }
  __syncthreads();
  {
    const int row = tid >> 5, j = tid & 31;
    c32* zrow = zbuf + row * FZP;
    c32 v[32];
#pragma unroll
    for (int m2 = 0; m2 < 32; ++m2) v[m2] = zrow[j + 32 * m2];
    __syncthreads();
    fir_fft_rows(v, zrow, tw1, j);
  }
  // samples 1024..2047 of each row are the block's output: even n = Re, odd n = -Im of element n/2, over 1024
  const float* fl = reinterpret_cast<const float*>(zbuf);
  float* y = yg + (size_t)b * d.y_stride;
  double s2 = 0.0;
  constexpr float inv = 1.f / (float)FN2;
  for (int i = tid; i < FROWS * FB; i += FNT) {
    const int r = i / FB, m = i - r * FB;
    const long gi = (long)(k0 + r) * FB + m;
    const float raw = fl[r * (2 * FZP) + FB + m];
    const float val = ((m & 1) ? -raw : raw) * inv;
    if (k0 + r < d.nblk && gi < d.L) {
      y[gi] = val;
      s2 += (double)val * (double)val;
    }
  }
  if (partial != nullptr) {
    s2 = wave_reduce_sum(s2);
    if ((tid & 63) == 0) red[tid >> 6] = s2;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int wv = 0; wv < FNT / 64; ++wv) s += red[wv];
      partial[(size_t)b * d.ntiles + tile] = s;
    }
  }
}

struct MixDims {
  long L, src_stride, pool_len;
  int nch, length;
  int clean_gain;   // 1: target = Clip(Cut(Shift(Gain(clean)))) (VADNoiseDataset), 0: Cut(Shift(clean)) (NoiseDataset)
};

// Gain then Shift (vacated samples are zero, no rollover): sample i of the transformed clip.
__device__ __forceinline__ float aug_sample(const float* __restrict__ x, long L, int shift, float gain, long i) {
  const long j = i - shift;
  const float v = x[min(max(j, 0L), L - 1)];
  return (j >= 0 && j < L) ? gain * v : 0.f;
}

__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_reduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[b][chunk][0..1] = sum of squares of the gained, shifted clip / of the noise segment over this chunk
__global__ __launch_bounds__(256) void aug_stats_kernel(const noise_aug_item* __restrict__ items, const float* __restrict__ noisy_src,
                                                        const float* __restrict__ noise_pool, double* __restrict__ partial,
                                                        MixDims d) {
  __shared__ double red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const noise_aug_item it = items[b];
  const float* x = noisy_src + (size_t)it.src * d.src_stride;
  const long i0 = (long)ch * AUG_CH;
  double s2 = 0.0, n2 = 0.0;
#pragma unroll 4
  for (int u = 0; u < AUG_CH / 256; ++u) {
    const long i = i0 + threadIdx.x + u * 256;
    const float s = aug_sample(x, d.L, it.shift, it.gain, i);
    const float nz = noise_pool[min(max(it.noise_off + min(i, (long)it.noise_len - 1), 0L), d.pool_len - 1)];
    if (i < d.L) s2 += (double)(s * s);
    if (i < it.noise_len) n2 += (double)(nz * nz);
  }
  s2 = block_sum(s2, red);
  n2 = block_sum(n2, red);
  if (threadIdx.x == 0) {
    double* p = partial + ((size_t)b * d.nch + ch) * 3;
    p[0] = s2;
    p[1] = n2;
  }
}

// Sum over the chunks of one clip of partial[.][which], in a fixed order (every thread gets the result).
__device__ __forceinline__ double clip_total(const double* __restrict__ p, int n, int stride, double* red) {
  double s = 0.0;
  for (int c = threadIdx.x; c < n; c += 256) s += p[(size_t)c * stride];
  return block_sum(s, red);
}

__global__ __launch_bounds__(256) void aug_mix_kernel(const noise_aug_item* __restrict__ items, const float* __restrict__ noisy_src,
                                                      const float* __restrict__ clean_src, const float* __restrict__ noise_pool,
                                                      double* __restrict__ partial, float* __restrict__ mixed,
                                                      float* __restrict__ out_noisy, float* __restrict__ out_clean, MixDims d) {
  __shared__ double red[4];
  const int b = blockIdx.y, ch = blockIdx.x;
  const noise_aug_item it = items[b];
  const long i0 = (long)ch * AUG_CH;
  const bool has_ir = it.ir >= 0;
  // a clip without an impulse response only leaves through the Cut window
  if (!has_ir && (i0 >= (long)it.cut + d.length || i0 + AUG_CH <= it.cut)) return;
  const double* pc = partial + (size_t)b * d.nch * 3;
  const double S2 = clip_total(pc, d.nch, 3, red);
  const double N2 = clip_total(pc + 1, d.nch, 3, red);
  float factor = 0.f;
  if (it.noise_len > 0) {
    // calculate_rms / calculate_desired_noise_rms of AddBackgroundNoise, as train/data.py:133-148 restates them
    const float clean_rms = (float)sqrt(S2 / (double)d.L), noise_rms = (float)sqrt(N2 / (double)it.noise_len);
    if (!(noise_rms < 1e-9f)) factor = (float)((double)clean_rms / pow(10.0, (double)it.snr_db / 20.0) / (double)noise_rms);
  }
  const float* x = noisy_src + (size_t)it.src * d.src_stride;
  const float* xc = clean_src + (size_t)it.src * d.src_stride;
  const int nlen = max(it.noise_len, 1);
  float* mrow = has_ir ? mixed + (size_t)it.ir_row * d.L : nullptr;
  double y2 = 0.0;
#pragma unroll 4
  for (int u = 0; u < AUG_CH / 256; ++u) {
    const long i = i0 + threadIdx.x + u * 256;
    const float s = aug_sample(x, d.L, it.shift, it.gain, i);
    float c = aug_sample(xc, d.L, it.shift, d.clean_gain ? it.gain : 1.f, i);
    if (d.clean_gain) c = fminf(fmaxf(c, -1.f), 1.f);
    const float nz = noise_pool[min(max(it.noise_off + (i % nlen), 0L), d.pool_len - 1)];
    const float y = s + nz * factor;
    const long j = i - it.cut;
    const bool in_cut = i < d.L && j >= 0 && j < d.length;
    if (in_cut) out_clean[(size_t)b * d.length + j] = c;
    if (has_ir) {
      if (i < d.L) {
        mrow[i] = y;
        y2 += (double)(y * y);
      }
    } else if (in_cut) {
      const float v = it.invert ? -y : y;
      out_noisy[(size_t)b * d.length + j] = fminf(fmaxf(v, -1.f), 1.f);
    }
  }
  if (has_ir) {
    y2 = block_sum(y2, red);
    if (threadIdx.x == 0) partial[((size_t)b * d.nch + ch) * 3 + 2] = y2;
  }
}

struct FinDims {
  long L;
  int nch, nrev, length;
};

// ImpulseResponse.apply after the convolution (train/data.py:194-208), then Cut, PolarityInversion, Clip.
__global__ __launch_bounds__(256) void aug_finish_kernel(const noise_aug_item* __restrict__ items, const int* __restrict__ item_of_row,
                                                         const float* __restrict__ reverb, const double* __restrict__ mix_partial,
                                                         const double* __restrict__ rev_partial, float* __restrict__ out_noisy,
                                                         FinDims d) {
  __shared__ double red[4];
  const int row = blockIdx.y, b = item_of_row[row];
  const noise_aug_item it = items[b];
  const double Sy = clip_total(mix_partial + (size_t)b * d.nch * 3 + 2, d.nch, 3, red);
  const double Sr = clip_total(rev_partial + (size_t)row * d.nrev, d.nrev, 1, red);
  const float in_rms = (float)sqrt(Sy / (double)d.L), out_rms = (float)sqrt(Sr / (double)d.L);
  const float scale = out_rms > 0.f ? in_rms / out_rms : 1.f;
  const float* r = reverb + (size_t)row * d.L;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < d.length; j += gridDim.x * 256) {
    const long i = min(max((long)it.cut + j, 0L), d.L - 1);
    float v = fminf(fmaxf(r[i] * scale, -1.f), 1.f);
    v = it.invert ? -v : v;
    out_noisy[(size_t)b * d.length + j] = fminf(fmaxf(v, -1.f), 1.f);
  }
}

}  // namespace

NOISE_EXPORT int noise_fir_block(void) { return FB; }
NOISE_EXPORT int noise_fir_blocks(long L) { return L > 0 ? (int)((L + FB - 1) / FB) : 0; }
NOISE_EXPORT int noise_fir_partials(long L) { return noise_cdiv(noise_fir_blocks(L), FROWS); }
NOISE_EXPORT int noise_aug_chunks(long L) { return L > 0 ? (int)((L + AUG_CH - 1) / AUG_CH) : 0; }

NOISE_EXPORT int noise_fir_spectra(const float* x, long x_stride, int n_rows, long L, int filter, const float* tw1,
                                   const float* tw2, float* spec, void* stream) {
  NOISE_REQUIRE(x && tw1 && tw2 && spec, "noise_fir_spectra: null pointer");
  NOISE_REQUIRE(n_rows > 0 && n_rows <= 65535, "noise_fir_spectra: n_rows=%d outside 1..65535", n_rows);
  NOISE_REQUIRE(L > 0 && x_stride >= 0, "noise_fir_spectra: L=%ld x_stride=%ld", L, x_stride);
  const size_t smem = fir_smem();
  NOISE_ALLOW_LDS(fir_spectra_kernel, smem);
  SpecDims d{L, x_stride, noise_fir_blocks(L), filter != 0};
  hipLaunchKernelGGL(fir_spectra_kernel, dim3(noise_cdiv(d.nblk, FROWS), n_rows), dim3(FNT), smem, (hipStream_t)stream, x,
                     (const c32*)tw1, (const c32*)tw2, (c32*)spec, d);
  NOISE_CHECK_LAUNCH("noise_fir_spectra");
  return 0;
}

NOISE_EXPORT int noise_fir_apply(const float* xspec, int n_rows, long L, const int* ir_of_row, const float* hspec,
                                 const long* ir_first_block, const int* ir_nblk, int n_ir, const float* tw1, const float* tw2,
                                 float* y, long y_stride, double* partial, void* stream) {
  NOISE_REQUIRE(xspec && ir_of_row && hspec && ir_first_block && ir_nblk && tw1 && tw2 && y, "noise_fir_apply: null pointer");
  NOISE_REQUIRE(n_rows > 0 && n_rows <= 65535, "noise_fir_apply: n_rows=%d outside 1..65535", n_rows);
  NOISE_REQUIRE(L > 0 && y_stride >= L && n_ir > 0, "noise_fir_apply: L=%ld y_stride=%ld n_ir=%d", L, y_stride, n_ir);
  const size_t smem = fir_smem();
  NOISE_ALLOW_LDS(fir_apply_kernel, smem);
  ApplyDims d{L, y_stride, noise_fir_blocks(L), n_ir, n_rows, noise_fir_partials(L)};
  NOISE_REQUIRE((long)n_rows * d.ntiles < (1L << 30), "noise_fir_apply: %d rows of %d tiles exceed the grid", n_rows, d.ntiles);
  hipLaunchKernelGGL(fir_apply_kernel, dim3(noise_pad8((long)n_rows * d.ntiles)), dim3(FNT), smem, (hipStream_t)stream,
                     (const c32*)xspec, ir_of_row, (const c32*)hspec, ir_first_block, ir_nblk, (const c32*)tw1, (const c32*)tw2, y,
                     partial, d);
  NOISE_CHECK_LAUNCH("noise_fir_apply");
  return 0;
}

NOISE_EXPORT int noise_aug_mix(const noise_aug_item* items, int n_items, const float* noisy_src, const float* clean_src,
                               long src_stride, long L, const float* noise_pool, long noise_pool_len, double* partial,
                               float* mixed, float* out_noisy, float* out_clean, int length, int clean_gain, void* stream) {
  NOISE_REQUIRE(items && noisy_src && clean_src && noise_pool && partial && out_noisy && out_clean, "noise_aug_mix: null pointer");
  NOISE_REQUIRE(n_items > 0 && n_items <= 65535, "noise_aug_mix: n_items=%d outside 1..65535", n_items);
  NOISE_REQUIRE(L > 0 && src_stride >= L && noise_pool_len > 0 && length > 0 && length <= L,
                "noise_aug_mix: L=%ld src_stride=%ld noise_pool_len=%ld length=%d", L, src_stride, noise_pool_len, length);
  MixDims d{L, src_stride, noise_pool_len, noise_aug_chunks(L), length, clean_gain != 0};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_stats_kernel, dim3(d.nch, n_items), dim3(256), 0, st, items, noisy_src, noise_pool, partial, d);
  hipLaunchKernelGGL(aug_mix_kernel, dim3(d.nch, n_items), dim3(256), 0, st, items, noisy_src, clean_src, noise_pool, partial,
                     mixed, out_noisy, out_clean, d);
  NOISE_CHECK_LAUNCH("noise_aug_mix");
  return 0;
}

NOISE_EXPORT int noise_aug_finish(const noise_aug_item* items, const int* item_of_row, int n_rows, const float* reverb, long L,
                                  const double* mix_partial, const double* rev_partial, float* out_noisy, int length,
                                  void* stream) {
  NOISE_REQUIRE(items && item_of_row && reverb && mix_partial && rev_partial && out_noisy, "noise_aug_finish: null pointer");
  NOISE_REQUIRE(n_rows > 0 && n_rows <= 65535 && L > 0 && length > 0 && length <= L, "noise_aug_finish: n_rows=%d L=%ld length=%d",
                n_rows, L, length);
  FinDims d{L, noise_aug_chunks(L), noise_fir_partials(L), length};
  hipLaunchKernelGGL(aug_finish_kernel, dim3(noise_grid(length, 256, 64), n_rows), dim3(256), 0, (hipStream_t)stream, items,
                     item_of_row, reverb, mix_partial, rev_partial, out_noisy, d);
  NOISE_CHECK_LAUNCH("noise_aug_finish");
  return 0;
}
