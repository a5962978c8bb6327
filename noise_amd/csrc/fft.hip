// STFT / iSTFT with an in-LDS FFT for MI355X (gfx950).
//
// Same operations as stft.hip (train/models.py:460-475 STFT.forward, :477-495 STFT.inverse and its adjoint), for bases
// that are a real DFT times a window, which is what the reference registers (models.py:413-458): forward_basis[f][n] =
// w[n] (cos, -sin)(2 pi f n / N) and inverse_basis[f][n] = c_f ws[n] (cos, -sin)(2 pi f n / N) / N with c = 1 for the DC and
// Nyquist rows and 2 otherwise.  The caller checks that structure and passes the two windows; any other basis goes
// through the dense products of stft.hip.
//
// A frame of N real samples is one complex FFT of N2 = N/2 points (even samples in the real parts, odd ones in the
// imaginary parts) plus a split step.  N2 = R1*R2 is done in two register passes (Cooley-Tukey): 16 threads per frame,
// each thread runs a whole radix-R2 (then radix-R1) transform in registers, the passes exchange data through one
// LDS buffer [frame][N2] that is reused in place.  Frames enter through an LDS stage filled with coalesced loads
// (reflect padding by index mirroring); spectra leave / enter with the lanes along the frame axis, so the [B][F][T]
// tensors are accessed in contiguous runs.  Overlap-add is atomic-free: every output sample gathers its N/hop frames.
#include "common.h"
#include "fft_common.h"

namespace {

struct FftDims {
  int B, L, T, hop, F, Lout, accum;
};

// Tables (device, caller-provided): win [N] window; tw1 [N2] = e^{-2 pi i j / N2}; tw2 [N2 + 1] = e^{-2 pi i k / N}.
template <int N2, int FR>
struct FftLds {
  static constexpr int ZP = N2 + 1;                     // row pitch of the complex buffer (odd: rows spread over banks)
  static constexpr size_t zbytes = (size_t)FR * ZP * 8;
};

// Pass A: thread (frame, m1) transforms z[m1 + R1 m2], m2 < R2, multiplies by e^{-2 pi i m1 k2 / N2} and stores at
// [m1 R2 + k2].  Pass B: thread (frame, k2) transforms [m1 R2 + k2], m1 < R1, in place: result index R2 k1 + k2.
// MODE 0: forward -> log-magnitude, phase, per-clip max |log-magnitude|.
// MODE 1: adjoint of the iSTFT: sig = dL/d(wave_out), masked where the output clamp was active and scaled by window_sum;
//         writes dL/d(log-magnitude).
template <int N2, int R1, int R2, int FR, int MODE>
__global__ __launch_bounds__(16 * FR) void stft_fft_kernel(const float* __restrict__ sig, const float* __restrict__ win,
                                                           const c32* __restrict__ tw1g, const c32* __restrict__ tw2g,
                                                           float* __restrict__ out0, float* __restrict__ out1,
                                                           unsigned* __restrict__ absmax_bits,
                                                           const float* __restrict__ wave_out, const float* __restrict__ wsum,
                                                           const float* __restrict__ logmag, const float* __restrict__ phase,
                                                           FftDims d) {
  constexpr int N = 2 * N2, NT = 16 * FR, ZP = FftLds<N2, FR>::ZP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  c32* zbuf = reinterpret_cast<c32*>(lds_raw);              // [FR][ZP]
  c32* tw1 = zbuf + FR * ZP;                                // [N2]
  c32* tw2 = tw1 + N2;                                      // [N2 + 1]
  float* wn = reinterpret_cast<float*>(tw2 + N2 + 1);       // [N]
  float* xs = wn + N;                                       // [(FR-1) hop + N]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * FR;
  const int hop = d.hop, half = N2;
  const int nS = (FR - 1) * hop + N;
  for (int i = tid; i < N2; i += NT) tw1[i] = tw1g[i];
  for (int i = tid; i <= N2; i += NT) tw2[i] = tw2g[i];
  for (int i = tid; i < N; i += NT) wn[i] = win[i];
  // ---- stage the samples of this tile (index in the padded signal = t0*hop + i)
  for (int i = tid; i < nS; i += NT) {
    int s = t0 * hop + i - half;
    float v;
    if (MODE == 0) {
      if (s < 0) s = -s;
      if (s >= d.L) s = 2 * (d.L - 1) - s;
      s = min(max(s, 0), d.L - 1);      // only reached by frames past T (never stored)
      v = sig[(size_t)b * d.L + s];
    } else {
      const bool in = s >= 0 && s < d.Lout;
      const size_t o = (size_t)b * d.Lout + (in ? s : 0);
      const float w = wave_out[o], g = sig[o], e = wsum[(in ? s : 0) + half];
      v = (in && w > -1.f && w < 1.f) ? g * e : 0.f;
    }
    xs[i] = v;
  }
  __syncthreads();
  const int fr = tid >> 4, j = tid & 15;
  c32* zrow = zbuf + fr * ZP;
  // ---- pass A
  if (j < R1) {
    const float* x = xs + fr * hop;
    c32 v[R2];
#pragma unroll
    for (int m2 = 0; m2 < R2; ++m2) {
      const int n = 2 * (j + R1 * m2);
      v[m2] = {x[n] * wn[n], x[n + 1] * wn[n + 1]};
    }
    Dft<R2>::run(v);
    zrow[j * R2] = v[0];
#pragma unroll
    for (int k2 = 1; k2 < R2; ++k2) zrow[j * R2 + k2] = cmul(v[k2], tw1[j * k2]);
  }
  __syncthreads();
  fft_pass_b<R1, R2>(zrow, j);
  __syncthreads();
  // ---- split step + epilogue: thread = (frame, bin group); X[k] = E + W^k O, X[N2-k] = conj(E - W^k O) with
  // E = (Z[k] + conj Z[N2-k])/2, O = (Z[k] - conj Z[N2-k])/(2i)
  const int f2 = tid % FR, kg = tid / FR;        // 16 bin groups
  const int t = t0 + f2;
  const c32* zr = zbuf + f2 * ZP;
  float amax = 0.f;
  const bool tin = t < d.T;
  const int tc = tin ? t : d.T - 1;
  for (int k = kg; k <= N2 / 2; k += 16) {
    const c32 zk = zr[k], zn = zr[k == 0 ? 0 : N2 - k];
    const c32 e = {0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)};
    const c32 o = {0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x)};
    const c32 wo = cmul(tw2[k], o);
    c32 xa = cadd(e, wo), xb = cconj(csub(e, wo));
    if (k == 0) {          // DC and Nyquist: both real
      xa = {zk.x + zk.y, 0.f};
      xb = {zk.x - zk.y, 0.f};
    }
    const int ka = k, kb = N2 - k;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int f = side == 0 ? ka : kb;
      if (side == 1 && kb == ka) continue;
      const c32 xv = side == 0 ? xa : xb;
      const size_t oidx = ((size_t)b * d.F + f) * d.T + tc;
      if (MODE == 0) {
        const float lm = logf(fmaxf(sqrtf(xv.x * xv.x + xv.y * xv.y), 1e-3f));
        if (tin) {
          out0[oidx] = lm;
          if (out1 != nullptr) out1[oidx] = atan2f(xv.y, xv.x);
          amax = fmaxf(amax, fabsf(lm));
        }
      } else {
        // rows of inverse_basis: c_f ws[n] (cos, -sin) / N -> the transform of the windowed gradient times c_f / N
        const float cf = (f == 0 || f == N2) ? 1.f / (float)N : 2.f / (float)N;
        float sn, cs;
        sincosf(phase[oidx], &sn, &cs);
        const float gv = expf(logmag[oidx]) * cf * (cs * xv.x + sn * xv.y);
        const float prev = d.accum ? out0[oidx] : 0.f;
        if (tin) out0[oidx] = prev + gv;
      }
    }
  }
  if (MODE == 0 && absmax_bits != nullptr) {
    amax = wave_reduce_max(amax);
    if ((tid & 63) == 0) atomicMax(absmax_bits + b, __float_as_uint(amax));
  }
}

// Inverse: spectrum -> frames (complex N2-point inverse transform via conjugation) -> synthesis window -> overlap-add.
// A workgroup owns FR frames = FR - (N/hop - 1) output hops.
template <int N2, int R1, int R2, int FR>
__global__ __launch_bounds__(16 * FR) void istft_fft_kernel(const float* __restrict__ logmag, const float* __restrict__ phase,
                                                            const float* __restrict__ win, const c32* __restrict__ tw1g,
                                                            const c32* __restrict__ tw2g, const float* __restrict__ wsum,
                                                            const float* __restrict__ vad_logit, float* __restrict__ out,
                                                            FftDims d) {
  constexpr int N = 2 * N2, NT = 16 * FR, ZP = FftLds<N2, FR>::ZP;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  c32* zbuf = reinterpret_cast<c32*>(lds_raw);              // [FR][ZP]
  c32* tw1 = zbuf + FR * ZP;
  c32* tw2 = tw1 + N2;
  float* wn = reinterpret_cast<float*>(tw2 + N2 + 1);
  const int tid = threadIdx.x;
  const int b = blockIdx.y, hop = d.hop;
  const int R = N / hop;
  const int TFo = FR - (R - 1);
  const int q0 = blockIdx.x * TFo;        // first output hop of this tile
  const int tstart = q0 - (R - 1);        // first frame of this tile
  for (int i = tid; i < N2; i += NT) tw1[i] = tw1g[i];
  for (int i = tid; i <= N2; i += NT) tw2[i] = tw2g[i];
  for (int i = tid; i < N; i += NT) wn[i] = win[i];
  __syncthreads();
  // ---- spectrum -> conj(Z): Z[k] = E + iO, Z[N2-k] = conj(E) + i conj(O) with E = (X[k] + conj X[N2-k])/2,
  // O = (X[k] - conj X[N2-k]) conj(W^k)/2; the imaginary parts of the DC and Nyquist bins are not used
  {
    const int f2 = tid % FR, kg = tid / FR;
    const int t = tstart + f2;
    const bool tin = t >= 0 && t < d.T;
    const int tc = t < 0 ? 0 : (t < d.T ? t : d.T - 1);
    c32* zr = zbuf + f2 * ZP;
    for (int k = kg; k <= N2 / 2; k += 16) {
      const int kb = N2 - k;
      const size_t oa = ((size_t)b * d.F + k) * d.T + tc, ob = ((size_t)b * d.F + kb) * d.T + tc;
      const float ma = expf(logmag[oa]), mb = expf(logmag[ob]);
      float sa, ca, sb, cb;
      sincosf(phase[oa], &sa, &ca);
      sincosf(phase[ob], &sb, &cb);
      c32 xa = {ma * ca, ma * sa}, xb = {mb * cb, mb * sb};
      if (!tin) xa = xb = {0.f, 0.f};
      if (k == 0) {
        zr[0] = {0.5f * (xa.x + xb.x), -0.5f * (xa.x - xb.x)};     // conj(E + iO), E and O real
      } else {
        const c32 e = {0.5f * (xa.x + xb.x), 0.5f * (xa.y - xb.y)};
        const c32 df = {0.5f * (xa.x - xb.x), 0.5f * (xa.y + xb.y)};
        const c32 o = cmul(df, cconj(tw2[k]));
        // Z[k] = E + iO = (e.x - o.y, e.y + o.x);  Z[N2-k] = conj(E) + i conj(O) = (e.x + o.y, -e.y + o.x)
        zr[k] = {e.x - o.y, -(e.y + o.x)};
        zr[kb] = {e.x + o.y, -(-e.y + o.x)};
      }
    }
  }
  __syncthreads();
  const int fr = tid >> 4, j = tid & 15;
  c32* zrow = zbuf + fr * ZP;
  // ---- pass A: every thread reads its inputs, then (after a barrier) the rows are rewritten in pass-B order
  c32 v[R2];
  if (j < R1) {
#pragma unroll
    for (int m2 = 0; m2 < R2; ++m2) v[m2] = zrow[j + R1 * m2];
  }
  __syncthreads();
  if (j < R1) {
    Dft<R2>::run(v);
    zrow[j * R2] = v[0];
#pragma unroll
    for (int k2 = 1; k2 < R2; ++k2) zrow[j * R2 + k2] = cmul(v[k2], tw1[j * k2]);
  }
  __syncthreads();
  fft_pass_b<R1, R2>(zrow, j);
  __syncthreads();
  // zbuf[fr][n] = N2 conj(z[n]): sample 2n = Re / N2, sample 2n+1 = -Im / N2
  // ---- atomic-free overlap-add: every output sample gathers its R frames
  float gate = 1.f;
  if (vad_logit != nullptr) gate = 1.f / (1.f + expf(-1e4f * vad_logit[b]));
  const float* fl = reinterpret_cast<const float*>(zbuf);
  constexpr float inv = 1.f / (float)N2;
  for (int i = tid; i < TFo * hop; i += NT) {
    const int jj = i / hop, m = i - jj * hop;
    const int s = (q0 + jj) * hop + m;
    const int so = s - N2;
    if (so < 0 || so >= d.Lout) continue;
    float y = 0.f;
    for (int r = 0; r < R; ++r) {
      const int n = r * hop + m;
      const float fv = fl[(jj + R - 1 - r) * (2 * ZP) + n];
      y += ((n & 1) ? -fv : fv) * wn[n];
    }
    y *= inv * wsum[s];
    y = fminf(fmaxf(y, -1.f), 1.f) * gate;
    out[(size_t)b * d.Lout + so] = y;
  }
}

template <int N2, int FR>
size_t fwd_smem(int hop) {
  return FftLds<N2, FR>::zbytes + (size_t)(N2 + N2 + 1) * 8 + (size_t)(2 * N2) * 4 + (size_t)((FR - 1) * hop + 2 * N2) * 4;
}
template <int N2, int FR>
size_t inv_smem() {
  return FftLds<N2, FR>::zbytes + (size_t)(N2 + N2 + 1) * 8 + (size_t)(2 * N2) * 4;
}

constexpr int FWD_FR = 16, INV_FR = 32;

template <int MODE>
int launch_fwd(int n_fft, const float* sig, const float* win, const float* tw1, const float* tw2, float* out0, float* out1,
               unsigned* absmax, const float* wave_out, const float* wsum, const float* logmag, const float* phase,
               FftDims d, hipStream_t st) {
  dim3 grid(noise_cdiv(d.T, FWD_FR), d.B);
#define FWD_CASE(NN, RA, RB)                                                                                          \
  if (n_fft == NN) {                                                                                                  \
    const size_t smem = fwd_smem<NN / 2, FWD_FR>(d.hop);                                                              \
    auto kfn = stft_fft_kernel<NN / 2, RA, RB, FWD_FR, MODE>;                                                         \
    NOISE_ALLOW_LDS(kfn, smem);                                                                                       \
    hipLaunchKernelGGL(kfn, grid, dim3(16 * FWD_FR), smem, st, sig, win, (const c32*)tw1, (const c32*)tw2, out0, out1, \
                       absmax, wave_out, wsum, logmag, phase, d);                                                     \
    return 0;                                                                                                         \
  }
  FWD_CASE(320, 16, 10)
  FWD_CASE(512, 16, 16)
#undef FWD_CASE
  return 1;
}

int check_fft(const char* who, int B, int n_fft, int hop) {
  NOISE_REQUIRE(B > 0, "%s: empty batch", who);
  NOISE_REQUIRE(n_fft == 320 || n_fft == 512, "%s: the FFT kernels serve n_fft 320 and 512 (got %d); use the dense-basis entry points", who, n_fft);
  NOISE_REQUIRE(hop > 0 && hop % 2 == 0 && n_fft % hop == 0, "%s: hop=%d must be even and divide n_fft=%d", who, hop, n_fft);
  return 0;
}

}  // namespace

NOISE_EXPORT int noise_fft_supported(int n_fft) { return n_fft == 320 || n_fft == 512; }

NOISE_EXPORT int noise_stft_fwd_fft(const float* wave, const float* window, const float* tw1, const float* tw2, float* logmag,
                                    float* phase, float* absmax, int B, int L, int n_fft, int hop, void* stream) {
  if (check_fft("noise_stft_fwd_fft", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(wave && window && tw1 && tw2 && logmag, "noise_stft_fwd_fft: null pointer");
  NOISE_REQUIRE(L > n_fft / 2, "noise_stft_fwd_fft: L=%d too short for reflect padding of %d", L, n_fft / 2);
  hipStream_t st = (hipStream_t)stream;
  FftDims d{B, L, L / hop + 1, hop, n_fft / 2 + 1, hop * (L / hop), 0};
  if (absmax) {
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * B, st);
    NOISE_REQUIRE(e == hipSuccess, "noise_stft_fwd_fft: memset failed: %s", hipGetErrorString(e));
  }
  launch_fwd<0>(n_fft, wave, window, tw1, tw2, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  NOISE_CHECK_LAUNCH("noise_stft_fwd_fft");
  return 0;
}

NOISE_EXPORT int noise_istft_fwd_fft(const float* logmag, const float* phase, const float* syn_window, const float* tw1,
                                     const float* tw2, const float* window_sum, const float* vad_logit, float* wave_out, int B,
                                     int T, int n_fft, int hop, void* stream) {
  if (check_fft("noise_istft_fwd_fft", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(logmag && phase && syn_window && tw1 && tw2 && window_sum && wave_out, "noise_istft_fwd_fft: null pointer");
  NOISE_REQUIRE(T > 1, "noise_istft_fwd_fft: need at least 2 frames");
  const int R = n_fft / hop;
  NOISE_REQUIRE(R < INV_FR, "noise_istft_fwd_fft: n_fft/hop=%d too large", R);
  hipStream_t st = (hipStream_t)stream;
  FftDims d{B, 0, T, hop, n_fft / 2 + 1, hop * (T - 1), 0};
  const int TFo = INV_FR - (R - 1), hops = T - 1 + R;
  dim3 grid(noise_cdiv(hops, TFo), B);
#define INV_CASE(NN, RA, RB)                                                                                            \
  if (n_fft == NN) {                                                                                                    \
    const size_t smem = inv_smem<NN / 2, INV_FR>();                                                                     \
    auto kfn = istft_fft_kernel<NN / 2, RA, RB, INV_FR>;                                                                \
    NOISE_ALLOW_LDS(kfn, smem);                                                                                         \
    hipLaunchKernelGGL(kfn, grid, dim3(16 * INV_FR), smem, st, logmag, phase, syn_window, (const c32*)tw1, (const c32*)tw2, \
                       window_sum, vad_logit, wave_out, d);                                                             \
  }
  INV_CASE(320, 16, 10)
  INV_CASE(512, 16, 16)
#undef INV_CASE
  NOISE_CHECK_LAUNCH("noise_istft_fwd_fft");
  return 0;
}

NOISE_EXPORT int noise_istft_bwd_fft(const float* d_wave, const float* wave_out, const float* logmag, const float* phase,
                                     const float* syn_window, const float* tw1, const float* tw2, const float* window_sum,
                                     float* d_logmag, int accumulate, int B, int T, int n_fft, int hop, void* stream) {
  if (check_fft("noise_istft_bwd_fft", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(d_wave && wave_out && logmag && phase && syn_window && tw1 && tw2 && window_sum && d_logmag,
                "noise_istft_bwd_fft: null pointer");
  NOISE_REQUIRE(T > 1, "noise_istft_bwd_fft: need at least 2 frames");
  FftDims d{B, 0, T, hop, n_fft / 2 + 1, hop * (T - 1), accumulate};
  launch_fwd<1>(n_fft, d_wave, syn_window, tw1, tw2, d_logmag, nullptr, nullptr, wave_out, window_sum, logmag, phase, d,
                (hipStream_t)stream);
  NOISE_CHECK_LAUNCH("noise_istft_bwd_fft");
  return 0;
}
