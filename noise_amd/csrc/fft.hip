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

// Cheaper transcendental functions for the bf16 inference pipeline (absolute error about 1e-6 in the sine / cosine and
// 2e-6 rad in the angle, far below that path's bf16 rounding); the fp32 paths keep the library functions.
__device__ __forceinline__ float fast_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mn * __frcp_rn(fmaxf(mx, 1e-37f));
  const float s = a * a;
  // minimax odd polynomial of atan on [0, 1], degree 13
  float r = -0.0117212f;
  r = fmaf(r, s, 0.05265332f);
  r = fmaf(r, s, -0.11643287f);
  r = fmaf(r, s, 0.19354346f);
  r = fmaf(r, s, -0.33262347f);
  r = fmaf(r, s, 0.99997726f);
  r = r * a;
  r = ay > ax ? 1.57079637f - r : r;
  r = x < 0.f ? 3.14159274f - r : r;
  return copysignf(r, y);
}
__device__ __forceinline__ void fast_sincos(float a, float* sn, float* cs) {
  *sn = __sinf(a);
  *cs = __cosf(a);
}

#ifdef FFT_STAMPS
__device__ unsigned long long g_fft_stamps[8];
#define FFT_T(i) do { if (blockIdx.x == 8 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_fft_stamps[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define FFT_T(i)
#endif

struct FftDims {
  int B, L, T, hop, F, Lout, accum;
  int ntiles;   // frame tiles per clip; the grid is B * ntiles workgroups, padded to a multiple of 8
  int mark_floor;   // forward: a clip with a bin far below the 1e-3 magnitude floor gets absmax = floor_mark() (see below)
  float floor_mag;  // 1e-3f (a run-time value on purpose, see log_floor)
};

// |log(1e-3)| as the device's logf gives it (evaluated at run time from `floor_mag` = 1e-3f, so that no host-side constant
// folding can round it differently), and the next float above it.  A clip whose per-clip max carries the mark holds a bin
// whose magnitude is below 0.9e-3: whatever the summation order, that bin sits on the floor and the clip's max
// |log-magnitude| is exactly the floor value (magnitudes of [-1, 1] audio stay far below e^6.9), so
// stft_absmax_refine_kernel has nothing to redo.
__device__ __forceinline__ float log_floor(float floor_mag) { return fabsf(logf(floor_mag)); }
__device__ __forceinline__ float floor_mark(float floor_mag) { return __uint_as_float(__float_as_uint(log_floor(floor_mag)) + 1u); }

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
template <int N2, int R1, int R2, int FR, int MODE, bool FAST>
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
  // The staged samples [(FR-1) hop + N] share the complex buffer: pass A reads its samples into registers, a barrier, and
  // only then writes its results over them.  One buffer less is another resident workgroup per CU.
  float* xs = reinterpret_cast<float*>(lds_raw);
  static_assert((size_t)FR * N * 4 <= FftLds<N2, FR>::zbytes, "the staged samples must fit in the complex buffer");
  const int tid = threadIdx.x;
#ifdef FFT_STAMPS
  unsigned long long last_ = clock64();
#endif
  // Consecutive frame tiles of a clip write (forward) or read (adjoint) neighbouring runs of the same [F][T] rows, and
  // with an odd T the runs share 128-byte lines: placed on one XCD they meet in one L2 (the writes merge into full
  // lines, the reads fetch a line once) instead of producing partial-line traffic from eight L2s.
  const int vblk = xcd_chunked_block();
  if (vblk >= d.B * d.ntiles) return;
  const int b = vblk / d.ntiles, t0 = (vblk - b * d.ntiles) * FR;
  const int hop = d.hop, half = N2;
  const int nS = (FR - 1) * hop + N;
  // ---- prologue: every global load (tables and this tile's samples; index in the padded signal = t0*hop + i) is
  // requested before the first LDS store, from clamped addresses, so the workgroup pays one memory round trip
  // instead of one per staging loop
  constexpr int XPT = N / 16;                    // samples per thread at the largest hop (nS <= FR*N = XPT*NT)
  constexpr int TWN = (N2 + NT) / NT, WNN = (N + NT - 1) / NT;
  float xv[XPT], xw[MODE == 1 ? XPT : 1], xe[MODE == 1 ? XPT : 1];
  c32 t1r[TWN], t2r[TWN];
  float wr[WNN];
#pragma unroll
  for (int u = 0; u < XPT; ++u) {
    const int i = tid + u * NT;
    int s = t0 * hop + (i < nS ? i : 0) - half;
    if (MODE == 0) {
      if (s < 0) s = -s;
      if (s >= d.L) s = 2 * (d.L - 1) - s;
      s = min(max(s, 0), d.L - 1);      // only reached by frames past T (never stored)
      xv[u] = sig[(size_t)b * d.L + s];
    } else {
      const bool in = s >= 0 && s < d.Lout;
      const size_t o = (size_t)b * d.Lout + (in ? s : 0);
      xw[u] = wave_out[o];
      xv[u] = sig[o];
      xe[u] = wsum[(in ? s : 0) + half];
    }
  }
#pragma unroll
  for (int u = 0; u < TWN; ++u) {
    const int k = tid + u * NT;
    t1r[u] = tw1g[k < N2 ? k : N2 - 1];
    t2r[u] = tw2g[k <= N2 ? k : N2];
  }
#pragma unroll
  for (int u = 0; u < WNN; ++u) wr[u] = win[min(tid + u * NT, N - 1)];
#pragma unroll
  for (int u = 0; u < TWN; ++u) {
    const int k = tid + u * NT;
    if (k < N2) tw1[k] = t1r[u];
    if (k <= N2) tw2[k] = t2r[u];
  }
#pragma unroll
  for (int u = 0; u < WNN; ++u)
    if (tid + u * NT < N) wn[tid + u * NT] = wr[u];
#pragma unroll
  for (int u = 0; u < XPT; ++u) {
    const int i = tid + u * NT;
    if (i < nS) {
      float v = xv[u];
      if (MODE == 1) {
        const int s = t0 * hop + i - half;
        v = (s >= 0 && s < d.Lout && xw[u] > -1.f && xw[u] < 1.f) ? v * xe[u] : 0.f;
      }
      xs[i] = v;
    }
  }
  FFT_T(0);
  __syncthreads();
  FFT_T(1);
  const int fr = tid >> 4, j = tid & 15;
  c32* zrow = zbuf + fr * ZP;
  // ---- pass A
  c32 v[R2];
  if (j < R1) {
    const float* x = xs + fr * hop;
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
  FFT_T(2);
  __syncthreads();
  FFT_T(3);
  fft_pass_b<R1, R2>(zrow, j);
  FFT_T(4);
  __syncthreads();
  FFT_T(5);
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
        const float lm = FAST ? 0.5f * __logf(fmaxf(xv.x * xv.x + xv.y * xv.y, 1e-6f))
                              : logf(fmaxf(sqrtf(xv.x * xv.x + xv.y * xv.y), 1e-3f));
#ifdef FFT_DBG_NOSTORE
        if (tin && lm == 123.456f) {
#else
        if (tin) {
#endif
          out0[oidx] = lm;
          if (out1 != nullptr) out1[oidx] = FAST ? fast_atan2(xv.y, xv.x) : atan2f(xv.y, xv.x);
          amax = fmaxf(amax, fabsf(lm));
          if (d.mark_floor && xv.x * xv.x + xv.y * xv.y < 0.81e-6f) amax = fmaxf(amax, floor_mark(d.floor_mag));
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
  FFT_T(6);
  if (MODE == 0 && absmax_bits != nullptr) {
    amax = wave_reduce_max(amax);
    if ((tid & 63) == 0) atomicMax(absmax_bits + b, __float_as_uint(amax));
  }
}

// Inverse: spectrum -> frames (complex N2-point inverse transform via conjugation) -> synthesis window -> overlap-add.
// A workgroup owns FR frames = FR - (N/hop - 1) output hops.
template <int N2, int R1, int R2, int FR, bool FAST>
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
  const int vblk = xcd_chunked_block();      // see stft_fft_kernel: neighbouring tiles share cache lines of the spectra
  if (vblk >= d.B * d.ntiles) return;
  const int b = vblk / d.ntiles, hop = d.hop;
  const int R = N / hop;
  const int TFo = FR - (R - 1);
  const int q0 = (vblk - b * d.ntiles) * TFo;        // first output hop of this tile
  const int tstart = q0 - (R - 1);        // first frame of this tile
  // ---- prologue: the tables and this thread's share of the spectrum are all requested before the first LDS store
  // (one memory round trip per workgroup)
  constexpr int TWN = (N2 + NT) / NT, WNN = (N + NT - 1) / NT;
  constexpr int KI = (N2 / 2 + 16) / 16;         // bin pairs (k, N2 - k) per thread: k = kg + 16 u <= N2/2
  const int f2 = tid % FR, kg = tid / FR;
  const int t = tstart + f2;
  const bool tin = t >= 0 && t < d.T;
  const int tc = t < 0 ? 0 : (t < d.T ? t : d.T - 1);
  float la[KI], lb[KI], pa[KI], pb[KI];
  c32 twk[KI];                                   // e^{-2 pi i k / N} of this thread's bins, straight from the table
#pragma unroll
  for (int u = 0; u < KI; ++u) {
    const int k = min(kg + 16 * u, N2 / 2), kb = N2 - k;
    const size_t oa = ((size_t)b * d.F + k) * d.T + tc, ob = ((size_t)b * d.F + kb) * d.T + tc;
    la[u] = logmag[oa];
    lb[u] = logmag[ob];
    pa[u] = phase[oa];
    pb[u] = phase[ob];
    twk[u] = tw2g[k];
  }
  c32 t1r[TWN];
  float wr[WNN];
#pragma unroll
  for (int u = 0; u < TWN; ++u) {
    const int k = tid + u * NT;
    t1r[u] = tw1g[k < N2 ? k : N2 - 1];
  }
#pragma unroll
  for (int u = 0; u < WNN; ++u) wr[u] = win[min(tid + u * NT, N - 1)];
#pragma unroll
  for (int u = 0; u < TWN; ++u) {
    const int k = tid + u * NT;
    if (k < N2) tw1[k] = t1r[u];
  }
#pragma unroll
  for (int u = 0; u < WNN; ++u)
    if (tid + u * NT < N) wn[tid + u * NT] = wr[u];
  // ---- spectrum -> conj(Z): Z[k] = E + iO, Z[N2-k] = conj(E) + i conj(O) with E = (X[k] + conj X[N2-k])/2,
  // O = (X[k] - conj X[N2-k]) conj(W^k)/2; the imaginary parts of the DC and Nyquist bins are not used.
  {
    c32* zr = zbuf + f2 * ZP;
#pragma unroll
    for (int u = 0; u < KI; ++u) {
      const int k = kg + 16 * u;
      if (k > N2 / 2) continue;
      const int kb = N2 - k;
      const float ma = FAST ? __expf(la[u]) : expf(la[u]), mb = FAST ? __expf(lb[u]) : expf(lb[u]);
      float sa, ca, sb, cb;
      if (FAST) {
        fast_sincos(pa[u], &sa, &ca);
        fast_sincos(pb[u], &sb, &cb);
      } else {
        sincosf(pa[u], &sa, &ca);
        sincosf(pb[u], &sb, &cb);
      }
      c32 xa = {ma * ca, ma * sa}, xb = {mb * cb, mb * sb};
      if (!tin) xa = xb = {0.f, 0.f};
      if (k == 0) {
        zr[0] = {0.5f * (xa.x + xb.x), -0.5f * (xa.x - xb.x)};     // conj(E + iO), E and O real
      } else {
        const c32 e = {0.5f * (xa.x + xb.x), 0.5f * (xa.y - xb.y)};
        const c32 df = {0.5f * (xa.x - xb.x), 0.5f * (xa.y + xb.y)};
        const c32 o = cmul(df, cconj(twk[u]));
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

// Per-clip normaliser max |log-magnitude| (train/models.py:614-620) in the accumulation order of the dense kernel
// (stft_frames_kernel in stft.hip: one k-ordered fmaf chain per bin, the order of the reference's conv1d).  The max is
// attained on the bin of smallest magnitude, a difference of large numbers whose fp32 rounding depends on the summation
// order by ~1e-4 of the log; the FFT spectrum is as accurate but rounds differently, and every normalised input (and
// gradient) of the clip inherits that difference.  Only the few bins within RANGE of the FFT max can attain the dense
// max, so those are recomputed as dense dot products: one lane per candidate, 2 x n_fft sequential fmaf.
constexpr int REFINE_MAX_CAND = 2048;
__global__ __launch_bounds__(256) void stft_absmax_refine_kernel(const float* __restrict__ wave, const float* __restrict__ basisT,
                                                                 const float* __restrict__ logmag, float* __restrict__ absmax,
                                                                 int L, int T, int F, int n_fft, int hop, int rpad,
                                                                 float floor_mag) {
  __shared__ int cand[REFINE_MAX_CAND];
  __shared__ int ncand;
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float m0 = absmax[b];
  if (m0 == floor_mark(floor_mag)) {   // a bin far below the floor: the max is the floor in any order
    if (tid == 0) absmax[b] = log_floor(floor_mag);
    return;
  }
  constexpr float RANGE = 2e-3f;
  const float* lm = logmag + (size_t)b * F * T;
  const float* x = wave + (size_t)b * L;
  const int half = n_fft / 2, total = F * T;
  float best = 0.f;
  const float thr = m0 - RANGE;
  // one lane per candidate: the k-ordered chain of the dense kernel; the operands of 8 steps are requested together
  auto recompute = [&](int n) {
    for (int c = tid; c < n; c += 256) {
      const int i = cand[c], f = i / T, t = i - f * T;
      const float* bre = basisT + 2 * f;
      const int s0 = t * hop - half;
      float re = 0.f, im = 0.f;
      for (int k0 = 0; k0 < n_fft; k0 += 8) {
        float xv[8], br[8], bi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = min(k0 + u, n_fft - 1);
          int sidx = s0 + k;
          if (sidx < 0) sidx = -sidx;
          if (sidx >= L) sidx = 2 * (L - 1) - sidx;
          xv[u] = x[sidx];
          br[u] = bre[(size_t)k * rpad];
          bi[u] = bre[(size_t)k * rpad + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (k0 + u < n_fft) {
            re = fmaf(br[u], xv[u], re);
            im = fmaf(bi[u], xv[u], im);
          }
        }
      }
      best = fmaxf(best, fabsf(logf(fmaxf(sqrtf(re * re + im * im), 1e-3f))));
    }
  };
  // Fast path: one sweep over the clip's spectrum (every thread requests 8 elements at a time); a list that overflows
  // (thousands of bins within RANGE of the max: not a signal the path meets) falls back to sweeping in rounds.
  if (tid == 0) ncand = 0;
  __syncthreads();
  for (int i0 = 0; i0 < total; i0 += 2048) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = lm[min(i0 + tid + 256 * u, total - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + tid + 256 * u;
      if (i < total && fabsf(v[u]) >= thr) {
        const int slot = atomicAdd(&ncand, 1);
        if (slot < REFINE_MAX_CAND) cand[slot] = i;
      }
    }
  }
  __syncthreads();
  const int nall = ncand;   // uniform
  if (nall <= REFINE_MAX_CAND) {
    recompute(nall);
  } else {
    int base = 0;
    while (base < total) {   // every value that steers the control flow is uniform over the workgroup
      __syncthreads();
      if (tid == 0) ncand = 0;
      __syncthreads();
      int i0 = base;
      while (i0 < total) {
        const int i = i0 + tid;
        if (i < total && fabsf(lm[i]) >= thr) cand[atomicAdd(&ncand, 1)] = i;
        i0 += 256;
        __syncthreads();
        const int nc = ncand;
        __syncthreads();
        if (nc > REFINE_MAX_CAND - 256) break;
      }
      base = i0;
      recompute(ncand);
    }
  }
  best = wave_reduce_max(best);
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) absmax[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

template <int N2, int FR>
size_t fwd_smem(int hop) {
  (void)hop;   // the staged samples ((FR-1) hop + N floats) live inside the complex buffer
  return FftLds<N2, FR>::zbytes + (size_t)(N2 + N2 + 1) * 8 + (size_t)(2 * N2) * 4;
}
template <int N2, int FR>
size_t inv_smem() {
  return FftLds<N2, FR>::zbytes + (size_t)(N2 + N2 + 1) * 8 + (size_t)(2 * N2) * 4;
}

#ifndef FWD_FR_320
#define FWD_FR_320 16
#endif
#ifndef FWD_FR_512
#define FWD_FR_512 16
#endif
constexpr int INV_FR = 16;   // frames per workgroup (measured: 32, one 128-byte line of [B][F][T] per bin, is equal at n_fft 320 and 7 % slower at 512, where it leaves two workgroups per CU)

template <int MODE, bool FAST>
int launch_fwd(int n_fft, const float* sig, const float* win, const float* tw1, const float* tw2, float* out0, float* out1,
               unsigned* absmax, const float* wave_out, const float* wsum, const float* logmag, const float* phase,
               FftDims d, hipStream_t st) {
#define FWD_CASE(NN, RA, RB, FRAMES)                                                                                  \
  if (n_fft == NN) {                                                                                                  \
    d.ntiles = noise_cdiv(d.T, FRAMES);                                                                               \
    const dim3 grid(noise_pad8((long)d.ntiles * d.B));                                                                \
    const size_t smem = fwd_smem<NN / 2, FRAMES>(d.hop);                                                              \
    auto kfn = stft_fft_kernel<NN / 2, RA, RB, FRAMES, MODE, FAST>;                                                   \
    NOISE_ALLOW_LDS(kfn, smem);                                                                                       \
    hipLaunchKernelGGL(kfn, grid, dim3(16 * FRAMES), smem, st, sig, win, (const c32*)tw1, (const c32*)tw2, out0, out1, \
                       absmax, wave_out, wsum, logmag, phase, d);                                                     \
    return 0;                                                                                                         \
  }
  FWD_CASE(320, 16, 10, FWD_FR_320)
  FWD_CASE(512, 16, 16, FWD_FR_512)
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

#ifdef FFT_STAMPS
NOISE_EXPORT int noise_debug_fft_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[8] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fft_stamps), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fft_stamps), 8 * sizeof(unsigned long long));
}
#endif

NOISE_EXPORT int noise_fft_supported(int n_fft) { return n_fft == 320 || n_fft == 512; }

// fast_math != 0: hardware-rate log / angle (the bf16 inference pipeline); 0: library functions (fp32 parity).
NOISE_EXPORT int noise_stft_fwd_fft(const float* wave, const float* window, const float* tw1, const float* tw2, float* logmag,
                                    float* phase, float* absmax, int B, int L, int n_fft, int hop, int fast_math,
                                    void* stream) {
  if (check_fft("noise_stft_fwd_fft", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(wave && window && tw1 && tw2 && logmag, "noise_stft_fwd_fft: null pointer");
  NOISE_REQUIRE(L > n_fft / 2, "noise_stft_fwd_fft: L=%d too short for reflect padding of %d", L, n_fft / 2);
  hipStream_t st = (hipStream_t)stream;
  FftDims d{B, L, L / hop + 1, hop, n_fft / 2 + 1, hop * (L / hop), 0, 0};
  if (absmax) {
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * B, st);
    NOISE_REQUIRE(e == hipSuccess, "noise_stft_fwd_fft: memset failed: %s", hipGetErrorString(e));
  }
  if (fast_math)
    launch_fwd<0, true>(n_fft, wave, window, tw1, tw2, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  else
    launch_fwd<0, false>(n_fft, wave, window, tw1, tw2, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  NOISE_CHECK_LAUNCH("noise_stft_fwd_fft");
  return 0;
}

// The same transform with the per-clip normaliser evaluated as noise_stft_fwd evaluates it (the reference's conv1d order):
// the training step's network input.  basisT: the dense kernel's basis layout [n_fft][noise_stft_rpad(n_fft)].
NOISE_EXPORT int noise_stft_fwd_fft_ref(const float* wave, const float* window, const float* tw1, const float* tw2,
                                        const float* basisT, float* logmag, float* phase, float* absmax, int B, int L,
                                        int n_fft, int hop, void* stream) {
  if (check_fft("noise_stft_fwd_fft_ref", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(wave && window && tw1 && tw2 && basisT && logmag && absmax, "noise_stft_fwd_fft_ref: null pointer");
  NOISE_REQUIRE(L > n_fft / 2, "noise_stft_fwd_fft_ref: L=%d too short for reflect padding of %d", L, n_fft / 2);
  hipStream_t st = (hipStream_t)stream;
  FftDims d{B, L, L / hop + 1, hop, n_fft / 2 + 1, hop * (L / hop), 0, 0, 1, 1e-3f};
  hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * B, st);
  NOISE_REQUIRE(e == hipSuccess, "noise_stft_fwd_fft_ref: memset failed: %s", hipGetErrorString(e));
  launch_fwd<0, false>(n_fft, wave, window, tw1, tw2, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  NOISE_CHECK_LAUNCH("noise_stft_fwd_fft_ref");
  const int F = n_fft / 2 + 1, rpad = (2 * F + 31) / 32 * 32;
  hipLaunchKernelGGL(stft_absmax_refine_kernel, dim3(B), dim3(256), 0, st, wave, basisT, logmag, absmax, L, d.T, F, n_fft, hop,
                     rpad, 1e-3f);
  NOISE_CHECK_LAUNCH("noise_stft_fwd_fft_ref/absmax");
  return 0;
}

NOISE_EXPORT int noise_istft_fwd_fft(const float* logmag, const float* phase, const float* syn_window, const float* tw1,
                                     const float* tw2, const float* window_sum, const float* vad_logit, float* wave_out, int B,
                                     int T, int n_fft, int hop, int fast_math, void* stream) {
  if (check_fft("noise_istft_fwd_fft", B, n_fft, hop)) return -1;
  NOISE_REQUIRE(logmag && phase && syn_window && tw1 && tw2 && window_sum && wave_out, "noise_istft_fwd_fft: null pointer");
  NOISE_REQUIRE(T > 1, "noise_istft_fwd_fft: need at least 2 frames");
  const int R = n_fft / hop;
  NOISE_REQUIRE(R < INV_FR, "noise_istft_fwd_fft: n_fft/hop=%d too large", R);
  hipStream_t st = (hipStream_t)stream;
  FftDims d{B, 0, T, hop, n_fft / 2 + 1, hop * (T - 1), 0, 0};
  const int TFo = INV_FR - (R - 1), hops = T - 1 + R;
  d.ntiles = noise_cdiv(hops, TFo);
  dim3 grid(noise_pad8((long)d.ntiles * B));
#define INV_CASE(NN, RA, RB, FAST)                                                                                      \
  if (n_fft == NN && (fast_math != 0) == FAST) {                                                                        \
    const size_t smem = inv_smem<NN / 2, INV_FR>();                                                                     \
    auto kfn = istft_fft_kernel<NN / 2, RA, RB, INV_FR, FAST>;                                                          \
    NOISE_ALLOW_LDS(kfn, smem);                                                                                         \
    hipLaunchKernelGGL(kfn, grid, dim3(16 * INV_FR), smem, st, logmag, phase, syn_window, (const c32*)tw1, (const c32*)tw2, \
                       window_sum, vad_logit, wave_out, d);                                                             \
  }
  INV_CASE(320, 16, 10, false)
  INV_CASE(512, 16, 16, false)
  INV_CASE(320, 16, 10, true)
  INV_CASE(512, 16, 16, true)
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
  FftDims d{B, 0, T, hop, n_fft / 2 + 1, hop * (T - 1), accumulate, 0};
  launch_fwd<1, false>(n_fft, d_wave, syn_window, tw1, tw2, d_logmag, nullptr, nullptr, wave_out, window_sum, logmag, phase, d,
                (hipStream_t)stream);
  NOISE_CHECK_LAUNCH("noise_istft_bwd_fft");
  return 0;
}
