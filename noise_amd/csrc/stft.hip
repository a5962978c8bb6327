// STFT / iSTFT for MI355X (gfx950): windowed real DFT and its inverse as fp32-MFMA tile products.
//
// Replaces train/models.py:460-475 (STFT.forward: reflect pad + strided conv1d with the windowed
// DFT basis + log-magnitude / atan2) and :477-495 (STFT.inverse: exp/cos/sin recombination,
// conv_transpose1d overlap-add, window-envelope scaling, crop, clamp).  The same frame-product
// kernel, run with the inverse basis, is the adjoint needed for the iSTFT backward pass.
//
// Layout: waveforms [B][L]; spectra [B][F][T] (T innermost, as the reference's (B,1,F,T)).
// One workgroup = 64 frames of one clip.  The clip samples the tile touches are staged once in
// LDS (coalesced, reflect-padded by index mirroring); frames are read from that staging with a
// row pitch of hop+1 so the 32 frames of an MFMA column tile hit 32 different banks.
// v_mfma_f32_32x32x2_f32 keeps the arithmetic exact fp32 (a k-ordered fmaf chain).
#include "common.h"

namespace {

struct StftDims {
  int B, L, T, n_fft, hop, F, Rpad, Lout, accum;
  int nFT;             // 32-row tiles covering n_fft/2 + 1 (frequency bins, or samples of a folded frame)
  unsigned hop_magic;  // floor(k / hop) == __umulhi(k, hop_magic) for k <= n_fft
};

constexpr int STFT_TF = 64;  // frames per workgroup

// Runs mma(step 0), mma(step 1), ... in order, with the operands of step j+1 (basis values from L2, samples from LDS)
// requested before the MFMAs of step j: two operand sets in turn, and scheduling barriers so that the compiler does not
// rotate the loop back into load -> wait -> MFMA.  fetch(j, ops) must accept a repeated last step.
template <class Ops, class Fetch, class Mma>
__device__ __forceinline__ void pipelined_steps(int steps, Fetch fetch, Mma mma) {
  Ops X, Y;
  fetch(0, X);
  int j = 0;
  for (; j + 1 < steps; j += 2) {
    fetch(j + 1, Y);
    __builtin_amdgcn_sched_barrier(0);
    mma(X);
    __builtin_amdgcn_sched_barrier(0);
    fetch(min(j + 2, steps - 1), X);
    __builtin_amdgcn_sched_barrier(0);
    mma(Y);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (j < steps) mma(X);
}

template <int NT>
struct DenseOps {   // one k-step of the dense inverse products: two spectrum values and two basis values per row tile
  float s[2];
  float a[NT][2];
};

// MODE 0: forward  -> log-magnitude, phase, per-clip max|log-magnitude|
// MODE 1: adjoint of the iSTFT: sig = dL/d(wave_out); writes dL/d(log-magnitude)
template <int MODE, int NT>
__global__ __launch_bounds__(256, (NT <= 3 ? 4 : 1)) void stft_frames_kernel(
    const float* __restrict__ sig, const float* __restrict__ basisT, float* __restrict__ out0,
    float* __restrict__ out1, unsigned* __restrict__ absmax_bits, const float* __restrict__ wave_out,
    const float* __restrict__ wsum, const float* __restrict__ logmag, const float* __restrict__ phase,
    StftDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STFT_TF;
  const int hop = d.hop, n_fft = d.n_fft, pitch = hop + 1;
  const int R = n_fft / hop;
  // a tile of a short clip holds fewer than STFT_TF frames: only their samples are staged (the launcher sizes the LDS for
  // the largest tile, so 0.5 s segments fit four workgroups per CU); the MFMA columns of the missing frames repeat the
  // last staged frame and are dropped in the epilogue
  const int nfr = min(STFT_TF, d.T - t0);
  const int nS = (nfr - 1) * hop + n_fft;
  const int half = n_fft / 2;

  // ---- stage the samples of this tile (index in the padded signal = t0*hop + i)
  for (int i = tid; i < nS; i += 256) {
    int s = t0 * hop + i - half;
    float v;
    if (MODE == 0) {
      if (s < 0) s = -s;
      if (s >= d.L) s = 2 * (d.L - 1) - s;
      s = min(max(s, 0), d.L - 1);  // only reached by frames past T (never stored)
      v = sig[(size_t)b * d.L + s];
    } else {
      v = 0.f;
      if (s >= 0 && s < d.Lout) {
        size_t o = (size_t)b * d.Lout + s;
        float w = wave_out[o];
        v = (w > -1.f && w < 1.f) ? sig[o] * wsum[s + half] : 0.f;
      }
    }
    lds[i + i / hop] = v;
  }
  __syncthreads();

  const int nRT = d.Rpad >> 5;
  f32x16 acc[NT][2];
#pragma unroll
  for (int r = 0; r < NT; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;

  const int nc0 = min(n, nfr - 1), nc1 = min(n + 32, nfr - 1);
  const float* bas = basisT + n;
  int rtc[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r) rtc[r] = 32 * (wave + 4 * r < nRT ? wave + 4 * r : nRT - 1);
  // k runs in steps of two rows (kk = 2s + h) in the order of the reference's conv1d; the operands of step s+1 (basis
  // values from L2, samples from LDS) are requested before the MFMAs of step s, so their latency overlaps 2*NT MFMAs
  // of this wave and those of the other resident waves instead of stalling every step
  const int steps = n_fft >> 1;
  const float* ap = bas + (size_t)h * d.Rpad;
  const size_t astep = (size_t)2 * d.Rpad;
  int l0 = nc0 * pitch + h, l1 = nc1 * pitch + h;   // sample kk of the lane's two frames: +2 per step, +1 per hop crossed
  int into = 0;
  (void)R;
  auto advance = [&]() {
    ap += astep;
    into += 2;
    l0 += 2;
    l1 += 2;
    if (into == hop) {
      into = 0;
      ++l0;
      ++l1;
    }
  };
  // branch-free over r: a wave without an r-th row tile recomputes its last one (discarded in the epilogue)
  auto fetch = [&](float (&a)[NT], float& b0, float& b1) {
#pragma unroll
    for (int r = 0; r < NT; ++r) a[r] = ap[rtc[r]];
    b0 = lds[l0];
    b1 = lds[l1];
  };
  auto mma = [&](const float (&a)[NT], float b0, float b1) {
#pragma unroll
    for (int r = 0; r < NT; ++r) {
      acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b0, acc[r][0], 0, 0, 0);
      acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b1, acc[r][1], 0, 0, 0);
    }
  };
  // two register sets in turn (steps is even: n_fft is a multiple of 32); the scheduling barriers keep each request
  // ahead of the MFMAs it overlaps
  float aX[NT], aY[NT], bX0, bX1, bY0, bY1;
  fetch(aX, bX0, bX1);
  for (int s = 0; s < steps; s += 2) {
    advance();
    fetch(aY, bY0, bY1);
    __builtin_amdgcn_sched_barrier(0);
    mma(aX, bX0, bX1);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 2 < steps) advance();   // the last step requests itself again
    fetch(aX, bX0, bX1);
    __builtin_amdgcn_sched_barrier(0);
    mma(aY, bY0, bY1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: rows are interleaved (2f = Re, 2f+1 = Im); a lane holds rows 8g+4h..+3
  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    const int rt = wave + 4 * r;
    if (rt >= nRT) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int t = t0 + 32 * c + n;
      if (t >= d.T) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int f = (rt * 32 + 8 * g + 4 * h) / 2 + q;
          if (f >= d.F) continue;
          const float re = acc[r][c][4 * g + 2 * q], im = acc[r][c][4 * g + 2 * q + 1];
          const size_t o = ((size_t)b * d.F + f) * d.T + t;
          if (MODE == 0) {
            const float lm = logf(fmaxf(sqrtf(re * re + im * im), 1e-3f));
            out0[o] = lm;
            if (out1 != nullptr) out1[o] = atan2f(im, re);
            amax = fmaxf(amax, fabsf(lm));
          } else {
            float sn, cs;
            sincosf(phase[o], &sn, &cs);
            const float gv = expf(logmag[o]) * (cs * re + sn * im);
            out0[o] = d.accum ? out0[o] + gv : gv;
          }
        }
      }
    }
  }
  if (MODE == 0 && absmax_bits != nullptr) {
    amax = wave_reduce_max(amax);
    if (lane == 0) atomicMax(absmax_bits + b, __float_as_uint(amax));
  }
}

constexpr int ISTFT_TF = 32;  // frames per workgroup (incl. the R-1 halo frames)

template <int NT>
__global__ __launch_bounds__(256) void istft_kernel(const float* __restrict__ logmag,
                                                    const float* __restrict__ phase,
                                                    const float* __restrict__ inv_basis,
                                                    const float* __restrict__ wsum,
                                                    const float* __restrict__ vad_logit,
                                                    float* __restrict__ out, StftDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int hop = d.hop, n_fft = d.n_fft, F = d.F;
  const int R = n_fft / hop;
  const int TFo = ISTFT_TF - (R - 1);
  const int q0 = blockIdx.x * TFo;        // first output hop of this tile
  const int tstart = q0 - (R - 1);        // first frame of this tile

  // ---- stage z = [mag*cos ; mag*sin] for the tile's 32 frames: zs[(2f+part)*32 + tl]
  for (int i = tid; i < F * 32; i += 256) {
    const int f = i >> 5, tl = i & 31, t = tstart + tl;
    float zr = 0.f, zi = 0.f;
    if (t >= 0 && t < d.T) {
      const size_t o = ((size_t)b * F + f) * d.T + t;
      const float m = expf(logmag[o]);
      float sn, cs;
      sincosf(phase[o], &sn, &cs);
      zr = m * cs;
      zi = m * sn;
    }
    lds[(2 * f) * 32 + tl] = zr;
    lds[(2 * f + 1) * 32 + tl] = zi;
  }
  __syncthreads();

  const int nNT = n_fft >> 5;
  f32x16 acc[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

  const float* bas = inv_basis + (size_t)h * F * n_fft + n;
  int ntc[NT];   // branch-free row tiles, see stft_frames_kernel
#pragma unroll
  for (int r = 0; r < NT; ++r) ntc[r] = 32 * (wave + 4 * r < nNT ? wave + 4 * r : nNT - 1);
  pipelined_steps<DenseOps<NT>>(
      F,
      [&](int f, DenseOps<NT>& o) {
        o.s[0] = lds[(2 * f + h) * 32 + n];
        const float* arow = bas + (size_t)f * n_fft;
#pragma unroll
        for (int r = 0; r < NT; ++r) o.a[r][0] = arow[ntc[r]];
      },
      [&](const DenseOps<NT>& o) {
#pragma unroll
        for (int r = 0; r < NT; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[r][0], o.s[0], acc[r], 0, 0, 0);
      });
  __syncthreads();

  // ---- time-domain frames to LDS: ft[tl][sample], pitch n_fft+1
  const int fp = n_fft + 1;
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    const int nt = wave + 4 * r;
    if (nt >= nNT) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      lds[n * fp + row] = acc[r][e];
    }
  }
  __syncthreads();

  // ---- atomic-free overlap-add: every output sample gathers its R frames
  float gate = 1.f;
  if (vad_logit != nullptr) gate = 1.f / (1.f + expf(-1e4f * vad_logit[b]));
  const int half = n_fft / 2;
  for (int i = tid; i < TFo * hop; i += 256) {
    const int jj = i / hop, m = i - jj * hop;
    const int s = (q0 + jj) * hop + m;
    const int so = s - half;
    if (so < 0 || so >= d.Lout) continue;
    float y = 0.f;
    for (int r = 0; r < R; ++r) y += lds[(jj + R - 1 - r) * fp + r * hop + m];
    y *= wsum[s];
    y = fminf(fmaxf(y, -1.f), 1.f) * gate;
    out[(size_t)b * d.Lout + so] = y;
  }
}


// ---- symmetric-window variants ---------------------------------------------------------------------------
// For a window with w[n] = w[n_fft - n] (every periodic window, centred-padded or not) the Re rows of the
// windowed DFT basis are even and the Im rows odd about n_fft/2:
//   Re X[f] = sum_{k=0}^{N/2} bRe[k][f] * (x[k] + x[N-k]),   Im X[f] = sum_k bIm[k][f] * (x[k] - x[N-k])
// (k = 0 and k = N/2 unpaired), so the product needs N/2 + 1 basis rows instead of N: half the MFMA work.
// basisF [N/2 + 1][2*FP]: column f = Re f, column FP + f = Im f (FP = 32 * nFT).  A wave owns NT frequency
// tiles for both column tiles of the 64 frames, with Re and Im accumulators side by side.
constexpr int STFT_SYM_MAX_WAVES = 9;

template <int MODE, int NT>
__global__ __launch_bounds__(64 * STFT_SYM_MAX_WAVES) void stft_frames_sym_kernel(
    const float* __restrict__ sig, const float* __restrict__ basisF, float* __restrict__ out0,
    float* __restrict__ out1, unsigned* __restrict__ absmax_bits, const float* __restrict__ wave_out,
    const float* __restrict__ wsum, const float* __restrict__ logmag, const float* __restrict__ phase,
    StftDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STFT_TF;
  const int hop = d.hop, n_fft = d.n_fft, pitch = hop + 1;
  const int nS = (STFT_TF - 1) * hop + n_fft;
  const int half = n_fft / 2;

  // four samples per thread and trip: the loads of a trip are independent and issued together (clamped addresses,
  // select afterwards), so staging costs nS / (4 * threads) memory round trips
  for (int i0 = tid; i0 < nS; i0 += 4 * nthr) {
    float v[4], w[4], e[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthr;
      int s = t0 * hop + i - half;
      if (MODE == 0) {
        if (s < 0) s = -s;
        if (s >= d.L) s = 2 * (d.L - 1) - s;
        s = min(max(s, 0), d.L - 1);
        v[u] = sig[(size_t)b * d.L + s];
      } else {
        in[u] = s >= 0 && s < d.Lout;
        const int sc = in[u] ? s : 0;
        const size_t o = (size_t)b * d.Lout + sc;
        w[u] = wave_out[o];
        v[u] = sig[o];
        e[u] = wsum[sc + half];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthr;
      float x = v[u];
      if (MODE != 0) x = (in[u] && w[u] > -1.f && w[u] < 1.f) ? x * e[u] : 0.f;
      if (i < nS) lds[i + (int)__umulhi((unsigned)i, d.hop_magic)] = x;
    }
  }
  __syncthreads();

  f32x16 acc[NT][2][2];  // [frequency tile][Re, Im][column tile]
#pragma unroll
  for (int r = 0; r < NT; ++r)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][p][c][e] = 0.f;

  const int FP = 32 * d.nFT;
  int ftc[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r) ftc[r] = 32 * min(wave * NT + r, d.nFT - 1);
  const float* bas = basisF + n;
  const int fa = n * pitch, fb = (n + 32) * pitch;
  const int steps = half / 2 + 1;  // k = 2j + h runs over 0 .. N/2 (+1: a padding step for odd counts)
#pragma unroll 4
  for (int j = 0; j < steps; ++j) {
    const int kp = 2 * j + h;
    const bool valid = kp <= half, paired = kp > 0 && kp < half;
    const int k1 = min(kp, half), k2 = n_fft - k1;
    const int i1 = k1 + (int)__umulhi((unsigned)k1, d.hop_magic);
    const int i2 = k2 + (int)__umulhi((unsigned)k2, d.hop_magic);
    float va = lds[fa + i1], vb = lds[fb + i1];
    float wa = lds[fa + i2], wb = lds[fb + i2];
    va = valid ? va : 0.f;
    vb = valid ? vb : 0.f;
    wa = paired ? wa : 0.f;
    wb = paired ? wb : 0.f;
    const float sa = va + wa, sb = vb + wb, da = va - wa, db = vb - wb;
    const float* arow = bas + (size_t)k1 * (2 * FP);
#pragma unroll
    for (int r = 0; r < NT; ++r) {
      const float are = arow[ftc[r]], aim = arow[FP + ftc[r]];
      acc[r][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(are, sa, acc[r][0][0], 0, 0, 0);
      acc[r][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(are, sb, acc[r][0][1], 0, 0, 0);
      acc[r][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, da, acc[r][1][0], 0, 0, 0);
      acc[r][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aim, db, acc[r][1][1], 0, 0, 0);
    }
  }

  float amax = 0.f;
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    const int ft = wave * NT + r;
    if (ft >= d.nFT) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int t = t0 + 32 * c + n;
      const bool tin = t < d.T;
      const int tc = tin ? t : d.T - 1;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = 32 * ft + (e & 3) + 8 * (e >> 2) + 4 * h;
        const bool in = tin && f < d.F;
        const float re = acc[r][0][c][e], im = acc[r][1][c][e];
        // clamped address: the loads of the gradient path need no branch, only the stores are conditional
        const size_t o = ((size_t)b * d.F + (f < d.F ? f : d.F - 1)) * d.T + tc;
        if (MODE == 0) {
          const float lm = logf(fmaxf(sqrtf(re * re + im * im), 1e-3f));
          if (in) {
            out0[o] = lm;
            if (out1 != nullptr) out1[o] = atan2f(im, re);
            amax = fmaxf(amax, fabsf(lm));
          }
        } else {
          float sn, cs;
          sincosf(phase[o], &sn, &cs);
          const float gv = expf(logmag[o]) * (cs * re + sn * im);
          const float prev = d.accum ? out0[o] : 0.f;
          if (in) out0[o] = prev + gv;
        }
      }
    }
  }
  if (MODE == 0 && absmax_bits != nullptr) {
    amax = wave_reduce_max(amax);
    if (lane == 0) atomicMax(absmax_bits + b, __float_as_uint(amax));
  }
}

// Inverse with the same symmetry: a frame is x[p] = E[p] + O[p], x[N-p] = E[p] - O[p] for p <= N/2 with
// E = sum_f zr[f]*ibRe[f][p], O = sum_f zi[f]*ibIm[f][p]; only columns 0..N/2 of inv_basis are read.
// A wave owns NT sample tiles of E and of O for the 32 frames of the workgroup.
template <int NT>
__global__ __launch_bounds__(64 * STFT_SYM_MAX_WAVES) void istft_sym_kernel(
    const float* __restrict__ logmag, const float* __restrict__ phase, const float* __restrict__ inv_basis,
    const float* __restrict__ wsum, const float* __restrict__ vad_logit, float* __restrict__ out, StftDims d) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
  const int n = lane & 31, h = lane >> 5;
  const int b = blockIdx.y;
  const int hop = d.hop, n_fft = d.n_fft, F = d.F;
  const int R = n_fft / hop;
  const int TFo = ISTFT_TF - (R - 1);
  const int q0 = blockIdx.x * TFo;
  const int tstart = q0 - (R - 1);

  // z rows 2f (mag*cos), 2f+1 (mag*sin); row pair F is zero (odd F: the k-loop pairs bins two by two)
#pragma unroll 2
  for (int i = tid; i < (F + 1) * 32; i += nthr) {   // clamped addresses, zero selected afterwards: no branch around loads
    const int f = i >> 5, tl = i & 31, t = tstart + tl;
    const bool in = f < F && t >= 0 && t < d.T;
    const size_t o = ((size_t)b * F + (f < F ? f : F - 1)) * d.T + (t < 0 ? 0 : (t < d.T ? t : d.T - 1));
    const float m = expf(logmag[o]);
    float sn, cs;
    sincosf(phase[o], &sn, &cs);
    lds[(2 * f) * 32 + tl] = in ? m * cs : 0.f;
    lds[(2 * f + 1) * 32 + tl] = in ? m * sn : 0.f;
  }
  __syncthreads();

  f32x16 acc[NT][2];
#pragma unroll
  for (int r = 0; r < NT; ++r)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[r][p][e] = 0.f;
  int ntc[NT];
#pragma unroll
  for (int r = 0; r < NT; ++r) ntc[r] = 32 * min(wave * NT + r, d.nFT - 1);
  const float* bre = inv_basis + n;
  const float* bim = inv_basis + (size_t)F * n_fft + n;
  const int steps = (F + 1) / 2;
  pipelined_steps<DenseOps<NT>>(
      steps,
      [&](int j, DenseOps<NT>& o) {
        const int f = 2 * j + h;
        const size_t fo = (size_t)min(f, F - 1) * n_fft;
        o.s[0] = lds[(2 * f) * 32 + n];
        o.s[1] = lds[(2 * f + 1) * 32 + n];
#pragma unroll
        for (int r = 0; r < NT; ++r) {
          o.a[r][0] = bre[fo + ntc[r]];
          o.a[r][1] = bim[fo + ntc[r]];
        }
      },
      [&](const DenseOps<NT>& o) {
#pragma unroll
        for (int r = 0; r < NT; ++r) {
          acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[r][0], o.s[0], acc[r][0], 0, 0, 0);
          acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[r][1], o.s[1], acc[r][1], 0, 0, 0);
        }
      });
  __syncthreads();

  // E to lds[frame][p], O to lds[32 + frame][p], pitch EP (odd)
  const int EP = 32 * d.nFT + 1;
#pragma unroll
  for (int r = 0; r < NT; ++r) {
    const int nt = wave * NT + r;
    if (nt >= d.nFT) continue;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) lds[(32 * p + n) * EP + nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[r][p][e];
  }
  __syncthreads();

  float gate = 1.f;
  if (vad_logit != nullptr) gate = 1.f / (1.f + expf(-1e4f * vad_logit[b]));
  const int half = n_fft / 2;
  for (int i = tid; i < TFo * hop; i += nthr) {
    const int jj = i / hop, m = i - jj * hop;
    const int s = (q0 + jj) * hop + m;
    const int so = s - half;
    if (so < 0 || so >= d.Lout) continue;
    float y = 0.f;
    for (int r = 0; r < R; ++r) {
      const int p = r * hop + m, fr = jj + R - 1 - r;
      const bool lower = p <= half;
      const int pf = lower ? p : n_fft - p;
      const float ev = lds[fr * EP + pf], ov = lds[(32 + fr) * EP + pf];
      y += lower ? ev + ov : ev - ov;
    }
    y *= wsum[s];
    y = fminf(fmaxf(y, -1.f), 1.f) * gate;
    out[(size_t)b * d.Lout + so] = y;
  }
}

int check_dims(const char* who, int B, int L, int n_fft, int hop) {
  NOISE_REQUIRE(B > 0 && L > 0, "%s: empty batch (B=%d, L=%d)", who, B, L);
  NOISE_REQUIRE(n_fft >= 64 && n_fft % 32 == 0 && n_fft <= 1024, "%s: n_fft=%d unsupported (multiple of 32 in [64,1024])", who, n_fft);
  NOISE_REQUIRE(hop > 0 && hop % 2 == 0 && n_fft % hop == 0, "%s: hop=%d must be even and divide n_fft=%d", who, hop, n_fft);
  NOISE_REQUIRE(L > n_fft / 2, "%s: L=%d too short for reflect padding of %d", who, L, n_fft / 2);
  return 0;
}

template <int MODE>
int launch_frames(const float* sig, const float* basisT, float* out0, float* out1, unsigned* absmax,
                  const float* wave_out, const float* wsum, const float* logmag, const float* phase,
                  StftDims d, hipStream_t st) {
  const int nRT = d.Rpad / 32;
  const int per_wave = (nRT + 3) / 4;
  const int nS = ((d.T < STFT_TF ? d.T : STFT_TF) - 1) * d.hop + d.n_fft;   // samples of the largest tile
  const size_t smem = (size_t)(nS + nS / d.hop + 1) * sizeof(float);
  dim3 grid(noise_cdiv(d.T, STFT_TF), d.B);
  if (per_wave <= 3) {
    NOISE_ALLOW_LDS((stft_frames_kernel<MODE, 3>), smem);
    hipLaunchKernelGGL((stft_frames_kernel<MODE, 3>), grid, dim3(256), smem, st, sig, basisT, out0, out1, absmax,
                       wave_out, wsum, logmag, phase, d);
  } else if (per_wave <= 5) {
    NOISE_ALLOW_LDS((stft_frames_kernel<MODE, 5>), smem);
    hipLaunchKernelGGL((stft_frames_kernel<MODE, 5>), grid, dim3(256), smem, st, sig, basisT, out0, out1, absmax,
                       wave_out, wsum, logmag, phase, d);
  } else {
    NOISE_ALLOW_LDS((stft_frames_kernel<MODE, 9>), smem);
    hipLaunchKernelGGL((stft_frames_kernel<MODE, 9>), grid, dim3(256), smem, st, sig, basisT, out0, out1, absmax,
                       wave_out, wsum, logmag, phase, d);
  }
  return 0;
}

void sym_dims(StftDims& d) {
  d.nFT = (d.n_fft / 2 + 1 + 31) / 32;
  d.hop_magic = (unsigned)(0x100000000ull / (unsigned)d.hop) + 1u;  // exact quotient for k*hop < 2^32
}

template <int MODE>
int launch_frames_sym(const float* sig, const float* basisF, float* out0, float* out1, unsigned* absmax,
                      const float* wave_out, const float* wsum, const float* logmag, const float* phase,
                      StftDims d, hipStream_t st) {
  sym_dims(d);
  const int nS = (STFT_TF - 1) * d.hop + d.n_fft;
  const size_t smem = (size_t)(nS + nS / d.hop + 1) * sizeof(float);
  dim3 grid(noise_cdiv(d.T, STFT_TF), d.B);
  if (d.nFT <= STFT_SYM_MAX_WAVES) {
    NOISE_ALLOW_LDS((stft_frames_sym_kernel<MODE, 1>), smem);
    hipLaunchKernelGGL((stft_frames_sym_kernel<MODE, 1>), grid, dim3(64 * d.nFT), smem, st, sig, basisF, out0, out1,
                       absmax, wave_out, wsum, logmag, phase, d);
  } else {
    NOISE_ALLOW_LDS((stft_frames_sym_kernel<MODE, 2>), smem);
    hipLaunchKernelGGL((stft_frames_sym_kernel<MODE, 2>), grid, dim3(64 * noise_cdiv(d.nFT, 2)), smem, st, sig, basisF,
                       out0, out1, absmax, wave_out, wsum, logmag, phase, d);
  }
  return 0;
}

}  // namespace

NOISE_EXPORT int noise_stft_rpad(int n_fft) { return ((2 * (n_fft / 2 + 1) + 31) / 32) * 32; }

NOISE_EXPORT int noise_stft_fwd(const float* wave, const float* basisT, float* logmag, float* phase,
                                float* absmax, int B, int L, int n_fft, int hop, void* stream) {
  if (check_dims("noise_stft_fwd", B, L, n_fft, hop)) return -1;
  NOISE_REQUIRE(wave && basisT && logmag, "noise_stft_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  StftDims d{B, L, L / hop + 1, n_fft, hop, n_fft / 2 + 1, noise_stft_rpad(n_fft), hop * (L / hop), 0};
  if (absmax) {
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * B, st);
    NOISE_REQUIRE(e == hipSuccess, "noise_stft_fwd: memset failed: %s", hipGetErrorString(e));
  }
  launch_frames<0>(wave, basisT, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  NOISE_CHECK_LAUNCH("noise_stft_fwd");
  return 0;
}

NOISE_EXPORT int noise_istft_fwd(const float* logmag, const float* phase, const float* inv_basis,
                                 const float* window_sum, const float* vad_logit, float* wave_out, int B, int T,
                                 int n_fft, int hop, void* stream) {
  NOISE_REQUIRE(B > 0 && T > 0, "noise_istft_fwd: empty batch (B=%d, T=%d)", B, T);
  if (check_dims("noise_istft_fwd", B, n_fft, n_fft, hop)) return -1;
  NOISE_REQUIRE(logmag && phase && inv_basis && window_sum && wave_out, "noise_istft_fwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int F = n_fft / 2 + 1, R = n_fft / hop;
  NOISE_REQUIRE(R < ISTFT_TF, "noise_istft_fwd: n_fft/hop=%d too large", R);
  StftDims d{B, 0, T, n_fft, hop, F, 0, hop * (T - 1), 0};
  NOISE_REQUIRE(d.Lout > 0, "noise_istft_fwd: need at least 2 frames");
  const int TFo = ISTFT_TF - (R - 1);
  const int hops = T - 1 + R;
  dim3 grid(noise_cdiv(hops, TFo), B);
  const size_t smem = (size_t)(2 * F) * 32 * sizeof(float);
  const int per_wave = (n_fft / 32 + 3) / 4;
  if (per_wave <= 3) {
    NOISE_ALLOW_LDS((istft_kernel<3>), smem);
    hipLaunchKernelGGL((istft_kernel<3>), grid, dim3(256), smem, st, logmag, phase, inv_basis, window_sum, vad_logit, wave_out, d);
  } else if (per_wave <= 4) {
    NOISE_ALLOW_LDS((istft_kernel<4>), smem);
    hipLaunchKernelGGL((istft_kernel<4>), grid, dim3(256), smem, st, logmag, phase, inv_basis, window_sum, vad_logit, wave_out, d);
  } else {
    NOISE_ALLOW_LDS((istft_kernel<8>), smem);
    hipLaunchKernelGGL((istft_kernel<8>), grid, dim3(256), smem, st, logmag, phase, inv_basis, window_sum, vad_logit, wave_out, d);
  }
  NOISE_CHECK_LAUNCH("noise_istft_fwd");
  return 0;
}

// Adjoint of noise_istft_fwd w.r.t. the log-magnitude input (the phase receives no gradient:
// it is taken from the noisy input).  d_wave [B][hop*(T-1)], inv_basisT laid out like the
// forward basisT ([n_fft][Rpad], row 2f = Re f, row 2f+1 = Im f).  accumulate != 0 adds into d_logmag.
NOISE_EXPORT int noise_istft_bwd(const float* d_wave, const float* wave_out, const float* logmag,
                                 const float* phase, const float* inv_basisT, const float* window_sum,
                                 float* d_logmag, int accumulate, int B, int T, int n_fft, int hop, void* stream) {
  NOISE_REQUIRE(B > 0 && T > 1, "noise_istft_bwd: empty batch (B=%d, T=%d)", B, T);
  if (check_dims("noise_istft_bwd", B, n_fft, n_fft, hop)) return -1;
  NOISE_REQUIRE(d_wave && wave_out && logmag && phase && inv_basisT && window_sum && d_logmag,
                "noise_istft_bwd: null pointer");
  hipStream_t st = (hipStream_t)stream;
  StftDims d{B, 0, T, n_fft, hop, n_fft / 2 + 1, noise_stft_rpad(n_fft), hop * (T - 1), accumulate};
  launch_frames<1>(d_wave, inv_basisT, d_logmag, nullptr, nullptr, wave_out, window_sum, logmag, phase, d, st);
  NOISE_CHECK_LAUNCH("noise_istft_bwd");
  return 0;
}

// ---- symmetric-window entry points (half the arithmetic; same results up to summation order) -------------
NOISE_EXPORT int noise_stft_fold_cols(int n_fft) { return 2 * 32 * ((n_fft / 2 + 1 + 31) / 32); }

NOISE_EXPORT int noise_stft_fwd_sym(const float* wave, const float* basisF, float* logmag, float* phase,
                                    float* absmax, int B, int L, int n_fft, int hop, void* stream) {
  if (check_dims("noise_stft_fwd_sym", B, L, n_fft, hop)) return -1;
  NOISE_REQUIRE(wave && basisF && logmag, "noise_stft_fwd_sym: null pointer");
  hipStream_t st = (hipStream_t)stream;
  StftDims d{B, L, L / hop + 1, n_fft, hop, n_fft / 2 + 1, 0, hop * (L / hop), 0, 0, 0};
  if (absmax) {
    hipError_t e = hipMemsetAsync(absmax, 0, sizeof(float) * B, st);
    NOISE_REQUIRE(e == hipSuccess, "noise_stft_fwd_sym: memset failed: %s", hipGetErrorString(e));
  }
  launch_frames_sym<0>(wave, basisF, logmag, phase, (unsigned*)absmax, nullptr, nullptr, nullptr, nullptr, d, st);
  NOISE_CHECK_LAUNCH("noise_stft_fwd_sym");
  return 0;
}

NOISE_EXPORT int noise_istft_fwd_sym(const float* logmag, const float* phase, const float* inv_basis,
                                     const float* window_sum, const float* vad_logit, float* wave_out, int B, int T,
                                     int n_fft, int hop, void* stream) {
  NOISE_REQUIRE(B > 0 && T > 0, "noise_istft_fwd_sym: empty batch (B=%d, T=%d)", B, T);
  if (check_dims("noise_istft_fwd_sym", B, n_fft, n_fft, hop)) return -1;
  NOISE_REQUIRE(logmag && phase && inv_basis && window_sum && wave_out, "noise_istft_fwd_sym: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int F = n_fft / 2 + 1, R = n_fft / hop;
  NOISE_REQUIRE(R < ISTFT_TF, "noise_istft_fwd_sym: n_fft/hop=%d too large", R);
  StftDims d{B, 0, T, n_fft, hop, F, 0, hop * (T - 1), 0, 0, 0};
  sym_dims(d);
  NOISE_REQUIRE(d.Lout > 0, "noise_istft_fwd_sym: need at least 2 frames");
  const int TFo = ISTFT_TF - (R - 1);
  const int hops = T - 1 + R;
  dim3 grid(noise_cdiv(hops, TFo), B);
  const size_t zs = (size_t)(2 * F + 2) * 32, eo = (size_t)64 * (32 * d.nFT + 1);
  const size_t smem = (zs > eo ? zs : eo) * sizeof(float);
  if (d.nFT <= STFT_SYM_MAX_WAVES) {
    NOISE_ALLOW_LDS((istft_sym_kernel<1>), smem);
    hipLaunchKernelGGL((istft_sym_kernel<1>), grid, dim3(64 * d.nFT), smem, st, logmag, phase, inv_basis, window_sum,
                       vad_logit, wave_out, d);
  } else {
    NOISE_ALLOW_LDS((istft_sym_kernel<2>), smem);
    hipLaunchKernelGGL((istft_sym_kernel<2>), grid, dim3(64 * noise_cdiv(d.nFT, 2)), smem, st, logmag, phase, inv_basis,
                       window_sum, vad_logit, wave_out, d);
  }
  NOISE_CHECK_LAUNCH("noise_istft_fwd_sym");
  return 0;
}

NOISE_EXPORT int noise_istft_bwd_sym(const float* d_wave, const float* wave_out, const float* logmag,
                                     const float* phase, const float* inv_basisF, const float* window_sum,
                                     float* d_logmag, int accumulate, int B, int T, int n_fft, int hop, void* stream) {
  NOISE_REQUIRE(B > 0 && T > 1, "noise_istft_bwd_sym: empty batch (B=%d, T=%d)", B, T);
  if (check_dims("noise_istft_bwd_sym", B, n_fft, n_fft, hop)) return -1;
  NOISE_REQUIRE(d_wave && wave_out && logmag && phase && inv_basisF && window_sum && d_logmag,
                "noise_istft_bwd_sym: null pointer");
  hipStream_t st = (hipStream_t)stream;
  StftDims d{B, 0, T, n_fft, hop, n_fft / 2 + 1, 0, hop * (T - 1), accumulate, 0, 0};
  launch_frames_sym<1>(d_wave, inv_basisF, d_logmag, nullptr, nullptr, wave_out, window_sum, logmag, phase, d, st);
  NOISE_CHECK_LAUNCH("noise_istft_bwd_sym");
  return 0;
}
