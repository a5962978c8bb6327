// Per-utterance SNR mixing and activity detection in float64, one workgroup per utterance.
//
// Replaces process/audiolib.py:176-216 (segmental_snr_mixer), :219-248 (active_rms; its
// threshold of -50 on a linear RMS passes every window, so it is the RMS of the whole signal),
// :30-35 (normalize_segmental_rms), :18-19 (is_clipped) and :251-289 (activitydetector).
//
// Two passes over an utterance pair: (1) max|c|, max|n|, sum c^2, sum n^2, sum c*n;
// (2) write clean/noise/noisy at the drawn level while tracking max|noisy|.  Only when the
// mix clips (rare) a third pass rescales the three outputs.  No atomics: all reductions are
// inside the workgroup, so results are reproducible run to run.
#include "common.h"

namespace {

constexpr int MIX_THREADS = 1024;
#define MIX_EPS 2.220446049250313e-16  // numpy.finfo(float).eps

__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

__device__ __forceinline__ double block_max(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmax(t, red[i]);
  return t;
}

__global__ __launch_bounds__(MIX_THREADS) void snr_mix_kernel(
    const double* __restrict__ clean, const double* __restrict__ noise, const int* __restrict__ snr,
    const int* __restrict__ level_draw, double* __restrict__ clean_out, double* __restrict__ noise_out,
    double* __restrict__ noisy_out, int* __restrict__ level_out, double* __restrict__ peaks_out,
    float* __restrict__ noisy_f32, long L, double target_level, double clipping_threshold) {
  __shared__ double red[MIX_THREADS / 64];
  const long u = blockIdx.x;
  const double* c = clean + u * L;
  const double* n = noise + u * L;
  const int tid = threadIdx.x;

  // ---- pass 1: one read of both signals
  double mc = 0, mn = 0, scc = 0, snn = 0, scn_ = 0;
  for (long i = tid; i < L; i += MIX_THREADS) {
    const double a = c[i], b = n[i];
    mc = fmax(mc, fabs(a));
    mn = fmax(mn, fabs(b));
    scc += a * a;
    snn += b * b;
    scn_ += a * b;
  }
  mc = block_max(mc, red);
  mn = block_max(mn, red);
  scc = block_sum(scc, red);
  snn = block_sum(snn, red);
  scn_ = block_sum(scn_, red);
  const double kc = mc + MIX_EPS, kn = mn + MIX_EPS;
  // RMS of the PEAK-NORMALISED signals (sum (x/k)^2 == sum x^2 / k^2)
  const double rmsclean = sqrt(scc / (kc * kc) / (double)L);
  const double rmsnoise = sqrt(snn / (kn * kn) / (double)L);
  const double tl = pow(10.0, target_level / 20.0);
  const double sc = tl / (rmsclean + MIX_EPS);
  const double sn = tl / (rmsnoise + MIX_EPS);
  const double noisescalar = rmsclean / pow(10.0, (double)snr[u] / 20.0) / (rmsnoise + MIX_EPS);
  // mix = alpha*c + beta*n, so its mean square follows from the three sums above
  const double alpha = sc / kc, beta = sn * noisescalar / kn;
  const double syy = alpha * alpha * scc + 2.0 * alpha * beta * scn_ + beta * beta * snn;
  const double rmsnoisy = sqrt(fmax(syy, 0.0) / (double)L);
  int level = level_draw[u];
  const double scn = pow(10.0, (double)level / 20.0) / (rmsnoisy + MIX_EPS);

  // ---- pass 2: write at the drawn level (same operation order as the reference), track the peak
  // The peaks of the three outputs (what is_clipped, audiolib.py:18-19, looks at afterwards) and a float32 copy of the
  // mix (the STFT features' input) come out of the same pass; both are optional.
  double my = 0, pc = 0, pn = 0, sy = 0;
  for (long i = tid; i < L; i += MIX_THREADS) {
    const double a = (c[i] / kc) * sc;
    const double b = ((n[i] / kn) * sn) * noisescalar;
    const double y = (a + b) * scn;
    const double ao = a * scn, bo = b * scn;
    clean_out[u * L + i] = ao;
    noise_out[u * L + i] = bo;
    noisy_out[u * L + i] = y;
    if (noisy_f32) noisy_f32[u * L + i] = (float)y;   // rewritten by pass 3 in the rare clipping case
    my = fmax(my, fabs(y));
    pc = fmax(pc, fabs(ao));
    pn = fmax(pn, fabs(bo));
    sy += y;
  }
  my = block_max(my, red);
  if (peaks_out) {
    pc = block_max(pc, red);
    pn = block_max(pn, red);
    sy = block_sum(sy, red);
  }

  // ---- pass 3 (only if the mix clips): renormalise all three signals by the peak
  const bool clips = my > clipping_threshold;
  const double peak = clips ? my / (clipping_threshold - MIX_EPS) : 1.0;
  if (clips) {
    sy = 0;
    for (long i = tid; i < L; i += MIX_THREADS) {
      clean_out[u * L + i] = clean_out[u * L + i] / peak;
      noise_out[u * L + i] = noise_out[u * L + i] / peak;
      const double y = noisy_out[u * L + i] / peak;
      noisy_out[u * L + i] = y;
      sy += y;
      if (noisy_f32) noisy_f32[u * L + i] = (float)y;
    }
    if (peaks_out) sy = block_sum(sy, red);
    level = (int)(20.0 * log10(scn / peak * (rmsnoisy + MIX_EPS)));
  }
  if (tid == 0) {
    level_out[u] = level;
    if (peaks_out) {   // division by the peak is monotonic: the largest element stays the largest
      peaks_out[4 * u + 0] = clips ? pc / peak : pc;
      peaks_out[4 * u + 1] = clips ? pn / peak : pn;
      peaks_out[4 * u + 2] = clips ? my / peak : my;
      peaks_out[4 * u + 3] = sy;   // sum of the mix: a checksum of the utterance (fixed summation order)
    }
  }
}

constexpr int ACT_THREADS = 256;

__global__ __launch_bounds__(ACT_THREADS) void activity_kernel(const double* __restrict__ audio,
                                                               double* __restrict__ perc_active, long L,
                                                               int win, double energy_thresh,
                                                               double target_level) {
  extern __shared__ double prob[];  // one energy probability per 50 ms frame
  __shared__ double red[ACT_THREADS / 64];
  __shared__ int cnt_red[ACT_THREADS / 64];
  const long u = blockIdx.x;
  const double* x = audio + u * L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0;
  for (long i = tid; i < L; i += ACT_THREADS) s += x[i] * x[i];
  s = block_sum(s, red);
  const double scalar = pow(10.0, target_level / 20.0) / (sqrt(s / (double)L) + MIX_EPS);
  const int nfr = (int)((L + win - 1) / win);
  for (int f = wave; f < nfr; f += ACT_THREADS / 64) {
    const long lo = (long)f * win, hi = min(lo + (long)win, L);
    double e = 0;
    for (long i = lo + lane; i < hi; i += 64) {
      const double v = x[i] * scalar;
      e += v * v;
    }
    e = wave_reduce_sum(e);
    if (lane == 0) {
      const double frame_rms = 20.0 * log10(e + MIX_EPS);
      prob[f] = 1.0 / (1.0 + exp(-(-1.0 + 0.2 * frame_rms)));
    }
  }
  __syncthreads();
  int active = 0;
  for (int f = tid; f < nfr; f += ACT_THREADS) {
    const double p = prob[f], prev = f > 0 ? prob[f - 1] : 0.0;  // the previous frame's UNsmoothed value
    const double sm = p > prev ? p * 0.8 + prev * (1 - 0.8) : p * 0.05 + prev * (1 - 0.05);
    active += sm > energy_thresh;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) active += __shfl_xor(active, o, 64);
  if (lane == 0) cnt_red[wave] = active;
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int i = 0; i < ACT_THREADS / 64; ++i) t += cnt_red[i];
    perc_active[u] = (double)t / (double)nfr;
  }
}

}  // namespace

// peaks_out (optional): [n_utt][4] = max |clean_out|, max |noise_out|, max |noisy_out|, sum noisy_out per utterance;
// noisy_f32 (optional): [n_utt][L] float32 copy of noisy_out.
NOISE_EXPORT int noise_snr_mix_peaks(const double* clean, const double* noise, const int* snr, const int* level_draw,
                                     double* clean_out, double* noise_out, double* noisy_out, int* level_out,
                                     double* peaks_out, float* noisy_f32, int n_utt, long L, double target_level,
                                     double clipping_threshold, void* stream) {
  NOISE_REQUIRE(n_utt >= 0 && L >= 0, "noise_snr_mix: negative size");
  if (n_utt == 0) return 0;
  NOISE_REQUIRE(L > 0, "noise_snr_mix: empty utterance");
  NOISE_REQUIRE(clean && noise && snr && level_draw && clean_out && noise_out && noisy_out && level_out,
                "noise_snr_mix: null pointer");
  hipLaunchKernelGGL(snr_mix_kernel, dim3(n_utt), dim3(MIX_THREADS), 0, (hipStream_t)stream, clean, noise, snr,
                     level_draw, clean_out, noise_out, noisy_out, level_out, peaks_out, noisy_f32, L, target_level,
                     clipping_threshold);
  NOISE_CHECK_LAUNCH("noise_snr_mix");
  return 0;
}

NOISE_EXPORT int noise_snr_mix(const double* clean, const double* noise, const int* snr, const int* level_draw,
                               double* clean_out, double* noise_out, double* noisy_out, int* level_out,
                               int n_utt, long L, double target_level, double clipping_threshold, void* stream) {
  return noise_snr_mix_peaks(clean, noise, snr, level_draw, clean_out, noise_out, noisy_out, level_out, nullptr, nullptr,
                             n_utt, L, target_level, clipping_threshold, stream);
}

NOISE_EXPORT int noise_activity(const double* audio, double* perc_active, int n_utt, long L, int fs,
                                double energy_thresh, double target_level, void* stream) {
  NOISE_REQUIRE(n_utt >= 0 && L >= 0, "noise_activity: negative size");
  if (n_utt == 0) return 0;
  NOISE_REQUIRE(L > 0 && fs > 0, "noise_activity: empty utterance");
  NOISE_REQUIRE(audio && perc_active, "noise_activity: null pointer");
  const int win = (int)(fs * 50 / 1000);
  const long nfr = (L + win - 1) / win;
  NOISE_REQUIRE(nfr * sizeof(double) <= 60000, "noise_activity: utterance too long (%ld frames)", nfr);
  hipLaunchKernelGGL(activity_kernel, dim3(n_utt), dim3(ACT_THREADS), nfr * sizeof(double), (hipStream_t)stream,
                     audio, perc_active, L, win, energy_thresh, target_level);
  NOISE_CHECK_LAUNCH("noise_activity");
  return 0;
}
