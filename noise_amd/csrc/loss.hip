// Training losses of the spectral model and their gradients (train/models.py:776-790), gfx950.
//
//   spectral  = mean |enh - clean_logmag|                                  (:777)
//   vad       = mean BCE-with-logits(vad, [count(|clean| > 0) >= 160])      (:780-783)
//   waveform  = l1_loss((B,1,1,L), (B,L)) which BROADCASTS to (B,1,B,L):    (:786-787)
//               mean over all (i, j, t) of |enh_wave[i][t] - clean_wave[j][t]|
//   loss      = waveform + spectral + vad                                   (:790)
// All reductions are per-workgroup partials folded by one finalize kernel (no atomics, reproducible).
#include "common.h"
#include "lazy.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
  return t;
}

// partial[blockIdx] = sum |enh - clm|;  d_enh = sign(enh - clm) * scale
__global__ __launch_bounds__(256) void spectral_l1_kernel(const float* __restrict__ enh, const float* __restrict__ clm,
                                                          long n, float scale, float* __restrict__ d_enh,
                                                          double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = enh[i] - clm[i];
    s += (double)fabsf(d);
    if (d_enh) d_enh[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
  }
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// label[b] = count(|clean[b]| > 0) >= min_active
__global__ __launch_bounds__(256) void vad_label_kernel(const float* __restrict__ clean, int L, int min_active,
                                                        float* __restrict__ label) {
  __shared__ int red[4];
  int c = 0;
  const float* x = clean + (size_t)blockIdx.x * L;
  for (int i = threadIdx.x; i < L; i += 256) c += fabsf(x[i]) > 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) label[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) >= min_active ? 1.f : 0.f;
}

// Cross-pair waveform L1.  One workgroup = WT consecutive time samples of every clip.
//   pair_abs partial, and sgn[i][t] = sum_j sign(w_i[t] - c_j[t])
constexpr int WT = 16;     // time samples per workgroup
constexpr int WJ = 1008;   // clean rows staged per LDS fill: 63 KiB, so that with the static reduction words the
                           // workgroup stays inside the 64 KiB a launch gets without an opt-in

__global__ __launch_bounds__(256) void wave_pair_l1_kernel(const float* __restrict__ w, const float* __restrict__ c, int B,
                                                           int L, float scale, float* __restrict__ d_w,
                                                           double* __restrict__ partial) {
  extern __shared__ float cs[];   // [WJ][WT]
  __shared__ double red[4];
  const int tid = threadIdx.x, tl = tid & (WT - 1), il = tid / WT;   // 16 time lanes x 16 clip lanes
  const int t0 = blockIdx.x * WT;
  const int t = t0 + tl;
  const bool tv = t < L;
  double total = 0.0;
  const int ni = (B + 15) / 16;   // clips per clip lane
  for (int i0 = 0; i0 < ni; i0 += 4) {
    // this thread owns clips il + 16*(i0 .. i0+3)
    float wv[4], ab[4], sg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = il + 16 * (i0 + r);
      wv[r] = (tv && i0 + r < ni && i < B) ? w[(size_t)i * L + t] : 0.f;
      ab[r] = 0.f;
      sg[r] = 0.f;
    }
    for (int j0 = 0; j0 < B; j0 += WJ) {
      const int nj = min(WJ, B - j0);
      __syncthreads();
      for (int e = tid; e < nj * WT; e += 256) {
        const int j = e / WT, tt = e - j * WT;
        cs[e] = (t0 + tt < L) ? c[(size_t)(j0 + j) * L + t0 + tt] : 0.f;
      }
      __syncthreads();
      for (int j = 0; j < nj; ++j) {
        const float cv = cs[j * WT + tl];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = wv[r] - cv;
          ab[r] += fabsf(d);
          sg[r] += (d > 0.f ? 1.f : 0.f) - (d < 0.f ? 1.f : 0.f);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = il + 16 * (i0 + r);
      if (tv && i0 + r < ni && i < B) {
        total += (double)ab[r];
        if (d_w) d_w[(size_t)i * L + t] = sg[r] * scale;
      }
    }
  }
  total = block_sum_d(total, red);
  if (tid == 0) partial[blockIdx.x] = total;
}

// Same result in O(B log B) per time sample: sort the B clean values of a sample, prefix-sum them, and
// every enhanced value finds its rank k by binary search:
//     sum_j |w - c_j| = w*k - P[k] + (P[B] - P[k2]) - w*(B - k2),   sum_j sign(w - c_j) = k - (B - k2)
// with k = #{c_j < w}, k2 = #{c_j <= w}.  One workgroup owns WTS time samples; the sorted columns and their
// double-precision prefix sums live in LDS.
__global__ __launch_bounds__(256) void wave_pair_sorted_kernel(const float* __restrict__ w, const float* __restrict__ c,
                                                               int B, int L, int NP, int WTS, float scale,
                                                               float* __restrict__ d_w, double* __restrict__ partial) {
  extern __shared__ double lds_d[];
  double* pd = lds_d;                                        // [WTS][B+1] prefix sums
  float* cs = reinterpret_cast<float*>(pd + (size_t)WTS * (B + 1));   // [WTS][NP] sorted clean values
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * WTS;
  // NP and WTS are powers of two: every index split below is a shift / mask
  const int lw = __ffs(WTS) - 1, lh = __ffs(NP) - 2;
  for (int e = tid; e < WTS * NP; e += 256) {
    const int tt = e & (WTS - 1), j = e >> lw;
    cs[tt * NP + j] = (j < B && t0 + tt < L) ? c[(size_t)j * L + t0 + tt] : INFINITY;
  }
  __syncthreads();
  // bitonic sort of each row (ascending)
  const int half = NP >> 1;
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1, lj = __ffs(k) - 2; j > 0; j >>= 1, --lj) {
      for (int e = tid; e < WTS * half; e += 256) {
        const int row = e >> lh, p = e & (half - 1);
        const int lo = ((p >> lj) << (lj + 1)) + (p & (j - 1)), hi = lo + j;
        float* r = cs + row * NP;
        const float a = r[lo], b = r[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) {
          r[lo] = b;
          r[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  // prefix sums: each wave scans rows wave, wave+4, ...
  for (int row = wave; row < WTS; row += 4) {
    double carry = 0.0;
    if (lane == 0) pd[(size_t)row * (B + 1)] = 0.0;
    for (int j0 = 0; j0 < B; j0 += 64) {
      const int j = j0 + lane;
      double v = j < B ? (double)cs[row * NP + j] : 0.0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
      }
      if (j < B) pd[(size_t)row * (B + 1) + j + 1] = carry + v;
      carry += __shfl(v, 63, 64);
    }
  }
  __syncthreads();
  double total = 0.0;
  for (int e = tid; e < B * WTS; e += 256) {
    const int tt = e & (WTS - 1), i = e >> lw;
    const int t = t0 + tt;
    if (t >= L) continue;
    const float wv = w[(size_t)i * L + t];
    const float* r = cs + tt * NP;
    int lo = 0, hi = B;                    // lower bound: first index with r[idx] >= wv
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (r[mid] < wv) lo = mid + 1; else hi = mid;
    }
    const int k = lo;
    int k2 = k;
    while (k2 < B && r[k2] == wv) ++k2;
    const double* P = pd + (size_t)tt * (B + 1);
    const double wd = (double)wv;
    total += wd * k - P[k] + (P[B] - P[k2]) - wd * (B - k2);
    if (d_w) d_w[(size_t)i * L + t] = (float)(k - (B - k2)) * scale;
  }
  total = block_sum_d(total, red);
  if (tid == 0) partial[blockIdx.x] = total;
}

// Same algorithm for 256 <= NP <= 2048, with the sort in registers.  A workgroup owns RT = 4 consecutive time samples:
// the clean values of all B clips at those samples are staged once (16-byte pieces per clip, transposed into LDS rows),
// and every thread reads and writes its own clips' RT waveform samples / gradients as one float4.  Per time sample the
// row is sorted with each thread holding EPT = NP/256 consecutive elements: compare-exchange partners at distance
// j < EPT are the thread's own registers, j < 64*EPT another lane of the wave (shuffle), and only the 1-3 largest
// distances go through LDS, so a 2048-element sort needs 6 barriers instead of 66.
constexpr int RT = 4;

template <int EPT>
__device__ __forceinline__ void bitonic_sort_row(float (&v)[EPT], float* xch, int tid) {
  constexpr int NP = 256 * EPT;
  const int lane = tid & 63;
#pragma unroll
  for (int k = 2; k <= NP; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64 * EPT) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) xch[tid * EPT + e] = v[e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const int idx = tid * EPT + e;
          const float o = xch[idx ^ j];
          const bool up = (idx & k) == 0, lower = (idx & j) == 0;
          v[e] = (lower == up) ? fminf(v[e], o) : fmaxf(v[e], o);
        }
      } else if (j >= EPT) {
        const int lj = j / EPT;
        const bool lower = (lane & lj) == 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const float o = __shfl_xor(v[e], lj, 64);
          const bool up = ((tid * EPT + e) & k) == 0;
          v[e] = (lower == up) ? fminf(v[e], o) : fmaxf(v[e], o);
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          if ((e & j) == 0) {
            const bool up = ((tid * EPT + e) & k) == 0;
            const float a = v[e], b = v[e | j];
            const float mn = fminf(a, b), mx = fmaxf(a, b);
            v[e] = up ? mn : mx;
            v[e | j] = up ? mx : mn;
          }
        }
      }
    }
  }
}

template <int EPT>
__global__ __launch_bounds__(256, 2) void wave_pair_rows_kernel(const float* __restrict__ w, const float* __restrict__ c,
                                                             int B, int L, int ntiles, float scale,
                                                             float* __restrict__ d_w, double* __restrict__ partial) {
  // a tile reads 16 bytes of every clip row: eight neighbouring tiles share each 128-byte line, so neighbours are placed
  // on one XCD (without this every line was fetched by up to 8 L2s: 6x the algorithmic traffic)
  const int vblk = xcd_chunked_block();
  if (vblk >= ntiles) return;
  constexpr int NP = 256 * EPT, R = EPT;   // R: clips per thread (B <= NP)
  extern __shared__ double lds_d[];
  double* P = lds_d;                                   // [B + 1] prefix sums of the sorted row
  float* srt = reinterpret_cast<float*>(P + B + 1);    // [NP] sorted row (also the exchange buffer of the sort)
  float* ct = srt + NP;                                // [RT][pitch] clean tile, row = time sample
  __shared__ double red[4], wtot[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pitch = B | 1;
  const int t0 = vblk * RT;
  const bool vec = (L & 3) == 0 && t0 + RT <= L;       // whole, 16-byte aligned pieces

  constexpr int Q = RT / 4;   // float4 pieces per clip
  for (int e = tid; e < Q * B; e += 256) {
    const int j = e / Q, hf = e - j * Q;
    float x[4];
    if (vec) {
      const float4 q = ld4(c + (size_t)j * L + t0 + 4 * hf);
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + 4 * hf + u;
        x[u] = t < L ? c[(size_t)j * L + t] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) ct[(4 * hf + u) * pitch + j] = x[u];
  }
  float wv[R][RT], dv[R][RT];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int u = 0; u < RT; ++u) dv[r][u] = 0.f;
    const int i = tid + 256 * r;
    const size_t row = (size_t)(i < B ? i : 0) * L + t0;
    if (vec) {
#pragma unroll
      for (int h = 0; h < Q; ++h) {
        const float4 q0 = ld4(w + row + 4 * h);
        wv[r][4 * h] = q0.x; wv[r][4 * h + 1] = q0.y; wv[r][4 * h + 2] = q0.z; wv[r][4 * h + 3] = q0.w;
      }
    } else {
#pragma unroll
      for (int u = 0; u < RT; ++u) wv[r][u] = t0 + u < L ? w[row + u] : 0.f;
    }
  }
  __syncthreads();

  double total = 0.0;
  // the loop stays rolled (its body holds the whole unrolled sorting network); wv / dv are therefore addressed with
  // compile-time indices and selects, never with tt
#pragma unroll 1
  for (int tt = 0; tt < RT; ++tt) {
    float v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = tid * EPT + e;
      v[e] = j < B ? ct[tt * pitch + j] : INFINITY;
    }
    bitonic_sort_row<EPT>(v, srt, tid);
    // prefix sums in double: own elements, then a scan over the wave, then over the four waves
    double run[EPT], own = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      own += tid * EPT + e < B ? (double)v[e] : 0.0;
      run[e] = own;
    }
    double inc = own;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    __syncthreads();              // the sort's last exchange (if any) has been read
    if (lane == 63) wtot[wave] = inc;
#pragma unroll
    for (int e = 0; e < EPT; ++e) srt[tid * EPT + e] = v[e];
    __syncthreads();
    double base = inc - own;
    for (int wv2 = 0; wv2 < wave; ++wv2) base += wtot[wv2];
    if (tid == 0) P[0] = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
      if (tid * EPT + e < B) P[tid * EPT + e + 1] = base + run[e];
    __syncthreads();
    const bool live = t0 + tt < L;
    // k = #{c < x} and k2 = #{c <= x} for the thread's R clips by branch-free binary searches over the padded row
    // (entries >= B are +inf, so no bound check): the 2R searches advance in lock step, their LDS reads overlap
    float x[R];
    int k[R], k2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      x[r] = wv[r][0];
#pragma unroll
      for (int u = 1; u < RT; ++u) x[r] = u == tt ? wv[r][u] : x[r];
      k[r] = k2[r] = 0;
    }
#pragma unroll
    for (int s2 = NP >> 1; s2 > 0; s2 >>= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float a = srt[k[r] + s2 - 1], b = srt[k2[r] + s2 - 1];
        k[r] += a < x[r] ? s2 : 0;
        k2[r] += b <= x[r] ? s2 : 0;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {   // the steps above reach NP - 1 at most: one more element decides a full row
      k[r] += srt[k[r]] < x[r] ? 1 : 0;
      k2[r] += srt[k2[r]] <= x[r] ? 1 : 0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = tid + 256 * r;
      const int ka = k[r] < B ? k[r] : B, kb = k2[r] < B ? k2[r] : B;   // x = +inf would count the padding
      const double xd = (double)x[r];
      if (live && i < B) total += xd * ka - P[ka] + (P[B] - P[kb]) - xd * (B - kb);
      const float gv = (float)(ka - (B - kb)) * scale;
#pragma unroll
      for (int u = 0; u < RT; ++u) dv[r][u] = u == tt ? gv : dv[r][u];
    }
    __syncthreads();              // srt / P are rewritten for the next time sample
  }
  if (d_w != nullptr) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = tid + 256 * r;
      if (i >= B) continue;
      const size_t row = (size_t)i * L + t0;
      if (vec) {
#pragma unroll
        for (int h = 0; h < Q; ++h)
          st4(d_w + row + 4 * h, make_float4(dv[r][4 * h], dv[r][4 * h + 1], dv[r][4 * h + 2], dv[r][4 * h + 3]));
      } else {
#pragma unroll
        for (int u = 0; u < RT; ++u)
          if (t0 + u < L) d_w[row + u] = dv[r][u];
      }
    }
  }
  total = block_sum_d(total, red);
  if (tid == 0) partial[vblk] = total;
}

// out[0..3] = loss, vad_loss, waveform_loss, spectral_loss;  d_vad[b] = (sigmoid(x) - y)/B
__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* __restrict__ spec_partial, int n_spec,
                                                            double spec_count, const double* __restrict__ wave_partial,
                                                            int n_wave, double wave_count, const float* __restrict__ vad,
                                                            const float* __restrict__ label, int B, float* __restrict__ d_vad,
                                                            float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0, wv = 0.0, bce = 0.0;
  for (int i = threadIdx.x; i < n_spec; i += 256) s += spec_partial[i];
  for (int i = threadIdx.x; i < n_wave; i += 256) wv += wave_partial[i];
  for (int b = threadIdx.x; b < B; b += 256) {
    const float x = vad[b], y = label[b];
    bce += (double)(fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x))));
    if (d_vad) d_vad[b] = (1.f / (1.f + expf(-x)) - y) / (float)B;
  }
  s = block_sum_d(s, red);
  wv = block_sum_d(wv, red);
  bce = block_sum_d(bce, red);
  if (threadIdx.x == 0) {
    const float spectral = (float)(s / spec_count), wave = (float)(wv / wave_count), v = (float)(bce / (double)B);
    out[0] = wave + spectral + v;
    out[1] = v;
    out[2] = wave;
    out[3] = spectral;
  }
}

// out[b] = max |x[b][:]|
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  __shared__ float red[4];
  const float* row = x + (size_t)blockIdx.x * n;
  float m = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(row[i]));
  m = wave_reduce_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace

// Per-clip max |x| (train/models.py:614-620) for callers that did not get it from the STFT kernel.
NOISE_EXPORT int noise_absmax(const float* x, float* out, int B, long n, void* stream) {
  NOISE_REQUIRE(x && out && B > 0 && n > 0, "noise_absmax: bad argument");
  hipLaunchKernelGGL(absmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, n, out);
  NOISE_CHECK_LAUNCH("noise_absmax");
  return 0;
}

NOISE_EXPORT int noise_loss_max_blocks(void) { return 1024; }

// Spectral L1.  partial: at least noise_loss_max_blocks() doubles; *n_partial receives the count written.
NOISE_EXPORT int noise_spectral_l1(const float* enh, const float* clean_logmag, float* d_enh, double* partial,
                                   int* n_partial, long n, void* stream) {
  NOISE_REQUIRE(enh && clean_logmag && partial && n > 0, "noise_spectral_l1: bad argument");
  const int grid = noise_grid(n, 256 * 8, 1024);
  if (n_partial) *n_partial = grid;
  hipLaunchKernelGGL(spectral_l1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, enh, clean_logmag, n,
                     (float)(1.0 / (double)n), d_enh, partial);
  NOISE_CHECK_LAUNCH("noise_spectral_l1");
  return 0;
}

NOISE_EXPORT int noise_vad_label(const float* clean, float* label, int B, int L, int min_active, void* stream) {
  NOISE_REQUIRE(clean && label && B > 0 && L > 0, "noise_vad_label: bad argument");
  hipLaunchKernelGGL(vad_label_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, clean, L, min_active, label);
  NOISE_CHECK_LAUNCH("noise_vad_label");
  return 0;
}

// Cross-pair waveform L1 (the reference's broadcast).  partial: at least ceil(L/2) doubles.
NOISE_EXPORT int noise_wave_pair_l1(const float* wave, const float* clean, float* d_wave, double* partial, int* n_partial,
                                    int B, int L, void* stream) {
  NOISE_REQUIRE(wave && clean && partial && B > 0 && L > 0, "noise_wave_pair_l1: bad argument");
  if (B >= 64 && B <= 2048) {   // sorted algorithm: B log B instead of B^2 per time sample
    int NP = 64;
    while (NP < B) NP <<= 1;
    if (NP >= 256) {            // register sort, RT time samples per workgroup
      const int pitch = B | 1;
      const size_t smem = (size_t)(B + 1) * sizeof(double) + (size_t)(NP + RT * pitch) * sizeof(float);
      const int grid = noise_cdiv(L, RT);
      if (n_partial) *n_partial = grid;
      const float scale = (float)(1.0 / ((double)B * (double)B * (double)L));
      hipStream_t st = (hipStream_t)stream;
#define ROWS(ept)                                                                                                  \
  do {                                                                                                             \
    NOISE_ALLOW_LDS(wave_pair_rows_kernel<ept>, smem);                                                             \
    hipLaunchKernelGGL(wave_pair_rows_kernel<ept>, dim3(noise_pad8(grid)), dim3(256), smem, st, wave, clean, B, L,  \
                       grid, scale, d_wave, partial);                                                                       \
  } while (0)
      if (NP == 256) ROWS(1);
      else if (NP == 512) ROWS(2);
      else if (NP == 1024) ROWS(4);
      else ROWS(8);
#undef ROWS
      NOISE_CHECK_LAUNCH("noise_wave_pair_l1/rows");
      return 0;
    }
    // columns per workgroup: the kernel is a chain of barriers and dependent LDS reads, so several small
    // workgroups per CU (<= 40 KiB of LDS each) hide far more latency than one large one
    const size_t per_col = (size_t)(B + 1) * sizeof(double) + (size_t)NP * sizeof(float);
    int WTS = 8;
    while (WTS > 2 && WTS * per_col > 40 * 1024) WTS >>= 1;
    const size_t smem = WTS * per_col;
    const int grid = noise_cdiv(L, WTS);
    if (n_partial) *n_partial = grid;
    const double cnt = (double)B * (double)B * (double)L;
    NOISE_ALLOW_LDS(wave_pair_sorted_kernel, smem);
    hipLaunchKernelGGL(wave_pair_sorted_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, wave, clean, B, L, NP, WTS,
                       (float)(1.0 / cnt), d_wave, partial);
    NOISE_CHECK_LAUNCH("noise_wave_pair_l1/sorted");
    return 0;
  }
  const int grid = noise_cdiv(L, WT);
  if (n_partial) *n_partial = grid;
  const size_t smem = (size_t)(B < WJ ? B : WJ) * WT * sizeof(float);
  const double cnt = (double)B * (double)B * (double)L;
  hipLaunchKernelGGL(wave_pair_l1_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, wave, clean, B, L,
                     (float)(1.0 / cnt), d_wave, partial);
  NOISE_CHECK_LAUNCH("noise_wave_pair_l1");
  return 0;
}

// out4 = [loss, vad_loss, waveform_loss, spectral_loss]
NOISE_EXPORT int noise_loss_finalize(const double* spec_partial, int n_spec, long spec_count, const double* wave_partial,
                                     int n_wave, int B, int L, const float* vad, const float* label, float* d_vad,
                                     float* out4, void* stream) {
  NOISE_REQUIRE(spec_partial && wave_partial && vad && label && out4 && B > 0, "noise_loss_finalize: bad argument");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, spec_partial, n_spec,
                     (double)spec_count, wave_partial, n_wave, (double)B * (double)B * (double)L, vad, label, B, d_vad, out4);
  NOISE_CHECK_LAUNCH("noise_loss_finalize");
  return 0;
}
