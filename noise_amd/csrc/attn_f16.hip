// Attention core of the bottleneck Transformer layer for the bf16 inference path, gfx950.
//
// softmax((q / sqrt(d_k)) k^T) v per head (train/models.py:27-37, :60-89), no dropout (eval mode), any clip length:
// half-precision operands on the matrix cores, fp32 scores, online softmax, so the n x n scores are never stored.
// K and V^T of one (clip, head) live in LDS; a wave owns 16 queries and walks the keys 32 at a time:
//     S^T [key][query] = K Q^T          two v_mfma_f32_16x16x16_f16 (d_k <= 16 is the reduction index)
//     running max / rescale (only when some query's maximum moved) / p = 2^(s - m), scores pre-scaled by log2(e)
//     O^T [d][query] += V^T P^T         one v_mfma_f32_16x16x32_f16 whose 32-deep reduction index is laid out so that
//                                       the lane's own 8 probabilities are its B fragment (no shuffles): slot j of
//                                       lane group q is key 4q + j (j < 4) or 16 + 4q + (j - 4); row d_k of V^T is
//                                       all ones, so output row d_k is the sum of the weights (the normaliser)
#include "common.h"

namespace {

using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct Attn16Dims {
  long ntok;
  int n, heads, dk, hd, n32, vp;   // n32 = n rounded up to the 64 keys of a step; vp = row pitch of V^T (halves)
  int kp, vrows;                   // columns of K kept in LDS (d_k rounded up to 4), rows of V^T kept (d_k + the ones row)
};

// KP / VROWS: columns of K and rows of V^T kept in LDS, compile-time so that the operand loads below do not branch
template <int KP, int VROWS>
__global__ __launch_bounds__(1024) void attn_core_f16_kernel(const float* __restrict__ qkv, float* __restrict__ o, Attn16Dims d,
                                                            int nsplit, int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  // Only what exists is kept: the lanes of an MFMA operand whose columns / rows lie beyond feed zeros without reading, and
  // a 10 s clip at hop 128 then takes 65 KB instead of 90 KB, i.e. two workgroups per CU instead of one.
  _Float16* ks = reinterpret_cast<_Float16*>(lds_raw);   // [n32][kp], rows >= n and columns >= d_k zero
  _Float16* vt = ks + (size_t)d.n32 * KP;               // [vrows][vp], V transposed, zero padded, row d_k all ones
  const int n = d.n, dk = d.dk, O = 3 * d.hd;
  // the splits of a (clip, head) restage the same K/V and the heads of a clip share every 32-byte sector of its qkv
  // rows: consecutive virtual indices run on one XCD, so those bytes come from HBM once, not once per L2
  const int vblk = xcd_chunked_block();
  if (vblk >= total) return;
  const int split = vblk % nsplit, bh = vblk / nsplit;
  const int b = bh / d.heads, h = bh % d.heads;
  const long base = (long)b * n;
  const int nthr = blockDim.x, nw = nthr >> 6;
  for (int i = threadIdx.x; i < d.n32 * 16; i += nthr) {     // clamped addresses, padding selected afterwards
    const int j = i >> 4, c = i & 15;
    const bool in = j < n && c < dk;
    const float* row = qkv + (base + (j < n ? j : n - 1)) * O + h * dk + (c < dk ? c : dk - 1);
    const float kv = row[d.hd], vv = row[2 * d.hd];
    if (c < KP) ks[j * KP + c] = (_Float16)(in ? kv : 0.f);
    // row d_k of V^T is all ones over the real keys: the P V product then carries sum_j p_j in output row d_k
    if (c < VROWS) vt[c * d.vp + j] = (_Float16)(in ? vv : ((c == dk && j < n) ? 1.f : 0.f));   // (d_k = 16 leaves no spare row)
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p = lane & 15, q = lane >> 4;
  const float qscale = 1.44269504f / sqrtf((float)dk);   // scores in units of log 2: p = 2^(s - m)
  const int ntile = (n + 15) >> 4, nblk = d.n32 >> 6;
  for (int qt = split * nw + wave; qt < ntile; qt += nw * nsplit) {
    const int qi = qt * 16 + p;
    const bool valid = qi < n;
    const float* qrow = qkv + (base + (valid ? qi : n - 1)) * O + h * dk;
    f16x4 qb;
#pragma unroll
    for (int r = 0; r < 4; ++r) qb[r] = (_Float16)((4 * q + r < dk) ? qrow[4 * q + r] * qscale : 0.f);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;      // lsum: the sum of the weights when d_k = 16 leaves no spare output row
    for (int kb = 0; kb < nblk; ++kb) {      // 64 keys per step: two independent score / weight chains in flight
      const int key0 = 64 * kb;
      f16x4 ka[4], va[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        ka[g] = 4 * q < KP ? *reinterpret_cast<const f16x4*>(ks + (key0 + 16 * g + p) * KP + 4 * q) : f16x4{};
        va[g] = p < VROWS ? *reinterpret_cast<const f16x4*>(vt + p * d.vp + key0 + 16 * g + 4 * q) : f16x4{};
      }
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      f32x4 sv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) sv[g] = __builtin_amdgcn_mfma_f32_16x16x16f16(ka[g], qb, zero, 0, 0, 0);
      if (key0 + 64 > n) {      // only the last step holds keys past the clip
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + 16 * g + 4 * q + r >= n) sv[g][r] = -INFINITY;
      }
      float bm = -INFINITY;
#pragma unroll
      for (int g = 0; g < 4; ++g) bm = fmaxf(bm, fmaxf(fmaxf(sv[g][0], sv[g][1]), fmaxf(sv[g][2], sv[g][3])));
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      if (__any(bm > m)) {                    // the running maximum of some query moved: rescale what is accumulated
        const float mn = fmaxf(m, bm);        // finite: every step holds at least one real key
        const float sc = __builtin_amdgcn_exp2f(m - mn);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] *= sc;
        lsum *= sc;
        m = mn;
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        f16x8 pb, vv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pb[r] = (_Float16)__builtin_amdgcn_exp2f(sv[2 * h2][r] - m);
          pb[4 + r] = (_Float16)__builtin_amdgcn_exp2f(sv[2 * h2 + 1][r] - m);
          vv[r] = va[2 * h2][r];
          vv[4 + r] = va[2 * h2 + 1][r];
        }
        if (dk == 16) {
#pragma unroll
          for (int r = 0; r < 8; ++r) lsum += (float)pb[r];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(vv, pb, acc, 0, 0, 0);
      }
    }
    // output row d_k (lane group q = d_k / 4, register d_k % 4) holds the sum of this query's weights
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((dk & 3) == r) l = acc[r];
    l = __shfl(l, ((dk >> 2) << 4) | p, 64);
    if (dk == 16) {
      lsum += __shfl_xor(lsum, 16, 64);
      lsum += __shfl_xor(lsum, 32, 64);
      l = lsum;
    }
    if (valid) {
      const float inv = 1.f / l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * q + r;
        if (c < dk) o[(base + qi) * d.hd + h * dk + c] = acc[r] * inv;
      }
    }
  }
}

}  // namespace

// o [ntok][heads*dk] fp32 from qkv [ntok][3*heads*dk] fp32 (q | k | v, head-major), n tokens per clip.
NOISE_EXPORT int noise_attn_core_fwd_f16(const float* qkv, float* o, long ntok, int n, int d_model, int heads, int dk,
                                         void* stream) {
  NOISE_REQUIRE(qkv && o, "noise_attn_core_fwd_f16: null pointer");
  NOISE_REQUIRE(heads > 0 && dk > 0 && dk <= 16 && heads * dk <= d_model, "noise_attn_core_fwd_f16: heads=%d d_k=%d unsupported",
                heads, dk);
  NOISE_REQUIRE(n > 0 && ntok > 0 && ntok % n == 0, "noise_attn_core_fwd_f16: %ld tokens is not a multiple of n=%d", ntok, n);
  Attn16Dims d{ntok, n, heads, dk, heads * dk, (n + 63) / 64 * 64, 0, 0, 0};
  d.vp = d.n32 + 8;
  // The reference configuration (d_k = 64 // 6 = 10) may keep K as [n][12] and V^T as 11 rows.  The narrow rows cost bank
  // conflicts (measured 0.18 -> 0.23 ms where nothing else changes), so they are used exactly when they let a second
  // workgroup onto the CU (10 s clips at hop 128: 90 KB -> 65 KB, 0.76 -> 0.66 ms).
  const size_t full_bytes = ((size_t)d.n32 * 16 + (size_t)16 * d.vp) * sizeof(_Float16);
  const size_t compact_bytes = ((size_t)d.n32 * 12 + (size_t)11 * d.vp) * sizeof(_Float16);
  const bool compact = dk == 10 && full_bytes > 80 * 1024 && compact_bytes <= 80 * 1024;
  d.kp = compact ? 12 : 16;
  d.vrows = compact ? 11 : 16;
  const size_t smem = ((size_t)d.n32 * d.kp + (size_t)d.vrows * d.vp) * sizeof(_Float16);
  NOISE_REQUIRE(smem <= 158 * 1024, "noise_attn_core_fwd_f16: n=%d tokens per clip exceed the LDS-resident K/V limit", n);
  const long bh = (ntok / n) * heads;
  const int threads = smem > 32 * 1024 ? 1024 : 256, nw = threads / 64;
  int nsplit = (int)(2048 / bh);
  const int max_split = ((n + 15) / 16 + nw - 1) / nw;
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  const int total = (int)(bh * nsplit);
  const dim3 grid((unsigned)noise_pad8(total));
  if (compact) {
    NOISE_ALLOW_LDS((attn_core_f16_kernel<12, 11>), smem);
    hipLaunchKernelGGL((attn_core_f16_kernel<12, 11>), grid, dim3(threads), smem, (hipStream_t)stream, qkv, o, d, nsplit, total);
  } else {
    NOISE_ALLOW_LDS((attn_core_f16_kernel<16, 16>), smem);
    hipLaunchKernelGGL((attn_core_f16_kernel<16, 16>), grid, dim3(threads), smem, (hipStream_t)stream, qkv, o, d, nsplit, total);
  }
  NOISE_CHECK_LAUNCH("noise_attn_core_fwd_f16");
  return 0;
}
