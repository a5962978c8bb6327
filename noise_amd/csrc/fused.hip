// Block-fused inference kernels of the denoiser U-Net for MI355X (gfx950): bf16 storage, fp32 accumulation.
//
// Eval-mode forward of train/models.py:348-407 (InvertedResidual), :299-345 (its transposed depthwise variant),
// :1151-1173 (first conv / last transposed conv) and :583-612 (Base._forward: padding, skip adds with the
// crop-from-the-end of :608, final crop :610) with BatchNorm folded into the weights.  One kernel per
// inverted-residual block: a workgroup owns a spatial tile, expands it (1x1, matrix cores), runs the 3x3 depthwise
// (transposed) convolution on the LDS-resident hidden tile (packed VALU) and projects it (1x1, matrix cores), so the
// t-times wider hidden tensors never reach HBM.  Activations between kernels are NHWC bf16.
//
// Inside a kernel the tiles are IEEE half precision: everything that lives in LDS is either a block input (bf16 values
// convert exactly) or sits behind a ReLU6 (bounded by 6), so half's 11 significant bits cost nothing in range, the
// depthwise stage runs on v_pk_fma_f16 (two multiply-adds per lane and instruction, no unpacking) and the folded
// weights keep 11 bits as well.  Only what leaves a kernel is rounded to bf16.
//
// MFMA arrangement (v_mfma_f32_16x16x32_f16): A = weights (rows = output channels), B = activations (columns =
// 16 pixels), so a lane ends up with 4 consecutive output channels of one pixel: one 8-byte store.  LDS tiles
// are [pixel][channels + 8] halves: the B fragment of a lane is 16 contiguous bytes and the odd 16-byte pitch spreads
// the 16 pixels of a fragment over the banks.
#include "common.h"
#include <type_traits>

namespace {

// Diagnostic build (-DFUSED_STAMPS, scripts/dbg/fused_stamps.py): cycles one workgroup spends per phase.
#ifdef FUSED_STAMPS
__device__ unsigned long long g_fused_stamps[32];
#define FU_T(i) do { if (blockIdx.x == 8 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_fused_stamps[i] += now_ - last_; last_ = now_; } } while (0)
#else
#define FU_T(i)
#endif
#define IR_T(i) do { if ((CIN == 24 && TR) || (CIN == 16 && !TR)) FU_T((TR ? 16 : 24) + (i)); } while (0)


using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }

// Hidden activations are kept divided by 6 (the 1/6 is folded into the weights and biases that produce them, the 6 into
// the weights that consume them): ReLU6 becomes a clamp to [0, 1], which the instruction that produces the value applies
// for free (the `clamp` output modifier of v_add_f32 / v_pk_fma_f16).
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 1.f); }

__device__ __forceinline__ f16x8 clamp01_h8(f16x8 v) {
  const f16x8 zero = (f16x8)((_Float16)0.f), one = (f16x8)((_Float16)1.f);
  return __builtin_elementwise_min(__builtin_elementwise_max(v, zero), one);
}

__device__ __forceinline__ f16x8 fma_h8(f16x8 a, f16x8 b, f16x8 c) { return __builtin_elementwise_fma(a, b, c); }

// two bf16 packed in a 32-bit word -> floats
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ f16x8 to_h8(const float (&f)[8]) {
  f16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (_Float16)f[j];
  return v;
}

struct IrArgs {
  const void* x;        // [B][Hi][Wi][CIN]
  const void* skip;     // optional [B][Hs][Ws][CIN], added to x at rows soy.., columns sox.. (same element type as x)
  const _Float16* w1;   // [HT*16][KP]  expand weights, BatchNorm scale folded, zero padded
  const float* b1;      // [HT*16]
  const _Float16* wd;   // [9][G*8]     depthwise weights (tap-major), BatchNorm scale folded
  const _Float16* b2;   // [G*8]
  const _Float16* w3;   // [CT*16][HP]  projection weights, BatchNorm scale folded, zero padded
  const float* b3;      // [CT*16]
  void* out;            // [B][Ho][Wo][COUT]
  int B, Hi, Wi, Ho, Wo, Hs, Ws, soy, sox, tiles_y, tiles_x;
};

// Geometry of one workgroup tile.  TR = false: stride-2 convolution, TH x 16 output pixels from a (2TH+1) x 33 input
// region.  TR = true: stride-2 transposed convolution, TH x 16 input cells -> 2TH x 32 output pixels from a
// (TH+1) x 17 input region (cell (i,j) -> outputs (2i..2i+1, 2j..2j+1) from inputs (i..i+1, j..j+1)).
template <bool TR, int TH>
struct TileGeom {
  static constexpr int TW = 16;
  static constexpr int IH = TR ? TH + 1 : 2 * TH + 1, IW = TR ? TW + 1 : 2 * TW + 1;
  static constexpr int PIN = IH * IW, NT1 = (PIN + 15) / 16;
  static constexpr int OH = TR ? 2 * TH : TH, OW = TR ? 2 * TW : TW, LOW = TR ? 5 : 4;   // OW = 1 << LOW
  static constexpr int POUT = OH * OW, NT3 = POUT / 16;
  // Transposed blocks run depthwise + project once per row of cells: the depthwise tile then holds 2 x 32 outputs instead
  // of 2TH x 32, and the LDS that saves is another resident workgroup per CU.
  static constexpr int PASSES = TR ? TH : 1, PPOUT = POUT / PASSES, PNT3 = PPOUT / 16;
};

template <int CIN, int HID, int COUT, bool TR, int TH>
struct IrCfg {
  using G_ = TileGeom<TR, TH>;
  // MFMA reduction steps cover 32 channels (KP, HP: the padded widths of the weight images); the tiles in LDS only hold
  // the channels that exist, rounded up to the 8 a lane reads at once (KR, HR): lanes whose 8 channels lie beyond feed
  // zeros without reading.  The narrower rows are what lets another workgroup fit on a CU.
  static constexpr int KP = rup(CIN, 32), KS1 = KP / 32, KR = rup(CIN, 8), IP = KR + 8;
  static constexpr int HT = (HID + 15) / 16, HP = rup(HID, 32), KS3 = HP / 32, HR = rup(HID, 8), HPI = HR + 8, G = (HID + 7) / 8;
  static_assert(HT * 16 <= HPI, "the expand epilogue writes 16 channels per tile into a row");
  static constexpr int CT = (COUT + 15) / 16;
  static constexpr int LDS_IN = G_::NT1 * 16 * IP, LDS_H1 = G_::NT1 * 16 * HPI, LDS_H2 = G_::PPOUT * HPI;   // halves
  static constexpr int LDS_W1 = HT * KS1 * 64 * 8, LDS_W3 = CT * KS3 * 64 * 8;                             // halves
  static constexpr int LDS_F = HT * 16 + G_::NT1 * 16;                                                     // floats: b1, pixel mask
  // The depthwise tile may live where the input tile was (dead once the expand has read it) at the price of one more
  // barrier per tile: done exactly when that buys another resident workgroup (160 KiB of LDS per CU).
  static constexpr size_t BYTES_SEP = (size_t)(LDS_IN + LDS_H1 + LDS_H2 + LDS_W1 + LDS_W3) * 2 + (size_t)LDS_F * 4;
  static constexpr bool ALIAS_H2 = LDS_H2 <= LDS_IN && 163840 / (BYTES_SEP - 2 * LDS_H2) > 163840 / BYTES_SEP;
  static constexpr int OFF_H2 = ALIAS_H2 ? 0 : LDS_IN + LDS_H1, OFF_W1 = LDS_IN + LDS_H1 + (ALIAS_H2 ? 0 : LDS_H2);
  static constexpr size_t LDS_BYTES = BYTES_SEP - (ALIAS_H2 ? 2 * LDS_H2 : 0);
};

template <int CIN, int HID, int COUT, bool TR, int TH, bool IN_F32, bool OUT_F32>
__global__ __launch_bounds__(256) void ir_block_kernel(IrArgs a) {
  using C = IrCfg<CIN, HID, COUT, TR, TH>;
  using T = TileGeom<TR, TH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  _Float16* in_t = reinterpret_cast<_Float16*>(lds_raw);   // [NT1*16][IP]
  _Float16* h1 = in_t + C::LDS_IN;                          // [NT1*16][HPI]  expand output (zero outside the image)
  _Float16* h2 = in_t + C::OFF_H2;                          // [PPOUT][HPI]   depthwise output (one pass)
  _Float16* wf1 = in_t + C::OFF_W1;                         // [HT][KS1][64][8] A fragments of the expand weights
  _Float16* wf3 = wf1 + C::LDS_W1;                          // [CT][KS3][64][8]
  float* b1s = reinterpret_cast<float*>(wf3 + C::LDS_W3);   // [HT*16]
  float* maskf = b1s + C::HT * 16;                          // [NT1*16]  1 inside the image, 0 outside
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;

  // ---- one-time set-up: zero the tiles (padding channels / pixels stay zero for ever), stage weight fragments
  for (int i = tid; i < C::OFF_W1 / 8; i += 256)
    reinterpret_cast<uint4*>(lds_raw)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < C::HT * C::KS1 * 64; i += 256) {
    const int l = i & 63, f = i >> 6, ks = f % C::KS1, mt = f / C::KS1;
    *reinterpret_cast<f16x8*>(wf1 + i * 8) =
        *reinterpret_cast<const f16x8*>(a.w1 + (size_t)(16 * mt + (l & 15)) * C::KP + 32 * ks + 8 * (l >> 4));
  }
  for (int i = tid; i < C::CT * C::KS3 * 64; i += 256) {
    const int l = i & 63, f = i >> 6, ks = f % C::KS3, mt = f / C::KS3;
    *reinterpret_cast<f16x8*>(wf3 + i * 8) =
        *reinterpret_cast<const f16x8*>(a.w3 + (size_t)(16 * mt + (l & 15)) * C::HP + 32 * ks + 8 * (l >> 4));
  }
  for (int i = tid; i < C::HT * 16; i += 256) b1s[i] = a.b1[i];
  for (int i = tid; i < T::NT1 * 16; i += 256) maskf[i] = 0.f;
  // depthwise weights of this thread's channel group stay in registers: thread = (channel group, pixel lane)
  constexpr int NPL = 256 / C::G;
  const int cg = tid % C::G, pl = tid / C::G;
  f16x8 wdr[9], b2r;
#pragma unroll
  for (int k = 0; k < 9; ++k) wdr[k] = *reinterpret_cast<const f16x8*>(a.wd + k * (C::G * 8) + 8 * cg);
  b2r = *reinterpret_cast<const f16x8*>(a.b2 + 8 * cg);
  float b3r[C::CT][4];
#pragma unroll
  for (int mt = 0; mt < C::CT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b3r[mt][r] = a.b3[16 * mt + 4 * q + r];

  // ---- input tile loader: pieces of 8 channels, prefetched into registers one tile ahead.  The region coordinates of a
  // thread's pieces do not depend on the tile: they are worked out once.
  constexpr int C8 = CIN / 8, PIECES = T::PIN * C8, NI = (PIECES + 255) / 256;
  int pry[NI], prx[NI], plds[NI], pc8[NI], ppix[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int it = tid + 256 * u;
    const int pi = it / C8;
    pc8[u] = 8 * (it - pi * C8);
    pry[u] = pi / T::IW;
    prx[u] = pi - pry[u] * T::IW;
    plds[u] = it < PIECES ? pi * C::IP + pc8[u] : -1;
    ppix[u] = pi;
  }
  // Raw prefetch registers: nothing is computed from a loaded value before the next iteration needs it, so the loads
  // stay in flight behind the current tile's arithmetic (a conversion placed here would wait for them at once).
  using raw_t = typename std::conditional<IN_F32, float4, uint4>::type;   // 16 bytes: 4 floats or 8 bf16
  constexpr int NR = IN_F32 ? 2 : 1;
  raw_t prx_[NI][NR], prs_[NI][NR];
  unsigned pin = 0;      // bit u: piece u lies inside the image
  const bool has_skip = a.skip != nullptr;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;      // < 2^31 (checked by the launcher)
  constexpr int ESZ = IN_F32 ? 4 : 2;
  int nb = 0, ny0 = 0, nx0 = 0;                        // origin of the prefetched tile
  auto prefetch = [&](int tile) {
    const unsigned tx = (unsigned)tile % (unsigned)a.tiles_x, r = (unsigned)tile / (unsigned)a.tiles_x;
    const unsigned ty = r % (unsigned)a.tiles_y;
    nb = (int)(r / (unsigned)a.tiles_y);
    ny0 = TR ? (int)ty * TH : 2 * (int)ty * TH - 1;    // image row / column of the region's first input pixel
    nx0 = TR ? (int)tx * T::TW : 2 * (int)tx * T::TW - 1;
    const int b = nb, y0 = ny0, x0 = nx0;
    const char* xb = static_cast<const char*>(a.x) + (size_t)b * a.Hi * a.Wi * CIN * ESZ;
    const char* sb = has_skip ? static_cast<const char*>(a.skip) + ((size_t)b * a.Hs * a.Ws + (size_t)a.soy * a.Ws + a.sox) * CIN * ESZ
                              : xb;      // without a skip tensor the second load re-reads x (never used)
    const int sW = has_skip ? a.Ws : a.Wi;
    pin = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int iy = y0 + pry[u], ix = x0 + prx[u];
      const bool in = (unsigned)iy < (unsigned)a.Hi && (unsigned)ix < (unsigned)a.Wi;
      const int iyc = in ? iy : 0, ixc = in ? ix : 0;     // out-of-image pixels are masked after the expand
      const raw_t* px = reinterpret_cast<const raw_t*>(xb + (size_t)((iyc * a.Wi + ixc) * CIN + pc8[u]) * ESZ);
      const raw_t* ps = reinterpret_cast<const raw_t*>(sb + (size_t)((iyc * sW + ixc) * CIN + pc8[u]) * ESZ);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        prx_[u][r] = px[r];
        if (has_skip) prs_[u][r] = ps[r];
      }
      pin |= in ? (1u << u) : 0u;
    }
  };
  auto raw_to_float = [&](const raw_t (&r)[NR], float (&f)[8]) {
    if constexpr (IN_F32) {
      f[0] = r[0].x; f[1] = r[0].y; f[2] = r[0].z; f[3] = r[0].w; f[4] = r[1].x; f[5] = r[1].y; f[6] = r[1].z; f[7] = r[1].w;
    } else {
      const unsigned w[4] = {r[0].x, r[0].y, r[0].z, r[0].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf_lo(w[j]);
        f[2 * j + 1] = bf_hi(w[j]);
      }
    }
  };

  // Neighbouring tiles share halo pixels and partial cache lines.  XCD j (workgroups j, j + 8, ...: one L2) owns the
  // contiguous tile range [j T8, (j + 1) T8) and its workgroups stride through it together, so the window of tiles in
  // flight on an L2 advances row by row through one clip after the other and the halo rows of a window are still in
  // that L2 when the next window needs them.
  const int T8 = (ntiles + 7) >> 3, tstep = (int)(gridDim.x >> 3);
  const int tend = min(ntiles, (int)((blockIdx.x & 7u) + 1) * T8);
  int tile = (int)(blockIdx.x & 7u) * T8 + (int)(blockIdx.x >> 3);
  if (tile < tend) prefetch(tile);
  __syncthreads();
#ifdef FUSED_STAMPS
  unsigned long long last_ = clock64();
#endif
  for (; tile < tend; tile += tstep) {
    const int b = nb, y0 = ny0, x0 = nx0;
    IR_T(5);
    // ---- registers -> input tile (+ the pixel mask, written by the thread that holds a pixel's first piece)
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      if (plds[u] >= 0) {
        float f[8];
        raw_to_float(prx_[u], f);
        if (has_skip) {
          float g[8];
          raw_to_float(prs_[u], g);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] += g[j];
        }
        *reinterpret_cast<f16x8*>(in_t + plds[u]) = to_h8(f);
        if (pc8[u] == 0) maskf[ppix[u]] = (pin >> u) & 1u ? 1.f : 0.f;
      }
    }
    __syncthreads();
    IR_T(0);
    if (tile + tstep < tend) prefetch(tile + tstep);
    IR_T(1);

    // ---- expand: h1[pixel][hidden] = relu6(W1 x + b1), zero outside the image (the depthwise padding)
    {
      f16x8 wa[C::HT][C::KS1];
#pragma unroll
      for (int mt = 0; mt < C::HT; ++mt)
#pragma unroll
        for (int ks = 0; ks < C::KS1; ++ks) wa[mt][ks] = *reinterpret_cast<const f16x8*>(wf1 + ((mt * C::KS1 + ks) * 64 + lane) * 8);
      for (int nt = wave; nt < T::NT1; nt += 4) {
        const int pi = 16 * nt + p;
        f16x8 xv[C::KS1];
#pragma unroll
        for (int ks = 0; ks < C::KS1; ++ks)
          xv[ks] = 32 * ks + 8 * q < C::KR ? *reinterpret_cast<const f16x8*>(in_t + pi * C::IP + 32 * ks + 8 * q) : f16x8{};
        const float m = maskf[pi];
        const bool inside = __all(m != 0.f);
#pragma unroll
        for (int mt = 0; mt < C::HT; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[mt][ks], xv[ks], acc, 0, 0, 0);
          const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * mt + 4 * q);
          f16x4 o;
          if (inside) {     // wave-uniform: all 16 pixels of the tile lie in the image (nearly every tile)
            o[0] = (_Float16)clamp01(acc[0] + bb.x);
            o[1] = (_Float16)clamp01(acc[1] + bb.y);
            o[2] = (_Float16)clamp01(acc[2] + bb.z);
            o[3] = (_Float16)clamp01(acc[3] + bb.w);
          } else {
            o[0] = (_Float16)(clamp01(acc[0] + bb.x) * m);
            o[1] = (_Float16)(clamp01(acc[1] + bb.y) * m);
            o[2] = (_Float16)(clamp01(acc[2] + bb.z) * m);
            o[3] = (_Float16)(clamp01(acc[3] + bb.w) * m);
          }
          *reinterpret_cast<f16x4*>(h1 + pi * C::HPI + 16 * mt + 4 * q) = o;   // HT*16 <= HPI: inside the row
        }
      }
    }
    __syncthreads();
    IR_T(2);

    f16x8 w3r[C::CT][C::KS3];
#pragma unroll
    for (int mt = 0; mt < C::CT; ++mt)
#pragma unroll
      for (int ks = 0; ks < C::KS3; ++ks) w3r[mt][ks] = *reinterpret_cast<const f16x8*>(wf3 + ((mt * C::KS3 + ks) * 64 + lane) * 8);
    const int oy0 = TR ? 2 * y0 : (y0 + 1) / 2, ox0 = TR ? 2 * x0 : (x0 + 1) / 2;
    char* ob = static_cast<char*>(a.out) + (size_t)b * a.Ho * a.Wo * COUT * (OUT_F32 ? 4 : 2);
#pragma unroll 1
    for (int pass = 0; pass < T::PASSES; ++pass) {
      if (pass > 0) __syncthreads();   // the previous pass's project has read h2
      // ---- depthwise 3x3 (stride 2) or its transpose on the hidden tile, + bias, ReLU6 (packed half arithmetic)
      if (pl < NPL) {
        if (!TR) {
          for (int po = pl; po < T::POUT; po += NPL) {
            const int oy = po >> T::LOW, ox = po & (T::OW - 1);
            const _Float16* src = h1 + ((2 * oy) * T::IW + 2 * ox) * C::HPI + 8 * cg;
            f16x8 acc = b2r;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx)
                acc = fma_h8(*reinterpret_cast<const f16x8*>(src + (ky * T::IW + kx) * C::HPI), wdr[3 * ky + kx], acc);
            *reinterpret_cast<f16x8*>(h2 + po * C::HPI + 8 * cg) = clamp01_h8(acc);
          }
        } else {
          // out[o] += in[i] * w[k] with o = 2i - 1 + k: even outputs take tap 1 of cell i, odd ones tap 2 of i and tap 0 of
          // i+1; this pass: the cells of row `pass`, i.e. output rows 2 pass and 2 pass + 1 (rows 0 and 1 of h2)
          for (int cx = pl; cx < T::TW; cx += NPL) {
            const _Float16* src = h1 + (pass * T::IW + cx) * C::HPI + 8 * cg;
            const f16x8 v00 = *reinterpret_cast<const f16x8*>(src), v01 = *reinterpret_cast<const f16x8*>(src + C::HPI);
            const f16x8 v10 = *reinterpret_cast<const f16x8*>(src + T::IW * C::HPI);
            const f16x8 v11 = *reinterpret_cast<const f16x8*>(src + (T::IW + 1) * C::HPI);
            const f16x8 o00 = fma_h8(v00, wdr[4], b2r);
            const f16x8 o01 = fma_h8(v01, wdr[3], fma_h8(v00, wdr[5], b2r));
            const f16x8 o10 = fma_h8(v10, wdr[1], fma_h8(v00, wdr[7], b2r));
            const f16x8 o11 = fma_h8(v11, wdr[0], fma_h8(v10, wdr[2], fma_h8(v01, wdr[6], fma_h8(v00, wdr[8], b2r))));
            _Float16* dst = h2 + (2 * cx) * C::HPI + 8 * cg;
            *reinterpret_cast<f16x8*>(dst) = clamp01_h8(o00);
            *reinterpret_cast<f16x8*>(dst + C::HPI) = clamp01_h8(o01);
            *reinterpret_cast<f16x8*>(dst + T::OW * C::HPI) = clamp01_h8(o10);
            *reinterpret_cast<f16x8*>(dst + (T::OW + 1) * C::HPI) = clamp01_h8(o11);
          }
        }
      }
      __syncthreads();
      IR_T(3);

      // ---- project: out[pixel][cout] = W3 h2 + b3 (no activation)
      for (int nt = wave; nt < T::PNT3; nt += 4) {
        const int po = 16 * nt + p;
        f16x8 hb[C::KS3];
#pragma unroll
        for (int ks = 0; ks < C::KS3; ++ks)
          hb[ks] = 32 * ks + 8 * q < C::HR ? *reinterpret_cast<const f16x8*>(h2 + po * C::HPI + 32 * ks + 8 * q) : f16x8{};
        const int oy = oy0 + (TR ? 2 * pass : 0) + (po >> T::LOW), ox = ox0 + (po & (T::OW - 1));
        const bool valid = oy < a.Ho && ox < a.Wo;
        const unsigned obase = (unsigned)((oy * a.Wo + ox) * COUT);
#pragma unroll
        for (int mt = 0; mt < C::CT; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3r[mt][ks], hb[ks], acc, 0, 0, 0);
          const int co = 16 * mt + 4 * q;
          if (valid && co < COUT) {
            if (OUT_F32) {
              *reinterpret_cast<float4*>(ob + 4ull * (obase + co)) =
                  make_float4(acc[0] + b3r[mt][0], acc[1] + b3r[mt][1], acc[2] + b3r[mt][2], acc[3] + b3r[mt][3]);
            } else {
              bf16x4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] = (__bf16)(acc[r] + b3r[mt][r]);
              *reinterpret_cast<bf16x4*>(ob + 2ull * (obase + co)) = o;
            }
          }
        }
      }
    }
    // the next iteration's first barrier separates this tile's reads of h2 / in_t from their next writes, unless h2 lives
    // in the input tile, which the next iteration's staging overwrites before that barrier
    if (C::ALIAS_H2) __syncthreads();
  }
}

template <int CIN, int HID, int COUT, bool TR, int TH, bool INF, bool OUTF>
int launch_ir(const IrArgs& a0, hipStream_t st) {
  using C = IrCfg<CIN, HID, COUT, TR, TH>;
  IrArgs a = a0;
  const int oh = TR ? 2 * TH : TH, ow = TR ? 32 : 16;
  a.tiles_y = (a.Ho + oh - 1) / oh;
  a.tiles_x = (a.Wo + ow - 1) / ow;
  const long ntiles = (long)a.B * a.tiles_y * a.tiles_x;
  if (ntiles >= (1L << 31)) {
    noise_set_error("noise_fused_ir_block: %ld tiles exceed the 32-bit tile index", ntiles);
    return -1;
  }
  const size_t smem = C::LDS_BYTES;
  auto kfn = ir_block_kernel<CIN, HID, COUT, TR, TH, INF, OUTF>;
  NOISE_ALLOW_LDS(kfn, smem);
  static int per_cu = 0;      // resident workgroups per CU of this instantiation (asked once)
  if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kfn, 256, smem) != hipSuccess || per_cu < 1)) per_cu = 1;
  const long cap = (long)noise_num_cus() * per_cu;
  hipLaunchKernelGGL(kfn, dim3((unsigned)noise_pad8(ntiles < cap ? ntiles : cap)), dim3(256), smem, st, a);
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Head: normalise + left/top pad (models.py:614-622, :589) -> first Conv2d 1->32 k3 s2 + BN + ReLU6 (:1151-1159, kept as
// skip tensor A0) -> block 1: depthwise 3x3 s1 + BN + ReLU6 -> 1x1 32->16 + BN (:385-396, skip tensor A1).
// The single-channel 3x3 convolution runs on the matrix cores as a K = 32 product: reduction slot k = 8q + 2t + s holds
// tap 4q + t of the input (q < 3, t < 4, taps past 8 unused), rounded to half for s = 0 and the rounding remainder for
// s = 1, so lane group q of the B operand touches only its own taps.  The input keeps ~22 significant bits that way;
// only the weights are rounded, like everywhere else.
struct HeadArgs {
  const float* logmag;    // [B][H][W]
  const float* absmax;    // [B]
  const _Float16* w0;     // [32][32]  rows = channels, k as described above
  const float* b0;        // [32]
  const _Float16* wd;     // [9][32]
  const _Float16* b2;     // [32]
  const _Float16* w3;     // [16][32]
  const float* b3;        // [16]
  __bf16* a0;             // [B][H1][W1][32]
  __bf16* a1;             // [B][H1][W1][16]
  int B, H, W, hp, wp, H1, W1, tiles_y, tiles_x;
  float pad_value;
};

constexpr int HD_OH = 12, HD_OW = 16;   // the A0 halo tile is then 14 x 18 = 252 pixels: 16 MFMA tiles, no idle round
constexpr int HD_AH = HD_OH + 2, HD_AW = HD_OW + 2, HD_PA = HD_AH * HD_AW, HD_NTA = (HD_PA + 15) / 16;   // A0 halo tile
constexpr int HD_XH = 2 * HD_AH + 1, HD_XW = 2 * HD_AW + 1, HD_XP = HD_XW + 1, HD_NX = HD_XH * HD_XW;     // input tile
constexpr int HD_C0 = 32, HD_C1 = 16, HD_CP = HD_C0 + 8;
constexpr int HD_POUT = HD_OH * HD_OW, HD_NT3 = HD_POUT / 16;

__global__ __launch_bounds__(256, 4) void head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[HD_XH * HD_XP];
  __shared__ __attribute__((aligned(16))) _Float16 h1[HD_NTA * 16 * HD_CP];   // A0 on the halo tile (zero outside the image)
  __shared__ __attribute__((aligned(16))) _Float16 h2[HD_POUT * HD_CP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
  const int cg = tid & 3, pl = tid >> 2;      // depthwise: 4 channel groups of 8, 64 pixel lanes
  // the depthwise taps (rows 0..8) and bias (row 9) sit in LDS: 36 registers less is a fourth workgroup per CU
  __shared__ __attribute__((aligned(16))) _Float16 wds[10 * HD_C0];
  if (tid < 36) *reinterpret_cast<f16x8*>(wds + 8 * tid) = *reinterpret_cast<const f16x8*>(a.wd + 8 * tid);
  else if (tid < 40) *reinterpret_cast<f16x8*>(wds + 8 * tid) = *reinterpret_cast<const f16x8*>(a.b2 + 8 * (tid - 36));
  f16x8 w0a[2], w3a;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) w0a[mt] = *reinterpret_cast<const f16x8*>(a.w0 + (16 * mt + p) * 32 + 8 * q);
  w3a = *reinterpret_cast<const f16x8*>(a.w3 + p * 32 + 8 * q);
  float b0r[2][4], b3r[4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b0r[mt][r] = a.b0[16 * mt + 4 * q + r];
#pragma unroll
  for (int r = 0; r < 4; ++r) b3r[r] = a.b3[4 * q + r];
  for (int i = tid; i < HD_NTA * 16 * HD_CP / 8; i += 256) reinterpret_cast<uint4*>(h1)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < HD_POUT * HD_CP / 8; i += 256) reinterpret_cast<uint4*>(h2)[i] = make_uint4(0, 0, 0, 0);

  int toff[4];                   // LDS offsets of this lane group's four taps of the 3x3 window (clamped to tap 8)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tap = min(4 * q + t, 8);
    toff[t] = (tap / 3) * HD_XP + tap % 3;
  }
  const int Hp = a.H + a.hp, Wp = a.W + a.wp;
  constexpr int NI = (HD_NX + 255) / 256;
  int pry[NI], prx[NI];
#pragma unroll
  for (int u = 0; u < NI; ++u) {
    const int i = tid + 256 * u;
    pry[u] = i / HD_XW;
    prx[u] = i - pry[u] * HD_XW;
  }
  // raw prefetch registers (nothing is computed from a loaded value before the next iteration: the loads stay in flight)
  float pre[NI], pre_amax = 0.f;
  unsigned freal = 0, finside = 0;
  const int ntiles = a.B * a.tiles_y * a.tiles_x;      // < 2^31 (checked by the launcher)
  int nb = 0, noy0 = 0, nox0 = 0;                      // origin of the prefetched tile
  auto prefetch = [&](int tile) {
    const unsigned tx = (unsigned)tile % (unsigned)a.tiles_x, r = (unsigned)tile / (unsigned)a.tiles_x;
    noy0 = (int)(r % (unsigned)a.tiles_y) * HD_OH;
    nox0 = (int)tx * HD_OW;
    nb = (int)(r / (unsigned)a.tiles_y);
    const int b = nb, oy0 = noy0, ox0 = nox0;
    pre_amax = a.absmax[b];
    const float* src = a.logmag + (size_t)b * a.H * a.W;
    freal = finside = 0;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int yy = 2 * oy0 - 3 + pry[u], xx = 2 * ox0 - 3 + prx[u];            // position in the padded image
      const bool inside = (unsigned)yy < (unsigned)Hp && (unsigned)xx < (unsigned)Wp;
      const bool real = inside && yy >= a.hp && xx >= a.wp;
      const int sy = real ? yy - a.hp : 0, sx = real ? xx - a.wp : 0;
      pre[u] = src[sy * a.W + sx];
      freal |= real ? (1u << u) : 0u;
      finside |= inside ? (1u << u) : 0u;
    }
  };
  // Neighbouring tiles share halo pixels and partial cache lines.  XCD j (workgroups j, j + 8, ...: one L2) owns the
  // contiguous tile range [j T8, (j + 1) T8) and its workgroups stride through it together, so the window of tiles in
  // flight on an L2 advances row by row through one clip after the other and the halo rows of a window are still in
  // that L2 when the next window needs them.
  const int T8 = (ntiles + 7) >> 3, tstep = (int)(gridDim.x >> 3);
  const int tend = min(ntiles, (int)((blockIdx.x & 7u) + 1) * T8);
  int tile = (int)(blockIdx.x & 7u) * T8 + (int)(blockIdx.x >> 3);
  if (tile < tend) prefetch(tile);
  __syncthreads();
#ifdef FUSED_STAMPS
  unsigned long long last_ = clock64();
#endif
  for (; tile < tend; tile += tstep) {
    const int b = nb, oy0 = noy0, ox0 = nox0;
    FU_T(13);
    {
      const float inv = 1.f / (pre_amax + 1e-6f);
#pragma unroll
      for (int u = 0; u < NI; ++u)
        if (tid + 256 * u < HD_NX)
          xs[pry[u] * HD_XP + prx[u]] = (freal >> u) & 1u ? pre[u] * inv : ((finside >> u) & 1u ? a.pad_value : 0.f);
    }
    __syncthreads();
    FU_T(8);
    if (tile + tstep < tend) prefetch(tile + tstep);
    FU_T(9);

    // ---- first conv on the A0 halo tile
    for (int nt = wave; nt < HD_NTA; nt += 4) {
      const int pi = 16 * nt + p;
      const int ry = pi / HD_AW, rx = pi - ry * HD_AW;
      const int ay = oy0 - 1 + ry, ax = ox0 - 1 + rx;                     // A0 pixel
      const bool valid = pi < HD_PA && (unsigned)ay < (unsigned)a.H1 && (unsigned)ax < (unsigned)a.W1;
      const float m = valid ? 1.f : 0.f;
      const float* xp = xs + (pi < HD_PA ? 2 * ry * HD_XP + 2 * rx : 0);
      // reduction slot 8q + 2t + s = tap 4q + t, rounded to half (s = 0) and its remainder (s = 1): the fp32 input keeps
      // ~22 bits through the half-precision MFMA; slots past tap 8 meet zero weights
      f16x8 xv;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float v = xp[toff[t]];
        const _Float16 hi = (_Float16)v;
        xv[2 * t] = hi;
        xv[2 * t + 1] = (_Float16)(v - (float)hi);
      }
      __bf16* a0p = a.a0 + (((size_t)b * a.H1 + ay) * a.W1 + ax) * HD_C0 + 4 * q;
      const bool interior = valid && ry >= 1 && ry <= HD_OH && rx >= 1 && rx <= HD_OW;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0a[mt], xv, acc, 0, 0, 0);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = clamp01(acc[r] + b0r[mt][r]) * m;
        f16x4 o;
        bf16x4 ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (_Float16)v[r];
          ob[r] = (__bf16)(6.f * v[r]);      // the skip tensor holds the activation itself
        }
        *reinterpret_cast<f16x4*>(h1 + pi * HD_CP + 16 * mt + 4 * q) = o;
        if (interior) *reinterpret_cast<bf16x4*>(a0p + 16 * mt) = ob;
      }
    }
    __syncthreads();
    FU_T(10);
    // ---- depthwise 3x3 stride 1 on A0
    for (int po = pl; po < HD_POUT; po += 64) {
      const int oy = po >> 4, ox = po & 15;
      const _Float16* src = h1 + (oy * HD_AW + ox) * HD_CP + 8 * cg;
      f16x8 acc = *reinterpret_cast<const f16x8*>(wds + 9 * HD_C0 + 8 * cg);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc = fma_h8(*reinterpret_cast<const f16x8*>(src + (ky * HD_AW + kx) * HD_CP),
                       *reinterpret_cast<const f16x8*>(wds + (3 * ky + kx) * HD_C0 + 8 * cg), acc);
      *reinterpret_cast<f16x8*>(h2 + po * HD_CP + 8 * cg) = clamp01_h8(acc);
    }
    __syncthreads();
    FU_T(11);
    // ---- project 32 -> 16
    for (int nt = wave; nt < HD_NT3; nt += 4) {
      const int po = 16 * nt + p;
      const f16x8 hb = *reinterpret_cast<const f16x8*>(h2 + po * HD_CP + 8 * q);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3a, hb, acc, 0, 0, 0);
      const int oy = oy0 + (po >> 4), ox = ox0 + (po & 15);
      if (oy < a.H1 && ox < a.W1) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)(acc[r] + b3r[r]);
        *reinterpret_cast<bf16x4*>(a.a1 + (((size_t)b * a.H1 + oy) * a.W1 + ox) * HD_C1 + 4 * q) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Tail: s = D2 + A1 (cropped skip) -> block 9: depthwise ConvTranspose2d 3x3 s1 + BN + ReLU6 -> 1x1 16->32 + BN
// (models.py:1193-1201) -> u = D3 + A0 (cropped skip) -> last ConvTranspose2d 32->1 k3 s2 with bias (:1164-1171) ->
// crop of the last H x W outputs (:610) -> enh = clamp(logmag/max + filt, -1, 1) * max (:762-763).
// The 32->1 transposed convolution is a product with M = 9 taps: P[tap][pixel] = sum_c w[c][tap] u[pixel][c]; an output
// sample then adds the 1, 2 or 4 entries of P that reach it.
struct TailArgs {
  const __bf16* d2;       // [B][h][w][16]
  const __bf16* a1;       // [B][Hs1][Ws1][16], read at (+oy1, +ox1)
  const __bf16* a0;       // [B][Hs0][Ws0][32], read at (+oy0, +ox0)
  const _Float16* wd;     // [9][16]
  const _Float16* b2;     // [16]
  const _Float16* w3;     // [32][32]  (k >= 16 zero)
  const float* b3;        // [32]
  const _Float16* wl;     // [16][32]  rows = taps (9 used), k = channel
  const float* logmag;    // [B][H][W]
  const float* absmax;    // [B]
  float* enh;             // [B][H][W]   (optional)
  float* filt;            // [B][H][W]   (optional: the raw network output)
  int B, h, w, Hs1, Ws1, oy1, ox1, Hs0, Ws0, oy0, ox0, H, W, offy, offx, tiles_y, tiles_x;
  float bias;
};

constexpr int TL_CH = 7, TL_CW = 15;   // the u region is then 8 x 16 = 128 pixels: one depthwise round, eight 16-pixel tiles
constexpr int TL_UH = TL_CH + 1, TL_UW = TL_CW + 1, TL_PU = TL_UH * TL_UW, TL_NTU = (TL_PU + 15) / 16;   // u region
constexpr int TL_SH = TL_CH + 3, TL_SW = TL_CW + 3, TL_PS = TL_SH * TL_SW;                                // s region
constexpr int TL_CP = 32 + 8, TL_SP = 16 + 8;

__global__ __launch_bounds__(256, 4) void tail_kernel(TailArgs a) {
  __shared__ __attribute__((aligned(16))) _Float16 st[TL_PS * TL_SP];          // s = D2 + A1 (zero outside the image)
  __shared__ __attribute__((aligned(16))) __bf16 a0t[TL_NTU * 16 * TL_CP];     // A0 on the u region
  // depthwise output (channels 0..15; the 1x1 weights of channels 16..31 are zero) and, in the same rows once a wave has
  // read them, u = D3 + A0 (zero outside the image): a wave reads its 16 pixels of h2 before it writes their u
  __shared__ __attribute__((aligned(16))) _Float16 h2[TL_NTU * 16 * TL_CP];
  _Float16* const ut = h2;
  __shared__ __attribute__((aligned(16))) float pt[TL_NTU * 16 * 12];          // P[pixel][tap]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
#ifdef FUSED_STAMPS
  unsigned long long last_ = clock64();
#endif
  const int cg = tid & 1, pl = tid >> 1;      // depthwise: 2 channel groups of 8, 128 pixel lanes
  // the depthwise taps sit in LDS: a thread needs them for one pixel per tile, and 36 registers less is a fourth workgroup
  __shared__ __attribute__((aligned(16))) _Float16 wds[10 * 16];
  if (tid < 18) *reinterpret_cast<f16x8*>(wds + 8 * tid) = *reinterpret_cast<const f16x8*>(a.wd + 8 * tid);
  else if (tid < 20) *reinterpret_cast<f16x8*>(wds + 8 * tid) = *reinterpret_cast<const f16x8*>(a.b2 + 8 * (tid - 18));   // row 9: the bias
  f16x8 w3a[2], wla;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) w3a[mt] = *reinterpret_cast<const f16x8*>(a.w3 + (16 * mt + p) * 32 + 8 * q);
  wla = *reinterpret_cast<const f16x8*>(a.wl + p * 32 + 8 * q);
  float b3r[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b3r[mt][r] = a.b3[16 * mt + 4 * q + r];
  for (int i = tid; i < TL_NTU * 16 * TL_CP / 8; i += 256) {
    reinterpret_cast<uint4*>(a0t)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(h2)[i] = make_uint4(0, 0, 0, 0);
  }

  constexpr int PCS = TL_PS * 2, NIS = (PCS + 255) / 256;        // s pieces (8 channels)
  constexpr int PCA = TL_PU * 4, NIA = (PCA + 255) / 256;        // A0 pieces
  int sry[NIS], srx[NIS], sld[NIS], ary[NIA], arx[NIA], ald[NIA];
#pragma unroll
  for (int u = 0; u < NIS; ++u) {
    const int it = tid + 256 * u, pi = it >> 1;
    sry[u] = pi / TL_SW;
    srx[u] = pi - sry[u] * TL_SW;
    sld[u] = it < PCS ? pi * TL_SP + 8 * (it & 1) : -1;
  }
#pragma unroll
  for (int u = 0; u < NIA; ++u) {
    const int it = tid + 256 * u, pi = it >> 2;
    ary[u] = pi / TL_UW;
    arx[u] = pi - ary[u] * TL_UW;
    ald[u] = it < PCA ? pi * TL_CP + 8 * (it & 3) : -1;
  }
  uint4 prd[NIS], prs[NIS];     // raw prefetch registers: D2 and A1 pieces (8 bf16 each), summed when they are stored
  unsigned sin = 0;
  bf16x8 prea[NIA];
  constexpr int NOUT = 4 * TL_CH * TL_CW, NIO = (NOUT + 255) / 256;       // outputs of a tile, per thread
  float plm[NIO], pmx = 1.f;    // the tile's log-magnitudes and the clip's normaliser, requested with the tile's inputs
  const int ntiles = a.B * a.tiles_y * a.tiles_x;      // < 2^31 (checked by the launcher)
  int nb = 0, ncy0 = 0, ncx0 = 0;                      // origin of the prefetched tile
  auto prefetch = [&](int tile) {
    const unsigned tx = (unsigned)tile % (unsigned)a.tiles_x, r = (unsigned)tile / (unsigned)a.tiles_x;
    ncy0 = (int)(r % (unsigned)a.tiles_y) * TL_CH;
    ncx0 = (int)tx * TL_CW;
    nb = (int)(r / (unsigned)a.tiles_y);
    const int b = nb, cy0 = ncy0, cx0 = ncx0;
    const __bf16* d2b = a.d2 + (size_t)b * a.h * a.w * 16;
    const __bf16* a1b = a.a1 + ((size_t)b * a.Hs1 * a.Ws1 + (size_t)a.oy1 * a.Ws1 + a.ox1) * 16;
    const __bf16* a0b = a.a0 + ((size_t)b * a.Hs0 * a.Ws0 + (size_t)a.oy0 * a.Ws0 + a.ox0) * 32;
    sin = 0;
#pragma unroll
    for (int u = 0; u < NIS; ++u) {
      const int c8 = 8 * ((tid + 256 * u) & 1);
      const int iy = cy0 - 1 + sry[u], ix = cx0 - 1 + srx[u];
      const bool in = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
      const int yc = in ? iy : 0, xc = in ? ix : 0;
      prd[u] = *reinterpret_cast<const uint4*>(d2b + (yc * a.w + xc) * 16 + c8);
      prs[u] = *reinterpret_cast<const uint4*>(a1b + (yc * a.Ws1 + xc) * 16 + c8);
      sin |= in ? (1u << u) : 0u;
    }
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
      const int c8 = 8 * ((tid + 256 * u) & 3);
      int iy = cy0 + ary[u], ix = cx0 + arx[u];
      iy = iy < a.h ? iy : a.h - 1;
      ix = ix < a.w ? ix : a.w - 1;
      prea[u] = *reinterpret_cast<const bf16x8*>(a0b + (iy * a.Ws0 + ix) * 32 + c8);
    }
    if (a.enh) {
      pmx = a.absmax[b];
#pragma unroll
      for (int u = 0; u < NIO; ++u) {
        const int oi = min(tid + 256 * u, NOUT - 1), Y = oi / (2 * TL_CW), X = oi - Y * (2 * TL_CW);
        int y = 2 * cy0 + Y - a.offy, x = 2 * cx0 + X - a.offx;
        y = min(max(y, 0), a.H - 1);
        x = min(max(x, 0), a.W - 1);
        plm[u] = a.logmag[((size_t)b * a.H + y) * a.W + x];
      }
    }
  };
  // Neighbouring tiles share halo pixels and partial cache lines.  XCD j (workgroups j, j + 8, ...: one L2) owns the
  // contiguous tile range [j T8, (j + 1) T8) and its workgroups stride through it together, so the window of tiles in
  // flight on an L2 advances row by row through one clip after the other and the halo rows of a window are still in
  // that L2 when the next window needs them.
  const int T8 = (ntiles + 7) >> 3, tstep = (int)(gridDim.x >> 3);
  const int tend = min(ntiles, (int)((blockIdx.x & 7u) + 1) * T8);
  int tile = (int)(blockIdx.x & 7u) * T8 + (int)(blockIdx.x >> 3);
  if (tile < tend) prefetch(tile);
  __syncthreads();
  FU_T(0);
  for (; tile < tend; tile += tstep) {
    const int b = nb, cy0 = ncy0, cx0 = ncx0;
    FU_T(6);
#pragma unroll
    for (int u = 0; u < NIS; ++u)
      if (sld[u] >= 0) {
        const float m = (sin >> u) & 1u ? 1.f : 0.f;
        const unsigned dw[4] = {prd[u].x, prd[u].y, prd[u].z, prd[u].w}, sw[4] = {prs[u].x, prs[u].y, prs[u].z, prs[u].w};
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[2 * j] = (_Float16)((bf_lo(dw[j]) + bf_lo(sw[j])) * m);
          v[2 * j + 1] = (_Float16)((bf_hi(dw[j]) + bf_hi(sw[j])) * m);
        }
        *reinterpret_cast<f16x8*>(st + sld[u]) = v;
      }
#pragma unroll
    for (int u = 0; u < NIA; ++u)
      if (ald[u] >= 0) *reinterpret_cast<bf16x8*>(a0t + ald[u]) = prea[u];
    float clm[NIO];
#pragma unroll
    for (int u = 0; u < NIO; ++u) clm[u] = plm[u];
    const float cmx = pmx + 1e-6f;
    __syncthreads();
    FU_T(1);
    if (tile + tstep < tend) prefetch(tile + tstep);
    FU_T(2);

    // ---- depthwise ConvTranspose2d 3x3 stride 1: out[o] = sum_k s[o + 1 - k] w[k]
    for (int pu = pl; pu < TL_PU; pu += 128) {
      const int uy = pu / TL_UW, ux = pu - uy * TL_UW;      // u pixel = region origin + (uy, ux); s index = +1
      const _Float16* src = st + ((uy + 2) * TL_SW + ux + 2) * TL_SP + 8 * cg;
      f16x8 acc = *reinterpret_cast<const f16x8*>(wds + 144 + 8 * cg);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
          acc = fma_h8(*reinterpret_cast<const f16x8*>(src - (ky * TL_SW + kx) * TL_SP),
                       *reinterpret_cast<const f16x8*>(wds + (3 * ky + kx) * 16 + 8 * cg), acc);
      *reinterpret_cast<f16x8*>(h2 + pu * TL_CP + 8 * cg) = clamp01_h8(acc);
    }
    __syncthreads();
    FU_T(3);
    // ---- 1x1 16 -> 32, + A0: u (zero outside the image: the transposed convolution only sees real pixels)
    for (int nt = wave; nt < TL_NTU; nt += 4) {
      const int pu = 16 * nt + p;
      const f16x8 hb = *reinterpret_cast<const f16x8*>(h2 + pu * TL_CP + 8 * q);
      const int uy = pu / TL_UW, ux = pu - uy * TL_UW;
      const float m = (pu < TL_PU && cy0 + uy < a.h && cx0 + ux < a.w) ? 1.f : 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3a[mt], hb, acc, 0, 0, 0);
        const bf16x4 sk = *reinterpret_cast<const bf16x4*>(a0t + pu * TL_CP + 16 * mt + 4 * q);
        f16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (_Float16)((acc[r] + b3r[mt][r] + (float)sk[r]) * m);
        *reinterpret_cast<f16x4*>(ut + pu * TL_CP + 16 * mt + 4 * q) = o;
      }
    }
    __syncthreads();
    FU_T(4);
    // ---- P[pixel][tap]
    for (int nt = wave; nt < TL_NTU; nt += 4) {
      const int pu = 16 * nt + p;
      const f16x8 ub = *reinterpret_cast<const f16x8*>(ut + pu * TL_CP + 8 * q);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wla, ub, acc, 0, 0, 0);
      if (q < 3) *reinterpret_cast<float4*>(pt + pu * 12 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    FU_T(5);
    // ---- output samples: the 2CH x 2CW outputs of the tile, 256 at a time
#pragma unroll
    for (int u = 0; u < NIO; ++u) {
      const bool live = tid + 256 * u < NOUT;
      const int oi = live ? tid + 256 * u : 0;
      const int Y = oi / (2 * TL_CW), X = oi - Y * (2 * TL_CW);
      const int cy = Y >> 1, dy = Y & 1, cx = X >> 1, dx = X & 1;
      float f = a.bias;
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          // contribution of cell (cy + aa, cx + bb): tap k = o - 2i + 1 per axis
          const int ky = dy == 0 ? 1 : (aa == 0 ? 2 : 0), kx = dx == 0 ? 1 : (bb == 0 ? 2 : 0);
          const bool on = (dy == 1 || aa == 0) && (dx == 1 || bb == 0);
          const float v = pt[((cy + aa) * TL_UW + cx + bb) * 12 + 3 * ky + kx];
          f += on ? v : 0.f;
        }
      const int y = 2 * cy0 + Y - a.offy, x = 2 * cx0 + X - a.offx;
      if (live && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W) {
        const size_t o = ((size_t)b * a.H + y) * a.W + x;
        if (a.filt) a.filt[o] = f;
        if (a.enh) a.enh[o] = fminf(fmaxf(clm[u] / cmx + f, -1.f), 1.f) * cmx;
      }
    }
    // the next iteration's first barrier separates these reads of pt from the next writes (pt is written after 3 more)
  }
}

}  // namespace

#ifdef FUSED_STAMPS
NOISE_EXPORT int noise_debug_fused_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[32] = {0};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fused_stamps), z, sizeof(z));
  }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_stamps), 32 * sizeof(unsigned long long));
}
#endif

// Padded weight shapes the fused kernels expect (so that a binding can size its buffers); "half" = IEEE binary16:
//   w1 [ceil(Hid/16)*16][roundup(Cin,32)] half, b1 [ceil(Hid/16)*16] f32, wd [9][ceil(Hid/8)*8] half (tap-major),
//   b2 [ceil(Hid/8)*8] half, w3 [ceil(Cout/16)*16][roundup(Hid,32)] half, b3 [ceil(Cout/16)*16] f32.
NOISE_EXPORT int noise_fused_ir_block(const void* x, const void* skip, int in_f32, int Hs, int Ws, int soy, int sox,
                                      const void* w1, const float* b1, const void* wd, const void* b2, const void* w3,
                                      const float* b3, void* out, int out_f32, int transposed, int B, int Hi, int Wi,
                                      int Cin, int Hid, int Cout, void* stream) {
  NOISE_REQUIRE(x && w1 && b1 && wd && b2 && w3 && b3 && out, "noise_fused_ir_block: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_fused_ir_block: empty input");
  NOISE_REQUIRE((long)Hi * Wi * (Cin > Cout ? Cin : 4 * Cout) < (1L << 30), "noise_fused_ir_block: image too large");
  IrArgs a{};
  a.x = x; a.skip = skip; a.w1 = (const _Float16*)w1; a.b1 = b1; a.wd = (const _Float16*)wd; a.b2 = (const _Float16*)b2;
  a.w3 = (const _Float16*)w3; a.b3 = b3;
  a.out = out; a.B = B; a.Hi = Hi; a.Wi = Wi;
  a.Ho = transposed ? 2 * Hi - 1 : (Hi - 1) / 2 + 1;
  a.Wo = transposed ? 2 * Wi - 1 : (Wi - 1) / 2 + 1;
  a.Hs = Hs; a.Ws = Ws; a.soy = soy; a.sox = sox;
  if (skip) NOISE_REQUIRE(soy >= 0 && sox >= 0 && soy + Hi <= Hs && sox + Wi <= Ws,
                          "noise_fused_ir_block: skip window %dx%d at (%d,%d) leaves the %dx%d skip tensor", Hi, Wi, soy, sox, Hs, Ws);
  hipStream_t st = (hipStream_t)stream;
  bool done = false;
#define IR_CASE(CI, HI, CO, TRV, THV, INF, OUTF)                                                                           \
  if (!done && Cin == CI && Hid == HI && Cout == CO && (transposed != 0) == TRV) {                                         \
    NOISE_REQUIRE((in_f32 != 0) == INF && (out_f32 != 0) == OUTF,                                                          \
                  "noise_fused_ir_block: the %d -> %d -> %d block is served with %s input and %s output", CI, HI, CO,      \
                  INF ? "fp32" : "bf16", OUTF ? "fp32" : "bf16");                                                          \
    if (launch_ir<CI, HI, CO, TRV, THV, INF, OUTF>(a, st)) return -1;                                                      \
    done = true;                                                                                                           \
  }
  // the inverted-residual blocks of train/configs/spectralv2.yaml:77-84 (t = 3): encoder 16->24, 24->32, 32->64 (its
  // output feeds the fp32 attention kernels) and the mirrored decoder 64->32 (fed by them), 32->24, 24->16
  IR_CASE(16, 48, 24, false, 2, false, false)
  IR_CASE(24, 72, 32, false, 2, false, false)
  IR_CASE(32, 96, 64, false, 2, false, true)
  IR_CASE(64, 192, 32, true, 2, true, false)
  IR_CASE(32, 96, 24, true, 2, false, false)
  IR_CASE(24, 72, 16, true, 4, false, false)
#undef IR_CASE
  NOISE_REQUIRE(done, "noise_fused_ir_block: no fused kernel for %d -> %d -> %d (%s); use the per-layer fp32 path", Cin, Hid,
                Cout, transposed ? "transposed" : "strided");
  NOISE_CHECK_LAUNCH("noise_fused_ir_block");
  return 0;
}

// Head of the fused network (see head_kernel).  w0 [32][32] half: row c holds tap 4q + t of output channel c at
// k = 8q + 2t and again at k = 8q + 2t + 1 (q < 3, t < 4, 4q + t <= 8), zeros elsewhere; wd [9][32] half, b2 [32] half,
// w3 [16][32] half.  a0 [B][H1][W1][32], a1 [B][H1][W1][16] bf16 with H1 = (H + hp - 1)/2 + 1, W1 = (W + wp - 1)/2 + 1.
NOISE_EXPORT int noise_fused_head(const float* logmag, const float* absmax, const void* w0, const float* b0,
                                  const void* wd, const void* b2, const void* w3, const float* b3, void* a0, void* a1,
                                  int B, int H, int W, int hp, int wp, float pad_value, void* stream) {
  NOISE_REQUIRE(logmag && absmax && w0 && b0 && wd && b2 && w3 && b3 && a0 && a1, "noise_fused_head: null pointer");
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && hp >= 0 && wp >= 0, "noise_fused_head: bad shape");
  NOISE_REQUIRE((long)H * W < (1L << 30), "noise_fused_head: image too large");
  HeadArgs a{};
  a.logmag = logmag; a.absmax = absmax; a.w0 = (const _Float16*)w0; a.b0 = b0; a.wd = (const _Float16*)wd;
  a.b2 = (const _Float16*)b2; a.w3 = (const _Float16*)w3; a.b3 = b3; a.a0 = (__bf16*)a0; a.a1 = (__bf16*)a1;
  a.B = B; a.H = H; a.W = W; a.hp = hp; a.wp = wp; a.pad_value = pad_value;
  a.H1 = (H + hp - 1) / 2 + 1;
  a.W1 = (W + wp - 1) / 2 + 1;
  a.tiles_y = noise_cdiv(a.H1, HD_OH);
  a.tiles_x = noise_cdiv(a.W1, HD_OW);
  const long ntiles = (long)B * a.tiles_y * a.tiles_x;
  NOISE_REQUIRE(ntiles < (1L << 31), "noise_fused_head: %ld tiles exceed the 32-bit tile index", ntiles);
  static int per_cu = 0;
  if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, head_kernel, 256, 0) != hipSuccess || per_cu < 1)) per_cu = 1;
  const long cap = (long)noise_num_cus() * per_cu;
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)noise_pad8(ntiles < cap ? ntiles : cap)), dim3(256), 0, (hipStream_t)stream, a);
  NOISE_CHECK_LAUNCH("noise_fused_head");
  return 0;
}

// Tail of the fused network (see tail_kernel).  d2 [B][h][w][16]; a1 / a0: the two skip tensors with their own sizes and
// crop offsets; wd [9][16] half, b2 [16] half, w3 [32][32] half (columns >= 16 zero), wl [16][32] half (row = tap of the
// last transposed convolution, column = channel), bias its scalar bias.  enh / filt [B][H][W] (either may be NULL).
NOISE_EXPORT int noise_fused_tail(const void* d2, const void* a1, int Hs1, int Ws1, int oy1, int ox1, const void* a0,
                                  int Hs0, int Ws0, int oy0, int ox0, const void* wd, const void* b2, const void* w3,
                                  const float* b3, const void* wl, float bias, const float* logmag, const float* absmax,
                                  float* enh, float* filt, int B, int h, int w, int H, int W, void* stream) {
  NOISE_REQUIRE(d2 && a1 && a0 && wd && b2 && w3 && b3 && wl && (enh || filt), "noise_fused_tail: null pointer");
  NOISE_REQUIRE(enh == nullptr || (logmag && absmax), "noise_fused_tail: enh needs logmag and absmax");
  NOISE_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "noise_fused_tail: bad shape");
  NOISE_REQUIRE(2 * h - 1 >= H && 2 * w - 1 >= W, "noise_fused_tail: network output %dx%d is smaller than its input %dx%d",
                2 * h - 1, 2 * w - 1, H, W);
  NOISE_REQUIRE(oy1 >= 0 && ox1 >= 0 && oy1 + h <= Hs1 && ox1 + w <= Ws1 && oy0 >= 0 && ox0 >= 0 && oy0 + h <= Hs0 && ox0 + w <= Ws0,
                "noise_fused_tail: a skip window leaves its tensor");
  NOISE_REQUIRE((long)Hs0 * Ws0 * 32 < (1L << 30), "noise_fused_tail: image too large");
  TailArgs a{};
  a.d2 = (const __bf16*)d2; a.a1 = (const __bf16*)a1; a.a0 = (const __bf16*)a0; a.wd = (const _Float16*)wd;
  a.b2 = (const _Float16*)b2; a.w3 = (const _Float16*)w3; a.b3 = b3; a.wl = (const _Float16*)wl; a.bias = bias;
  a.logmag = logmag; a.absmax = absmax;
  a.enh = enh; a.filt = filt; a.B = B; a.h = h; a.w = w; a.Hs1 = Hs1; a.Ws1 = Ws1; a.oy1 = oy1; a.ox1 = ox1;
  a.Hs0 = Hs0; a.Ws0 = Ws0; a.oy0 = oy0; a.ox0 = ox0; a.H = H; a.W = W;
  a.offy = 2 * h - 1 - H;
  a.offx = 2 * w - 1 - W;
  a.tiles_y = noise_cdiv(h, TL_CH);
  a.tiles_x = noise_cdiv(w, TL_CW);
  const long ntiles = (long)B * a.tiles_y * a.tiles_x;
  NOISE_REQUIRE(ntiles < (1L << 31), "noise_fused_tail: %ld tiles exceed the 32-bit tile index", ntiles);
  static int per_cu = 0;
  if (per_cu == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tail_kernel, 256, 0) != hipSuccess || per_cu < 1)) per_cu = 1;
  const long cap = (long)noise_num_cus() * per_cu;
  hipLaunchKernelGGL(tail_kernel, dim3((unsigned)noise_pad8(ntiles < cap ? ntiles : cap)), dim3(256), 0, (hipStream_t)stream, a);
  NOISE_CHECK_LAUNCH("noise_fused_tail");
  return 0;
}
