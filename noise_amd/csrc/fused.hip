// Block-fused inference kernels of the denoiser U-Net for MI355X (gfx950): bf16 storage, fp32 accumulation.
//
// Eval-mode forward of train/models.py:348-407 (InvertedResidual), :299-345 (its transposed depthwise variant),
// :1151-1173 (first conv / last transposed conv) and :583-612 (Base._forward: padding, skip adds with the
// crop-from-the-end of :608, final crop :610) with BatchNorm folded into the weights.  One kernel per
// inverted-residual block: a workgroup owns a spatial tile, expands it (1x1, bf16 MFMA), runs the 3x3 depthwise
// (transposed) convolution on the LDS-resident hidden tile (VALU, fp32) and projects it (1x1, bf16 MFMA), so the
// t-times wider hidden tensors never reach HBM.  Activations between blocks are NHWC bf16.
//
// MFMA arrangement (v_mfma_f32_16x16x32_bf16): A = weights (rows = output channels), B = activations (columns =
// 16 pixels), so a lane ends up with 4 consecutive output channels of one pixel: one 8-byte bf16 store.  LDS tiles
// are [pixel][channels + 8] bf16: the B fragment of a lane is 16 contiguous bytes and the odd 16-byte pitch spreads
// the 16 pixels of a fragment over the banks.
#include "common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int rup(int a, int b) { return (a + b - 1) / b * b; }

__device__ __forceinline__ float relu6_(float v) { return fminf(fmaxf(v, 0.f), 6.f); }

__device__ __forceinline__ void unpack8(const bf16x8 v, float (&f)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)v[j];
}

__device__ __forceinline__ bf16x8 pack8(const float (&f)[8]) {
  bf16x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (__bf16)f[j];
  return v;
}

// 8 consecutive channels of one pixel from a bf16 or fp32 NHWC tensor, as floats
template <bool F32>
__device__ __forceinline__ void load8(const void* base, size_t elem_off, float (&f)[8]) {
  if (F32) {
    const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem_off);
    const float4 b = *reinterpret_cast<const float4*>(static_cast<const float*>(base) + elem_off + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack8(*reinterpret_cast<const bf16x8*>(static_cast<const __bf16*>(base) + elem_off), f);
  }
}

struct IrArgs {
  const void* x;       // [B][Hi][Wi][CIN]
  const void* skip;    // optional [B][Hs][Ws][CIN], added to x at rows soy.., columns sox.. (same element type as x)
  const __bf16* w1;    // [HT*16][KP]  expand weights, BatchNorm scale folded, zero padded
  const float* b1;     // [HT*16]
  const float* wd;     // [9][G*8]     depthwise weights (tap-major), BatchNorm scale folded
  const float* b2;     // [G*8]
  const __bf16* w3;    // [CT*16][HP]  projection weights, BatchNorm scale folded, zero padded
  const float* b3;     // [CT*16]
  void* out;           // [B][Ho][Wo][COUT]
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
  static constexpr int OH = TR ? 2 * TH : TH, OW = TR ? 2 * TW : TW;
  static constexpr int POUT = OH * OW, NT3 = POUT / 16;
};

template <int CIN, int HID, int COUT, bool TR, int TH>
struct IrCfg {
  using G_ = TileGeom<TR, TH>;
  static constexpr int KP = rup(CIN, 32), KS1 = KP / 32, IP = KP + 8;
  static constexpr int HT = (HID + 15) / 16, HP = rup(HID, 32), KS3 = HP / 32, HPI = HP + 8, G = (HID + 7) / 8;
  static constexpr int CT = (COUT + 15) / 16;
  static constexpr int LDS_IN = G_::NT1 * 16 * IP, LDS_H1 = G_::NT1 * 16 * HPI, LDS_H2 = G_::POUT * HPI;   // bf16 elements
  static constexpr int LDS_W1 = HT * KS1 * 64 * 8, LDS_W3 = CT * KS3 * 64 * 8;                             // bf16 elements
  static constexpr int LDS_F = HT * 16 + CT * 16;                                                          // floats (b1, b3)
  static constexpr size_t LDS_BYTES = (size_t)(LDS_IN + LDS_H1 + LDS_H2 + LDS_W1 + LDS_W3) * 2 + (size_t)LDS_F * 4;
};

template <int CIN, int HID, int COUT, bool TR, int TH, bool IN_F32, bool OUT_F32>
__global__ __launch_bounds__(256) void ir_block_kernel(IrArgs a) {
  using C = IrCfg<CIN, HID, COUT, TR, TH>;
  using T = TileGeom<TR, TH>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __bf16* in_t = reinterpret_cast<__bf16*>(lds_raw);   // [NT1*16][IP]
  __bf16* h1 = in_t + C::LDS_IN;                        // [NT1*16][HPI]  expand output (zero outside the image)
  __bf16* h2 = h1 + C::LDS_H1;                          // [POUT][HPI]    depthwise output
  __bf16* wf1 = h2 + C::LDS_H2;                         // [HT][KS1][64][8] A fragments of the expand weights
  __bf16* wf3 = wf1 + C::LDS_W1;                        // [CT][KS3][64][8]
  float* b1s = reinterpret_cast<float*>(wf3 + C::LDS_W3);
  float* b3s = b1s + C::HT * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;

  // ---- one-time set-up: zero the tiles (padding channels / pixels stay zero for ever), stage weight fragments
  for (int i = tid; i < (C::LDS_IN + C::LDS_H1 + C::LDS_H2) / 8; i += 256)
    reinterpret_cast<uint4*>(lds_raw)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < C::HT * C::KS1 * 64; i += 256) {
    const int l = i & 63, f = i >> 6, ks = f % C::KS1, mt = f / C::KS1;
    *reinterpret_cast<bf16x8*>(wf1 + i * 8) =
        *reinterpret_cast<const bf16x8*>(a.w1 + (size_t)(16 * mt + (l & 15)) * C::KP + 32 * ks + 8 * (l >> 4));
  }
  for (int i = tid; i < C::CT * C::KS3 * 64; i += 256) {
    const int l = i & 63, f = i >> 6, ks = f % C::KS3, mt = f / C::KS3;
    *reinterpret_cast<bf16x8*>(wf3 + i * 8) =
        *reinterpret_cast<const bf16x8*>(a.w3 + (size_t)(16 * mt + (l & 15)) * C::HP + 32 * ks + 8 * (l >> 4));
  }
  for (int i = tid; i < C::HT * 16; i += 256) b1s[i] = a.b1[i];
  for (int i = tid; i < C::CT * 16; i += 256) b3s[i] = a.b3[i];
  // depthwise weights of this thread's channel group stay in registers: thread = (channel group, pixel lane)
  constexpr int NPL = 256 / C::G;
  const int cg = tid % C::G, pl = tid / C::G;
  float wdr[9][8], b2r[8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wdr[k][j] = a.wd[k * (C::G * 8) + 8 * cg + j];
#pragma unroll
  for (int j = 0; j < 8; ++j) b2r[j] = a.b2[8 * cg + j];

  // ---- input tile loader: pieces of 8 channels, prefetched into registers one tile ahead
  constexpr int PIECES = T::PIN * (CIN / 8), NI = (PIECES + 255) / 256;
  float pre[NI][8];
  const long ntiles = (long)a.B * a.tiles_y * a.tiles_x;
  auto tile_origin = [&](long tile, int& b, int& y0, int& x0) {
    const int tx = (int)(tile % a.tiles_x);
    const long r = tile / a.tiles_x;
    const int ty = (int)(r % a.tiles_y);
    b = (int)(r / a.tiles_y);
    y0 = TR ? ty * TH : 2 * ty * TH - 1;          // image row / column of the region's first input pixel
    x0 = TR ? tx * T::TW : 2 * tx * T::TW - 1;
  };
  auto prefetch = [&](long tile) {
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int it = tid + 256 * u;
      const int pi = it / (CIN / 8), c8 = it - pi * (CIN / 8);
      const int ry = pi / T::IW, rx = pi - ry * T::IW;
      int iy = y0 + ry, ix = x0 + rx;
      iy = iy < 0 ? 0 : (iy >= a.Hi ? a.Hi - 1 : iy);     // clamped: out-of-image pixels are masked after the expand
      ix = ix < 0 ? 0 : (ix >= a.Wi ? a.Wi - 1 : ix);
      load8<IN_F32>(a.x, (((size_t)b * a.Hi + iy) * a.Wi + ix) * CIN + 8 * c8, pre[u]);
      if (a.skip != nullptr) {
        float s[8];
        load8<IN_F32>(a.skip, (((size_t)b * a.Hs + iy + a.soy) * a.Ws + ix + a.sox) * CIN + 8 * c8, s);
#pragma unroll
        for (int j = 0; j < 8; ++j) pre[u][j] += s[j];
      }
    }
  };

  long tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int b, y0, x0;
    tile_origin(tile, b, y0, x0);
    // ---- registers -> input tile
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int it = tid + 256 * u;
      if (it < PIECES) {
        const int pi = it / (CIN / 8), c8 = it - pi * (CIN / 8);
        *reinterpret_cast<bf16x8*>(in_t + pi * C::IP + 8 * c8) = pack8(pre[u]);
      }
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);

    // ---- expand: h1[pixel][hidden] = relu6(W1 x + b1), zero outside the image (the depthwise padding)
    {
      bf16x8 wa[C::HT][C::KS1];
#pragma unroll
      for (int mt = 0; mt < C::HT; ++mt)
#pragma unroll
        for (int ks = 0; ks < C::KS1; ++ks) wa[mt][ks] = *reinterpret_cast<const bf16x8*>(wf1 + ((mt * C::KS1 + ks) * 64 + lane) * 8);
      for (int nt = wave; nt < T::NT1; nt += 4) {
        const int pi = 16 * nt + p;
        bf16x8 xb[C::KS1];
#pragma unroll
        for (int ks = 0; ks < C::KS1; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(in_t + pi * C::IP + 32 * ks + 8 * q);
        const int ry = pi / T::IW, rx = pi - ry * T::IW;
        const int iy = y0 + ry, ix = x0 + rx;
        const bool valid = pi < T::PIN && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
#pragma unroll
        for (int mt = 0; mt < C::HT; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS1; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt][ks], xb[ks], acc, 0, 0, 0);
          const float4 bb = *reinterpret_cast<const float4*>(b1s + 16 * mt + 4 * q);
          bf16x4 o;
          o[0] = (__bf16)(valid ? relu6_(acc[0] + bb.x) : 0.f);
          o[1] = (__bf16)(valid ? relu6_(acc[1] + bb.y) : 0.f);
          o[2] = (__bf16)(valid ? relu6_(acc[2] + bb.z) : 0.f);
          o[3] = (__bf16)(valid ? relu6_(acc[3] + bb.w) : 0.f);
          *reinterpret_cast<bf16x4*>(h1 + pi * C::HPI + 16 * mt + 4 * q) = o;   // HT*16 <= HP: inside the row
        }
      }
    }
    __syncthreads();

    // ---- depthwise 3x3 (stride 2) or its transpose on the hidden tile, + bias, ReLU6
    if (pl < NPL) {
      if (!TR) {
        for (int po = pl; po < T::POUT; po += NPL) {
          const int oy = po / T::OW, ox = po - oy * T::OW;
          float acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = b2r[j];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              float v[8];
              unpack8(*reinterpret_cast<const bf16x8*>(h1 + ((2 * oy + ky) * T::IW + 2 * ox + kx) * C::HPI + 8 * cg), v);
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], wdr[3 * ky + kx][j], acc[j]);
            }
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = relu6_(acc[j]);
          *reinterpret_cast<bf16x8*>(h2 + po * C::HPI + 8 * cg) = pack8(acc);
        }
      } else {
        // out[o] += in[i] * w[k] with o = 2i - 1 + k: even outputs take tap 1 of cell i, odd ones tap 2 of i and tap 0 of i+1
        for (int cell = pl; cell < TH * T::TW; cell += NPL) {
          const int cy = cell / T::TW, cx = cell - cy * T::TW;
          float v00[8], v01[8], v10[8], v11[8];
          unpack8(*reinterpret_cast<const bf16x8*>(h1 + (cy * T::IW + cx) * C::HPI + 8 * cg), v00);
          unpack8(*reinterpret_cast<const bf16x8*>(h1 + (cy * T::IW + cx + 1) * C::HPI + 8 * cg), v01);
          unpack8(*reinterpret_cast<const bf16x8*>(h1 + ((cy + 1) * T::IW + cx) * C::HPI + 8 * cg), v10);
          unpack8(*reinterpret_cast<const bf16x8*>(h1 + ((cy + 1) * T::IW + cx + 1) * C::HPI + 8 * cg), v11);
          float o00[8], o01[8], o10[8], o11[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o00[j] = relu6_(fmaf(v00[j], wdr[4][j], b2r[j]));
            o01[j] = relu6_(fmaf(v01[j], wdr[3][j], fmaf(v00[j], wdr[5][j], b2r[j])));
            o10[j] = relu6_(fmaf(v10[j], wdr[1][j], fmaf(v00[j], wdr[7][j], b2r[j])));
            o11[j] = relu6_(fmaf(v11[j], wdr[0][j], fmaf(v10[j], wdr[2][j], fmaf(v01[j], wdr[6][j], fmaf(v00[j], wdr[8][j], b2r[j])))));
          }
          __bf16* dst = h2 + ((2 * cy) * T::OW + 2 * cx) * C::HPI + 8 * cg;
          *reinterpret_cast<bf16x8*>(dst) = pack8(o00);
          *reinterpret_cast<bf16x8*>(dst + C::HPI) = pack8(o01);
          *reinterpret_cast<bf16x8*>(dst + T::OW * C::HPI) = pack8(o10);
          *reinterpret_cast<bf16x8*>(dst + (T::OW + 1) * C::HPI) = pack8(o11);
        }
      }
    }
    __syncthreads();

    // ---- project: out[pixel][cout] = W3 h2 + b3 (no activation)
    {
      bf16x8 wa[C::CT][C::KS3];
#pragma unroll
      for (int mt = 0; mt < C::CT; ++mt)
#pragma unroll
        for (int ks = 0; ks < C::KS3; ++ks) wa[mt][ks] = *reinterpret_cast<const bf16x8*>(wf3 + ((mt * C::KS3 + ks) * 64 + lane) * 8);
      const int oy0 = TR ? 2 * y0 : (y0 + 1) / 2, ox0 = TR ? 2 * x0 : (x0 + 1) / 2;
      for (int nt = wave; nt < T::NT3; nt += 4) {
        const int po = 16 * nt + p;
        bf16x8 hb[C::KS3];
#pragma unroll
        for (int ks = 0; ks < C::KS3; ++ks) hb[ks] = *reinterpret_cast<const bf16x8*>(h2 + po * C::HPI + 32 * ks + 8 * q);
        const int oy = oy0 + po / T::OW, ox = ox0 + po % T::OW;
        const bool valid = oy < a.Ho && ox < a.Wo;
        const size_t obase = (((size_t)b * a.Ho + oy) * a.Wo + ox) * COUT;
#pragma unroll
        for (int mt = 0; mt < C::CT; ++mt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < C::KS3; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt][ks], hb[ks], acc, 0, 0, 0);
          const int co = 16 * mt + 4 * q;
          const float4 bb = *reinterpret_cast<const float4*>(b3s + co);
          if (valid && co < COUT) {
            if (OUT_F32) {
              *reinterpret_cast<float4*>(static_cast<float*>(a.out) + obase + co) =
                  make_float4(acc[0] + bb.x, acc[1] + bb.y, acc[2] + bb.z, acc[3] + bb.w);
            } else {
              bf16x4 o;
              o[0] = (__bf16)(acc[0] + bb.x);
              o[1] = (__bf16)(acc[1] + bb.y);
              o[2] = (__bf16)(acc[2] + bb.z);
              o[3] = (__bf16)(acc[3] + bb.w);
              *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(a.out) + obase + co) = o;
            }
          }
        }
      }
    }
    // the next iteration's first barrier separates this tile's reads of h2 / in_t from their next writes
  }
}

template <int CIN, int HID, int COUT, bool TR, int TH>
int launch_ir(const IrArgs& a0, bool in_f32, bool out_f32, hipStream_t st) {
  using C = IrCfg<CIN, HID, COUT, TR, TH>;
  IrArgs a = a0;
  const int oh = TR ? 2 * TH : TH, ow = TR ? 32 : 16;
  a.tiles_y = (a.Ho + oh - 1) / oh;
  a.tiles_x = (a.Wo + ow - 1) / ow;
  const long ntiles = (long)a.B * a.tiles_y * a.tiles_x;
  const size_t smem = C::LDS_BYTES;
  const int per_cu = smem > 80 * 1024 ? 1 : (smem > 53 * 1024 ? 2 : 3);
  int grid = (int)(ntiles < (long)noise_num_cus() * per_cu ? ntiles : (long)noise_num_cus() * per_cu);
#define IR_LAUNCH(INF, OUTF)                                                                            \
  do {                                                                                                  \
    NOISE_ALLOW_LDS((ir_block_kernel<CIN, HID, COUT, TR, TH, INF, OUTF>), smem);                        \
    hipLaunchKernelGGL((ir_block_kernel<CIN, HID, COUT, TR, TH, INF, OUTF>), dim3(grid), dim3(256), smem, st, a); \
  } while (0)
  if (in_f32 && out_f32) IR_LAUNCH(true, true);
  else if (in_f32) IR_LAUNCH(true, false);
  else if (out_f32) IR_LAUNCH(false, true);
  else IR_LAUNCH(false, false);
#undef IR_LAUNCH
  return 0;
}


// ------------------------------------------------------------------------------------------------------------------
// Head: normalise + left/top pad (models.py:614-622, :589) -> first Conv2d 1->32 k3 s2 + BN + ReLU6 (:1151-1159, kept as
// skip tensor A0) -> block 1: depthwise 3x3 s1 + BN + ReLU6 -> 1x1 32->16 + BN (:385-396, skip tensor A1).
// The single-channel 3x3 convolution runs on the matrix cores as a K = 32 product whose k index holds the 9 taps of the
// input rounded to bf16 (k = 0..8) and the 9 rounding remainders (k = 9..17): the input itself keeps ~16 significant
// bits, only the weights are bf16 like everywhere else.
struct HeadArgs {
  const float* logmag;   // [B][H][W]
  const float* absmax;   // [B]
  const __bf16* w0;      // [32][32]  rows = channels, k = tap (0..8), tap again (9..17), zeros
  const float* b0;       // [32]
  const float* wd;       // [9][32]
  const float* b2;       // [32]
  const __bf16* w3;      // [16][32]
  const float* b3;       // [16]
  __bf16* a0;            // [B][H1][W1][32]
  __bf16* a1;            // [B][H1][W1][16]
  int B, H, W, hp, wp, H1, W1, tiles_y, tiles_x;
  float pad_value;
};

constexpr int HD_OH = 8, HD_OW = 16;
constexpr int HD_AH = HD_OH + 2, HD_AW = HD_OW + 2, HD_PA = HD_AH * HD_AW, HD_NTA = (HD_PA + 15) / 16;   // A0 halo tile
constexpr int HD_XH = 2 * HD_AH + 1, HD_XW = 2 * HD_AW + 1, HD_XP = HD_XW + 1, HD_NX = HD_XH * HD_XW;     // input tile
constexpr int HD_C0 = 32, HD_C1 = 16, HD_CP = HD_C0 + 8;
constexpr int HD_POUT = HD_OH * HD_OW, HD_NT3 = HD_POUT / 16;

__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[HD_XH * HD_XP];
  __shared__ __attribute__((aligned(16))) __bf16 h1[HD_NTA * 16 * HD_CP];   // A0 on the halo tile (zero outside the image)
  __shared__ __attribute__((aligned(16))) __bf16 h2[HD_POUT * HD_CP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
  const int cg = tid & 3, pl = tid >> 2;      // depthwise: 4 channel groups of 8, 64 pixel lanes
  float wdr[9][8], b2r[8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wdr[k][j] = a.wd[k * HD_C0 + 8 * cg + j];
#pragma unroll
  for (int j = 0; j < 8; ++j) b2r[j] = a.b2[8 * cg + j];
  bf16x8 w0a[2], w3a;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) w0a[mt] = *reinterpret_cast<const bf16x8*>(a.w0 + (16 * mt + p) * 32 + 8 * q);
  w3a = *reinterpret_cast<const bf16x8*>(a.w3 + p * 32 + 8 * q);
  float b0r[2][4], b3r[4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b0r[mt][r] = a.b0[16 * mt + 4 * q + r];
#pragma unroll
  for (int r = 0; r < 4; ++r) b3r[r] = a.b3[4 * q + r];
  for (int i = tid; i < HD_NTA * 16 * HD_CP / 8; i += 256) reinterpret_cast<uint4*>(h1)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < HD_POUT * HD_CP / 8; i += 256) reinterpret_cast<uint4*>(h2)[i] = make_uint4(0, 0, 0, 0);

  const int Hp = a.H + a.hp, Wp = a.W + a.wp;
  constexpr int NI = (HD_NX + 255) / 256;
  float pre[NI];
  const long ntiles = (long)a.B * a.tiles_y * a.tiles_x;
  auto origin = [&](long tile, int& b, int& oy0, int& ox0) {
    const int tx = (int)(tile % a.tiles_x);
    const long r = tile / a.tiles_x;
    oy0 = (int)(r % a.tiles_y) * HD_OH;
    ox0 = tx * HD_OW;
    b = (int)(r / a.tiles_y);
  };
  auto prefetch = [&](long tile) {
    int b, oy0, ox0;
    origin(tile, b, oy0, ox0);
    const float inv = 1.f / (a.absmax[b] + 1e-6f);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = tid + 256 * u;
      const int ry = i / HD_XW, rx = i - ry * HD_XW;
      const int yy = 2 * oy0 - 3 + ry, xx = 2 * ox0 - 3 + rx;            // position in the padded image
      const bool inside = yy >= 0 && yy < Hp && xx >= 0 && xx < Wp;
      const bool real = inside && yy >= a.hp && xx >= a.wp;
      const int sy = real ? yy - a.hp : 0, sx = real ? xx - a.wp : 0;
      const float v = a.logmag[((size_t)b * a.H + sy) * a.W + sx] * inv;
      pre[u] = real ? v : (inside ? a.pad_value : 0.f);
    }
  };
  long tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int b, oy0, ox0;
    origin(tile, b, oy0, ox0);
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = tid + 256 * u;
      if (i < HD_NX) xs[(i / HD_XW) * HD_XP + i % HD_XW] = pre[u];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);

    // ---- first conv on the A0 halo tile
    for (int nt = wave; nt < HD_NTA; nt += 4) {
      const int pi = 16 * nt + p;
      const int ry = pi / HD_AW, rx = pi - ry * HD_AW;
      const int ay = oy0 - 1 + ry, ax = ox0 - 1 + rx;                     // A0 pixel
      const bool valid = pi < HD_PA && ay >= 0 && ay < a.H1 && ax >= 0 && ax < a.W1;
      const int ryc = pi < HD_PA ? ry : 0, rxc = pi < HD_PA ? rx : 0;
      bf16x8 xb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;                       // k < 9: tap k, rounded; 9 <= k < 18: remainder of tap k - 9
        const int tap = k < 9 ? k : (k < 18 ? k - 9 : 0);
        const float v = xs[(2 * ryc + tap / 3) * HD_XP + 2 * rxc + tap % 3];
        const __bf16 hi = (__bf16)v;
        xb[j] = k < 9 ? hi : (k < 18 ? (__bf16)(v - (float)hi) : (__bf16)0.f);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0a[mt], xb, acc, 0, 0, 0);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)(valid ? relu6_(acc[r] + b0r[mt][r]) : 0.f);
        *reinterpret_cast<bf16x4*>(h1 + pi * HD_CP + 16 * mt + 4 * q) = o;
        if (valid && ry >= 1 && ry <= HD_OH && rx >= 1 && rx <= HD_OW)
          *reinterpret_cast<bf16x4*>(a.a0 + (((size_t)b * a.H1 + ay) * a.W1 + ax) * HD_C0 + 16 * mt + 4 * q) = o;
      }
    }
    __syncthreads();
    // ---- depthwise 3x3 stride 1 on A0
    for (int po = pl; po < HD_POUT; po += 64) {
      const int oy = po / HD_OW, ox = po - oy * HD_OW;
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = b2r[j];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float v[8];
          unpack8(*reinterpret_cast<const bf16x8*>(h1 + ((oy + ky) * HD_AW + ox + kx) * HD_CP + 8 * cg), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], wdr[3 * ky + kx][j], acc[j]);
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = relu6_(acc[j]);
      *reinterpret_cast<bf16x8*>(h2 + po * HD_CP + 8 * cg) = pack8(acc);
    }
    __syncthreads();
    // ---- project 32 -> 16
    for (int nt = wave; nt < HD_NT3; nt += 4) {
      const int po = 16 * nt + p;
      const bf16x8 hb = *reinterpret_cast<const bf16x8*>(h2 + po * HD_CP + 8 * q);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3a, hb, acc, 0, 0, 0);
      const int oy = oy0 + po / HD_OW, ox = ox0 + po % HD_OW;
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
  const __bf16* d2;      // [B][h][w][16]
  const __bf16* a1;      // [B][Hs1][Ws1][16], read at (+oy1, +ox1)
  const __bf16* a0;      // [B][Hs0][Ws0][32], read at (+oy0, +ox0)
  const float* wd;       // [9][16]
  const float* b2;       // [16]
  const __bf16* w3;      // [32][32]  (k >= 16 zero)
  const float* b3;       // [32]
  const __bf16* wl;      // [16][32]  rows = taps (9 used), k = channel
  const float* logmag;   // [B][H][W]
  const float* absmax;   // [B]
  float* enh;            // [B][H][W]   (optional)
  float* filt;           // [B][H][W]   (optional: the raw network output)
  int B, h, w, Hs1, Ws1, oy1, ox1, Hs0, Ws0, oy0, ox0, H, W, offy, offx, tiles_y, tiles_x;
  float bias;
};

constexpr int TL_CH = 4, TL_CW = 16;
constexpr int TL_UH = TL_CH + 1, TL_UW = TL_CW + 1, TL_PU = TL_UH * TL_UW, TL_NTU = (TL_PU + 15) / 16;   // u region
constexpr int TL_SH = TL_CH + 3, TL_SW = TL_CW + 3, TL_PS = TL_SH * TL_SW;                                // s region
constexpr int TL_CP = 32 + 8;

__global__ __launch_bounds__(256) void tail_kernel(TailArgs a) {
  __shared__ __attribute__((aligned(16))) float st[TL_PS * 16];              // s = D2 + A1, fp32 (zero outside)
  __shared__ __attribute__((aligned(16))) __bf16 a0t[TL_NTU * 16 * TL_CP];   // A0 on the u region
  __shared__ __attribute__((aligned(16))) __bf16 h2[TL_NTU * 16 * TL_CP];    // depthwise output (channels 16..31 zero)
  __shared__ __attribute__((aligned(16))) __bf16 ut[TL_NTU * 16 * TL_CP];    // u = D3 + A0 (zero outside the image)
  __shared__ __attribute__((aligned(16))) float pt[TL_NTU * 16 * 12];        // P[pixel][tap]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p = lane & 15, q = lane >> 4;
  const int cg = tid & 1, pl = tid >> 1;      // depthwise: 2 channel groups of 8, 128 pixel lanes
  float wdr[9][8], b2r[8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) wdr[k][j] = a.wd[k * 16 + 8 * cg + j];
#pragma unroll
  for (int j = 0; j < 8; ++j) b2r[j] = a.b2[8 * cg + j];
  bf16x8 w3a[2], wla;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) w3a[mt] = *reinterpret_cast<const bf16x8*>(a.w3 + (16 * mt + p) * 32 + 8 * q);
  wla = *reinterpret_cast<const bf16x8*>(a.wl + p * 32 + 8 * q);
  float b3r[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) b3r[mt][r] = a.b3[16 * mt + 4 * q + r];
  for (int i = tid; i < TL_NTU * 16 * TL_CP / 8; i += 256) {
    reinterpret_cast<uint4*>(a0t)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(h2)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(ut)[i] = make_uint4(0, 0, 0, 0);
  }

  constexpr int PCS = TL_PS * 2, NIS = (PCS + 255) / 256;        // s pieces (8 channels)
  constexpr int PCA = TL_PU * 4, NIA = (PCA + 255) / 256;        // A0 pieces
  float pres[NIS][8];
  bf16x8 prea[NIA];
  const long ntiles = (long)a.B * a.tiles_y * a.tiles_x;
  auto origin = [&](long tile, int& b, int& cy0, int& cx0) {
    const int tx = (int)(tile % a.tiles_x);
    const long r = tile / a.tiles_x;
    cy0 = (int)(r % a.tiles_y) * TL_CH;
    cx0 = tx * TL_CW;
    b = (int)(r / a.tiles_y);
  };
  auto prefetch = [&](long tile) {
    int b, cy0, cx0;
    origin(tile, b, cy0, cx0);
#pragma unroll
    for (int u = 0; u < NIS; ++u) {
      const int it = tid + 256 * u;
      const int pi = it >> 1, c8 = it & 1;
      const int ry = pi / TL_SW, rx = pi - ry * TL_SW;
      const int iy = cy0 - 1 + ry, ix = cx0 - 1 + rx;
      const bool in = iy >= 0 && iy < a.h && ix >= 0 && ix < a.w;
      const int yc = in ? iy : 0, xc = in ? ix : 0;
      float d[8], s[8];
      unpack8(*reinterpret_cast<const bf16x8*>(a.d2 + (((size_t)b * a.h + yc) * a.w + xc) * 16 + 8 * c8), d);
      unpack8(*reinterpret_cast<const bf16x8*>(a.a1 + (((size_t)b * a.Hs1 + yc + a.oy1) * a.Ws1 + xc + a.ox1) * 16 + 8 * c8), s);
#pragma unroll
      for (int j = 0; j < 8; ++j) pres[u][j] = in ? d[j] + s[j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
      const int it = tid + 256 * u;
      const int pi = it >> 2, c8 = it & 3;
      const int ry = pi / TL_UW, rx = pi - ry * TL_UW;
      int iy = cy0 + ry, ix = cx0 + rx;
      iy = iy < a.h ? iy : a.h - 1;
      ix = ix < a.w ? ix : a.w - 1;
      prea[u] = *reinterpret_cast<const bf16x8*>(a.a0 + (((size_t)b * a.Hs0 + iy + a.oy0) * a.Ws0 + ix + a.ox0) * 32 + 8 * c8);
    }
  };
  long tile = blockIdx.x;
  if (tile < ntiles) prefetch(tile);
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    int b, cy0, cx0;
    origin(tile, b, cy0, cx0);
#pragma unroll
    for (int u = 0; u < NIS; ++u) {
      const int it = tid + 256 * u;
      if (it < PCS) {
        float* dst = st + (it >> 1) * 16 + 8 * (it & 1);
        *reinterpret_cast<float4*>(dst) = make_float4(pres[u][0], pres[u][1], pres[u][2], pres[u][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(pres[u][4], pres[u][5], pres[u][6], pres[u][7]);
      }
    }
#pragma unroll
    for (int u = 0; u < NIA; ++u) {
      const int it = tid + 256 * u;
      if (it < PCA) *reinterpret_cast<bf16x8*>(a0t + (it >> 2) * TL_CP + 8 * (it & 3)) = prea[u];
    }
    __syncthreads();
    if (tile + gridDim.x < ntiles) prefetch(tile + gridDim.x);

    // ---- depthwise ConvTranspose2d 3x3 stride 1: out[o] = sum_k s[o + 1 - k] w[k]
    for (int pu = pl; pu < TL_PU; pu += 128) {
      const int uy = pu / TL_UW, ux = pu - uy * TL_UW;      // u pixel = region origin + (uy, ux); s index = +1
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = b2r[j];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* src = st + ((uy + 2 - ky) * TL_SW + ux + 2 - kx) * 16 + 8 * cg;
          const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
          const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(v[j], wdr[3 * ky + kx][j], acc[j]);
        }
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = relu6_(acc[j]);
      *reinterpret_cast<bf16x8*>(h2 + pu * TL_CP + 8 * cg) = pack8(acc);
    }
    __syncthreads();
    // ---- 1x1 16 -> 32, + A0: u (zero outside the image: the transposed convolution only sees real pixels)
    for (int nt = wave; nt < TL_NTU; nt += 4) {
      const int pu = 16 * nt + p;
      const bf16x8 hb = *reinterpret_cast<const bf16x8*>(h2 + pu * TL_CP + 8 * q);
      const int uy = pu / TL_UW, ux = pu - uy * TL_UW;
      const bool valid = pu < TL_PU && cy0 + uy < a.h && cx0 + ux < a.w;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3a[mt], hb, acc, 0, 0, 0);
        const bf16x4 sk = *reinterpret_cast<const bf16x4*>(a0t + pu * TL_CP + 16 * mt + 4 * q);
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)(valid ? acc[r] + b3r[mt][r] + (float)sk[r] : 0.f);
        *reinterpret_cast<bf16x4*>(ut + pu * TL_CP + 16 * mt + 4 * q) = o;
      }
    }
    __syncthreads();
    // ---- P[pixel][tap]
    for (int nt = wave; nt < TL_NTU; nt += 4) {
      const int pu = 16 * nt + p;
      const bf16x8 ub = *reinterpret_cast<const bf16x8*>(ut + pu * TL_CP + 8 * q);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wla, ub, acc, 0, 0, 0);
      if (q < 3) *reinterpret_cast<float4*>(pt + pu * 12 + 4 * q) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
    __syncthreads();
    // ---- output samples: thread = one of the 2CH x 2CW outputs of the tile
    {
      const int Y = tid >> 5, X = tid & 31;
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
      if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
        const size_t o = ((size_t)b * a.H + y) * a.W + x;
        if (a.filt) a.filt[o] = f;
        if (a.enh) {
          const float mx = a.absmax[b] + 1e-6f;
          a.enh[o] = fminf(fmaxf(a.logmag[o] / mx + f, -1.f), 1.f) * mx;
        }
      }
    }
    // the next iteration's first barrier separates these reads of pt from the next writes (pt is written after 3 more)
  }
}

}  // namespace

// Padded weight shapes the fused kernels expect (so that a binding can size its buffers):
//   w1 [ceil(Hid/16)*16][roundup(Cin,32)] bf16, b1 [ceil(Hid/16)*16] f32, wd [9][ceil(Hid/8)*8] f32 (tap-major),
//   b2 [ceil(Hid/8)*8] f32, w3 [ceil(Cout/16)*16][roundup(Hid,32)] bf16, b3 [ceil(Cout/16)*16] f32.
NOISE_EXPORT int noise_fused_ir_block(const void* x, const void* skip, int in_f32, int Hs, int Ws, int soy, int sox,
                                      const void* w1, const float* b1, const float* wd, const float* b2, const void* w3,
                                      const float* b3, void* out, int out_f32, int transposed, int B, int Hi, int Wi,
                                      int Cin, int Hid, int Cout, void* stream) {
  NOISE_REQUIRE(x && w1 && b1 && wd && b2 && w3 && b3 && out, "noise_fused_ir_block: null pointer");
  NOISE_REQUIRE(B > 0 && Hi > 0 && Wi > 0, "noise_fused_ir_block: empty input");
  IrArgs a{};
  a.x = x; a.skip = skip; a.w1 = (const __bf16*)w1; a.b1 = b1; a.wd = wd; a.b2 = b2; a.w3 = (const __bf16*)w3; a.b3 = b3;
  a.out = out; a.B = B; a.Hi = Hi; a.Wi = Wi;
  a.Ho = transposed ? 2 * Hi - 1 : (Hi - 1) / 2 + 1;
  a.Wo = transposed ? 2 * Wi - 1 : (Wi - 1) / 2 + 1;
  a.Hs = Hs; a.Ws = Ws; a.soy = soy; a.sox = sox;
  if (skip) NOISE_REQUIRE(soy >= 0 && sox >= 0 && soy + Hi <= Hs && sox + Wi <= Ws,
                          "noise_fused_ir_block: skip window %dx%d at (%d,%d) leaves the %dx%d skip tensor", Hi, Wi, soy, sox, Hs, Ws);
  hipStream_t st = (hipStream_t)stream;
  bool done = false;
#define IR_CASE(CI, HI, CO, TRV, THV)                                                     \
  if (!done && Cin == CI && Hid == HI && Cout == CO && (transposed != 0) == TRV) {        \
    launch_ir<CI, HI, CO, TRV, THV>(a, in_f32 != 0, out_f32 != 0, st);                    \
    done = true;                                                                          \
  }
  // the inverted-residual blocks of train/configs/spectralv2.yaml:77-84 (t = 3): encoder 16->24, 24->32, 32->64 and
  // the mirrored decoder 64->32, 32->24, 24->16
  IR_CASE(16, 48, 24, false, 2)
  IR_CASE(24, 72, 32, false, 2)
  IR_CASE(32, 96, 64, false, 2)
  IR_CASE(64, 192, 32, true, 2)
  IR_CASE(32, 96, 24, true, 2)
  IR_CASE(24, 72, 16, true, 2)
#undef IR_CASE
  NOISE_REQUIRE(done, "noise_fused_ir_block: no fused kernel for %d -> %d -> %d (%s); use the per-layer fp32 path", Cin, Hid,
                Cout, transposed ? "transposed" : "strided");
  NOISE_CHECK_LAUNCH("noise_fused_ir_block");
  return 0;
}


// Head of the fused network (see head_kernel).  w0 [32][32] bf16: row c holds the 9 folded taps of output channel c at
// k = 0..8 and again at k = 9..17, zeros after; wd [9][32], w3 [16][32] bf16.  a0 [B][H1][W1][32], a1 [B][H1][W1][16]
// with H1 = (H + hp - 1)/2 + 1, W1 = (W + wp - 1)/2 + 1.
NOISE_EXPORT int noise_fused_head(const float* logmag, const float* absmax, const void* w0, const float* b0,
                                  const float* wd, const float* b2, const void* w3, const float* b3, void* a0, void* a1,
                                  int B, int H, int W, int hp, int wp, float pad_value, void* stream) {
  NOISE_REQUIRE(logmag && absmax && w0 && b0 && wd && b2 && w3 && b3 && a0 && a1, "noise_fused_head: null pointer");
  NOISE_REQUIRE(B > 0 && H > 0 && W > 0 && hp >= 0 && wp >= 0, "noise_fused_head: bad shape");
  HeadArgs a{};
  a.logmag = logmag; a.absmax = absmax; a.w0 = (const __bf16*)w0; a.b0 = b0; a.wd = wd; a.b2 = b2;
  a.w3 = (const __bf16*)w3; a.b3 = b3; a.a0 = (__bf16*)a0; a.a1 = (__bf16*)a1;
  a.B = B; a.H = H; a.W = W; a.hp = hp; a.wp = wp; a.pad_value = pad_value;
  a.H1 = (H + hp - 1) / 2 + 1;
  a.W1 = (W + wp - 1) / 2 + 1;
  a.tiles_y = noise_cdiv(a.H1, HD_OH);
  a.tiles_x = noise_cdiv(a.W1, HD_OW);
  const long ntiles = (long)B * a.tiles_y * a.tiles_x;
  const long cap = (long)noise_num_cus() * 4;
  hipLaunchKernelGGL(head_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), 0, (hipStream_t)stream, a);
  NOISE_CHECK_LAUNCH("noise_fused_head");
  return 0;
}

// Tail of the fused network (see tail_kernel).  d2 [B][h][w][16]; a1 / a0: the two skip tensors with their own sizes and
// crop offsets; wd [9][16], w3 [32][32] bf16 (columns >= 16 zero), wl [16][32] bf16 (row = tap of the last transposed
// convolution, column = channel), bias its scalar bias.  enh / filt [B][H][W] (either may be NULL).
NOISE_EXPORT int noise_fused_tail(const void* d2, const void* a1, int Hs1, int Ws1, int oy1, int ox1, const void* a0,
                                  int Hs0, int Ws0, int oy0, int ox0, const float* wd, const float* b2, const void* w3,
                                  const float* b3, const void* wl, float bias, const float* logmag, const float* absmax,
                                  float* enh, float* filt, int B, int h, int w, int H, int W, void* stream) {
  NOISE_REQUIRE(d2 && a1 && a0 && wd && b2 && w3 && b3 && wl && (enh || filt), "noise_fused_tail: null pointer");
  NOISE_REQUIRE(enh == nullptr || (logmag && absmax), "noise_fused_tail: enh needs logmag and absmax");
  NOISE_REQUIRE(B > 0 && h > 0 && w > 0 && H > 0 && W > 0, "noise_fused_tail: bad shape");
  NOISE_REQUIRE(2 * h - 1 >= H && 2 * w - 1 >= W, "noise_fused_tail: network output %dx%d is smaller than its input %dx%d",
                2 * h - 1, 2 * w - 1, H, W);
  NOISE_REQUIRE(oy1 >= 0 && ox1 >= 0 && oy1 + h <= Hs1 && ox1 + w <= Ws1 && oy0 >= 0 && ox0 >= 0 && oy0 + h <= Hs0 && ox0 + w <= Ws0,
                "noise_fused_tail: a skip window leaves its tensor");
  TailArgs a{};
  a.d2 = (const __bf16*)d2; a.a1 = (const __bf16*)a1; a.a0 = (const __bf16*)a0; a.wd = wd; a.b2 = b2;
  a.w3 = (const __bf16*)w3; a.b3 = b3; a.wl = (const __bf16*)wl; a.bias = bias; a.logmag = logmag; a.absmax = absmax;
  a.enh = enh; a.filt = filt; a.B = B; a.h = h; a.w = w; a.Hs1 = Hs1; a.Ws1 = Ws1; a.oy1 = oy1; a.ox1 = ox1;
  a.Hs0 = Hs0; a.Ws0 = Ws0; a.oy0 = oy0; a.ox0 = ox0; a.H = H; a.W = W;
  a.offy = 2 * h - 1 - H;
  a.offx = 2 * w - 1 - W;
  a.tiles_y = noise_cdiv(h, TL_CH);
  a.tiles_x = noise_cdiv(w, TL_CW);
  const long ntiles = (long)B * a.tiles_y * a.tiles_x;
  const long cap = (long)noise_num_cus() * 4;
  hipLaunchKernelGGL(tail_kernel, dim3((unsigned)(ntiles < cap ? ntiles : cap)), dim3(256), 0, (hipStream_t)stream, a);
  NOISE_CHECK_LAUNCH("noise_fused_tail");
  return 0;
}
