// Column-walking 3x3 depthwise stencils (NHWC fp32, equal stride 1 or 2 in both axes), gfx950.
//
// A thread owns one column of one clip (or of one band of rows of it) for 4 consecutive channels and walks
// down the rows, keeping the rows the stencil shares between consecutive steps in registers.  Compared with
// one thread per output pixel this loads every source element 3x (stride 1), 1.5x (stride 2) or 1x
// (transposed stride 2) instead of 9x / 2.25x, evaluates the lazy BatchNorm(+ReLU6) or BatchNorm-backward
// expression of each loaded element once, and has no divergent taps.
//
// Two geometries cover every 3x3 / pad 1 depthwise layer of the U-Net and its adjoints:
//   gather<S>:  dst[o] = sum_k src[o*S - 1 + k] * w[k]                      (walks dst rows)
//   cell     :  dst[o] = sum_k src[(o + 1 - k) / 2] * w[k], (o+1-k) even     (walks src rows; 2x2 dst per step)
//
//   layer (train/models.py)                  forward            backward (dgrad + wgrad)
//   Conv2d s1           :385-393             gather<1>          gather<1>, taps flipped, on dy
//   Conv2d s2           :385-393             gather<2>          cell on dy
//   ConvTranspose2d s1  :299-345             gather<1> flipped  gather<1> on dy
//   ConvTranspose2d s2  :299-345             cell               gather<2> on dy
// (flipped: tap k uses w[8-k]; the weight gradient is written back with the same permutation.)
#pragma once
#include "lazy.h"

namespace {

struct WalkDims {
  int B, Hs, Ws;   // source tensor (B, Hs, Ws, C)
  int Hd, Wd;      // destination tensor (B, Hd, Wd, C)
  int C;
  int wown;        // columns owned by threads: Wd (gather) or Ws (cell)
  int nb, bh;      // bands per clip and walked rows per band (dst rows for gather, src rows for cell)
  int ngroups;     // number of groups of PPB = 256 / (C/4) column slots
};

__device__ __forceinline__ void load_taps(const float* __restrict__ w, int c, bool flip, float4 (&wk)[9]) {
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int kk = flip ? 8 - k : k;
    wk[k] = make_float4(w[(c + 0) * 9 + kk], w[(c + 1) * 9 + kk], w[(c + 2) * 9 + kk], w[(c + 3) * 9 + kk]);
  }
}

__device__ __forceinline__ float4 keep4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// Decode a column slot: clip, band rows [r0, r1) and owned column.
__device__ __forceinline__ bool walk_slot(const WalkDims& d, int slot, int nrows, int& bi, int& r0, int& r1, int& col) {
  if (slot >= d.B * d.nb * d.wown) return false;
  col = slot % d.wown;
  const int u = slot / d.wown;
  const int band = u % d.nb;
  bi = u / d.nb;
  r0 = band * d.bh;
  r1 = r0 + d.bh < nrows ? r0 + d.bh : nrows;
  return r0 < r1;
}

// ------------------------------------------------------------------------------------------------
// forward, gather geometry
// HAS_B (here and below): the input is the sum of two lazy sources.  Optional operands are template flags, not
// run-time tests: a branch around a load makes the compiler wait for ALL outstanding loads behind it.
template <int S, bool FLIP, bool HAS_B>
__global__ __launch_bounds__(256) void walk_gather_fwd_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                              float* __restrict__ out, float* __restrict__ slab,
                                                              WalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (pl < PPB) {
    const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
    float4 wk[9];
    load_taps(w, c, FLIP, wk);
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      int bi, r0, r1, ox;
      if (!walk_slot(d, g * PPB + pl, d.Hd, bi, r0, r1, ox)) continue;
      const int ix0 = ox * S - 1;
      const bool v0 = ix0 >= 0, v1 = ix0 + 1 < d.Ws, v2 = ix0 + 2 < d.Ws;
      const size_t clip = (size_t)bi * d.Hs * d.Ws * C + c;
      // branch-free: a tap outside the source reads a clamped (valid) address and is zeroed afterwards, so all
      // loads of a step are issued back to back
      const int cx0 = v0 ? ix0 : 0, cx1 = v1 ? ix0 + 1 : 0, cx2 = v2 ? ix0 + 2 : 0;
      auto load_row = [&](int iy, float4(&r)[3]) {
        const bool vy = iy >= 0 && iy < d.Hs;
        const size_t row = clip + (size_t)(vy ? iy : 0) * d.Ws * C;
        const size_t o0 = row + (size_t)cx0 * C, o1 = row + (size_t)cx1 * C, o2 = row + (size_t)cx2 * C;
        float4 x0 = lazy4_at(a, o0), x1 = lazy4_at(a, o1), x2 = lazy4_at(a, o2);
        if (HAS_B) {
          x0 = x0 + lazy4_at(b, o0);
          x1 = x1 + lazy4_at(b, o1);
          x2 = x2 + lazy4_at(b, o2);
        }
        r[0] = keep4(vy && v0, x0);
        r[1] = keep4(vy && v1, x1);
        r[2] = keep4(vy && v2, x2);
      };
      float4 t0[3], t1[3], t2[3];
      load_row(r0 * S - 1, t0);
      if (S == 1) load_row(r0, t1);
      float* o = out + (((size_t)bi * d.Hd + r0) * d.Wd + ox) * C + c;
      for (int oy = r0; oy < r1; ++oy, o += (size_t)d.Wd * C) {
        if (S == 1) {
          load_row(oy + 1, t2);
        } else {
          load_row(2 * oy, t1);
          load_row(2 * oy + 1, t2);
        }
        float4 acc = t0[0] * wk[0];
        acc = fma4(t0[1], wk[1], acc);
        acc = fma4(t0[2], wk[2], acc);
        acc = fma4(t1[0], wk[3], acc);
        acc = fma4(t1[1], wk[4], acc);
        acc = fma4(t1[2], wk[5], acc);
        acc = fma4(t2[0], wk[6], acc);
        acc = fma4(t2[1], wk[7], acc);
        acc = fma4(t2[2], wk[8], acc);
        st4(o, acc);
        s1 = s1 + acc;
        s2 = fma4(acc, acc, s2);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (S == 1) {
            t0[j] = t1[j];
            t1[j] = t2[j];
          } else {
            t0[j] = t2[j];
          }
        }
      }
    }
  }
  if (slab != nullptr) {
    const float4 v[2] = {s1, s2};
    block_channel_reduce<2>(v, C, CG, PPB, red, slab + (size_t)blockIdx.x * 2 * C);
  }
}

// forward, cell geometry (transposed convolution, stride 2).  STORE = false: statistics only -- the output is left to
// consumers that rebuild it from the layer input (pwvirt.hip), `out` is not dereferenced.
template <bool HAS_B, bool STORE = true>
__global__ __launch_bounds__(256) void walk_cell_fwd_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                            float* __restrict__ out, float* __restrict__ slab,
                                                            WalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (pl < PPB) {
    const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
    float4 wk[9];
    load_taps(w, c, false, wk);
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      int bi, m0, m1, mx;
      if (!walk_slot(d, g * PPB + pl, d.Hs, bi, m0, m1, mx)) continue;
      const bool vx1 = mx + 1 < d.Ws;         // source column mx+1 exists
      const bool dx1 = 2 * mx + 1 < d.Wd;     // destination column 2mx+1 exists
      const size_t clip = (size_t)bi * d.Hs * d.Ws * C + c;
      auto load_row = [&](int iy, float4& r0, float4& r1) {   // branch-free, see walk_gather_fwd_kernel
        const bool vy = iy < d.Hs;
        const size_t o0 = clip + ((size_t)(vy ? iy : 0) * d.Ws + mx) * C, o1 = vx1 ? o0 + C : o0;
        float4 x0 = lazy4_at(a, o0), x1 = lazy4_at(a, o1);
        if (HAS_B) {
          x0 = x0 + lazy4_at(b, o0);
          x1 = x1 + lazy4_at(b, o1);
        }
        r0 = keep4(vy, x0);
        r1 = keep4(vy && vx1, x1);
      };
      float4 s00, s01, s10, s11;
      load_row(m0, s00, s01);
      float* o = out + (((size_t)bi * d.Hd + 2 * m0) * d.Wd + 2 * mx) * C + c;
      const size_t drow = (size_t)d.Wd * C;
      for (int m = m0; m < m1; ++m, o += 2 * drow) {
        load_row(m + 1, s10, s11);
        const float4 d00 = s00 * wk[4];
        if (STORE) st4(o, d00);
        s1 = s1 + d00;
        s2 = fma4(d00, d00, s2);
        if (dx1) {
          const float4 d01 = fma4(s01, wk[3], s00 * wk[5]);
          if (STORE) st4(o + C, d01);
          s1 = s1 + d01;
          s2 = fma4(d01, d01, s2);
        }
        if (2 * m + 1 < d.Hd) {
          const float4 d10 = fma4(s10, wk[1], s00 * wk[7]);
          if (STORE) st4(o + drow, d10);
          s1 = s1 + d10;
          s2 = fma4(d10, d10, s2);
          if (dx1) {
            float4 d11 = s00 * wk[8];
            d11 = fma4(s01, wk[6], d11);
            d11 = fma4(s10, wk[2], d11);
            d11 = fma4(s11, wk[0], d11);
            if (STORE) st4(o + drow + C, d11);
            s1 = s1 + d11;
            s2 = fma4(d11, d11, s2);
          }
        }
        s00 = s10;
        s01 = s11;
      }
    }
  }
  if (slab != nullptr) {
    const float4 v[2] = {s1, s2};
    block_channel_reduce<2>(v, C, CG, PPB, red, slab + (size_t)blockIdx.x * 2 * C);
  }
}

// ------------------------------------------------------------------------------------------------
// backward: destination = the layer's input gradient g_in (and the activation a at the same pixels, for the
// weight gradient), source = the lazy gradient dy of the layer's output.
struct BwdPixel {
  Lazy4 a, b;
  Epi4 epi;
  float* g_in;
};

// Finishes one destination pixel: weight-gradient contribution is done by the caller (it needs av), this applies
// the epilogue and stores.
template <bool HAS_B>
__device__ __forceinline__ float4 bwd_activation(const BwdPixel& p, size_t off, float4& yraw) {
  yraw = ld4(p.a.y + off);
  float4 av = lazy4_act(p.a, yraw);
  if (HAS_B) av = av + lazy4_at(p.b, off);
  return av;
}

// Per-channel constants of a backward walker (9 taps, 6 gradient vectors, scale/shift of both sources, 5 epilogue
// vectors) would sit in ~96 registers of every thread for the whole walk.  They live in LDS instead: one copy per
// workgroup, written once, read back with a few ds_read_b128 per step, so that two waves fit a SIMD.  The table shares
// the reduction buffer, which is not touched before the walk has ended.
enum { BT_W = 0, BT_G = 9, BT_A = 15, BT_B = 17, BT_E = 19, BT_ROWS = 24 };

__device__ __forceinline__ void bwd_table_fill(float* tab, int C, int c, const LazyGrad& G, const LazySrc& A,
                                               const LazySrc& Bsrc, const BwdEpilogue& E, const float* __restrict__ w,
                                               bool flip) {
  float4 wk[9];
  load_taps(w, c, flip, wk);
#pragma unroll
  for (int k = 0; k < 9; ++k) st4(tab + (BT_W + k) * C + c, wk[k]);
  const LazyGrad4 g = lazygrad4_init(G, C, c);
  st4(tab + (BT_G + 0) * C + c, g.s);
  st4(tab + (BT_G + 1) * C + c, g.k1h);
  st4(tab + (BT_G + 2) * C + c, g.k1l);
  st4(tab + (BT_G + 3) * C + c, g.a2);
  st4(tab + (BT_G + 4) * C + c, g.mh);
  st4(tab + (BT_G + 5) * C + c, g.ml);
  const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
  st4(tab + (BT_A + 0) * C + c, a.scale);
  st4(tab + (BT_A + 1) * C + c, a.shift);
  st4(tab + (BT_B + 0) * C + c, b.scale);
  st4(tab + (BT_B + 1) * C + c, b.shift);
  const Epi4 e = epi4_init(E, C, c);
  st4(tab + (BT_E + 0) * C + c, e.scale);
  st4(tab + (BT_E + 1) * C + c, e.shift);
  st4(tab + (BT_E + 2) * C + c, e.mean);
  st4(tab + (BT_E + 3) * C + c, e.mlo);
  st4(tab + (BT_E + 4) * C + c, e.istd);
}

// The structs below are used with their scalar members only (pointers, bounds, flags); these return a copy whose
// vector members come from the table at t = tab + c.
__device__ __forceinline__ float4 bt_w(const float* t, int C, int k) { return ld4(t + (BT_W + k) * C); }
__device__ __forceinline__ LazyGrad4 bt_grad(LazyGrad4 m, const float* t, int C) {
  m.s = ld4(t + (BT_G + 0) * C);
  m.k1h = ld4(t + (BT_G + 1) * C);
  m.k1l = ld4(t + (BT_G + 2) * C);
  m.a2 = ld4(t + (BT_G + 3) * C);
  m.mh = ld4(t + (BT_G + 4) * C);
  m.ml = ld4(t + (BT_G + 5) * C);
  return m;
}
__device__ __forceinline__ Lazy4 bt_src(Lazy4 m, const float* t, int C, int row) {
  m.scale = ld4(t + row * C);
  m.shift = ld4(t + (row + 1) * C);
  return m;
}
__device__ __forceinline__ Epi4 bt_epi(Epi4 m, const float* t, int C) {
  m.scale = ld4(t + (BT_E + 0) * C);
  m.shift = ld4(t + (BT_E + 1) * C);
  m.mean = ld4(t + (BT_E + 2) * C);
  m.mlo = ld4(t + (BT_E + 3) * C);
  m.istd = ld4(t + (BT_E + 4) * C);
  return m;
}
// tab + c behind an offset the compiler cannot see through (it is zero): reads through it stay inside the loop
// iteration they are written in instead of being hoisted back into registers.
__device__ __forceinline__ const float* bt_here(const float* tc) {
  int z = 0;
  asm volatile("" : "+v"(z));
  return tc + z;
}

// RECOMP (stride-2 transposed layers with a BatchNorm behind them): the raw layer output y that the lazy gradient
// needs (dy = s*(dz - k1) + a2*(y - mean)) is recomputed from the layer input, which is four times smaller and walked
// anyway, instead of being read back: y(2i-1+r, 2j-1+c) = sum a(i, j) * wk[3r+c] is the adjoint of the gather below.
// The thread keeps the activated input rows oy and oy+1 at columns ox-1..ox+1; the centre of row oy doubles as the
// destination pixel's activation.
template <int S, bool FLIP, bool HAS_B, bool HAS_GO, bool RECOMP = false>
__global__ __launch_bounds__(256, 2) void walk_gather_bwd_kernel(LazyGrad G, LazySrc A, LazySrc Bsrc,
                                                              const float* __restrict__ w, float* __restrict__ g_in,
                                                              float* __restrict__ dw_slab, BwdEpilogue E, WalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 dwk[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) dwk[k] = f4(0.f);
  if (pl == 0) bwd_table_fill(red, C, c, G, A, Bsrc, E, w, FLIP);
  __syncthreads();
  if (pl < PPB) {
    // scalar members only; the per-channel vectors are read from the table where they are used
    const Lazy4 am = lazy4_init(A, C, c), bm = lazy4_init(Bsrc, C, c);
    const Epi4 em = epi4_init(E, C, c);
    const LazyGrad4 gm = lazygrad4_init(G, C, c);
    const float* const tc = red + c;
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      int bi, r0, r1, ox;
      if (!walk_slot(d, g * PPB + pl, d.Hd, bi, r0, r1, ox)) continue;
      const int ix0 = ox * S - 1;
      const bool v0 = ix0 >= 0, v1 = ix0 + 1 < d.Ws, v2 = ix0 + 2 < d.Ws;
      const size_t clip = (size_t)bi * d.Hs * d.Ws * C + c;
      const int cx0 = v0 ? ix0 : 0, cx1 = v1 ? ix0 + 1 : 0, cx2 = v2 ? ix0 + 2 : 0;
      auto load_row = [&](int iy, float4(&r)[3]) {   // branch-free, see walk_gather_fwd_kernel
        const LazyGrad4 gr = bt_grad(gm, tc, C);
        const bool vy = iy >= 0 && iy < d.Hs;
        const size_t row = clip + (size_t)(vy ? iy : 0) * d.Ws * C;
        const float4 x0 = lazygrad4_at(gr, row + (size_t)cx0 * C);
        const float4 x1 = lazygrad4_at(gr, row + (size_t)cx1 * C);
        const float4 x2 = lazygrad4_at(gr, row + (size_t)cx2 * C);
        r[0] = keep4(vy && v0, x0);
        r[1] = keep4(vy && v1, x1);
        r[2] = keep4(vy && v2, x2);
      };
      float4 t0[3], t1[3], t2[3];
      if (!RECOMP) load_row(r0 * S - 1, t0);
      if (S == 1) load_row(r0, t1);
      size_t ioff = (((size_t)bi * d.Hd + r0) * d.Wd + ox) * C + c;
      const size_t drow = (size_t)d.Wd * C;
      // Software pipeline: everything the NEXT step needs from memory (S source rows of dz / y, the destination
      // pixel's activation and second gradient) is requested before the arithmetic of the current step, so a
      // thread always has one step of loads in flight.
      float4 rz[S][3], ry[RECOMP ? 1 : S][3], ray, rby, rgo;
      float4 rna[RECOMP ? 3 : 1], rnb[(RECOMP && HAS_B) ? 3 : 1];   // RECOMP: raw input row oy+1 at columns ox-1..ox+1
      const bool vxm = ox > 0, vxp = ox + 1 < d.Wd;
      const size_t nxm = (size_t)(vxm ? ox - 1 : ox) * C, nx = (size_t)ox * C, nxp = (size_t)(vxp ? ox + 1 : ox) * C;
      auto issue = [&](int oy, size_t off) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
          const int iy = S == 1 ? oy + 1 : 2 * oy + u;
          const size_t row = clip + (size_t)((iy >= 0 && iy < d.Hs) ? iy : 0) * d.Ws * C;
          rz[u][0] = ld4(gm.dz + row + (size_t)cx0 * C);
          rz[u][1] = ld4(gm.dz + row + (size_t)cx1 * C);
          rz[u][2] = ld4(gm.dz + row + (size_t)cx2 * C);
          if (!RECOMP) {
            ry[u][0] = ld4(gm.y + row + (size_t)cx0 * C);
            ry[u][1] = ld4(gm.y + row + (size_t)cx1 * C);
            ry[u][2] = ld4(gm.y + row + (size_t)cx2 * C);
          }
        }
        if (RECOMP) {
          const size_t nrow = ((size_t)bi * d.Hd + (oy + 1 < d.Hd ? oy + 1 : oy)) * d.Wd * C + c;
          rna[0] = ld4(am.y + nrow + nxm);
          rna[1] = ld4(am.y + nrow + nx);
          rna[2] = ld4(am.y + nrow + nxp);
          if (HAS_B) {
            rnb[0] = ld4(bm.y + nrow + nxm);
            rnb[1] = ld4(bm.y + nrow + nx);
            rnb[2] = ld4(bm.y + nrow + nxp);
          }
        } else {
          ray = ld4(am.y + off);
          rby = HAS_B ? ld4(bm.y + off) : f4(0.f);
        }
        rgo = HAS_GO ? ld4(em.g_other + off) : f4(0.f);
      };
      float4 a0[RECOMP ? 3 : 1], a1[RECOMP ? 3 : 1], ymid = f4(0.f);   // activated input rows oy, oy+1; raw centre of row oy
      if (RECOMP) {
        const Lazy4 pa = bt_src(am, tc, C, BT_A), pb = bt_src(bm, tc, C, BT_B);
        const LazyGrad4 gr = bt_grad(gm, tc, C);
        float4 wk[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) wk[q] = bt_w(tc, C, q);
        const size_t row = ((size_t)bi * d.Hd + r0) * d.Wd * C + c;
        const float4 q0 = ld4(am.y + row + nxm), q1 = ld4(am.y + row + nx), q2 = ld4(am.y + row + nxp);
        a0[0] = lazy4_act(pa, q0);
        a0[1] = lazy4_act(pa, q1);
        a0[2] = lazy4_act(pa, q2);
        if (HAS_B) {
          a0[0] = a0[0] + lazy4_at(pb, row + nxm);
          a0[1] = a0[1] + lazy4_at(pb, row + nx);
          a0[2] = a0[2] + lazy4_at(pb, row + nxp);
        }
        a0[0] = keep4(vxm, a0[0]);
        a0[2] = keep4(vxp, a0[2]);
        ymid = q1;
        // source row 2*r0 - 1 above a band that does not start at the top: its y comes from input rows r0 - 1 and r0
        // like every other row's (the layer's output is never read, so it need not exist in memory)
        const bool vp = r0 > 0;
        const size_t prow = ((size_t)bi * d.Hd + (vp ? r0 - 1 : r0)) * d.Wd * C + c;
        float4 ap[3];
        ap[0] = lazy4_at(pa, prow + nxm);
        ap[1] = lazy4_at(pa, prow + nx);
        ap[2] = lazy4_at(pa, prow + nxp);
        if (HAS_B) {
          ap[0] = ap[0] + lazy4_at(pb, prow + nxm);
          ap[1] = ap[1] + lazy4_at(pb, prow + nx);
          ap[2] = ap[2] + lazy4_at(pb, prow + nxp);
        }
        ap[0] = keep4(vxm, ap[0]);
        ap[2] = keep4(vxp, ap[2]);
        float4 yb[3];
        yb[0] = fma4(a0[1], wk[0], fma4(a0[0], wk[2], fma4(ap[1], wk[6], ap[0] * wk[8])));
        yb[1] = fma4(a0[1], wk[1], ap[1] * wk[7]);
        yb[2] = fma4(a0[2], wk[0], fma4(a0[1], wk[2], fma4(ap[2], wk[6], ap[1] * wk[8])));
        const size_t srow = clip + (size_t)(vp ? 2 * r0 - 1 : 0) * d.Ws * C;
        t0[0] = keep4(vp && v0, lazygrad4_eval(gr, ld4(gr.dz + srow + (size_t)cx0 * C), yb[0]));
        t0[1] = keep4(vp && v1, lazygrad4_eval(gr, ld4(gr.dz + srow + (size_t)cx1 * C), yb[1]));
        t0[2] = keep4(vp && v2, lazygrad4_eval(gr, ld4(gr.dz + srow + (size_t)cx2 * C), yb[2]));
      }
      issue(r0, ioff);
#pragma unroll 1
      for (int oy = r0; oy < r1; ++oy, ioff += drow) {
        const float* t = bt_here(tc);
        const LazyGrad4 gr = bt_grad(gm, t, C);
        if (S == 1) {
          const bool vy = oy + 1 < d.Hs;
#pragma unroll
          for (int j = 0; j < 3; ++j) t2[j] = lazygrad4_eval(gr, rz[0][j], ry[0][j]);
          t2[0] = keep4(vy && v0, t2[0]);
          t2[1] = keep4(vy && v1, t2[1]);
          t2[2] = keep4(vy && v2, t2[2]);
        } else {
          const bool vy2 = 2 * oy + 1 < d.Hs;   // row 2*oy always exists
          if (RECOMP) {
            const bool vn = oy + 1 < d.Hd;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              a1[j] = lazy4_act(bt_src(am, t, C, BT_A), rna[j]);
              if (HAS_B) a1[j] = a1[j] + lazy4_act(bt_src(bm, t, C, BT_B), rnb[j]);
            }
            a1[0] = keep4(vn && vxm, a1[0]);
            a1[1] = keep4(vn, a1[1]);
            a1[2] = keep4(vn && vxp, a1[2]);
            float4 y0[3], y1[3];
            y0[0] = fma4(a0[0], bt_w(t, C, 5), a0[1] * bt_w(t, C, 3));
            y0[1] = a0[1] * bt_w(t, C, 4);
            y0[2] = fma4(a0[2], bt_w(t, C, 3), a0[1] * bt_w(t, C, 5));
            y1[0] = fma4(a1[1], bt_w(t, C, 0), fma4(a1[0], bt_w(t, C, 2), fma4(a0[1], bt_w(t, C, 6), a0[0] * bt_w(t, C, 8))));
            y1[1] = fma4(a1[1], bt_w(t, C, 1), a0[1] * bt_w(t, C, 7));
            y1[2] = fma4(a1[2], bt_w(t, C, 0), fma4(a1[1], bt_w(t, C, 2), fma4(a0[2], bt_w(t, C, 6), a0[1] * bt_w(t, C, 8))));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              t1[j] = lazygrad4_eval(gr, rz[0][j], y0[j]);
              t2[j] = lazygrad4_eval(gr, rz[S - 1][j], y1[j]);
            }
          } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              t1[j] = lazygrad4_eval(gr, rz[0][j], ry[0][j]);
              t2[j] = lazygrad4_eval(gr, rz[S - 1][j], ry[RECOMP ? 0 : S - 1][j]);
            }
          }
          t1[0] = keep4(v0, t1[0]);
          t1[1] = keep4(v1, t1[1]);
          t1[2] = keep4(v2, t1[2]);
          t2[0] = keep4(vy2 && v0, t2[0]);
          t2[1] = keep4(vy2 && v1, t2[1]);
          t2[2] = keep4(vy2 && v2, t2[2]);
        }
        const float4 yraw = RECOMP ? ymid : ray;
        float4 av;
        if (RECOMP) {
          av = a0[1];
          ymid = rna[1];
#pragma unroll
          for (int j = 0; j < 3; ++j) a0[j] = a1[j];
        } else {
          av = lazy4_act(bt_src(am, t, C, BT_A), yraw);
          if (HAS_B) av = av + lazy4_act(bt_src(bm, t, C, BT_B), rby);
        }
        const float4 go = rgo;
        issue(oy + 1 < r1 ? oy + 1 : oy, oy + 1 < r1 ? ioff + drow : ioff);   // next step (a harmless repeat at the end)
        __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arithmetic below
        const float* u = bt_here(tc);   // a second window: the taps are not held across the requests above
        float4 acc = t0[0] * bt_w(u, C, 0);
        acc = fma4(t0[1], bt_w(u, C, 1), acc);
        acc = fma4(t0[2], bt_w(u, C, 2), acc);
        acc = fma4(t1[0], bt_w(u, C, 3), acc);
        acc = fma4(t1[1], bt_w(u, C, 4), acc);
        acc = fma4(t1[2], bt_w(u, C, 5), acc);
        acc = fma4(t2[0], bt_w(u, C, 6), acc);
        acc = fma4(t2[1], bt_w(u, C, 7), acc);
        acc = fma4(t2[2], bt_w(u, C, 8), acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          dwk[j] = fma4(t0[j], av, dwk[j]);
          dwk[3 + j] = fma4(t1[j], av, dwk[3 + j]);
          dwk[6 + j] = fma4(t2[j], av, dwk[6 + j]);
        }
        acc = epi4_core(bt_epi(em, u, C), acc + go, yraw, dwk[9], dwk[10]);
        if (g_in != nullptr) st4(g_in + ioff, acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (S == 1) {
            t0[j] = t1[j];
            t1[j] = t2[j];
          } else {
            t0[j] = t2[j];
          }
        }
      }
    }
    if (FLIP) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 t = dwk[k];
        dwk[k] = dwk[8 - k];
        dwk[8 - k] = t;
      }
    }
  }
  block_channel_reduce<11>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// backward of a stride-2 Conv2d: cell geometry over dy
template <bool HAS_B, bool HAS_GO>
__global__ __launch_bounds__(256, 2) void walk_cell_bwd_kernel(LazyGrad G, LazySrc A, LazySrc Bsrc,
                                                            const float* __restrict__ w, float* __restrict__ g_in,
                                                            float* __restrict__ dw_slab, BwdEpilogue E, WalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 dwk[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) dwk[k] = f4(0.f);
  if (pl == 0) bwd_table_fill(red, C, c, G, A, Bsrc, E, w, false);
  __syncthreads();
  if (pl < PPB) {
    // scalar members only; the per-channel vectors are read from the table where they are used
    const Lazy4 am = lazy4_init(A, C, c), bm = lazy4_init(Bsrc, C, c);
    const Epi4 em = epi4_init(E, C, c);
    const LazyGrad4 gm = lazygrad4_init(G, C, c);
    const float* const tc = red + c;
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      int bi, m0, m1, mx;
      if (!walk_slot(d, g * PPB + pl, d.Hs, bi, m0, m1, mx)) continue;
      const bool vx1 = mx + 1 < d.Ws;
      const bool dx1 = 2 * mx + 1 < d.Wd;
      const size_t clip = (size_t)bi * d.Hs * d.Ws * C + c;
      auto load_row = [&](int iy, float4& r0, float4& r1) {   // branch-free, see walk_gather_fwd_kernel
        const LazyGrad4 gr = bt_grad(gm, tc, C);
        const bool vy = iy < d.Hs;
        const size_t o0 = clip + ((size_t)(vy ? iy : 0) * d.Ws + mx) * C;
        const float4 x0 = lazygrad4_at(gr, o0), x1 = lazygrad4_at(gr, vx1 ? o0 + C : o0);
        r0 = keep4(vy, x0);
        r1 = keep4(vy && vx1, x1);
      };
      float4 s00, s01, s10, s11;
      load_row(m0, s00, s01);
      size_t ioff = (((size_t)bi * d.Hd + 2 * m0) * d.Wd + 2 * mx) * C + c;
      const size_t drow = (size_t)d.Wd * C;
      // Software pipeline (see walk_gather_bwd_kernel): the source row and the four raw destination activations of the
      // NEXT cell are requested before the arithmetic of the current one.  A destination pixel outside the tensor is
      // redirected to (2m, 2mx): its loads are harmless and its results are neither accumulated nor stored.
      float4 rz0, rz1, ry0, ry1, ra[4];
      auto issue = [&](int m, size_t off) {
        const int iy = m + 1;
        const size_t o0 = clip + ((size_t)(iy < d.Hs ? iy : 0) * d.Ws + mx) * C, o1 = vx1 ? o0 + C : o0;
        rz0 = ld4(gm.dz + o0);
        rz1 = ld4(gm.dz + o1);
        ry0 = ld4(gm.y + o0);
        ry1 = ld4(gm.y + o1);
        const bool k10 = 2 * m + 1 < d.Hd;
        ra[0] = ld4(am.y + off);
        ra[1] = ld4(am.y + (dx1 ? off + C : off));
        ra[2] = ld4(am.y + (k10 ? off + drow : off));
        ra[3] = ld4(am.y + ((k10 && dx1) ? off + drow + C : off));
      };
      const float* t = tc;   // table window of the current step
      auto activate = [&](float4 yraw, size_t off) {
        float4 av = lazy4_act(bt_src(am, t, C, BT_A), yraw);
        if (HAS_B) av = av + lazy4_at(bt_src(bm, t, C, BT_B), off);
        return av;
      };
      auto finish = [&](float4 g, float4 yraw, size_t off, bool ok) {   // + second gradient, mask, producer sums
        if (HAS_GO) g = g + keep4(ok, ld4(em.g_other + off));
        return epi4_core(bt_epi(em, t, C), g, yraw, dwk[9], dwk[10]);
      };
      issue(m0, ioff);
#pragma unroll 1
      for (int m = m0; m < m1; ++m, ioff += 2 * drow) {
        const bool vy = m + 1 < d.Hs;
        t = bt_here(tc);
        const LazyGrad4 gr = bt_grad(gm, t, C);
        s10 = keep4(vy, lazygrad4_eval(gr, rz0, ry0));
        s11 = keep4(vy && vx1, lazygrad4_eval(gr, rz1, ry1));
        const bool ok01 = dx1, ok10 = 2 * m + 1 < d.Hd, ok11 = ok10 && dx1;
        const size_t o00 = ioff, o01 = ok01 ? ioff + C : ioff, o10 = ok10 ? ioff + drow : ioff,
                     o11 = ok11 ? ioff + drow + C : ioff;
        const float4 y00 = ra[0], y01 = ra[1], y10 = ra[2], y11 = ra[3];
        issue(m + 1 < m1 ? m + 1 : m, m + 1 < m1 ? ioff + 2 * drow : ioff);
        __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arithmetic below
        t = bt_here(tc);
        const float4 a00 = activate(y00, o00), a01 = keep4(ok01, activate(y01, o01));
        const float4 a10 = keep4(ok10, activate(y10, o10)), a11 = keep4(ok11, activate(y11, o11));
        float4 g00 = s00 * bt_w(t, C, 4);
        float4 g01 = fma4(s01, bt_w(t, C, 3), s00 * bt_w(t, C, 5));
        float4 g10 = fma4(s10, bt_w(t, C, 1), s00 * bt_w(t, C, 7));
        float4 g11 = s00 * bt_w(t, C, 8);
        g11 = fma4(s01, bt_w(t, C, 6), g11);
        g11 = fma4(s10, bt_w(t, C, 2), g11);
        g11 = fma4(s11, bt_w(t, C, 0), g11);
        dwk[4] = fma4(s00, a00, dwk[4]);
        dwk[5] = fma4(s00, a01, dwk[5]);
        dwk[3] = fma4(s01, a01, dwk[3]);
        dwk[7] = fma4(s00, a10, dwk[7]);
        dwk[1] = fma4(s10, a10, dwk[1]);
        dwk[8] = fma4(s00, a11, dwk[8]);
        dwk[6] = fma4(s01, a11, dwk[6]);
        dwk[2] = fma4(s10, a11, dwk[2]);
        dwk[0] = fma4(s11, a11, dwk[0]);
        g00 = finish(g00, y00, o00, true);
        g01 = finish(keep4(ok01, g01), y01, o01, ok01);
        g10 = finish(keep4(ok10, g10), y10, o10, ok10);
        g11 = finish(keep4(ok11, g11), y11, o11, ok11);
        if (g_in != nullptr) {
          st4(g_in + o00, g00);
          if (ok01) st4(g_in + o01, g01);
          if (ok10) st4(g_in + o10, g10);
          if (ok11) st4(g_in + o11, g11);
        }
        s00 = s10;
        s01 = s11;
      }
    }
  }
  block_channel_reduce<11>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 11 * C);
}

// Band layout: enough column slots to give every CU several workgroups, bands of at least 8 walked rows.
inline int walk_env(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
inline void walk_plan(WalkDims& d, int walked_rows, int max_blocks, int& grid) {
  const int PPB = 256 / (d.C / 4);
  const long slots1 = (long)d.B * d.wown;
  int nb = 1;
  const long want = 2048L * PPB;
  if (slots1 < want) {
    nb = (int)((want + slots1 - 1) / slots1);
    const int max_nb = walked_rows / 8 > 1 ? walked_rows / 8 : 1;
    if (nb > max_nb) nb = max_nb;
  }
  // Tall images: bands of 16 rows on at most 1024 workgroups.  Measured on 1280 x 81 x 33 tensors (scripts/walk_sweep.py):
  // 4.5 -> 5.0 TB/s forward and 4.7 -> 5.0 TB/s backward for 32 channels, nothing lost for 16; bands of 4-8 rows or
  // one workgroup per band lose 10-30 %, so the order in which the bands are visited is not what bounds these kernels.
  if (walked_rows >= 64) {
    nb = (walked_rows + 15) / 16;
    if (max_blocks > 1024) max_blocks = 1024;
  }
  // NOISE_WALK_BH / NOISE_WALK_CAP (diagnostic, scripts/walk_sweep.py): band height and grid cap
  const int ebh = walk_env("NOISE_WALK_BH", 0);
  if (ebh > 0) nb = (walked_rows + ebh - 1) / ebh;
  max_blocks = walk_env("NOISE_WALK_CAP", max_blocks);
  d.bh = (walked_rows + nb - 1) / nb;
  d.nb = (walked_rows + d.bh - 1) / d.bh;
  d.ngroups = (int)(((long)d.B * d.nb * d.wown + PPB - 1) / PPB);
  grid = d.ngroups < max_blocks ? d.ngroups : max_blocks;
}


// ------------------------------------------------------------------------------------------------
// Last layer (train/models.py:1164-1171 with the crop of :610 and the mask of :762-763), stride 2, C/4 a power
// of two in [4,16]: the same two geometries with the channel sum spread over the C/4 lanes of a column.
struct LastWalkDims {
  int B, Hi, Wi, C;     // input (B, Hi, Wi, C)
  int H, W, oy0, ox0;   // output crop: H x W starting at (oy0, ox0) of the full (2Hi-1) x (2Wi-1) output
  int m_lo, x_lo;       // first walked input row / owned input column (earlier ones only feed cropped outputs)
  int wown, nb, bh, ngroups;
};

// forward: filt(Y, X) = bias + sum_c sum_k a[(Y+1-ky)/2][(X+1-kx)/2][c] * w[c][k];  enh = clamp(x/max + filt)*max
template <bool HAS_B>
__global__ __launch_bounds__(256) void last_cell_fwd_kernel(LazySrc A, LazySrc Bsrc, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ logmag,
                                                            const float* __restrict__ absmax, float* __restrict__ enh,
                                                            float* __restrict__ filt_out, LastWalkDims d) {
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  const Lazy4 a = lazy4_init(A, C, c), b = lazy4_init(Bsrc, C, c);
  float4 wk[9];
  load_taps(w, c, false, wk);
  const float bv = bias[0];
  const int nslots = d.B * d.nb * d.wown;
  const int p = (cg >> 1) & 1, q = cg & 1;   // lanes 0..3 of a column finish outputs (p, q) of its 2x2 cell
  for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {   // uniform trip counts: the lane shuffles need whole waves
    const int slot = g * PPB + pl;
    const bool valid = slot < nslots;
    const int sl = valid ? slot : 0;
    const int mx = d.x_lo + sl % d.wown;
    const int u = sl / d.wown;
    const int band = u % d.nb, bi = u / d.nb;
    const int m0 = d.m_lo + band * d.bh;
    const int m1 = valid ? (m0 + d.bh < d.Hi ? m0 + d.bh : d.Hi) : m0;
    const bool vx1 = mx + 1 < d.Wi;
    const size_t clip = (size_t)bi * d.Hi * d.Wi * C + c;
    auto load_row = [&](int iy, float4& r0, float4& r1) {   // branch-free: clamped address, zeroed afterwards
      const bool vy = iy < d.Hi;
      const size_t o0 = clip + ((size_t)(vy ? iy : 0) * d.Wi + mx) * C, o1 = vx1 ? o0 + C : o0;
      float4 x0 = lazy4_at(a, o0), x1 = lazy4_at(a, o1);
      if (HAS_B) {
        x0 = x0 + lazy4_at(b, o0);
        x1 = x1 + lazy4_at(b, o1);
      }
      r0 = keep4(vy, x0);
      r1 = keep4(vy && vx1, x1);
    };
    const float mxv = absmax ? absmax[bi] + 1e-6f : 1.f;
    float4 s00, s01, s10, s11;
    load_row(m0, s00, s01);
    const int X = 2 * mx + q - d.ox0;
    const int Xc = X < 0 ? 0 : (X < d.W ? X : d.W - 1);
    for (int m = m0; m < m0 + d.bh; ++m) {
      load_row(m + 1, s10, s11);
      // the log-magnitude this lane may need at the end of the step is requested now, from a clamped address
      const int Y = 2 * m + p - d.oy0;
      const size_t oc = ((size_t)bi * d.H + (Y < 0 ? 0 : (Y < d.H ? Y : d.H - 1))) * d.W + Xc;
      const float lmv = enh ? logmag[oc] : 0.f;
      float4 t00 = s00 * wk[4];
      float4 t01 = fma4(s01, wk[3], s00 * wk[5]);
      float4 t10 = fma4(s10, wk[1], s00 * wk[7]);
      float4 t11 = s00 * wk[8];
      t11 = fma4(s01, wk[6], t11);
      t11 = fma4(s10, wk[2], t11);
      t11 = fma4(s11, wk[0], t11);
      float p00 = (t00.x + t00.y) + (t00.z + t00.w), p01 = (t01.x + t01.y) + (t01.z + t01.w);
      float p10 = (t10.x + t10.y) + (t10.z + t10.w), p11 = (t11.x + t11.y) + (t11.z + t11.w);
      for (int o = CG >> 1; o > 0; o >>= 1) {
        p00 += __shfl_xor(p00, o, 64);
        p01 += __shfl_xor(p01, o, 64);
        p10 += __shfl_xor(p10, o, 64);
        p11 += __shfl_xor(p11, o, 64);
      }
      const float f = (p ? (q ? p11 : p10) : (q ? p01 : p00)) + bv;
      if (cg < 4 && m < m1 && Y >= 0 && Y < d.H && X >= 0 && X < d.W) {
        if (filt_out) filt_out[oc] = f;
        if (enh) {
          const float v = lmv / mxv + f;
          enh[oc] = fminf(fmaxf(v, -1.f), 1.f) * mxv;
        }
      }
      s00 = s10;
      s01 = s11;
    }
  }
}

// backward: dy0(Y, X) = d_enh * max * [|enh| < max] inside the crop, 0 outside;
//   g_in[i][c] = sum_k dy0(2i - 1 + k) * w[c][k];   dW[c][k] += a[i][c] * dy0(2i - 1 + k);   db = sum dy0
// Slab rows are 12*C floats: 9 taps | bias gradient at column 9*C | producer BatchNorm sums at [10C, 12C).
template <bool HAS_B, bool HAS_GO, bool COOP>
__global__ __launch_bounds__(256) void last_gather_bwd_kernel(const float* __restrict__ d_enh, const float* __restrict__ enh,
                                                              const float* __restrict__ absmax, LazySrc A, LazySrc Bsrc,
                                                              const float* __restrict__ w, float* __restrict__ g_in,
                                                              float* __restrict__ dw_slab, BwdEpilogue E, LastWalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 dwk[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) dwk[k] = f4(0.f);
  if (pl < PPB) {
    const BwdPixel px{lazy4_init(A, C, c), lazy4_init(Bsrc, C, c), epi4_init(E, C, c), g_in};
    float4 wk[9];
    load_taps(w, c, false, wk);
    float db = 0.f;
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      const int slot = g * PPB + pl;
      if (slot >= d.B * d.nb * d.wown) continue;
      const int ix = slot % d.wown;
      const int u = slot / d.wown;
      const int band = u % d.nb, bi = u / d.nb;
      const int r0 = band * d.bh, r1 = r0 + d.bh < d.Hi ? r0 + d.bh : d.Hi;
      const float mxv = absmax[bi] + 1e-6f;
      const int X0 = 2 * ix - 1 - d.ox0;   // cropped column of tap kx = 0
      const bool v0 = X0 >= 0 && X0 < d.W, v1 = X0 + 1 >= 0 && X0 + 1 < d.W, v2 = X0 + 2 >= 0 && X0 + 2 < d.W;
      const int c0 = v0 ? X0 : 0, c1 = v1 ? X0 + 1 : 0, c2 = v2 ? X0 + 2 : 0;
      const size_t oclip = (size_t)bi * d.H * d.W;
      auto load_row = [&](int Y, float (&r)[3]) {   // Y: cropped output row; branch-free
        const bool vy = Y >= 0 && Y < d.H;
        const size_t row = oclip + (size_t)(vy ? Y : 0) * d.W;
        const float e0 = enh[row + c0], e1 = enh[row + c1], e2 = enh[row + c2];
        const float g0 = d_enh[row + c0], g1 = d_enh[row + c1], g2 = d_enh[row + c2];
        r[0] = (vy && v0 && e0 > -mxv && e0 < mxv) ? g0 * mxv : 0.f;
        r[1] = (vy && v1 && e1 > -mxv && e1 < mxv) ? g1 * mxv : 0.f;
        r[2] = (vy && v2 && e2 > -mxv && e2 < mxv) ? g2 * mxv : 0.f;
      };
      float t0[3], t1[3], t2[3];
      load_row(2 * r0 - 1 - d.oy0, t0);
      size_t ioff = (((size_t)bi * d.Hi + r0) * d.Wi + ix) * C + c;
      const size_t drow = (size_t)d.Wi * C;
      // software pipeline (see walk_gather_bwd_kernel): the next input row's operands are requested first
      // COOP (C/4 a power of two >= 8): lane cg < 6 of a column fetches enh / d_enh of window element (cg / 3, cg % 3) and
      // forms its dy0; the six values are exchanged with lane shuffles (2 scalar loads per lane instead of 12).
      float re[COOP ? 1 : 2][COOP ? 1 : 3], rg[COOP ? 1 : 2][COOP ? 1 : 3];
      float4 ray, rby, rgo;
      const int ce = cg < 6 ? cg : 0, cu = ce / 3, cj = ce - 3 * cu;
      const int ccol = cj == 0 ? c0 : (cj == 1 ? c1 : c2);
      const bool cval = cj == 0 ? v0 : (cj == 1 ? v1 : v2);
      const int gbase = (int)(threadIdx.x & 63) & ~(CG - 1);
      auto issue = [&](int iy, size_t off) {
        if (COOP) {
          const int Y = 2 * iy + cu - d.oy0;
          const size_t row = oclip + (size_t)((Y >= 0 && Y < d.H) ? Y : 0) * d.W;
          re[0][0] = enh[row + ccol];
          rg[0][0] = d_enh[row + ccol];
        } else {
#pragma unroll
          for (int u = 0; u < (COOP ? 0 : 2); ++u) {
            const int Y = 2 * iy + u - d.oy0;
            const size_t row = oclip + (size_t)((Y >= 0 && Y < d.H) ? Y : 0) * d.W;
            re[u][0] = enh[row + c0];
            re[u][COOP ? 0 : 1] = enh[row + c1];
            re[u][COOP ? 0 : 2] = enh[row + c2];
            rg[u][0] = d_enh[row + c0];
            rg[u][COOP ? 0 : 1] = d_enh[row + c1];
            rg[u][COOP ? 0 : 2] = d_enh[row + c2];
          }
        }
        ray = ld4(px.a.y + off);
        rby = HAS_B ? ld4(px.b.y + off) : f4(0.f);
        rgo = HAS_GO ? ld4(px.epi.g_other + off) : f4(0.f);
      };
      issue(r0, ioff);
#pragma unroll 1
      for (int iy = r0; iy < r1; ++iy, ioff += drow) {
        const int Y1 = 2 * iy - d.oy0, Y2 = Y1 + 1;
        const bool vy1 = Y1 >= 0 && Y1 < d.H, vy2 = Y2 >= 0 && Y2 < d.H;
        if (COOP) {
          const bool vy = cu == 0 ? vy1 : vy2;
          const float mine = (vy && cval && re[0][0] > -mxv && re[0][0] < mxv) ? rg[0][0] * mxv : 0.f;
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            t1[j] = __shfl(mine, gbase + j, 64);
            t2[j] = __shfl(mine, gbase + 3 + j, 64);
          }
        } else {
          t1[0] = (vy1 && v0 && re[0][0] > -mxv && re[0][0] < mxv) ? rg[0][0] * mxv : 0.f;
          t1[1] = (vy1 && v1 && re[0][COOP ? 0 : 1] > -mxv && re[0][COOP ? 0 : 1] < mxv) ? rg[0][COOP ? 0 : 1] * mxv : 0.f;
          t1[2] = (vy1 && v2 && re[0][COOP ? 0 : 2] > -mxv && re[0][COOP ? 0 : 2] < mxv) ? rg[0][COOP ? 0 : 2] * mxv : 0.f;
          t2[0] = (vy2 && v0 && re[COOP ? 0 : 1][0] > -mxv && re[COOP ? 0 : 1][0] < mxv) ? rg[COOP ? 0 : 1][0] * mxv : 0.f;
          t2[1] = (vy2 && v1 && re[COOP ? 0 : 1][COOP ? 0 : 1] > -mxv && re[COOP ? 0 : 1][COOP ? 0 : 1] < mxv)
                      ? rg[COOP ? 0 : 1][COOP ? 0 : 1] * mxv : 0.f;
          t2[2] = (vy2 && v2 && re[COOP ? 0 : 1][COOP ? 0 : 2] > -mxv && re[COOP ? 0 : 1][COOP ? 0 : 2] < mxv)
                      ? rg[COOP ? 0 : 1][COOP ? 0 : 2] * mxv : 0.f;
        }
        const float4 yraw = ray;
        float4 av = lazy4_act(px.a, yraw);
        if (HAS_B) av = av + lazy4_act(px.b, rby);
        const float4 go = rgo;
        issue(iy + 1 < r1 ? iy + 1 : iy, iy + 1 < r1 ? ioff + drow : ioff);
        __builtin_amdgcn_sched_barrier(0);   // keep the requests ahead of the arithmetic below
        float4 acc = f4(t0[0]) * wk[0];
        acc = fma4(f4(t0[1]), wk[1], acc);
        acc = fma4(f4(t0[2]), wk[2], acc);
        acc = fma4(f4(t1[0]), wk[3], acc);
        acc = fma4(f4(t1[1]), wk[4], acc);
        acc = fma4(f4(t1[2]), wk[5], acc);
        acc = fma4(f4(t2[0]), wk[6], acc);
        acc = fma4(f4(t2[1]), wk[7], acc);
        acc = fma4(f4(t2[2]), wk[8], acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          dwk[j] = fma4(f4(t0[j]), av, dwk[j]);
          dwk[3 + j] = fma4(f4(t1[j]), av, dwk[3 + j]);
          dwk[6 + j] = fma4(f4(t2[j]), av, dwk[6 + j]);
        }
        // every output (Y, X) is tap (1|2, 1|2) of exactly one input pixel: those four carry the bias gradient
        db += (t1[1] + t1[2]) + (t2[1] + t2[2]);
        acc = epi4_core(px.epi, acc + go, yraw, dwk[10], dwk[11]);
        st4(g_in + ioff, acc);
#pragma unroll
        for (int j = 0; j < 3; ++j) t0[j] = t2[j];
      }
    }
    if (cg == 0) dwk[9].x = db;
  }
  block_channel_reduce<12>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 12 * C);
}

inline void last_walk_plan(LastWalkDims& d, bool fwd, int max_blocks, int& grid) {
  const int PPB = 256 / (d.C / 4);
  d.m_lo = fwd ? d.oy0 / 2 : 0;
  d.x_lo = fwd ? d.ox0 / 2 : 0;
  d.wown = d.Wi - d.x_lo;
  const int rows = d.Hi - d.m_lo;
  const long slots1 = (long)d.B * d.wown;
  int nb = 1;
  const long want = 2048L * PPB;
  if (slots1 < want) {
    nb = (int)((want + slots1 - 1) / slots1);
    const int max_nb = rows / 8 > 1 ? rows / 8 : 1;
    if (nb > max_nb) nb = max_nb;
  }
  d.bh = (rows + nb - 1) / nb;
  d.nb = (rows + d.bh - 1) / d.bh;
  d.ngroups = (int)(((long)d.B * d.nb * d.wown + PPB - 1) / PPB);
  grid = d.ngroups < max_blocks ? d.ngroups : max_blocks;
}

}  // namespace

namespace {

// ------------------------------------------------------------------------------------------------
// First layer (train/models.py:1151-1159 on the input padded by :589 and normalised by :614-622), stride 2:
// gather geometry over a single-channel input, so the 3x3 window is nine scalars shared by the channel lanes.
//   x_in(py, px) = 0 outside the padded input, pad_value inside the F.pad border, logmag / max elsewhere
struct FirstWalkDims {
  int B, H, W, hp, wp, Ho, Wo, C;
  float pad_value;
  int nb, bh, ngroups;
};

struct FirstWindow {
  const float* x;   // this clip's log-magnitudes (H, W)
  float mx;         // per-clip max + 1e-6 (divided, as the reference does)
  int H, W, hp, wp;
  float pad_value;
  int px0;          // padded column of tap kx = 0
  __device__ __forceinline__ void row(int py, float (&r)[3]) const {   // py: padded row; branch-free
    const int y = py - hp;
    const bool rin = py >= 0 && py < H + hp, rdata = y >= 0 && rin;
    const float* xr = x + (size_t)(rdata ? y : 0) * W;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int px = px0 + j, xx = px - wp;
      const bool cin = px >= 0 && px < W + wp, cdata = xx >= 0 && cin;
      const float v = xr[cdata ? xx : 0] / mx;
      r[j] = (rin && cin) ? ((rdata && cdata) ? v : pad_value) : 0.f;
    }
  }
  // one element of the window: padded row py, tap column j
  __device__ __forceinline__ float at(int py, int j) const {
    const int y = py - hp, px = px0 + j, xx = px - wp;
    const bool rin = py >= 0 && py < H + hp, rdata = y >= 0 && rin;
    const bool cin = px >= 0 && px < W + wp, cdata = xx >= 0 && cin;
    const float v = x[(size_t)(rdata ? y : 0) * W + (cdata ? xx : 0)] / mx;
    return (rin && cin) ? ((rdata && cdata) ? v : pad_value) : 0.f;
  }
  // The two new rows of a stride-2 step (padded rows py, py + 1).  With COOP (C/4 a power of two >= 8: the lanes of a
  // column are 8+ consecutive lanes of one wave) lane cg < 6 fetches and normalises element (cg / 3, cg % 3) and
  // the six values are exchanged with lane shuffles: one load and one division per lane instead of six.
  template <bool COOP>
  __device__ __forceinline__ void rows2(int py, int cg, int CG, float (&r1)[3], float (&r2)[3]) const {
    if (COOP) {
      const int e = cg < 6 ? cg : 0;
      const float mine = at(py + e / 3, e % 3);
      const int base = (int)(threadIdx.x & 63) & ~(CG - 1);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        r1[j] = __shfl(mine, base + j, 64);
        r2[j] = __shfl(mine, base + 3 + j, 64);
      }
    } else {
      row(py, r1);
      row(py + 1, r2);
    }
  }
};

template <bool COOP>
__global__ __launch_bounds__(256) void first_walk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ absmax,
                                                             const float* __restrict__ w, float* __restrict__ out,
                                                             float* __restrict__ slab, FirstWalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if (pl < PPB) {
    float4 wk[9];
    load_taps(w, c, false, wk);
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      const int slot = g * PPB + pl;
      if (slot >= d.B * d.nb * d.Wo) continue;
      const int ox = slot % d.Wo;
      const int u = slot / d.Wo;
      const int band = u % d.nb, bi = u / d.nb;
      const int r0 = band * d.bh, r1 = r0 + d.bh < d.Ho ? r0 + d.bh : d.Ho;
      const FirstWindow win{x + (size_t)bi * d.H * d.W, absmax[bi] + 1e-6f, d.H, d.W, d.hp, d.wp, d.pad_value,
                            2 * ox - 1};
      float t0[3], t1[3], t2[3];
      win.row(2 * r0 - 1, t0);
      float* o = out + (((size_t)bi * d.Ho + r0) * d.Wo + ox) * C + c;
      // the window element this lane contributes to the NEXT step is requested before this step's arithmetic
      const int e = cg < 6 ? cg : 0;
      float mine = COOP ? win.at(2 * r0 + e / 3, e % 3) : 0.f;
      for (int oy = r0; oy < r1; ++oy, o += (size_t)d.Wo * C) {
        if (COOP) {
          const int base = (int)(threadIdx.x & 63) & ~(CG - 1);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            t1[j] = __shfl(mine, base + j, 64);
            t2[j] = __shfl(mine, base + 3 + j, 64);
          }
          mine = win.at(2 * (oy + 1) + e / 3, e % 3);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          win.row(2 * oy, t1);
          win.row(2 * oy + 1, t2);
        }
        float4 acc = f4(t0[0]) * wk[0];
        acc = fma4(f4(t0[1]), wk[1], acc);
        acc = fma4(f4(t0[2]), wk[2], acc);
        acc = fma4(f4(t1[0]), wk[3], acc);
        acc = fma4(f4(t1[1]), wk[4], acc);
        acc = fma4(f4(t1[2]), wk[5], acc);
        acc = fma4(f4(t2[0]), wk[6], acc);
        acc = fma4(f4(t2[1]), wk[7], acc);
        acc = fma4(f4(t2[2]), wk[8], acc);
        st4(o, acc);
        s1 = s1 + acc;
        s2 = fma4(acc, acc, s2);
#pragma unroll
        for (int j = 0; j < 3; ++j) t0[j] = t2[j];
      }
    }
  }
  if (slab != nullptr) {
    const float4 v[2] = {s1, s2};
    block_channel_reduce<2>(v, C, CG, PPB, red, slab + (size_t)blockIdx.x * 2 * C);
  }
}

// dW[c][k] = sum_o dy[o][c] * x_in(2o - 1 + k)   (the input is data: no input gradient)
// RECOMP: the layer's raw output y, which the lazy BatchNorm backward needs, is rebuilt from the input window the
// thread holds anyway (9 multiply-adds per channel) instead of being read back (C floats per pixel).
template <bool COOP, bool RECOMP>
__global__ __launch_bounds__(256) void first_walk_wgrad_kernel(LazyGrad G, const float* __restrict__ x,
                                                               const float* __restrict__ absmax,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ dw_slab, FirstWalkDims d) {
  extern __shared__ float red[];
  const int C = d.C, CG = C >> 2, PPB = 256 / CG;
  const int tid = threadIdx.x, pl = tid / CG, cg = tid - pl * CG, c = 4 * cg;
  float4 dwk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) dwk[k] = f4(0.f);
  if (pl < PPB) {
    const LazyGrad4 gr = lazygrad4_init(G, C, c);
    float4 wk[9];
    if (RECOMP) load_taps(w, c, false, wk);
    for (int g = blockIdx.x; g < d.ngroups; g += gridDim.x) {
      const int slot = g * PPB + pl;
      if (slot >= d.B * d.nb * d.Wo) continue;
      const int ox = slot % d.Wo;
      const int u = slot / d.Wo;
      const int band = u % d.nb, bi = u / d.nb;
      const int r0 = band * d.bh, r1 = r0 + d.bh < d.Ho ? r0 + d.bh : d.Ho;
      const FirstWindow win{x + (size_t)bi * d.H * d.W, absmax[bi] + 1e-6f, d.H, d.W, d.hp, d.wp, d.pad_value,
                            2 * ox - 1};
      float t0[3], t1[3], t2[3];
      win.row(2 * r0 - 1, t0);
      size_t off = (((size_t)bi * d.Ho + r0) * d.Wo + ox) * C + c;
      const size_t orow = (size_t)d.Wo * C;
      // software pipeline: the window element this lane contributes to the NEXT step and the next gradient row are
      // requested before the arithmetic of the current step
      const int e = cg < 6 ? cg : 0;
      float mine = COOP ? win.at(2 * r0 + e / 3, e % 3) : 0.f;
      float4 nz = ld4(gr.dz + off), ny = RECOMP ? f4(0.f) : ld4(gr.y + off);   // gr.y aliases dz without a BatchNorm
      for (int oy = r0; oy < r1; ++oy, off += orow) {
        if (COOP) {
          const int base = (int)(threadIdx.x & 63) & ~(CG - 1);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            t1[j] = __shfl(mine, base + j, 64);
            t2[j] = __shfl(mine, base + 3 + j, 64);
          }
        } else {
          win.row(2 * oy, t1);
          win.row(2 * oy + 1, t2);
        }
        const float4 cz = nz, cy = ny;
        const size_t noff = oy + 1 < r1 ? off + orow : off;
        if (COOP) mine = win.at(2 * (oy + 1) + e / 3, e % 3);
        nz = ld4(gr.dz + noff);
        if (!RECOMP) ny = ld4(gr.y + noff);
        __builtin_amdgcn_sched_barrier(0);
        float4 dy;
        if (RECOMP) {
          float4 y = f4(t0[0]) * wk[0];
#pragma unroll
          for (int j = 1; j < 3; ++j) y = fma4(f4(t0[j]), wk[j], y);
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            y = fma4(f4(t1[j]), wk[3 + j], y);
            y = fma4(f4(t2[j]), wk[6 + j], y);
          }
          dy = lazygrad4_eval(gr, cz, y);
        } else {
          dy = lazygrad4_eval(gr, cz, cy);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          dwk[j] = fma4(dy, f4(t0[j]), dwk[j]);
          dwk[3 + j] = fma4(dy, f4(t1[j]), dwk[3 + j]);
          dwk[6 + j] = fma4(dy, f4(t2[j]), dwk[6 + j]);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) t0[j] = t2[j];
      }
    }
  }
  block_channel_reduce<9>(dwk, C, CG, PPB, red, dw_slab + (size_t)blockIdx.x * 9 * C);
}

inline void first_walk_plan(FirstWalkDims& d, int max_blocks, int& grid) {
  const int PPB = 256 / (d.C / 4);
  const long slots1 = (long)d.B * d.Wo;
  int nb = 1;
  const long want = 2048L * PPB;
  if (slots1 < want) {
    nb = (int)((want + slots1 - 1) / slots1);
    const int max_nb = d.Ho / 8 > 1 ? d.Ho / 8 : 1;
    if (nb > max_nb) nb = max_nb;
  }
  d.bh = (d.Ho + nb - 1) / nb;
  d.nb = (d.Ho + d.bh - 1) / d.bh;
  d.ngroups = (int)(((long)d.B * d.nb * d.Wo + PPB - 1) / PPB);
  grid = d.ngroups < max_blocks ? d.ngroups : max_blocks;
}

}  // namespace
