// Device-side loaders for lazily normalised tensors (see common.h: LazySrc / LazyGrad).
// All activations are NHWC fp32; channel counts are multiples of 4 so a lane moves float4.
#pragma once
#include "common.h"

#ifdef __HIPCC__
enum { ACT_NONE = 0, ACT_RELU6 = 1 };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
// Packed fp32 arithmetic: two lanes of a float4 per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32); every
// operation is the IEEE one, so results equal the scalar forms bit for bit.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ float4 pack4(f32x2 lo, f32x2 hi) { return make_float4(lo.x, lo.y, hi.x, hi.y); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) {
  return pack4(f32x2{a.x, a.y} + f32x2{b.x, b.y}, f32x2{a.z, a.w} + f32x2{b.z, b.w});
}
__device__ __forceinline__ float4 operator*(float4 a, float4 b) {
  return pack4(f32x2{a.x, a.y} * f32x2{b.x, b.y}, f32x2{a.z, a.w} * f32x2{b.z, b.w});
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return pack4(__builtin_elementwise_fma(f32x2{a.x, a.y}, f32x2{b.x, b.y}, f32x2{c.x, c.y}),
               __builtin_elementwise_fma(f32x2{a.z, a.w}, f32x2{b.z, b.w}, f32x2{c.z, c.w}));
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
  return pack4(f32x2{a.x, a.y} - f32x2{b.x, b.y}, f32x2{a.z, a.w} - f32x2{b.z, b.w});
}
__device__ __forceinline__ float4 relu6_4(float4 v) { return make_float4(relu6f(v.x), relu6f(v.y), relu6f(v.z), relu6f(v.w)); }

// Per-thread view of one lazy source restricted to 4 consecutive channels c..c+3.
struct Lazy4 {
  const float* y;
  float4 scale, shift;
  float lo, hi;   // clamp bounds of the activation: (0, 6) for ReLU6, (-inf, +inf) for none, so it needs no branch
  int act;
  bool present;
};

// min(max(v, lo), hi) as one median-of-three per lane (lo <= hi; the bounds may be infinite)
__device__ __forceinline__ float4 clamp4(float4 v, float lo, float hi) {
  return make_float4(__builtin_amdgcn_fmed3f(v.x, lo, hi), __builtin_amdgcn_fmed3f(v.y, lo, hi),
                     __builtin_amdgcn_fmed3f(v.z, lo, hi), __builtin_amdgcn_fmed3f(v.w, lo, hi));
}

__device__ __forceinline__ Lazy4 lazy4_init(const LazySrc& s, int C, int c) {
  Lazy4 r;
  r.y = s.y;
  r.present = s.y != nullptr;
  r.act = s.relu6;
  r.lo = s.relu6 ? 0.f : -INFINITY;
  r.hi = s.relu6 ? 6.f : INFINITY;
  r.scale = f4(1.f);
  r.shift = f4(0.f);
  if (r.present && s.coef != nullptr) {
    r.scale = ld4(s.coef + c);
    r.shift = ld4(s.coef + C + c);
  }
  return r;
}

// value of the source at flat element offset `off` (= pixel*C + c)
// activation of an already loaded raw value
__device__ __forceinline__ float4 lazy4_act(const Lazy4& s, float4 raw) {
  return clamp4(fma4(raw, s.scale, s.shift), s.lo, s.hi);
}

__device__ __forceinline__ float4 lazy4_at(const Lazy4& s, size_t off) { return lazy4_act(s, ld4(s.y + off)); }

// Per-thread view of a lazy gradient restricted to 4 consecutive channels.
// coef3 holds six per-channel vectors [s | k1_hi | k1_lo | a2 | mean_hi | mean_lo] and
//     dy = s*((dz - k1_hi) - k1_lo) + a2*((y - mean_hi) - mean_lo)
// i.e. BatchNorm backward in its centred form.  k1 (the mean of dz) and the batch mean are carried as
// hi+lo float pairs: when the incoming gradient is nearly constant over the pixels, dz - k1 is a small
// difference of close numbers and a single-float k1 would cap the accuracy of dy at eps*|k1|.
struct LazyGrad4 {
  const float* dz;
  const float* y;
  float4 s, k1h, k1l, a2, mh, ml;
  bool affine;
};

__device__ __forceinline__ LazyGrad4 lazygrad4_init(const LazyGrad& g, int C, int c) {
  LazyGrad4 r;
  r.dz = g.dz;
  r.affine = g.coef3 != nullptr;
  r.y = r.affine ? g.y : g.dz;   // without coefficients the formula below is the identity on dz (s = 1, a2 = 0)
  r.s = f4(1.f);
  r.k1h = r.k1l = r.a2 = r.mh = r.ml = f4(0.f);
  if (r.affine) {
    r.s = ld4(g.coef3 + c);
    r.k1h = ld4(g.coef3 + C + c);
    r.k1l = ld4(g.coef3 + 2 * C + c);
    r.a2 = ld4(g.coef3 + 3 * C + c);
    r.mh = ld4(g.coef3 + 4 * C + c);
    r.ml = ld4(g.coef3 + 5 * C + c);
  }
  return r;
}

// dy from already loaded dz and y (y is ignored when there are no coefficients)
__device__ __forceinline__ float4 lazygrad4_eval(const LazyGrad4& g, float4 d, float4 y) {
  const float4 t1 = sub4(sub4(d, g.k1h), g.k1l);
  const float4 t2 = sub4(sub4(y, g.mh), g.ml);
  return fma4(g.a2, t2, g.s * t1);
}

__device__ __forceinline__ float4 lazygrad4_at(const LazyGrad4& g, size_t off) {
  return lazygrad4_eval(g, ld4(g.dz + off), ld4(g.y + off));
}

__device__ __forceinline__ float lazygrad1(float dz, float y, float s, float k1h, float k1l, float a2, float mh, float ml) {
  return fmaf(a2, (y - mh) - ml, s * ((dz - k1h) - k1l));
}

// Per-thread state of a BwdEpilogue restricted to 4 consecutive channels of the producer.
struct Epi4 {
  const float* g_other;
  float4 scale, shift, mean, mlo, istd;
  float rlo, rhi;   // the gradient passes where rlo < scale*y + shift < rhi: (0, 6) for ReLU6, (-inf, +inf) otherwise
  bool stats, relu6;
};

__device__ __forceinline__ Epi4 epi4_init(const BwdEpilogue& e, int C, int c) {
  Epi4 r;
  r.g_other = e.g_other;
  r.stats = e.coef != nullptr;
  r.relu6 = e.relu6 != 0;
  r.rlo = (r.stats && r.relu6) ? 0.f : -INFINITY;
  r.rhi = (r.stats && r.relu6) ? 6.f : INFINITY;
  r.scale = f4(1.f);
  r.shift = r.mean = r.mlo = f4(0.f);
  r.istd = f4(1.f);
  if (r.stats) {
    r.scale = ld4(e.coef + c);
    r.shift = ld4(e.coef + C + c);
    r.mean = ld4(e.coef + 2 * C + c);
    r.istd = ld4(e.coef + 3 * C + c);
    r.mlo = ld4(e.coef + 4 * C + c);
  }
  return r;
}

// Applies the epilogue to gradient g of the element whose raw producer output is yraw; accumulates the
// two BatchNorm-backward sums.  Returns the value to store.
// Core of the epilogue on a gradient that already includes g_other.
// Branch-free: without producer coefficients the mask bounds are infinite and the two sums are accumulated from
// identity coefficients (nobody reads them); yraw must be a finite, really loaded value.
__device__ __forceinline__ float4 epi4_core(const Epi4& e, float4 g, float4 yraw, float4& s1, float4& s2) {
  const float4 v = fma4(yraw, e.scale, e.shift);
  g.x = (v.x > e.rlo && v.x < e.rhi) ? g.x : 0.f;
  g.y = (v.y > e.rlo && v.y < e.rhi) ? g.y : 0.f;
  g.z = (v.z > e.rlo && v.z < e.rhi) ? g.z : 0.f;
  g.w = (v.w > e.rlo && v.w < e.rhi) ? g.w : 0.f;
  const float4 xh = sub4(sub4(yraw, e.mean), e.mlo) * e.istd;
  s1 = s1 + g;
  s2 = fma4(g, xh, s2);
  return g;
}

// `ok` = false marks a pixel that does not exist (its g must already be zero): nothing is accumulated for it.
__device__ __forceinline__ float4 epi4_apply(const Epi4& e, float4 g, float4 yraw, size_t off, float4& s1, float4& s2,
                                             bool ok = true) {
  if (e.g_other) {
    const float4 o = ld4(e.g_other + off);
    g = g + make_float4(ok ? o.x : 0.f, ok ? o.y : 0.f, ok ? o.z : 0.f, ok ? o.w : 0.f);
  }
  if (e.stats) {
    if (e.relu6) {
      const float4 v = fma4(yraw, e.scale, e.shift);
      g.x = (v.x > 0.f && v.x < 6.f) ? g.x : 0.f;
      g.y = (v.y > 0.f && v.y < 6.f) ? g.y : 0.f;
      g.z = (v.z > 0.f && v.z < 6.f) ? g.z : 0.f;
      g.w = (v.w > 0.f && v.w < 6.f) ? g.w : 0.f;
    }
    const float4 xh = sub4(sub4(yraw, e.mean), e.mlo) * e.istd;
    s1 = s1 + g;
    s2 = fma4(g, xh, s2);
  }
  return g;
}

// Gather geometry of a 3x3, pad-1 stencil along one axis.
//   GEOM_CONV : source index of tap k for destination o  ->  o*s - 1 + k
//   GEOM_CONVT: source index of tap k for destination o  ->  (o + 1 - k)/s when divisible
// Returns -1 when the tap falls outside [0, n_src) or does not exist.
enum { GEOM_CONV = 0, GEOM_CONVT = 1 };

template <int GEOM>
__device__ __forceinline__ int tap_src(int o, int k, int s, int n_src) {
  int i;
  if (GEOM == GEOM_CONV) {
    i = o * s - 1 + k;
  } else {
    const int t = o + 1 - k;
    if (t < 0) return -1;
    if (s == 2) {
      if (t & 1) return -1;
      i = t >> 1;
    } else {
      i = t;
    }
  }
  return (i >= 0 && i < n_src) ? i : -1;
}
#endif
