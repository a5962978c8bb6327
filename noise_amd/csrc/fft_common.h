// Register-resident small DFTs and complex helpers shared by the in-LDS FFT kernels (gfx950).
#pragma once
#include "common.h"

namespace {

struct c32 {
  float x, y;
};
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 cconj(c32 a) { return {a.x, -a.y}; }
__device__ __forceinline__ c32 cmul_mi(c32 a) { return {a.y, -a.x}; }   // a * (-i)

// ---- small forward DFTs (e^{-i...}) on registers, natural order in and out
template <int R>
struct Dft;

template <>
struct Dft<2> {
  static __device__ __forceinline__ void run(c32 (&v)[2]) {
    const c32 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
  }
};

template <>
struct Dft<4> {
  static __device__ __forceinline__ void run(c32 (&v)[4]) {
    const c32 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]), s1 = cadd(v[1], v[3]), d1 = cmul_mi(csub(v[1], v[3]));
    v[0] = cadd(s0, s1);
    v[1] = cadd(d0, d1);
    v[2] = csub(s0, s1);
    v[3] = csub(d0, d1);
  }
};

template <>
struct Dft<5> {
  static __device__ __forceinline__ void run(c32 (&v)[5]) {
    constexpr float c1 = 0.30901699437494742f, c2 = -0.80901699437494742f;   // cos(2 pi/5), cos(4 pi/5)
    constexpr float s1 = 0.95105651629515357f, s2 = 0.58778525229247313f;    // sin(2 pi/5), sin(4 pi/5)
    const c32 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]), a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
    const c32 x0 = v[0];
    v[0] = {x0.x + a1.x + a2.x, x0.y + a1.y + a2.y};
    const c32 p1 = {x0.x + c1 * a1.x + c2 * a2.x, x0.y + c1 * a1.y + c2 * a2.y};
    const c32 p2 = {x0.x + c2 * a1.x + c1 * a2.x, x0.y + c2 * a1.y + c1 * a2.y};
    // -i (s1 b1 + s2 b2) and -i (s2 b1 - s1 b2)
    const c32 q1 = {s1 * b1.y + s2 * b2.y, -(s1 * b1.x + s2 * b2.x)};
    const c32 q2 = {s2 * b1.y - s1 * b2.y, -(s2 * b1.x - s1 * b2.x)};
    v[1] = cadd(p1, q1);
    v[4] = csub(p1, q1);
    v[2] = cadd(p2, q2);
    v[3] = csub(p2, q2);
  }
};

// e^{-2 pi i k / R} for compile-time k, R (double precision constants folded by the compiler)
template <int R>
__device__ __forceinline__ c32 unit_root(int k) {
  const double a = -6.283185307179586476925286766559 * (double)(k % R) / (double)R;
  return {(float)__builtin_cos(a), (float)__builtin_sin(a)};
}

// R = RA * RB: n = n1 + RA n2, k = RB k1 + k2
template <int RA, int RB>
__device__ __forceinline__ void dft_ct(c32 (&v)[RA * RB]) {
  c32 t[RA][RB];
#pragma unroll
  for (int n1 = 0; n1 < RA; ++n1) {
#pragma unroll
    for (int n2 = 0; n2 < RB; ++n2) t[n1][n2] = v[n1 + RA * n2];
    Dft<RB>::run(t[n1]);
#pragma unroll
    for (int k2 = 1; k2 < RB; ++k2)
      if (n1 != 0) t[n1][k2] = cmul(t[n1][k2], unit_root<RA * RB>(n1 * k2));
  }
#pragma unroll
  for (int k2 = 0; k2 < RB; ++k2) {
    c32 u[RA];
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) u[n1] = t[n1][k2];
    Dft<RA>::run(u);
#pragma unroll
    for (int k1 = 0; k1 < RA; ++k1) v[RB * k1 + k2] = u[k1];
  }
}

template <>
struct Dft<8> {
  static __device__ __forceinline__ void run(c32 (&v)[8]) { dft_ct<2, 4>(v); }
};
template <>
struct Dft<10> {
  static __device__ __forceinline__ void run(c32 (&v)[10]) { dft_ct<2, 5>(v); }
};
template <>
struct Dft<16> {
  static __device__ __forceinline__ void run(c32 (&v)[16]) { dft_ct<4, 4>(v); }
};

template <>
struct Dft<32> {
  static __device__ __forceinline__ void run(c32 (&v)[32]) { dft_ct<4, 8>(v); }
};

// Second pass of a two-pass N2 = R1*R2 transform whose first pass left element (m1, k2) at [m1 R2 + k2]: thread j < R2
// transforms the R1 elements [m1 R2 + j] in place; the result index is R2 k1 + j.
template <int R1, int R2>
__device__ __forceinline__ void fft_pass_b(c32* zrow, int j) {
  if (j < R2) {
    c32 v[R1];
#pragma unroll
    for (int m1 = 0; m1 < R1; ++m1) v[m1] = zrow[m1 * R2 + j];
    Dft<R1>::run(v);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) zrow[R2 * k1 + j] = v[k1];
  }
}

}  // namespace
