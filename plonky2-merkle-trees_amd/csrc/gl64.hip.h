// gl64.hip.h -- Goldilocks field (p = 2^64 - 2^32 + 1) on the gfx950 32-bit VALU.
//
// Replaces plonky2_field 0.1.0 goldilocks_field.rs (third-party, absent from the reference tree;
// field order pinned by /root/reference/src/mmr/common.rs:3).
//
// Representation: a field element is any u64 congruent to it ("loose" form, like plonky2's
// non-canonical GoldilocksField); canon() maps to [0,p).  Every routine states what it accepts.
// Identities used: 2^64 = 2^32 - 1 =: EPS (mod p), 2^96 = -1 (mod p).
//
// The multiply is 4 x v_mad_u64_u32 (32x32+64 -> 64, the only full-product instruction on the
// VALU) plus a 13-instruction reduction; nothing here is GEMM-shaped, so no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GL_DEV __device__ __forceinline__

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ull;
constexpr u64 EPS = 0xFFFFFFFFull;

GL_DEV u64 canon(u64 x) { return x >= P ? x - P : x; }

// a: any u64, c: canonical (< p).  Result: loose u64.   (a + c < 2^64 + p => one fix-up suffices)
GL_DEV u64 add_c(u64 a, u64 c) {
  u64 s = a + c;
  return s < a ? s + EPS : s;
}

// a, b: any u64.  Result: loose u64.
GL_DEV u64 add(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) {
    s += EPS;
    if (s < EPS) s += EPS;  // only when both inputs were within 2^32 of 2^64
  }
  return s;
}

// a: any u64, b: canonical (< p).  Result: loose u64.
GL_DEV u64 sub_c(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d - EPS : d;  // wrapped by +2^64 == +EPS; d >= 2^32 here so no second borrow
}

// 64x64 -> 128 schoolbook on 32-bit halves; each line is one v_mad_u64_u32 (no overflow possible:
// (2^32-1)^2 + 2*(2^32-1) = 2^64 - 1).
GL_DEV void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 t0 = (u64)a0 * b0;
  const u64 t1 = (u64)a0 * b1 + (t0 >> 32);
  const u64 t2 = (u64)a1 * b0 + (u32)t1;
  const u64 t3 = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
  lo = (t2 << 32) | (u32)t0;
  hi = t3;
}

// x = lo + hl*2^64 + hh*2^96  ==  lo - hh + hl*EPS (mod p).  Result: loose u64.
GL_DEV u64 reduce128(u64 lo, u64 hi) {
  const u64 hh = hi >> 32, hl = hi & EPS;
  u64 t0 = lo - hh;
  if (lo < hh) t0 -= EPS;
  const u64 t1 = (hl << 32) - hl;  // hl * EPS, < 2^64
  u64 t2 = t0 + t1;
  if (t2 < t1) t2 += EPS;
  return t2;
}

// x = lo + hi*2^64 with hi < 2^32.  Result: loose u64.
GL_DEV u64 reduce96(u64 lo, u32 hi) {
  const u64 t1 = ((u64)hi << 32) - hi;
  u64 t2 = lo + t1;
  if (t2 < t1) t2 += EPS;
  return t2;
}

// a, b: any u64.  Result: loose u64.
GL_DEV u64 mul(u64 a, u64 b) {
  u64 lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
}

// a*b + c, all any u64 (product + c < 2^128).  Result: loose u64.
GL_DEV u64 mul_add(u64 a, u64 b, u64 c) {
  u64 lo, hi;
  mul_wide(a, b, lo, hi);
  lo += c;
  hi += (lo < c);
  return reduce128(lo, hi);
}

GL_DEV u64 sqr(u64 a) { return mul(a, a); }

// x^7: two squarings + two multiplies (x2, x4, x3 = x2*x, x7 = x4*x3).
GL_DEV u64 pow7(u64 x) {
  const u64 x2 = sqr(x);
  const u64 x4 = sqr(x2);
  const u64 x3 = mul(x2, x);
  return mul(x4, x3);
}

GL_DEV u64 pow(u64 a, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, a);
    a = mul(a, a);
    e >>= 1;
  }
  return r;
}

}  // namespace gl
