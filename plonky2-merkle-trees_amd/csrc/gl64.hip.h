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
// VALU) plus a 7-instruction reduction written in inline assembly (carries as SGPR lane masks); the plain-C
// form (*_ref) stays for the reference variants of the permutation.
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

// a, b: any u64.  Result: loose u64.  Plain C: the compiler schedules it (~20 VALU instructions); the exact REFERENCE variants of
// the permutation (poseidon.hip.h: the redo path of the flag form and the A/B baselines) stay on this form on purpose.
GL_DEV u64 mul_ref(u64 a, u64 b) {
  u64 lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi);
}
GL_DEV u64 mul_add_ref(u64 a, u64 b, u64 c) {
  u64 lo, hi;
  mul_wide(a, b, lo, hi);
  lo += c;
  hi += (lo < c);
  return reduce128(lo, hi);
}
GL_DEV u64 pow7_ref(u64 x) {
  const u64 x2 = mul_ref(x, x);
  const u64 x4 = mul_ref(x2, x2);
  const u64 x3 = mul_ref(x2, x);
  return mul_ref(x4, x3);
}

// ------------------------------------------------------------------ issue-optimised exact forms
// Every VALU instruction other than add / sub / xor / mov costs the same issue slot on gfx950, v_mad_u64_u32 (32 x 32 + 64) included,
// so these are written for the fewest instructions: carries travel as SGPR lane masks, "+ x" of a 32-bit word into a 64-bit pair is an
// x*1 mad, and 2^64 == EPS folds as a multiply by the inline constant -1.
// d = a * b + c, carry-out as a lane mask (SGPR pair)
GL_DEV u64 mad_carry(u32 a, u32 b, u64 c, u64& carry) {
  u64 d;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return d;
}
// d = a * 0xFFFFFFFF + c   (a * 2^64 folded: 2^64 = 2^32 - 1 mod p), carry-out as a lane mask
GL_DEV u64 mad_eps_carry(u32 a, u64 c, u64& carry) {
  u64 d;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(d), "=s"(carry) : "v"(a), "v"(c));
  return d;
}
// d = a + c (32-bit a into a 64-bit pair) as one mad; caller guarantees no overflow
GL_DEV u64 add32(u32 a, u64 c) {
  u64 d, unused;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(unused) : "v"(a), "v"(c));
  return d;
}
// mask ? 0xFFFFFFFF : 0
GL_DEV u32 eps_if(u64 mask) {
  u32 m;
  asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(m) : "s"(mask));
  return m;
}
// (hi:lo) - h, borrow-out as a lane mask
GL_DEV u64 sub32_borrow(u64 x, u32 h, u64& borrow) {
  u32 lo, hi;
  asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\tv_subbrev_co_u32_e64 %1, %2, 0, %4, %2"
      : "=&v"(lo), "=v"(hi), "=&s"(borrow)
      : "v"((u32)x), "v"((u32)(x >> 32)), "v"(h));
  return ((u64)hi << 32) | lo;
}
// (hi:lo) - h - (cin ? 1 : 0), borrow-out as a lane mask
GL_DEV u64 sub32_borrow_in(u64 x, u32 h, u64 cin, u64& borrow) {
  u32 lo, hi;
  asm("v_subb_co_u32_e64 %0, %2, %3, %5, %6\n\tv_subbrev_co_u32_e64 %1, %2, 0, %4, %2"
      : "=&v"(lo), "=v"(hi), "=&s"(borrow)
      : "v"((u32)x), "v"((u32)(x >> 32)), "v"(h), "s"(cin));
  return ((u64)hi << 32) | lo;
}
// 64 x 64 -> 128 in four mads and nothing else: the second cross term is added to the WHOLE first one (a1 b0 + t1, 65 bits), its
// carry-out stays a lane mask `c` of weight 2^96 == -1 (mod p), and the reduction takes it as the borrow-in of its "- hh".  (gfx90a+
// wants 64-bit operands in even-aligned pairs, so every (word, 0) addend costs a v_mov; this form needs three.)
// a b = lo + (hi + c 2^32) 2^64.
GL_DEV void mul_wide_c(u64 a, u64 b, u64& lo, u64& hi, u64& c) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 t0 = (u64)a0 * b0;
  const u64 t1 = (u64)a0 * b1 + (t0 >> 32);        // <= 2^64 - 2^32: no carry
  const u64 t2 = mad_carry(a1, b0, t1, c);
  hi = (u64)a1 * b1 + (t2 >> 32);                  // < 2^64
  lo = (t2 << 32) | (u32)t0;
}
// x = lo + hl 2^64 + (hh + c) 2^96 == lo + hl EPS - hh - c, exactly (hh + c < 2^32).  Result: loose u64.
GL_DEV u64 reduce128_c(u64 lo, u64 hi, u64 c) {
  const u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 c1, b, b2;
  const u64 d1 = mad_eps_carry(hl, lo, c1);                // wrapped by 2^64 in lanes of c1
  const u64 d2 = add32(eps_if(c1), d1);                    // + EPS there; cannot wrap again (d1 <= 2^64 - 2^33 when wrapped)
  const u64 d3 = sub32_borrow_in(d2, hh, c, b);            // wrapped by +2^64 == +EPS in lanes of b ...
  return sub32_borrow(d3, eps_if(b), b2);                  // ... take it back (d3 >= 2^64 - 2^32 there: no 2nd borrow)
}
// a, b: any u64.  Result: loose u64.  11 issue slots + 3 v_mov.  (Compile-time constant operands keep the C form: it folds.)
GL_DEV u64 mul(u64 a, u64 b) {
  if (__builtin_constant_p(a) || __builtin_constant_p(b)) return mul_ref(a, b);
  u64 lo, hi, c;
  mul_wide_c(a, b, lo, hi, c);
  return reduce128_c(lo, hi, c);
}
// a*b + c, all any u64 (product + c < 2^128).  Result: loose u64.  c rides in as addends of the first two partial products.
GL_DEV u64 mul_add(u64 a, u64 b, u64 c) {
  if (__builtin_constant_p(a) || __builtin_constant_p(b)) return mul_add_ref(a, b, c);
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 cy;
  const u64 t0 = (u64)a0 * b0 + (u32)c;                    // <= 2^64 - 2^32
  u64 t1 = (u64)a0 * b1 + (t0 >> 32);                      // <= 2^64 - 2^32
  t1 = add32((u32)(c >> 32), t1);                          // <= 2^64 - 1
  const u64 t2 = mad_carry(a1, b0, t1, cy);
  const u64 hi = (u64)a1 * b1 + (t2 >> 32);
  const u64 lo = (t2 << 32) | (u32)t0;
  return reduce128_c(lo, hi, cy);
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
