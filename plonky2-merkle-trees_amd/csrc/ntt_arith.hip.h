// ntt_arith.hip.h -- Goldilocks arithmetic of the transform kernels (NTT / IFFT / coset LDE), written for the gfx950 issue
// costs (profiles/r01_valu_issue_rates_gfx950.txt: every VALU instruction other than a plain 32-bit add / sub / xor / mov costs
// one ~4.3-cycle issue slot, v_mad_u64_u32, v_lshl_add_u64 and the 64-bit shifts included).  The transform kernels sit on that
// issue cadence (profiles/r04_commit_phase.txt), so everything here is counted in slots:
//
//   add      4   v_lshl_add_u64 + v_cmp_lt_u64 + v_cndmask + v_mad_u64_u32 (+EPS where the 64-bit sum wrapped)
//   sub      5   v_sub_co + v_subb_co + v_cndmask + v_sub_co + v_subbrev_co (-EPS where it borrowed)
//   bfly     9   (a + b, a - b) in one block, the two chains interleaved so that their carries need no padding
//   mul      9 (+3 v_mov)  four v_mad_u64_u32 + the 5-slot 128 -> 64 bit fold
//   mul_pow2<K>  5 (K < 32) / 7 (32 <= K < 96): shifts + the fold; 2^96 = -1, so K >= 96 is a sign the butterflies absorb
//
// Replaces plonky2_field 0.1.0 fft.rs / goldilocks_field.rs arithmetic inside CircuitData::prove (absent third-party source;
// call sites /root/reference/src/mmr/mmr_plonky2_verifier.rs:148, mmr_plonky2_verifier_1_recursion.rs:192,218).  In Goldilocks
// 2 has order 192 and plonky2's 2^k-th roots of unity (g^((p-1)/2^k), g = 7^((p-1)/2^32)) are powers of two up to k = 6:
// w_64 = 2^39, w_16 = 2^156 = -2^60, w_8 = 2^120 = -2^24, w_4 = 2^48 -- so a radix-16 butterfly needs no multiplier at all.
//
// Values are "loose" u64 (any representative, as gl64.hip.h).  add / sub / mul leave the astronomically rare second wrap (both
// operands within 2^32 of 2^64, or a borrow against a value < 2^32) to a sticky lane mask in SGPRs, exactly as
// poseidon_fast.hip.h does: the caller redoes a flagged tile with the exact radix-2 code, so results are exact for every input.
// Each primitive is ONE asm block that keeps its carries and ORs its rare-event mask into the sticky pair itself (round 3's
// per-instruction asm with C-level `sticky |= c` had every mask spilled to VGPR lanes and OR-ed at the end of the kernel:
// 880 v_readlane / v_writelane + 750 s_nop in the 4 800-instruction k_coset_lde12).
#pragma once
#include "poseidon_fast.hip.h"

namespace ntt {

using gl::u32;
using gl::u64;

// Wait states: LLVM's gfx940+ rule (GCNHazardRecognizer, "VALU writes SGPR -> VALU reads that SGPR": 2 wait states) is not applied
// inside an asm string, so every carry / mask consumer below sits at least two instructions (or an s_nop) behind its producer.
// The sticky OR is an s_or_b64 INSIDE the block: a C-level `sticky |= c` lets the compiler re-associate the ORs into one tree at
// the end of the kernel, which keeps every carry mask alive and spills them to VGPR lanes.

// s = a + b, d = a - b (mod p), loose.  Exact unless a corrected result wraps a second time (-> sticky).  9 slots.
GL_DEV void bfly(u64 a, u64 b, u64& s_out, u64& d_out, u64& sticky) {
  u64 s, c, w;
  u32 dl, dh, ma, ms;
  asm("v_sub_co_u32_e64 %[dl], %[w], %[a0], %[b0]\n\t"
      "v_lshl_add_u64 %[s], %[a], 0, %[b]\n\t"
      "v_cmp_lt_u64_e64 %[c], %[s], %[a]\n\t"
      "v_subb_co_u32_e64 %[dh], %[w], %[a1], %[b1], %[w]\n\t"
      "s_nop 0\n\t"
      "v_cndmask_b32_e64 %[ma], 0, -1, %[c]\n\t"
      "v_cndmask_b32_e64 %[ms], 0, -1, %[w]\n\t"
      "v_mad_u64_u32 %[s], %[c], %[ma], 1, %[s]\n\t"
      "v_sub_co_u32_e64 %[dl], %[w], %[dl], %[ms]\n\t"
      "s_or_b64 %[st], %[st], %[c]\n\t"
      "s_nop 0\n\t"
      "v_subbrev_co_u32_e64 %[dh], %[w], 0, %[dh], %[w]\n\t"
      "s_or_b64 %[st], %[st], %[w]"
      : [s] "=&v"(s), [dl] "=&v"(dl), [dh] "=&v"(dh), [ma] "=&v"(ma), [ms] "=&v"(ms), [c] "=&s"(c), [w] "=&s"(w), [st] "+s"(sticky)
      : [a] "v"(a), [b] "v"(b), [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
      : "scc");
  s_out = s;
  d_out = ((u64)dh << 32) | dl;
}

// a + b (mod p), loose.  4 slots.
GL_DEV u64 add(u64 a, u64 b, u64& sticky) {
  u64 s, c;
  u32 m;
  asm("v_lshl_add_u64 %[s], %[a], 0, %[b]\n\t"
      "v_cmp_lt_u64_e64 %[c], %[s], %[a]\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %[m], 0, -1, %[c]\n\t"
      "v_mad_u64_u32 %[s], %[c], %[m], 1, %[s]\n\t"
      "s_or_b64 %[st], %[st], %[c]"
      : [s] "=&v"(s), [c] "=&s"(c), [m] "=&v"(m), [st] "+s"(sticky)
      : [a] "v"(a), [b] "v"(b)
      : "scc");
  return s;
}

// a - b (mod p), loose.  5 slots.
GL_DEV u64 sub(u64 a, u64 b, u64& sticky) {
  u32 lo, hi, m;
  u64 w;
  asm("v_sub_co_u32_e64 %[lo], %[w], %[a0], %[b0]\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32_e64 %[hi], %[w], %[a1], %[b1], %[w]\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %[m], 0, -1, %[w]\n\t"
      "v_sub_co_u32_e64 %[lo], %[w], %[lo], %[m]\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32_e64 %[hi], %[w], 0, %[hi], %[w]\n\t"
      "s_or_b64 %[st], %[st], %[w]"
      : [lo] "=&v"(lo), [hi] "=&v"(hi), [m] "=&v"(m), [w] "=&s"(w), [st] "+s"(sticky)
      : [a0] "v"((u32)a), [a1] "v"((u32)(a >> 32)), [b0] "v"((u32)b), [b1] "v"((u32)(b >> 32))
      : "scc");
  return ((u64)hi << 32) | lo;
}

// lo + h 2^64 (h < 2^32) folded to 64 bits: exact (a wrapped value is < h EPS, so + EPS cannot wrap again).  3 slots.
GL_DEV u64 fold96(u32 h, u64 lo) {
  u64 c;
  u32 m;
  asm("v_mad_u64_u32 %[lo], %[c], %[h], -1, %[lo]\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32_e64 %[m], 0, -1, %[c]\n\t"
      "v_mad_u64_u32 %[lo], %[c], %[m], 1, %[lo]"
      : [lo] "+v"(lo), [c] "=&s"(c), [m] "=&v"(m)
      : [h] "v"(h));
  return lo;
}

// d - h - (cin ? 1 : 0) for a 32-bit h: the "- hh" of the 128 -> 64 bit fold.  A borrow means d < 2^32 (-> sticky).  2 slots.
GL_DEV u64 sub32_flag(u64 d, u32 h, u64& sticky) {
  u32 lo, hi;
  u64 w;
  asm("v_sub_co_u32_e64 %[lo], %[w], %[d0], %[h]\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32_e64 %[hi], %[w], 0, %[d1], %[w]\n\t"
      "s_or_b64 %[st], %[st], %[w]"
      : [lo] "=&v"(lo), [hi] "=&v"(hi), [w] "=&s"(w), [st] "+s"(sticky)
      : [d0] "v"((u32)d), [d1] "v"((u32)(d >> 32)), [h] "v"(h)
      : "scc");
  return ((u64)hi << 32) | lo;
}
GL_DEV u64 sub32_flag_cin(u64 d, u32 h, u64 cin, u64& sticky) {
  u32 lo, hi;
  u64 w;
  asm("v_subb_co_u32_e64 %[lo], %[w], %[d0], %[h], %[cin]\n\t"
      "s_nop 1\n\t"
      "v_subbrev_co_u32_e64 %[hi], %[w], 0, %[d1], %[w]\n\t"
      "s_or_b64 %[st], %[st], %[w]"
      : [lo] "=&v"(lo), [hi] "=&v"(hi), [w] "=&s"(w), [st] "+s"(sticky)
      : [d0] "v"((u32)d), [d1] "v"((u32)(d >> 32)), [h] "v"(h), [cin] "s"(cin)
      : "scc");
  return ((u64)hi << 32) | lo;
}

// a * b (mod p), loose: gl::mul_wide_c (four mads, the 65th bit of the cross terms as a lane mask of weight 2^96 = -1) + the
// fold.  9 slots + 3 v_mov.
GL_DEV u64 mul(u64 a, u64 b, u64& sticky) {
  u64 lo, hi, c;
  gl::mul_wide_c(a, b, lo, hi, c);
  asm volatile("s_nop 0" ::"s"(c));  // (c is one instruction old at most when the fold below reads it two instructions later)
  return sub32_flag_cin(fold96((u32)hi, lo), (u32)(hi >> 32), c, sticky);
}

// x * 2^K (mod p), 0 <= K < 96, loose.
template <int K>
GL_DEV u64 mul_pow2(u64 x, u64& sticky) {
  static_assert(K >= 0 && K < 96, "2^96 = -1: fold the sign into the butterfly");
  if constexpr (K == 0) {
    return x;
  } else if constexpr (K < 32) {
    // x 2^K = lo64 + h 2^64, h < 2^K.  5 slots, exact.
    return fold96((u32)(x >> 32) >> (32 - K), x << K);
  } else if constexpr (K < 64) {
    // x 2^K = lo64 + (hl + hh 2^32) 2^64: the 128 -> 64 bit fold of the general multiply.  7 slots.
    const u64 lo = x << K;
    const u64 hi = x >> (64 - K);
    return sub32_flag(fold96((u32)hi, lo), (u32)(hi >> 32), sticky);
  } else {
    // K = 64 + s: x 2^s = l + Q 2^32 with l = low word, Q = x >> (32 - s) < 2^64, and 2^64 (l + Q 2^32) = l EPS - Q  (2^96 = -1).
    // 7 slots.
    constexpr int s = K - 64;
    const u32 l = (u32)x << s;
    const u64 q = x >> (32 - s);
    u32 nlo, nhi;
    u64 w;
    asm("v_sub_co_u32_e64 %[lo], %[w], 1, %[q0]\n\t"
        "s_nop 1\n\t"
        "v_subb_co_u32_e64 %[hi], %[w], -1, %[q1], %[w]\n\t"
        "s_or_b64 %[st], %[st], %[w]"
        : [lo] "=&v"(nlo), [hi] "=&v"(nhi), [w] "=&s"(w), [st] "+s"(sticky)
        : [q0] "v"((u32)q), [q1] "v"((u32)(q >> 32))
      : "scc");  // p - Q; Q > p (Q within 2^32 of 2^64) -> sticky
    return fold96(l, ((u64)nhi << 32) | nlo);             // l EPS + (p - Q) < 2^64 + p: one wrap at most
  }
}

// In-register DIF over 16 points: natural slots in, bit-reversed slots out (slot r holds frequency brev4(r)), twiddles
// zeta^(e << s) with zeta = 2^Z16 the 16th root of unity: Z16 = 156 forward (plonky2's w_16), 36 = -156 mod 192 inverse.
// A twiddle -2^K is the butterfly with its operands swapped (b - a) times 2^K.
template <int E>
GL_DEV void bfly_pow2(u64& x0, u64& x1, u64& sticky) {
  constexpr int e = ((E % 192) + 192) % 192;
  u64 s, d;
  if constexpr (e >= 96) {
    bfly(x1, x0, s, d, sticky);
    x1 = mul_pow2<e - 96>(d, sticky);
  } else {
    bfly(x0, x1, s, d, sticky);
    x1 = mul_pow2<e>(d, sticky);
  }
  x0 = s;
}
template <int Z16>
GL_DEV void dif16(u64 (&x)[16], u64& sticky) {
  poseidon::static_for<0, 4>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    constexpr int half = 8 >> s;
    poseidon::static_for<0, 8>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      constexpr int blk = b / half, e = b % half;
      constexpr int i0 = blk * 2 * half + e;
      bfly_pow2<Z16 * (e << s)>(x[i0], x[i0 + half], sticky);
    });
  });
}

// the same over 4 points: zeta_4 = 2^Z4 (48 forward, 144 inverse)
template <int Z4>
GL_DEV void dif4(u64 (&x)[4], u64& sticky) {
  bfly_pow2<0>(x[0], x[2], sticky);
  bfly_pow2<Z4>(x[1], x[3], sticky);
  bfly_pow2<0>(x[0], x[1], sticky);
  bfly_pow2<0>(x[2], x[3], sticky);
}

}  // namespace ntt
