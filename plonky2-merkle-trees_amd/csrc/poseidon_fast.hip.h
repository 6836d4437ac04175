// poseidon_fast.hip.h -- issue-optimised Poseidon permutation for gfx950 (the shipped hot path).
//
// Same function as poseidon.hip.h (bit-identical; tests/test_parity_gpu.py), restructured around the MEASURED
// gfx950 issue costs (profiles/r01_valu_issue_rates_gfx950.txt): add/sub/xor/mov issue at ~4 wave-instr/CU/ns,
// every other VALU op -- including v_mad_u64_u32 (32x32+64) -- at ~2.  The kernel is issue-bound, so the design
// goal is simply the fewest issue units per hash:
//   * MDS row = two 12-long v_mad_u64_u32 chains (32-bit halves x 6-bit constants).  The NEXT round's constant
//     is folded in as the chains' initial addend (halves of the constant sit in SGPR pairs), so the per-round
//     "add constants" layer disappears.
//   * hi-chain += lo-chain >> 32 and the 96 -> 64 bit fold are two more mads:  x*1 + acc  and
//     top*0xFFFFFFFF + (mid:lo)  (2^64 = 2^32 - 1 mod p), whose carry-out lands in an SGPR lane mask.
//   * 128 -> 64 bit reduction after a field multiply: hl*0xFFFFFFFF + lo64 in one mad (carry mask c1),
//     + (c1 ? EPS : 0) through v_cndmask + one x*1 mad, then - hh with v_sub_co/v_subbrev.
//   * Events that need a further correction are RARE (probability <= 2^-22 per op: the 96->64 fold overflowing,
//     the "- hh" borrowing).  Instead of paying fix-up instructions on every op, their lane masks are OR-ed
//     (scalar pipe, s_or_b64) into a wave-uniform sticky flag; a wave whose flag is set (~0.5 % of waves)
//     reloads its inputs and recomputes with the exact reference variant.  Results are therefore exact always.
#pragma once
#include "poseidon.hip.h"

namespace poseidon_fast {

using gl::u32;
using gl::u64;

// The constant table seen through the CONSTANT address space: a uniform load through such a pointer is a scalar load whatever else
// the kernel does.  Through the plain (generic) pointer the compiler must prove that nothing in the kernel can write the table, and
// it cannot when the kernel also stores through computed pointers or uses atomics (the batched proof-of-work queue read its round
// constants and the 168 matrix words of every group of partial rounds with VECTOR loads: ~400 KB per wave-permutation).  Valid for the
// global table only (written once at p2mt_init) -- never for the LDS copy the 12-lane layout keeps.
typedef const u64 __attribute__((address_space(4))) * ctab;
typedef const u32 __attribute__((address_space(4))) * ctab32;
GL_DEV ctab as_const_table(const u64* global_table) { return (ctab)(unsigned long long)global_table; }
GL_DEV const u32* as_u32(const u64* p) { return reinterpret_cast<const u32*>(p); }
GL_DEV ctab32 as_u32(ctab p) { return (ctab32)p; }

// sticky |= mask on the scalar pipe, as an opaque statement: a plain C "|=" lets the optimiser re-associate the ORs of a loop body into
// a tree at its end, which keeps every lane mask alive until then -- in SGPRs spilled to VGPR lanes (44 v_writelane / v_readlane per group
// of four partial rounds, measured)
GL_DEV void raise(u64& sticky, u64 mask) { asm("s_or_b64 %0, %0, %1" : "+s"(sticky) : "s"(mask) : "scc"); }

using gl::add32;
using gl::eps_if;
using gl::mad_carry;
using gl::mad_eps_carry;
using gl::mul_wide_c;
using gl::sub32_borrow;
using gl::sub32_borrow_in;
// acc += a * K, K an inline constant (all MDS entries are <= 41 < 64); forced mad: the compiler would otherwise
// strength-reduce small constants into v_mov + v_lshl_add_u64 (3 issue units instead of 2)
template <u32 K>
GL_DEV void mac_const(u64& acc, u32 a) {
  u64 unused;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(unused) : "v"(a), "n"(K));
}
// first link of a chain: d = a * K + init, init a wave-uniform 64-bit value (SGPR pair: no VGPR initialisation)
template <u32 K>
GL_DEV u64 mac_const_first(u32 a, u64 init_uniform) {
  u64 d, unused;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(unused) : "v"(a), "n"(K), "s"(init_uniform));
  return d;
}
template <u32 K>
GL_DEV u64 mac_const_first0(u32 a) {
  u64 d, unused;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(unused) : "v"(a), "n"(K));
  return d;
}
// x = lo + hl*2^64 + hh*2^96 == lo + hl*EPS - hh.  Loose result; the rare borrow of "- hh" goes to `sticky`.
GL_DEV u64 reduce128(u64 lo, u64 hi, u64& sticky) {
  const u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 c1, b;
  const u64 d1 = mad_eps_carry(hl, lo, c1);     // wrapped by 2^64 in lanes of c1
  const u64 d2 = add32(eps_if(c1), d1);         // + EPS there; cannot wrap again (d1 < 2^64 - 2^33 when wrapped)
  const u64 d3 = sub32_borrow(d2, hh, b);       // borrows only if d2 < hh < 2^32
  raise(sticky, b);
  return d3;
}

// 64 x 64 -> 128: four mads; the last high-word accumulation is an x*1 mad (2 units) instead of the
// v_mov + v_lshl_add_u64 (3 units) the compiler would pick for  t3 + (t2 >> 32).
GL_DEV void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 t0 = (u64)a0 * b0;
  const u64 t1 = (u64)a0 * b1 + (t0 >> 32);
  const u64 t2 = (u64)a1 * b0 + (u32)t1;
  const u64 t3 = add32((u32)(t2 >> 32), (u64)a1 * b1 + (t1 >> 32));
  lo = (t2 << 32) | (u32)t0;
  hi = t3;
}
// x = lo + hl 2^64 + (hh + c) 2^96 == lo + hl EPS - hh - c.  Loose result; the rare borrow goes to `sticky`.
GL_DEV u64 reduce128_c(u64 lo, u64 hi, u64 c, u64& sticky) {
  const u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 c1, b;
  const u64 d1 = mad_eps_carry(hl, lo, c1);
  const u64 d2 = add32(eps_if(c1), d1);
  const u64 d3 = sub32_borrow_in(d2, hh, c, b);    // hh + c < 2^32 (the product is < 2^128): borrows only if d2 < 2^32
  raise(sticky, b);
  return d3;
}

GL_DEV u64 mul(u64 a, u64 b, u64& sticky) {
  u64 lo, hi, c;
  mul_wide_c(a, b, lo, hi, c);
  return reduce128_c(lo, hi, c, sticky);
}

// the previous form (x*1 mad for the last high-word accumulation), kept for the A/B (p2mt_set_variant(2, 6))
GL_DEV u64 mul_v0(u64 a, u64 b, u64& sticky) {
  u64 lo, hi;
  mul_wide(a, b, lo, hi);
  return reduce128(lo, hi, sticky);
}
template <int MULV = 0>
GL_DEV u64 pow7(u64 x, u64& sticky) {
  if constexpr (MULV == 1) {
    const u64 x2 = mul_v0(x, x, sticky);
    const u64 x4 = mul_v0(x2, x2, sticky);
    const u64 x3 = mul_v0(x2, x, sticky);
    return mul_v0(x4, x3, sticky);
  } else {
    const u64 x2 = mul(x, x, sticky);
    const u64 x4 = mul(x2, x2, sticky);
    const u64 x3 = mul(x2, x, sticky);
    return mul(x4, x3, sticky);
  }
}

// ------------------------------------------------------------------ exact forms (no sticky flag)
// Same instruction sequences with the rare corrections applied explicitly (+3 instructions per multiply, +2 per MDS
// row).  Used by the latency-bound layouts (4 and 12 lanes per hash), where a flagged wave redoing its work
// serially would set the duration of the whole launch; the throughput-bound one-hash-per-lane kernels keep the
// flag + fallback, which is cheaper on average.
namespace exact {

GL_DEV u64 reduce128(u64 lo, u64 hi) {
  const u32 hl = (u32)hi, hh = (u32)(hi >> 32);
  u64 c1, b, b2;
  const u64 d1 = mad_eps_carry(hl, lo, c1);
  const u64 d2 = add32(eps_if(c1), d1);
  const u64 d3 = sub32_borrow(d2, hh, b);                  // wrapped by +2^64 == +EPS in lanes of b ...
  return sub32_borrow(d3, eps_if(b), b2);                  // ... take it back (d3 >= 2^64 - 2^32 there: no 2nd borrow)
}
GL_DEV u64 mul(u64 a, u64 b) {  // gl::mul without the constant-folding detour
  u64 lo, hi, c;
  mul_wide_c(a, b, lo, hi, c);
  return gl::reduce128_c(lo, hi, c);
}
GL_DEV u64 pow7(u64 x) {
  const u64 x2 = mul(x, x);
  const u64 x4 = mul(x2, x2);
  const u64 x3 = mul(x2, x);
  return mul(x4, x3);
}
// top * 2^64 + val  (top < 2^10) folded to 64 bits
GL_DEV u64 fold96(u32 top, u64 val) {
  u64 cm;
  const u64 d = mad_eps_carry(top, val, cm);
  return add32(eps_if(cm), d);                             // wrapped value < top * EPS < 2^42: + EPS cannot wrap
}

}  // namespace exact

// out[r] = sum_c MDS[r][c] * s[c] + add[r] for r < ROWS  (add = next round's constants, canonical; ADD = false
// for the last round).  Rows >= ROWS are left untouched (two_to_one only needs 4 output words of the last layer).
template <bool ADD, int ROWS = 12, bool EXACT = false, typename P = const u64*>
GL_DEV void mds_layer(u64 (&s)[12], P add, u64& sticky) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  poseidon::static_for<0, ROWS>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    constexpr u32 k0 = poseidon::mds_entry(r, 0);
    u64 al, ah;
    if constexpr (ADD) {
      const u64 c = add[r];  // (c_lo, 0) and (c_hi, 0) become SGPR pairs: scalar pipe only
      al = mac_const_first<k0>(lo[0], (u64)(u32)c);
      ah = mac_const_first<k0>(hi[0], (u64)(u32)(c >> 32));
    } else {
      al = mac_const_first0<k0>(lo[0]);
      ah = mac_const_first0<k0>(hi[0]);
    }
    poseidon::static_for<1, 12>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr u32 k = poseidon::mds_entry(r, c);
      mac_const<k>(al, lo[c]);   // < 264 * 2^32 + 2^32
      mac_const<k>(ah, hi[c]);
    });
    ah = add32((u32)(al >> 32), ah);                       // X = ah * 2^32 + (u32)al, < 2^74
    const u64 val = ((u64)(u32)ah << 32) | (u32)al;
    if constexpr (EXACT) {
      s[r] = exact::fold96((u32)(ah >> 32), val);
    } else {
      u64 cm;
      s[r] = mad_eps_carry((u32)(ah >> 32), val, cm);      // top * EPS + val; wraps with probability ~2^-22
      raise(sticky, cm);
    }
  });
}

// one row of the last layer (no constants): the proof-of-work grind looks at word 7 only
template <int ROW>
GL_DEV u64 mds_row(const u64 (&s)[12], u64& sticky) {
  constexpr u32 k0 = poseidon::mds_entry(ROW, 0);
  u64 al = mac_const_first0<k0>((u32)s[0]), ah = mac_const_first0<k0>((u32)(s[0] >> 32));
  poseidon::static_for<1, 12>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr u32 k = poseidon::mds_entry(ROW, c);
    mac_const<k>(al, (u32)s[c]);
    mac_const<k>(ah, (u32)(s[c] >> 32));
  });
  ah = add32((u32)(al >> 32), ah);
  u64 cm;
  const u64 r = mad_eps_carry((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al, cm);
  raise(sticky, cm);
  return r;
}

// ------------------------------------------------------------------ MDS layer on the matrix pipe (round 3 A/B)
// The MDS is a 12 x 12 contraction with 6-bit constants.  On 8-bit limbs of the state words it fits
// v_mfma_i32_4x4x4_16b_i8, whose 16 blocks are independent 4x4x4 products over the lanes 4b..4b+3: with one hash per lane the B
// operand of lane n is four bytes of hash n (byte t of words j0..j0+3), the A operand is a per-lane constant (MDS[r0 + lane%4][j0..j0+3])
// and the four result registers of lane n are rows r0..r0+3 of ITS OWN hash -- no cross-lane movement (layout confirmed by
// tools/ubench_mfma.hip).  Per layer: 24 v_xor (bytes -> signed), 48 v_perm_b32 (six 4x4 byte transposes), 72 MFMAs
// (8 limbs x 3 row blocks x 3 K steps; the accumulators start at 128 * rowsum, which undoes the signed-byte offset), 96
// v_mad_u64_u32 putting the 18-bit limb sums back together (the next round's constant rides in as the chain's initial addend), then
// the same 96 -> 64 bit fold as the VALU form.
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
struct MfmaCtx {
  u32 a[3][3];       // a[rb][jb]: bytes k = 0..3 = MDS[4 rb + (lane & 3)][4 jb + k]
  mfma_v4i cinit[2]; // 128 * rowsum of rows 0..3 / of any other four rows
  mfma_v4i a32;      // mds_layer_mfma32: the lane's 16 K slots of its row of the block-structured 32 x 32 A operand (below)
};
constexpr u32 mds_rowsum(int r) {
  u32 t = 0;
  for (int c = 0; c < 12; ++c) t += poseidon::mds_entry(r, c);
  return t;
}
GL_DEV MfmaCtx mfma_ctx_init() {
  MfmaCtx c;
  const unsigned i = threadIdx.x & 3;
  poseidon::static_for<0, 3>([&](auto rbc) {
    constexpr int rb = decltype(rbc)::value;
    poseidon::static_for<0, 3>([&](auto jbc) {
      constexpr int jb = decltype(jbc)::value;
      auto pack = [](int row) constexpr -> u32 {
        return poseidon::mds_entry(row, 4 * jb) | (poseidon::mds_entry(row, 4 * jb + 1) << 8) |
               (poseidon::mds_entry(row, 4 * jb + 2) << 16) | (poseidon::mds_entry(row, 4 * jb + 3) << 24);
      };
      constexpr u32 p0 = pack(4 * rb), p1 = pack(4 * rb + 1), p2 = pack(4 * rb + 2), p3 = pack(4 * rb + 3);
      c.a[rb][jb] = i == 0 ? p0 : (i == 1 ? p1 : (i == 2 ? p2 : p3));
    });
  });
  static_assert(mds_rowsum(1) == 256 && mds_rowsum(2) == 256 && mds_rowsum(3) == 256 && mds_rowsum(11) == 256, "circulant rows");
  c.cinit[0] = mfma_v4i{(int)(128 * mds_rowsum(0)), 128 * 256, 128 * 256, 128 * 256};
  c.cinit[1] = mfma_v4i{128 * 256, 128 * 256, 128 * 256, 128 * 256};
  return c;
}
// acc += a * k, k wave-uniform (SGPR)
GL_DEV void mac_sgpr(u64& acc, u32 a, u32 k) {
  u64 unused;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(unused) : "v"(a), "s"(k));
}
// first link: d = a * 1 + init, init a wave-uniform 64-bit value (SGPR pair)
GL_DEV u64 mac_one_first(u32 a, u64 init_uniform) {
  u64 d, unused;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(unused) : "v"(a), "s"(init_uniform));
  return d;
}
GL_DEV u64 mac_one_first0(u32 a) {
  u64 d, unused;
  asm("v_mad_u64_u32 %0, %1, %2, 1, 0" : "=v"(d), "=s"(unused) : "v"(a));
  return d;
}
template <bool ADD, int ROWS = 12, typename P = const u64*>
GL_DEV void mds_layer_mfma(u64 (&s)[12], P add, u64& sticky, const MfmaCtx& mc) {
  // B operands: bt[jb][t] = (byte t of words 4jb .. 4jb+3) ^ 0x80, t = 0..7 over the 64-bit word
  u32 bt[3][8];
#pragma unroll
  for (int jb = 0; jb < 3; ++jb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32 w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (u32)(s[4 * jb + k] >> (32 * h)) ^ 0x80808080u;
      const u32 p01l = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), p01h = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
      const u32 p23l = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), p23h = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
      bt[jb][4 * h + 0] = __builtin_amdgcn_perm(p23l, p01l, 0x05040100u);
      bt[jb][4 * h + 1] = __builtin_amdgcn_perm(p23l, p01l, 0x07060302u);
      bt[jb][4 * h + 2] = __builtin_amdgcn_perm(p23h, p01h, 0x05040100u);
      bt[jb][4 * h + 3] = __builtin_amdgcn_perm(p23h, p01h, 0x07060302u);
    }
  }
  poseidon::static_for<0, (ROWS + 3) / 4>([&](auto rbc) {
    constexpr int rb = decltype(rbc)::value;
    // eight independent accumulation chains, K step by K step: no MFMA waits for the one issued just before it
    mfma_v4i d[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) d[t] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)mc.a[rb][0], (int)bt[0][t], mc.cinit[rb ? 1 : 0], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) d[t] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)mc.a[rb][1], (int)bt[1][t], d[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 8; ++t) d[t] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)mc.a[rb][2], (int)bt[2][t], d[t], 0, 0, 0);
    // The results are consumed by inline-asm mads, which the compiler's hazard recogniser does not look into: a VALU read of an
    // XDL result needs passes + 3 = 5 wait states on gfx950 (software-managed).  This statement depends on all eight
    // accumulators, so it sits behind the last MFMA and in front of every consumer.
    asm volatile("s_nop 7" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
    poseidon::static_for<0, 4>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int r = 4 * rb + i;
      if constexpr (r < ROWS) {
        u64 al, ah;  // sum_t d_t 2^(8t) over the low / high half, every d_t in [0, 2^18)
        if constexpr (ADD) {
          const u64 c = add[r];
          al = mac_one_first((u32)d[0][i], (u64)(u32)c);
          ah = mac_one_first((u32)d[4][i], (u64)(u32)(c >> 32));
        } else {
          al = mac_one_first0((u32)d[0][i]);
          ah = mac_one_first0((u32)d[4][i]);
        }
        mac_sgpr(al, (u32)d[1][i], 1u << 8);
        mac_sgpr(ah, (u32)d[5][i], 1u << 8);
        mac_sgpr(al, (u32)d[2][i], 1u << 16);
        mac_sgpr(ah, (u32)d[6][i], 1u << 16);
        mac_sgpr(al, (u32)d[3][i], 1u << 24);
        mac_sgpr(ah, (u32)d[7][i], 1u << 24);
        ah = add32((u32)(al >> 32), ah);
        const u64 val = ((u64)(u32)ah << 32) | (u32)al;
        u64 cm;
        s[r] = mad_eps_carry((u32)(ah >> 32), val, cm);
        raise(sticky, cm);
      }
    });
  });
}

// ------------------------------------------------------------------ MDS layer on ONE large MFMA per 8-bit limb (round 3, second A/B)
// The 4x4x4 form above is a wash because the matrix pipe's cost to the VALU is per INSTRUCTION (72 per layer x ~1.1 mad slots).
// v_mfma_i32_32x32x32_i8 costs ~4 mad slots (tools/ubench_mfma32.hip: 8.4-11 ns of VALU time at four waves per SIMD) and, with a
// block-structured A operand, does a whole limb of the layer for all 64 hashes of the wave without cross-lane movement:
//   * lanes 0..31 feed K slots 0..15 of column n = lane, lanes 32..63 feed K slots 16..31 of column n = lane - 32;
//   * result register v of lane (n, half) is row 8 (v / 4) + 4 half + v % 4 of column n;
//   * so rows {0-3, 8-11, 16-19} of A carry M in K slots 0..11 only and rows {4-7, 12-15, 20-23} carry M in K slots 16..27 only:
//     register v < 12 of EVERY lane = sum_j M[v][j] * (byte l of word j of the lane's own hash)   (probe: tools/ubench_mfma32.hip).
// K slots 12..15 carry the signed-byte offset: the byte -128 in B against -rowsum / 4 in A adds 128 rowsum, so the results are the
// unsigned limb sums.  Per layer: 24 v_xor (bytes -> signed) + 48 v_perm_b32 (six 4x4 byte transposes, as above) + 8 MFMAs (srcC =
// inline 0) + per row two chains of four v_mad_u64_u32 (18-bit limb sums x 2^(8l), starting at the next round's constant half) + the
// same 96 -> 64 bit fold: 120 mads against 312 in the VALU form.
typedef int mfma_v16i __attribute__((ext_vector_type(16)));
// Call at kernel entry, with EVERY lane of the wave active (1-D workgroups of whole 64-lane wavefronts): the operand is a function of the
// lane id, and an MFMA reads it from all 64 lanes whatever EXEC says later.
GL_DEV void mfma32_ctx_init(MfmaCtx& c) {
  const unsigned l = threadIdx.x & 63, i = l & 31, half = l >> 5, sub = (i >> 2) & 1, v = 4 * (i >> 3) + (i & 3);
  mfma_v4i a = {0, 0, 0, 0};
  poseidon::static_for<0, 12>([&](auto rc_) {
    constexpr int r = decltype(rc_)::value;
    auto pack = [](int jb) constexpr -> u32 {
      return poseidon::mds_entry(r, 4 * jb) | (poseidon::mds_entry(r, 4 * jb + 1) << 8) | (poseidon::mds_entry(r, 4 * jb + 2) << 16) |
             (poseidon::mds_entry(r, 4 * jb + 3) << 24);
    };
    constexpr u32 p0 = pack(0), p1 = pack(1), p2 = pack(2);
    // K slots 12..15 meet the constant byte -128 in B: 4 x (-rowsum / 4) x (-128) = 128 rowsum undoes the signed-byte offset
    static_assert(mds_rowsum(r) % 4 == 0 && mds_rowsum(r) / 4 <= 128, "offset term fits four signed bytes");
    constexpr u32 p3 = 0x01010101u * (u32)(256 - mds_rowsum(r) / 4);
    if (sub == half && v == (unsigned)r) a = mfma_v4i{(int)p0, (int)p1, (int)p2, (int)p3};
  });
  c.a32 = a;
}
// XF: the 96 -> 64 bit fold of a row in its exact form (two more instructions per row, no flag: see permute_impl)
template <bool ADD, int ROWS = 12, typename P = const u64*, bool XF = false>
GL_DEV void mds_layer_mfma32(u64 (&s)[12], P add, u64& sticky, const MfmaCtx& mc) {
  u64 acc[2][12];
  poseidon::static_for<0, 2>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    u32 t[3][4];  // t[g][b] = (byte b of the halves of words 4g .. 4g+3) ^ 0x80
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      u32 w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (u32)(s[4 * g + k] >> (32 * h)) ^ 0x80808080u;
      const u32 p01l = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), p01h = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
      const u32 p23l = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), p23h = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
      t[g][0] = __builtin_amdgcn_perm(p23l, p01l, 0x05040100u);
      t[g][1] = __builtin_amdgcn_perm(p23l, p01l, 0x07060302u);
      t[g][2] = __builtin_amdgcn_perm(p23h, p01h, 0x05040100u);
      t[g][3] = __builtin_amdgcn_perm(p23h, p01h, 0x07060302u);
    }
    mfma_v16i c[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      mfma_v4i bop;  // K slots 12..15: the byte -128 against -rowsum / 4 in A, i.e. + 128 rowsum: the results are the UNSIGNED limb sums
      bop[0] = (int)t[0][b], bop[1] = (int)t[1][b], bop[2] = (int)t[2][b], bop[3] = (int)0x80808080u;
      c[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(mc.a32, bop, mfma_v16i{}, 0, 0, 0);
    }
    // The results are consumed by inline-asm mads, which the compiler's hazard recogniser does not look into: a VALU read of the
    // result of an 8-pass XDL op needs 11 wait states.  This statement depends on all four results, so it sits behind the last MFMA
    // and in front of every consumer.
    asm volatile("s_nop 15" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
    poseidon::static_for<0, ROWS>([&](auto rcst) {
      constexpr int r = decltype(rcst)::value;
      u64 a;  // sum_l c_l 2^(8l) over this half, every c_l in [0, 2^18)
      if constexpr (ADD) a = mac_one_first((u32)c[0][r], (u64)(u32)(add[r] >> (32 * h)));
      else a = mac_one_first0((u32)c[0][r]);
      mac_sgpr(a, (u32)c[1][r], 1u << 8);
      mac_sgpr(a, (u32)c[2][r], 1u << 16);
      mac_sgpr(a, (u32)c[3][r], 1u << 24);
      acc[h][r] = a;  // < 2^43
    });
  });
  poseidon::static_for<0, ROWS>([&](auto rcst) {
    constexpr int r = decltype(rcst)::value;
    const u64 al = acc[0][r];
    const u64 ah = add32((u32)(al >> 32), acc[1][r]);
    const u64 val = ((u64)(u32)ah << 32) | (u32)al;
    if constexpr (XF) {
      s[r] = exact::fold96((u32)(ah >> 32), val);
    } else {
      u64 cm;
      s[r] = mad_eps_carry((u32)(ah >> 32), val, cm);
      raise(sticky, cm);
    }
  });
}

// ------------------------------------------------------------------ sparse partial rounds (SPARSE = true)
// plonky2's "fast" partial rounds (poseidon.rs partial_first_constant_layer / mds_partial_layer_init / mds_partial_layer_fast):
//   s += FIRST;  s[1..] = INIT * s[1..]  (11 x 11, once);  22 x { s0 = sbox(s[0]) + K_r;  d = 25 s0 + sum_j W_rj s[j];
//                                                               s[j] += V_rj s0;  s[0] = d }
// with full 64-bit constants.  A dense round costs 26 v_mad_u64_u32 per MDS row because the MDS entries are 6-bit numbers; here a
// term of a dot product costs 6: the constant is split into limbs of 22 + 22 + 20 bits and the state word into its two dwords,
// so that all six 32 x 22-bit partial products of a term accumulate in 64-bit registers without ever overflowing (12 terms x
// 2^54 < 2^58) -- no carry handling per product.  The six sums are put together and reduced once per dot product.
// Table layout behind the round constants (u64 indices from ctx.rc; runtime.hip builds it):
constexpr int kSpFirst = 370;                  // 12: FAST_PARTIAL_FIRST_ROUND_CONSTANT
constexpr int kSpK = 382;                      // 22: FAST_PARTIAL_ROUND_CONSTANTS
constexpr int kSpV = 404;                      // 22 x 11: FAST_PARTIAL_ROUND_VS
constexpr int kSpW = 646;                      // 22 x 11 x {limb0, limb1, limb2, 0} as u32: FAST_PARTIAL_ROUND_W_HATS
constexpr int kSpInit = kSpW + 22 * 11 * 2;    // 11 x 11 x {limb0, limb1, limb2, 0}: FAST_PARTIAL_ROUND_INITIAL_MATRIX
constexpr int kSpTableWords = kSpInit + 121 * 2;
// behind those: the tables of the batched partial rounds (partial_rounds3 below; runtime.hip builds them)
constexpr int kP3Groups = 7;                   // rounds 4..24 in threes; round 25 stays a single dense round
constexpr int kP3Tab = kSpTableWords;          // per group a copy of u32[168]: M^3 row-major [0, 144), row 0 of M^2 [144, 156), (M m0)[r] [156, 168)
constexpr int kP3K = kP3Tab + 84 * kP3Groups;  // per group of three rounds 14 u64: c1[0], K2, K3[0..12)
constexpr int kP3W = kP3K + 14 * kP3Groups;    // right behind the addends: u32[14][14] rows for the 12-lane layout -- lane 0: row 0 of M,
                                               // lane 1: row 0 of M^2, lane 2 + r: row r of M^3; then the lane's d1 and d2 coefficients
constexpr int kP3WaveWords = 14 * kP3Groups + 98;  // what stage_round_constants() copies behind the 360 round constants
constexpr int kLeafPairK0 = kP3W + 98;         // 12 u64: sum_{k not in {0, 4}} MDS[r][k] (rc[k])^7 + rc[12 + r] (two_to_one of two leaf digests)
// behind those: the tables of partial_rounds_g (round 4): round 3's MDS layer and the 22 partial rounds as five groups of FOUR and one of three.  Per
// group of G rounds G + 11 rows of 16 u32 -- rows 0 .. G-2: row 0 of M^(i+1) (the S-box input of the group's round i+1; row 0 of M itself
// is written as immediates and its slot unused), rows G-1 .. G+10: M^G -- with the coefficients of the earlier rounds' d_k in
// entries 12 .., then 16 u64 of addends (the round constants pushed through the powers of M).  Rows are 64 bytes: one s_load_dwordx16.
constexpr int kPGTab = (kLeafPairK0 + 12 + 7) & ~7;
constexpr int pg_words(int G) { return 8 * (G + 11) + 16; }
constexpr int kPG4Groups = 5, kPG3Groups = 1;  // MDS layers of rounds 3-6 (the first one follows a FULL S-box layer), 7-10, 11-14, 15-18, 19-22 | 23-25
constexpr int kTableWords = kPGTab + kPG4Groups * pg_words(4) + kPG3Groups * pg_words(3);

struct Dot {
  u64 a0l, a0h, a1l, a1h, a2l, a2h;
};
// first term: no addend
GL_DEV void dot_first(Dot& d, u64 x, const u32* __restrict__ w) {
  const u32 xl = (u32)x, xh = (u32)(x >> 32);
  d.a0l = (u64)xl * w[0], d.a0h = (u64)xh * w[0];
  d.a1l = (u64)xl * w[1], d.a1h = (u64)xh * w[1];
  d.a2l = (u64)xl * w[2], d.a2h = (u64)xh * w[2];
}
GL_DEV void dot_term(Dot& d, u64 x, const u32* __restrict__ w) {
  const u32 xl = (u32)x, xh = (u32)(x >> 32);
  d.a0l += (u64)xl * w[0], d.a0h += (u64)xh * w[0];
  d.a1l += (u64)xl * w[1], d.a1h += (u64)xh * w[1];
  d.a2l += (u64)xl * w[2], d.a2h += (u64)xh * w[2];
}
// sum_k (a_kl + a_kh 2^32) 2^(22 k) mod p, every accumulator < 2^59.  Loose result; rare corrections go to `sticky`.
GL_DEV u64 dot_finish(const Dot& d, u64& sticky) {
  using u128 = unsigned __int128;
  // T_k = t + U 2^32 with U = (a_kl >> 32) + a_kh < 2^60; U = ul + uh 2^32  =>  T_k = (t + ul 2^32) + uh (2^32 - 1)  (mod p), < 2^65
  auto fold = [](u64 al, u64 ah) -> u128 {
    const u64 U = add32((u32)(al >> 32), ah);
    const u64 low = ((u64)(u32)U << 32) | (u32)al;
    const u64 uh = U >> 32;  // < 2^28
    return (u128)low + (((u64)uh << 32) - uh);
  };
  const u128 v = fold(d.a0l, d.a0h) + (fold(d.a1l, d.a1h) << 22) + (fold(d.a2l, d.a2h) << 44);  // < 2^110
  return reduce128((u64)v, (u64)(v >> 64), sticky);
}
// x * k + c (mod p), all three any u64
GL_DEV u64 mul_add_flag(u64 x, u64 k, u64 c, u64& sticky) {
  u64 lo, hi;
  mul_wide(x, k, lo, hi);
  const u64 l = lo + c;
  hi += (l < c);  // x k + c < 2^128
  return reduce128(l, hi, sticky);
}

// ------------------------------------------------------------------ three partial rounds per MDS application (round 3)
// In a partial round only word 0 goes through the S-box: with y = the state after that S-box, c1, c2, c3 the next three rounds'
// constant vectors, m0 the first column of M and d1, d2 = S(x) - x of the two later rounds' word 0,
//     v1 = M y + c1
//     v2 = M v1 + d1 m0 + c2 = M^2 y + d1 m0 + (M c1 + c2)
//     v3 = M v2 + d2 m0 + c3 = M^3 y + d1 (M m0) + d2 m0 + (M^2 c1 + M c2 + c3).
// The two intermediate S-box inputs need only ROW 0 of M y and of M^2 y, and the entries of M^2 / M^3 are still small integers
// (row sums 2^16 / 2^24, M's are 2^8), so a term of a row stays ONE v_mad_u64_u32 per 32-bit half and the accumulators stay below
// 2^58.  Three rounds then cost 2 x 24 + 12 x 28 mads instead of 3 x 288: the one-hash-per-lane layout spends 58 % of its
// instructions in MDS layers, two thirds of them in the partial rounds.  (plonky2's own "fast" partial rounds trade the MDS for ~22
// full 64 x 64 multiplications per round -- right for a CPU, measured 2.3 % slower here, profiles/r02_sparse_flag_form_ab.txt.)
// Folds are the exact form: the top word of a row reaches 2^26 here, too often for the flag.
// (hi:lo) = a - b as two words, borrow-out as a lane mask
GL_DEV u64 sub64_borrow(u64 a, u64 b, u64& borrow) {
  u32 lo, hi;
  asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\tv_subb_co_u32_e64 %1, %2, %4, %6, %2"
      : "=&v"(lo), "=v"(hi), "=&s"(borrow)
      : "v"((u32)a), "v"((u32)(a >> 32)), "v"((u32)b), "v"((u32)(b >> 32)));
  return ((u64)hi << 32) | lo;
}
GL_DEV u64 sub_any(u64 a, u64 b) {  // a - b mod p for any u64 a, b; loose result
  u64 m, m2, m3;
  const u64 d = sub64_borrow(a, b, m);                 // wrapped by +2^64 == +EPS (mod p) in lanes of m: take it back ...
  const u64 d2 = sub32_borrow(d, eps_if(m), m2);       // ... a second time if that wraps too (d < EPS: b within 2^32 of 2^64)
  return sub32_borrow(d2, eps_if(m2), m3);
}
// the same with the second wrap (probability 2^-32) left to the sticky flag
GL_DEV u64 sub_flag(u64 a, u64 b, u64& sticky) {
  u64 m, m2;
  const u64 d = sub64_borrow(a, b, m);
  const u64 d2 = sub32_borrow(d, eps_if(m), m2);
  raise(sticky, m2);
  return d2;
}
// acc += a * k, k wave-uniform (SGPR)
GL_DEV void mac_s(u64& acc, u32 a, u32 k) {
  u64 unused;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(unused) : "v"(a), "s"(k));
}
template <typename P, typename Sbox>
GL_DEV void partial_rounds3(u64 (&s)[12], P rc, int g, Sbox&& sbox, u64& sticky) {
  // The 168 matrix words are the same for every group, but each group reads ITS OWN copy of them: hoisted out of the loop they
  // would sit in SGPRs spilled to VGPR lanes (274 v_readlane per group), and behind an offset the optimiser cannot see through they
  // become vector loads (no proof that the kernel's stores leave them alone); a copy per group is a plain loop-variant scalar load.
  const auto T = as_u32(rc + kP3Tab + 84 * g);
  const P K = rc + kP3K + 14 * g;
  auto finish = [](u64 al, u64 ah) -> u64 {  // (al + ah 2^32) mod p, loose; al, ah < 2^58
    ah = add32((u32)(al >> 32), ah);
    return exact::fold96((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al);
  };
  s[0] = sbox(s[0]);
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  // v1[0] = row 0 of M y + c1[0]
  u64 x1;
  {
    const u64 c = K[0];
    constexpr u32 k0 = poseidon::mds_entry(0, 0);
    u64 al = mac_const_first<k0>(lo[0], (u64)(u32)c), ah = mac_const_first<k0>(hi[0], (u64)(u32)(c >> 32));
    poseidon::static_for<1, 12>([&](auto cc) {
      constexpr int c2 = decltype(cc)::value;
      constexpr u32 k = poseidon::mds_entry(0, c2);
      mac_const<k>(al, lo[c2]);
      mac_const<k>(ah, hi[c2]);
    });
    x1 = finish(al, ah);
  }
  const u64 d1 = sub_flag(sbox(x1), x1, sticky);
  const u32 d1l = (u32)d1, d1h = (u32)(d1 >> 32);
  // v2[0] = row 0 of M^2 y + d1 m0[0] + K2
  u64 x2;
  {
    const u64 c = K[1];
    constexpr u32 m00 = poseidon::mds_entry(0, 0);
    u64 al = mac_const_first<m00>(d1l, (u64)(u32)c), ah = mac_const_first<m00>(d1h, (u64)(u32)(c >> 32));
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const u32 k = T[144 + j];
      mac_s(al, lo[j], k);
      mac_s(ah, hi[j], k);
    }
    x2 = finish(al, ah);
  }
  const u64 d2 = sub_flag(sbox(x2), x2, sticky);
  const u32 d2l = (u32)d2, d2h = (u32)(d2 >> 32);
  // v3 = M^3 y + d1 (M m0) + d2 m0 + K3
  poseidon::static_for<0, 12>([&](auto rcst) {
    constexpr int r = decltype(rcst)::value;
    constexpr u32 m0r = poseidon::mds_entry(r, 0);
    const u64 c = K[2 + r];
    u64 al = mac_const_first<m0r>(d2l, (u64)(u32)c), ah = mac_const_first<m0r>(d2h, (u64)(u32)(c >> 32));
    const u32 km = T[156 + r];
    mac_s(al, d1l, km);
    mac_s(ah, d1h, km);
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const u32 k = T[12 * r + j];
      mac_s(al, lo[j], k);
      mac_s(ah, hi[j], k);
    }
    s[r] = finish(al, ah);
  });
}

// ------------------------------------------------------------------ FOUR partial rounds per MDS application (round 4)
// The same identity one round further: v4 = M^4 y + d1 (M^2 m0) + d2 (M m0) + d3 m0 + (M^3 c1 + M^2 c2 + M c3 + c4), with the three
// intermediate S-box inputs from row 0 of M y, M^2 y, M^3 y.  The entries of M^4 are 29-bit numbers, still one v_mad_u64_u32 per term
// and 32-bit half, but the row sums reach 1.04 x 2^32: a chain of twelve terms can pass 2^64 -- only in its LAST link, because every
// entry of M^4 is above 0.083 x 2^32 (runtime.hip checks both facts when it builds the table) -- and only for states whose twelve
// halves are all within 4 % of 2^32.  The last link's carry-out therefore goes to the sticky flag (exact redo), and so does the
// carry of "hi chain += lo chain >> 32".  Per round 180 instructions against 208 for groups of three.  The first group takes the MDS
// layer of the last full round of the first half as its first application (LEAD = false): 1 + 22 = 5 x 4 + 3 layers replace that
// matrix-pipe layer, 7 x 3 and one dense round.
GL_DEV void mac_s_flag(u64& acc, u32 a, u32 k, u64& sticky) {
  u64 cy;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(cy) : "v"(a), "s"(k));
  raise(sticky, cy);
}
GL_DEV u64 add32_flag(u32 a, u64 c, u64& sticky) {
  u64 d, cy;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cy) : "v"(a), "v"(c));
  raise(sticky, cy);
  return d;
}
// `tab`: the group's table (kPGTab + ...), a loop-variant address for the reason partial_rounds3 gives
// LEAD: the group starts with the S-box of its first round on word 0.  Without it the caller has applied a full S-box layer: the MDS
// layer of the last full round of the first half then is the group's first MDS application (and one matrix-pipe layer fewer).
template <int G, bool LEAD = true, typename P, typename Sbox>
GL_DEV void partial_rounds_g(u64 (&s)[12], P tab, Sbox&& sbox, u64& sticky) {
  static_assert(G == 3 || G == 4, "groups of three or four");
  const auto T = as_u32(tab);
  const P K = tab + 8 * (G + 11);
  // (al + ah 2^32) mod p, loose, for al, ah < 2^58: ah = ahl + ahh 2^32 and 2^64 == EPS, so al + ahh EPS (< 2^59: no carry) and then
  // ahl onto the high word, whose carry (one row in ~2^6) is another EPS.  No (word, word) pair has to be put together: no v_mov.
  auto finish = [](u64 al, u64 ah) -> u64 {
    u64 unused, k;
    const u64 r = mad_eps_carry((u32)(ah >> 32), al, unused);
    u32 rh = (u32)(r >> 32);
    asm("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(rh), "=s"(k) : "v"((u32)ah));
    return add32(eps_if(k), ((u64)rh << 32) | (u32)r);  // wrapped high word < 2^27: + EPS cannot wrap again
  };
  if constexpr (LEAD) s[0] = sbox(s[0]);
  u32 lo[12], hi[12], dl[G - 1], dh[G - 1];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  constexpr u32 m00 = poseidon::mds_entry(0, 0);
  poseidon::static_for<1, G>([&](auto ic) {  // x_i = word 0 in front of the S-box of the group's round i; d_i = S(x_i) - x_i
    constexpr int i = decltype(ic)::value;
    const u64 c = K[i - 1];
    u64 al, ah;
    if constexpr (i == 1) {
      al = mac_const_first<m00>(lo[0], (u64)(u32)c), ah = mac_const_first<m00>(hi[0], (u64)(u32)(c >> 32));
      poseidon::static_for<1, 12>([&](auto cc) {
        constexpr int j = decltype(cc)::value;
        constexpr u32 k = poseidon::mds_entry(0, j);
        mac_const<k>(al, lo[j]);
        mac_const<k>(ah, hi[j]);
      });
    } else {
      const auto R = T + 16 * (i - 1);
      al = mac_const_first<m00>(dl[i - 2], (u64)(u32)c), ah = mac_const_first<m00>(dh[i - 2], (u64)(u32)(c >> 32));  // d_(i-1) m0[0]
#pragma unroll
      for (int k = 0; k < i - 2; ++k) {
        const u32 w = R[12 + k];
        mac_s(al, dl[k], w);
        mac_s(ah, dh[k], w);
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const u32 w = R[j];
        mac_s(al, lo[j], w);
        mac_s(ah, hi[j], w);
      }
    }
    const u64 x = finish(al, ah);  // accumulators < 2^57 (row sums of M^3 < 2^25)
    const u64 d = sub_flag(sbox(x), x, sticky);
    dl[i - 1] = (u32)d, dh[i - 1] = (u32)(d >> 32);
  });
  poseidon::static_for<0, 12>([&](auto rcst) {
    constexpr int r = decltype(rcst)::value;
    constexpr u32 m0r = poseidon::mds_entry(r, 0);
    const auto R = T + 16 * (G - 1 + r);
    const u64 c = K[G - 1 + r];
    u64 al = mac_const_first<m0r>(dl[G - 2], (u64)(u32)c), ah = mac_const_first<m0r>(dh[G - 2], (u64)(u32)(c >> 32));
#pragma unroll
    for (int k = 0; k < G - 2; ++k) {
      const u32 w = R[12 + k];
      mac_s(al, dl[k], w);
      mac_s(ah, dh[k], w);
    }
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const u32 w = R[j];
      mac_s(al, lo[j], w);
      mac_s(ah, hi[j], w);
    }
    if constexpr (G == 4) {  // the one link that can pass 2^64, then a full 32-bit top word
      mac_s_flag(al, lo[11], R[11], sticky);
      mac_s_flag(ah, hi[11], R[11], sticky);
      ah = add32_flag((u32)(al >> 32), ah, sticky);
      s[r] = exact::fold96((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al);
    } else {
      mac_s(al, lo[11], R[11]);
      mac_s(ah, hi[11], R[11]);
      s[r] = finish(al, ah);
    }
  });
}

// Input: any u64 words.  Output: loose u64 words, valid iff the returned sticky mask is 0 for the whole wave.
// `rc`: the 360 round constants in GLOBAL memory (kernel argument: base + immediate offsets let the compiler
// fetch a whole round with wide s_load_dwordx8/x16; the __constant__ symbol would cost a PC-relative address
// computation per element).
// CAP_ZERO: the caller guarantees s[8..11] == 0 on entry (two_to_one): their first S-box input is the round
//   constant itself, so (rc[8+i])^7 is read precomputed from rc[360 + i].
// OUT_ROWS: number of output words the caller needs (4 for a hash => the last MDS layer computes 4 rows).
// EXACT: use the exact-form primitives (no flag is ever raised; ~8 % more instructions) -- the redo path of a
//   flagged wave.
// LEAF_PAIR (with CAP_ZERO): the caller additionally guarantees s[1..3] == s[5..7] == 0 -- two_to_one of two leaf digests
//   [leaf, 0, 0, 0] (hash_or_noop's zero padding, quirk Q1), i.e. half of all hashes of a tree build.  Only words 0 and 4 go
//   through the first S-box layer; the other ten S-box outputs are the constants (rc[i])^7, read from rc[360 + ..].
// SPARSE: the 22 partial rounds in the sparse form above (same function; the dense form is the default and the redo path).
// MFMA: 1 = every dense MDS layer on the matrix pipe, 2 = only those of the 22 partial rounds (mds_layer_mfma; `mc` from
//   mfma_ctx_init(), made while every lane of the wave was still active).  Same function, same flag semantics.
//   3 = every 12-row dense MDS layer (seven of the 8 full rounds + the last partial round) as one v_mfma_i32_32x32x32_i8 per limb
//   (mds_layer_mfma32; `mc` from mfma32_ctx_init()), the other partial rounds batched as in P3.
// P3: 1 = the 22 partial rounds as 7 groups of three with one MDS application each (partial_rounds3) + one single round;
//   2 = the MDS layer of round 3 and the 22 partial rounds as five groups of four and one of three (partial_rounds_g).
// XF: every 96 -> 64 bit fold of an MDS row in its exact form (+2 instructions per row, +1.6 % per hash).  What is left to raise the
//   flag then has probability 2^-32 per operation, so a launch practically never redoes a hash -- for kernels of ONE hash per lane,
//   where a redo (28 k instructions on one wave, ~60 us) in the last wavefronts sets the duration of the whole launch.
// FIRST_DONE: the caller passes the state in front of round 1's S-boxes (it did round 0 itself: the proof-of-work grind shares eleven of
//   the twelve first-round S-boxes between all candidates of a proof).  LAST_ROW >= 0: only that word of the result is computed.
template <bool CAP_ZERO = false, int OUT_ROWS = 12, bool EXACT = false, bool LEAF_PAIR = false, bool SPARSE = false, int MFMA = 0,
          int P3 = (!EXACT && !SPARSE && MFMA == 0), typename RC = const u64*, int MULV = 0, bool FIRST_DONE = false, int LAST_ROW = -1,
          bool XF = false>
GL_DEV u64 permute_impl(u64 (&s)[12], RC rc, const MfmaCtx* mc) {
  static_assert(!P3 || (!EXACT && !SPARSE && (MFMA == 0 || MFMA == 3)), "partial_rounds3 belongs to the dense flag form");
  u64 sticky = 0;
  static_assert(!(MFMA && (EXACT || SPARSE)), "the matrix-pipe MDS exists in the flag form with dense partial rounds only");
  auto mds4 = [&](auto add_tag, auto rows_tag, RC add, auto in_partial_round) {
    constexpr bool kAdd = decltype(add_tag)::value;
    constexpr int kRows = decltype(rows_tag)::value;
    if constexpr (MFMA == 1 || (MFMA == 2 && decltype(in_partial_round)::value)) mds_layer_mfma<kAdd, kRows>(s, add, sticky, *mc);
    else if constexpr (MFMA == 3 && kRows == 12) mds_layer_mfma32<kAdd, kRows, RC, XF>(s, add, sticky, *mc);  // (4 rows: 104 mads beat it)
    else mds_layer<kAdd, kRows, EXACT || XF>(s, add, sticky);
  };
  auto mds = [&](auto add_tag, auto rows_tag, RC add) { mds4(add_tag, rows_tag, add, std::false_type{}); };
  using T = std::true_type;
  using R12 = std::integral_constant<int, 12>;
  auto sbox = [&](u64 x) -> u64 {
    if constexpr (EXACT) return exact::pow7(x);
    else return pow7<MULV>(x, sticky);
  };
  static_assert(!LEAF_PAIR || CAP_ZERO, "LEAF_PAIR implies zero capacity words");
  constexpr int kVar = CAP_ZERO ? 8 : 12;
  static_assert(!FIRST_DONE || (!CAP_ZERO && !LEAF_PAIR), "FIRST_DONE: the caller did round 0 itself");
  if constexpr (FIRST_DONE) {  // s = the state in front of round 1's S-boxes (round 0's MDS layer with round 1's constants in)
  } else if constexpr (LEAF_PAIR && MFMA == 3) {  // round 0: words 0 and 4 only, and of its MDS layer only their two columns --
    const u64 y0 = sbox(gl::add_c(s[0], rc[0])), y4 = sbox(gl::add_c(s[4], rc[4]));  // the other ten S-box outputs are constants,
    const u32 y0l = (u32)y0, y0h = (u32)(y0 >> 32), y4l = (u32)y4, y4h = (u32)(y4 >> 32);  // their share sits in the table
    poseidon::static_for<0, 12>([&](auto rcst) {
      constexpr int r = decltype(rcst)::value;
      constexpr u32 m0 = poseidon::mds_entry(r, 0), m4 = poseidon::mds_entry(r, 4);
      const u64 c = rc[kLeafPairK0 + r];
      u64 al = mac_const_first<m0>(y0l, (u64)(u32)c), ah = mac_const_first<m0>(y0h, (u64)(u32)(c >> 32));
      mac_const<m4>(al, y4l);
      mac_const<m4>(ah, y4h);
      ah = add32((u32)(al >> 32), ah);
      if constexpr (XF) {
        s[r] = exact::fold96((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al);
      } else {
        u64 cm;
        s[r] = mad_eps_carry((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al, cm);
        raise(sticky, cm);
      }
    });
  } else if constexpr (LEAF_PAIR) {  // round 0: words 0 and 4 only
    s[0] = sbox(gl::add_c(s[0], rc[0]));
    s[4] = sbox(gl::add_c(s[4], rc[4]));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      s[1 + i] = rc[364 + i];
      s[5 + i] = rc[367 + i];
    }
#pragma unroll
    for (int i = 8; i < 12; ++i) s[i] = rc[360 + (i - 8)];
    mds(T{}, R12{}, rc + 12);
  } else {  // round 0
#pragma unroll
    for (int i = 0; i < kVar; ++i) s[i] = gl::add_c(s[i], rc[i]);  // round 0 constants, exact
#pragma unroll
    for (int i = 0; i < kVar; ++i) s[i] = sbox(s[i]);
    if constexpr (CAP_ZERO) {
#pragma unroll
      for (int i = 8; i < 12; ++i) s[i] = rc[360 + (i - 8)];
    }
    mds(T{}, R12{}, rc + 12);
  }
  static_assert(!(SPARSE && EXACT), "the exact redo path keeps the dense partial rounds");
#pragma unroll 1
  for (int r = 1; r < POSEIDON_HALF_FULL_ROUNDS - (SPARSE || P3 == 2 ? 1 : 0); ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(T{}, R12{}, rc + 12 * (r + 1));
  }
  if constexpr (SPARSE) {
    {  // last full round of the first half: the addend of its MDS layer is FIRST instead of round 4's constants
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
      mds_layer<true, 12, false>(s, rc + kSpFirst, sticky);
    }
    {  // s[1..] = INIT * s[1..]
      u64 t[11];
      const u32* init = reinterpret_cast<const u32*>(rc + kSpInit);
#pragma unroll
      for (int rr = 0; rr < 11; ++rr) {
        Dot d;
        dot_first(d, s[1], init + 4 * (11 * rr));
#pragma unroll
        for (int c = 1; c < 11; ++c) dot_term(d, s[c + 1], init + 4 * (11 * rr + c));
        t[rr] = dot_finish(d, sticky);
      }
#pragma unroll
      for (int rr = 0; rr < 11; ++rr) s[rr + 1] = t[rr];
    }
#pragma unroll 1
    for (int pr = 0; pr < POSEIDON_PARTIAL_ROUNDS; ++pr) {
      const u64 s0 = gl::add_c(sbox(s[0]), rc[kSpK + pr]);
      const u32* w = reinterpret_cast<const u32*>(rc + kSpW) + 4 * 11 * pr;
      const u64* v = rc + kSpV + 11 * pr;
      Dot d;
      {  // 25 s0: limb 0 only
        const u32 xl = (u32)s0, xh = (u32)(s0 >> 32);
        d.a0l = (u64)xl * 25u, d.a0h = (u64)xh * 25u;
      }
      {
        const u32 xl = (u32)s[1], xh = (u32)(s[1] >> 32);
        d.a0l += (u64)xl * w[0], d.a0h += (u64)xh * w[0];
        d.a1l = (u64)xl * w[1], d.a1h = (u64)xh * w[1];
        d.a2l = (u64)xl * w[2], d.a2h = (u64)xh * w[2];
      }
#pragma unroll
      for (int j = 1; j < 11; ++j) dot_term(d, s[j + 1], w + 4 * j);
#pragma unroll
      for (int j = 0; j < 11; ++j) s[j + 1] = mul_add_flag(s0, v[j], s[j + 1], sticky);
      s[0] = dot_finish(d, sticky);
    }
    // the constants of the first full round of the second half (the dense form folds them into the previous MDS layer)
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::add_c(s[i], rc[12 * (POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS) + i]);
  } else if constexpr (P3 == 2) {
    static_assert(POSEIDON_PARTIAL_ROUNDS + 1 == 4 * kPG4Groups + 3 * kPG3Groups, "round 3's layer + 22 partial rounds = five groups of four + one of three");
    {  // the last full round of the first half: its MDS layer opens the first group
#pragma unroll
      for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
      partial_rounds_g<4, false>(s, rc + kPGTab, sbox, sticky);
    }
#pragma unroll 1
    for (int g = 1; g < kPG4Groups; ++g) partial_rounds_g<4>(s, rc + kPGTab + pg_words(4) * g, sbox, sticky);
#pragma unroll 1
    for (int g = 0; g < kPG3Groups; ++g) partial_rounds_g<3>(s, rc + kPGTab + pg_words(4) * kPG4Groups + pg_words(3) * g, sbox, sticky);
  } else if constexpr (P3) {
    static_assert(POSEIDON_PARTIAL_ROUNDS == 3 * kP3Groups + 1, "7 groups of three + one round");
#pragma unroll 1
    for (int g = 0; g < kP3Groups; ++g) partial_rounds3(s, rc, g, sbox, sticky);
    {
      constexpr int r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS - 1;
      s[0] = sbox(s[0]);
      mds4(T{}, R12{}, rc + 12 * (r + 1), T{});
    }
  } else {
#pragma unroll 1
    for (int r = POSEIDON_HALF_FULL_ROUNDS; r < POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS; ++r) {
      s[0] = sbox(s[0]);
      mds4(T{}, R12{}, rc + 12 * (r + 1), T{});
    }
  }
#pragma unroll 1
  for (int r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS; r < POSEIDON_ROUNDS - 1; ++r) {
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
    mds(T{}, R12{}, rc + 12 * (r + 1));
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = sbox(s[i]);
  if constexpr (LAST_ROW >= 0) s[LAST_ROW] = mds_row<LAST_ROW>(s, sticky);  // only this word of the result is valid
  else mds(std::false_type{}, std::integral_constant<int, OUT_ROWS>{}, RC{});
  return sticky;
}
// `rc` = the GLOBAL constant table (p2mt::perm_ctx().rc), never the LDS copy of the 12-lane layout
template <bool CAP_ZERO = false, int OUT_ROWS = 12, bool EXACT = false, bool LEAF_PAIR = false, bool SPARSE = false, int MFMA = 0,
          int P3 = (!EXACT && !SPARSE && MFMA == 0), int MULV = 0, bool FIRST_DONE = false, int LAST_ROW = -1, bool XF = false>
GL_DEV u64 permute(u64 (&s)[12], const u64* __restrict__ rc, const MfmaCtx* mc = nullptr) {
  if constexpr (SPARSE) return permute_impl<CAP_ZERO, OUT_ROWS, EXACT, LEAF_PAIR, SPARSE, MFMA, P3, const u64*, MULV, FIRST_DONE, LAST_ROW, XF>(s, rc, mc);
  else return permute_impl<CAP_ZERO, OUT_ROWS, EXACT, LEAF_PAIR, SPARSE, MFMA, P3, ctab, MULV, FIRST_DONE, LAST_ROW, XF>(s, as_const_table(rc), mc);
}

}  // namespace poseidon_fast
