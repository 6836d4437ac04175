// poseidon.hip.h -- width-12 Poseidon permutation over Goldilocks, one hash per lane.
//
// Replaces plonky2 @3b21b87 hash/poseidon.rs + hash/hashing.rs (third-party, absent), reached from
// /root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:23,33,45 and
// /root/reference/src/mmr/merkle_mountain_ranges.rs:91,96,111,125 (PoseidonHash::{two_to_one,hash_or_noop}).
//
// Mapping to CDNA4: the whole 12 x u64 state lives in 24 VGPRs of ONE lane, so a wave64 runs 64
// independent permutations with no cross-lane traffic (the alternative "one wavefront per node" layout
// leaves 52/64 lanes idle in the 22 partial rounds, where only lane 0 has non-linear work).  Round bodies
// are fully unrolled over the 12 state words; the round loops themselves are kept rolled
// (#pragma unroll 1) so the kernel body (~30 KB) stays inside the 64 KB instruction cache, and the round
// constants are wave-uniform scalar loads (s_load) from __constant__ memory, costing no VGPRs.
//
// MDS layer (small constants < 2^6): two formulations, selected by the MDS template argument:
//   MDS_MAD64  32-bit halves x constant with v_mad_u64_u32 (24 per output word);
//   MDS_DOT2   16-bit limbs of two neighbouring state words packed per VGPR (v_perm_b32) and
//              v_dot2_u32_u16 against packed constant pairs: 2 MACs per instruction, 6 per (row, limb).
// Partial rounds: PARTIAL_NAIVE (spec form, reuses the MDS layer) or PARTIAL_FAST (sparse-matrix form,
// constants derived in tools/poseidon_spec.py).  All variants are bit-identical (tests/test_parity_gpu.py).
#pragma once
#include <type_traits>

#include "gl64.hip.h"
#define P2MT_QUAL static __device__ __constant__ const
#include "poseidon_constants.h"

namespace poseidon {

using gl::u32;
using gl::u64;

enum { MDS_MAD64 = 0, MDS_DOT2 = 1 };
enum { PARTIAL_NAIVE = 0, PARTIAL_FAST = 1 };

// The generated tables live in constant memory (statically initialised, served by the scalar cache).
#define kRC POSEIDON_RC
#define kFastFirst POSEIDON_FAST_FIRST
#define kFastK POSEIDON_FAST_K
#define kFastInit POSEIDON_FAST_INIT
#define kFastV POSEIDON_FAST_V
#define kFastWHat POSEIDON_FAST_W_HAT

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- indices stay constant expressions
// whatever the optimiser's unroll thresholds are.
template <int I, int N, typename F>
GL_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// MDS[r][c] = CIRC[(c - r) mod 12] + (r == c ? DIAG[r] : 0)
constexpr u32 mds_entry(int r, int c) {
  constexpr u32 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  return circ[((c - r) % 12 + 12) % 12] + ((r == 0 && c == 0) ? 8u : 0u);
}

// ------------------------------------------------------------------ MDS, v_mad_u64_u32 formulation
GL_DEV void mds_mad64(u64 (&s)[12]) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  static_for<0, 12>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    u64 al = 0, ah = 0;  // each < 264 * 2^32
    static_for<0, 12>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      constexpr u32 k = mds_entry(r, c);
      al += (u64)lo[c] * k;
      ah += (u64)hi[c] * k;
    });
    // X = al + ah * 2^32  (< 2^73)
    const u64 xl = al + (ah << 32);
    const u32 xh = (u32)(ah >> 32) + (xl < al ? 1u : 0u);
    s[r] = gl::reduce96(xl, xh);
  });
}

// ------------------------------------------------------------------ MDS, v_dot2_u32_u16 formulation
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

GL_DEV u32 dot2(u32 a, u32 b, u32 c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b), c, false);
}

// packed constant pair for row r, column pair j: (MDS[r][2j], MDS[r][2j+1])
constexpr u32 mds_pair(int r, int j) { return mds_entry(r, 2 * j) | (mds_entry(r, 2 * j + 1) << 16); }

GL_DEV void mds_dot2(u64 (&s)[12]) {
  // pk[k][j]: 16-bit limb k of s[2j] (low half) and of s[2j+1] (high half)
  u32 pk[4][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const u32 l0 = (u32)s[2 * j], h0 = (u32)(s[2 * j] >> 32);
    const u32 l1 = (u32)s[2 * j + 1], h1 = (u32)(s[2 * j + 1] >> 32);
    pk[0][j] = __builtin_amdgcn_perm(l1, l0, 0x05040100u);
    pk[1][j] = __builtin_amdgcn_perm(l1, l0, 0x07060302u);
    pk[2][j] = __builtin_amdgcn_perm(h1, h0, 0x05040100u);
    pk[3][j] = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
  }
  static_for<0, 12>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    u32 acc[4];  // each < 264 * 2^16 < 2^25
    static_for<0, 4>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      u32 a = 0;
      static_for<0, 6>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        a = dot2(pk[k][j], mds_pair(r, j), a);
      });
      acc[k] = a;
    });
    // X = acc0 + acc1*2^16 + acc2*2^32 + acc3*2^48  (< 2^74)
    const u64 a01 = (u64)acc[0] + ((u64)acc[1] << 16);  // < 2^42
    const u64 a23 = (u64)acc[2] + ((u64)acc[3] << 16);  // < 2^42
    const u64 xl = a01 + (a23 << 32);
    const u32 xh = (u32)(a23 >> 32) + (xl < a01 ? 1u : 0u);
    s[r] = gl::reduce96(xl, xh);
  });
}

template <int MDS>
GL_DEV void mds_layer(u64 (&s)[12]) {
  if constexpr (MDS == MDS_DOT2) mds_dot2(s);
  else mds_mad64(s);
}

// ------------------------------------------------------------------ rounds
template <int MDS>
GL_DEV void full_round(u64 (&s)[12], const u64* __restrict__ rc) {
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl::pow7_ref(gl::add_c(s[i], rc[i]));
  mds_layer<MDS>(s);
}

template <int MDS>
GL_DEV void partial_rounds_naive(u64 (&s)[12]) {
#pragma unroll 1
  for (int r = POSEIDON_HALF_FULL_ROUNDS; r < POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS; ++r) {
    const u64* rc = kRC + 12 * r;
#pragma unroll
    for (int i = 0; i < 12; ++i) s[i] = gl::add_c(s[i], rc[i]);
    s[0] = gl::pow7_ref(s[0]);
    mds_layer<MDS>(s);
  }
}

// 3-word accumulator for sums of up to 2^32 128-bit products: value = lo + hi*2^64 + top*2^128
struct Acc192 {
  u64 lo, hi;
  u32 top;
};
GL_DEV void acc_mul(Acc192& a, u64 x, u64 y) {
  u64 pl, ph;
  gl::mul_wide(x, y, pl, ph);
  a.lo += pl;
  const u64 c0 = a.lo < pl;
  a.hi += ph;
  u32 c1 = a.hi < ph;
  a.hi += c0;
  c1 += a.hi < c0;
  a.top += c1;
}
// 2^128 = -2^32 (mod p)
GL_DEV u64 acc_reduce(const Acc192& a) {
  const u64 r = gl::reduce128(a.lo, a.hi);
  return gl::sub_c(r, (u64)a.top << 32);  // top < 2^8 here => (top << 32) is canonical
}

GL_DEV void partial_rounds_fast(u64 (&s)[12]) {
#pragma unroll
  for (int i = 0; i < 12; ++i) s[i] = gl::add_c(s[i], kFastFirst[i]);
  {  // s[1..] <- init * s[1..]   (11x11 dense, once per permutation)
    u64 t[11];
#pragma unroll
    for (int r = 0; r < 11; ++r) {
      Acc192 a = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < 11; ++c) acc_mul(a, s[c + 1], kFastInit[r * 11 + c]);
      t[r] = acc_reduce(a);
    }
#pragma unroll
    for (int r = 0; r < 11; ++r) s[r + 1] = t[r];
  }
#pragma unroll 1
  for (int i = 0; i < POSEIDON_PARTIAL_ROUNDS; ++i) {
    const u64 s0 = gl::add_c(gl::pow7_ref(s[0]), kFastK[i]);
    Acc192 d = {0, 0, 0};
    acc_mul(d, s0, (u64)POSEIDON_M00);
    const u64* wh = kFastWHat + 11 * i;
    const u64* v = kFastV + 11 * i;
#pragma unroll
    for (int j = 0; j < 11; ++j) acc_mul(d, s[j + 1], wh[j]);
#pragma unroll
    for (int j = 0; j < 11; ++j) s[j + 1] = gl::mul_add(s0, v[j], s[j + 1]);
    s[0] = acc_reduce(d);
  }
}

// The permutation.  Input: any u64 words; output: loose u64 words (canonicalise what leaves the kernel).
template <int MDS, int PARTIAL>
GL_DEV void permute(u64 (&s)[12]) {
#pragma unroll 1
  for (int r = 0; r < POSEIDON_HALF_FULL_ROUNDS; ++r) full_round<MDS>(s, kRC + 12 * r);
  if constexpr (PARTIAL == PARTIAL_FAST) partial_rounds_fast(s);
  else partial_rounds_naive<MDS>(s);
#pragma unroll 1
  for (int r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS; r < POSEIDON_ROUNDS; ++r)
    full_round<MDS>(s, kRC + 12 * r);
}

// Hasher::two_to_one: perm([l, r, 0, 0, 0, 0])[0..4], canonical output.
template <int MDS, int PARTIAL>
GL_DEV void two_to_one(const u64 (&l)[4], const u64 (&r)[4], u64 (&out)[4]) {
  u64 s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  permute<MDS, PARTIAL>(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = gl::canon(s[i]);
}

}  // namespace poseidon
