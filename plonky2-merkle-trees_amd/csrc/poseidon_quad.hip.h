// poseidon_quad.hip.h -- the Poseidon permutation on FOUR lanes (one DPP quad) per hash.
//
// Third layout of the same function (bit-identical to poseidon.hip.h / poseidon_fast.hip.h), for batches that are
// too small to fill the chip one-hash-per-lane (2^12 .. 2^16 items: every launch there costs a full single-hash
// latency of ~60 us) and too big for one-wavefront-per-node.  Lane q of a quad owns state words 3q, 3q+1, 3q+2:
//   * S-boxes: 3 per lane in a full round (4x shorter), word 0's in a partial round;
//   * MDS: every word is broadcast inside the quad with v_mov_b32_dpp quad_perm:[s,s,s,s] (full-rate moves, no
//     LDS), then each lane runs the two mad chains of ITS three rows against per-lane row constants in VGPRs;
//   * the next round's constants are folded into the chains and prefetched one round ahead; arithmetic is the exact
//     form of poseidon_fast (explicit fix-ups, no sticky flag): in a latency-bound launch a flagged wave redoing
//     its work serially would set the duration of the whole launch.
// ~5.6 k instructions per wave for 16 hashes: 0.35x the latency of the one-hash-per-lane kernel at 0.73x its
// throughput, so it is used only where latency, not issue rate, is the bound.
#pragma once
#include "poseidon_fast.hip.h"

namespace poseidon_quad {

using gl::u32;
using gl::u64;

// value of lane SRC of this lane's quad
template <int SRC>
GL_DEV u32 quad_bcast(u32 v) {
  return (u32)__builtin_amdgcn_mov_dpp((int)v, SRC * 0x55, 0xf, 0xf, true);
}

struct Lane {
  u32 q;          // lane index inside the quad
  u32 k[3][12];   // k[i][c] = MDS[3q + i][c]
  const u64* rc;  // round-constant table + 3q: rc[12 r + i] is this lane's constant for word 3q+i of round r
  // batched partial rounds (poseidon_fast::partial_rounds3): this lane's three rows of M^3, its coefficients of d1 and d2, and the
  // table of addends (per group g: p3k[14 g] = c1[0], [14 g + 1] = K2, [14 g + 2 + r] = K3[r]) and row 0 of M^2
  u32 k3[3][12], cf1[3], cf2[3];
  const u64* p3k;
  const u32* m2row0;
};

GL_DEV void lane_init(Lane& ln, const u64* __restrict__ rc_table) {
  ln.q = threadIdx.x & 3;
  ln.rc = rc_table + 3 * ln.q;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const u32 r = 3 * ln.q + i;
#pragma unroll
    for (int c = 0; c < 12; ++c) ln.k[i][c] = (u32)POSEIDON_MDS_CIRC[(c + 12 - r) % 12] + ((r == 0 && c == 0) ? 8u : 0u);
  }
  // (rc_table is the GLOBAL constant table here -- the quad kernels do not stage it --, so the regions behind the 360 round
  // constants are there)
  const u32* t = reinterpret_cast<const u32*>(rc_table + poseidon_fast::kP3Tab);  // copy 0 of [M^3 | row 0 of M^2 | M m0]
  ln.p3k = rc_table + poseidon_fast::kP3K;
  ln.m2row0 = t + 144;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const u32 r = 3 * ln.q + i;
#pragma unroll
    for (int c = 0; c < 12; ++c) ln.k3[i][c] = t[12 * r + c];
    ln.cf1[i] = t[156 + r];
    ln.cf2[i] = (u32)POSEIDON_MDS_CIRC[(12 - r) % 12] + (r == 0 ? 8u : 0u);  // m0[r] = MDS[r][0]
  }
}

// One permutation of the state spread over the quad: x[i] = word 3q+i.  Input: any u64; output: loose u64 (exact).
GL_DEV void permute(u64 (&x)[3], const Lane& ln) {
  u64 cn[3];  // constants of round r+1, fetched one round ahead
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    x[i] = gl::add_c(x[i], ln.rc[i]);
    cn[i] = ln.rc[12 + i];
  }
  auto round = [&](bool full, bool add, const u64 (&cf)[3]) {
    if (full) {
#pragma unroll
      for (int i = 0; i < 3; ++i) x[i] = poseidon_fast::exact::pow7(x[i]);
    } else {
      const u64 y = poseidon_fast::exact::pow7(x[0]);
      if (ln.q == 0) x[0] = y;
    }
    u32 lo[12], hi[12];
    poseidon::static_for<0, 3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const u32 l = (u32)x[i], h = (u32)(x[i] >> 32);
      lo[0 + i] = quad_bcast<0>(l); hi[0 + i] = quad_bcast<0>(h);
      lo[3 + i] = quad_bcast<1>(l); hi[3 + i] = quad_bcast<1>(h);
      lo[6 + i] = quad_bcast<2>(l); hi[6 + i] = quad_bcast<2>(h);
      lo[9 + i] = quad_bcast<3>(l); hi[9 + i] = quad_bcast<3>(h);
    });
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      u64 al = add ? (u64)(u32)cf[i] : 0, ah = add ? (u64)(u32)(cf[i] >> 32) : 0;
#pragma unroll
      for (int c = 0; c < 12; ++c) {
        al += (u64)lo[c] * ln.k[i][c];
        ah += (u64)hi[c] * ln.k[i][c];
      }
      ah = poseidon_fast::add32((u32)(al >> 32), ah);
      const u64 val = ((u64)(u32)ah << 32) | (u32)al;
      x[i] = poseidon_fast::exact::fold96((u32)(ah >> 32), val);
    }
  };
  auto fold = [](u64 al, u64 ah) -> u64 {
    ah = poseidon_fast::add32((u32)(al >> 32), ah);
    return poseidon_fast::exact::fold96((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al);
  };
  // rounds k = 4 + 3g, k + 1, k + 2 with ONE broadcast of the state (poseidon_fast::partial_rounds3 has the algebra): every lane of
  // the quad computes the two intermediate S-box inputs (row 0 of M y and of M^2 y: the same values in all four lanes, so d1 and d2
  // need no broadcast), then its own three rows of M^3 y + d1 (M m0) + d2 m0 + K3
  auto group = [&](int g) {
    const u64* K = ln.p3k + 14 * g;
    const u64 y = poseidon_fast::exact::pow7(x[0]);
    if (ln.q == 0) x[0] = y;
    u32 lo[12], hi[12];
    poseidon::static_for<0, 3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const u32 l = (u32)x[i], h = (u32)(x[i] >> 32);
      lo[0 + i] = quad_bcast<0>(l); hi[0 + i] = quad_bcast<0>(h);
      lo[3 + i] = quad_bcast<1>(l); hi[3 + i] = quad_bcast<1>(h);
      lo[6 + i] = quad_bcast<2>(l); hi[6 + i] = quad_bcast<2>(h);
      lo[9 + i] = quad_bcast<3>(l); hi[9 + i] = quad_bcast<3>(h);
    });
    u64 x1;
    {
      const u64 c = K[0];
      u64 al = (u64)(u32)c, ah = (u64)(u32)(c >> 32);
      poseidon::static_for<0, 12>([&](auto cc) {
        constexpr int c2 = decltype(cc)::value;
        constexpr u32 k = poseidon::mds_entry(0, c2);
        al += (u64)lo[c2] * k;
        ah += (u64)hi[c2] * k;
      });
      x1 = fold(al, ah);
    }
    const u64 d1 = poseidon_fast::sub_any(poseidon_fast::exact::pow7(x1), x1);
    const u32 d1l = (u32)d1, d1h = (u32)(d1 >> 32);
    u64 x2;
    {
      const u64 c = K[1];
      constexpr u32 m00 = poseidon::mds_entry(0, 0);
      u64 al = (u64)(u32)c + (u64)d1l * m00, ah = (u64)(u32)(c >> 32) + (u64)d1h * m00;
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const u32 k = ln.m2row0[j];
        al += (u64)lo[j] * k;
        ah += (u64)hi[j] * k;
      }
      x2 = fold(al, ah);
    }
    const u64 d2 = poseidon_fast::sub_any(poseidon_fast::exact::pow7(x2), x2);
    const u32 d2l = (u32)d2, d2h = (u32)(d2 >> 32);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const u64 c = K[2 + 3 * ln.q + i];
      u64 al = (u64)(u32)c + (u64)d1l * ln.cf1[i] + (u64)d2l * ln.cf2[i];
      u64 ah = (u64)(u32)(c >> 32) + (u64)d1h * ln.cf1[i] + (u64)d2h * ln.cf2[i];
#pragma unroll
      for (int c2 = 0; c2 < 12; ++c2) {
        al += (u64)lo[c2] * ln.k3[i][c2];
        ah += (u64)hi[c2] * ln.k3[i][c2];
      }
      x[i] = fold(al, ah);
    }
  };
#pragma unroll 1
  for (int r = 0; r < POSEIDON_HALF_FULL_ROUNDS; ++r) {
    const u64 cf[3] = {cn[0], cn[1], cn[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) cn[i] = ln.rc[12 * (r + 2) + i];
    round(true, true, cf);
  }
  static_assert(POSEIDON_PARTIAL_ROUNDS == 3 * poseidon_fast::kP3Groups + 1, "7 groups of three + one round");
#pragma unroll 1
  for (int g = 0; g < poseidon_fast::kP3Groups; ++g) group(g);
#pragma unroll
  for (int i = 0; i < 3; ++i) cn[i] = ln.rc[12 * (POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS) + i];
#pragma unroll 1
  for (int r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS - 1; r < POSEIDON_ROUNDS - 1; ++r) {
    const u64 cf[3] = {cn[0], cn[1], cn[2]};
    if (r + 2 < POSEIDON_ROUNDS) {
#pragma unroll
      for (int i = 0; i < 3; ++i) cn[i] = ln.rc[12 * (r + 2) + i];
    }
    round(r >= POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS, true, cf);
  }
  const u64 zero[3] = {0, 0, 0};
  round(true, false, zero);
}

}  // namespace poseidon_quad
