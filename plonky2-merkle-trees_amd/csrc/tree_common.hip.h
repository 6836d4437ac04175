// tree_common.hip.h -- device helpers shared by p2mt_hash.hip and p2mt_mmr.hip: HashOut load/store, post-order
// position, the re-loadable permutation wrappers of the fast path, the sponge, and the 12-lane / 4-lane two_to_one.
#pragma once
#include "poseidon_quad.hip.h"
#include "runtime.h"

namespace p2mt_dev {

using gl::u32;
using gl::u64;
using p2mt::PermCtx;
using p2mt::BatchArg;

constexpr int kBlock = 256;

// batched pipelines (runtime.h BatchCtx): a pointer into block 0 moves to the block of this workgroup's proof (grid z)
template <typename T>
GL_DEV T* bp_at(T* p, const BatchArg& ba, unsigned proof) {
  return ((uint64_t)p - ba.base) < ba.span ? reinterpret_cast<T*>((uint64_t)p + (uint64_t)proof * ba.stride) : p;
}
template <typename T>
GL_DEV T* bp(T* p, const BatchArg& ba) {
  return bp_at(p, ba, blockIdx.z);
}

GL_DEV void load_hash(const u64* p, u64 (&h)[4]) {
  const ulonglong2* q = reinterpret_cast<const ulonglong2*>(p);
  const ulonglong2 a = q[0], b = q[1];
  h[0] = a.x; h[1] = a.y; h[2] = b.x; h[3] = b.y;
}
GL_DEV void store_hash(u64* p, const u64 (&h)[4]) {
  ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
  q[0] = make_ulonglong2(h[0], h[1]);
  q[1] = make_ulonglong2(h[2], h[3]);
}

GL_DEV size_t node_pos(size_t last_leaf, unsigned h) { return 2 * last_leaf - (size_t)__popcll(last_leaf) + h; }

constexpr int IMPL_FAST = 2;  // M == 2: poseidon_fast with exact fallback; M in {0,1}: exact variants of poseidon.hip.h

// One permutation whose input can be re-materialised: `load` fills the state (it is called again if the fast
// path raised its sticky flag, so that no copy of the input has to stay live in VGPRs).
template <int M, int PR, typename Load>
GL_DEV void permute_reloadable(u64 (&s)[12], const PermCtx& ctx, Load&& load, const poseidon_fast::MfmaCtx* mc = nullptr) {
  load(s);
  if constexpr (M == IMPL_FAST) {  // PR == 1: sparse partial rounds; PR == 5: dense MDS layers on the matrix pipe (`mc`, every lane of the wave active)
    const u64 sticky = poseidon_fast::permute<false, 12, false, false, PR == 1, (PR == 5 ? 3 : 0), (PR == 5 ? 2 : PR == 0), 0, false, -1, PR == 5>(s, ctx.rc, mc) | ctx.force_fallback;
    if (__builtin_expect(sticky != 0, 0)) {
      load(s);
      poseidon::permute<poseidon::MDS_MAD64, poseidon::PARTIAL_NAIVE>(s);
    }
  } else {
    poseidon::permute<M, PR>(s);
  }
}

// two_to_one(l, r) with l/r produced by `load_lr`.  LEAF_PAIR: l and r are leaf digests [leaf, 0, 0, 0] (fast path only).
// PR (fast path): 0 = dense partial rounds, 1 = sparse partial rounds, 2 = dense with the MDS layers on the matrix pipe (`mc`),
// 3 = the same for the MDS layers of the partial rounds only, 4 = dense with one MDS layer per partial round (round 2's form; 0 batches
// three partial rounds per MDS application, poseidon_fast::partial_rounds3), 5 = the MDS layers of the full rounds as one 32x32x32
// MFMA per 8-bit limb (poseidon_fast::mds_layer_mfma32) and the partial rounds in groups of four (poseidon_fast::partial_rounds_g),
// 7 = 5 with the partial rounds in groups of three (round 3's default), 8 = 5 with flag-form folds in the MDS layers.
// XF: exact folds in the MDS layers (poseidon_fast::permute_impl) -- what PR == 5, the default, does anyway since round 4: a wave that
// redoes a hash finishes ~60-200 us after its neighbours, and the last such wave of a launch sets its duration (stage 1: -1.8 % with
// 1.6 % MORE instructions; the first level launch above it: 210 -> 167 us).  PR == 8 is PR == 5 with the flag-form folds (A/B).
template <int M, int PR, bool LEAF_PAIR = false, bool XF = false, typename LoadLR>
GL_DEV void two_to_one_r(const PermCtx& ctx, u64 (&o)[4], LoadLR&& load_lr, const poseidon_fast::MfmaCtx* mc = nullptr) {
  u64 s[12];
  auto load = [&](u64 (&st)[12]) {
    u64 l[4], r[4];
    load_lr(l, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      st[k] = l[k];
      st[4 + k] = r[k];
      st[8 + k] = 0;
    }
  };
  if constexpr (M == IMPL_FAST) {  // capacity words are zero and only 4 output words are needed
    load(s);
    const u64 sticky = poseidon_fast::permute<true, 4, false, LEAF_PAIR, PR == 1, (PR == 2 || PR == 3 ? PR - 1 : (PR == 5 || PR == 7 || PR == 8 ? 3 : 0)), (PR == 5 || PR == 8 ? 2 : PR == 0 || PR == 6 || PR == 7), (PR == 6 ? 1 : 0), false, -1, XF || PR == 5>(s, ctx.rc, mc) | ctx.force_fallback;
    if (__builtin_expect(sticky != 0, 0)) {  // ~0.5 % of waves.  (Redoing with the exact fast-form instead was
      load(s);                               //  measured 2 % slower overall: bigger kernel, worse allocation.)
      poseidon::permute<poseidon::MDS_MAD64, poseidon::PARTIAL_NAIVE>(s);
    }
  } else {
    permute_reloadable<M, PR>(s, ctx, load);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = gl::canon(s[k]);
}


// hash_or_noop / hash_no_pad of `len`-element rows: overwrite-mode sponge, rate 8, one row per lane.
// `get(k)` returns element k of the row.  With the fast variant the whole row is redone exactly if any of its
// permutations raised the sticky flag.
template <int M, int PR, typename Get>
GL_DEV void sponge(size_t len, const PermCtx& ctx, u64 (&o)[4], Get&& get) {
  u64 s[12];
  auto run = [&](auto fast) -> u64 {
    u64 sticky = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) s[k] = 0;
#pragma unroll 1
    for (size_t off = 0; off < len; off += 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (off + k < len) s[k] = get(off + k);
      // (exact folds unless the sparse A/B form is selected: the flag is collected over the whole row and a flagged wave redoes all of it)
      if constexpr (decltype(fast)::value) sticky |= poseidon_fast::permute<false, 12, false, false, PR == 1, 0, (PR == 1 ? 0 : 1), 0, false, -1, PR != 1>(s, ctx.rc);
      else if constexpr (M == IMPL_FAST) poseidon::permute<poseidon::MDS_MAD64, poseidon::PARTIAL_NAIVE>(s);
      else poseidon::permute<M, PR>(s);
    }
    return sticky;
  };
  if constexpr (M == IMPL_FAST) {
    const u64 sticky = run(std::true_type{}) | ctx.force_fallback;
    if (__builtin_expect(sticky != 0, 0)) run(std::false_type{});
  } else {
    run(std::false_type{});
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = gl::canon(s[k]);
}


// ---------------------------------------------------------------- one wavefront per node (latency path)
// Near the top of the tree a level has fewer nodes than the chip has lanes, and a lane-per-hash launch costs one
// full single-hash latency (~65 us) whatever its size.  Here 12 lanes of a wave share ONE permutation: lane i owns
// state word i, the twelve S-boxes of a full round run in parallel, and the MDS row of lane r is two mad chains
// over the words broadcast with v_readlane (SGPR operands) against that lane's row of constants.  All 30 rounds
// are unrolled with the lane's round constants preloaded, so a node takes ~3.5k instructions instead of ~28k.
// Used for levels of <= 2^12 nodes; bit-identical to the lane-per-hash kernels.  The flag-form primitives are tried first
// (~10 % fewer instructions per round); a wave whose sticky flag is set (probability ~4e-4 per permutation) redoes THIS
// permutation with the exact form on the same 12 lanes -- one more 11 us, not the 60 us serial redo of the lane-per-hash
// layout that once set the duration of whole latency-bound launches.
// One permutation by the calling wave (all 64 lanes must call it): lane w < 12 passes state word w (any u64) and
// receives word w of the permuted state (loose u64, exact); lanes >= 12 shadow lane 0 and their result is unused.
// `hook(r, x)` sees this lane's state word at the start of round r, after the round constant and before the S-box --
// the value plonky2's PoseidonGate keeps as a witness wire (p2mt_circuit.hip records it; the hashing kernels pass nothing).
// Copy the 360 round constants into LDS (all threads of the workgroup must call it) and return a context that reads them
// from there: a wave-permutation fetches one constant per round and lane, and an LDS round trip hides under one round
// where an L2 one does not (measured: ~2 us of exposed load latency per 10.5 us permutation).
constexpr int kWaveRcWords = 360 + poseidon_fast::kP3WaveWords;  // size of the LDS array behind stage_round_constants()
GL_DEV PermCtx stage_round_constants(u64* lds /*[kWaveRcWords]*/, const PermCtx& ctx) {
  for (unsigned k = threadIdx.x; k < 360; k += blockDim.x) lds[k] = ctx.rc[k];
  for (unsigned k = threadIdx.x; k < (unsigned)poseidon_fast::kP3WaveWords; k += blockDim.x) lds[360 + k] = ctx.w3[k];
  __syncthreads();
  return PermCtx{lds, ctx.force_fallback, lds + 360};
}

// lane C of this lane's 16-lane row, as ONE instruction (v_mov_b64_dpp row_newbcast:C).  The twelve state words live in lanes 0..11 of
// row 0; the other rows compute on their own lanes' values and nobody reads them (round 4: two v_readlane per word before -- 24 of the
// ~100 instructions of a round).
template <int C>
GL_DEV u64 row_bcast64(u64 v) {
  return (u64)__builtin_amdgcn_update_dpp((long long)0, (long long)v, 0x150 + C, 0xf, 0xf, true);
}
// ROWS4: every 16-lane row of the wavefront holds a state of its own in its lanes 0..11 -- four permutations per call.  Everything
// below is row-local already (row_newbcast, row_shl), so the only difference is which lane index picks the MDS row and the constants.
template <bool EXACT, bool ROWS4 = false, typename Hook>
GL_DEV u64 permute_wave_impl(u64 x, const PermCtx& ctx, Hook&& hook, u64& sticky) {
  const unsigned lane = ROWS4 ? (threadIdx.x & 15) : (threadIdx.x & 63);
  const unsigned w = lane < 12 ? lane : 0;
  u32 kk[12];  // this lane's MDS row: MDS[w][c] = CIRC[(c - w) mod 12] (+8 at [0][0]), picked out of two packed
               // immediates (bytes 17,15,41,16,2,28,13,13 | 39,18,34,20) instead of a per-lane table load
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const unsigned idx = (unsigned)c >= w ? (unsigned)c - w : (unsigned)c + 12 - w;
    const u64 word = idx < 8 ? 0x0D0D1C0210290F11ull : 0x14221227ull;
    kk[c] = (u32)((word >> (8 * (idx & 7))) & 0xFF) + ((w == 0 && c == 0) ? 8u : 0u);
  }

  // Batched partial rounds (round 3; poseidon_fast::partial_rounds3 has the algebra): rounds 4..24 run as 7 groups of three with
  // ONE broadcast-and-accumulate step each -- lane 0 accumulates row 0 of M y (the next S-box input), lane 1 row 0 of M^2 y, lanes
  // 2..13 the rows of M^3 y, all in the same 24 mads; the two later S-box inputs are broadcast, so every lane holds d1 / d2 and adds
  // its own multiple of them; the result slides from lanes 2..13 to lanes 0..11.  Per-lane rows and addends come from ctx.w3.
  const unsigned L = lane < 14 ? lane : 0;
  const u32* __restrict__ w3 = reinterpret_cast<const u32*>(ctx.w3 + 14 * poseidon_fast::kP3Groups) + 14 * L;
  u32 k3[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) k3[c] = w3[c];
  const u32 cf1 = w3[12], cf2 = w3[13];

  const u64* rcw = ctx.rc + w;     // this lane's column of the round-constant table
  u64 c_next = rcw[12];            // constant of round r+1, fetched one round ahead (hidden under the S-box)
  x = gl::add_c(x, rcw[0]);
  auto sbox = [&](u64 v) -> u64 {
    if constexpr (EXACT) return poseidon_fast::exact::pow7(v);
    else return poseidon_fast::pow7(v, sticky);
  };
  // one round: S-box (every lane in a full round, lane 0 in a partial one), then this lane's MDS row with the
  // next round's constant folded into the two mad chains
  auto round = [&](int r, bool full, bool add, u64 c_fold) {
    hook(r, x);
    const u64 y = sbox(x);
    if (full || lane == 0) x = y;
    u64 al = add ? (u64)(u32)c_fold : 0, ah = add ? (u64)(u32)(c_fold >> 32) : 0;
    poseidon::static_for<0, 12>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const u64 sc = row_bcast64<c>(x);
      al += (u64)(u32)sc * kk[c];
      ah += (u64)(u32)(sc >> 32) * kk[c];
    });
    ah = poseidon_fast::add32((u32)(al >> 32), ah);
    const u64 val = ((u64)(u32)ah << 32) | (u32)al;
    if constexpr (EXACT) {
      x = poseidon_fast::exact::fold96((u32)(ah >> 32), val);
    } else {
      u64 cm;
      x = poseidon_fast::mad_eps_carry((u32)(ah >> 32), val, cm);  // top * EPS + val; wraps with probability ~2^-22
      sticky |= cm;
    }
  };
  auto fold = [](u64 al, u64 ah) -> u64 {  // (al + ah 2^32) mod p, loose; exact form (the top word reaches 2^26 here)
    ah = poseidon_fast::add32((u32)(al >> 32), ah);
    return poseidon_fast::exact::fold96((u32)(ah >> 32), ((u64)(u32)ah << 32) | (u32)al);
  };

  auto group = [&](int g) {  // rounds k = 4 + 3g, k + 1, k + 2
    const int k = POSEIDON_HALF_FULL_ROUNDS + 3 * g;
    const u64 init = ctx.w3[14 * g + L];
    hook(k, x);
    const u64 y = sbox(x);
    if (lane == 0) x = y;
    u64 al = (u64)(u32)init, ah = (u64)(u32)(init >> 32);
    poseidon::static_for<0, 12>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const u64 sc = row_bcast64<c>(x);
      al += (u64)(u32)sc * k3[c];
      ah += (u64)(u32)(sc >> 32) * k3[c];
    });
    const u64 x1 = row_bcast64<0>(fold(al, ah));  // v1[0]
    hook(k + 1, x1);
    const u64 d1 = poseidon_fast::sub_any(sbox(x1), x1);
    al += (u64)(u32)d1 * cf1;
    ah += (u64)(u32)(d1 >> 32) * cf1;
    const u64 x2 = row_bcast64<1>(fold(al, ah));  // v2[0]
    hook(k + 2, x2);
    const u64 d2 = poseidon_fast::sub_any(sbox(x2), x2);
    al += (u64)(u32)d2 * cf2;
    ah += (u64)(u32)(d2 >> 32) * cf2;
    const u64 v3 = fold(al, ah);            // lanes 2..13: words 0..11 of the state in front of round k + 3
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v3, 0x102, 0xF, 0xF, true);          // row_shl:2: lane i <- lane i + 2
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v3 >> 32), 0x102, 0xF, 0xF, true);
    x = ((u64)hi << 32) | lo;
  };
#pragma unroll 1
  for (int r = 0; r < POSEIDON_HALF_FULL_ROUNDS; ++r) {
    const u64 c_fold = c_next;
    c_next = rcw[12 * (r + 2)];
    round(r, true, true, c_fold);
  }
  static_assert(POSEIDON_PARTIAL_ROUNDS == 3 * poseidon_fast::kP3Groups + 1, "7 groups of three + one round");
#pragma unroll 1
  for (int g = 0; g < poseidon_fast::kP3Groups; ++g) group(g);
  c_next = rcw[12 * (POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS)];
#pragma unroll 1
  for (int r = POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS - 1; r < POSEIDON_ROUNDS - 1; ++r) {
    const u64 c_fold = c_next;
    if (r + 2 < POSEIDON_ROUNDS) c_next = rcw[12 * (r + 2)];
    round(r, r >= POSEIDON_HALF_FULL_ROUNDS + POSEIDON_PARTIAL_ROUNDS, true, c_fold);
  }
  round(POSEIDON_ROUNDS - 1, true, false, 0);
  return x;
}
template <typename Hook>
GL_DEV u64 permute_wave_hook(u64 x, const PermCtx& ctx, Hook&& hook) {
  u64 sticky = ctx.force_fallback;  // lane masks OR-ed on the scalar pipe: wave-uniform
  const u64 y = permute_wave_impl<false>(x, ctx, hook, sticky);
  // (the flag of row 0 only: the other rows of the wave run the same instructions on values nobody reads)
  if (__builtin_expect((sticky & 0xFFFFull) != 0, 0)) return permute_wave_impl<true>(x, ctx, hook, sticky);
  return y;
}
GL_DEV u64 permute_wave(u64 x, const PermCtx& ctx) {
  return permute_wave_hook(x, ctx, [](int, u64) {});
}
// four states per wavefront (one per 16-lane row): same instructions, same latency, a quarter of the wavefronts -- a tree level of a
// few thousand nodes then runs at one wavefront per SIMD instead of four sharing it
GL_DEV u64 permute_wave4(u64 x, const PermCtx& ctx) {
  auto nohook = [](int, u64) {};
  u64 sticky = ctx.force_fallback;
  const u64 y = permute_wave_impl<false, true>(x, ctx, nohook, sticky);
  if (__builtin_expect(sticky != 0, 0)) return permute_wave_impl<true, true>(x, ctx, nohook, sticky);
  return y;
}
// out[4 j ..) = two_to_one(in[8 j .. 8 j + 4), in[8 j + 4 .. 8 j + 8)) for the four nodes j = j0 + row of the calling wave (all 64 lanes
// call it; rows whose node is >= n_out compute on zeros and store nothing)
GL_DEV void two_to_one_wave4(const u64* __restrict__ in, u64* __restrict__ out, size_t j0, size_t n_out, const PermCtx& ctx) {
  const unsigned rl = threadIdx.x & 15, row = (threadIdx.x >> 4) & 3;
  const size_t j = j0 + row;
  u64 x = (rl < 8 && j < n_out) ? in[8 * j + rl] : 0;
  x = permute_wave4(x, ctx);
  if (rl < 4 && j < n_out) out[4 * j + rl] = gl::canon(x);
}

// out[0..4) = two_to_one(lp[0..4), rp[0..4)) computed by the calling wave (all 64 lanes must call it)
GL_DEV void two_to_one_wave(const u64* __restrict__ lp, const u64* __restrict__ rp, u64* __restrict__ out,
                            const PermCtx& ctx) {
  const unsigned lane = threadIdx.x & 63;
  u64 x = lane < 4 ? lp[lane] : (lane < 8 ? rp[lane - 4] : 0);
  x = permute_wave(x, ctx);
  if (lane < 4) out[lane] = gl::canon(x);
}


// ---------------------------------------------------------------- four lanes per node (poseidon_quad.hip.h)
// out[0..4) = two_to_one(lp, rp) computed by the calling quad (all four lanes call it with the same pointers)
GL_DEV void two_to_one_quad(const u64* __restrict__ lp, const u64* __restrict__ rp, u64* __restrict__ out,
                            const poseidon_quad::Lane& ln) {
  u64 x[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const u32 w = 3 * ln.q + i;
    x[i] = w < 4 ? lp[w] : (w < 8 ? rp[w - 4] : 0);
  }
  poseidon_quad::permute(x, ln);
  if (ln.q == 0) {
    out[0] = gl::canon(x[0]);
    out[1] = gl::canon(x[1]);
    out[2] = gl::canon(x[2]);
  } else if (ln.q == 1) {
    out[3] = gl::canon(x[0]);
  }
}


inline unsigned grid_for(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
// grid with the proofs of a batch in z
inline dim3 bgrid(unsigned x, unsigned y = 1) { return dim3(x, y, p2mt::batch().B); }
inline BatchArg barg() { return p2mt::batch().arg; }

}  // namespace p2mt_dev
