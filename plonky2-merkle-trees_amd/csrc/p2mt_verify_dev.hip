// p2mt_verify_dev.hip -- the field arithmetic of CircuitData::verify on the device (round 4; until round 3 this ran on host
// threads with a 128-bit division per multiply, p2mt_verify_host.hip, and cost as much as a whole prove).
//
// Replaces what `circuit_data.verify(proof)` (/root/reference/src/mmr/mmr_plonky2_verifier.rs:150,
// mmr_plonky2_verifier_1_recursion.rs:193,220) runs inside plonky2 (git rev 3b21b87d, not in the reference tree; parity
// unpinned): plonk/verifier.rs verify_with_challenges -- plonk/vanishing_poly.rs eval_vanishing_poly with every gate's
// eval_unfiltered over the quadratic extension at zeta, against Z_H(zeta) * the quotient openings -- and the field side of
// fri/verifier.rs (PrecomputedReducedOpenings, fri_combine_initial, compute_evaluation, the final polynomial).  The transcript and
// the Merkle paths were on the device already (p2mt_circuit.hip); with these three kernels a verification is launches only, one
// small copy back, and the proofs of a batch ride in grid z:
//   verify_item()      the query's leaf index from the transcript's challenge and the (row, path, cap) triple of one (query, tree)
//                      pair, worked out by the wavefront that folds the path (the host used to build them between two synchronisations);
//   k_verify_fri       16 lanes per query: fri_combine_initial as 16 interleaved Horner sums in alpha^16, the two denominators'
//                      inverses on two lanes, every reduction layer's 2^arity-point interpolation as ONE barycentric term per lane
//                      (the points are a coset c<g> of the 2^arity-th roots of unity, so prod_{j != i}(p_i - p_j) = arity p_i^(arity-1)
//                      and prod_j(beta - p_j) = beta^arity - c^arity: one base-field inversion per layer instead of arity),
//                      the final polynomial as 16 interleaved Horner sums; first failing (query, check) by atomicMin;
//   k_verify_openings  one workgroup per proof, one role per wavefront: the PoseidonGate constraints replayed on 12 lanes (one
//                      per sponge word, MDS through LDS), the permutation argument on one lane per (challenge, chunk), the
//                      small gates one lane per constraint, each gate type of the in-circuit verifier (gates_recursion.hip.h, the
//                      source the prover's quotient kernel instantiates over the base field) on its own wavefront; then the
//                      filtered sum per constraint index, and sum_t alpha^t term_t as lane-interleaved Horner sums.
// Every value is an exact field element, so the verdicts and reason codes are those of the host code they replace
// (tests/test_circuit_gpu.py, test_recursion_gpu.py, test_reference_exhaustive_gpu.py run both on every case).
#include "tree_common.hip.h"
#include "circuit_types.h"
#include "gates_recursion.hip.h"

#include <new>

using namespace p2mt_dev;
using p2mt::VerifyDevArgs;

namespace {

// ---------------------------------------------------------------- F[X]/(X^2 - 7), loose u64 components, exact operations
struct E {
  u64 a, b;
};
GL_DEV E e_of(u64 a) { return E{a, 0}; }
GL_DEV E e_add(E x, E y) { return E{gl::add(x.a, y.a), gl::add(x.b, y.b)}; }
GL_DEV E e_sub(E x, E y) { return E{poseidon_fast::sub_any(x.a, y.a), poseidon_fast::sub_any(x.b, y.b)}; }
GL_DEV E e_mul(E x, E y) {
  const u64 bb7 = gl::mul(gl::mul(x.b, y.b), 7);
  return E{gl::mul_add(x.a, y.a, bb7), gl::mul_add(x.a, y.b, gl::mul(x.b, y.a))};
}
GL_DEV E e_scale(E x, u64 s) { return E{gl::mul(x.a, s), gl::mul(x.b, s)}; }
GL_DEV E e_canon(E x) { return E{gl::canon(x.a), gl::canon(x.b)}; }
GL_DEV bool e_eq(E x, E y) { return gl::canon(x.a) == gl::canon(y.a) && gl::canon(x.b) == gl::canon(y.b); }
GL_DEV bool e_is_zero(E x) { return gl::canon(x.a) == 0 && gl::canon(x.b) == 0; }
GL_DEV u64 f_inv(u64 x) { return gl::pow(x, gl::P - 2); }  // 0 -> 0
GL_DEV E e_inv(E x) {
  const u64 ni = f_inv(poseidon_fast::sub_any(gl::mul(x.a, x.a), gl::mul(gl::mul(x.b, x.b), 7)));
  return E{gl::mul(x.a, ni), gl::mul(poseidon_fast::sub_any(0, x.b), ni)};
}
GL_DEV E e_pow(E x, u64 e) {
  E r{1, 0};
  while (e) {
    if (e & 1) r = e_mul(r, x);
    x = e_mul(x, x);
    e >>= 1;
  }
  return r;
}
GL_DEV E e_at(const u64* w, size_t i) { return E{w[2 * i], w[2 * i + 1]}; }
GL_DEV u32 brev_bits(u32 x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }
// sum over the 16 lanes of a group (lanes g*16 .. g*16+15 of the wave)
GL_DEV E group16_sum(E v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    const u64 oa = __shfl_xor((unsigned long long)v.a, m, 16), ob = __shfl_xor((unsigned long long)v.b, m, 16);
    v = E{gl::add(v.a, oa), gl::add(v.b, ob)};
  }
  return v;
}
GL_DEV E group16_bcast(E v, int src) {
  return E{(u64)__shfl((unsigned long long)v.a, src, 16), (u64)__shfl((unsigned long long)v.b, src, 16)};
}

// ---------------------------------------------------------------- the (row, path, cap) triples
// mirrors the item construction of verify_pass (p2mt_circuit.hip, rounds 1-3): per query 4 oracle rows, then one coset per layer
struct VItem {
  u32 leaf_off, width, index, sib_off, n_sib, cap_off;
};
// the triple of ONE (query, tree) pair, computed by the wavefront that folds its path (until the end of round 4 a k_verify_items launch
// sat between the end of the transcript and the path folds: 21 us of launch and dependent-load latency for 28 lanes of index arithmetic)
GL_DEV VItem verify_item(const u64* __restrict__ dv, const VerifyDevArgs& a, u32 q, u32 tr) {
  const u32 log_big = a.fri.degree_bits + a.fri.rate_bits, cap_h = a.fri.cap_height, nred = a.fri.num_reductions;
  u32 x_index = (u32)(dv[a.o_out + 27 + q] & (((u64)1 << log_big) - 1));
  u32 w = a.o_proof + a.off_fri + nred * 64 + q * a.query_words;
  VItem r{0, 0, 0, 0, 0, 0};
  u32 log_sz = log_big;
  for (u32 k = 0; k <= tr; ++k) {
    if (k < 4) {
      const u32 np = (u32)a.n_polys[k], n_sib = log_big - cap_h;
      const u32 cap_off = k == 0 ? a.o_cscap : a.o_proof + 64 * (k - 1);
      if (k == tr) r = VItem{w, np, x_index, w + np, n_sib, cap_off};
      w += np + 4 * n_sib;
    } else {
      const u32 ab = a.fri.reduction_arity_bits[k - 4], arity = 1u << ab;
      x_index >>= ab;
      const u32 n_sib = log_sz - ab - cap_h;
      if (k == tr) r = VItem{w, 2 * arity, x_index, w + 2 * arity, n_sib, a.o_proof + a.off_fri + 64 * (k - 4)};
      w += 2 * arity + 4 * n_sib;
      log_sz -= ab;
    }
  }
  return r;
}

// ---------------------------------------------------------------- Merkle paths in two halves
// verify_merkle_proof_to_cap of every (query, tree) pair.  The digest of an opened row does not depend on the query's index -- only on
// where the row sits in the proof -- so the sponges (up to 17 chained permutations for a 135-wire row) run from the moment the proof
// is on the device, beside the transcript; what has to wait for the indices is the fold along the path (one permutation per level).
// One wavefront per pair on the 12-lanes-per-permutation layout, as k_verify_merkle (p2mt_circuit.hip, rounds 1-3) did both at once.
constexpr int kMerkleBlock = 256;
__global__ __launch_bounds__(kMerkleBlock) void k_verify_leaf_digests(const u64* __restrict__ dv, u64* __restrict__ digests, VerifyDevArgs a,
                                                                      BatchArg ba, PermCtx ctx) {
  dv = bp(dv, ba);
  digests = bp(digests, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const u32 nred = a.fri.num_reductions, per_q = 4 + nred, n_items = a.fri.num_query_rounds * per_q;
  const u32 item = blockIdx.x * (kMerkleBlock / 64) + (threadIdx.x >> 6);
  if (item >= n_items) return;  // wave-uniform
  const u32 lane = threadIdx.x & 63, q = item / per_q, tr = item % per_q;
  const u32 log_big = a.fri.degree_bits + a.fri.rate_bits, cap_h = a.fri.cap_height;
  u32 w = a.o_proof + a.off_fri + nred * 64 + q * a.query_words, width = 0;
  {
    u32 log_sz = log_big;
    for (u32 k = 0; k <= tr; ++k) {
      u32 wd, n_sib;
      if (k < 4) {
        wd = (u32)a.n_polys[k];
        n_sib = log_big - cap_h;
      } else {
        const u32 ab = a.fri.reduction_arity_bits[k - 4];
        wd = 2u << ab;
        n_sib = log_sz - ab - cap_h;
        log_sz -= ab;
      }
      if (k == tr) width = wd;
      else w += wd + 4 * n_sib;
    }
  }
  u64 x = 0;
  if (width <= 4) {
    if (lane < width) x = gl::canon(dv[w + lane]);
  } else {
#pragma unroll 1
    for (u32 off = 0; off < width; off += 8) {
      if (lane < 8 && off + lane < width) x = dv[w + off + lane];
      x = permute_wave(x, ctx);
    }
    x = gl::canon(x);
  }
  if (lane < 4) digests[4 * (size_t)item + lane] = x;
}

__global__ __launch_bounds__(kMerkleBlock) void k_verify_paths(const u64* __restrict__ dv, VerifyDevArgs a,
                                                               const u64* __restrict__ digests, u32 n_items, int* bad, BatchArg ba, PermCtx ctx) {
  dv = bp(dv, ba);
  digests = bp(digests, ba);
  bad = bp(bad, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const u32 item = blockIdx.x * (kMerkleBlock / 64) + (threadIdx.x >> 6);
  if (item >= n_items) return;  // wave-uniform
  const u32 lane = threadIdx.x & 63;
  const u32 per_q = 4 + a.fri.num_reductions;
  const VItem it = verify_item(dv, a, item / per_q, item % per_q);
  u64 x = lane < 4 ? digests[4 * (size_t)item + lane] : 0;
  u32 index = it.index;
  u64 sib_next = (lane < 8 && it.n_sib) ? dv[it.sib_off + (lane & 3)] : 0;  // the next sibling is fetched under the current permutation
#pragma unroll 1
  for (u32 s = 0; s < it.n_sib; ++s, index >>= 1) {
    const u64 up = __shfl_up((unsigned long long)x, 4);  // lanes 4..7 see the current digest
    const u64 sib = sib_next;
    if (lane < 8 && s + 1 < it.n_sib) sib_next = dv[it.sib_off + 4 * (s + 1) + (lane & 3)];
    const bool sib_left = index & 1;
    u64 y = 0;
    if (lane < 4) y = sib_left ? sib : x;
    else if (lane < 8) y = sib_left ? up : sib;
    x = gl::canon(permute_wave(y, ctx));
  }
  const bool mismatch = lane < 4 && x != dv[it.cap_off + 4 * index + lane];
  if (__any(mismatch) && lane == 0) atomicMin(bad, (int)item + 1);
}

// ---------------------------------------------------------------- k_verify_fri
constexpr int kFriBlock = 256;  // 16 queries per workgroup, 16 lanes each

__global__ __launch_bounds__(kFriBlock) void k_verify_fri(const u64* __restrict__ dv, int* __restrict__ res, VerifyDevArgs a, BatchArg ba) {
  dv = bp(dv, ba);
  res = bp(res, ba);
  __shared__ E s_red[kFriBlock];
  __shared__ E s_apow[17];   // alpha^0 .. alpha^16
  __shared__ E s_reduced[2];
  __shared__ E s_alpha_next;
  const u32 t = threadIdx.x, lane = t & 15, grp = t >> 4;
  const u64* out = dv + a.o_out;
  const u64* fo = dv + a.o_fo;  // openings in transcript order: batch 0 = every polynomial at zeta, then batch 1 at g zeta
  const E alpha = e_at(out, 4), zeta = E{out[6], out[7]};
  const u32 n_all = (u32)(a.n_polys[0] + a.n_polys[1] + a.n_polys[2] + a.n_polys[3]), n_next = a.d.num_challenges;
  // ---- PrecomputedReducedOpenings + the powers of alpha the queries need
  if (t < 17) s_apow[t] = e_pow(alpha, t);
  {
    E acc = e_of(0);
    for (u32 j = t; j < n_all; j += kFriBlock) acc = e_add(acc, e_mul(e_at(fo, j), e_pow(alpha, j)));
    s_red[t] = acc;
  }
  __syncthreads();
  for (int m = kFriBlock / 2; m >= 1; m >>= 1) {
    if (t < (u32)m) s_red[t] = e_add(s_red[t], s_red[t + m]);
    __syncthreads();
  }
  if (t == 0) {
    s_reduced[0] = s_red[0];
    E acc = e_of(0);
    for (u32 j = n_next; j-- > 0;) acc = e_add(e_mul(acc, alpha), e_at(fo, n_all + j));
    s_reduced[1] = acc;
    s_alpha_next = e_pow(alpha, n_next);
  }
  __syncthreads();
  // ---- one query per 16 lanes
  const u32 log_big = a.fri.degree_bits + a.fri.rate_bits, cap_h = a.fri.cap_height, nred = a.fri.num_reductions;
  const u32 q = blockIdx.x * (kFriBlock / 16) + grp;
  const bool live = q < a.fri.num_query_rounds;
  const u32 qq = live ? q : 0;  // dead groups recompute query 0 (the group shuffles want all lanes) and report nothing
  u32 x_index = (u32)(out[27 + qq] & (((u64)1 << log_big) - 1));
  const u64* w = dv + a.o_proof + a.off_fri + nred * 64 + (size_t)qq * a.query_words;
  const u64* leaf_of[4];
  u32 start[5];
  start[0] = 0;
  for (u32 o = 0; o < 4; ++o) {
    leaf_of[o] = w;
    start[o + 1] = start[o] + (u32)a.n_polys[o];
    w += a.n_polys[o] + 4 * (log_big - cap_h);
  }
  u64 subgroup_x = gl::mul(7, gl::pow(a.w_big, brev_bits(x_index, log_big)));
  const E a16 = s_apow[16];
  int reason = 0;
  E old_eval;
  {
    // fri_combine_initial: sum_pos leaf(pos) alpha^pos, pos = lane + 16 m  ->  alpha^lane * Horner in alpha^16
    E acc = e_of(0);
    const u32 last_m = (n_all - 1) / 16;
    for (u32 m = last_m + 1; m-- > 0;) {
      const u32 pos = lane + 16 * m;
      u64 v = 0;
      if (pos < n_all) {
        const u32 o = pos < start[1] ? 0 : pos < start[2] ? 1 : pos < start[3] ? 2 : 3;
        v = leaf_of[o][pos - start[o]];
      }
      acc = e_mul(acc, a16);
      acc.a = gl::add(acc.a, v);
    }
    const E acc0 = group16_sum(e_mul(acc, s_apow[lane]));
    E t1 = e_of(0);
    if (lane < n_next) t1 = e_scale(s_apow[lane], leaf_of[2][lane]);
    const E acc1 = group16_sum(t1);
    // the two denominators: lane parity picks which one a lane inverts (all lanes run the same instruction stream)
    const E gzeta = e_scale(zeta, a.w_n);
    const E den = e_sub(e_of(subgroup_x), (lane & 1) ? gzeta : zeta);
    const E inv = e_inv(den);
    const E inv0 = group16_bcast(inv, 0), inv1 = group16_bcast(inv, 1);
    E sum = e_mul(e_sub(acc0, s_reduced[0]), inv0);
    sum = e_add(e_mul(sum, s_alpha_next), e_mul(e_sub(acc1, s_reduced[1]), inv1));
    old_eval = e_scale(sum, subgroup_x);
  }
  u32 log_sz = log_big;
  for (u32 l = 0; l < nred; ++l) {
    const u32 ab = a.fri.reduction_arity_bits[l], arity = 1u << ab;
    const u32 within = x_index & (arity - 1);
    if (reason == 0 && !e_eq(e_at(w, within), old_eval)) reason = 3;
    // compute_evaluation at beta_l: points p_i = subgroup_x g^(i - brev(within)), values w[brev(i)]
    const E beta = e_at(out, 5 + l);
    const u32 li = lane & (arity - 1);  // lanes beyond the arity repeat a point and contribute nothing
    const u32 e = (li + arity - brev_bits(within, ab)) & (arity - 1);
    const u64 p_i = gl::mul(subgroup_x, gl::pow(a.w16, (u64)e << (4 - ab)));  // g = w16^(16 / arity)
    const E y_i = e_at(w, brev_bits(li, ab));
    const E d_i = e_sub(beta, e_of(p_i));
    const bool on_point = lane < arity && e_is_zero(d_i);
    E term = e_of(0);
    if (lane < arity) term = e_mul(e_scale(y_i, p_i), e_inv(d_i));
    const E s = group16_sum(term);
    u64 c_ar = subgroup_x;  // c^arity = subgroup_x^arity (g^arity = 1)
    for (u32 k = 0; k < ab; ++k) c_ar = gl::mul(c_ar, c_ar);
    E b_ar = beta;
    for (u32 k = 0; k < ab; ++k) b_ar = e_mul(b_ar, b_ar);
    const E z = e_sub(b_ar, e_of(c_ar));
    E ev = e_scale(e_mul(s, z), f_inv(gl::mul(arity, c_ar)));
    // beta on one of the points (a prover cannot aim for it, a forger might): the interpolant's value there is that point's value
    const unsigned long long hit = __ballot(on_point) >> ((threadIdx.x & 63) & ~15u) & 0xFFFFull;
    if (hit) ev = group16_bcast(y_i, __ffsll(hit) - 1);
    old_eval = ev;
    w += 2 * arity + 4 * (log_sz - ab - cap_h);
    subgroup_x = c_ar;
    x_index >>= ab;
    log_sz -= ab;
  }
  {
    // the final polynomial at subgroup_x: lane-interleaved Horner in x^16
    const u64* fin = dv + a.o_proof + a.off_final;
    const u32 final_len = a.final_len;
    u64 x16 = subgroup_x;
    for (int k = 0; k < 4; ++k) x16 = gl::mul(x16, x16);
    E acc = e_of(0);
    const u32 last_m = (final_len - 1) / 16;
    for (u32 m = last_m + 1; m-- > 0;) {
      const u32 i = lane + 16 * m;
      acc = e_scale(acc, x16);
      if (i < final_len) acc = e_add(acc, e_at(fin, i));
    }
    const E fe = group16_sum(e_scale(acc, gl::pow(subgroup_x, lane)));
    if (reason == 0 && !e_eq(fe, old_eval)) reason = 5;
  }
  if (live && lane == 0 && reason) atomicMin(res, (int)(q * 8 + reason));
  // the proof-of-work response next to the verdict words: one copy brings everything back (res points at the second int of its word)
  if (blockIdx.x == 0 && t == 0) reinterpret_cast<u64*>(res - 1)[1] = out[26];
}

// ---------------------------------------------------------------- k_verify_openings
struct FExtDev {
  typedef E T;
  GL_DEV static T add(T x, T y) { return e_add(x, y); }
  GL_DEV static T sub(T x, T y) { return e_sub(x, y); }
  GL_DEV static T mul(T x, T y) { return e_mul(x, y); }
  GL_DEV static T mulc(T x, u64 c) { return e_scale(x, c); }
  GL_DEV static T addc(T x, u64 c) { return E{gl::add(x.a, c), x.b}; }
  GL_DEV static T subc(T x, u64 c) { return E{poseidon_fast::sub_any(x.a, c), x.b}; }
  GL_DEV static T fromc(u64 c) { return e_of(c); }
};
GL_DEV E sbox7(E x) {
  const E x2 = e_mul(x, x), x4 = e_mul(x2, x2), x3 = e_mul(x2, x);
  return e_mul(x4, x3);
}

constexpr u32 kMaxTerms = 2 * (1 + 10) + 123;  // num_challenges (1 + num_chunks) + gate constraints, standard_recursion_config
constexpr int kOpenBlock = 1024;                // 16 wavefronts: 0 Poseidon, 1 permutation + L_0, 2 small gates, 3.. one per other gate

// phase 0: the whole check.  A single verification with the transcript on the host splits it: the gates' unfiltered constraints are
// functions of the OPENED VALUES alone (no challenge enters them), and replaying the PoseidonGate's 30 rounds over the extension
// field on twelve lanes is ~100 us of the ~110 -- phase 1 does that part as soon as the proof is on the device, beside the row
// sponges and the host's transcript, and leaves the constraints in `gcs`; phase 2, behind the challenges, filters and folds them
// with the powers of alpha, checks the permutation argument and compares with Z_H(zeta) q(zeta) (~10 us).
constexpr u32 kGcsWords = 2 * p2mt_cb::kMaxGateTypes * p2mt_cb::kNumGateConstraints + 2;  // constraints, then the "bad" word
__global__ __launch_bounds__(kOpenBlock) void k_verify_openings(const u64* __restrict__ dv, int* __restrict__ res_ok,
                                                                const u64* __restrict__ k_is, VerifyDevArgs a, BatchArg ba,
                                                                u64* __restrict__ gcs, int phase) {
  dv = bp(dv, ba);
  res_ok = bp(res_ok, ba);
  __shared__ E s_w[p2mt_cb::kNumWires];
  __shared__ E s_cs[p2mt_cb::kMaxGateTypes][p2mt_cb::kNumGateConstraints];  // unfiltered constraints per gate of the circuit
  __shared__ E s_terms[kMaxTerms];
  __shared__ E s_state[12];
  __shared__ E s_van[2][64];
  __shared__ int s_bad;
  const p2mt::VerifyDesc& d = a.d;
  const u32 t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const u32 nch = d.num_challenges, qf = d.quotient_degree_factor;
  const u32 num_chunks = (d.num_routed + qf - 1) / qf, num_prods = num_chunks - 1, n_consts = d.num_selectors + d.num_constants;
  const u64* out = dv + a.o_out;
  const u64* open = dv + a.o_proof + a.off_open;  // OpeningSet order: constants | sigmas | wires | zs | zs_next | pps | quotient
  const u64 *consts = open, *sigmas = consts + 2 * n_consts, *wires = sigmas + 2 * d.num_routed, *zs = wires + 2 * d.num_wires;
  const u64 *zs_next = zs + 2 * nch, *pps = zs_next + 2 * nch, *quot = pps + 2 * nch * num_prods;
  const u64 *betas = out, *gammas = out + nch, *alphas = out + 2 * nch;
  const u64* pi_hash = dv + 4;
  const E zeta{out[3 * nch], out[3 * nch + 1]};
  const u32 n_terms = nch * (1 + num_chunks) + p2mt_cb::kNumGateConstraints;
  for (u32 j = t; j < d.num_wires; j += kOpenBlock) s_w[j] = e_at(wires, j);
  for (u32 j = t; j < p2mt_cb::kMaxGateTypes * p2mt_cb::kNumGateConstraints; j += kOpenBlock)
    (&s_cs[0][0])[j] = phase == 2 ? E{gcs[2 * j], gcs[2 * j + 1]} : e_of(0);
  if (t == 0) s_bad = phase == 2 ? (int)gcs[kGcsWords - 2] : 0;
  __syncthreads();
  E zn = zeta;
  for (u32 i = 0; i < d.degree_bits; ++i) zn = e_mul(zn, zn);
  const E zh = e_sub(zn, e_of(1));
  const u64* gc = consts + 2 * d.num_selectors;
  // which gate of the circuit (index into d.kind) a wavefront evaluates: Poseidon on wave 0, every other non-trivial kind from wave 3
  int my_gate = -1;
  {
    u32 next_wave = 3;
    for (u32 g = 0; g < d.n_kinds; ++g) {
      const u32 k = d.kind[g];
      if (k == p2mt_cb::G_POSEIDON) {
        if (wave == 0) my_gate = (int)g;
      } else if (k >= p2mt_cb::G_BASE_SUM) {
        if (wave == next_wave) my_gate = (int)g;
        ++next_wave;
      }
    }
  }
  if (wave == 0) {
    if (my_gate >= 0 && phase != 2) {
      // PoseidonGate::eval_unfiltered: the permutation replayed from the opened wires, sponge word = lane (123 constraints)
      E* cs = s_cs[my_gate];
      const bool act = lane < 12;
      const u32 i = act ? lane : 0;
      const E swap = s_w[24];
      if (lane == 0) cs[0] = e_mul(swap, e_sub(swap, e_of(1)));
      if (lane < 4) cs[1 + lane] = e_sub(e_mul(swap, e_sub(s_w[lane + 4], s_w[lane])), s_w[25 + lane]);
      E s = i < 4 ? e_add(s_w[i], s_w[25 + i]) : i < 8 ? e_sub(s_w[i], s_w[25 + i - 4]) : s_w[i];
      u32 idx = 5;
      for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
        s.a = gl::add(s.a, POSEIDON_RC[12 * r + i]);
        if (r >= 4 && r < 26) {
          if (lane == 0) {
            const E in = s_w[65 + (r - 4)];
            cs[idx] = e_sub(s, in);
            s = sbox7(in);
          }
          idx += 1;
        } else {
          if (r != 0) {
            const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
            if (act) cs[idx + i] = e_sub(s, s_w[base + i]);
            s = s_w[base + i];
            idx += 12;
          }
          s = sbox7(s);
        }
        // MDS layer through LDS (one wavefront: program order is the only synchronisation needed besides the LDS counter)
        if (act) s_state[i] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        E acc = e_scale(s_state[i], POSEIDON_MDS_DIAG[i]);
        for (int k = 0; k < 12; ++k) acc = e_add(acc, e_scale(s_state[(k + i) % 12], POSEIDON_MDS_CIRC[k]));
        __builtin_amdgcn_wave_barrier();
        s = acc;
      }
      if (act) cs[idx + i] = e_sub(s, s_w[12 + i]);
    }
  } else if (wave == 1) {
    if (phase != 1) {
    // permutation argument: lane (c, q) = one chunk's check; lanes 32 + c: L_0(zeta) (Z_c(zeta) - 1)
    const bool subgroup = e_eq(zn, e_of(1));  // zeta in the subgroup: L_0 / Z_H degenerate (the prover refuses such a zeta)
    if (lane == 0 && subgroup) s_bad = 1;
    if (lane < nch * num_chunks) {
      const u32 c = lane / num_chunks, q = lane % num_chunks;
      const E bx = e_scale(zeta, betas[c]);
      E num = e_of(1), den = e_of(1);
      for (u32 j = q * qf; j < d.num_routed && j < (q + 1) * qf; ++j) {
        const E wg = E{gl::add(s_w[j].a, gammas[c]), s_w[j].b};
        num = e_mul(num, e_add(wg, e_scale(bx, k_is[j])));
        den = e_mul(den, e_add(wg, e_scale(e_at(sigmas, j), betas[c])));
      }
      const E prev = q == 0 ? e_at(zs, c) : e_at(pps, c * num_prods + q - 1);
      const E next = q == num_prods ? e_at(zs_next, c) : e_at(pps, c * num_prods + q);
      s_terms[nch + c * num_chunks + q] = e_sub(e_mul(prev, num), e_mul(next, den));
    } else if (lane >= 32 && lane < 32 + nch) {
      const u32 c = lane - 32;
      const E l0 = e_mul(zh, e_inv(e_scale(e_sub(zeta, e_of(1)), ((u64)1 << d.degree_bits) % gl::P)));
      s_terms[c] = e_mul(l0, e_sub(e_at(zs, c), e_of(1)));
    }
    }
  } else if (wave == 2) {
    // the small gates, one lane per constraint (phase 2: the public-input hash arrives with the challenges)
    for (u32 g = 0; phase != 1 && g < d.n_kinds; ++g) {
      const u32 k = d.kind[g];
      if (k == p2mt_cb::G_CONSTANT) {
        if (lane < d.num_constants) s_cs[g][lane] = e_sub(e_at(gc, lane), s_w[lane]);
      } else if (k == p2mt_cb::G_PUBLIC_INPUT) {
        if (lane < 4) s_cs[g][lane] = E{poseidon_fast::sub_any(s_w[lane].a, pi_hash[lane]), s_w[lane].b};
      } else if (k == p2mt_cb::G_ARITHMETIC) {
        if (lane < d.num_routed / 4)
          s_cs[g][lane] = e_sub(s_w[4 * lane + 3], e_add(e_mul(e_mul(s_w[4 * lane], s_w[4 * lane + 1]), e_at(gc, 0)),
                                                         e_mul(s_w[4 * lane + 2], e_at(gc, 1))));
      }
    }
  } else if (my_gate >= 0 && lane == 0 && phase != 2) {
    // one gate type of the in-circuit verifier, sequentially on one lane (a few hundred extension multiplications)
    E* cs = s_cs[my_gate];
    auto W = [&](int j) { return s_w[j]; };
    auto emit = [&](int j, E v) { cs[j] = v; };
    const E c0 = e_at(gc, 0), c1 = e_at(gc, 1);
    switch (d.kind[my_gate]) {
      case p2mt_cb::G_BASE_SUM: gates_rec::base_sum_gate<FExtDev>(W, emit); break;
      case p2mt_cb::G_ARITHMETIC_EXT: gates_rec::arithmetic_ext_gate<FExtDev>(W, c0, c1, emit); break;
      case p2mt_cb::G_MUL_EXT: gates_rec::mul_ext_gate<FExtDev>(W, c0, emit); break;
      case p2mt_cb::G_REDUCING: gates_rec::reducing_gate<FExtDev>(W, emit); break;
      case p2mt_cb::G_REDUCING_EXT: gates_rec::reducing_ext_gate<FExtDev>(W, emit); break;
      case p2mt_cb::G_RANDOM_ACCESS: gates_rec::random_access_gate<FExtDev>(W, c0, c1, emit); break;
      case p2mt_cb::G_COSET_INTERPOLATION: gates_rec::coset_interpolation_gate<FExtDev>(W, emit); break;
      case p2mt_cb::G_POSEIDON_MDS: gates_rec::poseidon_mds_gate<FExtDev>(W, emit); break;
      default: s_bad = 1; break;  // unknown gate type
    }
  }
  __syncthreads();
  if (phase == 1) {  // (workgroup-uniform)
    for (u32 j = t; j < p2mt_cb::kMaxGateTypes * p2mt_cb::kNumGateConstraints; j += kOpenBlock) {
      const E v = (&s_cs[0][0])[j];
      gcs[2 * j] = v.a, gcs[2 * j + 1] = v.b;
    }
    if (t == 0) gcs[kGcsWords - 2] = (u64)s_bad;
    return;
  }
  // filtered sum per constraint index: sum_g f_g(selector openings) cs_g[j]
  if (t < p2mt_cb::kNumGateConstraints) {
    E acc = e_of(0);
    for (u32 g = 0; g < d.n_kinds; ++g) {
      if (d.kind[g] == p2mt_cb::G_NOOP) continue;
      const E s = e_at(consts, d.sel[g]);
      E f = e_of(1);
      for (u32 k = d.gs[g]; k < d.ge[g]; ++k)
        if (k != g) f = e_mul(f, E{poseidon_fast::sub_any(k, s.a), poseidon_fast::sub_any(0, s.b)});
      if (d.num_selectors > 1) f = e_mul(f, E{poseidon_fast::sub_any(0xFFFFFFFFull, s.a), poseidon_fast::sub_any(0, s.b)});
      acc = e_add(acc, e_mul(f, s_cs[g][t]));
    }
    s_terms[nch * (1 + num_chunks) + t] = acc;
  }
  __syncthreads();
  // sum_t alpha_c^t term_t on wave c: lane l takes t = l, l + 64, ... (Horner in alpha^64), times alpha^l, tree sum
  if (wave < nch) {
    const u32 c = wave;
    const u64 al = alphas[c];
    const u64 a64 = gl::pow(al, 64);
    E acc = e_of(0);
    const u32 last_m = (n_terms - 1) / 64;
    for (u32 m = last_m + 1; m-- > 0;) {
      const u32 ti = lane + 64 * m;
      acc = e_scale(acc, a64);
      if (ti < n_terms) acc = e_add(acc, s_terms[ti]);
    }
    s_van[c][lane] = e_scale(acc, gl::pow(al, lane));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
      E van = e_of(0);
      for (int l = 0; l < 64; ++l) van = e_add(van, s_van[c][l]);
      E q = e_of(0);
      for (u32 k = qf; k-- > 0;) q = e_add(e_mul(q, zn), e_at(quot, c * qf + k));
      if (!e_eq(e_mul(zh, q), van)) atomicOr(&s_bad, 1);
    }
  }
  __syncthreads();
  if (t == 0) *res_ok = s_bad ? 0 : 1;
}

}  // namespace

// ---------------------------------------------------------------- staging (host)
// A verification is a chain of ~100 dependent permutations (the transcript) with everything else hanging off it: the row sponges
// need nothing but the proof, the vanishing-polynomial check needs the challenges up to zeta, the FRI arithmetic and the path folds
// need the end of the transcript.  Two side streams carry what does not have to wait; events tie them to the library stream.
namespace {
struct VerifyStreams {
  hipStream_t s_leaf = nullptr, s_open = nullptr;
  hipEvent_t e_proof = nullptr, e_zeta = nullptr, e_leaf = nullptr, e_open = nullptr;
  u64* d_gcs = nullptr;  // the gates' unfiltered constraints between the two phases of a single verification (k_verify_openings)
  bool early = false;    // phase 1 is in flight for the verification at hand
};
bool args_ok(const VerifyDevArgs& a) {
  const unsigned chunks = (a.d.num_routed + a.d.quotient_degree_factor - 1) / a.d.quotient_degree_factor;
  // fail closed on everything the kernels hard-code (none of it reachable with standard_recursion_config): k_verify_fri runs 16 lanes
  // per query and shifts by 4 - arity_bits; cap strides are 64 words (cap height 4); the final polynomial and the query list are
  // indexed from length - 1; k_verify_openings gives the gate kinds from BaseSum on one wave each, waves 3..15
  if (a.fri.num_query_rounds == 0 || a.final_len == 0 || a.fri.cap_height != 4 || a.fri.num_reductions > 8) return false;
  for (uint32_t l = 0; l < a.fri.num_reductions; ++l)
    if (a.fri.reduction_arity_bits[l] == 0 || a.fri.reduction_arity_bits[l] > 4) return false;
  unsigned wide_kinds = 0;
  for (uint32_t g = 0; g < a.d.n_kinds && g < 16; ++g) wide_kinds += a.d.kind[g] >= p2mt_cb::G_BASE_SUM;
  if (wide_kinds > 13) return false;
  return a.d.num_challenges <= 2 && a.d.num_wires <= p2mt_cb::kNumWires && a.d.n_kinds <= p2mt_cb::kMaxGateTypes && a.d.n_kinds <= 16 &&
         a.d.num_challenges * chunks <= 32 && a.d.num_challenges * (1 + chunks) + p2mt_cb::kNumGateConstraints <= kMaxTerms;
}
}  // namespace

int p2mt::verify_streams_create(void** out) {
  VerifyStreams* v = new (std::nothrow) VerifyStreams();
  if (!v) return p2mt::fail(P2MT_ENOMEM, "verify: out of host memory");
  if (hipStreamCreateWithFlags(&v->s_leaf, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&v->s_open, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&v->e_proof, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&v->e_zeta, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&v->e_leaf, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&v->e_open, hipEventDisableTiming) != hipSuccess) {
    p2mt::verify_streams_destroy(v);
    return p2mt::fail(P2MT_EHIP, "verify: cannot create streams / events");
  }
  if (hipMalloc((void**)&v->d_gcs, kGcsWords * 8) != hipSuccess) {
    (void)hipGetLastError();
    p2mt::verify_streams_destroy(v);
    return p2mt::fail(P2MT_ENOMEM, "verify: hipMalloc(gate constraints) failed");
  }
  *out = v;
  return P2MT_OK;
}
void p2mt::verify_streams_destroy(void* p) {
  VerifyStreams* v = static_cast<VerifyStreams*>(p);
  if (!v) return;
  if (v->s_leaf) (void)hipStreamDestroy(v->s_leaf);
  if (v->s_open) (void)hipStreamDestroy(v->s_open);
  for (hipEvent_t e : {v->e_proof, v->e_zeta, v->e_leaf, v->e_open})
    if (e) (void)hipEventDestroy(e);
  if (v->d_gcs) (void)hipFree(v->d_gcs);
  delete v;
}

// wait (host) for whatever is in flight on the side streams: error paths of the caller
void p2mt::verify_streams_join(void* p) {
  VerifyStreams* v = static_cast<VerifyStreams*>(p);
  if (!v) return;
  (void)hipStreamSynchronize(v->s_leaf);
  (void)hipStreamSynchronize(v->s_open);
}

// (1) the proof is on the device: the row sponges start on their own stream
int p2mt::verify_dev_begin(void* vs, const uint64_t* dv, uint64_t* d_digests, const VerifyDevArgs& a, bool early_gates) {
  VerifyStreams* v = static_cast<VerifyStreams*>(vs);
  v->early = false;
  if (!args_ok(a)) return p2mt::fail(P2MT_EINVAL, "verify: circuit configuration beyond standard_recursion_config");
  const unsigned n_items = a.fri.num_query_rounds * (4 + a.fri.num_reductions);
  P2MT_HIP(hipEventRecord(v->e_proof, p2mt::rt().stream));
  P2MT_HIP(hipStreamWaitEvent(v->s_leaf, v->e_proof, 0));
  hipLaunchKernelGGL(k_verify_leaf_digests, bgrid((n_items + kMerkleBlock / 64 - 1) / (kMerkleBlock / 64)), dim3(kMerkleBlock), 0, v->s_leaf, dv,
                     d_digests, a, barg(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipEventRecord(v->e_leaf, v->s_leaf));
  if (early_gates) {  // single verification, transcript on the host: the challenge-free part of the vanishing-polynomial check starts now
    P2MT_HIP(hipStreamWaitEvent(v->s_open, v->e_proof, 0));
    hipLaunchKernelGGL(k_verify_openings, dim3(1), dim3(kOpenBlock), 0, v->s_open, dv, (int*)nullptr, (const u64*)nullptr, a, barg(), v->d_gcs, 1);
    P2MT_LAUNCH_CHECK();
    v->early = true;
  }
  return P2MT_OK;
}
// (2) betas, gammas, alphas and zeta are out: the vanishing-polynomial check runs beside the rest of the transcript.
// d_res[0] = 1 if the openings satisfy the identity.
int p2mt::verify_dev_after_zeta(void* vs, const uint64_t* dv, int* d_res, const uint64_t* d_k_is, const VerifyDevArgs& a) {
  VerifyStreams* v = static_cast<VerifyStreams*>(vs);
  P2MT_HIP(hipEventRecord(v->e_zeta, p2mt::rt().stream));
  P2MT_HIP(hipStreamWaitEvent(v->s_open, v->e_zeta, 0));
  hipLaunchKernelGGL(k_verify_openings, bgrid(1), dim3(kOpenBlock), 0, v->s_open, dv, d_res, d_k_is, a, barg(), (u64*)nullptr, 0);
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipEventRecord(v->e_open, v->s_open));
  return P2MT_OK;
}
// (3) the transcript is complete: items, FRI arithmetic (d_res[1] = INT-max pattern or 8 * first failing query + reason: 3 layer value,
// 5 final polynomial) and the path folds (*d_flag = smallest failing item + 1) on the library stream, which then waits for the side
// streams: after this call one copy of {d_flag, d_res} tells the verdict.
int p2mt::verify_dev_finish(void* vs, const uint64_t* dv, void* d_items, const uint64_t* d_digests, int* d_flag, int* d_res,
                            const VerifyDevArgs& a) {
  VerifyStreams* v = static_cast<VerifyStreams*>(vs);
  hipStream_t st = p2mt::rt().stream;
  const unsigned nq = a.fri.num_query_rounds, n_items = nq * (4 + a.fri.num_reductions);
  // The FRI arithmetic (the longer of the two, ~50 us) stays on the library stream, right behind the transcript; the path folds (~38 us,
  // each wavefront works out its own (row, path, cap) triple) go beside it on the side stream the vanishing-polynomial check used
  // (long done by now), which also has to have seen the leaf digests.
  (void)d_items;
  P2MT_HIP(hipEventRecord(v->e_zeta, st));  // (re-used: "the transcript is complete")
  P2MT_HIP(hipStreamWaitEvent(v->s_open, v->e_zeta, 0));
  P2MT_HIP(hipStreamWaitEvent(v->s_open, v->e_leaf, 0));
  hipLaunchKernelGGL(k_verify_paths, bgrid((n_items + kMerkleBlock / 64 - 1) / (kMerkleBlock / 64)), dim3(kMerkleBlock), 0, v->s_open, dv,
                     a, d_digests, n_items, d_flag, barg(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipEventRecord(v->e_open, v->s_open));  // behind k_verify_openings AND k_verify_paths on that stream
  hipLaunchKernelGGL(k_verify_fri, bgrid((nq + kFriBlock / 16 - 1) / (kFriBlock / 16)), dim3(kFriBlock), 0, st, dv, d_res + 1, a, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipStreamWaitEvent(st, v->e_open, 0));
  return P2MT_OK;
}

// The host derived every challenge (host_poseidon.h) and sent them up on the library stream: the three checks that need them start
// together -- the vanishing-polynomial check on its side stream, the path folds behind the row sponges on theirs, the FRI arithmetic
// on the library stream, which then waits for both.  After this call one copy of {d_flag, d_res} tells the verdict.
int p2mt::verify_dev_with_challenges(void* vs, const uint64_t* dv, const uint64_t* d_digests, int* d_flag, int* d_res,
                                     const uint64_t* d_k_is, const VerifyDevArgs& a) {
  VerifyStreams* v = static_cast<VerifyStreams*>(vs);
  hipStream_t st = p2mt::rt().stream;
  const unsigned nq = a.fri.num_query_rounds, n_items = nq * (4 + a.fri.num_reductions);
  P2MT_HIP(hipEventRecord(v->e_zeta, st));  // "the challenges are on the device"
  // the longest of the three first, on the stream the challenges came up on (no event between them): it was the last to be
  // enqueued and its ~54 us were the tail of a verification
  hipLaunchKernelGGL(k_verify_fri, bgrid((nq + kFriBlock / 16 - 1) / (kFriBlock / 16)), dim3(kFriBlock), 0, st, dv, d_res + 1, a, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipStreamWaitEvent(v->s_leaf, v->e_zeta, 0));  // (behind k_verify_leaf_digests on that stream)
  hipLaunchKernelGGL(k_verify_paths, bgrid((n_items + kMerkleBlock / 64 - 1) / (kMerkleBlock / 64)), dim3(kMerkleBlock), 0, v->s_leaf, dv,
                     a, d_digests, n_items, d_flag, barg(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  P2MT_HIP(hipEventRecord(v->e_leaf, v->s_leaf));
  P2MT_HIP(hipStreamWaitEvent(v->s_open, v->e_zeta, 0));
  hipLaunchKernelGGL(k_verify_openings, bgrid(1), dim3(kOpenBlock), 0, v->s_open, dv, d_res, d_k_is, a, barg(), v->d_gcs, v->early ? 2 : 0);
  P2MT_LAUNCH_CHECK();
  v->early = false;
  P2MT_HIP(hipEventRecord(v->e_open, v->s_open));
  P2MT_HIP(hipStreamWaitEvent(st, v->e_open, 0));
  P2MT_HIP(hipStreamWaitEvent(st, v->e_leaf, 0));
  return P2MT_OK;
}
