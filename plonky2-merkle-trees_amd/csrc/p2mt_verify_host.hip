// p2mt_verify_host.hip -- the arithmetic half of CircuitData::verify (host code; the hashing half -- transcript and Merkle
// paths -- runs on the device, p2mt_circuit.hip).
//
// Replaces what `circuit_data.verify(proof)` (/root/reference/src/mmr/mmr_plonky2_verifier.rs:150,
// mmr_plonky2_verifier_1_recursion.rs:193,220) runs inside plonky2 (git rev 3b21b87d, not in the reference tree; parity
// unpinned): plonk/verifier.rs verify_with_challenges (vanishing polynomial at zeta against the quotient openings:
// plonk/vanishing_poly.rs eval_vanishing_poly, gates/*::eval_unfiltered over the quadratic extension) and the field side of
// fri/verifier.rs (PrecomputedReducedOpenings, fri_combine_initial, compute_evaluation, final polynomial).  A few thousand
// extension-field multiplications per proof: host work, as in the reference.
#include "runtime.h"
#include "poseidon_constants.h"  // host copy of the round constants / MDS (P2MT_QUAL defaults to static const)
#include "gates_recursion.hip.h"  // the in-circuit verifier's gate types, one source for base field (device) and extension (here)

#include <algorithm>
#include <vector>

namespace {

typedef uint64_t u64;
typedef unsigned __int128 u128;
constexpr u64 P = 0xFFFFFFFF00000001ull;

inline u64 f_add(u64 a, u64 b) { return (u64)(((u128)a + b) % P); }
inline u64 f_sub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
inline u64 f_mul(u64 a, u64 b) { return (u64)(((u128)a * b) % P); }
inline u64 f_pow(u64 a, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1, a = f_mul(a, a))
    if (e & 1) r = f_mul(r, a);
  return r;
}
inline u64 f_inv(u64 a) { return f_pow(a, P - 2); }
inline u64 root_of_unity(unsigned log_n) {
  u64 g = f_pow(7, (P - 1) >> 32);
  for (unsigned i = log_n; i < 32; ++i) g = f_mul(g, g);
  return g;
}

struct E {  // a + bX, X^2 = 7
  u64 a, b;
};
inline E e_of(u64 a) { return E{a, 0}; }
inline E operator+(E x, E y) { return E{f_add(x.a, y.a), f_add(x.b, y.b)}; }
inline E operator-(E x, E y) { return E{f_sub(x.a, y.a), f_sub(x.b, y.b)}; }
inline E operator*(E x, E y) { return E{f_add(f_mul(x.a, y.a), f_mul(7, f_mul(x.b, y.b))), f_add(f_mul(x.a, y.b), f_mul(x.b, y.a))}; }
inline E e_scale(E x, u64 s) { return E{f_mul(x.a, s), f_mul(x.b, s)}; }
inline bool operator==(E x, E y) { return x.a == y.a && x.b == y.b; }
inline E e_inv(E x) {
  const u64 ni = f_inv(f_sub(f_mul(x.a, x.a), f_mul(7, f_mul(x.b, x.b))));
  return E{f_mul(x.a, ni), f_mul(f_sub(0, x.b), ni)};
}
inline E e_pow(E x, u64 e) {
  E r = e_of(1);
  for (; e; e >>= 1, x = x * x)
    if (e & 1) r = r * x;
  return r;
}
inline E e_at(const u64* w, size_t i) { return E{w[2 * i], w[2 * i + 1]}; }
inline size_t brev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

inline E sbox7(E x) {
  const E x2 = x * x, x4 = x2 * x2, x3 = x2 * x;
  return x4 * x3;
}
void mds_layer(E (&s)[12]) {
  E out[12];
  for (int r = 0; r < 12; ++r) {
    E acc = e_scale(s[r], POSEIDON_MDS_DIAG[r]);
    for (int i = 0; i < 12; ++i) acc = acc + e_scale(s[(i + r) % 12], POSEIDON_MDS_CIRC[i]);
    out[r] = acc;
  }
  for (int r = 0; r < 12; ++r) s[r] = out[r];
}
// PoseidonGate::eval_unfiltered: the permutation replayed from the opened wire values (123 constraints)
void poseidon_gate_eval(const E* w, std::vector<E>& out) {
  const E swap = w[24];
  out.push_back(swap * (swap - e_of(1)));
  for (int i = 0; i < 4; ++i) out.push_back(swap * (w[i + 4] - w[i]) - w[25 + i]);
  E s[12];
  for (int i = 0; i < 4; ++i) {
    s[i] = w[i] + w[25 + i];
    s[i + 4] = w[i + 4] - w[25 + i];
  }
  for (int i = 8; i < 12; ++i) s[i] = w[i];
  for (int r = 0; r < POSEIDON_ROUNDS; ++r) {
    for (int i = 0; i < 12; ++i) s[i] = s[i] + e_of(POSEIDON_RC[12 * r + i]);
    if (r >= 4 && r < 26) {
      const E in = w[65 + (r - 4)];
      out.push_back(s[0] - in);
      s[0] = sbox7(in);
    } else {
      if (r != 0) {
        const int base = r < 4 ? 29 + 12 * (r - 1) : 87 + 12 * (r - 26);
        for (int i = 0; i < 12; ++i) {
          out.push_back(s[i] - w[base + i]);
          s[i] = w[base + i];
        }
      }
      for (int i = 0; i < 12; ++i) s[i] = sbox7(s[i]);
    }
    mds_layer(s);
  }
  for (int i = 0; i < 12; ++i) out.push_back(s[i] - w[12 + i]);
}

// the evaluation field of the verifier: the quadratic extension (gates_recursion.hip.h's F)
struct FExtHost {
  typedef E T;
  static T add(T a, T b) { return a + b; }
  static T sub(T a, T b) { return a - b; }
  static T mul(T a, T b) { return a * b; }
  static T mulc(T a, u64 c) { return e_scale(a, c); }
  static T addc(T a, u64 c) { return a + e_of(c); }
  static T subc(T a, u64 c) { return a - e_of(c); }
  static T fromc(u64 c) { return e_of(c); }
};

}  // namespace

// openings: the OpeningSet of the proof (constants | sigmas | wires | zs | zs_next | partial products | quotient), 2 words each
int p2mt::verify_openings_host(const VerifyDesc& d, const uint64_t* k_is, const uint64_t zeta_w[2], const uint64_t* openings,
                               const uint64_t pi_hash[4], const uint64_t* betas, const uint64_t* gammas, const uint64_t* alphas) {
  const unsigned nch = d.num_challenges, qf = d.quotient_degree_factor;
  const unsigned num_chunks = (d.num_routed + qf - 1) / qf, num_prods = num_chunks - 1;
  const unsigned n_consts = d.num_selectors + d.num_constants;
  const uint64_t* o = openings;
  auto take = [&](size_t count) {
    const uint64_t* p = o;
    o += 2 * count;
    return p;
  };
  const uint64_t *consts = take(n_consts), *sigmas = take(d.num_routed), *wires = take(d.num_wires), *zs = take(nch);
  const uint64_t *zs_next = take(nch), *pps = take((size_t)nch * num_prods), *quot = take((size_t)nch * qf);
  const E zeta{zeta_w[0], zeta_w[1]};
  E zn = zeta;
  for (unsigned i = 0; i < d.degree_bits; ++i) zn = zn * zn;
  if (zn == e_of(1)) return 0;  // zeta in the subgroup: L_0 / Z_H degenerate (the prover refuses such a zeta)
  const E zh = zn - e_of(1);
  const E l0 = zh * e_inv(e_scale(zeta - e_of(1), ((u64)1 << d.degree_bits) % P));
  std::vector<E> w(d.num_wires);
  for (unsigned j = 0; j < d.num_wires; ++j) w[j] = e_at(wires, j);
  // vanishing terms: L_0 (Z - 1) | partial-product checks | gate constraints
  std::vector<E> terms;
  for (unsigned c = 0; c < nch; ++c) terms.push_back(l0 * (e_at(zs, c) - e_of(1)));
  for (unsigned c = 0; c < nch; ++c) {
    const E bx = e_scale(zeta, betas[c]);
    for (unsigned q = 0; q < num_chunks; ++q) {
      E num = e_of(1), den = e_of(1);
      for (unsigned j = q * qf; j < d.num_routed && j < (q + 1) * qf; ++j) {
        const E wg = w[j] + e_of(gammas[c]);
        num = num * (wg + e_scale(bx, k_is[j]));
        den = den * (wg + e_scale(e_at(sigmas, j), betas[c]));
      }
      const E prev = q == 0 ? e_at(zs, c) : e_at(pps, c * num_prods + q - 1);
      const E next = q == num_prods ? e_at(zs_next, c) : e_at(pps, c * num_prods + q);
      terms.push_back(prev * num - next * den);
    }
  }
  std::vector<E> gate_terms(123, e_of(0));
  const uint64_t* gc = consts + 2 * d.num_selectors;
  for (unsigned g = 0; g < d.n_kinds; ++g) {
    std::vector<E> cs;
    switch (d.kind[g]) {
      case 1:  // ConstantGate
        for (unsigned i = 0; i < d.num_constants; ++i) cs.push_back(e_at(gc, i) - w[i]);
        break;
      case 2:  // PublicInputGate
        for (int i = 0; i < 4; ++i) cs.push_back(w[i] - e_of(pi_hash[i]));
        break;
      case 3:  // ArithmeticGate
        for (unsigned i = 0; i < d.num_routed / 4; ++i)
          cs.push_back(w[4 * i + 3] - (w[4 * i] * w[4 * i + 1] * e_at(gc, 0) + w[4 * i + 2] * e_at(gc, 1)));
        break;
      case 4:  // PoseidonGate
        poseidon_gate_eval(w.data(), cs);
        break;
      case 0: break;  // NoopGate
      default: {      // the gate types of the in-circuit verifier (5..12)
        cs.assign(123, e_of(0));
        size_t used = 0;
        auto W = [&](int j) { return w[j]; };
        auto emit = [&](int j, E v) {
          cs[j] = v;
          used = std::max(used, (size_t)j + 1);
        };
        const E c0 = e_at(gc, 0), c1 = e_at(gc, 1);
        switch (d.kind[g]) {
          case 5: gates_rec::base_sum_gate<FExtHost>(W, emit); break;
          case 6: gates_rec::arithmetic_ext_gate<FExtHost>(W, c0, c1, emit); break;
          case 7: gates_rec::mul_ext_gate<FExtHost>(W, c0, emit); break;
          case 8: gates_rec::reducing_gate<FExtHost>(W, emit); break;
          case 9: gates_rec::reducing_ext_gate<FExtHost>(W, emit); break;
          case 10: gates_rec::random_access_gate<FExtHost>(W, c0, c1, emit); break;
          case 11: gates_rec::coset_interpolation_gate<FExtHost>(W, emit); break;
          case 12: gates_rec::poseidon_mds_gate<FExtHost>(W, emit); break;
          default: return 0;  // unknown gate type
        }
        cs.resize(used);
        break;
      }
    }
    const E s = e_at(consts, d.sel[g]);
    E f = e_of(1);
    for (unsigned k = d.gs[g]; k < d.ge[g]; ++k)
      if (k != g) f = f * (e_of(k) - s);
    if (d.num_selectors > 1) f = f * (e_of(0xFFFFFFFFull) - s);
    for (size_t j = 0; j < cs.size(); ++j) gate_terms[j] = gate_terms[j] + f * cs[j];
  }
  terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
  for (unsigned c = 0; c < nch; ++c) {
    E van = e_of(0);
    for (size_t t = terms.size(); t-- > 0;) van = e_scale(van, alphas[c]) + terms[t];
    E acc = e_of(0);
    for (unsigned k = qf; k-- > 0;) acc = acc * zn + e_at(quot, c * qf + k);
    if (!(zh * acc == van)) return 0;
  }
  return 1;
}

// The field side of verify_fri_proof for every query.  openings: FriOpenings order (batch 0 = every polynomial of the
// oracles in order at zeta, batch 1 = the first num_challenges polynomials of oracle 2 at g zeta).  Returns 0 or the
// reason (3 layer value inconsistent with the previous layer, 5 final polynomial).
int p2mt::verify_fri_queries_host(const p2mt_fri_params& p, const uint64_t* n_polys, size_t n_oracles, size_t n_next,
                                  const uint64_t zeta_w[2], const uint64_t* openings, const uint64_t alpha_w[2],
                                  const uint64_t* betas_w, const uint64_t* fri_proof, size_t fri_len, const uint64_t* x_indices) {
  const unsigned log_big = p.degree_bits + p.rate_bits;
  unsigned total_arity = 0;
  for (uint32_t l = 0; l < p.num_reductions; ++l) total_arity += p.reduction_arity_bits[l];
  const size_t cap_words = (size_t)4 << p.cap_height, final_len = (size_t)1 << (p.degree_bits - total_arity);
  const uint64_t* final_words = fri_proof + fri_len - 1 - 2 * final_len;
  const E alpha{alpha_w[0], alpha_w[1]}, zeta{zeta_w[0], zeta_w[1]};
  const E gzeta = e_scale(zeta, root_of_unity(p.degree_bits));
  size_t n_all = 0;
  for (size_t o = 0; o < n_oracles; ++o) n_all += n_polys[o];
  // PrecomputedReducedOpenings
  E reduced[2] = {e_of(0), e_of(0)};
  for (size_t j = n_all; j-- > 0;) reduced[0] = reduced[0] * alpha + e_at(openings, j);
  for (size_t j = n_next; j-- > 0;) reduced[1] = reduced[1] * alpha + e_at(openings, n_all + j);
  const E alpha_n_next = e_pow(alpha, n_next);
  const uint64_t* w = fri_proof + p.num_reductions * cap_words;
  for (uint32_t q = 0; q < p.num_query_rounds; ++q) {
    size_t x_index = (size_t)x_indices[q];
    const uint64_t* leaf_of[8];
    for (size_t o = 0; o < n_oracles; ++o) {
      leaf_of[o] = w;
      w += n_polys[o] + 4 * (size_t)(log_big - p.cap_height);
    }
    u64 subgroup_x = f_mul(7, f_pow(root_of_unity(log_big), brev(x_index, log_big)));
    // fri_combine_initial: batch 0 then batch 1
    E acc0 = e_of(0), acc1 = e_of(0);
    for (size_t o = n_oracles; o-- > 0;)
      for (size_t j = n_polys[o]; j-- > 0;) acc0 = acc0 * alpha + e_of(leaf_of[o][j]);
    for (size_t j = n_next; j-- > 0;) acc1 = acc1 * alpha + e_of(leaf_of[2][j]);
    E sum = (acc0 - reduced[0]) * e_inv(e_of(subgroup_x) - zeta);
    sum = sum * alpha_n_next + (acc1 - reduced[1]) * e_inv(e_of(subgroup_x) - gzeta);
    E old_eval = e_scale(sum, subgroup_x);
    unsigned log_sz = log_big;
    for (uint32_t l = 0; l < p.num_reductions; ++l) {
      const unsigned ab = p.reduction_arity_bits[l];
      const size_t arity = (size_t)1 << ab, coset_index = x_index >> ab, within = x_index & (arity - 1);
      if (!(e_at(w, within) == old_eval)) return 3;
      {  // compute_evaluation: interpolate the coset's values (committed in bit-reversed order) and evaluate at beta
        const E beta{betas_w[2 * l], betas_w[2 * l + 1]};
        const u64 g = root_of_unity(ab);
        u64 pts[16];
        E ys[16];
        u64 y = f_mul(subgroup_x, f_pow(g, arity - brev(within, ab)));
        for (size_t i = 0; i < arity; ++i, y = f_mul(y, g)) {
          pts[i] = y;
          ys[i] = e_at(w, brev(i, ab));
        }
        E s = e_of(0);
        for (size_t i = 0; i < arity; ++i) {
          E num = e_of(1);
          u64 den = 1;
          for (size_t j = 0; j < arity; ++j) {
            if (j == i) continue;
            num = num * (beta - e_of(pts[j]));
            den = f_mul(den, f_sub(pts[i], pts[j]));
          }
          s = s + ys[i] * e_scale(num, f_inv(den));
        }
        old_eval = s;
      }
      w += 2 * arity + 4 * (size_t)(log_sz - ab - p.cap_height);
      for (unsigned k = 0; k < ab; ++k) subgroup_x = f_mul(subgroup_x, subgroup_x);
      x_index = coset_index;
      log_sz -= ab;
    }
    E fe = e_of(0);
    for (size_t i = final_len; i-- > 0;) fe = e_scale(fe, subgroup_x) + e_at(final_words, i);
    if (!(fe == old_eval)) return 5;
  }
  return 0;
}
