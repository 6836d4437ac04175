// p2mt_recursion.hip -- plonky2's in-circuit verifier for the reference's recursion circuit (host code of the library).
//
// Replaces what /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:95-104 runs inside plonky2 (git rev 3b21b87d, NOT in
// the reference tree -- parity unpinned, checked gate for gate against the tests' CPU restatement of the same algorithm):
//   builder.add_virtual_proof_with_pis(&inner.common)                    (:95)
//   builder.add_virtual_verifier_data(cap_height)                        (:98)
//   builder.verify_proof::<PoseidonGoldilocksConfig>(proof, vd, common)  (:101-104)
// and, for the witness (:201-202), pw.set_proof_with_pis_target / pw.set_verifier_data_target.
// Sources restated: plonk/circuit_builder.rs, gadgets/{arithmetic_extension, split_join, split_base, range_check, random_access,
// interpolation, hash}.rs, util/reducing.rs (ReducingFactorTarget), iop/challenger.rs (RecursiveChallenger),
// plonk/get_challenges.rs, recursion/recursive_verifier.rs, plonk/vanishing_poly.rs (eval_vanishing_poly_circuit),
// plonk/plonk_common.rs, gates/*.rs eval_unfiltered_circuit (Noop, Constant, PublicInput, Arithmetic, Poseidon through
// PoseidonMdsGate), fri/recursive_verifier.rs, hash/merkle_proofs.rs.
//
// A ProofWithPublicInputsTarget is FLAT: one target per proof word in the word order of p2mt_circuit_prove, so that
// set_proof_with_pis_target is target[i] <- word[i].  A VerifierCircuitTarget is 64 cap targets followed by 4 digest targets.
// The circuit this builds for an inner MMR-verifier circuit of 2^6 rows has ~2 500 gate rows -> 2^12 after padding.
#include "circuit_types.h"
#include "poseidon_constants.h"  // host copy of ALL_ROUND_CONSTANTS (P2MT_QUAL defaults to static const)

#include <algorithm>
#include <vector>

using namespace p2mt_cb;

namespace {

typedef p2mt_circuit_builder B;

// ------------------------------------------------------------------------------------------------ base-field gadgets
u64 zero(B* b) { return cb_constant(b, 0); }
u64 one(B* b) { return cb_constant(b, 1); }
u64 arith(B* b, u64 c0, u64 c1, u64 m0, u64 m1, u64 ad) {
  u64 out = 0;
  (void)cb_arithmetic(b, c0, c1, m0, m1, ad, &out);  // operands come from this file: always valid targets
  return out;
}
u64 mul(B* b, u64 x, u64 y) { return arith(b, 1, 0, x, y, x); }
u64 mul_add(B* b, u64 x, u64 y, u64 z) { return arith(b, 1, 1, x, y, z); }
u64 mul_const_add(B* b, u64 c, u64 x, u64 y) { return arith(b, c, 1, one(b), x, y); }
void connect(B* b, u64 x, u64 y) { b->copies.emplace_back(x, y); }
void assert_zero(B* b, u64 x) { connect(b, x, zero(b)); }

u64 exp_power_of_2(B* b, u64 base, unsigned power_log) {
  for (unsigned k = 0; k < power_log; ++k) base = mul(b, base, base);
  return base;
}
// base^(sum bits_i 2^i) for a constant base: product = (base^(2^i) - 1) * product * bit + product per bit (at most 20 bits here;
// plonky2 switches to an ExponentiationGate beyond num_base_arithmetic_ops_per_gate)
u64 exp_from_bits_const_base(B* b, u64 base, const std::vector<u64>& bits) {
  (void)cb_constant(b, base);  // `let base_t = self.constant(base)`: registered even when unused
  u64 product = one(b);
  for (size_t i = 0; i < bits.size(); ++i) product = arith(b, h_sub(h_pow(base, 1ull << i), 1), 1, product, bits[i], product);
  return product;
}
u64 le_sum(B* b, const std::vector<u64>& bits) {
  if (bits.empty()) return zero(b);
  const u64 two = cb_constant(b, 2);
  u64 acc = bits.back();
  for (size_t k = bits.size() - 1; k-- > 0;) acc = mul_add(b, two, acc, bits[k]);
  return acc;
}
// gadgets/split_join.rs split_le: num_bits BoolTargets (little-endian) through ceil(num_bits / 63) BaseSumGate<2> rows
std::vector<u64> split_le(B* b, u64 integer, unsigned num_bits) {
  std::vector<u64> bits;
  if (num_bits == 0) return bits;
  const unsigned k = (num_bits + kBaseSumLimbs - 1) / kBaseSumLimbs;
  std::vector<u32> rows;
  for (unsigned r = 0; r < k; ++r) rows.push_back(cb_add_gate(b, G_BASE_SUM));
  for (u32 r : rows)
    for (u32 j = 0; j < kBaseSumLimbs; ++j) bits.push_back(wire_t(r, 1 + j));
  for (size_t j = num_bits; j < bits.size(); ++j) assert_zero(b, bits[j]);
  bits.resize(num_bits);
  const u64 base = h_pow(2, kBaseSumLimbs);
  u64 acc = zero(b);
  for (size_t r = rows.size(); r-- > 0;) acc = mul_const_add(b, base, acc, wire_t(rows[r], 0));
  connect(b, acc, integer);
  Gen g{};
  g.kind = GEN_WIRE_SPLIT;
  g.t.push_back(integer);
  for (u32 r : rows) g.t.push_back(wire_t(r, 0));
  b->gens.push_back(g);
  return bits;
}

// ------------------------------------------------------------------------------------------------ extension-field gadgets
Ext ext_const(B* b, ExtConst c) { return Ext{cb_constant(b, c.a), cb_constant(b, c.b)}; }
Ext zero_ext(B* b) { return ext_const(b, {0, 0}); }
Ext one_ext(B* b) { return ext_const(b, {1, 0}); }
Ext to_ext(B* b, u64 t) { return Ext{t, zero(b)}; }
void connect_ext(B* b, Ext x, Ext y) {
  connect(b, x[0], y[0]);
  connect(b, x[1], y[1]);
}
Ext wire_ext(u32 row, u32 col) { return Ext{wire_t(row, col), wire_t(row, col + 1)}; }
bool as_const(const B* b, Ext t, ExtConst* out) {
  auto a = b->target_to_const.find(t[0]), c = b->target_to_const.find(t[1]);
  if (a == b->target_to_const.end() || c == b->target_to_const.end()) return false;
  *out = {a->second, c->second};
  return true;
}

// gadgets/arithmetic_extension.rs arithmetic_extension: const_0 * m0 * m1 + const_1 * addend, with plonky2's special cases, its
// operation cache, and MulExtensionGate when the addend is the zero constant
Ext arithmetic_ext(B* b, u64 c0, u64 c1, Ext m0, Ext m1, Ext ad) {
  c0 %= kP;
  c1 %= kP;
  {
    const Ext z = zero_ext(b);
    ExtConst m0c{}, m1c{}, adc{};
    const bool h0 = as_const(b, m0, &m0c), h1 = as_const(b, m1, &m1c), ha = as_const(b, ad, &adc);
    const bool first_zero = c0 == 0 || m0 == z || m1 == z, second_zero = c1 == 0 || ad == z;
    const bool first_known = first_zero || (h0 && h1), second_known = second_zero || ha;
    if (first_known && second_known) {
      const ExtConst f = first_zero ? ExtConst{0, 0} : ec_scale(ec_mul(m0c, m1c), c0);
      const ExtConst s = second_zero ? ExtConst{0, 0} : ec_scale(adc, c1);
      return ext_const(b, ec_add(f, s));
    }
    if (first_zero && c1 == 1) return ad;
    if (second_zero) {
      if (h0) {
        const ExtConst t = ec_scale(m0c, c0);
        if (t.a == 1 && t.b == 0) return m1;
      }
      if (h1) {
        const ExtConst t = ec_scale(m1c, c0);
        if (t.a == 1 && t.b == 0) return m0;
      }
    }
  }
  const auto op = std::make_tuple(c0, c1, m0[0], m0[1], m1[0], m1[1], ad[0], ad[1]);
  auto hit = b->ext_arith_results.find(op);
  if (hit != b->ext_arith_results.end()) return hit->second;
  ExtConst adc{};
  Ext res;
  Gen g{};
  if (as_const(b, ad, &adc) && adc.a == 0 && adc.b == 0) {  // "If the addend is zero, we use a multiplication gate."
    u32 row, i;
    cb_find_slot(b, G_MUL_EXT, c0, 0, kMulExtOps, &row, &i);
    connect_ext(b, m0, wire_ext(row, 6 * i));
    connect_ext(b, m1, wire_ext(row, 6 * i + 2));
    g.kind = GEN_MUL_EXT;
    g.row = row, g.i = i, g.c0 = c0;
    res = wire_ext(row, 6 * i + 4);
  } else {
    u32 row, i;
    cb_find_slot(b, G_ARITHMETIC_EXT, c0, c1, kArithExtOps, &row, &i);
    connect_ext(b, m0, wire_ext(row, 8 * i));
    connect_ext(b, m1, wire_ext(row, 8 * i + 2));
    connect_ext(b, ad, wire_ext(row, 8 * i + 4));
    g.kind = GEN_ARITH_EXT;
    g.row = row, g.i = i, g.c0 = c0, g.c1 = c1;
    res = wire_ext(row, 8 * i + 6);
  }
  b->gens.push_back(g);
  b->ext_arith_results[op] = res;
  return res;
}
Ext add_ext(B* b, Ext x, Ext y) { return arithmetic_ext(b, 1, 1, one_ext(b), x, y); }
Ext sub_ext(B* b, Ext x, Ext y) { return arithmetic_ext(b, 1, kP - 1, one_ext(b), x, y); }
Ext mul_ext(B* b, Ext x, Ext y) { return arithmetic_ext(b, 1, 0, x, y, zero_ext(b)); }
Ext mul_add_ext(B* b, Ext x, Ext y, Ext z) { return arithmetic_ext(b, 1, 1, x, y, z); }
Ext mul_sub_ext(B* b, Ext x, Ext y, Ext z) { return arithmetic_ext(b, 1, kP - 1, x, y, z); }
Ext mul_many_ext(B* b, const std::vector<Ext>& terms) {
  Ext acc = one_ext(b);
  for (const Ext& t : terms) acc = mul_ext(b, acc, t);
  return acc;
}
Ext scalar_mul_ext(B* b, u64 a, Ext x) { return mul_ext(b, to_ext(b, a), x); }
Ext mul_const_ext(B* b, u64 c, Ext x) { return mul_ext(b, ext_const(b, {c % kP, 0}), x); }
Ext exp_power_of_2_ext(B* b, Ext base, unsigned power_log) {
  for (unsigned k = 0; k < power_log; ++k) base = mul_ext(b, base, base);
  return base;
}
Ext exp_u64_ext(B* b, Ext base, u64 exponent) {
  if (exponent == 0) return one_ext(b);
  if (exponent == 1) return base;
  if (exponent == 2) return mul_ext(b, base, base);
  if (exponent == 3) return mul_many_ext(b, {base, base, base});
  Ext current = base, product = one_ext(b);
  for (unsigned j = 0; (exponent >> j) != 0; ++j) {
    if (j != 0) current = mul_ext(b, current, current);
    if ((exponent >> j) & 1) product = mul_ext(b, product, current);
  }
  return product;
}
// x / y + z: the inverse of y is a generated witness (QuotientGeneratorExtension) pinned by y * inv == 1
Ext div_add_ext(B* b, Ext x, Ext y, Ext z) {
  const Ext inv{cb_virtual(b), cb_virtual(b)};
  const Ext o = one_ext(b);
  Gen g{};
  g.kind = GEN_QUOTIENT_EXT;
  g.t = {o[0], o[1], y[0], y[1], inv[0], inv[1]};
  b->gens.push_back(g);
  connect_ext(b, mul_ext(b, y, inv), o);
  return mul_add_ext(b, x, inv, z);
}
Ext div_ext(B* b, Ext x, Ext y) { return div_add_ext(b, x, y, zero_ext(b)); }

// util/reducing.rs ReducingFactorTarget: sum_i terms[i] * base^i
Ext reduce_arithmetic(B* b, Ext base, const std::vector<Ext>& terms) {
  Ext acc = zero_ext(b);
  for (size_t k = terms.size(); k-- > 0;) acc = mul_add_ext(b, base, acc, terms[k]);
  return acc;
}
Ext reduce_base(B* b, Ext base, const std::vector<u64>& terms) {
  if (terms.size() <= kArithExtOps + 1) {  // "For small reductions, use an arithmetic gate."
    std::vector<Ext> e;
    for (u64 t : terms) e.push_back(to_ext(b, t));
    return reduce_arithmetic(b, base, e);
  }
  std::vector<u64> rev(terms);
  while (rev.size() % kReducingCoeffs) rev.push_back(zero(b));
  std::reverse(rev.begin(), rev.end());
  Ext acc = zero_ext(b);
  for (size_t off = 0; off < rev.size(); off += kReducingCoeffs) {
    const u32 row = cb_add_gate(b, G_REDUCING);
    connect_ext(b, base, wire_ext(row, 2));
    connect_ext(b, acc, wire_ext(row, 4));
    for (u32 j = 0; j < kReducingCoeffs; ++j) connect(b, rev[off + j], wire_t(row, 6 + j));
    acc = wire_ext(row, 0);
  }
  return acc;
}
Ext reduce_ext(B* b, Ext base, const std::vector<Ext>& terms) {
  if (terms.size() <= kArithExtOps + 1) return reduce_arithmetic(b, base, terms);
  std::vector<Ext> rev(terms);
  const Ext z = zero_ext(b);
  while (rev.size() % kReducingExtCoeffs) rev.push_back(z);
  std::reverse(rev.begin(), rev.end());
  Ext acc = z;
  for (size_t off = 0; off < rev.size(); off += kReducingExtCoeffs) {
    const u32 row = cb_add_gate(b, G_REDUCING_EXT);
    connect_ext(b, base, wire_ext(row, 2));
    connect_ext(b, acc, wire_ext(row, 4));
    for (u32 j = 0; j < kReducingExtCoeffs; ++j) connect_ext(b, rev[off + j], wire_ext(row, 6 + 2 * j));
    acc = wire_ext(row, 0);
  }
  return acc;
}
Ext reducing_shift(B* b, Ext base, u64 count, Ext x) {
  const Ext z = zero_ext(b);
  const Ext e = x == z ? z : exp_u64_ext(b, base, count);
  return mul_ext(b, e, x);
}

// gadgets/random_access.rs (vectors of 16: the Merkle caps and the FRI cosets of arity 16)
u64 random_access(B* b, u64 access_index, const std::vector<u64>& v) {
  if (v.size() == 1) return v[0];
  const u64 claimed = cb_virtual(b);
  u32 row, copy;
  cb_find_slot(b, G_RANDOM_ACCESS, 0, 0, kRaCopies, &row, &copy);
  const u32 base = 18 * copy;
  for (u32 i = 0; i < 16; ++i) connect(b, v[i], wire_t(row, base + 2 + i));
  connect(b, access_index, wire_t(row, base));
  connect(b, claimed, wire_t(row, base + 1));
  Gen g{};
  g.kind = GEN_RANDOM_ACCESS;
  g.row = row, g.i = copy;
  b->gens.push_back(g);
  return claimed;
}
Ext random_access_ext(B* b, u64 access_index, const std::vector<Ext>& v) {
  Ext out;
  for (int k = 0; k < 2; ++k) {
    std::vector<u64> col;
    for (const Ext& e : v) col.push_back(e[k]);
    out[k] = random_access(b, access_index, col);
  }
  return out;
}
// gadgets/interpolation.rs interpolate_coset with CosetInterpolationGate::with_max_degree(4, 8)
Ext interpolate_coset(B* b, u64 coset_shift, const std::vector<Ext>& values, Ext evaluation_point) {
  const u32 row = cb_add_gate(b, G_COSET_INTERPOLATION);
  connect(b, coset_shift, wire_t(row, 0));
  for (u32 i = 0; i < 16; ++i) connect_ext(b, values[i], wire_ext(row, 1 + 2 * i));
  connect_ext(b, evaluation_point, wire_ext(row, 33));
  return wire_ext(row, 35);
}

// hash gadgets
void permute(B* b, u64 (&state)[12]) { (void)cb_permute_swapped(b, state, zero(b)); }
void hash_no_pad(B* b, const std::vector<u64>& in, u64 (&out)[4]) { (void)cb_hash_no_pad(b, in.data(), in.size(), out); }
void hash_or_noop(B* b, const std::vector<u64>& in, u64 (&out)[4]) {
  if (in.size() <= 4) {
    for (size_t k = 0; k < 4; ++k) out[k] = k < in.size() ? in[k] : zero(b);
    return;
  }
  hash_no_pad(b, in, out);
}

// ------------------------------------------------------------------------------------------------ proof targets
struct Layout {
  size_t wires_cap = 0, zs_cap = 64, quotient_cap = 128;
  size_t constants, n_constants, sigmas, wires, zs, zs_next, pps, quotient, commit_caps, final_poly, final_len, pow_witness, public_inputs, end;
  struct Q {
    size_t leaves[4], widths[4], siblings[4], plen0;
    std::vector<std::array<size_t, 4>> steps;  // evals offset, arity, siblings offset, path length
  };
  std::vector<Q> queries;
};
Layout layout_of(const p2mt_common_data& cd) {
  Layout L;
  const size_t capw = 4u << kCapHeight;
  size_t off = 3 * capw;
  L.n_constants = cd.num_selectors + kNumConsts;
  L.constants = off, off += 2 * L.n_constants;
  L.sigmas = off, off += 2 * kNumRouted;
  L.wires = off, off += 2 * kNumWires;
  L.zs = off, off += 2 * kNumCh;
  L.zs_next = off, off += 2 * kNumCh;
  L.pps = off, off += 2 * kNumCh * kNumProds;
  L.quotient = off, off += 2 * kNumQuot;
  L.commit_caps = off, off += capw * cd.fri.num_reductions;
  const size_t widths[4] = {L.n_constants + kNumRouted, kNumWires, kNumZs, kNumQuot};
  const unsigned lde_bits = cd.degree_bits + kRateBits;
  unsigned total_arity = 0;
  for (u32 l = 0; l < cd.fri.num_reductions; ++l) total_arity += cd.fri.reduction_arity_bits[l];
  for (u32 q = 0; q < cd.fri.num_query_rounds; ++q) {
    Layout::Q Q;
    size_t plen = lde_bits - kCapHeight;
    Q.plen0 = plen;
    for (int o = 0; o < 4; ++o) {
      Q.leaves[o] = off, Q.widths[o] = widths[o], Q.siblings[o] = off + widths[o];
      off += widths[o] + 4 * plen;
    }
    for (u32 l = 0; l < cd.fri.num_reductions; ++l) {
      const size_t ar = (size_t)1 << cd.fri.reduction_arity_bits[l];
      plen -= cd.fri.reduction_arity_bits[l];
      Q.steps.push_back({off, ar, off + 2 * ar, plen});
      off += 2 * ar + 4 * plen;
    }
    L.queries.push_back(Q);
  }
  L.final_len = (size_t)1 << (cd.degree_bits - total_arity);
  L.final_poly = off, off += 2 * L.final_len;
  L.pow_witness = off, off += 1;
  L.public_inputs = off;
  L.end = off + cd.num_public_inputs;
  return L;
}

struct Hash4 {
  u64 e[4];
};
std::vector<Hash4> hashes_at(const u64* flat, size_t off, size_t n) {
  std::vector<Hash4> out(n);
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 4; ++k) out[i].e[k] = flat[off + 4 * i + k];
  return out;
}
std::vector<Ext> exts_at(const u64* flat, size_t off, size_t n) {
  std::vector<Ext> out(n);
  for (size_t i = 0; i < n; ++i) out[i] = Ext{flat[off + 2 * i], flat[off + 2 * i + 1]};
  return out;
}

// ------------------------------------------------------------------------------------------------ RecursiveChallenger
struct Challenger {
  B* b;
  u64 state[12];
  std::vector<u64> inp, out;
  explicit Challenger(B* bb) : b(bb) {
    for (auto& s : state) s = zero(bb);
  }
  void observe(u64 t) {
    out.clear();  // any buffered outputs are now invalid
    inp.push_back(t);
  }
  void observe_hash(const u64* h) {
    for (int k = 0; k < 4; ++k) observe(h[k]);
  }
  void observe_cap(const std::vector<Hash4>& cap) {
    for (const auto& h : cap) observe_hash(h.e);
  }
  void observe_exts(const std::vector<Ext>& v) {
    for (const Ext& e : v) observe(e[0]), observe(e[1]);
  }
  void absorb() {
    if (inp.empty()) return;
    for (size_t off = 0; off < inp.size(); off += 8) {
      for (size_t k = 0; k < 8 && off + k < inp.size(); ++k) state[k] = inp[off + k];  // overwrite mode
      permute(b, state);
    }
    out.assign(state, state + 8);
    inp.clear();
  }
  u64 get() {
    absorb();
    if (out.empty()) {
      permute(b, state);
      out.assign(state, state + 8);
    }
    const u64 t = out.back();
    out.pop_back();
    return t;
  }
  Ext get_ext() {
    const u64 a = get(), c = get();
    return Ext{a, c};
  }
};

// ------------------------------------------------------------------------------------------------ gate constraints in-circuit
std::vector<Ext> mds_layer_circuit(B* b, const std::vector<Ext>& state) {  // Poseidon::mds_layer_circuit through the PoseidonMdsGate
  const u32 row = cb_add_gate(b, G_POSEIDON_MDS);
  std::vector<Ext> out(12);
  for (u32 i = 0; i < 12; ++i) {
    connect_ext(b, state[i], wire_ext(row, 2 * i));
    out[i] = wire_ext(row, 24 + 2 * i);
  }
  return out;
}
void constant_layer_circuit(B* b, std::vector<Ext>& state, unsigned round_ctr) {
  for (u32 i = 0; i < 12; ++i) state[i] = add_ext(b, state[i], ext_const(b, {POSEIDON_RC[i + 12 * round_ctr], 0}));
}
// PoseidonGate::eval_unfiltered_circuit (use_mds_gate): 123 constraints over the opened wires
std::vector<Ext> poseidon_gate_eval_circuit(B* b, const std::vector<Ext>& w) {
  std::vector<Ext> cons;
  const Ext swap = w[24];
  cons.push_back(mul_sub_ext(b, swap, swap, swap));
  for (u32 i = 0; i < 4; ++i) cons.push_back(mul_sub_ext(b, swap, sub_ext(b, w[i + 4], w[i]), w[25 + i]));
  std::vector<Ext> state(12, zero_ext(b));
  for (u32 i = 0; i < 4; ++i) {
    state[i] = add_ext(b, w[i], w[25 + i]);
    state[i + 4] = sub_ext(b, w[i + 4], w[25 + i]);
  }
  for (u32 i = 8; i < 12; ++i) state[i] = w[i];
  unsigned round_ctr = 0;
  for (u32 r = 0; r < 4; ++r) {
    constant_layer_circuit(b, state, round_ctr);
    if (r != 0)
      for (u32 i = 0; i < 12; ++i) {
        const Ext sbox_in = w[29 + 12 * (r - 1) + i];
        cons.push_back(sub_ext(b, state[i], sbox_in));
        state[i] = sbox_in;
      }
    for (u32 i = 0; i < 12; ++i) state[i] = exp_u64_ext(b, state[i], 7);
    state = mds_layer_circuit(b, state);
    ++round_ctr;
  }
  for (u32 r = 0; r < 22; ++r) {
    constant_layer_circuit(b, state, round_ctr);
    const Ext sbox_in = w[65 + r];
    cons.push_back(sub_ext(b, state[0], sbox_in));
    state[0] = exp_u64_ext(b, sbox_in, 7);
    state = mds_layer_circuit(b, state);
    ++round_ctr;
  }
  for (u32 r = 0; r < 4; ++r) {
    constant_layer_circuit(b, state, round_ctr);
    for (u32 i = 0; i < 12; ++i) {
      const Ext sbox_in = w[87 + 12 * r + i];
      cons.push_back(sub_ext(b, state[i], sbox_in));
      state[i] = sbox_in;
    }
    for (u32 i = 0; i < 12; ++i) state[i] = exp_u64_ext(b, state[i], 7);
    state = mds_layer_circuit(b, state);
    ++round_ctr;
  }
  for (u32 i = 0; i < 12; ++i) cons.push_back(sub_ext(b, state[i], w[12 + i]));
  return cons;
}

int gate_eval_unfiltered_circuit(B* b, u32 kind, const std::vector<Ext>& gc, const std::vector<Ext>& w, const u64* pi_hash,
                                 std::vector<Ext>* out) {
  out->clear();
  switch (kind) {
    case G_NOOP: return P2MT_OK;
    case G_CONSTANT:
      for (u32 i = 0; i < kNumConsts; ++i) out->push_back(sub_ext(b, gc[i], w[i]));
      return P2MT_OK;
    case G_PUBLIC_INPUT:
      for (u32 i = 0; i < 4; ++i) out->push_back(sub_ext(b, w[i], to_ext(b, pi_hash[i])));
      return P2MT_OK;
    case G_ARITHMETIC:
      for (u32 i = 0; i < kNumOps; ++i) {
        const Ext scaled_mul = mul_many_ext(b, {gc[0], w[4 * i], w[4 * i + 1]});
        const Ext computed = mul_add_ext(b, gc[1], w[4 * i + 2], scaled_mul);
        out->push_back(sub_ext(b, w[4 * i + 3], computed));
      }
      return P2MT_OK;
    case G_POSEIDON: *out = poseidon_gate_eval_circuit(b, w); return P2MT_OK;
    default:
      return p2mt::fail(P2MT_EINVAL, "verify_proof: the inner circuit contains a gate type whose in-circuit evaluation is not built "
                                     "(the reference's inner circuits use Noop / Constant / PublicInput / Arithmetic / Poseidon only)");
  }
}

// plonk/vanishing_poly.rs eval_vanishing_poly_circuit
int eval_vanishing_poly_circuit(B* b, const p2mt_common_data& cd, Ext x, Ext x_pow_deg, const std::vector<Ext>& constants,
                                const std::vector<Ext>& sigmas, const std::vector<Ext>& wires, const std::vector<Ext>& zs,
                                const std::vector<Ext>& zs_next, const std::vector<Ext>& pps, const u64* pi_hash, const u64* betas,
                                const u64* gammas, const u64* alphas, std::vector<Ext>* out) {
  // evaluate_gate_constraints_circuit
  u32 num_gate_constraints = 0;
  for (u32 g = 0; g < cd.n_kinds; ++g) num_gate_constraints = std::max(num_gate_constraints, kGateNumConstraints[cd.kind[g]]);
  std::vector<Ext> constraint_terms(num_gate_constraints, zero_ext(b));
  const std::vector<Ext> gate_consts(constants.begin() + cd.num_selectors, constants.end());
  for (u32 g = 0; g < cd.n_kinds; ++g) {
    const Ext s = constants[cd.sel[g]];
    std::vector<Ext> terms;
    for (u32 j = cd.gs[g]; j < cd.ge[g]; ++j)
      if (j != g) terms.push_back(sub_ext(b, ext_const(b, {j, 0}), s));
    if (cd.num_selectors > 1) terms.push_back(sub_ext(b, ext_const(b, {kUnusedSelector, 0}), s));
    const Ext filter = mul_many_ext(b, terms);
    std::vector<Ext> mine;
    P2MT_TRY(gate_eval_unfiltered_circuit(b, cd.kind[g], gate_consts, wires, pi_hash, &mine));
    for (size_t j = 0; j < mine.size(); ++j) constraint_terms[j] = mul_add_ext(b, filter, mine[j], constraint_terms[j]);
  }
  // eval_l_0_circuit: (x^n - 1) / (n (x - 1))
  Ext l_0_x;
  {
    const Ext o = one_ext(b);
    const Ext neg_one = to_ext(b, cb_constant(b, kP - 1));
    const Ext eval_zero_poly = sub_ext(b, x_pow_deg, o);
    const u64 n = (1ull << cd.degree_bits) % kP;
    const Ext denominator = arithmetic_ext(b, n, n, x, o, neg_one);
    l_0_x = div_ext(b, eval_zero_poly, denominator);
  }
  std::vector<Ext> s_ids;
  for (u32 j = 0; j < kNumRouted; ++j) s_ids.push_back(scalar_mul_ext(b, cb_constant(b, cd.k_is[j]), x));
  std::vector<Ext> z1_terms, pp_terms;
  for (u32 i = 0; i < kNumCh; ++i) {
    const Ext z_x = zs[i], z_gx = zs_next[i];
    z1_terms.push_back(mul_sub_ext(b, l_0_x, z_x, l_0_x));
    std::vector<Ext> nums, dens;
    for (u32 j = 0; j < kNumRouted; ++j) {
      const Ext beta_ext = to_ext(b, betas[i]), gamma_ext = to_ext(b, gammas[i]);
      const Ext wire_value_plus_gamma = add_ext(b, wires[j], gamma_ext);
      nums.push_back(mul_add_ext(b, beta_ext, s_ids[j], wire_value_plus_gamma));
      dens.push_back(mul_add_ext(b, beta_ext, sigmas[j], wire_value_plus_gamma));
    }
    std::vector<Ext> accs{z_x};
    accs.insert(accs.end(), pps.begin() + i * kNumProds, pps.begin() + (i + 1) * kNumProds);
    accs.push_back(z_gx);
    for (u32 q = 0; q + 1 < accs.size(); ++q) {  // check_partial_products_circuit
      const Ext nume = mul_many_ext(b, std::vector<Ext>(nums.begin() + q * kQF, nums.begin() + (q + 1) * kQF));
      const Ext deno = mul_many_ext(b, std::vector<Ext>(dens.begin() + q * kQF, dens.begin() + (q + 1) * kQF));
      const Ext next_acc_deno = mul_ext(b, accs[q + 1], deno);
      pp_terms.push_back(mul_sub_ext(b, accs[q], nume, next_acc_deno));
    }
  }
  std::vector<Ext> terms(z1_terms);
  terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
  terms.insert(terms.end(), constraint_terms.begin(), constraint_terms.end());
  out->clear();
  for (u32 i = 0; i < kNumCh; ++i) out->push_back(reduce_ext(b, to_ext(b, alphas[i]), terms));
  return P2MT_OK;
}

// hash/merkle_proofs.rs verify_merkle_proof_to_cap_with_cap_index
void verify_merkle_proof_to_cap(B* b, const std::vector<u64>& leaf_data, const std::vector<u64>& leaf_index_bits, u64 cap_index,
                                const std::vector<Hash4>& cap, const std::vector<Hash4>& siblings) {
  u64 state[4];
  hash_or_noop(b, leaf_data, state);
  for (size_t k = 0; k < siblings.size() && k < leaf_index_bits.size(); ++k) {
    u64 perm[12];
    for (int j = 0; j < 4; ++j) perm[j] = state[j], perm[4 + j] = siblings[k].e[j], perm[8 + j] = zero(b);
    (void)cb_permute_swapped(b, perm, leaf_index_bits[k]);
    for (int j = 0; j < 4; ++j) state[j] = perm[j];
  }
  for (int i = 0; i < 4; ++i) {
    std::vector<u64> col;
    for (const auto& h : cap) col.push_back(h.e[i]);
    connect(b, random_access(b, cap_index, col), state[i]);
  }
}

}  // namespace

// ==================================================================================================== C ABI
extern "C" int p2mt_cb_add_virtual_proof_with_pis(p2mt_circuit_builder* b, const p2mt_circuit_data* inner, p2mt_target* out, size_t out_len) {
  return p2mt::abi_guard([&]() -> int {
  if (!b || !inner || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  p2mt_common_data cd;
  P2MT_TRY(p2mt_circuit_common_data(inner, &cd));
  if (out_len != cd.proof_len) return p2mt::fail(P2MT_EINVAL, "add_virtual_proof_with_pis: out_len must be the inner circuit's proof_len");
  for (size_t k = 0; k < out_len; ++k) out[k] = cb_virtual(b);
  return P2MT_OK;
  });
}

extern "C" int p2mt_cb_add_virtual_verifier_data(p2mt_circuit_builder* b, unsigned cap_height, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!b || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (cap_height != kCapHeight) return p2mt::fail(P2MT_EINVAL, "add_virtual_verifier_data: cap_height must be 4 (standard_recursion_config)");
  for (size_t k = 0; k < 64 + 4; ++k) out[k] = cb_virtual(b);
  return P2MT_OK;
  });
}

extern "C" int p2mt_cb_verify_proof(p2mt_circuit_builder* b, const p2mt_target* proof_with_pis, size_t proof_len,
                                    const p2mt_target* verifier_data, const p2mt_circuit_data* inner) {
  return p2mt::abi_guard([&]() -> int {
  if (!b || !proof_with_pis || !verifier_data || !inner) return p2mt::fail(P2MT_EINVAL, "null pointer");
  p2mt_common_data cd;
  P2MT_TRY(p2mt_circuit_common_data(inner, &cd));
  if (proof_len != cd.proof_len) return p2mt::fail(P2MT_EINVAL, "verify_proof: proof target does not match the inner circuit");
  for (size_t k = 0; k < proof_len; ++k) P2MT_TRY(cb_check(b, proof_with_pis[k], true));
  for (size_t k = 0; k < 68; ++k) P2MT_TRY(cb_check(b, verifier_data[k], true));
  const Layout L = layout_of(cd);
  if (L.end != proof_len) return p2mt::fail(P2MT_EINVAL, "internal: proof layout mismatch");
  for (u32 l = 0; l < cd.fri.num_reductions; ++l)
    if (cd.fri.reduction_arity_bits[l] != 4) return p2mt::fail(P2MT_EINVAL, "verify_proof: FRI arity must be 16 (ConstantArityBits(4, 5))");
  const u64* flat = proof_with_pis;
  const std::vector<Hash4> wires_cap = hashes_at(flat, L.wires_cap, 16), zs_cap = hashes_at(flat, L.zs_cap, 16),
                           quotient_cap = hashes_at(flat, L.quotient_cap, 16), cs_cap = hashes_at(verifier_data, 0, 16);
  const std::vector<Ext> constants = exts_at(flat, L.constants, L.n_constants), sigmas = exts_at(flat, L.sigmas, kNumRouted),
                         wires = exts_at(flat, L.wires, kNumWires), zs = exts_at(flat, L.zs, kNumCh),
                         zs_next = exts_at(flat, L.zs_next, kNumCh), pps = exts_at(flat, L.pps, kNumCh * kNumProds),
                         quotient = exts_at(flat, L.quotient, kNumQuot), final_poly = exts_at(flat, L.final_poly, L.final_len);
  std::vector<std::vector<Hash4>> commit_caps;
  for (u32 l = 0; l < cd.fri.num_reductions; ++l) commit_caps.push_back(hashes_at(flat, L.commit_caps + 64 * l, 16));
  const std::vector<u64> public_inputs(flat + L.public_inputs, flat + L.public_inputs + cd.num_public_inputs);

  u64 pi_hash[4];
  hash_no_pad(b, public_inputs, pi_hash);
  // ---- get_challenges
  Challenger ch(b);
  ch.observe_hash(verifier_data + 64);
  ch.observe_hash(pi_hash);
  ch.observe_cap(wires_cap);
  u64 betas[kNumCh], gammas[kNumCh], alphas[kNumCh];
  for (auto& t : betas) t = ch.get();
  for (auto& t : gammas) t = ch.get();
  ch.observe_cap(zs_cap);
  for (auto& t : alphas) t = ch.get();
  ch.observe_cap(quotient_cap);
  const Ext zeta = ch.get_ext();
  std::vector<Ext> zeta_batch;  // OpeningSetTarget::to_fri_openings
  for (const auto* v : {&constants, &sigmas, &wires, &zs, &pps, &quotient}) zeta_batch.insert(zeta_batch.end(), v->begin(), v->end());
  ch.observe_exts(zeta_batch);
  ch.observe_exts(zs_next);
  const Ext fri_alpha = ch.get_ext();
  std::vector<Ext> fri_betas;
  for (const auto& cap : commit_caps) {
    ch.observe_cap(cap);
    fri_betas.push_back(ch.get_ext());
  }
  ch.observe_exts(final_poly);
  ch.observe(flat[L.pow_witness]);
  const u64 fri_pow_response = ch.get();
  std::vector<u64> fri_query_indices;
  for (u32 q = 0; q < cd.fri.num_query_rounds; ++q) fri_query_indices.push_back(ch.get());
  // ---- verify_proof_with_challenges
  const Ext o = one_ext(b);
  const Ext zeta_pow_deg = exp_power_of_2_ext(b, zeta, cd.degree_bits);
  std::vector<Ext> vanishing;
  P2MT_TRY(eval_vanishing_poly_circuit(b, cd, zeta, zeta_pow_deg, constants, sigmas, wires, zs, zs_next, pps, pi_hash, betas, gammas,
                                       alphas, &vanishing));
  const Ext z_h_zeta = sub_ext(b, zeta_pow_deg, o);
  for (u32 i = 0; i < kNumCh; ++i) {
    const Ext recombined = reduce_ext(b, zeta_pow_deg, std::vector<Ext>(quotient.begin() + i * kQF, quotient.begin() + (i + 1) * kQF));
    connect_ext(b, vanishing[i], mul_ext(b, z_h_zeta, recombined));
  }
  const std::vector<Hash4>* caps[4] = {&cs_cap, &wires_cap, &zs_cap, &quotient_cap};
  const Ext zeta_next = mul_const_ext(b, h_root_of_unity(cd.degree_bits), zeta);  // get_fri_instance_target
  // ---- verify_fri_proof
  (void)split_le(b, fri_pow_response, 64 - cd.fri.proof_of_work_bits);  // fri_verify_proof_of_work: assert_leading_zeros -> range_check
  const Ext reduced_openings[2] = {reduce_ext(b, fri_alpha, zeta_batch), reduce_ext(b, fri_alpha, zs_next)};
  const unsigned n_log = cd.degree_bits + kRateBits;
  for (u32 q = 0; q < cd.fri.num_query_rounds; ++q) {  // fri_verifier_query_round
    const Layout::Q& Q = L.queries[q];
    std::vector<u64> x_index_bits = split_le(b, fri_query_indices[q], 64);  // low_bits(x_index, n_log, F::BITS)
    x_index_bits.resize(n_log);
    const u64 cap_index = le_sum(b, std::vector<u64>(x_index_bits.end() - kCapHeight, x_index_bits.end()));
    std::vector<std::vector<u64>> leaves(4);
    for (int oi = 0; oi < 4; ++oi) {  // fri_verify_initial_proof
      leaves[oi].assign(flat + Q.leaves[oi], flat + Q.leaves[oi] + Q.widths[oi]);
      verify_merkle_proof_to_cap(b, leaves[oi], x_index_bits, cap_index, *caps[oi], hashes_at(flat, Q.siblings[oi], Q.plen0));
    }
    u64 subgroup_x;
    {
      const u64 g = cb_constant(b, 7);  // F::coset_shift()
      const u64 phi = exp_from_bits_const_base(b, h_root_of_unity(n_log), std::vector<u64>(x_index_bits.rbegin(), x_index_bits.rend()));
      subgroup_x = mul(b, g, phi);
    }
    Ext old_eval;
    {  // fri_combine_initial
      const Ext sx = to_ext(b, subgroup_x);
      std::vector<u64> all_evals, next_evals(leaves[2].begin(), leaves[2].begin() + kNumCh);
      for (int oi = 0; oi < 4; ++oi) all_evals.insert(all_evals.end(), leaves[oi].begin(), leaves[oi].end());
      Ext total = zero_ext(b);
      u64 count = 0;
      const std::vector<u64>* batches[2] = {&all_evals, &next_evals};
      const Ext points[2] = {zeta, zeta_next};
      for (int bi = 0; bi < 2; ++bi) {
        const Ext reduced_evals = reduce_base(b, fri_alpha, *batches[bi]);
        count += batches[bi]->size();
        const Ext numerator = sub_ext(b, reduced_evals, reduced_openings[bi]);
        const Ext denominator = sub_ext(b, sx, points[bi]);
        total = reducing_shift(b, fri_alpha, count, total);
        count = 0;
        total = div_add_ext(b, numerator, denominator, total);
      }
      old_eval = mul_ext(b, total, sx);  // "Multiply the final polynomial by X" (plonky2 #436)
    }
    for (u32 l = 0; l < cd.fri.num_reductions; ++l) {
      const unsigned arity_bits = cd.fri.reduction_arity_bits[l];
      const std::vector<Ext> evals = exts_at(flat, Q.steps[l][0], Q.steps[l][1]);
      const std::vector<u64> coset_index_bits(x_index_bits.begin() + arity_bits, x_index_bits.end());
      const std::vector<u64> within_bits(x_index_bits.begin(), x_index_bits.begin() + arity_bits);
      const u64 x_index_within_coset = le_sum(b, within_bits);
      connect_ext(b, random_access_ext(b, x_index_within_coset, evals), old_eval);
      {  // compute_evaluation
        const u64 g = h_root_of_unity(arity_bits), g_inv = h_pow(g, (1u << arity_bits) - 1);
        std::vector<Ext> ev(evals.size());
        for (size_t i = 0; i < evals.size(); ++i) {  // reverse_index_bits_in_place
          size_t r = 0;
          for (unsigned k = 0; k < arity_bits; ++k) r |= ((i >> k) & 1) << (arity_bits - 1 - k);
          ev[i] = evals[r];
        }
        const u64 start = exp_from_bits_const_base(b, g_inv, std::vector<u64>(within_bits.rbegin(), within_bits.rend()));
        const u64 coset_start = mul(b, start, subgroup_x);
        old_eval = interpolate_coset(b, coset_start, ev, fri_betas[l]);
      }
      std::vector<u64> flat_evals;
      for (const Ext& e : evals) flat_evals.push_back(e[0]), flat_evals.push_back(e[1]);
      verify_merkle_proof_to_cap(b, flat_evals, coset_index_bits, cap_index, commit_caps[l], hashes_at(flat, Q.steps[l][2], Q.steps[l][3]));
      subgroup_x = exp_power_of_2(b, subgroup_x, arity_bits);
      x_index_bits = coset_index_bits;
    }
    connect_ext(b, reduce_ext(b, to_ext(b, subgroup_x), final_poly), old_eval);  // final_poly.eval_scalar
  }
  return P2MT_OK;
  });
}

// pw.set_proof_with_pis_target(&target, &proof) (mmr_plonky2_verifier_1_recursion.rs:201): target[i] <- word[i]
extern "C" int p2mt_pw_set_proof_with_pis_target(p2mt_partial_witness* pw, const p2mt_target* proof_target, const uint64_t* proof_words,
                                                 size_t proof_len) {
  return p2mt::abi_guard([&]() -> int {
  if (!pw || !proof_target || !proof_words) return p2mt::fail(P2MT_EINVAL, "null pointer");
  for (size_t k = 0; k < proof_len; ++k) P2MT_TRY(p2mt_pw_set_target(pw, proof_target[k], proof_words[k]));
  return P2MT_OK;
  });
}

// pw.set_verifier_data_target(&target, &inner.verifier_only) (:202): constants_sigmas_cap and circuit_digest of the inner circuit
extern "C" int p2mt_pw_set_verifier_data_target(p2mt_partial_witness* pw, const p2mt_target* verifier_data_target,
                                                const p2mt_circuit_data* inner) {
  return p2mt::abi_guard([&]() -> int {
  if (!pw || !verifier_data_target || !inner) return p2mt::fail(P2MT_EINVAL, "null pointer");
  p2mt_common_data cd;
  P2MT_TRY(p2mt_circuit_common_data(inner, &cd));
  for (size_t k = 0; k < 64; ++k) P2MT_TRY(p2mt_pw_set_target(pw, verifier_data_target[k], cd.cs_cap[k]));
  for (size_t k = 0; k < 4; ++k) P2MT_TRY(p2mt_pw_set_target(pw, verifier_data_target[64 + k], cd.digest[k]));
  return P2MT_OK;
  });
}
