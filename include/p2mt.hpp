// p2mt.hpp -- C++ host-side mirror of the reference's public Rust API over the C ABI (p2mt.h).
//
// The reference is compiled code (Rust) and this image has no Rust toolchain, so the host side above the C ABI
// is written in C++ with the reference's names, argument meaning and error behaviour:
//   src/simple_merkle_tree/simple_merkle_tree.rs : MerkleTree::{build, get_merkle_proof, get_in_between_hashes},
//                                                  verify_merkle_proof
//   src/mmr/merkle_mountain_ranges.rs            : MMR::{new_, add_leaf, bagging_the_peaks, get_peaks, get_proof,
//                                                  get_proof_normal_index, get_subtree_proof_elm}, MMR_proof::verify,
//                                                  get_mmr_index, get_heights_bitmap_for_mmr_size
// and, for the prover the reference drives through CircuitData::prove (mmr_plonky2_verifier.rs:148,
// mmr_plonky2_verifier_1_recursion.rs:192,218), plonky2's PolynomialBatch, Challenger, prove_openings (FRI), and
// CircuitBuilder / PartialWitness / CircuitData::prove with the reference's own circuit code on top
//   src/mmr/common.rs : equal, or_list, pick_hash;  src/mmr/mmr_plonky2_verifier.rs : verify_mmr_proof_circuit.
// Where the reference panics (assert!/unwrap/log2_strict) this throws p2mt::panic carrying the status code.
// Getters take `this` by const reference instead of consuming `self` (Quirk Q7: callers no longer clone).
// All hashing happens in libp2mt_hip.so on the GPU; this header only moves buffers.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "p2mt.h"

namespace p2mt {

using GoldilocksField = std::uint64_t;  // transparent u64, canonical on output
constexpr std::uint64_t GOLDILOCKS_FIELD_ORDER = P2MT_GOLDILOCKS_FIELD_ORDER;  // src/mmr/common.rs:3

struct HashOut {
  std::array<std::uint64_t, 4> elements{};
  bool operator==(const HashOut& o) const { return elements == o.elements; }
  bool operator!=(const HashOut& o) const { return !(*this == o); }
};
static_assert(sizeof(HashOut) == 32, "HashOut must be the 32-byte record of the C ABI");

struct panic : std::runtime_error {
  int code;
  panic(int c, const std::string& m) : std::runtime_error("p2mt panic (" + std::to_string(c) + "): " + m), code(c) {}
};
inline void check(int rc) {
  if (rc != P2MT_OK) throw panic(rc, p2mt_last_error());
}

// ---------------------------------------------------------------- simple_merkle_tree.rs
struct MerkleTree {
  std::size_t count_levels = 0;
  std::vector<std::vector<HashOut>> tree;  // levels 0 .. count_levels-1
  HashOut root;

  static MerkleTree build(const std::vector<GoldilocksField>& leaves) {  // :28-51
    const std::size_t n = leaves.size();
    std::vector<HashOut> flat(n >= 2 ? 2 * n - 2 : 1);
    MerkleTree t;
    check(p2mt_merkle_build_pow2(leaves.data(), n, flat[0].elements.data(), t.root.elements.data()));
    for (std::size_t cnt = n, off = 0; cnt >= 2; off += cnt, cnt /= 2) {
      t.tree.emplace_back(flat.begin() + off, flat.begin() + off + cnt);
      ++t.count_levels;
    }
    return t;
  }

  std::vector<HashOut> get_merkle_proof(std::size_t leaf_index) const {  // :55-74
    if (tree.empty() || leaf_index >= tree[0].size()) throw panic(P2MT_EINVAL, "assert!(leaf_index < self.tree[0].len())");
    std::vector<HashOut> proof;
    std::size_t idx = leaf_index;
    for (std::size_t i = 0; i < count_levels; ++i, idx /= 2) proof.push_back(tree[i][idx ^ 1]);
    return proof;
  }

  std::vector<HashOut> get_in_between_hashes(std::size_t leaf_index) const {  // :76-86
    if (tree.empty() || leaf_index >= tree[0].size()) throw panic(P2MT_EINVAL, "assert!(leaf_index < self.tree[0].len())");
    std::vector<HashOut> hashes;
    std::size_t idx = leaf_index / 2;
    for (std::size_t i = 1; i < count_levels; ++i, idx /= 2) hashes.push_back(tree[i][idx]);
    hashes.push_back(root);
    return hashes;
  }
};

inline bool verify_merkle_proof(GoldilocksField leaf, std::size_t leaf_index, const HashOut& root,
                                const std::vector<HashOut>& hashes) {  // :91-109
  const std::uint64_t idx = leaf_index;
  std::uint8_t ok = 0;
  check(p2mt_verify_merkle_proof_batch(&leaf, &idx, root.elements.data(),
                                       hashes.empty() ? nullptr : hashes[0].elements.data(), hashes.size(), 1, &ok));
  return ok != 0;
}

// ---------------------------------------------------------------- merkle_mountain_ranges.rs
inline std::pair<std::uint64_t, std::size_t> get_heights_bitmap_for_mmr_size(std::size_t mmr_size) {  // :39-81
  std::size_t rem = 0;
  const std::uint64_t bm = p2mt_get_heights_bitmap_for_mmr_size(mmr_size, &rem);
  return {bm, rem};
}

inline std::size_t get_mmr_index(std::size_t leaf_normal_index) {  // :257-270
  const std::int64_t r = p2mt_get_mmr_index(leaf_normal_index);
  if (r < 0) throw panic((int)r, "get_mmr_index: i32 overflow (n >= 2^30)");
  return (std::size_t)r;
}

struct MMR_proof {  // :15-23
  std::size_t mmr_size = 0;
  std::vector<std::pair<HashOut, bool>> merkle_proof;  // (sibling, sibling_on_left)
  std::vector<HashOut> peaks;

  bool verify(GoldilocksField leaf, const HashOut& root) const {  // :232-252 (panics at :245 if not among peaks)
    std::vector<HashOut> sib(merkle_proof.size());
    std::vector<std::uint8_t> lefts(merkle_proof.size());
    for (std::size_t i = 0; i < merkle_proof.size(); ++i) {
      sib[i] = merkle_proof[i].first;
      lefts[i] = merkle_proof[i].second ? 1 : 0;
    }
    int result = 0;
    check(p2mt_mmr_proof_verify(sib.empty() ? nullptr : sib[0].elements.data(), lefts.data(), (int)sib.size(),
                                peaks.empty() ? nullptr : peaks[0].elements.data(), (int)peaks.size(), leaf,
                                root.elements.data(), &result));
    return result != 0;
  }
};

class MMR {  // struct MMR { elements: Vec<HashOut> }, device-resident (:8-12)
 public:
  MMR() { check(p2mt_mmr_create(&h_)); }
  ~MMR() { p2mt_mmr_destroy(h_); }
  MMR(const MMR&) = delete;
  MMR& operator=(const MMR&) = delete;
  MMR(MMR&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }

  static MMR new_() { return MMR(); }  // MMR::new (:84-86)
  static MMR from_leaves(const std::vector<GoldilocksField>& leaves) {
    MMR m;
    m.extend(leaves);
    return m;
  }

  void add_leaf(GoldilocksField leaf) { check(p2mt_mmr_add_leaf(h_, leaf)); }  // :89-120 (write-combined)
  void extend(const std::vector<GoldilocksField>& leaves) { check(p2mt_mmr_extend(h_, leaves.data(), leaves.size())); }
  void reserve(std::size_t n_leaves) { check(p2mt_mmr_reserve(h_, n_leaves)); }

  std::size_t len() const { return p2mt_mmr_len(h_); }
  std::size_t num_leaves() const { return p2mt_mmr_num_leaves(h_); }
  std::vector<HashOut> elements() const {
    std::vector<HashOut> out(len());
    if (!out.empty()) check(p2mt_mmr_copy_elements(h_, 0, out.size(), out[0].elements.data()));
    return out;
  }

  HashOut bagging_the_peaks() const {  // :122-127
    HashOut r;
    check(p2mt_mmr_root(h_, r.elements.data()));
    return r;
  }
  std::vector<HashOut> get_peaks() const {  // :179-200
    std::vector<HashOut> p(P2MT_MAX_PROOF_LEN);
    int n = 0;
    check(p2mt_mmr_peaks(h_, p[0].elements.data(), &n));
    p.resize(n);
    return p;
  }
  MMR_proof get_proof(std::size_t mmr_index) const {  // :209-223
    std::vector<HashOut> sib(P2MT_MAX_PROOF_LEN), peaks(P2MT_MAX_PROOF_LEN);
    std::vector<std::uint8_t> lefts(P2MT_MAX_PROOF_LEN);
    int ns = 0, np = 0;
    MMR_proof pr;
    check(p2mt_mmr_proof(h_, mmr_index, sib[0].elements.data(), lefts.data(), &ns, peaks[0].elements.data(), &np,
                         &pr.mmr_size));
    for (int i = 0; i < ns; ++i) pr.merkle_proof.emplace_back(sib[i], lefts[i] != 0);
    pr.peaks.assign(peaks.begin(), peaks.begin() + np);
    return pr;
  }
  MMR_proof get_proof_normal_index(std::size_t normal_index) const { return get_proof(get_mmr_index(normal_index)); }  // :203-205
  std::vector<std::pair<HashOut, bool>> get_subtree_proof_elm(std::size_t mmr_index) const {  // :147-176
    return get_proof(mmr_index).merkle_proof;
  }
  p2mt_mmr* handle() const { return h_; }

 private:
  p2mt_mmr* h_ = nullptr;
};

// ---------------------------------------------------------------- plonky2 prover pieces (fri/oracle.rs, iop/challenger.rs)
using Extension = std::array<std::uint64_t, 2>;  // a + bX in F[X]/(X^2 - 7)

struct FriParams : p2mt_fri_params {
  // CircuitConfig::standard_recursion_config().fri_config.fri_params(degree_bits, false)
  static FriParams standard(unsigned degree_bits) {
    FriParams p;
    check(p2mt_fri_params_standard(degree_bits, &p));
    return p;
  }
};

// PolynomialBatch { polynomials, merkle_tree { leaves, digests, cap } } with blinding = false
struct PolynomialBatch {
  std::size_t n_polys = 0;
  unsigned degree_log = 0, rate_bits = 3, cap_height = 4;
  std::vector<GoldilocksField> polynomials;  // coefficients [n_polys][2^degree_log]
  std::vector<GoldilocksField> leaves;       // [2^(degree_log+rate_bits)][n_polys], bit-reversed point order
  std::vector<HashOut> digests, cap;         // digests level-major

  static PolynomialBatch commit(std::vector<GoldilocksField> polys, std::size_t n_polys, bool is_values, unsigned rate_bits,
                                unsigned cap_height) {
    PolynomialBatch b;
    if (n_polys == 0 || polys.size() % n_polys) throw panic(P2MT_EINVAL, "PolynomialBatch: ragged input");
    const std::size_t n = polys.size() / n_polys;
    if (n == 0 || (n & (n - 1))) throw panic(P2MT_EINVAL, "log2_strict: not a power of two");
    b.n_polys = n_polys;
    b.rate_bits = rate_bits;
    b.cap_height = cap_height;
    while ((std::size_t(1) << b.degree_log) < n) ++b.degree_log;
    const std::size_t big = n << rate_bits;
    std::size_t nd = 0;
    for (std::size_t r = big; r > (std::size_t(1) << cap_height); r >>= 1) nd += r;
    b.leaves.resize(big * n_polys);
    b.digests.resize(nd ? nd : 1);
    b.cap.resize(std::size_t(1) << cap_height);
    check(p2mt_polynomial_batch_commit(polys.data(), is_values ? 1 : 0, n_polys, b.degree_log, rate_bits, cap_height,
                                       b.leaves.data(), b.digests[0].elements.data(), b.cap[0].elements.data()));
    b.digests.resize(nd);
    if (is_values) check(p2mt_ntt_batch(polys.data(), b.degree_log, n_polys, 1));  // keep coefficients, as plonky2 does
    b.polynomials = std::move(polys);
    return b;
  }
  static PolynomialBatch from_values(std::vector<GoldilocksField> values, std::size_t n_polys, unsigned rate_bits = 3,
                                     unsigned cap_height = 4) {
    return commit(std::move(values), n_polys, true, rate_bits, cap_height);
  }
  static PolynomialBatch from_coeffs(std::vector<GoldilocksField> coeffs, std::size_t n_polys, unsigned rate_bits = 3,
                                     unsigned cap_height = 4) {
    return commit(std::move(coeffs), n_polys, false, rate_bits, cap_height);
  }
};

class Challenger {  // Challenger<F, PoseidonHash>, sponge resident on the device
 public:
  Challenger() { check(p2mt_challenger_create(&h_)); }
  ~Challenger() { p2mt_challenger_destroy(h_); }
  Challenger(const Challenger& o) { check(p2mt_challenger_clone(o.h_, &h_)); }
  Challenger& operator=(const Challenger&) = delete;
  void observe_element(GoldilocksField e) { check(p2mt_challenger_observe(h_, &e, 1)); }
  void observe_elements(const std::vector<GoldilocksField>& e) { check(p2mt_challenger_observe(h_, e.data(), e.size())); }
  void observe_hash(const HashOut& h) { check(p2mt_challenger_observe(h_, h.elements.data(), 4)); }
  void observe_cap(const std::vector<HashOut>& cap) {
    if (!cap.empty()) check(p2mt_challenger_observe(h_, cap[0].elements.data(), 4 * cap.size()));
  }
  void observe_extension_elements(const std::vector<Extension>& e) {
    if (!e.empty()) check(p2mt_challenger_observe(h_, e[0].data(), 2 * e.size()));
  }
  GoldilocksField get_challenge() { return get_n_challenges(1)[0]; }
  std::vector<GoldilocksField> get_n_challenges(std::size_t n) {
    std::vector<GoldilocksField> out(n);
    check(p2mt_challenger_get_challenges(h_, n, out.data()));
    return out;
  }
  Extension get_extension_challenge() {
    auto c = get_n_challenges(2);
    return Extension{c[0], c[1]};
  }
  p2mt_challenger* handle() const { return h_; }

 private:
  p2mt_challenger* h_ = nullptr;
};

// FriBatchInfo { point, polynomials: [(oracle index, polynomial index)] }
struct FriBatchInfo {
  Extension point{};
  std::vector<std::pair<std::uint32_t, std::uint32_t>> polynomials;
};

namespace detail {
inline void marshal(const std::vector<const PolynomialBatch*>& oracles, const std::vector<FriBatchInfo>& batches,
                    std::vector<p2mt_fri_oracle>& o, std::vector<p2mt_fri_batch>& b, std::vector<std::vector<std::uint32_t>>& flat) {
  for (auto* pb : oracles)
    o.push_back(p2mt_fri_oracle{pb->polynomials.data(), pb->leaves.data(),
                                pb->digests.empty() ? nullptr : pb->digests[0].elements.data(), pb->n_polys});
  flat.resize(batches.size());
  for (std::size_t i = 0; i < batches.size(); ++i) {
    for (auto& pr : batches[i].polynomials) {
      flat[i].push_back(pr.first);
      flat[i].push_back(pr.second);
    }
    b.push_back(p2mt_fri_batch{{batches[i].point[0], batches[i].point[1]}, flat[i].data(), batches[i].polynomials.size()});
  }
}
}  // namespace detail

// OpeningSet::to_fri_openings: values[batch][poly] at the batch's point
inline std::vector<std::vector<Extension>> fri_openings(const std::vector<FriBatchInfo>& batches,
                                                        const std::vector<const PolynomialBatch*>& oracles) {
  std::vector<p2mt_fri_oracle> o;
  std::vector<p2mt_fri_batch> b;
  std::vector<std::vector<std::uint32_t>> flat;
  detail::marshal(oracles, batches, o, b, flat);
  std::size_t total = 0;
  for (auto& bi : batches) total += bi.polynomials.size();
  std::vector<Extension> out(total ? total : 1);
  check(p2mt_fri_openings(o.data(), o.size(), b.data(), b.size(), oracles.at(0)->degree_log, out[0].data()));
  std::vector<std::vector<Extension>> res;
  std::size_t off = 0;
  for (auto& bi : batches) {
    res.emplace_back(out.begin() + off, out.begin() + off + bi.polynomials.size());
    off += bi.polynomials.size();
  }
  return res;
}

// PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params) -> FriProof words (layout: p2mt.h)
inline std::vector<std::uint64_t> prove_openings(const std::vector<FriBatchInfo>& batches,
                                                 const std::vector<const PolynomialBatch*>& oracles, Challenger& challenger,
                                                 const FriParams& params) {
  std::vector<p2mt_fri_oracle> o;
  std::vector<p2mt_fri_batch> b;
  std::vector<std::vector<std::uint32_t>> flat;
  detail::marshal(oracles, batches, o, b, flat);
  std::vector<std::uint64_t> np;
  for (auto* pb : oracles) np.push_back(pb->n_polys);
  const std::size_t total = p2mt_fri_proof_len(&params, np.size(), np.data());
  if (total == 0) throw panic(P2MT_EINVAL, "unsupported FriParams");
  std::vector<std::uint64_t> proof(total);
  check(p2mt_fri_prove_openings(o.data(), o.size(), b.data(), b.size(), &params, challenger.handle(), proof.data()));
  return proof;
}

// ---------------------------------------------------------------- plonky2 CircuitBuilder / CircuitData (plonk/circuit_builder.rs,
// circuit_data.rs, iop/witness.rs) as the reference uses them at mmr_plonky2_verifier.rs:13-151
using Target = p2mt_target;
struct BoolTarget {
  Target target;
};
struct HashOutTarget {
  std::array<Target, 4> elements{};
};

class PartialWitness {  // PartialWitness::new() (:123)
 public:
  PartialWitness() { check(p2mt_pw_create(&h_)); }
  ~PartialWitness() { p2mt_pw_destroy(h_); }
  PartialWitness(const PartialWitness&) = delete;
  PartialWitness& operator=(const PartialWitness&) = delete;
  void set_target(Target t, GoldilocksField v) { check(p2mt_pw_set_target(h_, t, v)); }
  void set_bool_target(BoolTarget t, bool v) { set_target(t.target, v ? 1 : 0); }
  void set_hash_target(const HashOutTarget& t, const HashOut& v) {
    for (int i = 0; i < 4; ++i) set_target(t.elements[i], v.elements[i]);
  }
  // pw.set_proof_with_pis_target / pw.set_verifier_data_target (mmr_plonky2_verifier_1_recursion.rs:201-202): defined below
  void set_proof_with_pis_target(const struct ProofWithPublicInputsTarget& target, const struct ProofWithPublicInputs& proof);
  void set_verifier_data_target(const struct VerifierCircuitTarget& target, const class CircuitData& inner);
  p2mt_partial_witness* handle() const { return h_; }

 private:
  p2mt_partial_witness* h_ = nullptr;
};

// ProofWithPublicInputsTarget<2>: one target per word of a proof of the inner circuit (the order p2mt_circuit_prove writes)
struct ProofWithPublicInputsTarget {
  std::vector<Target> targets;
  std::vector<Target> public_inputs;  // the last num_public_inputs targets
};
// VerifierCircuitTarget: constants_sigmas_cap [16][4] then circuit_digest [4]
struct VerifierCircuitTarget {
  std::vector<Target> targets;
};

// ProofWithPublicInputs as the words p2mt_circuit_prove returns (field order of plonky2's serialisation, see p2mt.h)
struct ProofWithPublicInputs {
  std::vector<std::uint64_t> words;
  std::vector<GoldilocksField> public_inputs;
};

class CircuitData {  // CircuitData<GoldilocksField, PoseidonGoldilocksConfig, 2>; prover data resident on the device
 public:
  explicit CircuitData(p2mt_circuit_data* h) : h_(h) {
    check(p2mt_circuit_get_info(h_, &info));
    prover_only.public_inputs.resize(info.num_public_inputs);
    check(p2mt_circuit_public_inputs(h_, prover_only.public_inputs.data()));
  }
  ~CircuitData() { p2mt_circuit_destroy(h_); }
  CircuitData(const CircuitData&) = delete;
  CircuitData& operator=(const CircuitData&) = delete;
  CircuitData(CircuitData&& o) noexcept : info(o.info), prover_only(std::move(o.prover_only)), h_(o.h_) { o.h_ = nullptr; }

  ProofWithPublicInputs prove(const PartialWitness& pw) {  // circuit_data.prove(pw) (:148)
    ProofWithPublicInputs p;
    p.words.resize(info.proof_len);
    check(p2mt_circuit_prove(h_, pw.handle(), p.words.data(), p.words.size()));
    p.public_inputs.assign(p.words.end() - info.num_public_inputs, p.words.end());
    return p;
  }
  // circuit_data.verify(proof) (:150): throws where plonky2 returns Err
  void verify(const ProofWithPublicInputs& proof) {
    int accepted = 0, reason = 0;
    check(p2mt_circuit_verify(h_, proof.words.data(), proof.words.size(), &accepted, &reason));
    if (!accepted) throw panic(P2MT_EINVAL, "proof rejected (reason " + std::to_string(reason) + ")");
  }
  // circuit_data.verify for many proofs in passes of up to 256 (p2mt_circuit_verify_batch): accepted[i] per proof
  std::vector<bool> verify_batch(const std::vector<ProofWithPublicInputs>& proofs) {
    const std::size_t n = proofs.size(), len = info.proof_len;
    std::vector<uint64_t> words(n * len);
    for (std::size_t i = 0; i < n; ++i) {
      if (proofs[i].words.size() != len) throw panic(P2MT_EINVAL, "proof does not belong to this circuit");
      std::copy(proofs[i].words.begin(), proofs[i].words.end(), words.begin() + i * len);
    }
    std::vector<int> acc(n), reason(n);
    check(p2mt_circuit_verify_batch(h_, words.data(), n, len, acc.data(), reason.data()));
    return std::vector<bool>(acc.begin(), acc.end());
  }
  // ProofWithPublicInputs::to_bytes / from_bytes in plonky2's Buffer order
  std::vector<uint8_t> to_bytes(const ProofWithPublicInputs& proof) const {
    std::vector<uint8_t> out(p2mt_proof_bytes_len(h_));
    check(p2mt_proof_to_bytes(h_, proof.words.data(), proof.words.size(), out.data(), out.size()));
    return out;
  }
  ProofWithPublicInputs from_bytes(const std::vector<uint8_t>& bytes) const {
    ProofWithPublicInputs p;
    p.words.resize(info.proof_len);
    check(p2mt_proof_from_bytes(h_, bytes.data(), bytes.size(), p.words.data(), p.words.size()));
    p.public_inputs.assign(p.words.end() - info.num_public_inputs, p.words.end());
    return p;
  }
  HashOut circuit_digest() const {
    HashOut d;
    check(p2mt_circuit_constants_sigmas(h_, nullptr, nullptr, d.elements.data()));
    return d;
  }
  p2mt_circuit_data* handle() const { return h_; }
  // `inner_circuit_data.common` / `.verifier_only`: the handle carries both
  const CircuitData& common() const { return *this; }
  const CircuitData& verifier_only() const { return *this; }
  p2mt_circuit_info info{};
  struct {
    std::vector<Target> public_inputs;
  } prover_only;

 private:
  p2mt_circuit_data* h_ = nullptr;
};

// Batched prover (p2mt_batch_prover_*): up to `batch` proofs of one circuit per pass of the pipeline; each proof equals
// circuit.prove(pw) word for word.  Borrows the circuit (keep it alive; one thread at a time).
class BatchProver {
 public:
  BatchProver(CircuitData& circuit, std::size_t batch) : circuit_(circuit) { check(p2mt_batch_prover_create(circuit.handle(), batch, &h_)); }
  ~BatchProver() { p2mt_batch_prover_destroy(h_); }
  BatchProver(const BatchProver&) = delete;
  BatchProver& operator=(const BatchProver&) = delete;
  std::vector<ProofWithPublicInputs> prove(const std::vector<const PartialWitness*>& witnesses) {
    const std::size_t n = witnesses.size(), len = circuit_.info.proof_len;
    std::vector<const p2mt_partial_witness*> hs(n);
    for (std::size_t i = 0; i < n; ++i) hs[i] = witnesses[i]->handle();
    std::vector<uint64_t> words(n * len);
    check(p2mt_batch_prover_prove(h_, hs.data(), n, words.data(), len, nullptr));
    std::vector<ProofWithPublicInputs> out(n);
    for (std::size_t i = 0; i < n; ++i) {
      out[i].words.assign(words.begin() + i * len, words.begin() + (i + 1) * len);
      out[i].public_inputs.assign(out[i].words.end() - circuit_.info.num_public_inputs, out[i].words.end());
    }
    return out;
  }

 private:
  CircuitData& circuit_;
  p2mt_batch_prover* h_ = nullptr;
};

inline void PartialWitness::set_proof_with_pis_target(const ProofWithPublicInputsTarget& target, const ProofWithPublicInputs& proof) {
  if (target.targets.size() != proof.words.size()) throw panic(P2MT_EINVAL, "proof does not match its target");
  check(p2mt_pw_set_proof_with_pis_target(h_, target.targets.data(), proof.words.data(), proof.words.size()));
}
inline void PartialWitness::set_verifier_data_target(const VerifierCircuitTarget& target, const CircuitData& inner) {
  check(p2mt_pw_set_verifier_data_target(h_, target.targets.data(), inner.handle()));
}

class CircuitBuilder {  // CircuitBuilder::<F, 2>::new(CircuitConfig::standard_recursion_config()) (:30-31)
 public:
  CircuitBuilder() { check(p2mt_cb_create(&h_)); }
  ~CircuitBuilder() { p2mt_cb_destroy(h_); }
  CircuitBuilder(const CircuitBuilder&) = delete;
  CircuitBuilder& operator=(const CircuitBuilder&) = delete;

  Target add_virtual_target() { return out1(p2mt_cb_add_virtual_target); }
  HashOutTarget add_virtual_hash() {
    HashOutTarget h;
    for (auto& e : h.elements) e = add_virtual_target();
    return h;
  }
  BoolTarget add_virtual_bool_target_safe() { return BoolTarget{out1(p2mt_cb_add_virtual_bool_target_safe)}; }
  Target constant(GoldilocksField c) {
    Target t;
    check(p2mt_cb_constant(h_, c, &t));
    return t;
  }
  Target zero() { return constant(0); }
  Target one() { return constant(1); }
  void connect(Target x, Target y) { check(p2mt_cb_connect(h_, x, y)); }
  Target mul(Target x, Target y) { return out3(p2mt_cb_mul, x, y); }
  Target add(Target x, Target y) { return out3(p2mt_cb_add, x, y); }
  Target sub(Target x, Target y) { return out3(p2mt_cb_sub, x, y); }
  Target mul_add(Target x, Target y, Target z) {
    Target t;
    check(p2mt_cb_mul_add(h_, x, y, z, &t));
    return t;
  }
  BoolTarget not_(BoolTarget b) {
    Target t;
    check(p2mt_cb_not(h_, b.target, &t));
    return BoolTarget{t};
  }
  BoolTarget or_(BoolTarget a, BoolTarget b) { return BoolTarget{out3(p2mt_cb_or, a.target, b.target)}; }
  BoolTarget is_equal(Target x, Target y) { return BoolTarget{out3(p2mt_cb_is_equal, x, y)}; }
  HashOutTarget hash_or_noop(const std::vector<Target>& inputs) {  // hash_or_noop::<PoseidonHash>
    HashOutTarget h;
    check(p2mt_cb_hash_or_noop(h_, inputs.data(), inputs.size(), h.elements.data()));
    return h;
  }
  HashOutTarget hash_n_to_hash_no_pad(const std::vector<Target>& inputs) {
    HashOutTarget h;
    check(p2mt_cb_hash_n_to_hash_no_pad(h_, inputs.data(), inputs.size(), h.elements.data()));
    return h;
  }
  // ---- recursion (mmr_plonky2_verifier_1_recursion.rs:95-104): plonky2's in-circuit verifier, built inside the library
  ProofWithPublicInputsTarget add_virtual_proof_with_pis(const CircuitData& inner_common) {
    ProofWithPublicInputsTarget t;
    t.targets.resize(inner_common.info.proof_len);
    check(p2mt_cb_add_virtual_proof_with_pis(h_, inner_common.handle(), t.targets.data(), t.targets.size()));
    t.public_inputs.assign(t.targets.end() - inner_common.info.num_public_inputs, t.targets.end());
    return t;
  }
  VerifierCircuitTarget add_virtual_verifier_data(unsigned cap_height) {
    VerifierCircuitTarget t;
    t.targets.resize(68);
    check(p2mt_cb_add_virtual_verifier_data(h_, cap_height, t.targets.data()));
    return t;
  }
  void verify_proof(const ProofWithPublicInputsTarget& proof_with_pis, const VerifierCircuitTarget& inner_verifier_data,
                    const CircuitData& inner_common) {  // builder.verify_proof::<PoseidonGoldilocksConfig>(...)
    check(p2mt_cb_verify_proof(h_, proof_with_pis.targets.data(), proof_with_pis.targets.size(), inner_verifier_data.targets.data(),
                               inner_common.handle()));
  }
  void register_public_inputs(const std::array<Target, 4>& t) { check(p2mt_cb_register_public_inputs(h_, t.data(), 4)); }
  void register_public_input(Target t) { check(p2mt_cb_register_public_inputs(h_, &t, 1)); }
  CircuitData build() {  // builder.build::<C>() (:89)
    p2mt_circuit_data* c = nullptr;
    check(p2mt_cb_build(h_, &c));
    return CircuitData(c);
  }

 private:
  Target out1(int (*f)(p2mt_circuit_builder*, p2mt_target*)) {
    Target t;
    check(f(h_, &t));
    return t;
  }
  Target out3(int (*f)(p2mt_circuit_builder*, p2mt_target, p2mt_target, p2mt_target*), Target x, Target y) {
    Target t;
    check(f(h_, x, y, &t));
    return t;
  }
  p2mt_circuit_builder* h_ = nullptr;
};

// ---------------------------------------------------------------- src/mmr/common.rs:5-58
inline BoolTarget equal(CircuitBuilder& builder, const HashOutTarget& first, const HashOutTarget& second) {
  const BoolTarget elm0 = builder.is_equal(first.elements[0], second.elements[0]);
  const BoolTarget elm1 = builder.is_equal(first.elements[1], second.elements[1]);
  const BoolTarget elm2 = builder.is_equal(first.elements[2], second.elements[2]);
  const BoolTarget elm3 = builder.is_equal(first.elements[3], second.elements[3]);
  const BoolTarget elm0_or_elm1 = builder.or_(elm0, elm1);
  const BoolTarget elm2_or_elm3 = builder.or_(elm2, elm3);
  return builder.or_(elm0_or_elm1, elm2_or_elm3);
}
inline BoolTarget or_list(CircuitBuilder& builder, const std::vector<BoolTarget>& ins) {
  if (ins.empty()) throw panic(P2MT_EINVAL, "assert!(ins.len() > 0)");
  if (ins.size() == 1) return ins[0];
  if (ins.size() == 2) return builder.or_(ins[0], ins[1]);
  std::vector<BoolTarget> pairs;
  for (std::size_t i = 0; i < ins.size(); i += 2) pairs.push_back(i + 1 < ins.size() ? builder.or_(ins[i], ins[i + 1]) : ins[i]);
  return or_list(builder, pairs);
}
inline HashOutTarget pick_hash(CircuitBuilder& builder, const HashOutTarget& option1, const HashOutTarget& option2,
                               BoolTarget pick_left) {
  const BoolTarget opposite = builder.not_(pick_left);
  // call order as in common.rs:48-55 -- four `mul`s, then four `mul_add`s: the builder packs arithmetic operations into gate slots
  // in call order, so a different interleaving gives different gate rows as soon as a gate is partly filled
  Target t[4];
  for (int i = 0; i < 4; ++i) t[i] = builder.mul(option2.elements[i], opposite.target);
  HashOutTarget out;
  for (int i = 0; i < 4; ++i) out.elements[i] = builder.mul_add(option1.elements[i], pick_left.target, t[i]);
  return out;
}

// ---------------------------------------------------------------- src/mmr/mmr_plonky2_verifier.rs:13-91
struct MmrVerifierCircuit {
  CircuitData data;
  Target leaf_to_prove;
  std::vector<std::pair<HashOutTarget, BoolTarget>> proof_targets;
  std::vector<HashOutTarget> peak_targets;
};
inline MmrVerifierCircuit verify_mmr_proof_circuit(std::size_t nr_merkle_proof_elms, std::size_t nr_peaks) {
  std::vector<std::pair<HashOutTarget, BoolTarget>> proof_targets;
  std::vector<HashOutTarget> peak_targets;
  CircuitBuilder builder;
  const Target leaf_to_prove = builder.add_virtual_target();
  HashOutTarget next_hash = builder.hash_or_noop({leaf_to_prove});
  auto cat = [](const HashOutTarget& a, const HashOutTarget& b) {
    std::vector<Target> v(a.elements.begin(), a.elements.end());
    v.insert(v.end(), b.elements.begin(), b.elements.end());
    return v;
  };
  for (std::size_t k = 0; k < nr_merkle_proof_elms; ++k) {
    const HashOutTarget merkle_proof_elm = builder.add_virtual_hash();
    const BoolTarget elm_on_left = builder.add_virtual_bool_target_safe();
    proof_targets.emplace_back(merkle_proof_elm, elm_on_left);
    const HashOutTarget option1 = builder.hash_or_noop(cat(merkle_proof_elm, next_hash));  // sibling on the left
    const HashOutTarget option2 = builder.hash_or_noop(cat(next_hash, merkle_proof_elm));  // sibling on the right
    next_hash = pick_hash(builder, option1, option2, elm_on_left);
  }
  std::vector<HashOutTarget> peaks;
  std::vector<BoolTarget> equals;
  for (std::size_t k = 0; k < nr_peaks; ++k) {
    const HashOutTarget peak = builder.add_virtual_hash();
    peaks.push_back(peak);
    peak_targets.push_back(peak);
    equals.push_back(equal(builder, peak, next_hash));
  }
  const BoolTarget hash_in_peaks = or_list(builder, equals);
  builder.connect(builder.one(), hash_in_peaks.target);
  if (peaks.size() > 1) {
    std::vector<Target> all;
    for (auto& p : peaks) all.insert(all.end(), p.elements.begin(), p.elements.end());
    builder.register_public_inputs(builder.hash_n_to_hash_no_pad(all).elements);
  } else {
    builder.register_public_inputs(peaks.at(0).elements);
  }
  return MmrVerifierCircuit{builder.build(), leaf_to_prove, std::move(proof_targets), std::move(peak_targets)};
}


// ---------------------------------------------------------------- src/mmr/mmr_plonky2_verifier_1_recursion.rs:20-75 (inner circuit)
struct InnerMerkleProofCircuit {
  CircuitData data;
  Target leaf_to_prove;
  std::vector<std::pair<HashOutTarget, BoolTarget>> proof_targets;
};
inline InnerMerkleProofCircuit verify_inner_merkle_proof_circuit(std::size_t nr_merkle_proof_elms, std::size_t nr_peaks) {
  std::vector<std::pair<HashOutTarget, BoolTarget>> proof_targets;
  CircuitBuilder builder;
  const Target leaf_to_prove = builder.add_virtual_target();
  HashOutTarget next_hash = builder.hash_or_noop({leaf_to_prove});
  auto cat = [](const HashOutTarget& a, const HashOutTarget& b) {
    std::vector<Target> v(a.elements.begin(), a.elements.end());
    v.insert(v.end(), b.elements.begin(), b.elements.end());
    return v;
  };
  for (std::size_t k = 0; k < nr_merkle_proof_elms; ++k) {
    const HashOutTarget merkle_proof_elm = builder.add_virtual_hash();
    const BoolTarget elm_on_left = builder.add_virtual_bool_target_safe();
    proof_targets.emplace_back(merkle_proof_elm, elm_on_left);
    const HashOutTarget option1 = builder.hash_or_noop(cat(merkle_proof_elm, next_hash));
    const HashOutTarget option2 = builder.hash_or_noop(cat(next_hash, merkle_proof_elm));
    next_hash = pick_hash(builder, option1, option2, elm_on_left);
  }
  std::vector<BoolTarget> equals;
  for (std::size_t k = 0; k < nr_peaks; ++k) {
    const HashOutTarget peak = builder.add_virtual_hash();
    for (Target elm : peak.elements) builder.register_public_input(elm);
    equals.push_back(equal(builder, peak, next_hash));
  }
  const BoolTarget hash_in_peaks = or_list(builder, equals);
  builder.connect(builder.one(), hash_in_peaks.target);
  return InnerMerkleProofCircuit{builder.build(), leaf_to_prove, std::move(proof_targets)};
}

// ---------------------------------------------------------------- :84-140 (outer circuit)
struct CompleteVerificationCircuit {
  CircuitData data;
  ProofWithPublicInputsTarget prev_proof_target;
  VerifierCircuitTarget prev_proof_verifier_data;
  std::vector<HashOutTarget> targets;
};
inline CompleteVerificationCircuit complete_verification_circuit_with_inner_proof(const CircuitData& inner_proof_circuit_data_common,
                                                                                  std::size_t nr_peaks) {
  CircuitBuilder builder;
  ProofWithPublicInputsTarget prev_proof_target = builder.add_virtual_proof_with_pis(inner_proof_circuit_data_common);
  VerifierCircuitTarget prev_proof_verifier_data = builder.add_virtual_verifier_data(4);  // common.config.fri_config.cap_height
  builder.verify_proof(prev_proof_target, prev_proof_verifier_data, inner_proof_circuit_data_common);
  std::vector<HashOutTarget> targets, peaks;
  std::vector<BoolTarget> equals;
  HashOutTarget prev_hash;  // HashOutTarget::from_vec(prev_proof_target.public_inputs[0..4]): the FIRST peak (quirk Q4)
  for (int k = 0; k < 4; ++k) prev_hash.elements[k] = prev_proof_target.public_inputs.at(k);
  for (std::size_t k = 0; k < nr_peaks; ++k) {
    const HashOutTarget peak = builder.add_virtual_hash();
    peaks.push_back(peak);
    targets.push_back(peak);
    equals.push_back(equal(builder, peak, prev_hash));
  }
  const BoolTarget hash_in_peaks = or_list(builder, equals);
  builder.connect(builder.one(), hash_in_peaks.target);
  if (peaks.size() > 1) {
    std::vector<Target> all;
    for (auto& p : peaks) all.insert(all.end(), p.elements.begin(), p.elements.end());
    builder.register_public_inputs(builder.hash_n_to_hash_no_pad(all).elements);
  } else {
    builder.register_public_inputs(peaks.at(0).elements);
  }
  return CompleteVerificationCircuit{builder.build(), std::move(prev_proof_target), std::move(prev_proof_verifier_data), std::move(targets)};
}

}  // namespace p2mt
