// The reference's own unit tests, re-expressed against the C++ mirror (include/p2mt.hpp).  Runs on a GPU box.
//   simple_merkle_tree.rs:117-309  (5 tests), merkle_mountain_ranges.rs:278-374 (4 tests)
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../include/p2mt.hpp"

using namespace p2mt;
#define REQUIRE(x) do { if (!(x)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); std::exit(1); } } while (0)

static HashOut H(std::uint64_t a, std::uint64_t b, std::uint64_t c, std::uint64_t d) { return HashOut{{a, b, c, d}}; }

static const std::vector<GoldilocksField> k16 = {
    14786323743454721611ull, 976503040092093812ull,  4644130751253292674ull,  6522877527545910706ull,
    11021172818651636092ull, 12048403458499719587ull, 11457874926809001558ull, 14982007443548219923ull,
    4546369223935415035ull,  7205140577604465038ull, 4644130751253292674ull,  4208177174652750506ull,
    16147116534354400672ull, 18147003476480002882ull, 14133393155459789216ull, 9890944065319669426ull};

static void test_build_merkle_tree_4_leaves() {  // :119-144, values printed at :136-140
  MerkleTree t = MerkleTree::build({2890852870ull, 156728478ull, 2876514289ull, 984286162ull});
  REQUIRE(t.count_levels == 2);
  REQUIRE(t.tree[0][0] == H(2890852870ull, 0, 0, 0) && t.tree[0][3] == H(984286162ull, 0, 0, 0));
  REQUIRE(t.tree[1][0] == H(6678006133445961348ull, 15827935749738443865ull, 6295652393730592048ull, 1546515167911236130ull));
  REQUIRE(t.tree[1][1] == H(6698018865469624861ull, 12486244005715193285ull, 11330639022572315007ull, 6059804404595156248ull));
  REQUIRE(t.root == H(13451271846715771774ull, 4069913004933160254ull, 14528216580130305557ull, 9716424959297545638ull));
}

static void test_build_merkle_tree_16_leaves() {  // :146-194
  MerkleTree t = MerkleTree::build(k16);
  REQUIRE(t.count_levels == 4);
  REQUIRE(t.tree[3][1] == H(14079844864384152521ull, 6499705357519308869ull, 16026207645313349904ull, 15079809878245341298ull));
  REQUIRE(t.root == H(2659148958598424285ull, 16496267010313658247ull, 12216516055477211974ull, 15749220035779350537ull));
}

static void test_merkle_proof_small_tree() {  // :196-212 (the asserted values)
  MerkleTree t = MerkleTree::build({2890852870ull, 156728478ull, 2876514289ull, 984286162ull});
  auto p = t.get_merkle_proof(0);
  REQUIRE(p[0] == H(156728478ull, 0, 0, 0));
  REQUIRE(p[1] == H(6698018865469624861ull, 12486244005715193285ull, 11330639022572315007ull, 6059804404595156248ull));
}

static void test_verify_merkle_proof_16() {  // :236-308 incl. the negative cases
  MerkleTree t = MerkleTree::build(k16);
  for (std::size_t i = 0; i < 16; ++i) REQUIRE(verify_merkle_proof(k16[i], i, t.root, t.get_merkle_proof(i)));
  auto p0 = t.get_merkle_proof(0), p1 = t.get_merkle_proof(1);
  REQUIRE(!verify_merkle_proof(k16[1], 0, t.root, p0));        // wrong leaf
  REQUIRE(!verify_merkle_proof(k16[0], 1, t.root, p0));        // wrong index
  REQUIRE(!verify_merkle_proof(k16[0], 0, t.root, p1));        // wrong proof
  REQUIRE(!verify_merkle_proof(k16[0], 0, t.tree[0][0], p0));  // wrong root
  bool threw = false;
  try { MerkleTree::build({1, 2, 3}); } catch (const panic&) { threw = true; }  // log2_strict panic :30
  REQUIRE(threw);
}

static void test_heights_bitmap() {  // merkle_mountain_ranges.rs:279-303
  const std::pair<std::size_t, std::uint64_t> tv[] = {{1, 1}, {3, 2}, {4, 3}, {7, 4}, {10, 6}, {15, 8}, {22, 12}, {25, 14},
                                                      {26, 15}, {31, 16}, {32, 17}, {34, 18}, {35, 19}, {38, 20}, {41, 22}, {42, 23}};
  for (auto& v : tv) {
    REQUIRE(get_heights_bitmap_for_mmr_size(v.first).first == v.second);
    REQUIRE(get_heights_bitmap_for_mmr_size(v.first).second == 0);
  }
}

static void test_get_mmr_index() {  // :305-329
  const std::size_t tv[][2] = {{0, 0}, {1, 1}, {2, 3}, {3, 4}, {4, 7}, {5, 8}, {6, 10}, {7, 11}, {8, 15}, {9, 16}, {10, 18},
                               {11, 19}, {12, 22}, {13, 23}, {14, 25}, {15, 26}};
  for (auto& v : tv) REQUIRE(get_mmr_index(v[0]) == v[1]);
}

static std::vector<GoldilocksField> random_leaves(std::size_t n, unsigned seed) {
  std::mt19937_64 rng(seed);
  std::uniform_int_distribution<std::uint64_t> d(0, GOLDILOCKS_FIELD_ORDER - 1);  // gen_range(0..GOLDILOCKS_FIELD_ORDER)
  std::vector<GoldilocksField> v(n);
  for (auto& x : v) x = d(rng);
  return v;
}

static void test_mmr_add_leaf() {  // :331-341: 100 leaves one by one
  MMR mmr = MMR::new_();
  auto leaves = random_leaves(100, 1);
  for (auto l : leaves) mmr.add_leaf(l);
  REQUIRE(mmr.len() == 197);  // 2*100 - popcount(100)
  MMR bulk = MMR::from_leaves(leaves);
  REQUIRE(mmr.elements() == bulk.elements());
}

static void test_get_proof() {  // :343-374 (the reference prints; here the result is asserted)
  auto leaves = random_leaves(16, 2);
  MMR mmr = MMR::new_();
  for (auto l : leaves) mmr.add_leaf(l);
  const std::size_t standard_index = 4, leaf_index = 7;
  MMR_proof proof = mmr.get_proof(leaf_index);
  HashOut root = mmr.bagging_the_peaks();
  REQUIRE(proof.mmr_size == 31 && proof.peaks.size() == 1 && proof.merkle_proof.size() == 4);
  REQUIRE(proof.verify(leaves[standard_index], root));
  REQUIRE(mmr.get_proof_normal_index(standard_index).merkle_proof == proof.merkle_proof);
  bool threw = false;
  try { proof.verify(leaves[standard_index + 1], root); } catch (const panic& e) { threw = e.code == P2MT_ENOTPEAK; }  // :245
  REQUIRE(threw);
  // every leaf of MMRs with 3, 7, 31, 70 leaves (mmr_plonky2_verifier.rs:153-193, native part :113-117)
  for (std::size_t n : {3u, 7u, 31u, 70u}) {
    auto lv = random_leaves(n, 10 + (unsigned)n);
    MMR m = MMR::from_leaves(lv);
    HashOut r = m.bagging_the_peaks();
    for (std::size_t i = 0; i < n; ++i) REQUIRE(m.get_proof_normal_index(i).verify(lv[i], r));
  }
}

// Prover pieces (plonky2; no test in the reference beyond prove+verify): known answers derived from the CPU restatement
// (tools/gen_fri_golden.py), whose permutation is pinned by the reference's own vectors.
static void test_challenger_and_fri() {
  Challenger ch;
  ch.observe_elements({1, 2, 3});
  auto c = ch.get_n_challenges(3);
  REQUIRE(c[0] == 12398646804117377360ull && c[1] == 15781308336284228359ull && c[2] == 17027997015668057891ull);
  ch.observe_elements({0, 1, 2, 3, 4, 5, 6, 7, 8, 9});
  REQUIRE(ch.get_challenge() == 5672640524457059960ull);

  // two polynomials of degree < 32: f0 = sum (i+1) x^i, f1 = sum (2i+1) x^i, opened at 3 + 5X
  std::vector<GoldilocksField> coeffs(64);
  for (std::uint64_t i = 0; i < 32; ++i) { coeffs[i] = i + 1; coeffs[32 + i] = 2 * i + 1; }
  PolynomialBatch pb = PolynomialBatch::from_coeffs(coeffs, 2);
  REQUIRE(pb.leaves.size() == 256 * 2 && pb.cap.size() == 16 && pb.digests.size() == 256 + 128 + 64 + 32);
  // from_values(fft(coeffs)) is the same commitment
  std::vector<GoldilocksField> values = coeffs;
  check(p2mt_ntt_batch(values.data(), 5, 2, 0));
  PolynomialBatch pv = PolynomialBatch::from_values(values, 2);
  REQUIRE(pv.cap == pb.cap && pv.polynomials == pb.polynomials);

  std::vector<FriBatchInfo> batches(1);
  batches[0].point = Extension{3, 5};
  batches[0].polynomials = {{0, 0}, {0, 1}};
  FriParams params = FriParams::standard(5);
  REQUIRE(params.num_reductions == 0 && params.rate_bits == 3 && params.cap_height == 4 && params.num_query_rounds == 28);
  params.proof_of_work_bits = 4;
  params.num_query_rounds = 2;
  Challenger tr;
  tr.observe_cap(pb.cap);
  Challenger tr2(tr);
  auto proof = prove_openings(batches, {&pb}, tr, params);
  REQUIRE(proof.size() == 101 && proof.back() == 6);
  REQUIRE(proof[96] == 1587238328117080795ull && proof[99] == 6829301891844123817ull);
  std::uint64_t x = 0;
  for (auto w : proof) x ^= w;
  REQUIRE(x == 6717879834928653870ull);
  REQUIRE(prove_openings(batches, {&pb}, tr2, params) == proof);  // deterministic (smallest PoW witness)
  auto op = fri_openings(batches, {&pb});
  REQUIRE(op.size() == 1 && op[0].size() == 2);
  // 1 + 2z + ... at z = 3 + 5X must also be what the restatement gives for the first 8 coefficients' polynomial
  std::vector<GoldilocksField> small = {1, 2, 3, 4, 5, 6, 7, 8};
  Extension pt{3, 5}, out{};
  check(p2mt_eval_polys_ext(small.data(), 1, 3, pt.data(), out.data()));
  REQUIRE(out[0] == 1210260379ull && out[1] == 490064140ull);
}

// mmr_plonky2_verifier.rs:102-151 (test_mmr_verifier) with fixed leaves 1, 2, 3 and leaf index 1: MMR -> proof -> circuit
// -> witness -> prove.  The reference ends in plonky2's `circuit_data.verify(proof)`; here the proof words are compared
// with the committed vector tests/golden/prove_vectors.json ("mmr_leaves_1_2_3_index_1", generated from the CPU
// restatement, whose verifier accepts it).
static void test_mmr_verifier_3leaves() {
  const std::vector<GoldilocksField> leaves = {1, 2, 3};
  MMR mmr = MMR::new_();
  for (auto l : leaves) mmr.add_leaf(l);
  const MMR_proof pr = mmr.get_proof(get_mmr_index(1));
  const HashOut root = mmr.bagging_the_peaks();
  REQUIRE(pr.verify(leaves[1], root));
  MmrVerifierCircuit c = verify_mmr_proof_circuit(pr.merkle_proof.size(), pr.peaks.size());
  PartialWitness pw;
  pw.set_target(c.leaf_to_prove, leaves[1]);
  for (std::size_t i = 0; i < pr.merkle_proof.size(); ++i) {
    pw.set_hash_target(c.proof_targets[i].first, pr.merkle_proof[i].first);
    pw.set_bool_target(c.proof_targets[i].second, pr.merkle_proof[i].second);
  }
  for (std::size_t i = 0; i < pr.peaks.size(); ++i) pw.set_hash_target(c.peak_targets[i], pr.peaks[i]);
  for (int i = 0; i < 4; ++i) pw.set_target(c.data.prover_only.public_inputs[i], root.elements[i]);
  const ProofWithPublicInputs proof = c.data.prove(pw);
  REQUIRE(c.data.info.degree_bits == 4 && proof.words.size() == 9227);
  REQUIRE(c.data.circuit_digest() == H(10066954030287170541ull, 2095877397257724682ull, 6270894316388847652ull, 800445324960154887ull));
  REQUIRE(proof.public_inputs.size() == 4);
  for (int i = 0; i < 4; ++i) REQUIRE(proof.public_inputs[i] == root.elements[i]);
  REQUIRE(root == H(14051017894672733496ull, 17897758925374905203ull, 11557515652286392125ull, 12346532418229956107ull));
  std::uint64_t acc = 0;
  for (std::size_t i = 0; i < proof.words.size(); ++i) acc += (std::uint64_t)(i + 1) * proof.words[i];
  REQUIRE(acc == 713678955804639150ull);  // proof_weighted_checksum of the golden vector
  c.data.verify(proof);                      // circuit_data.verify(proof) (:150)
  ProofWithPublicInputs forged = proof;
  forged.words[200] ^= 1;
  bool rejected = false;
  try {
    c.data.verify(forged);
  } catch (const panic&) {
    rejected = true;
  }
  REQUIRE(rejected);
  // a wrong side bit contradicts the circuit: plonky2 panics in witness generation
  PartialWitness bad;
  bad.set_target(c.leaf_to_prove, leaves[1]);
  for (std::size_t i = 0; i < pr.merkle_proof.size(); ++i) {
    bad.set_hash_target(c.proof_targets[i].first, pr.merkle_proof[i].first);
    bad.set_bool_target(c.proof_targets[i].second, !pr.merkle_proof[i].second);
  }
  for (std::size_t i = 0; i < pr.peaks.size(); ++i) bad.set_hash_target(c.peak_targets[i], pr.peaks[i]);
  bool panicked = false;
  try {
    c.data.prove(bad);
  } catch (const panic& e) {
    panicked = e.code == P2MT_EINVAL;
  }
  REQUIRE(panicked);
}

// mmr_plonky2_verifier_1_recursion.rs:152-192 (the inner proof of test_mmr_verifier_1_recursion): peaks are the public inputs
static void test_inner_merkle_proof_3leaves() {
  MMR mmr = MMR::from_leaves({1, 2, 3});
  const MMR_proof pr = mmr.get_proof(get_mmr_index(1));
  InnerMerkleProofCircuit c = verify_inner_merkle_proof_circuit(pr.merkle_proof.size(), pr.peaks.size());
  REQUIRE(c.data.prover_only.public_inputs.size() == 4 * pr.peaks.size());
  PartialWitness pw;
  pw.set_target(c.leaf_to_prove, 2);
  for (std::size_t i = 0; i < pr.merkle_proof.size(); ++i) {
    pw.set_hash_target(c.proof_targets[i].first, pr.merkle_proof[i].first);
    pw.set_bool_target(c.proof_targets[i].second, pr.merkle_proof[i].second);
  }
  for (std::size_t i = 0; i < pr.peaks.size(); ++i)
    for (int k = 0; k < 4; ++k) pw.set_target(c.data.prover_only.public_inputs[4 * i + k], pr.peaks[i].elements[k]);
  const ProofWithPublicInputs proof = c.data.prove(pw);
  c.data.verify(proof);
  for (std::size_t i = 0; i < pr.peaks.size(); ++i)
    for (int k = 0; k < 4; ++k) REQUIRE(proof.public_inputs[4 * i + k] == pr.peaks[i].elements[k]);
  // the batched prover: three copies of the statement in passes of two -> the same words as prove
  BatchProver batch(c.data, 2);
  const std::vector<ProofWithPublicInputs> many = batch.prove({&pw, &pw, &pw});
  REQUIRE(many.size() == 3);
  for (const auto& p : many) REQUIRE(p.words == proof.words);
  REQUIRE(c.data.prove(pw).words == proof.words);  // the circuit handle is its own again
  // batched verify and the byte form of the proof
  ProofWithPublicInputs broken = proof;
  broken.words[200] ^= 1;
  const std::vector<bool> verdicts = c.data.verify_batch({proof, broken, proof});
  REQUIRE(verdicts.size() == 3 && verdicts[0] && !verdicts[1] && verdicts[2]);
  const std::vector<uint8_t> bytes = c.data.to_bytes(proof);
  REQUIRE(bytes.size() > 8 * proof.words.size());
  REQUIRE(c.data.from_bytes(bytes).words == proof.words);
}

// mmr_plonky2_verifier_1_recursion.rs:152-221 test_complete_verification_circuit_with_inner_proof, leaves (1..7), leaf index 5
// (:229-236: 3 peaks; the leaf sits in the second mountain, quirk Q4 lets the outer check pass)
static void test_mmr_verifier_1_recursion_7leaves() {
  MMR mmr = MMR::from_leaves({1, 2, 3, 4, 5, 6, 7});
  const MMR_proof pr = mmr.get_proof(get_mmr_index(5));
  InnerMerkleProofCircuit inner = verify_inner_merkle_proof_circuit(pr.merkle_proof.size(), pr.peaks.size());
  PartialWitness pw1;
  pw1.set_target(inner.leaf_to_prove, 6);
  for (std::size_t i = 0; i < pr.merkle_proof.size(); ++i) {
    pw1.set_hash_target(inner.proof_targets[i].first, pr.merkle_proof[i].first);
    pw1.set_bool_target(inner.proof_targets[i].second, pr.merkle_proof[i].second);
  }
  for (std::size_t i = 0; i < pr.peaks.size(); ++i)
    for (int k = 0; k < 4; ++k) pw1.set_target(inner.data.prover_only.public_inputs[4 * i + k], pr.peaks[i].elements[k]);
  const ProofWithPublicInputs inner_proof = inner.data.prove(pw1);
  CompleteVerificationCircuit outer = complete_verification_circuit_with_inner_proof(inner.data.common(), pr.peaks.size());
  REQUIRE(outer.data.info.degree_bits == 12);
  PartialWitness pw2;
  pw2.set_proof_with_pis_target(outer.prev_proof_target, inner_proof);
  pw2.set_verifier_data_target(outer.prev_proof_verifier_data, inner.data.verifier_only());
  for (std::size_t i = 0; i < pr.peaks.size(); ++i) pw2.set_hash_target(outer.targets[i], pr.peaks[i]);
  const HashOut root = mmr.bagging_the_peaks();
  for (int k = 0; k < 4; ++k) pw2.set_target(outer.data.prover_only.public_inputs[k], root.elements[k]);
  const ProofWithPublicInputs final_proof = outer.data.prove(pw2);
  outer.data.verify(final_proof);
  for (int k = 0; k < 4; ++k) REQUIRE(final_proof.public_inputs[k] == root.elements[k]);
}

// A caller that does arithmetic BEFORE pick_hash (common.rs:42-58): three `mul`s leave an ArithmeticGate partly filled, so the
// order in which pick_hash issues its eight operations decides which gate slots they land in.  The digest is printed;
// tests/test_cpp_mirror.py compares it with the Python mirror's and the oracle's for the same circuit.
static void test_pick_hash_call_order() {
  CircuitBuilder builder;
  const Target x = builder.add_virtual_target(), y = builder.add_virtual_target();
  const Target m1 = builder.mul(x, y), m2 = builder.mul(m1, y), m3 = builder.mul(m2, x);
  const HashOutTarget h1 = builder.add_virtual_hash(), h2 = builder.add_virtual_hash();
  const BoolTarget pick_left = builder.add_virtual_bool_target_safe();
  const HashOutTarget out = pick_hash(builder, h1, h2, pick_left);
  builder.register_public_inputs(out.elements);
  builder.register_public_input(m3);
  CircuitData data = builder.build();
  const HashOut d = data.circuit_digest();
  std::printf("pick_hash_call_order digest: %llu %llu %llu %llu\n", (unsigned long long)d.elements[0],
              (unsigned long long)d.elements[1], (unsigned long long)d.elements[2], (unsigned long long)d.elements[3]);
}

int main() {
  if (p2mt_device_count() == 0) { std::fprintf(stderr, "no GPU: the product has no CPU fallback\n"); return 77; }
  check(p2mt_init(0));
  test_build_merkle_tree_4_leaves();
  test_build_merkle_tree_16_leaves();
  test_merkle_proof_small_tree();
  test_verify_merkle_proof_16();
  test_heights_bitmap();
  test_get_mmr_index();
  test_mmr_add_leaf();
  test_get_proof();
  test_challenger_and_fri();
  test_mmr_verifier_3leaves();
  test_inner_merkle_proof_3leaves();
  test_mmr_verifier_1_recursion_7leaves();
  test_pick_hash_call_order();
  std::puts("cpp mirror: 8 reference tests + prover pieces + the recursion passed");
  return 0;
}
