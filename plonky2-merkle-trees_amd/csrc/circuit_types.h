// circuit_types.h -- host-side types shared by the circuit translation units (p2mt_circuit.hip: CircuitBuilder primitives, build(),
// prove, verify; p2mt_recursion.hip: extension-field gadgets and plonky2's in-circuit verifier).
//
// Everything here restates plonky2 (git rev 3b21b87d, NOT in /root/reference -- parity unpinned) under
// CircuitConfig::standard_recursion_config() (/root/reference/src/mmr/mmr_plonky2_verifier.rs:30, mmr_plonky2_verifier_1_recursion.rs:28,92).
#pragma once
#include "runtime.h"

#include <array>
#include <map>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace p2mt_cb {

typedef uint64_t u64;
typedef uint32_t u32;
constexpr u64 kP = 0xFFFFFFFF00000001ull;

// CircuitConfig::standard_recursion_config()
constexpr u32 kNumWires = 135, kNumRouted = 80, kNumConsts = 2, kNumCh = 2, kQF = 8, kRateBits = 3, kCapHeight = 4;
constexpr u32 kNumChunks = (kNumRouted + kQF - 1) / kQF, kNumProds = kNumChunks - 1, kNumOps = kNumRouted / 4;
constexpr u32 kNumZs = kNumCh * (1 + kNumProds), kNumQuot = kNumCh * kQF, kNumGateConstraints = 123;

// Conventions of plonky2 @3b21b87d restated from recall that have a plausible alternative: ONE switch each (the twin in the
// CPU restatement is CONVENTIONS in its circuit module; tools/plonky2_crosscheck carries the vectors of every alternative).
//   circuit_digest = hash_no_pad(constants_sigmas_cap || D || degree_bits), D = the digest of the (empty) domain separator:
enum { kDomainSepHashPad = 0,   // D = hash_pad([]) = hash_no_pad([1,0,0,0,0,0,0,1])  (circuit_builder.rs build(), pad10*1)  [default]
       kDomainSepZeroHash = 1,  // D = hash_no_pad([]) = [0,0,0,0]                      (rounds 1-2 of this repository)
       kDomainSepNone = 2 };    // no D term                                            (a revision without domain separators)
constexpr int kDigestDomainSeparator = kDomainSepHashPad;
//   proof of work: the response of a candidate w is `observe(w); get_challenge()` on a copy of the transcript (fri/prover.rs
//   fri_proof_of_work, one duplexing).  Implemented by k_fri_pow*, the transcript tail of p2mt_fri.hip, verify_pass (p2mt_circuit.hip) and
//   the in-circuit check of p2mt_recursion.hip; if plonky2's verify reports "Invalid proof-of-work witness" on the cross-check,
//   those four sites are what to change.
constexpr int kPowRule = 0;

// Gate types (the numbering is the oracle's ORACLE_GATE_*).  The last eight are what builder.verify_proof adds; their parameters
// are what each gate's new_from_config yields for 80 routed / 135 wires / 2 constants, D = 2.
enum {
  G_NOOP = 0,
  G_CONSTANT,
  G_PUBLIC_INPUT,
  G_ARITHMETIC,
  G_POSEIDON,
  G_BASE_SUM,             // BaseSumGate<2> { num_limbs: 63 }
  G_ARITHMETIC_EXT,       // ArithmeticExtensionGate { num_ops: 10 }
  G_MUL_EXT,              // MulExtensionGate { num_ops: 13 }
  G_REDUCING,             // ReducingGate { num_coeffs: 43 }
  G_REDUCING_EXT,         // ReducingExtensionGate { num_coeffs: 32 }
  G_RANDOM_ACCESS,        // RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }
  G_COSET_INTERPOLATION,  // CosetInterpolationGate { subgroup_bits: 4, degree: 6 }
  G_POSEIDON_MDS,         // PoseidonMdsGate
  G_KINDS
};
constexpr u32 kGateDegree[G_KINDS] = {0, 1, 1, 3, 7, 2, 3, 3, 2, 2, 5, 6, 1};
constexpr u32 kGateNumConsts[G_KINDS] = {0, 2, 0, 2, 0, 0, 2, 1, 0, 0, 2, 0, 0};
constexpr u32 kGateNumConstraints[G_KINDS] = {0, 2, 4, 20, 123, 64, 20, 26, 86, 64, 26, 12, 24};
// plonky2 sorts the gate types of a circuit by (degree, id): Noop(0) < Constant(1) < PoseidonMds(1) < PublicInput(1) <
// BaseSum(2) < ReducingExtension(2) < Reducing(2) < ArithmeticExtension(3) < Arithmetic(3) < MulExtension(3) <
// RandomAccess(5) < CosetInterpolation(6) < Poseidon(7)
constexpr int kSortedKinds[G_KINDS] = {G_NOOP,     G_CONSTANT,       G_POSEIDON_MDS, G_PUBLIC_INPUT, G_BASE_SUM,      G_REDUCING_EXT,        G_REDUCING,
                                       G_ARITHMETIC_EXT, G_ARITHMETIC, G_MUL_EXT,      G_RANDOM_ACCESS, G_COSET_INTERPOLATION, G_POSEIDON};
constexpr u32 kBaseSumLimbs = 63, kArithExtOps = 10, kMulExtOps = 13, kReducingCoeffs = 43, kReducingExtCoeffs = 32;
constexpr u32 kRaBits = 4, kRaCopies = 4, kRaExtra = 2;
constexpr u32 kMaxGateTypes = 16;  // array bound of the per-gate-type tables (G_KINDS <= 16)

constexpr u64 kWireFlag = 1ull << 63, kUnusedSelector = 0xFFFFFFFFull;
constexpr u32 kNoSlot = 0xFFFFFFFFu;  // a wire nothing reads: no entry in the value table

inline u64 h_mul(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % kP); }
inline u64 h_add(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % kP); }
inline u64 h_sub(u64 a, u64 b) { return a >= b ? a - b : a + (kP - b); }
inline u64 h_pow(u64 a, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1, a = h_mul(a, a))
    if (e & 1) r = h_mul(r, a);
  return r;
}
inline u64 h_root_of_unity(unsigned log_n) {
  u64 g = h_pow(7, (kP - 1) >> 32);
  for (unsigned i = log_n; i < 32; ++i) g = h_mul(g, g);
  return g;
}

inline u64 wire_t(u32 row, u32 col) { return kWireFlag | ((u64)row << 8) | col; }
inline bool is_wire(u64 t) { return (t & kWireFlag) != 0; }
inline u32 wire_row(u64 t) { return (u32)((t & ~kWireFlag) >> 8); }
inline u32 wire_col(u64 t) { return (u32)(t & 0xFF); }

struct GateInst {
  int kind;
  u64 c[2];
};

// generators (iop/generator.rs + the gates' own): what generate_partial_witness runs
enum {
  GEN_POSEIDON = 0,
  GEN_ARITH = 1,
  GEN_EQUALITY = 2,
  GEN_CONST = 3,
  GEN_ARITH_EXT = 4,      // row, i, c0, c1
  GEN_MUL_EXT = 5,        // row, i, c0
  GEN_QUOTIENT_EXT = 6,   // t = numerator[2], denominator[2], quotient[2]
  GEN_REDUCING = 7,       // row
  GEN_REDUCING_EXT = 8,   // row
  GEN_WIRE_SPLIT = 9,     // t = integer, then the BaseSumGate rows (as wire targets of their sum wires)
  GEN_BASE_SPLIT = 10,    // row
  GEN_RANDOM_ACCESS = 11, // row, i = copy
  GEN_INTERPOLATION = 12, // row
  GEN_POSEIDON_MDS = 13,  // row, i = output element
  // The pre-pass of a ReducingGate / ReducingExtensionGate (round 5; not plonky2 generators: they set no wire the gate's own generator
  // would not set to the same value).  A gate's output is old_acc * alpha^n + H(coefficients): LOCAL computes H and alpha^n without
  // old_acc (t = 4 virtual targets), COMBINE the output from them -- so a chain of k gates hands its accumulator on after one short
  // multiplication per gate instead of one n-step Horner chain per gate, while the gates' own generators fill the n - 1 intermediate wires side by side.
  GEN_REDUCING_LOCAL = 14,      // row; t = H[2], alpha^n[2]
  GEN_REDUCING_EXT_LOCAL = 15,  // row; t as above
  GEN_REDUCING_COMBINE = 16,    // row; t as above
  GEN_KINDS
};
struct Gen {
  int kind;
  u32 row, i;
  u64 c0, c1;
  u64 x, y, eq, inv;   // EqualityGenerator targets
  std::vector<u64> t;  // targets of the generators that are not tied to one gate row
};

typedef std::array<u64, 2> Ext;  // ExtensionTarget<2>
struct ExtConst {
  u64 a, b;
};
inline ExtConst ec_add(ExtConst x, ExtConst y) { return {h_add(x.a, y.a), h_add(x.b, y.b)}; }
inline ExtConst ec_mul(ExtConst x, ExtConst y) { return {h_add(h_mul(x.a, y.a), h_mul(7, h_mul(x.b, y.b))), h_add(h_mul(x.a, y.b), h_mul(x.b, y.a))}; }
inline ExtConst ec_scale(ExtConst x, u64 c) { return {h_mul(x.a, c), h_mul(x.b, c)}; }

}  // namespace p2mt_cb

struct p2mt_circuit_builder {
  typedef uint64_t u64;
  typedef uint32_t u32;
  u64 n_virtual = 0;
  std::vector<p2mt_cb::GateInst> gates;
  std::vector<std::pair<u64, u64>> copies;
  std::vector<p2mt_cb::Gen> gens;
  std::map<u64, u64> const_to_target;  // iterated in increasing canonical order at build()
  std::unordered_map<u64, u64> target_to_const;
  std::map<std::tuple<u64, u64, u64, u64, u64>, u64> arith_results;
  std::map<std::tuple<u64, u64, u64, u64, u64, u64, u64, u64>, p2mt_cb::Ext> ext_arith_results;  // (c0, c1, m0, m1, addend)
  std::map<std::tuple<int, u64, u64>, std::pair<u32, u32>> slots;  // find_slot: (gate kind, params) -> (row, next free operation)
  std::vector<std::array<u32, 3>> constant_generators;              // (row, constant index, wire): consumed in order by build()
  std::vector<u64> public_inputs;
  bool built = false;  // build() consumes the builder, as plonky2's does
};

namespace p2mt_cb {
// ---- CircuitBuilder primitives (p2mt_circuit.hip)
int cb_check(const p2mt_circuit_builder* b, u64 t, bool routable);
u64 cb_virtual(p2mt_circuit_builder* b);
u64 cb_constant(p2mt_circuit_builder* b, u64 c);
int cb_connect(p2mt_circuit_builder* b, u64 x, u64 y);
u32 cb_add_gate(p2mt_circuit_builder* b, int kind, u64 c0 = 0, u64 c1 = 0);
void cb_find_slot(p2mt_circuit_builder* b, int kind, u64 p0, u64 p1, u32 num_ops, u32* row, u32* i);
int cb_arithmetic(p2mt_circuit_builder* b, u64 c0, u64 c1, u64 m0, u64 m1, u64 ad, u64* out);
int cb_permute_swapped(p2mt_circuit_builder* b, u64 (&state)[12], u64 swap);
int cb_hash_no_pad(p2mt_circuit_builder* b, const u64* in, size_t n, u64* out);
int cb_is_equal(p2mt_circuit_builder* b, u64 x, u64 y, u64* out);
}  // namespace p2mt_cb

// What verify_proof reads from `inner_circuit_data.common` / `.verifier_only` (p2mt_circuit.hip fills it from a built circuit)
struct p2mt_common_data {
  uint32_t degree_bits, num_selectors, n_kinds, num_public_inputs;
  uint32_t kind[p2mt_cb::kMaxGateTypes], sel[p2mt_cb::kMaxGateTypes], gs[p2mt_cb::kMaxGateTypes], ge[p2mt_cb::kMaxGateTypes];
  uint64_t k_is[p2mt_cb::kNumRouted];
  p2mt_fri_params fri;
  size_t proof_len;
  uint64_t cs_cap[64], digest[4];
};
int p2mt_circuit_common_data(const p2mt_circuit_data* c, p2mt_common_data* out);  // p2mt_circuit.hip
