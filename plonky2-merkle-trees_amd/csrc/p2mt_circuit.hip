// p2mt_circuit.hip -- CircuitBuilder / CircuitData::prove for the reference's MMR-verifier circuits, device-resident.
//
// Replaces what /root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91 (circuit construction through plonky2's
// CircuitBuilder; gadgets of src/mmr/common.rs:5-58) and :148 (`circuit_data.prove(pw)`) run inside plonky2 (git rev 3b21b87d,
// NOT in the reference tree -- parity unpinned, checked bit for bit against the tests' CPU restatement):
//   host   CircuitBuilder under CircuitConfig::standard_recursion_config() restricted to the gate set those circuits use
//          (NoopGate, ConstantGate, PublicInputGate, ArithmeticGate, PoseidonGate); build(): selector / constant / sigma
//          polynomials and their commitment; the generator schedule (levels of independent generators).
//   device witness fill (k_witness_lds / k_witness_run: one workgroup interprets the levelled generator list out of an LDS or
//          global value table, one wavefront per PoseidonGate row on the 12-lanes-per-permutation layout; k_poseidon_rows then
//          replays every row in parallel to record its S-box input wires), wires commitment, challenger, Z / partial
//          products, the quotient polynomials (k_quotient: four role-wavefronts per 64 points of the 8n-point LDE coset
//          evaluate every gate's constraints, the permutation checks and L_0 (Z - 1), combined with powers of alpha and
//          divided by Z_H), coset IFFT, quotient commitment, openings and the FRI proof (p2mt_fri.hip);
//          verify: the transcript here; Merkle paths and the field arithmetic in p2mt_verify_dev.hip, staged beside it.
// Nothing here is GEMM-shaped: 64-bit modular arithmetic on the integer VALU, latency-bound at these sizes (64..4096 rows).
#include "tree_common.hip.h"
#include "circuit_types.h"
#include "gates_recursion.hip.h"
#include "host_poseidon.h"

#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <new>
#include <thread>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <mutex>
#include <vector>

using namespace p2mt_dev;
using namespace p2mt_cb;
using p2mt::rt;

namespace {

// one generator as the device interpreter sees it.  GEN_POSEIDON / GEN_ARITH / GEN_EQUALITY: a, b, c, out, out2 are value slots
// (GEN_POSEIDON: a = row, b = index into pslots).  The generators of the recursion gates are tied to one gate row: a = row,
// b = operation / copy index, slots come from the dense wire -> slot table; GEN_QUOTIENT_EXT / GEN_WIRE_SPLIT: a = offset of
// their slot list in the argument array, b = its length.
struct WOp {
  u32 kind, a, b, c, out, out2;
  u64 c0, c1;
};

}  // namespace

struct p2mt_partial_witness {
  std::vector<std::pair<u64, u64>> sets;  // (target, canonical value), in call order
};

struct p2mt_circuit_data {
  u32 degree_bits = 0, n = 0, num_selectors = 0, n_kinds = 0, n_cs = 0, n_slots = 0, n_pi = 0, n_act = 0;
  u32 kind[kMaxGateTypes] = {}, sel[kMaxGateTypes] = {}, gs[kMaxGateTypes] = {}, ge[kMaxGateTypes] = {}, counts[kMaxGateTypes] = {};
  bool has_recursion_gates = false;  // any gate type beyond Noop / Constant / PublicInput / Arithmetic / Poseidon
  // memo of fill_witness: the last PartialWitness's target sequence resolved to slots
  bool memo_valid = false;
  std::vector<u64> memo_targets;
  std::vector<u32> memo_slot;
  std::vector<int> memo_first;
  std::unordered_map<u32, u64> memo_const_value;
  bool force_single_workgroup = false;  // set for the one retry after the grid-wide interpreter gave up waiting (see p2mt_circuit_prove)
  u64 n_virtual = 0;
  std::vector<Gen> gens;
  std::vector<u64> public_inputs;
  std::vector<u32> slot_of;  // target index -> value slot (one per copy-constraint class)
  std::vector<std::pair<u32, u64>> const_inits;
  std::vector<u64> h_cs;     // constants_sigmas values [n_cs][n]
  u64 cs_cap[64] = {}, digest[4] = {};
  p2mt_fri_params fri = {};
  size_t fri_len = 0, proof_len = 0, n_digests = 0;
  // generator schedule, valid for the input-slot set it was computed for
  std::vector<u32> sched_inputs;
  bool sched_valid = false;
  u32 n_levels = 0;
  // PoseidonGate rows the HOST evaluates before the launch (single proves; see select_host_chain): in evaluation order
  struct HostOp {
    bool poseidon;   // PoseidonGenerator (in: 12 state words + swap, out: 12) or ArithmeticBaseGenerator (in: m0, m1, addend; out: 1)
    u32 n_in, n_out, in[13], out[12];
    u32 check_mask;  // bit k: out[k] holds a value already when this runs (compared, as a second set_target would be)
    u64 c0, c1;
  };
  std::vector<HostOp> host_chain;
  u32 host_rows = 0;                 // PoseidonGate rows among them
  std::vector<u64> h_vals;           // slot -> value for the slots the host chain reads and writes (constants written once)
  std::vector<u32> memo_input_slots; // the input-slot set of the memoised target sequence (to re-schedule when the mode changes)
  bool sched_host_chain = false;
  // device memory: one allocation, carved
  u64* d_base = nullptr;
  u64 *d_cs_vals = nullptr, *d_cs_coeffs = nullptr, *d_cs_lde = nullptr, *d_cs_leaves = nullptr, *d_cs_dig = nullptr;
  u64 *d_w_vals = nullptr, *d_w_coeffs = nullptr, *d_w_lde = nullptr, *d_w_leaves = nullptr, *d_w_dig = nullptr;
  u64 *d_z_vals = nullptr, *d_z_coeffs = nullptr, *d_z_lde = nullptr, *d_z_leaves = nullptr, *d_z_dig = nullptr, *d_pp_q = nullptr;
  u64 *d_q_vals = nullptr, *d_q_coeffs = nullptr, *d_q_lde = nullptr, *d_q_leaves = nullptr, *d_q_dig = nullptr;
  u64 *d_head = nullptr, *d_open = nullptr, *d_chal = nullptr, *d_kis = nullptr, *d_vals = nullptr, *d_init = nullptr;
  u64* d_q_extra = nullptr;  // [gate type][kNumCh][8n]: the recursion gates' shares of the vanishing polynomial (k_quotient_extra)
  u32 *d_set = nullptr, *d_wire_slot = nullptr, *d_pi_slot = nullptr, *d_lvl = nullptr, *d_pslots = nullptr, *d_prows = nullptr;
  u32 *d_slot_tab = nullptr, *d_args = nullptr, *d_sync = nullptr;  // dense wire -> slot table; slot lists of free-standing generators; grid barrier
  WOp* d_ops = nullptr;
  int* d_err = nullptr;  // [0] witness conflict (op index + 1, or -1 unset public input), [1] zero denominator
  size_t init_cap = 0, ops_cap = 0, args_cap = 0, lds_bytes = 0;  // lds_bytes != 0: the value table fits LDS (k_witness_lds)
  p2mt_challenger* ch = nullptr;
  // verifier scratch (allocated on the first p2mt_circuit_verify)
  u64* d_verify = nullptr;
  void* vstreams = nullptr;  // side streams + events of the staged verifier (p2mt_verify_dev.hip)
  const void* vconst_in[2] = {nullptr, nullptr};
  p2mt::HostLink* link = nullptr;  // single proves with the transcript on the host: caps down, challenges up (runtime.h)
  u64* h_vpin = nullptr;  // pinned staging of the host-side verifier transcript: [pi hash 4 | FriOpenings order | challenges]
  size_t h_vpin_words = 0;
  unsigned vconst_B[2] = {0, 0};  // the block (single / batch) that holds this circuit's digest and constants cap already
  // batched verifier (p2mt_circuit_verify_batch): one block of the same layout per proof, a challenger state behind each
  u64* d_trace = nullptr;  // debug (p2mt_debug_witness_trace): completion tick of every generator of the dataflow interpreter
  char* d_vbatch = nullptr;
  size_t vbatch_cap = 0, vbatch_stride = 0;
  p2mt_challenger* vbch = nullptr;
  // pinned host staging: [0..2) zeta, [2..4) err flags, [8..8+proof_len) proof, then the witness assignments (H2D)
  u64* h_pin = nullptr;
  size_t pin_pairs_off = 0;
  size_t pin_pitch = 0;  // words between the staging areas of consecutive proofs (batched prover; one area otherwise)
  p2mt_challenger* vch = nullptr;
  u64 k_is[kNumRouted] = {};
};

// ------------------------------------------------------------------------------------------------ builder (host)
namespace p2mt_cb {

int cb_check(const p2mt_circuit_builder* b, u64 t, bool routable) {
  if (is_wire(t)) {
    if (wire_row(t) >= b->gates.size() || wire_col(t) >= kNumWires || (t & ~kWireFlag) >> 40) return p2mt::fail(P2MT_EINVAL, "circuit: bad wire target");
    if (routable && wire_col(t) >= kNumRouted) return p2mt::fail(P2MT_EINVAL, "circuit: tried to route a wire that isn't routable");
    return P2MT_OK;
  }
  if (t >= b->n_virtual) return p2mt::fail(P2MT_EINVAL, "circuit: unknown virtual target");
  return P2MT_OK;
}
u64 cb_virtual(p2mt_circuit_builder* b) { return b->n_virtual++; }
u64 cb_constant(p2mt_circuit_builder* b, u64 c) {
  c %= gl::P;
  auto it = b->const_to_target.find(c);
  if (it != b->const_to_target.end()) return it->second;
  const u64 t = cb_virtual(b);
  b->const_to_target[c] = t;
  b->target_to_const[t] = c;
  return t;
}
int cb_connect(p2mt_circuit_builder* b, u64 x, u64 y) {
  P2MT_TRY(cb_check(b, x, true));
  P2MT_TRY(cb_check(b, y, true));
  b->copies.emplace_back(x, y);
  return P2MT_OK;
}
u32 cb_add_gate(p2mt_circuit_builder* b, int kind, u64 c0, u64 c1) {
  b->gates.push_back(GateInst{kind, {c0, c1}});
  const u32 row = (u32)b->gates.size() - 1;
  // Gate::extra_constant_wires: routed wires that can carry a constant; build() hands them out before adding ConstantGates
  if (kind == G_RANDOM_ACCESS)
    for (u32 k = 0; k < kRaExtra; ++k) b->constant_generators.push_back({row, k, 72 + k});
  else if (kind == G_CONSTANT)
    for (u32 k = 0; k < kNumConsts; ++k) b->constant_generators.push_back({row, k, k});
  // Gate::generators of the gates whose generator covers the whole row (per-operation generators are added where the slot is
  // taken: build() drops the generators of unused slots, circuit_builder.rs "Remove unused generators, if any")
  int gk = -1;
  switch (kind) {
    case G_BASE_SUM: gk = GEN_BASE_SPLIT; break;
    case G_REDUCING: gk = GEN_REDUCING; break;
    case G_REDUCING_EXT: gk = GEN_REDUCING_EXT; break;
    case G_COSET_INTERPOLATION: gk = GEN_INTERPOLATION; break;
    case G_POSEIDON_MDS: gk = GEN_POSEIDON_MDS; break;
    default: break;
  }
  if (gk >= 0) {
    Gen g{};
    g.kind = gk;
    g.row = row;
    // PoseidonMdsGenerator writes 12 extension elements, each a 12-term sum of its own: one record per output element (i = r), so that
    // they run on 12 lanes side by side -- as ONE lane generator it took ~18 us, and the in-circuit evaluation of the inner PoseidonGate
    // constraint chains 21 of them (round 5: the witness's critical path once the transcript had left the device)
    const u32 parts = gk == GEN_POSEIDON_MDS ? 12 : 1;
    for (u32 r = 0; r < parts; ++r) {
      g.i = r;
      b->gens.push_back(g);
    }
    if (gk == GEN_REDUCING || gk == GEN_REDUCING_EXT) {  // the pre-pass (circuit_types.h): four virtual targets between its two halves
      Gen l{};
      l.row = row;
      for (int k = 0; k < 4; ++k) l.t.push_back(cb_virtual(b));
      l.kind = gk == GEN_REDUCING ? GEN_REDUCING_LOCAL : GEN_REDUCING_EXT_LOCAL;
      b->gens.push_back(l);
      l.kind = GEN_REDUCING_COMBINE;
      b->gens.push_back(l);
    }
  }
  return row;
}
// circuit_builder.rs find_slot: the next free operation of a multi-operation gate with these parameters
void cb_find_slot(p2mt_circuit_builder* b, int kind, u64 p0, u64 p1, u32 num_ops, u32* row, u32* i) {
  const auto key = std::make_tuple(kind, p0, p1);
  auto sl = b->slots.find(key);
  if (sl != b->slots.end()) {
    *row = sl->second.first;
    *i = sl->second.second;
  } else {
    *row = cb_add_gate(b, kind, p0, p1);
    *i = 0;
  }
  if (*i == num_ops - 1) b->slots.erase(key);
  else b->slots[key] = std::make_pair(*row, *i + 1);
}
// gadgets/arithmetic.rs arithmetic(): const_0 * m0 * m1 + const_1 * addend
int cb_arithmetic(p2mt_circuit_builder* b, u64 c0, u64 c1, u64 m0, u64 m1, u64 ad, u64* out) {
  P2MT_TRY(cb_check(b, m0, true));
  P2MT_TRY(cb_check(b, m1, true));
  P2MT_TRY(cb_check(b, ad, true));
  c0 %= gl::P;
  c1 %= gl::P;
  {  // arithmetic_special_cases
    const u64 zero = cb_constant(b, 0);
    auto cst = [&](u64 t, u64* v) {
      auto it = b->target_to_const.find(t);
      if (it == b->target_to_const.end()) return false;
      *v = it->second;
      return true;
    };
    u64 m0c = 0, m1c = 0, adc = 0;
    const bool h0 = cst(m0, &m0c), h1 = cst(m1, &m1c), ha = cst(ad, &adc);
    const bool first_zero = c0 == 0 || m0 == zero || m1 == zero, second_zero = c1 == 0 || ad == zero;
    const bool first_known = first_zero || (h0 && h1), second_known = second_zero || ha;
    if (first_known && second_known) {
      const u64 f = first_zero ? 0 : h_mul(h_mul(m0c, m1c), c0), s = second_zero ? 0 : h_mul(adc, c1);
      *out = cb_constant(b, h_add(f, s));
      return P2MT_OK;
    }
    if (first_zero && c1 == 1) {
      *out = ad;
      return P2MT_OK;
    }
    if (second_zero) {
      if (h0 && h_mul(m0c, c0) == 1) {
        *out = m1;
        return P2MT_OK;
      }
      if (h1 && h_mul(m1c, c0) == 1) {
        *out = m0;
        return P2MT_OK;
      }
    }
  }
  const auto op = std::make_tuple(c0, c1, m0, m1, ad);
  auto hit = b->arith_results.find(op);
  if (hit != b->arith_results.end()) {
    *out = hit->second;
    return P2MT_OK;
  }
  u32 row, i;
  cb_find_slot(b, G_ARITHMETIC, c0, c1, kNumOps, &row, &i);
  b->copies.emplace_back(m0, wire_t(row, 4 * i));
  b->copies.emplace_back(m1, wire_t(row, 4 * i + 1));
  b->copies.emplace_back(ad, wire_t(row, 4 * i + 2));
  Gen g{};
  g.kind = GEN_ARITH;
  g.row = row;
  g.i = i;
  g.c0 = c0;
  g.c1 = c1;
  b->gens.push_back(g);
  *out = wire_t(row, 4 * i + 3);
  b->arith_results[op] = *out;
  return P2MT_OK;
}
// hash/poseidon.rs permute_swapped: one PoseidonGate row
int cb_permute_swapped(p2mt_circuit_builder* b, u64 (&state)[12], u64 swap) {
  const u32 row = cb_add_gate(b, G_POSEIDON);
  b->copies.emplace_back(swap, wire_t(row, 24));
  for (u32 i = 0; i < 12; ++i) b->copies.emplace_back(state[i], wire_t(row, i));
  Gen g{};
  g.kind = GEN_POSEIDON;
  g.row = row;
  b->gens.push_back(g);
  for (u32 i = 0; i < 12; ++i) state[i] = wire_t(row, 12 + i);
  return P2MT_OK;
}
int cb_hash_no_pad(p2mt_circuit_builder* b, const u64* in, size_t n, u64* out) {
  for (size_t k = 0; k < n; ++k) P2MT_TRY(cb_check(b, in[k], true));
  const u64 zero = cb_constant(b, 0);
  u64 state[12];
  for (auto& s : state) s = zero;
  for (size_t off = 0; off < n; off += 8) {
    for (size_t k = 0; k < 8 && off + k < n; ++k) state[k] = in[off + k];
    P2MT_TRY(cb_permute_swapped(b, state, zero));  // swap = _false()
  }
  for (int k = 0; k < 4; ++k) out[k] = state[k];
  return P2MT_OK;
}
int cb_is_equal(p2mt_circuit_builder* b, u64 x, u64 y, u64* out) {
  P2MT_TRY(cb_check(b, x, true));
  P2MT_TRY(cb_check(b, y, true));
  const u64 zero = cb_constant(b, 0), one = cb_constant(b, 1);
  const u64 equal = cb_virtual(b);
  u64 not_equal, diff, not_equal_check, diff_normalized;
  P2MT_TRY(cb_arithmetic(b, 1, gl::P - 1, one, one, equal, &not_equal));
  const u64 inv = cb_virtual(b);
  Gen g{};
  g.kind = GEN_EQUALITY;
  g.x = x;
  g.y = y;
  g.eq = equal;
  g.inv = inv;
  b->gens.push_back(g);
  P2MT_TRY(cb_arithmetic(b, 1, gl::P - 1, x, one, y, &diff));
  P2MT_TRY(cb_arithmetic(b, 1, 0, diff, inv, diff, &not_equal_check));
  P2MT_TRY(cb_arithmetic(b, 1, 0, diff, equal, diff, &diff_normalized));
  P2MT_TRY(cb_connect(b, diff_normalized, zero));
  P2MT_TRY(cb_connect(b, not_equal, not_equal_check));
  *out = equal;
  return P2MT_OK;
}

}  // namespace p2mt_cb

namespace {

// ------------------------------------------------------------------------------------------------ witness fill (device)
GL_DEV u64 ld64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV void st64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV u32 ld32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV void st32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
GL_DEV u64 fsub(u64 a, u64 b) { return gl::sub_c(a, gl::canon(b)); }

// x^(p-2); inverse of 0 is 0
GL_DEV u64 gl_inv(u64 x) {
  auto sqn = [](u64 v, int k) {
    for (int i = 0; i < k; ++i) v = gl::sqr(v);
    return v;
  };
  const u64 t2 = gl::mul(gl::sqr(x), x);
  const u64 t4 = gl::mul(sqn(t2, 2), t2);
  const u64 t8 = gl::mul(sqn(t4, 4), t4);
  const u64 t16 = gl::mul(sqn(t8, 8), t8);
  const u64 t24 = gl::mul(sqn(t16, 8), t8);
  const u64 t28 = gl::mul(sqn(t24, 4), t4);
  const u64 t30 = gl::mul(sqn(t28, 2), t2);
  const u64 t31 = gl::mul(gl::sqr(t30), x);
  const u64 t32 = gl::mul(gl::sqr(t31), x);
  return gl::mul(sqn(t31, 33), t32);
}

// quadratic extension F[X]/(X^2 - 7) on loose u64 pairs
struct DE {
  u64 a, b;
};
GL_DEV DE de_add(DE x, DE y) { return DE{gl::add(x.a, y.a), gl::add(x.b, y.b)}; }
GL_DEV DE de_sub(DE x, DE y) { return DE{fsub(x.a, y.a), fsub(x.b, y.b)}; }
GL_DEV DE de_mul(DE x, DE y) {
  return DE{gl::mul_add(gl::mul(x.b, y.b), 7, gl::mul(x.a, y.a)), gl::mul_add(x.a, y.b, gl::mul(x.b, y.a))};
}
GL_DEV DE de_scale(DE x, u64 c) { return DE{gl::mul(x.a, c), gl::mul(x.b, c)}; }
// multiplication by a FIXED element (the alpha of a Horner chain): 7 y.b is formed once, and the four products of a step are
// independent of each other -- one multiplication deep instead of two (a chain step is latency-bound on its single wavefront)
struct DEFixed {
  u64 a, b, b7;
};
GL_DEV DEFixed de_fix(DE y) { return DEFixed{y.a, y.b, gl::mul(y.b, 7)}; }
GL_DEV DE de_mul_fixed(DE x, const DEFixed& y) {
  return DE{gl::add(gl::mul(x.a, y.a), gl::mul(x.b, y.b7)), gl::add(gl::mul(x.a, y.b), gl::mul(x.b, y.a))};
}
GL_DEV DE de_inv(DE x) {  // (a - bX) / (a^2 - 7 b^2)
  const u64 n = gl::canon(fsub(gl::sqr(x.a), gl::mul(gl::sqr(x.b), 7)));
  const u64 ni = gl_inv(n);
  return DE{gl::mul(x.a, ni), gl::mul(fsub(0, x.b), ni)};
}

// Where the value table lives while the generators run: global memory (any circuit size; agent-scope accesses so that
// the waves of the workgroup see each other's writes) or LDS (circuits of up to ~18 k value slots, e.g. the 64-row ones:
// a dependent level then costs an LDS round trip instead of an L2 one).
struct GMem {
  u64* vals;
  u32* set;
  GL_DEV u64 get(u32 s) const { return ld64(vals + s); }
  GL_DEV bool is_set(u32 s) const { return ld32(set + s) != 0; }
  GL_DEV void store(u32 s, u64 v) const {
    st64(vals + s, v);
    st32(set + s, 1);
  }
  GL_DEV void sync() const {
    __threadfence();
    __syncthreads();
  }
};
struct LMem {
  u64* vals;
  uint8_t* set;
  GL_DEV u64 get(u32 s) const { return vals[s]; }
  GL_DEV bool is_set(u32 s) const { return set[s] != 0; }
  GL_DEV void store(u32 s, u64 v) const {
    vals[s] = v;
    set[s] = 1;
  }
  GL_DEV void sync() const { __syncthreads(); }
};

// Dataflow table (k_witness_flow): a value is its own "set" flag -- canonical field elements are < p, so the all-ones word never
// occurs as a value and marks an unset slot.  get() WAITS for the slot (bounded: a wait that exceeds its budget raises err[2] and
// from then on every wait returns at once, so the launch always drains).  `set` is kept up to date for k_witness_scatter.
constexpr u64 kUnsetValue = ~0ull;
constexpr u32 kFlowSpinBudget = 1u << 22;
struct FMem {
  u64* vals;
  u32* set;
  int* err;
  GL_DEV u64 get(u32 s) const {
    u64 v = ld64(vals + s);
    u32 spins = 0;
    while (v == kUnsetValue) {
      if (ld32(reinterpret_cast<const u32*>(err) + 2) != 0 || ++spins > kFlowSpinBudget) {
        atomicExch(err + 2, 1);
        return 0;
      }
      // back off: thousands of lanes poll at once, and their loads share the fabric with the producers' stores
      if (spins < 4) __builtin_amdgcn_s_sleep(2);
      else if (spins < 16) __builtin_amdgcn_s_sleep(8);
      else __builtin_amdgcn_s_sleep(32);
      v = ld64(vals + s);
    }
    return v;
  }
  // N operands at once: all N loads are in flight together (a get() per operand is one exposed ~1.2 us round trip each -- a
  // ReducingExtensionGate generator reads 68 operands); the slots that are not written yet fall back to the waiting get()
  template <int N>
  GL_DEV void get_many(const u32 (&sl)[N], u64 (&v)[N]) const {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ld64(vals + sl[i]);
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (v[i] == kUnsetValue) v[i] = get(sl[i]);
  }
  GL_DEV bool is_set(u32 s) const {  // a generator that re-derives an already written slot waits for its first writer, then compares
    (void)get(s);
    return true;
  }
  GL_DEV void store(u32 s, u64 v) const {
    st64(vals + s, v);
    st32(set + s, 1);
  }
  GL_DEV void sync() const {}
};
// the same for the tables that need no waiting (LMem, GMem)
template <typename Mem, int N>
GL_DEV void get_many(const Mem& m, const u32 (&sl)[N], u64 (&v)[N]) {
  if constexpr (std::is_same<Mem, FMem>::value) {
    m.template get_many<N>(sl, v);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = m.get(sl[i]);
  }
}

// PartitionWitness::set_target: a slot already holding a value must agree (plonky2 panics otherwise)
template <typename Mem>
GL_DEV void put(const Mem& m, u32 slot, u64 v, int* err, u32 op_index) {
  v = gl::canon(v);
  if (m.is_set(slot)) {
    if (m.get(slot) != v) atomicCAS(err, 0, (int)op_index + 1);
  } else {
    m.store(slot, v);
  }
}

__global__ __launch_bounds__(kBlock) void k_witness_init(const u64* __restrict__ pairs, u32 n_pairs, u64* vals, u32* set, BatchArg ba) {
  pairs = bp(pairs, ba);
  vals = bp(vals, ba);
  set = bp(set, ba);
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= n_pairs) return;
  const u32 slot = (u32)pairs[2 * t];
  vals[slot] = pairs[2 * t + 1];
  set[slot] = 1;
}

// Which of a generator's outputs it is the FIRST writer of (the scheduler knows).  A first write stores without reading the slot
// (a `put` is a dependent global round trip, and a gate-row generator has up to 86 of them in sequence); any other output is a
// check: the slot is compared with the value derived here -- plonky2's "set twice with different values" panic -- which in the
// dataflow interpreter first waits for the slot's first writer.  Encoding: bit 8 of WOp::kind = every output is a first write (the
// common case).  Otherwise one bit per output, in the order gen_targets() lists them: ArithmeticBase / Equality in bits 9-10 of
// kind; every other generator in the 96 bits (c, out, out2), which those kinds do not use otherwise.
constexpr u32 kFreshOutputs = 0x100, kFreshBit0 = 0x200, kFreshBit1 = 0x400;
constexpr u32 kGenNop = 0xFF;  // padding record of the schedule: no generator
struct Fresh {
  bool all;
  u32 m[3];
  GL_DEV bool at(u32 ord) const { return all || ((m[ord >> 5] >> (ord & 31)) & 1); }
};
GL_DEV Fresh fresh_of(const WOp& op) {
  const u32 kind = op.kind & 0xFF;
  if (kind == GEN_ARITH || kind == GEN_EQUALITY) return Fresh{(op.kind & kFreshOutputs) != 0, {(op.kind >> 9) & 3, 0, 0}};
  return Fresh{(op.kind & kFreshOutputs) != 0, {op.c, op.out, op.out2}};
}
template <typename Mem>
GL_DEV void put_out(const Mem& m, u32 slot, u64 v, bool fresh, int* err, u32 op_index) {
  if (fresh) m.store(slot, gl::canon(v));
  else put(m, slot, v, err, op_index);
}

// The generators of the recursion gates (one lane each).  tab: dense wire -> slot table [row * 135 + col]; args: the slot lists of
// the generators that are not tied to a gate row.  What each computes is the gate's own generator in plonky2:
// ArithmeticExtensionGenerator, MulExtensionGenerator, QuotientGeneratorExtension, ReducingGenerator (both gates),
// WireSplitGenerator, BaseSplitGenerator<2>, RandomAccessGenerator, InterpolationGenerator, PoseidonMdsGenerator.
// Operands are loaded up front (independent loads in flight together), then the dependent arithmetic, then the stores.
// (Inlined on purpose: a noinline version -- tried to keep its operand arrays out of the interpreter's register allocation -- was
// slower in the level-synchronous kernels and faulted in the dataflow one; the role split of k_witness_flow does that job.)
template <typename Mem>
GL_DEV void run_recursion_generator(const Mem& m, const WOp& op, u32 o, const u32* __restrict__ tab, const u32* __restrict__ args,
                                    int* err) {
  const Fresh fr = fresh_of(op);
  const u32* S = tab + (size_t)op.a * kNumWires;  // slots of the row's wires (row-tied generators only)
  auto G = [&](u32 col) { return m.get(S[col]); };
  switch (op.kind & 0xFF) {
    case GEN_ARITH_EXT: {
      const u32 at = 8 * op.b;
      u32 sl[6], os[2];
      u64 in[6];
#pragma unroll
      for (u32 i = 0; i < 6; ++i) sl[i] = S[at + i];
      os[0] = S[at + 6], os[1] = S[at + 7];  // (output slots looked up with the operands: a store never waits for its address)
      get_many(m, sl, in);
      const DE r = de_add(de_scale(de_mul(DE{in[0], in[1]}, DE{in[2], in[3]}), op.c0), de_scale(DE{in[4], in[5]}, op.c1));
      put_out(m, os[0], r.a, fr.at(0), err, o);
      put_out(m, os[1], r.b, fr.at(1), err, o);
      break;
    }
    case GEN_MUL_EXT: {
      const u32 at = 6 * op.b;
      u32 sl[4], os[2];
      u64 in[4];
#pragma unroll
      for (u32 i = 0; i < 4; ++i) sl[i] = S[at + i];
      os[0] = S[at + 4], os[1] = S[at + 5];
      get_many(m, sl, in);
      const DE r = de_scale(de_mul(DE{in[0], in[1]}, DE{in[2], in[3]}), op.c0);
      put_out(m, os[0], r.a, fr.at(0), err, o);
      put_out(m, os[1], r.b, fr.at(1), err, o);
      break;
    }
    case GEN_QUOTIENT_EXT: {
      const u32* A = args + op.a;
      const DE num{m.get(A[0]), m.get(A[1])}, den{m.get(A[2]), m.get(A[3])};
      const DE q = de_mul(num, de_inv(den));
      put_out(m, A[4], q.a, fr.at(0), err, o);
      put_out(m, A[5], q.b, fr.at(1), err, o);
      break;
    }
    // ReducingGate / ReducingExtensionGate: a Horner chain whose every intermediate accumulator is a wire.  The chain is a loop
    // over chunks of 8 coefficient words (fully unrolled it is ~100 KB of straight-line code per generator kind -- more than the
    // instruction cache, and the steps ran at instruction-fetch speed, ~1 us each); a chunk's operands are loaded together, the
    // next chunk's while this one computes.
    case GEN_REDUCING: {
      constexpr u32 kChunk = 8;
      u32 sl[4];
      u64 in[4];
#pragma unroll
      for (u32 i = 0; i < 4; ++i) sl[i] = S[2 + i];
      get_many(m, sl, in);
      const DEFixed alpha = de_fix(DE{in[0], in[1]});
      DE acc{in[2], in[3]};
      // per chunk: the coefficient values and the slots of the accumulator wires the steps write (table lookups: fetched a chunk
      // ahead, or every store would wait for its own address)
      u32 cs[kChunk], os_cur[2 * kChunk], os_nxt[2 * kChunk];
      u64 cur[kChunk], nxt[kChunk];
      auto out_col = [](u32 i) { return i == kReducingCoeffs - 1 ? 0u : 6 + kReducingCoeffs + 2 * i; };
      auto load = [&](u32 base, u64 (&dst)[kChunk], u32 (&od)[2 * kChunk]) {
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) {
          const u32 i = base + j < kReducingCoeffs ? base + j : kReducingCoeffs - 1;
          cs[j] = S[6 + i];
          od[2 * j] = S[out_col(i)];
          od[2 * j + 1] = S[out_col(i) + 1];
        }
        get_many(m, cs, dst);
      };
      load(0, cur, os_cur);
#pragma unroll 1
      for (u32 base = 0; base < kReducingCoeffs; base += kChunk) {
        if (base + kChunk < kReducingCoeffs) load(base + kChunk, nxt, os_nxt);
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) {
          const u32 i = base + j;
          if (i < kReducingCoeffs) {
            acc = de_mul_fixed(acc, alpha);
            acc.a = gl::add(acc.a, cur[j]);
            const u32 ord = i == kReducingCoeffs - 1 ? 0 : 2 + 2 * i;
            put_out(m, os_cur[2 * j], acc.a, fr.at(ord), err, o);
            put_out(m, os_cur[2 * j + 1], acc.b, fr.at(ord + 1), err, o);
          }
        }
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) cur[j] = nxt[j];
#pragma unroll
        for (u32 j = 0; j < 2 * kChunk; ++j) os_cur[j] = os_nxt[j];
      }
      break;
    }
    case GEN_REDUCING_EXT: {
      constexpr u32 kChunk = 4;  // extension coefficients per chunk
      u32 sl[4];
      u64 in[4];
#pragma unroll
      for (u32 i = 0; i < 4; ++i) sl[i] = S[2 + i];
      get_many(m, sl, in);
      const DEFixed alpha = de_fix(DE{in[0], in[1]});
      DE acc{in[2], in[3]};
      u32 cs[2 * kChunk], os_cur[2 * kChunk], os_nxt[2 * kChunk];
      u64 cur[2 * kChunk], nxt[2 * kChunk];
      auto out_col = [](u32 i) { return i == kReducingExtCoeffs - 1 ? 0u : 6 + 2 * kReducingExtCoeffs + 2 * i; };
      auto load = [&](u32 base, u64 (&dst)[2 * kChunk], u32 (&od)[2 * kChunk]) {
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) {
          const u32 i = base + j < kReducingExtCoeffs ? base + j : kReducingExtCoeffs - 1;
          cs[2 * j] = S[6 + 2 * i];
          cs[2 * j + 1] = S[7 + 2 * i];
          od[2 * j] = S[out_col(i)];
          od[2 * j + 1] = S[out_col(i) + 1];
        }
        get_many(m, cs, dst);
      };
      load(0, cur, os_cur);
#pragma unroll 1
      for (u32 base = 0; base < kReducingExtCoeffs; base += kChunk) {
        if (base + kChunk < kReducingExtCoeffs) load(base + kChunk, nxt, os_nxt);
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) {
          const u32 i = base + j;
          if (i < kReducingExtCoeffs) {
            acc = de_add(de_mul_fixed(acc, alpha), DE{cur[2 * j], cur[2 * j + 1]});
            const u32 ord = i == kReducingExtCoeffs - 1 ? 0 : 2 + 2 * i;
            put_out(m, os_cur[2 * j], acc.a, fr.at(ord), err, o);
            put_out(m, os_cur[2 * j + 1], acc.b, fr.at(ord + 1), err, o);
          }
        }
#pragma unroll
        for (u32 j = 0; j < 2 * kChunk; ++j) cur[j] = nxt[j], os_cur[j] = os_nxt[j];
      }
      break;
    }
    // The pre-pass of the two reducing gates (circuit_types.h): H = Horner(coefficients) from a zero accumulator and alpha^n, without
    // the gate's old_acc; then output = old_acc * alpha^n + H.  args: LOCAL alpha[2], coefficient words, then H[2], alpha^n[2];
    // COMBINE old_acc[2], H[2], alpha^n[2], then the output wires.
    case GEN_REDUCING_LOCAL:
    case GEN_REDUCING_EXT_LOCAL: {
      const bool ext = (op.kind & 0xFF) == GEN_REDUCING_EXT_LOCAL;
      const u32* A = args + op.a;
      const u32 n = ext ? kReducingExtCoeffs : kReducingCoeffs, words = ext ? 2 * kReducingExtCoeffs : kReducingCoeffs;
      u32 sl[2] = {A[0], A[1]};
      u64 al[2];
      get_many(m, sl, al);
      const DE a1{al[0], al[1]};
      const DEFixed alpha = de_fix(a1);
      constexpr u32 kChunk = 8;  // coefficient words per batch of loads
      u32 cs[kChunk];
      u64 cur[kChunk], nxt[kChunk];
      auto load = [&](u32 base, u64 (&dst)[kChunk]) {
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) cs[j] = A[2 + (base + j < words ? base + j : words - 1)];
        get_many(m, cs, dst);
      };
      DE acc{0, 0};
      load(0, cur);
#pragma unroll 1
      for (u32 base = 0; base < words; base += kChunk) {
        if (base + kChunk < words) load(base + kChunk, nxt);
        if (ext) {
#pragma unroll
          for (u32 j = 0; j < kChunk; j += 2)
            if (base + j < words) acc = de_add(de_mul_fixed(acc, alpha), DE{cur[j], cur[j + 1]});
        } else {
#pragma unroll
          for (u32 j = 0; j < kChunk; ++j)
            if (base + j < words) {
              acc = de_mul_fixed(acc, alpha);
              acc.a = gl::add(acc.a, cur[j]);
            }
        }
#pragma unroll
        for (u32 j = 0; j < kChunk; ++j) cur[j] = nxt[j];
      }
      DE pw{1, 0}, sq = a1;  // alpha^n, n = 43 or 32
#pragma unroll 1
      for (u32 e = n; e; e >>= 1) {
        if (e & 1) pw = de_mul(pw, sq);
        sq = de_mul(sq, sq);
      }
      const u32* O = A + 2 + words;
      put_out(m, O[0], acc.a, fr.at(0), err, o);
      put_out(m, O[1], acc.b, fr.at(1), err, o);
      put_out(m, O[2], pw.a, fr.at(2), err, o);
      put_out(m, O[3], pw.b, fr.at(3), err, o);
      break;
    }
    case GEN_REDUCING_COMBINE: {
      const u32* A = args + op.a;
      u32 sl[6], os[2] = {A[6], A[7]};
      u64 in[6];
#pragma unroll
      for (u32 i = 0; i < 6; ++i) sl[i] = A[i];
      get_many(m, sl, in);
      const DE r = de_add(de_mul(DE{in[0], in[1]}, DE{in[4], in[5]}), DE{in[2], in[3]});
      put_out(m, os[0], r.a, fr.at(0), err, o);
      put_out(m, os[1], r.b, fr.at(1), err, o);
      break;
    }
    case GEN_WIRE_SPLIT: {  // args: integer, then the sum wires of the BaseSumGates, low limbs first
      const u32* A = args + op.a;
      u64 v = gl::canon(m.get(A[0]));
      for (u32 k = 1; k < op.b; ++k) {
        put_out(m, A[k], v & ((1ull << kBaseSumLimbs) - 1), fr.at(k - 1), err, o);
        v >>= kBaseSumLimbs;
      }
      if (v) atomicCAS(err, 0, (int)o + 1);  // "Integer too large to fit in the BaseSumGates"
      break;
    }
    case GEN_BASE_SPLIT: {
      const u64 v = gl::canon(G(0));
      if (v >> kBaseSumLimbs) atomicCAS(err, 0, (int)o + 1);  // "Integer too large to fit in given number of limbs"
      u32 os[kBaseSumLimbs];  // the limb wires' slots, looked up together
#pragma unroll
      for (u32 j = 0; j < kBaseSumLimbs; ++j) os[j] = S[1 + j];
#pragma unroll
      for (u32 j = 0; j < kBaseSumLimbs; ++j) put_out(m, os[j], (v >> j) & 1, fr.at(j), err, o);
      break;
    }
    case GEN_RANDOM_ACCESS: {
      const u32 at = 18 * op.b;
      u32 os[1 + kRaBits];  // claimed element, then the index bits
      os[0] = S[at + 1];
#pragma unroll
      for (u32 j = 0; j < kRaBits; ++j) os[1 + j] = S[74 + kRaBits * op.b + j];
      const u64 idx = gl::canon(G(at));
      if (idx >= 16) {
        atomicCAS(err, 0, (int)o + 1);  // "Access index is larger than the vector size"
        break;
      }
      put_out(m, os[0], G(at + 2 + (u32)idx), fr.at(0), err, o);
#pragma unroll
      for (u32 j = 0; j < kRaBits; ++j) put_out(m, os[1 + j], (idx >> j) & 1, fr.at(1 + j), err, o);
      break;
    }
    case GEN_INTERPOLATION: {
      u32 sl[35];
      u64 in[35];  // shift, 16 values, evaluation point: wires 0 .. 34
#pragma unroll
      for (u32 i = 0; i < 35; ++i) sl[i] = S[i];
      get_many(m, sl, in);
      DE vals[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) vals[i] = DE{in[1 + 2 * i], in[2 + 2 * i]};
      u32 os[12];  // wires 35 .. 46: the outputs
#pragma unroll
      for (u32 i = 0; i < 12; ++i) os[i] = S[35 + i];
      auto PO = [&](u32 col, DE v, u32 ord) {
        put_out(m, os[col - 35], v.a, fr.at(ord), err, o);
        put_out(m, os[col - 34], v.b, fr.at(ord + 1), err, o);
      };
      const DE point{in[33], in[34]};
      const DE x = de_scale(point, gl_inv(gl::canon(in[0])));  // shifted_evaluation_point = evaluation_point / shift
      PO(45, x, 10);
      DE ev{0, 0}, pr{1, 0};
      auto step = [&](int i) {
        const DE term{gl::sub_c(x.a, gates_rec::kCosetDomainDev[i]), x.b};
        const DE weighted = de_scale(vals[i], gates_rec::kCosetWeightsDev[i]);
        ev = de_add(de_mul(ev, term), de_mul(weighted, pr));
        pr = de_mul(pr, term);
      };
#pragma unroll
      for (int i = 0; i < 6; ++i) step(i);
      PO(37, ev, 2);
      PO(41, pr, 6);
#pragma unroll
      for (int i = 6; i < 11; ++i) step(i);
      PO(39, ev, 4);
      PO(43, pr, 8);
#pragma unroll
      for (int i = 11; i < 16; ++i) step(i);
      PO(35, ev, 0);
      break;
    }
    case GEN_POSEIDON_MDS: {
      u32 sl[24];
      u64 in[24];
#pragma unroll
      for (u32 i = 0; i < 24; ++i) sl[i] = S[i];
      get_many(m, sl, in);
      const u32 r = op.b;  // this record's output element: sum_i circ[i] * state[(i + r) % 12] (+ diag[0] * state[0] for r = 0)
      const u32 os0 = S[24 + 2 * r], os1 = S[25 + 2 * r];
      DE st[12];
#pragma unroll
      for (u32 i = 0; i < 12; ++i) st[i] = DE{in[2 * i], in[2 * i + 1]};
      DE acc = r == 0 ? de_scale(st[0], 8) : DE{0, 0};
      // state word j meets circ[(j - r) mod 12]: the state stays in registers under static indices, the small constants (< 64) are
      // picked from two packed words by the run-time r
      u64 lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) lo |= (u64)gates_rec::mds_circ(k) << (6 * k), hi |= (u64)gates_rec::mds_circ(6 + k) << (6 * k);
#pragma unroll
      for (u32 j = 0; j < 12; ++j) {
        const u32 idx = j >= r ? j - r : j + 12 - r;
        const u64 cj = ((idx < 6 ? lo >> (6 * idx) : hi >> (6 * (idx - 6)))) & 63;
        acc = de_add(acc, de_scale(st[j], cj));
      }
      put_out(m, os0, acc.a, fr.at(0), err, o);
      put_out(m, os1, acc.b, fr.at(1), err, o);
      break;
    }
    default: break;
  }
}

// ArithmeticBaseGenerator / EqualityGenerator (slots in the record itself) or one of the generators above
template <typename Mem>
GL_DEV void run_lane_generator(const Mem& m, const WOp& op, u32 o, const u32* __restrict__ tab, const u32* __restrict__ args, int* err) {
  const u32 kind = op.kind & 0xFF;
  if (kind == GEN_ARITH) {
    const u64 m0 = m.get(op.a), m1 = m.get(op.b), ad = m.get(op.c);
    put_out(m, op.out, gl::mul_add(gl::mul(m0, m1), op.c0, gl::mul(ad, op.c1)), (op.kind & (kFreshOutputs | kFreshBit0)) != 0, err, o);
  } else if (kind == GEN_EQUALITY) {
    const u64 x = m.get(op.a), y = m.get(op.b);
    put_out(m, op.out, x == y ? 1 : 0, (op.kind & (kFreshOutputs | kFreshBit0)) != 0, err, o);
    put_out(m, op.out2, gl_inv(gl::canon(fsub(x, y))), (op.kind & (kFreshOutputs | kFreshBit1)) != 0, err, o);
  } else {
    run_recursion_generator(m, op, o, tab, args, err);
  }
}

// PoseidonGenerator on one wavefront (lane w < 12 owns state word w): reads the 12 inputs and the swap bit, writes the 12 outputs.
// ps: this lane's slot among the row's wires 0..24 (lane < 25).  fresh_mask: bit w = output word w is a first write.
// CHAIN (dataflow interpreter): the wavefront remembers the outputs of the row it ran last (slot and value per lane); an input of
// this row that IS one of those slots is taken from the register instead of waiting for the store -> load hand-off -- the
// transcript is a chain of ~110 such rows on one wavefront, each feeding its whole state (or its capacity words) to the next.
template <bool CHAIN, typename Mem>
GL_DEV void run_poseidon_generator(const Mem& m, u32 ps, u32 lane, u32 o, u32 fresh_mask, int* err, const PermCtx& ctx,
                                   u32* prev_slot = nullptr, u64* prev_val = nullptr) {
  const u32 swap_slot = __shfl(ps, 24), out_slot = __shfl(ps, (lane + 12) & 31);
  u64 x = 0;
  bool have = lane >= 12;
  if constexpr (CHAIN) {
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
      const u32 sj = __shfl(*prev_slot, j);
      const u64 vj = __shfl((unsigned long long)*prev_val, j);
      if (!have && sj == ps) {
        x = vj;
        have = true;
      }
    }
  }
  if (!have) x = m.get(ps);
  const u64 swap = m.get(swap_slot);
  const u64 partner = __shfl_xor((unsigned long long)x, 4);
  if (lane < 4) x = gl::add(x, gl::mul(swap, fsub(partner, x)));       // the permutation runs on the swapped state
  else if (lane < 8) x = fsub(x, gl::mul(swap, fsub(x, partner)));
  x = permute_wave(x, ctx);  // outputs only: the row's delta / S-box wires are filled afterwards (k_poseidon_rows)
  x = gl::canon(x);
  if (lane < 12) put_out(m, out_slot, x, ((fresh_mask >> lane) & 1) != 0, err, o);
  if constexpr (CHAIN) {
    *prev_slot = lane < 12 ? out_slot : kNoSlot;
    *prev_val = x;
  }
}
GL_DEV u32 poseidon_fresh_mask(u32 kind, u32 c) { return (kind & kFreshOutputs) ? 0xFFFu : c; }

// One workgroup runs the generators level by level (a level = generators whose inputs are all known; the host orders them
// and puts the PoseidonGate rows first).  PoseidonGenerator: one wavefront per row, lane w < 12 owns state word w.
// The PoseidonGate's non-routed wires (delta and S-box inputs: 110 of the row's 135) are never read by a generator nor
// copy-constrained, so they have no value slot and are not computed here at all: the level loop is a dependency chain, and
// recording them costs ~40 % on top of every chained permutation.  k_poseidon_rows recomputes every row in parallel, with
// the recording hook, once the wire matrix holds the rows' inputs.
template <typename Mem, u32 BLK = kBlock>
GL_DEV void run_levels(const Mem& m, const WOp* __restrict__ ops, const u32* __restrict__ lvl, u32 n_levels,
                       const u32* __restrict__ pslots, const u32* __restrict__ tab, const u32* __restrict__ args, int* err,
                       const PermCtx& ctx) {
  // A level should cost its LDS reads, its arithmetic and a barrier, not a chain of global-memory round trips: the level
  // bounds sit in LDS (when they fit), the generator records are staged into LDS a chunk of whole levels at a time (one
  // exposed load latency per <= 512 generators instead of one per level), and each wavefront's PoseidonGate row for level
  // l + 1 is looked up while level l runs.
  constexpr u32 kLvlLds = 1024, kChunk = 512;
  __shared__ u32 lvl_lds[2 * kLvlLds + 1];
  __shared__ WOp ops_lds[kChunk];
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = BLK / 64;
  const u32* L = lvl;
  if (n_levels <= kLvlLds) {
    for (u32 k = tid; k < 2 * n_levels + 1; k += BLK) lvl_lds[k] = lvl[k];
    __syncthreads();
    L = lvl_lds;
  }
  u32 ch_begin = 0, ch_end = 0;  // generators [ch_begin, ch_end) are in ops_lds
  auto OP = [&](u32 o) -> WOp { return (o >= ch_begin && o < ch_end) ? ops_lds[o - ch_begin] : ops[o]; };
  auto stage = [&](u32 l) {  // all threads, between two barriers of the level loop
    const u32 s = L[2 * l], e = L[2 * l + 2];
    if (e <= ch_end && s >= ch_begin) return;
    ch_begin = ch_end = 0;
    if (e - s > kChunk) return;  // a level wider than the chunk is read from global memory
    u32 last = l;
    while (last + 1 < n_levels && L[2 * (last + 2)] - s <= kChunk) ++last;
    const u32 end = L[2 * (last + 1)];
    for (u32 k = tid; k < end - s; k += BLK) ops_lds[k] = ops[s + k];
    __syncthreads();
    ch_begin = s;
    ch_end = end;
  };
  auto lookup = [&](u32 l, u32& ps) {  // this wavefront's first PoseidonGate row of level l
    const u32 s = L[2 * l], np = L[2 * l + 1];
    if (wave < np) {
      const WOp po = OP(s + wave);
      ps = lane < 25 ? pslots[(size_t)po.b * 32 + lane] : 0;
    }
  };
  u32 cur_ps = 0, nxt_ps = 0;
  if (n_levels) {
    stage(0);
    lookup(0, cur_ps);
  }
  for (u32 l = 0; l < n_levels; ++l) {
    const u32 s = L[2 * l], np = L[2 * l + 1], e = L[2 * l + 2];
    if (l + 1 < n_levels && L[2 * (l + 1) + 2] <= ch_end) lookup(l + 1, nxt_ps);  // else: after the next staging
    for (u32 o = s + wave; o < s + np; o += n_waves) {  // wave-uniform
      u32 ps = cur_ps;
      const WOp po = OP(o);
      if (o != s + wave) ps = lane < 25 ? pslots[(size_t)po.b * 32 + lane] : 0;  // more PoseidonGate rows in this level than wavefronts
      run_poseidon_generator<false>(m, ps, lane, o, poseidon_fresh_mask(po.kind, po.c), err, ctx);
    }
    for (u32 o = s + np + tid; o < e; o += BLK) run_lane_generator(m, OP(o), o, tab, args, err);
    m.sync();
    if (l + 1 < n_levels) {
      if (L[2 * (l + 1) + 2] <= ch_end) {
        cur_ps = nxt_ps;
      } else {
        stage(l + 1);
        lookup(l + 1, cur_ps);
      }
    }
  }
}

// BLK = 256: the fallback of a single proof; BLK = 1024 (16 wavefronts: 16 PoseidonGate rows of a level at a time) for the proofs
// of a batch, which get one workgroup each.
template <u32 BLK>
__global__ __launch_bounds__(BLK) void k_witness_run(const WOp* __restrict__ ops, const u32* __restrict__ lvl, u32 n_levels,
                                                     u64* vals, u32* set, const u32* __restrict__ pslots,
                                                     const u32* __restrict__ tab, const u32* __restrict__ args, int* err,
                                                     BatchArg ba, PermCtx ctx) {
  vals = bp(vals, ba);
  set = bp(set, ba);
  err = bp(err, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  run_levels<GMem, BLK>(GMem{vals, set}, ops, lvl, n_levels, pslots, tab, args, err, ctx);
}

// The same interpreter spread over the whole grid, for circuits with wide levels (the recursion's outer circuit: 28 FRI query
// rounds side by side, each a chain of PoseidonGate rows).  Every level: the PoseidonGate rows go one per wavefront over ALL
// wavefronts of the grid, the other generators one per lane over all lanes; then a grid-wide barrier.  The barrier is a counter
// in global memory: one lane per workgroup adds 1 and waits until every workgroup of this level has (counter >= (level + 1) *
// gridDim.x).  Every workgroup must be resident for this to terminate, so the launch uses at most kGridBlocks workgroups
// (far below what the device holds); as a last resort a wait that exceeds its spin budget raises err[2] and every workgroup
// leaves its loop -- a hung launch is never left behind.
constexpr u32 kGridBlocks = 64, kGridSpinBudget = 1u << 24;
__global__ __launch_bounds__(kBlock) void k_witness_grid(const WOp* __restrict__ ops, const u32* __restrict__ lvl, u32 n_levels,
                                                         u64* vals, u32* set, const u32* __restrict__ pslots,
                                                         const u32* __restrict__ tab, const u32* __restrict__ args, int* err,
                                                         u32* sync, PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  __shared__ int abort_flag;
  ctx = stage_round_constants(rc_lds, ctx);
  const GMem m{vals, set};
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = kBlock / 64;
  const u32 gwave = blockIdx.x * waves_per_block + wave, n_gwaves = gridDim.x * waves_per_block;
  const u32 gtid = blockIdx.x * kBlock + tid, n_gthreads = gridDim.x * kBlock;
  if (tid == 0) abort_flag = 0;
  __syncthreads();
  for (u32 l = 0; l < n_levels; ++l) {
    const u32 s = lvl[2 * l], np = lvl[2 * l + 1], e = lvl[2 * l + 2];
    for (u32 o = s + gwave; o < s + np; o += n_gwaves) {  // wave-uniform
      const u32 pb = ops[o].b, pk = ops[o].kind, pc = ops[o].c;
      const u32 ps = lane < 25 ? pslots[(size_t)pb * 32 + lane] : 0;
      run_poseidon_generator<false>(m, ps, lane, o, poseidon_fresh_mask(pk, pc), err, ctx);
    }
    for (u32 o = s + np + gtid; o < e; o += n_gthreads) run_lane_generator(m, ops[o], o, tab, args, err);
    // grid-wide barrier
    __threadfence();
    __syncthreads();
    if (tid == 0) {
      const u32 target = (l + 1) * gridDim.x;
      atomicAdd(sync, 1u);
      u32 spins = 0;
      while (ld32(sync) < target) {
        if (ld32(reinterpret_cast<const u32*>(err) + 2) != 0 || ++spins > kGridSpinBudget) {
          atomicExch(err + 2, 1);
          abort_flag = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (abort_flag) return;  // block-uniform
  }
}

// Dataflow form of the same schedule (the default for circuits whose table lives in global memory): no barrier at all.  The
// generators keep their level order and their static assignment (PoseidonGate rows one per wavefront over all wavefronts of the
// grid, the others one per lane over all lanes), but each simply WAITS for its operands (FMem::get) instead of for the whole
// level.  Deadlock-free as long as every workgroup is resident (kGridBlocks << what the device holds): every wavefront runs its
// generators in non-decreasing level order and a generator only reads slots written at strictly lower levels, so the unfinished
// generator of lowest level can always run.  What it buys: a dependent step costs one store -> load hand-off (~2 us) instead of
// a grid-wide barrier plus the slowest generator of the level (~15 us on top of a PoseidonGate row's 9.5 us); the outer
// recursion circuit's 124 levels took 3.2 ms with barriers, 1.7 ms this way (critical path: the 110-permutation transcript).
__global__ __launch_bounds__(kBlock) void k_witness_flow(const WOp* __restrict__ ops, const u32* __restrict__ lvl, u32 n_levels,
                                                         u64* vals, u32* set, const u32* __restrict__ pslots,
                                                         const u32* __restrict__ tab, const u32* __restrict__ args, int* err,
                                                         u64* __restrict__ trace, PermCtx ctx) {
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const FMem m{vals, set, err};
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves_per_block = kBlock / 64;
  const u32 gwave = blockIdx.x * waves_per_block + wave, n_gwaves = gridDim.x * waves_per_block;
  // Roles: three quarters of the wavefronts run PoseidonGate rows only, the rest the lane generators only.  A wavefront that did
  // both would tie the transcript -- the critical chain, always the first PoseidonGate row of its level -- to the progress of
  // unrelated arithmetic chains, and the whole schedule would degenerate to level-synchronous (measured: no gain over barriers).
  const u32 n_pw = n_gwaves - n_gwaves / 4;
  if (gwave < n_pw) {
    u32 prev_slot = kNoSlot;
    u64 prev_val = 0;
    for (u32 l = 0; l < n_levels; ++l) {
      const u32 s = lvl[2 * l], np = lvl[2 * l + 1];
      for (u32 o = s + gwave; o < s + np; o += n_pw) {  // wave-uniform
        const u32 pb = ops[o].b, pk = ops[o].kind, pc = ops[o].c;
        const u32 ps = lane < 25 ? pslots[(size_t)pb * 32 + lane] : 0;
        run_poseidon_generator<true>(m, ps, lane, o, poseidon_fresh_mask(pk, pc), err, ctx, &prev_slot, &prev_val);
        if (trace && lane == 0) trace[o] = wall_clock64();  // (debug: completion time of every generator, 100 MHz ticks)
      }
    }
  } else {
    const u32 lt = (gwave - n_pw) * 64 + lane, n_lt = (n_gwaves - n_pw) * 64;
    for (u32 l = 0; l < n_levels; ++l) {
      const u32 s = lvl[2 * l], np = lvl[2 * l + 1], e = lvl[2 * l + 2];
      for (u32 o = s + np + lt; o < e; o += n_lt) {
        run_lane_generator(m, ops[o], o, tab, args, err);
        if (trace) trace[o] = wall_clock64();
      }
    }
  }
}


// The whole witness fill in one launch with the value table in LDS: initial assignments, generator levels, and
// full_witness (wires[col][row], public inputs) straight from LDS.
__global__ __launch_bounds__(kBlock) void k_witness_lds(const u64* __restrict__ pairs, u32 n_pairs, u32 n_slots,
                                                        const WOp* __restrict__ ops, const u32* __restrict__ lvl, u32 n_levels,
                                                        const u32* __restrict__ pslots, const u32* __restrict__ wire_slot, u32 n_act,
                                                        u32 log_n, u64* __restrict__ wires,
                                                        const u32* __restrict__ pi_slot, u32 n_pi, u64* __restrict__ pi_out,
                                                        const u32* __restrict__ tab, const u32* __restrict__ args,
                                                        int* err, BatchArg ba, PermCtx ctx) {
  pairs = bp(pairs, ba);
  wires = bp(wires, ba);
  pi_out = bp(pi_out, ba);
  err = bp(err, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  extern __shared__ __attribute__((aligned(16))) u64 sh[];
  const LMem m{sh, reinterpret_cast<uint8_t*>(sh + n_slots)};
  for (u32 k = threadIdx.x; k < n_slots; k += kBlock) m.set[k] = 0;
  for (u32 t = threadIdx.x; t < (kNumWires << log_n); t += kBlock) wires[t] = 0;  // wires nothing sets are zero
  if (threadIdx.x < 3) err[threadIdx.x] = 0;
  __syncthreads();
  for (u32 k = threadIdx.x; k < n_pairs; k += kBlock) m.store((u32)pairs[2 * k], pairs[2 * k + 1]);
  __syncthreads();
  run_levels(m, ops, lvl, n_levels, pslots, tab, args, err, ctx);
  // full_witness: only the wires that own a slot (list of (wire index, slot) pairs); everything else was zero-filled before
  // the launch or written by the PoseidonGate rows above
  for (u32 k = threadIdx.x; k < n_act; k += kBlock) {
    const u32 t = wire_slot[2 * k], s = wire_slot[2 * k + 1];
    if (m.set[s]) wires[t] = m.vals[s];
  }
  for (u32 t = threadIdx.x; t < n_pi; t += kBlock) {
    const u32 s = pi_slot[t];
    if (!m.set[s]) atomicCAS(err, 0, -1);
    pi_out[t] = m.vals[s];
  }
}

// PartitionWitness::full_witness for the global-memory path: wires[col][row] of every wire that owns a slot + the public inputs
__global__ __launch_bounds__(kBlock) void k_witness_scatter(const u64* __restrict__ vals, const u32* __restrict__ set,
                                                            const u32* __restrict__ act, u32 n_act, u64* __restrict__ wires,
                                                            const u32* __restrict__ pi_slot, u32 n_pi, u64* __restrict__ pi_out,
                                                            int* err, BatchArg ba) {
  vals = bp(vals, ba);
  set = bp(set, ba);
  wires = bp(wires, ba);
  pi_out = bp(pi_out, ba);
  err = bp(err, ba);
  const u32 k = blockIdx.x * kBlock + threadIdx.x;
  if (k < n_pi) {
    const u32 s = pi_slot[k];
    if (!set[s]) atomicCAS(err, 0, -1);
    pi_out[k] = vals[s];
  }
  if (k >= n_act) return;
  const u32 t = act[2 * k], s = act[2 * k + 1];
  if (set[s]) wires[t] = vals[s];
}

// PoseidonGenerator's non-routed outputs for every PoseidonGate row at once: one wavefront per row reads the row's inputs and
// swap bit from the wire matrix and replays the permutation with a hook that stores delta (wires 25-28) and every S-box
// input (wires 29-134).
__global__ __launch_bounds__(kBlock) void k_poseidon_rows(const u32* __restrict__ rows, u32 n_rows, u64* __restrict__ wires,
                                                          u32 log_n, BatchArg ba, PermCtx ctx) {
  wires = bp(wires, ba);
  __shared__ u64 rc_lds[kWaveRcWords];
  ctx = stage_round_constants(rc_lds, ctx);
  const u32 k = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (k >= n_rows) return;  // wave-uniform
  u64* wrow = wires + rows[k];  // wire (row, col) = wrow[col << log_n]
  u64 x = lane < 12 ? wrow[(size_t)lane << log_n] : 0;
  const u64 swap = wrow[(size_t)24 << log_n];
  const u64 partner = __shfl_xor((unsigned long long)x, 4);
  if (lane < 4) {  // delta_i = swap * (in[i+4] - in[i])
    const u64 d = gl::canon(gl::mul(swap, fsub(partner, x)));
    wrow[(size_t)(25 + lane) << log_n] = d;
    x = gl::add(x, d);
  } else if (lane < 8) {
    x = fsub(x, gl::mul(swap, fsub(x, partner)));
  }
  (void)permute_wave_hook(x, ctx, [&](int r, u64 xv) {
    if (lane >= 12 || r == 0) return;
    u32 col;
    if (r < 4) col = 29 + 12 * (r - 1) + lane;
    else if (r < 26) {
      if (lane != 0) return;
      col = 65 + (r - 4);
    } else col = 87 + 12 * (r - 26) + lane;
    wrow[(size_t)col << log_n] = gl::canon(xv);
  });
}

// ------------------------------------------------------------------------------------------------ quotient (device)
struct QDesc {
  u32 log_n, n_kinds, num_selectors, n_cs;
  u32 kind[kMaxGateTypes], sel[kMaxGateTypes], gs[kMaxGateTypes], ge[kMaxGateTypes];
  u64 zh[kQF], zh_inv[kQF];  // Z_H(x_i) = 7^n w_8^(i mod 8) - 1 and its inverse
  u64 n_inv, w_big;
};

// compute_quotient_polys + eval_vanishing_poly_base on the LDE coset x_i = 7 w^i.  A workgroup owns 64 points; its four
// wavefronts split every point's work by ROLE (wave-uniform branches, so nothing diverges) and meet in LDS:
//   role 0  PoseidonGate: the transition out of the first half + the 22-round partial chain (the one sequential piece)
//   role 1  PoseidonGate: input layer, full rounds 0-2 of the first half, first full round of the second half; 2 pp chunks
//   role 2  PoseidonGate: last three full rounds; Arithmetic / Constant / PublicInput gates; L_0(x)(Z - 1); 4 pp chunks
//   role 3  14 of the 20 partial-product checks
// (every full round is independent of the others because its S-box inputs are wires; only the partial rounds chain).
// terms: L_0 (Z_c - 1) | partial-product checks | gate constraints (filtered), reduced with powers of alpha_c from a table
// built once per workgroup in LDS.  The committed batches are read poly-major in leaf order ([poly][brev(i)]: coalesced
// over lanes); the quotient values go out in natural order for the coset IFFT.
constexpr u32 kNumTerms = kNumCh + kNumCh * kNumChunks + kNumGateConstraints;
__global__ __launch_bounds__(256) void k_quotient(const QDesc d, const u64* __restrict__ cs, const u64* __restrict__ wl,
                                                  const u64* __restrict__ zl, const u64* __restrict__ pi_hash,
                                                  const u64* __restrict__ chal, const u64* __restrict__ k_is,
                                                  const u64* __restrict__ rc, const u64* __restrict__ extra, u32 n_extra,
                                                  u64* __restrict__ qvals, BatchArg ba) {
  wl = bp(wl, ba);
  zl = bp(zl, ba);
  pi_hash = bp(pi_hash, ba);
  chal = bp(chal, ba);
  extra = bp(extra, ba);
  qvals = bp(qvals, ba);
  __shared__ u64 apow[kNumCh][kNumTerms];
  __shared__ u64 part[4][64][kNumCh];
  const u32 log_big = d.log_n + 3, big = 1u << log_big;
  const u32 role = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const u32 r_raw = blockIdx.x * 64 + lane;
  const bool valid = r_raw < big;
  const u32 r = valid ? r_raw : big - 1;
  for (u32 k = threadIdx.x; k < kNumCh * kNumTerms; k += 256) {
    const u32 c = k / kNumTerms, e = k % kNumTerms;
    apow[c][e] = gl::canon(gl::pow(chal[2 * kNumCh + c], e));
  }
  __syncthreads();
  const u32 i = __brev(r) >> (32 - log_big);
  const u32 r_next = __brev((i + kQF) & (big - 1)) >> (32 - log_big);
  auto CS = [&](u32 j) { return cs[(size_t)j * big + r]; };
  auto W = [&](u32 j) { return wl[(size_t)j * big + r]; };
  auto Z = [&](u32 j) { return zl[(size_t)j * big + r]; };
  auto ZN = [&](u32 j) { return zl[(size_t)j * big + r_next]; };
  u64 acc[kNumCh], gacc[kNumCh];
#pragma unroll
  for (u32 c = 0; c < kNumCh; ++c) acc[c] = gacc[c] = 0;
  auto T = [&](u32 k, u64 t) {  // term k of the vanishing polynomial
#pragma unroll
    for (u32 c = 0; c < kNumCh; ++c) acc[c] = gl::mul_add(apow[c][k], t, acc[c]);
  };
  auto G = [&](u32 j, u64 t) {  // constraint j of the gate being evaluated (filtered when the gate is closed)
#pragma unroll
    for (u32 c = 0; c < kNumCh; ++c) gacc[c] = gl::mul_add(apow[c][kNumCh + kNumCh * kNumChunks + j], t, gacc[c]);
  };
  auto close_gate = [&](u32 g) {  // compute_filter, then acc += filter * gate sum
    const u64 sv = CS(d.sel[g]);
    u64 f = 1;
    for (u32 k = d.gs[g]; k < d.ge[g]; ++k)
      if (k != g) f = gl::mul(f, fsub((u64)k, sv));
    if (d.num_selectors > 1) f = gl::mul(f, fsub(kUnusedSelector, sv));
#pragma unroll
    for (u32 c = 0; c < kNumCh; ++c) {
      acc[c] = gl::mul_add(f, gacc[c], acc[c]);
      gacc[c] = 0;
    }
  };
  // position of each gate type this kernel evaluates in the circuit's sorted list, -1 if absent (recursion gates: k_quotient_extra)
  int gp = -1, g_arith = -1, g_const = -1, g_pi = -1;
  for (u32 g = 0; g < d.n_kinds; ++g) {
    const u32 k = d.kind[g];
    if (k == G_POSEIDON) gp = (int)g;
    else if (k == G_ARITHMETIC) g_arith = (int)g;
    else if (k == G_CONSTANT) g_const = (int)g;
    else if (k == G_PUBLIC_INPUT) g_pi = (int)g;
  }
  const u32 sig0 = d.num_selectors + kNumConsts;
  // partial-product check `idx` = challenge * 10 + chunk (term kNumCh + idx)
  auto pp_check = [&](u32 idx) {
    const u32 c = idx / kNumChunks, q = idx % kNumChunks;
    const u64 x = gl::mul(7, gl::pow(d.w_big, i));
    const u64 beta = chal[c], gamma = chal[kNumCh + c], bx = gl::mul(beta, x);
    u64 num = 1, den = 1;
#pragma unroll 1
    for (u32 j = q * kQF; j < (q + 1) * kQF && j < kNumRouted; ++j) {
      const u64 wg = gl::add(W(j), gamma);
      num = gl::mul(num, gl::mul_add(bx, k_is[j], wg));
      den = gl::mul(den, gl::mul_add(beta, CS(sig0 + j), wg));
    }
    const u64 prev = q == 0 ? Z(c) : Z(kNumCh + c * kNumProds + q - 1);
    const u64 next = q == kNumProds ? ZN(c) : Z(kNumCh + c * kNumProds + q);
    T(kNumCh + idx, fsub(gl::mul(prev, num), gl::mul(next, den)));
  };
  // one full round of the PoseidonGate from its S-box inputs: S-box, MDS, next round's constants (none after the last
  // round), constraints j0.. against the wires starting at cmp
  auto full_step = [&](u64 (&s)[12], const u64* rc_next, u32 cmp, u32 j0) {
#pragma unroll
    for (u32 k = 0; k < 12; ++k) s[k] = gl::pow7(s[k]);
    poseidon::mds_layer<poseidon::MDS_MAD64>(s);
    if (rc_next) {
#pragma unroll
      for (u32 k = 0; k < 12; ++k) s[k] = gl::add_c(s[k], rc_next[k]);
    }
#pragma unroll
    for (u32 k = 0; k < 12; ++k) G(j0 + k, fsub(s[k], W(cmp + k)));
  };
  auto load_row = [&](u64 (&s)[12], u32 base) {
#pragma unroll
    for (u32 k = 0; k < 12; ++k) s[k] = W(base + k);
  };
  if (role == 0) {
    if (gp >= 0) {
      u64 s[12];
      load_row(s, 53);  // S-box inputs of full round 3
#pragma unroll
      for (u32 k = 0; k < 12; ++k) s[k] = gl::pow7(s[k]);
      poseidon::mds_layer<poseidon::MDS_MAD64>(s);
#pragma unroll
      for (u32 k = 0; k < 12; ++k) s[k] = gl::add_c(s[k], rc[48 + k]);
#pragma unroll 1
      for (u32 pr = 0; pr < POSEIDON_PARTIAL_ROUNDS; ++pr) {
        const u64 in = W(65 + pr);
        G(41 + pr, fsub(s[0], in));
        s[0] = gl::pow7(in);
        poseidon::mds_layer<poseidon::MDS_MAD64>(s);
        const u64* rcn = rc + 12 * (5 + pr);
#pragma unroll
        for (u32 k = 0; k < 12; ++k) s[k] = gl::add_c(s[k], rcn[k]);
      }
#pragma unroll
      for (u32 k = 0; k < 12; ++k) G(63 + k, fsub(s[k], W(87 + k)));
      close_gate((u32)gp);
    }
  } else if (role == 1) {
    if (gp >= 0) {
      const u64 swap = W(24);
      G(0, gl::mul(swap, gl::sub_c(swap, 1)));
      u64 s[12];
#pragma unroll
      for (u32 k = 0; k < 4; ++k) {
        const u64 lhs = W(k), rhs = W(k + 4), delta = W(25 + k);
        G(1 + k, fsub(gl::mul(swap, gl::sub_c(rhs, lhs)), delta));
        s[k] = gl::add(lhs, delta);
        s[k + 4] = gl::sub_c(rhs, delta);
      }
#pragma unroll
      for (u32 k = 8; k < 12; ++k) s[k] = W(k);
#pragma unroll
      for (u32 k = 0; k < 12; ++k) s[k] = gl::add_c(s[k], rc[k]);
      full_step(s, rc + 12, 29, 5);
      load_row(s, 29);
      full_step(s, rc + 24, 41, 17);
      load_row(s, 41);
      full_step(s, rc + 36, 53, 29);
      load_row(s, 87);
      full_step(s, rc + 12 * 27, 99, 75);
      close_gate((u32)gp);
    }
    pp_check(18);
    pp_check(19);
  } else if (role == 2) {
    if (gp >= 0) {
      u64 s[12];
      load_row(s, 99);
      full_step(s, rc + 12 * 28, 111, 87);
      load_row(s, 111);
      full_step(s, rc + 12 * 29, 123, 99);
      load_row(s, 123);
      full_step(s, nullptr, 12, 111);
      close_gate((u32)gp);
    }
    if (g_arith >= 0) {
      const u64 c0 = CS(d.num_selectors), c1 = CS(d.num_selectors + 1);
#pragma unroll 1
      for (u32 o = 0; o < kNumOps; ++o) {
        const u64 prod = gl::mul(gl::mul(W(4 * o), W(4 * o + 1)), c0);
        G(o, fsub(W(4 * o + 3), gl::mul_add(W(4 * o + 2), c1, prod)));
      }
      close_gate((u32)g_arith);
    }
    if (g_const >= 0) {
      for (u32 j = 0; j < kNumConsts; ++j) G(j, gl::sub_c(CS(d.num_selectors + j), W(j)));
      close_gate((u32)g_const);
    }
    if (g_pi >= 0) {
      for (u32 j = 0; j < 4; ++j) G(j, gl::sub_c(W(j), pi_hash[j]));
      close_gate((u32)g_pi);
    }
    {  // L_0(x) (Z_c(x) - 1), L_0(x) = (x^n - 1) / (n (x - 1))
      const u64 x = gl::mul(7, gl::pow(d.w_big, i));
      const u64 l0 = gl::mul(gl::mul(d.zh[i & (kQF - 1)], d.n_inv), gl_inv(gl::canon(gl::sub_c(x, 1))));
      for (u32 c = 0; c < kNumCh; ++c) T(c, gl::mul(l0, gl::sub_c(Z(c), 1)));
    }
#pragma unroll 1
    for (u32 idx = 14; idx < 18; ++idx) pp_check(idx);
  } else {
#pragma unroll 1
    for (u32 idx = 0; idx < 14; ++idx) pp_check(idx);
  }
#pragma unroll
  for (u32 c = 0; c < kNumCh; ++c) part[role][lane][c] = acc[c];
  __syncthreads();
  if (role == 0 && valid) {
    const u64 zh_inv = d.zh_inv[i & (kQF - 1)];
#pragma unroll
    for (u32 c = 0; c < kNumCh; ++c) {
      u64 sum = gl::add(gl::add(part[0][lane][c], part[1][lane][c]), gl::add(part[2][lane][c], part[3][lane][c]));
      for (u32 e = 0; e < n_extra; ++e)  // the recursion gates' filtered constraints, one plane per gate type (k_quotient_extra)
        sum = gl::add(sum, extra[((size_t)e * kNumCh + c) * big + i]);
      qvals[(size_t)c * big + i] = gl::canon(gl::mul(sum, zh_inv));
    }
  }
}

// The gate types of the in-circuit verifier (gates_recursion.hip.h) on the LDE coset: one lane per point evaluates every such gate
// type present in the circuit, weights constraint j with alpha_c^(22 + j) (the gate constraints follow the 2 + 20 permutation terms),
// multiplies by the gate's selector filter and leaves the per-challenge sums in extra[c][i] (natural order) for k_quotient to add
// before the division by Z_H.  ~2 k field multiplications per point -- two orders of magnitude below the PoseidonGate's share.
struct FBaseDev {
  typedef u64 T;
  GL_DEV static T add(T a, T b) { return gl::add(a, b); }
  GL_DEV static T sub(T a, T b) { return fsub(a, b); }
  GL_DEV static T mul(T a, T b) { return gl::mul(a, b); }
  GL_DEV static T mulc(T a, u64 c) { return gl::mul(a, c); }
  GL_DEV static T addc(T a, u64 c) { return gl::add_c(a, c); }
  GL_DEV static T subc(T a, u64 c) { return gl::sub_c(a, c); }
  GL_DEV static T fromc(u64 c) { return c; }
};
__global__ __launch_bounds__(kBlock) void k_quotient_extra(const QDesc d, const u64* __restrict__ cs, const u64* __restrict__ wl,
                                                           const u64* __restrict__ chal, u64* __restrict__ extra, BatchArg ba) {
  wl = bp(wl, ba);
  chal = bp(chal, ba);
  extra = bp(extra, ba);
  __shared__ u64 apow[kNumCh][kNumGateConstraints];
  const u32 log_big = d.log_n + 3, big = 1u << log_big;
  for (u32 k = threadIdx.x; k < kNumCh * kNumGateConstraints; k += kBlock) {
    const u32 c = k / kNumGateConstraints, e = k % kNumGateConstraints;
    apow[c][e] = gl::canon(gl::pow(chal[2 * kNumCh + c], kNumCh + kNumCh * kNumChunks + e));
  }
  __syncthreads();
  const u32 r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= big) return;
  const u32 i = __brev(r) >> (32 - log_big);
  // blockIdx.y = which of the circuit's recursion gate types this workgroup evaluates (a lane per point evaluating all eight in
  // sequence was a 220 us latency chain on half a wavefront per SIMD; one type per workgroup row runs them side by side)
  u32 g = 0, seen = 0;
  for (; g < d.n_kinds; ++g)
    if (d.kind[g] > G_POSEIDON && seen++ == blockIdx.y) break;
  auto CS = [&](u32 j) { return cs[(size_t)j * big + r]; };
  auto W = [&](int j) { return wl[(size_t)j * big + r]; };
  u64 gacc[kNumCh];
#pragma unroll
  for (u32 c = 0; c < kNumCh; ++c) gacc[c] = 0;
  auto G = [&](int j, u64 t) {
#pragma unroll
    for (u32 c = 0; c < kNumCh; ++c) gacc[c] = gl::mul_add(apow[c][j], t, gacc[c]);
  };
  const u64 c0 = CS(d.num_selectors), c1 = CS(d.num_selectors + 1);
  switch (d.kind[g]) {  // block-uniform
    case G_BASE_SUM: gates_rec::base_sum_gate<FBaseDev>(W, G); break;
    case G_ARITHMETIC_EXT: gates_rec::arithmetic_ext_gate<FBaseDev>(W, c0, c1, G); break;
    case G_MUL_EXT: gates_rec::mul_ext_gate<FBaseDev>(W, c0, G); break;
    case G_REDUCING: gates_rec::reducing_gate<FBaseDev>(W, G); break;
    case G_REDUCING_EXT: gates_rec::reducing_ext_gate<FBaseDev>(W, G); break;
    case G_RANDOM_ACCESS: gates_rec::random_access_gate<FBaseDev>(W, c0, c1, G); break;
    case G_COSET_INTERPOLATION: gates_rec::coset_interpolation_gate<FBaseDev>(W, G); break;
    case G_POSEIDON_MDS: gates_rec::poseidon_mds_gate<FBaseDev>(W, G); break;
    default: break;
  }
  const u64 sv = CS(d.sel[g]);  // compute_filter
  u64 f = 1;
  for (u32 k = d.gs[g]; k < d.ge[g]; ++k)
    if (k != g) f = gl::mul(f, fsub((u64)k, sv));
  if (d.num_selectors > 1) f = gl::mul(f, fsub(kUnusedSelector, sv));
#pragma unroll
  for (u32 c = 0; c < kNumCh; ++c) extra[((size_t)blockIdx.y * kNumCh + c) * big + i] = gl::canon(gl::mul(f, gacc[c]));
}

// OpeningSet order (constants | sigmas | wires | zs | zs_next | partial products | quotient) from the FriOpenings order the
// challenger observed (... | zs | partial products | quotient || zs_next)
// Also plonky2's `ensure!(zeta^n != 1)`: err[3] is set when the opening point lies in the subgroup.
__global__ __launch_bounds__(kBlock) void k_opening_set(const u64* __restrict__ fri_order, u64* __restrict__ out, u32 n_cs,
                                                        const u64* __restrict__ zeta, u32 log_n, int* __restrict__ err, BatchArg ba) {
  fri_order = bp(fri_order, ba);
  out = bp(out, ba);
  zeta = bp(zeta, ba);
  err = bp(err, ba);
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  if (t == 0) {
    u64 a = zeta[0], b = zeta[1];
    for (u32 k = 0; k < log_n; ++k) {
      const u64 na = gl::mul_add(a, a, gl::mul(7, gl::mul(b, b))), nb = gl::mul(2, gl::mul(a, b));
      a = na;
      b = nb;
    }
    err[3] = (gl::canon(a) == 1 && gl::canon(b) == 0) ? 1 : 0;
  }
  const u32 a = 2 * (n_cs + kNumWires + kNumCh), tail = 2 * (kNumCh * kNumProds + kNumQuot), total = a + 2 * kNumCh + tail;
  if (t >= total) return;
  u32 src;
  if (t < a) src = t;
  else if (t < a + 2 * kNumCh) src = a + tail + (t - a);
  else src = t - 2 * kNumCh;
  out[t] = fri_order[src];
}

// inverse of k_opening_set: the order the challenger observes (FriOpenings) from the proof's OpeningSet order
__global__ __launch_bounds__(kBlock) void k_opening_unset(const u64* __restrict__ set_order, u64* __restrict__ out, u32 n_cs, BatchArg ba) {
  set_order = bp(set_order, ba);
  out = bp(out, ba);
  const u32 t = blockIdx.x * kBlock + threadIdx.x;
  const u32 a = 2 * (n_cs + kNumWires + kNumCh), tail = 2 * (kNumCh * kNumProds + kNumQuot), total = a + 2 * kNumCh + tail;
  if (t >= total) return;
  u32 dst;
  if (t < a) dst = t;
  else if (t < a + 2 * kNumCh) dst = a + tail + (t - a);
  else dst = t - 2 * kNumCh;
  out[dst] = set_order[t];
}

// one (query, tree) pair of a proof under verification: where its opened row, its path and its cap sit in the proof block.  (Reserved:
// since round 4 every wavefront of k_verify_paths works its own triple out; the layout below still keeps the space, nothing writes it.)
struct VItem {
  u32 leaf_off, width, index, sib_off, n_sib, cap_off;
};

size_t digests_count(size_t n, unsigned cap_height) {
  size_t c = 0;
  for (unsigned j = 0; ((size_t)n >> j) > ((size_t)1 << cap_height); ++j) c += n >> j;
  return c;
}

inline u32 target_index(const p2mt_circuit_data* c, u64 t) {
  return is_wire(t) ? wire_row(t) * kNumWires + wire_col(t) : (u32)(c->n * kNumWires + t);
}

int valid_target(const p2mt_circuit_data* c, u64 t) {
  if (is_wire(t)) return wire_row(t) < c->n && wire_col(t) < kNumWires && !((t & ~kWireFlag) >> 40);
  return t < c->n_virtual;
}

// Targets a generator watches (ins) and sets (outs), as wire / virtual targets.  These are the dependencies() of plonky2's
// generators; build() gives every one of them a value slot.
void gen_targets(const Gen& g, std::vector<u64>& ins, std::vector<u64>& outs) {
  ins.clear();
  outs.clear();
  auto W = [&](u32 col) { return wire_t(g.row, col); };
  switch (g.kind) {
    case GEN_POSEIDON:
      for (u32 k = 0; k < 12; ++k) ins.push_back(W(k));
      ins.push_back(W(24));
      for (u32 k = 12; k < 24; ++k) outs.push_back(W(k));
      break;
    case GEN_ARITH:
      for (u32 k = 0; k < 3; ++k) ins.push_back(W(4 * g.i + k));
      outs.push_back(W(4 * g.i + 3));
      break;
    case GEN_EQUALITY:
      ins = {g.x, g.y};
      outs = {g.eq, g.inv};
      break;
    case GEN_CONST: outs.push_back(W(g.i)); break;
    case GEN_ARITH_EXT:
      for (u32 k = 0; k < 6; ++k) ins.push_back(W(8 * g.i + k));
      outs = {W(8 * g.i + 6), W(8 * g.i + 7)};
      break;
    case GEN_MUL_EXT:
      for (u32 k = 0; k < 4; ++k) ins.push_back(W(6 * g.i + k));
      outs = {W(6 * g.i + 4), W(6 * g.i + 5)};
      break;
    case GEN_QUOTIENT_EXT:
      ins.assign(g.t.begin(), g.t.begin() + 4);
      outs.assign(g.t.begin() + 4, g.t.begin() + 6);
      break;
    // (The two reducing generators list their OUTPUT wires among the inputs: the pre-pass's COMBINE is the first writer of those,
    // as early as old_acc allows, and the gate's own generator -- which fills the n - 1 intermediate accumulators -- re-derives
    // them at its end and compares.  Without this the scheduler, which orders by dependency depth, makes the one-level Horner
    // chain the first writer and the two-level pre-pass the check.)
    case GEN_REDUCING:
      ins = {W(0), W(1)};
      for (u32 k = 2; k < 6 + kReducingCoeffs; ++k) ins.push_back(W(k));
      outs = {W(0), W(1)};
      for (u32 k = 0; k + 1 < kReducingCoeffs; ++k) outs.push_back(W(6 + kReducingCoeffs + 2 * k)), outs.push_back(W(7 + kReducingCoeffs + 2 * k));
      break;
    case GEN_REDUCING_EXT:
      ins = {W(0), W(1)};
      for (u32 k = 2; k < 6 + 2 * kReducingExtCoeffs; ++k) ins.push_back(W(k));
      outs = {W(0), W(1)};
      for (u32 k = 0; k + 1 < kReducingExtCoeffs; ++k)
        outs.push_back(W(6 + 2 * kReducingExtCoeffs + 2 * k)), outs.push_back(W(7 + 2 * kReducingExtCoeffs + 2 * k));
      break;
    case GEN_REDUCING_LOCAL:
      ins = {W(2), W(3)};
      for (u32 k = 0; k < kReducingCoeffs; ++k) ins.push_back(W(6 + k));
      outs = g.t;
      break;
    case GEN_REDUCING_EXT_LOCAL:
      ins = {W(2), W(3)};
      for (u32 k = 0; k < 2 * kReducingExtCoeffs; ++k) ins.push_back(W(6 + k));
      outs = g.t;
      break;
    case GEN_REDUCING_COMBINE:
      ins = {W(4), W(5), g.t[0], g.t[1], g.t[2], g.t[3]};
      outs = {W(0), W(1)};
      break;
    case GEN_WIRE_SPLIT:
      ins.push_back(g.t[0]);
      outs.assign(g.t.begin() + 1, g.t.end());
      break;
    case GEN_BASE_SPLIT:
      ins.push_back(W(0));
      for (u32 k = 0; k < kBaseSumLimbs; ++k) outs.push_back(W(1 + k));
      break;
    case GEN_RANDOM_ACCESS:
      ins.push_back(W(18 * g.i));
      for (u32 k = 0; k < 16; ++k) ins.push_back(W(18 * g.i + 2 + k));
      outs.push_back(W(18 * g.i + 1));
      for (u32 k = 0; k < kRaBits; ++k) outs.push_back(W(74 + kRaBits * g.i + k));
      break;
    case GEN_INTERPOLATION:
      for (u32 k = 0; k < 35; ++k) ins.push_back(W(k));
      for (u32 k = 35; k < 47; ++k) outs.push_back(W(k));
      break;
    case GEN_POSEIDON_MDS:  // (one record per output element g.i)
      for (u32 k = 0; k < 24; ++k) ins.push_back(W(k));
      outs = {W(24 + 2 * g.i), W(25 + 2 * g.i)};
      break;
    default: break;
  }
}

// single verifications / proves keep their transcript on a host core (host_poseidon.h); env P2MT_HOST_TRANSCRIPT=0 or
// p2mt_debug_host_transcript(0) puts it back on the device (A/B, and what the batched passes always do)
int& host_transcript_flag() {
  static int v = [] {
    const char* e = getenv("P2MT_HOST_TRANSCRIPT");
    return e ? atoi(e) : 1;
  }();
  return v;
}
bool host_transcript_on() { return host_transcript_flag() != 0; }
// ... and with it their long PoseidonGate chains (select_host_chain below); env P2MT_HOST_CHAIN=0 / p2mt_debug_host_chain(0): A/B
int& host_chain_flag() {
  static int v = [] {
    const char* e = getenv("P2MT_HOST_CHAIN");
    return e ? atoi(e) : 1;
  }();
  return v;
}

// Which PoseidonGate rows the HOST evaluates before the launch (single proves with the transcript on the host, csrc/host_poseidon.hip).
// A dependency chain of L PoseidonGate rows costs L x ~10 us on a wavefront and L x 1.35 us on a host core, and the recursion's outer
// circuit has one of 105-110 rows -- the inner proof's transcript: it was the critical path of the witness (tools/critical_path.py).
// Every input of such a row is a witness input, a constant or an output of an earlier row of the chain, i.e. something the host holds
// or can derive before anything is launched; its outputs then go down WITH the witness inputs, the row leaves the device schedule, and
// what depended on the challenges starts at level 1.  Structural (values do not matter), decided with the schedule:
//   evaluable   all 13 inputs (state + swap) are constants, witness inputs or outputs of evaluable rows;
//   length      rows on the longest evaluable chain through the row (depth from the inputs + height to the last consumer - 1);
//   chosen      every row of length >= T and its ancestors, for the smallest T >= kHostChainMinLen whose set stays within kHostChainCap
//               rows (the host is one core: 28 x 4 leaf sponges of 17 rows are evaluable too, and are better left to 112 wavefronts).
// k_poseidon_rows still fills the non-routed wires of every row from the wire matrix, chosen or not.
constexpr u32 kHostChainMinLen = 8, kHostChainCap = 192, kHostChainArithCap = 2048;
void select_host_chain(p2mt_circuit_data* c, const std::vector<u32>& io, const std::vector<u32>& io_off, const std::vector<u32>& n_ins,
                       const std::vector<int>& set_level, std::vector<char>& on_host) {
  const size_t n_gens = c->gens.size();
  auto is_p = [&](u32 gi) { return c->gens[gi].kind == GEN_POSEIDON; };
  auto n_outs = [&](u32 gi) { return io_off[gi + 1] - io_off[gi] - n_ins[gi]; };
  std::vector<char> known(c->n_slots, 0);
  for (u32 s = 0; s < c->n_slots; ++s) known[s] = set_level[s] == 0;
  // depth / height count the PoseidonGate rows on the longest evaluable path up to / from a generator (itself included)
  std::vector<int> producer(c->n_slots, -1), depth(n_gens, 0), height(n_gens, 0);
  std::vector<u32> missing(n_gens, 0), order;
  std::vector<std::vector<u32>> watchers(c->n_slots);
  for (size_t gi = 0; gi < n_gens; ++gi) {
    // PoseidonGenerator, and ArithmeticBaseGenerator: the selects between two hashes (pick_hash of the reference's inner circuit,
    // common.rs:42-58) sit between the rows of its chain
    if (c->gens[gi].kind != GEN_POSEIDON && c->gens[gi].kind != GEN_ARITH) continue;
    const u32* in = io.data() + io_off[gi];
    bool usable = true;
    for (u32 k = 0; k < io_off[gi + 1] - io_off[gi]; ++k) usable &= in[k] != kNoSlot;
    if (!usable) continue;
    for (u32 k = 0; k < n_ins[gi]; ++k) {
      bool dup = false;
      for (u32 j = 0; j < k; ++j) dup |= in[j] == in[k];
      if (!dup && !known[in[k]]) {
        ++missing[gi];
        watchers[in[k]].push_back((u32)gi);
      }
    }
    if (!missing[gi]) order.push_back((u32)gi);
  }
  for (size_t head = 0; head < order.size(); ++head) {
    const u32 gi = order[head];
    const u32 *in = io.data() + io_off[gi], *out = in + n_ins[gi];
    int d = 0;
    for (u32 k = 0; k < n_ins[gi]; ++k)
      if (producer[in[k]] >= 0) d = std::max(d, depth[producer[in[k]]]);
    depth[gi] = d + (is_p(gi) ? 1 : 0);
    for (u32 k = 0; k < n_outs(gi); ++k) {
      if (known[out[k]]) continue;  // (not its first writer)
      known[out[k]] = 1;
      producer[out[k]] = (int)gi;
      for (u32 w : watchers[out[k]])
        if (--missing[w] == 0) order.push_back(w);
    }
  }
  int longest = 0;
  for (size_t k = order.size(); k-- > 0;) {  // consumers come later in `order`: their heights are final when a producer is visited
    const u32 gi = order[k];
    height[gi] = std::max(height[gi], is_p(gi) ? 1 : 0);
    const u32* in = io.data() + io_off[gi];
    for (u32 j = 0; j < n_ins[gi]; ++j) {
      const int pr = producer[in[j]];
      if (pr >= 0) height[pr] = std::max(height[pr], height[gi] + (is_p((u32)pr) ? 1 : 0));
    }
  }
  auto length = [&](u32 gi) { return depth[gi] + height[gi] - (is_p(gi) ? 1 : 0); };
  for (u32 gi : order) longest = std::max(longest, length(gi));
  std::vector<char> best;
  for (int T = longest; T >= (int)kHostChainMinLen; --T) {
    std::vector<char> pick(n_gens, 0);
    size_t n_rows = 0, n_arith = 0;
    for (size_t k = order.size(); k-- > 0;) {  // reverse order: a picked generator picks its producers
      const u32 gi = order[k];
      if (!pick[gi] && is_p(gi) && length(gi) >= T) pick[gi] = 1;
      if (!pick[gi]) continue;
      (is_p(gi) ? n_rows : n_arith) += 1;
      const u32* in = io.data() + io_off[gi];
      for (u32 j = 0; j < n_ins[gi]; ++j)
        if (producer[in[j]] >= 0) pick[producer[in[j]]] = 1;
    }
    if (n_rows > kHostChainCap || n_arith > kHostChainArithCap) break;
    best.swap(pick);
  }
  if (best.empty()) return;
  // an output is a first write on the host unless the host holds the slot already (an input, a constant, an earlier chosen generator):
  // then it is compared.  (A slot whose first writer stays on the device becomes a check THERE: it is set before the launch.)
  std::vector<char> held(c->n_slots, 0);
  for (u32 s = 0; s < c->n_slots; ++s) held[s] = set_level[s] == 0;
  for (u32 gi : order) {
    if (!best[gi]) continue;
    const u32 *in = io.data() + io_off[gi], *out = in + n_ins[gi];
    p2mt_circuit_data::HostOp r{};
    r.poseidon = is_p(gi);
    r.n_in = n_ins[gi];
    r.n_out = n_outs(gi);
    r.c0 = c->gens[gi].c0;
    r.c1 = c->gens[gi].c1;
    for (u32 k = 0; k < r.n_in; ++k) r.in[k] = in[k];
    for (u32 k = 0; k < r.n_out; ++k) {
      r.out[k] = out[k];
      if (held[out[k]]) r.check_mask |= 1u << k;
      held[out[k]] = 1;
    }
    c->host_chain.push_back(r);
    c->host_rows += r.poseidon;
    on_host[gi] = 1;
  }
  c->h_vals.assign(c->n_slots, 0);
  for (const auto& ci : c->const_inits) c->h_vals[ci.first] = ci.second;
}

// Order the generators into levels for the given set of externally set slots (generate_partial_witness's watch lists,
// resolved ahead of time: readiness does not depend on values).  Returns P2MT_EINVAL if some generator can never run.
int schedule(p2mt_circuit_data* c, const std::vector<u32>& input_slots, bool host_chain) {
  if (c->sched_valid && c->sched_inputs == input_slots && c->sched_host_chain == host_chain) return P2MT_OK;
  c->sched_valid = false;
  c->host_chain.clear();
  c->host_rows = 0;
  std::vector<int> set_level(c->n_slots, -1);
  for (const auto& ci : c->const_inits) set_level[ci.first] = 0;
  for (u32 s : input_slots) set_level[s] = 0;
  auto slot_w = [&](u32 row, u32 col) { return c->slot_of[row * kNumWires + col]; };
  // slots of every generator's inputs / outputs, once
  const size_t n_gens = c->gens.size();
  std::vector<u32> io;            // concatenated slot lists
  std::vector<u32> io_off(n_gens + 1, 0), n_ins(n_gens, 0);
  {
    std::vector<u64> ins, outs;
    for (size_t gi = 0; gi < n_gens; ++gi) {
      gen_targets(c->gens[gi], ins, outs);
      n_ins[gi] = (u32)ins.size();
      for (u64 t : ins) io.push_back(c->slot_of[target_index(c, t)]);
      for (u64 t : outs) io.push_back(c->slot_of[target_index(c, t)]);
      io_off[gi + 1] = (u32)io.size();
    }
  }
  std::vector<char> on_host(n_gens, 0);
  size_t n_on_host = 0;
  if (host_chain) {
    select_host_chain(c, io, io_off, n_ins, set_level, on_host);
    n_on_host = c->host_chain.size();
    for (const auto& r : c->host_chain)
      for (u32 k = 0; k < r.n_out; ++k) set_level[r.out[k]] = 0;  // known before the launch, like a witness input
  }
  // worklist: a generator becomes ready when its last unset input slot gets a level
  std::vector<u32> missing(n_gens, 0);
  std::vector<std::vector<u32>> watchers(c->n_slots);
  std::vector<u32> ready;
  for (size_t gi = 0; gi < n_gens; ++gi) {
    if (on_host[gi]) continue;
    const u32* in = io.data() + io_off[gi];
    for (u32 k = 0; k < n_ins[gi]; ++k) {
      if (in[k] == kNoSlot) return p2mt::fail(P2MT_EINVAL, "internal: a generator input has no value slot");
      bool dup = false;
      for (u32 j = 0; j < k; ++j) dup |= in[j] == in[k];
      if (!dup && set_level[in[k]] < 0) {
        ++missing[gi];
        watchers[in[k]].push_back((u32)gi);
      }
    }
    if (!missing[gi]) ready.push_back((u32)gi);
  }
  struct Item {
    int level;
    WOp op;
  };
  std::vector<Item> items;
  items.reserve(n_gens);
  std::vector<u32> pslots;  // per PoseidonGate row: the value slots of wires 0..24 (inputs, outputs, swap), padded to 32
  std::vector<u32> args;    // slot lists of the generators that are not tied to a gate row
  u32 n_poseidon = 0;
  size_t head = 0;
  while (head < ready.size()) {
    const u32 gi = ready[head++];
    const Gen& g = c->gens[gi];
    const u32 *in = io.data() + io_off[gi], *out = in + n_ins[gi];
    const u32 n_out = io_off[gi + 1] - io_off[gi] - n_ins[gi];
    int level = 0;
    for (u32 k = 0; k < n_ins[gi]; ++k) level = std::max(level, set_level[in[k]]);
    for (u32 k = 0; k < n_out; ++k) {
      if (out[k] == kNoSlot) return p2mt::fail(P2MT_EINVAL, "internal: a generator output has no value slot");
      level = std::max(level, set_level[out[k]]);  // a check waits for the value it checks
    }
    ++level;
    // which outputs this generator is the first writer of (a slot listed twice counts once; any other output is a check)
    u32 fmask[3] = {0, 0, 0};
    bool fresh = true;
    for (u32 k = 0; k < n_out; ++k) {
      bool f = set_level[out[k]] < 0;
      for (u32 j = 0; j < k; ++j) f &= out[j] != out[k];
      if (f && k < 96) fmask[k >> 5] |= 1u << (k & 31);
      fresh &= f;
    }
    if (n_out > 96) return p2mt::fail(P2MT_EINVAL, "internal: generator with more than 96 outputs");
    for (u32 k = 0; k < n_out; ++k)
      if (set_level[out[k]] < 0) {
        set_level[out[k]] = level;
        for (u32 w : watchers[out[k]])
          if (--missing[w] == 0) ready.push_back(w);
        watchers[out[k]].clear();
      }
    WOp op{};
    op.kind = (u32)g.kind | (fresh ? kFreshOutputs : 0);
    op.c0 = g.c0;
    op.c1 = g.c1;
    if (g.kind != GEN_ARITH && g.kind != GEN_EQUALITY) op.c = fmask[0], op.out = fmask[1], op.out2 = fmask[2];
    else op.kind |= (fmask[0] & 3) << 9;
    switch (g.kind) {
      case GEN_ARITH: op.a = in[0], op.b = in[1], op.c = in[2], op.out = out[0]; break;
      case GEN_EQUALITY: op.a = in[0], op.b = in[1], op.out = out[0], op.out2 = out[1]; break;
      case GEN_POSEIDON:
        op.a = g.row;
        op.b = n_poseidon++;
        pslots.resize((size_t)n_poseidon * 32, 0);
        for (u32 k = 0; k < 25; ++k) pslots[(size_t)op.b * 32 + k] = slot_w(g.row, k);
        break;
      case GEN_QUOTIENT_EXT:
      case GEN_WIRE_SPLIT:
      case GEN_REDUCING_LOCAL:
      case GEN_REDUCING_EXT_LOCAL:
      case GEN_REDUCING_COMBINE:
        op.a = (u32)args.size();
        op.b = io_off[gi + 1] - io_off[gi];
        args.insert(args.end(), in, in + op.b);
        break;
      default:  // tied to a gate row: slots through the dense table
        op.a = g.row;
        op.b = g.i;
        break;
    }
    items.push_back(Item{level, op});
  }
  if (items.size() + n_on_host != n_gens) return p2mt::fail(P2MT_EINVAL, "prove: some generators weren't run (a target they depend on was never set)");
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) {
    return a.level != b.level ? a.level < b.level : ((a.op.kind & 0xFF) == GEN_POSEIDON) > ((b.op.kind & 0xFF) == GEN_POSEIDON);
  });
  // Emit level by level: the PoseidonGate rows first (one per wavefront), then the lane generators.  In the dataflow interpreter a
  // wavefront executes the code paths of ALL the generator kinds its 64 lanes hold, one after the other; the long single-lane
  // generators (a ReducingExtensionGate's 32-step Horner chain, a CosetInterpolationGate, an extension inversion: 25-35 us each)
  // would add up inside one wavefront.  So each of those kinds starts at a multiple of 64 lanes (padding with no-op records):
  // different kinds of a level run on different wavefronts, side by side.  (Tables in LDS are walked by one workgroup: no padding.)
  auto weight_class = [](u32 kind) -> int {
    switch (kind & 0xFF) {
      case GEN_REDUCING: return 1;
      case GEN_REDUCING_EXT: return 2;
      case GEN_INTERPOLATION: return 3;
      case GEN_QUOTIENT_EXT: return 4;
      case GEN_POSEIDON_MDS: return 5;
      case GEN_BASE_SPLIT: return 6;
      case GEN_WIRE_SPLIT: return 7;
      case GEN_RANDOM_ACCESS: return 8;
      case GEN_REDUCING_LOCAL: return 9;
      case GEN_REDUCING_EXT_LOCAL: return 10;
      default: return 0;
    }
  };
  std::vector<WOp> ops;
  std::vector<u32> lvl;
  for (int pass = 0; pass < 2; ++pass) {
    const bool pad = pass == 0 && !c->lds_bytes && c->has_recursion_gates;
    ops.clear();
    lvl.clear();
    size_t k = 0;
    while (k < items.size()) {
      size_t k2 = k;
      u32 np = 0;
      while (k2 < items.size() && items[k2].level == items[k].level) {
        np += (items[k2].op.kind & 0xFF) == GEN_POSEIDON;
        ++k2;
      }
      lvl.push_back((u32)ops.size());
      lvl.push_back(np);
      for (size_t j = k; j < k + np; ++j) ops.push_back(items[j].op);
      const size_t base = ops.size();
      for (int cls = 0; cls <= 10; ++cls) {
        bool first = true;
        for (size_t j = k + np; j < k2; ++j) {
          if (weight_class(items[j].op.kind) != cls) continue;
          if (first && pad && ops.size() != base) {
            WOp nop{};
            nop.kind = kGenNop;
            while ((ops.size() - base) % 64) ops.push_back(nop);
          }
          first = false;
          ops.push_back(items[j].op);
        }
      }
      k = k2;
    }
    lvl.push_back((u32)ops.size());
    if (ops.size() <= c->ops_cap) break;  // (the padded table did not fit: emit it dense)
  }
  if (ops.size() > c->ops_cap) return p2mt::fail(P2MT_EINVAL, "internal: generator table overflow");
  c->n_levels = (u32)(lvl.size() / 2);
  if (args.size() > c->args_cap) return p2mt::fail(P2MT_EINVAL, "internal: generator argument table overflow");
  hipStream_t st = rt().stream;
  P2MT_HIP(hipStreamSynchronize(st));  // the previous schedule may still be in use
  if (!ops.empty()) P2MT_HIP(hipMemcpy(c->d_ops, ops.data(), ops.size() * sizeof(WOp), hipMemcpyHostToDevice));
  P2MT_HIP(hipMemcpy(c->d_lvl, lvl.data(), lvl.size() * 4, hipMemcpyHostToDevice));
  if (!pslots.empty()) P2MT_HIP(hipMemcpy(c->d_pslots, pslots.data(), pslots.size() * 4, hipMemcpyHostToDevice));
  if (!args.empty()) P2MT_HIP(hipMemcpy(c->d_args, args.data(), args.size() * 4, hipMemcpyHostToDevice));
  c->sched_inputs = input_slots;
  c->sched_host_chain = host_chain;
  c->sched_valid = true;
  return P2MT_OK;
}

int fill_poseidon_rows(p2mt_circuit_data* c) {
  const u32 n_rows = c->counts[G_POSEIDON];
  if (!n_rows) return P2MT_OK;
  hipLaunchKernelGGL(k_poseidon_rows, bgrid((n_rows + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, rt().stream,
                     (const u32*)c->d_prows, n_rows, c->d_w_vals, c->degree_bits, barg(), p2mt::perm_ctx());
  P2MT_LAUNCH_CHECK();
  return P2MT_OK;
}

// witness fill for one PartialWitness: enqueue init / run / scatter (no synchronisation); wires -> d_w_vals, public inputs
// -> the tail of the proof buffer
int fill_witness(p2mt_circuit_data* c, const p2mt_partial_witness* const* pws, unsigned B) {
  // Provers call this with the same TARGET sequence proof after proof (only the values change): the target -> slot resolution,
  // the duplicate detection and the schedule lookup are memoised on that sequence, so the per-proof host work is one linear pass
  // (the outer recursion circuit sets ~11 300 targets per proof: 1.07 ms with a hash map per proof, ~0.03 ms this way).
  // The proofs of a batch must all set that one sequence.
  const p2mt_partial_witness* pw = pws[0];
  const size_t n_sets = pw->sets.size();
  // single proves with the transcript on the host also take their long PoseidonGate chains there (select_host_chain); the proofs of a
  // batch keep everything on the device.  The schedule differs (the chosen rows are not in it): switching re-schedules.
  const bool want_host_chain = B == 1 && host_chain_flag() != 0 && host_transcript_on();
  bool same = c->memo_valid && c->memo_targets.size() == n_sets;
  for (size_t k = 0; same && k < n_sets; ++k) same = c->memo_targets[k] == pw->sets[k].first;
  if (!same) {
    c->memo_valid = false;
    c->memo_targets.resize(n_sets);
    c->memo_slot.resize(n_sets);
    c->memo_first.assign(n_sets, -2);  // -2: first assignment of its slot; -1: the slot holds a constant; k >= 0: same slot as set k
    std::unordered_map<u32, int> seen;  // slot -> index of the first set (or -1 for a constant)
    for (const auto& ci : c->const_inits) seen[ci.first] = -1;
    std::vector<u32> input_slots;
    for (size_t k = 0; k < n_sets; ++k) {
      const u64 t = pw->sets[k].first;
      if (!valid_target(c, t)) return p2mt::fail(P2MT_EINVAL, "prove: witness sets a target that is not part of this circuit");
      const u32 slot = c->slot_of[target_index(c, t)];
      if (slot == kNoSlot) return p2mt::fail(P2MT_EINVAL, "prove: witness sets a wire that no generator or copy constraint uses");
      c->memo_targets[k] = t;
      c->memo_slot[k] = slot;
      auto it = seen.find(slot);
      if (it != seen.end()) {
        c->memo_first[k] = it->second;
      } else {
        seen[slot] = (int)k;
        input_slots.push_back(slot);
      }
    }
    std::sort(input_slots.begin(), input_slots.end());
    c->memo_input_slots = input_slots;
    P2MT_TRY(schedule(c, input_slots, want_host_chain));
    c->memo_const_value.clear();
    for (const auto& ci : c->const_inits) c->memo_const_value[ci.first] = ci.second;
    c->memo_valid = true;
  } else if (c->sched_host_chain != want_host_chain) {
    P2MT_TRY(schedule(c, c->memo_input_slots, want_host_chain));
  }
  const bool on_host = want_host_chain && !c->host_chain.empty();
  u64* const hv = on_host ? c->h_vals.data() : nullptr;
  // (slot, value) pairs straight into the pinned staging area: constants first, then every first assignment
  size_t np2 = 0;
  for (unsigned bi = 0; bi < B; ++bi) {
    const p2mt_partial_witness* w = pws[bi];
    if (bi) {
      bool seq = w->sets.size() == n_sets;
      for (size_t k = 0; seq && k < n_sets; ++k) seq = w->sets[k].first == c->memo_targets[k];
      if (!seq) return p2mt::fail(P2MT_EINVAL, "prove_batch: every witness of a batch must set the same targets in the same order");
    }
    u64* pairs = c->h_pin + (size_t)bi * c->pin_pitch + c->pin_pairs_off;
    np2 = 0;
    for (const auto& ci : c->const_inits) {
      pairs[np2++] = ci.first;
      pairs[np2++] = ci.second;
    }
    for (size_t k = 0; k < n_sets; ++k) {
      const int first = c->memo_first[k];
      const u64 v = w->sets[k].second;
      if (first == -2) {
        pairs[np2++] = c->memo_slot[k];
        pairs[np2++] = v;
        if (hv) hv[c->memo_slot[k]] = v;
      } else {  // PartitionWitness::set_target on an already set partition: the values must agree
        const u64 prev = first >= 0 ? w->sets[(size_t)first].second : c->memo_const_value[c->memo_slot[k]];
        if (prev != v) return p2mt::fail(P2MT_EINVAL, "prove: partition was set twice with different values");
      }
    }
  }
  if (on_host) {
    // PoseidonGenerator on the host core, row after row: the state with the swap applied as the gate's own linear form (any swap value),
    // the permutation, the outputs appended to the assignments (or compared where the slot holds a value already)
    u64* pairs = c->h_pin + c->pin_pairs_off;
    for (const auto& r : c->host_chain) {
      u64 x[12];
      if (r.poseidon) {
        for (u32 k = 0; k < 12; ++k) x[k] = hv[r.in[k]];
        const u64 sw = hv[r.in[12]];
        if (sw) {
          for (u32 k = 0; k < 4; ++k) {
            const u64 a = x[k], b = x[k + 4], d = h_mul(sw, h_sub(b, a));
            x[k] = h_add(a, d);
            x[k + 4] = h_sub(b, d);
          }
        }
        host_poseidon::permute(x);
      } else {  // const_0 * m0 * m1 + const_1 * addend
        x[0] = h_add(h_mul(h_mul(hv[r.in[0]], hv[r.in[1]]), r.c0), h_mul(hv[r.in[2]], r.c1));
      }
      for (u32 k = 0; k < r.n_out; ++k) {
        if ((r.check_mask >> k) & 1) {
          if (hv[r.out[k]] != x[k])
            return p2mt::fail(P2MT_EINVAL, "prove: partition was set twice with different values (the witness contradicts the circuit)");
        } else {
          hv[r.out[k]] = x[k];
          pairs[np2++] = r.out[k];
          pairs[np2++] = x[k];
        }
      }
    }
  }
  const size_t n_pairs = np2 / 2;
  hipStream_t st = rt().stream;  // (pinned source: a DMA copy, no blit kernel on the queue)
  if (B > 1) {
    P2MT_HIP(hipMemcpy2DAsync(c->d_init, p2mt::batch().arg.stride, c->h_pin + c->pin_pairs_off, c->pin_pitch * 8, np2 * 8, B,
                              hipMemcpyHostToDevice, st));
  } else {
    P2MT_HIP(hipMemcpyAsync(c->d_init, c->h_pin + c->pin_pairs_off, np2 * 8, hipMemcpyHostToDevice, st));
  }
  u64* d_pi_out = c->d_head + 8 + (c->proof_len - c->n_pi);
  if (!c->lds_bytes) {
    P2MT_TRY(p2mt::batch_fill(c->d_err, 0, 4 * sizeof(int)));
    P2MT_TRY(p2mt::batch_fill(c->d_w_vals, 0, (size_t)kNumWires * c->n * 8));  // wires nothing sets are zero
  }
  if (c->lds_bytes) {
    hipLaunchKernelGGL(k_witness_lds, bgrid(1), dim3(kBlock), c->lds_bytes, st, (const u64*)c->d_init, (u32)n_pairs, c->n_slots,
                       (const WOp*)c->d_ops, (const u32*)c->d_lvl, c->n_levels, (const u32*)c->d_pslots, (const u32*)c->d_wire_slot,
                       c->n_act, c->degree_bits,
                       c->d_w_vals, (const u32*)c->d_pi_slot, c->n_pi, d_pi_out, (const u32*)c->d_slot_tab, (const u32*)c->d_args,
                       c->d_err, barg(), p2mt::perm_ctx());
    P2MT_LAUNCH_CHECK();
    return fill_poseidon_rows(c);
  }
  P2MT_TRY(p2mt::batch_fill(c->d_set, 0, ((size_t)c->n_slots * 4 + 7) & ~(size_t)7));
  P2MT_TRY(p2mt::batch_fill(c->d_vals, 0xFF, (size_t)c->n_slots * 8));  // kUnsetValue everywhere (k_witness_flow waits on it)
  hipLaunchKernelGGL(k_witness_init, bgrid(grid_for(n_pairs)), dim3(kBlock), 0, st, (const u64*)c->d_init, (u32)n_pairs, c->d_vals,
                     c->d_set, barg());
  P2MT_LAUNCH_CHECK();
  // Interpreter for tables in global memory: 2 (default) = dataflow over the whole grid (k_witness_flow), 1 = level-synchronous
  // over the whole grid (k_witness_grid), 0 = one workgroup (k_witness_run); env P2MT_WITNESS_GRID selects (A/B and fallback).
  // The proofs of a batch take the one-workgroup interpreter each: the grid-wide ones assume that their workgroups are all
  // resident, which B grids at once are not.
  static const int env_mode = [] {
    const char* e = getenv("P2MT_WITNESS_GRID");
    return e ? atoi(e) : 2;
  }();
  const int mode = (c->force_single_workgroup || B > 1) ? 0 : env_mode;
  if (mode == 2) {
    hipLaunchKernelGGL(k_witness_flow, dim3(kGridBlocks), dim3(kBlock), 0, st, (const WOp*)c->d_ops, (const u32*)c->d_lvl, c->n_levels,
                       c->d_vals, c->d_set, (const u32*)c->d_pslots, (const u32*)c->d_slot_tab, (const u32*)c->d_args, c->d_err,
                       c->d_trace, p2mt::perm_ctx());
  } else if (mode == 1) {
    P2MT_HIP(hipMemsetAsync(c->d_sync, 0, 8, st));
    hipLaunchKernelGGL(k_witness_grid, dim3(kGridBlocks), dim3(kBlock), 0, st, (const WOp*)c->d_ops, (const u32*)c->d_lvl, c->n_levels,
                       c->d_vals, c->d_set, (const u32*)c->d_pslots, (const u32*)c->d_slot_tab, (const u32*)c->d_args, c->d_err,
                       c->d_sync, p2mt::perm_ctx());
  } else {
    if (B > 1) {
      hipLaunchKernelGGL(k_witness_run<1024>, bgrid(1), dim3(1024), 0, st, (const WOp*)c->d_ops, (const u32*)c->d_lvl, c->n_levels,
                         c->d_vals, c->d_set, (const u32*)c->d_pslots, (const u32*)c->d_slot_tab, (const u32*)c->d_args, c->d_err, barg(),
                         p2mt::perm_ctx());
    } else {
      hipLaunchKernelGGL(k_witness_run<256>, bgrid(1), dim3(kBlock), 0, st, (const WOp*)c->d_ops, (const u32*)c->d_lvl, c->n_levels,
                         c->d_vals, c->d_set, (const u32*)c->d_pslots, (const u32*)c->d_slot_tab, (const u32*)c->d_args, c->d_err, barg(),
                         p2mt::perm_ctx());
    }
  }
  P2MT_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_witness_scatter, bgrid(grid_for(std::max<size_t>(c->n_act, c->n_pi))), dim3(kBlock), 0, st,
                     (const u64*)c->d_vals, (const u32*)c->d_set, (const u32*)c->d_wire_slot, c->n_act, c->d_w_vals,
                     (const u32*)c->d_pi_slot, c->n_pi, d_pi_out, c->d_err, barg());
  P2MT_LAUNCH_CHECK();
  return fill_poseidon_rows(c);
}

int witness_status(p2mt_circuit_data* c, const int* err) {
  (void)c;
  if (err[3] != 0) return p2mt::fail(P2MT_EINVAL, "prove: opening point is in the subgroup");
  if (err[2] != 0) return p2mt::fail(P2MT_EHIP, "prove: the grid-wide witness interpreter gave up waiting (level barrier / operand never written)");
  if (err[0] == -1) return p2mt::fail(P2MT_EINVAL, "prove: a public input target was never set");
  if (err[0] != 0) return p2mt::fail(P2MT_EINVAL, "prove: partition was set twice with different values (the witness contradicts the circuit)");
  if (err[1] != 0) return p2mt::fail(P2MT_EINVAL, "prove: zero denominator in the permutation argument (plonky2 panics on this division)");
  return P2MT_OK;
}

}  // namespace

// ==================================================================================================== C ABI: builder
extern "C" int p2mt_cb_create(p2mt_circuit_builder** out) {
  return p2mt::abi_guard([&]() -> int {
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *out = new (std::nothrow) p2mt_circuit_builder;
  return *out ? P2MT_OK : p2mt::fail(P2MT_ENOMEM, "out of host memory");
  });
}
extern "C" int p2mt_cb_destroy(p2mt_circuit_builder* b) {
  return p2mt::abi_guard([&]() -> int {
  delete b;
  return P2MT_OK;
  });
}
#define CB_ARGS(b, out) \
  if (!(b) || !(out)) return p2mt::fail(P2MT_EINVAL, "null pointer")
extern "C" int p2mt_cb_add_virtual_target(p2mt_circuit_builder* b, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  *out = cb_virtual(b);
  return P2MT_OK;
  });
}
extern "C" int p2mt_cb_constant(p2mt_circuit_builder* b, uint64_t c, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  *out = cb_constant(b, c);
  return P2MT_OK;
  });
}
extern "C" int p2mt_cb_connect(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y) {
  return p2mt::abi_guard([&]() -> int {
  if (!b) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return cb_connect(b, x, y);
  });
}
extern "C" int p2mt_cb_arithmetic(p2mt_circuit_builder* b, uint64_t const_0, uint64_t const_1, p2mt_target multiplicand_0,
                                  p2mt_target multiplicand_1, p2mt_target addend, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, const_0, const_1, multiplicand_0, multiplicand_1, addend, out);
  });
}
extern "C" int p2mt_cb_add(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, 1, 1, x, cb_constant(b, 1), y, out);
  });
}
extern "C" int p2mt_cb_sub(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, 1, gl::P - 1, x, cb_constant(b, 1), y, out);
  });
}
extern "C" int p2mt_cb_mul(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, 1, 0, x, y, x, out);
  });
}
extern "C" int p2mt_cb_mul_add(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target z, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, 1, 1, x, y, z, out);
  });
}
extern "C" int p2mt_cb_mul_sub(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target z, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_arithmetic(b, 1, gl::P - 1, x, y, z, out);
  });
}
extern "C" int p2mt_cb_not(p2mt_circuit_builder* b, p2mt_target x, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  const u64 one = cb_constant(b, 1);
  return cb_arithmetic(b, 1, gl::P - 1, one, one, x, out);
  });
}
extern "C" int p2mt_cb_or(p2mt_circuit_builder* b, p2mt_target b1, p2mt_target b2, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  u64 res_minus_b2;
  P2MT_TRY(cb_arithmetic(b, gl::P - 1, 1, b1, b2, b1, &res_minus_b2));
  return cb_arithmetic(b, 1, 1, res_minus_b2, cb_constant(b, 1), b2, out);
  });
}
extern "C" int p2mt_cb_assert_bool(p2mt_circuit_builder* b, p2mt_target x) {
  return p2mt::abi_guard([&]() -> int {
  if (!b) return p2mt::fail(P2MT_EINVAL, "null pointer");
  u64 z;
  P2MT_TRY(cb_arithmetic(b, 1, gl::P - 1, x, x, x, &z));
  return cb_connect(b, z, cb_constant(b, 0));
  });
}
extern "C" int p2mt_cb_add_virtual_bool_target_safe(p2mt_circuit_builder* b, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  *out = cb_virtual(b);
  return p2mt_cb_assert_bool(b, *out);
  });
}
extern "C" int p2mt_cb_is_equal(p2mt_circuit_builder* b, p2mt_target x, p2mt_target y, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  return cb_is_equal(b, x, y, out);
  });
}
extern "C" int p2mt_cb_hash_n_to_hash_no_pad(p2mt_circuit_builder* b, const p2mt_target* inputs, size_t n, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  if (n && !inputs) return p2mt::fail(P2MT_EINVAL, "null pointer");
  return cb_hash_no_pad(b, inputs, n, out);
  });
}
extern "C" int p2mt_cb_hash_or_noop(p2mt_circuit_builder* b, const p2mt_target* inputs, size_t n, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  if (n && !inputs) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (n <= 4) {
    const u64 zero = cb_constant(b, 0);
    for (size_t k = 0; k < n; ++k) P2MT_TRY(cb_check(b, inputs[k], false));
    for (size_t k = 0; k < 4; ++k) out[k] = k < n ? inputs[k] : zero;
    return P2MT_OK;
  }
  return cb_hash_no_pad(b, inputs, n, out);
  });
}
extern "C" int p2mt_cb_register_public_inputs(p2mt_circuit_builder* b, const p2mt_target* targets, size_t n) {
  return p2mt::abi_guard([&]() -> int {
  if (!b || (n && !targets)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  for (size_t k = 0; k < n; ++k) P2MT_TRY(cb_check(b, targets[k], true));
  b->public_inputs.insert(b->public_inputs.end(), targets, targets + n);
  return P2MT_OK;
  });
}
extern "C" size_t p2mt_cb_num_gates(const p2mt_circuit_builder* b) { return b ? b->gates.size() : 0; }

// ==================================================================================================== build()
extern "C" int p2mt_circuit_destroy(p2mt_circuit_data* c) {
  return p2mt::abi_guard([&]() -> int {
  if (!c) return P2MT_OK;
  if (c->ch) p2mt_challenger_destroy(c->ch);
  if (c->vch) p2mt_challenger_destroy(c->vch);
  if (c->d_verify) (void)hipFree(c->d_verify);
  if (c->vstreams) p2mt::verify_streams_destroy(c->vstreams);
  if (c->h_vpin) (void)hipHostFree(c->h_vpin);
  p2mt::hostlink_destroy(c->link);
  if (c->d_trace) (void)hipFree(c->d_trace);
  if (c->vbch) p2mt::challenger_unwrap(c->vbch);
  if (c->d_vbatch) (void)hipFree(c->d_vbatch);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->d_base) {
    (void)hipStreamSynchronize(rt().stream);
    (void)hipFree(c->d_base);
  }
  delete c;
  return P2MT_OK;
  });
}

extern "C" int p2mt_cb_build(p2mt_circuit_builder* b, p2mt_circuit_data** out) {
  return p2mt::abi_guard([&]() -> int {
  CB_ARGS(b, out);
  P2MT_TRY(p2mt::ensure_init());
  if (b->built) return p2mt::fail(P2MT_EINVAL, "build: this builder was already built (CircuitBuilder::build consumes self)");
  b->built = true;
  // public-input hash routed into a PublicInputGate
  u64 pi_hash_t[4];
  P2MT_TRY(cb_hash_no_pad(b, b->public_inputs.data(), b->public_inputs.size(), pi_hash_t));
  const u32 pi_gate = cb_add_gate(b, G_PUBLIC_INPUT);
  for (u32 k = 0; k < 4; ++k) b->copies.emplace_back(pi_hash_t[k], wire_t(pi_gate, k));
  // one constant generator per distinct constant, in increasing canonical order: the spare constant wires of the gates that have
  // them (RandomAccessGate) first, in row order, then ConstantGates added as needed
  while (b->const_to_target.size() > b->constant_generators.size()) cb_add_gate(b, G_CONSTANT);
  {
    size_t k = 0;
    for (const auto& ct : b->const_to_target) {
      const auto cg = b->constant_generators[k++];
      b->gates[cg[0]].c[cg[1]] = ct.first;
      b->copies.emplace_back(wire_t(cg[0], cg[2]), ct.second);
      Gen g{};
      g.kind = GEN_CONST;
      g.row = cg[0];
      g.i = cg[2];
      g.c0 = ct.first;
      b->gens.push_back(g);
    }
  }
  while (b->gates.size() < 2 || (b->gates.size() & (b->gates.size() - 1))) cb_add_gate(b, G_NOOP);
  const size_t n = b->gates.size();
  if (n > ((size_t)1 << 12)) return p2mt::fail(P2MT_EINVAL, "build: more than 2^12 rows (LDE kernels cover degree_bits <= 12)");

  p2mt_circuit_data* c = new (std::nothrow) p2mt_circuit_data;
  if (!c) return p2mt::fail(P2MT_ENOMEM, "out of host memory");
  struct Guard {
    p2mt_circuit_data* c;
    ~Guard() {
      if (c) p2mt_circuit_destroy(c);
    }
  } guard{c};
  c->n = (u32)n;
  c->degree_bits = (u32)__builtin_ctzll((unsigned long long)n);
  c->n_virtual = b->n_virtual;
  c->public_inputs = b->public_inputs;
  c->n_pi = (u32)b->public_inputs.size();
  for (const auto& g : b->gates) ++c->counts[g.kind];
  // gate types present, sorted by (degree, id); selector groups (gates/selectors.rs selector_polynomials)
  u32 index_of[G_KINDS] = {};
  for (u32 k = 0; k < G_KINDS; ++k) {
    const u32 kind = (u32)kSortedKinds[k];
    if (c->counts[kind]) {
      index_of[kind] = c->n_kinds;
      c->kind[c->n_kinds++] = kind;
      if (kind > G_POSEIDON) c->has_recursion_gates = true;
    }
  }
  const u32 max_degree = kQF + 1;
  u32 group_of[kMaxGateTypes] = {};
  if (kGateDegree[c->kind[c->n_kinds - 1]] + c->n_kinds - 1 <= max_degree) {
    c->num_selectors = 1;
    for (u32 g = 0; g < c->n_kinds; ++g) {
      c->gs[g] = 0;
      c->ge[g] = c->n_kinds;
      group_of[g] = 0;
    }
  } else {
    u32 start = 0;
    while (start < c->n_kinds) {
      u32 size = 0;
      while (start + size < c->n_kinds && size + kGateDegree[c->kind[start + size]] < max_degree) ++size;
      for (u32 g = start; g < start + size; ++g) {
        c->gs[g] = start;
        c->ge[g] = start + size;
        group_of[g] = c->num_selectors;
      }
      ++c->num_selectors;
      start += size;
    }
  }
  for (u32 g = 0; g < c->n_kinds; ++g) c->sel[g] = group_of[g];
  const u32 n_cs = c->n_cs = c->num_selectors + kNumConsts + kNumRouted;
  c->h_cs.assign((size_t)n_cs * n, 0);
  for (size_t row = 0; row < n; ++row) {
    const u32 gi = index_of[b->gates[row].kind];
    for (u32 s = 0; s < c->num_selectors; ++s) c->h_cs[(size_t)s * n + row] = s == group_of[gi] ? gi : kUnusedSelector;
    for (u32 k = 0; k < kNumConsts; ++k) c->h_cs[(size_t)(c->num_selectors + k) * n + row] = b->gates[row].c[k];
  }
  // copy constraints -> classes (Forest) -> value slots and sigma (plonk/permutation_argument.rs)
  const size_t n_targets = n * kNumWires + b->n_virtual;
  std::vector<u32> parent(n_targets);
  for (size_t k = 0; k < n_targets; ++k) parent[k] = (u32)k;
  auto find = [&](u32 x) {
    while (parent[x] != x) {
      parent[x] = parent[parent[x]];
      x = parent[x];
    }
    return x;
  };
  for (const auto& cp : b->copies) {
    const u32 a = find(target_index(c, cp.first)), d = find(target_index(c, cp.second));
    if (a != d) parent[d] = a;
  }
  // classes of targets under the copy constraints (all of them: sigma needs every routed wire's class) ...
  std::vector<u32> cls(n_targets, 0);
  u32 n_cls = 0;
  {
    std::vector<u32> cls_of_rep(n_targets, kNoSlot);
    for (size_t k = 0; k < n_targets; ++k) {
      const u32 r = find((u32)k);
      if (cls_of_rep[r] == kNoSlot) cls_of_rep[r] = n_cls++;
      cls[k] = cls_of_rep[r];
    }
  }
  // ... and a value slot only for the classes something reads or sets: virtual targets, copy-constrained wires, generator
  // inputs / outputs, constants, public inputs.  Unused wires and the PoseidonGate's 110 non-routed wires get none, which
  // keeps the table of a 64-row circuit at ~2 k slots (18 KB) and lets circuits of a few hundred rows stay in LDS.
  {
    std::vector<char> active(n_cls, 0);
    auto mark = [&](u64 t) { active[cls[target_index(c, t)]] = 1; };
    for (u64 v = 0; v < b->n_virtual; ++v) mark(v);
    for (const auto& cp : b->copies) {
      mark(cp.first);
      mark(cp.second);
    }
    {
      std::vector<u64> ins, outs;
      for (const auto& g : b->gens) {
        gen_targets(g, ins, outs);
        for (u64 t : ins) mark(t);
        for (u64 t : outs) mark(t);
      }
    }
    for (u64 t : b->public_inputs) mark(t);
    std::vector<u32> slot_of_cls(n_cls, kNoSlot);
    for (u32 k = 0; k < n_cls; ++k)
      if (active[k]) slot_of_cls[k] = c->n_slots++;
    c->slot_of.resize(n_targets);
    for (size_t k = 0; k < n_targets; ++k) c->slot_of[k] = slot_of_cls[cls[k]];
  }
  u64 k_is[kNumRouted];
  k_is[0] = 1;
  for (u32 j = 1; j < kNumRouted; ++j) k_is[j] = h_mul(k_is[j - 1], 7);
  {
    // members of every class in row-major order; sigma sends a wire to the next member of its class (cyclically)
    std::vector<u32> first(n_cls, 0xFFFFFFFFu), last(n_cls, 0xFFFFFFFFu), next(n * kNumRouted, 0);
    for (size_t row = 0; row < n; ++row)
      for (u32 col = 0; col < kNumRouted; ++col) {
        const u32 s = cls[row * kNumWires + col], id = (u32)(row * kNumRouted + col);
        if (first[s] == 0xFFFFFFFFu) first[s] = id;
        else next[last[s]] = id;
        last[s] = id;
        next[id] = first[s];
      }
    std::vector<u64> subgroup(n);
    const u64 g = h_root_of_unity(c->degree_bits);
    subgroup[0] = 1;
    for (size_t k = 1; k < n; ++k) subgroup[k] = h_mul(subgroup[k - 1], g);
    const u32 s0 = c->num_selectors + kNumConsts;
    for (u32 col = 0; col < kNumRouted; ++col)
      for (size_t row = 0; row < n; ++row) {
        const u32 nb = next[row * kNumRouted + col];
        c->h_cs[(size_t)(s0 + col) * n + row] = h_mul(k_is[nb % kNumRouted], subgroup[nb / kNumRouted]);
      }
  }
  for (const auto& g : b->gens) {
    if (g.kind == GEN_CONST) c->const_inits.emplace_back(c->slot_of[g.row * kNumWires + g.i], g.c0);
    else c->gens.push_back(g);
  }
  P2MT_TRY(p2mt_fri_params_standard(c->degree_bits, &c->fri));
  {
    const uint64_t np[4] = {n_cs, kNumWires, kNumZs, kNumQuot};
    c->fri_len = p2mt_fri_proof_len(&c->fri, 4, np);
    if (c->fri_len == 0) return p2mt::fail(P2MT_EINVAL, "build: unsupported FRI shape for this degree");
  }
  const size_t n_open = n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot;
  c->proof_len = 3 * 64 + 2 * n_open + c->fri_len + c->n_pi;

  {  // value table in LDS when it fits (160 KB per CU on gfx950; env P2MT_WITNESS_LDS=0 forces the global-memory path)
    const size_t need = (size_t)c->n_slots * 9 + 16;
    const char* e = getenv("P2MT_WITNESS_LDS");
    if (need <= 160 * 1024 - 36 * 1024 && !(e && e[0] == '0')) {  // minus the static round-constant, level and generator tables
      c->lds_bytes = (need + 15) & ~(size_t)15;
      if (c->lds_bytes > 64 * 1024) {
        // the limit is a property of the kernel, shared by every circuit (and thread): only ever raise it
        static std::atomic<size_t> raised{64 * 1024};
        size_t cur = raised.load();
        while (c->lds_bytes > cur && !raised.compare_exchange_weak(cur, c->lds_bytes)) {
        }
        P2MT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_witness_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)raised.load()));
      }
    }
  }
  // ---- device memory
  const size_t big = n << kRateBits;
  const size_t nd = c->n_digests = digests_count(big, kCapHeight);
  size_t words = 0;
  auto carve = [&](size_t w) {
    const size_t at = words;
    words += (w + 3) & ~(size_t)3;
    return at;
  };
  struct Batch {
    size_t vals, coeffs, lde, leaves, dig;
  };
  auto carve_batch = [&](size_t polys) { return Batch{carve(polys * n), carve(polys * n), carve(polys * big), carve(polys * big), carve(4 * (nd ? nd : 1))}; };
  const Batch bc = carve_batch(n_cs), bw = carve_batch(kNumWires), bz = carve_batch(kNumZs), bq = carve_batch(kNumQuot);
  const size_t o_qvals = carve(kNumCh * big), o_ppq = carve((size_t)kNumCh * kNumChunks * n);
  const size_t o_head = carve(8 + c->proof_len + 2), o_open = carve(2 * n_open), o_chal = carve(8), o_kis = carve(kNumRouted);
  const size_t o_vals = carve(c->n_slots), o_set = carve((c->n_slots + 1) / 2);
  c->init_cap = c->const_inits.size() + n_targets + 12 * kHostChainCap + kHostChainArithCap;  // (+ the outputs of the host-evaluated PoseidonGate rows)
  const size_t o_init = carve(2 * c->init_cap);
  const size_t o_wslot = carve(n * kNumWires + 1), o_pislot = carve((c->n_pi + 2) / 2);
  c->ops_cap = c->gens.size() + (c->has_recursion_gates ? 3 * c->gens.size() + 8192 : 0);  // room for the schedule's padding records
  const size_t o_ops = carve((c->ops_cap + 1) * sizeof(WOp) / 8 + 1), o_lvl = carve(c->ops_cap + 2);
  const size_t o_pslots = carve((size_t)c->counts[G_POSEIDON] * 16 + 16), o_prows = carve(c->counts[G_POSEIDON] / 2 + 1);
  c->args_cap = 0;
  for (const auto& g : c->gens)
    if (g.kind == GEN_QUOTIENT_EXT || g.kind == GEN_WIRE_SPLIT) c->args_cap += g.t.size();
    else if (g.kind == GEN_REDUCING_LOCAL) c->args_cap += 2 + kReducingCoeffs + 4;
    else if (g.kind == GEN_REDUCING_EXT_LOCAL) c->args_cap += 2 + 2 * kReducingExtCoeffs + 4;
    else if (g.kind == GEN_REDUCING_COMBINE) c->args_cap += 8;
  const size_t o_slot_tab = carve((n * kNumWires + 1) / 2 + 1), o_args = carve(c->args_cap / 2 + 1), o_sync = carve(4);
  const size_t o_q_extra = c->has_recursion_gates ? carve((size_t)(G_KINDS - G_POSEIDON - 1) * kNumCh * big) : 0;
  if (hipMalloc((void**)&c->d_base, words * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(circuit) failed");
  c->pin_pairs_off = 8 + c->proof_len + 2;
  if (hipHostMalloc((void**)&c->h_pin, (c->pin_pairs_off + 2 * c->init_cap) * 8, hipHostMallocDefault) != hipSuccess)
    return p2mt::fail(P2MT_ENOMEM, "hipHostMalloc(circuit staging) failed");
  u64* base = c->d_base;
  c->d_cs_vals = base + bc.vals, c->d_cs_coeffs = base + bc.coeffs, c->d_cs_lde = base + bc.lde, c->d_cs_leaves = base + bc.leaves, c->d_cs_dig = base + bc.dig;
  c->d_w_vals = base + bw.vals, c->d_w_coeffs = base + bw.coeffs, c->d_w_lde = base + bw.lde, c->d_w_leaves = base + bw.leaves, c->d_w_dig = base + bw.dig;
  c->d_z_vals = base + bz.vals, c->d_z_coeffs = base + bz.coeffs, c->d_z_lde = base + bz.lde, c->d_z_leaves = base + bz.leaves, c->d_z_dig = base + bz.dig;
  c->d_q_coeffs = base + bq.coeffs, c->d_q_lde = base + bq.lde, c->d_q_leaves = base + bq.leaves, c->d_q_dig = base + bq.dig;
  c->d_q_vals = base + o_qvals, c->d_pp_q = base + o_ppq;
  c->d_head = base + o_head, c->d_open = base + o_open, c->d_chal = base + o_chal, c->d_kis = base + o_kis;
  c->d_vals = base + o_vals, c->d_set = (u32*)(base + o_set), c->d_init = base + o_init;
  c->d_wire_slot = (u32*)(base + o_wslot), c->d_pi_slot = (u32*)(base + o_pislot);
  c->d_ops = (WOp*)(base + o_ops), c->d_lvl = (u32*)(base + o_lvl), c->d_err = (int*)(base + o_head + 8 + c->proof_len);
  c->d_pslots = (u32*)(base + o_pslots);
  c->d_prows = (u32*)(base + o_prows);
  c->d_slot_tab = (u32*)(base + o_slot_tab), c->d_args = (u32*)(base + o_args), c->d_sync = (u32*)(base + o_sync);
  c->d_q_extra = c->has_recursion_gates ? base + o_q_extra : nullptr;
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemsetAsync(c->d_base, 0, words * 8, st));
  std::vector<u32> pi_slot(c->n_pi);
  for (u32 k = 0; k < c->n_pi; ++k) pi_slot[k] = c->slot_of[target_index(c, c->public_inputs[k])];
  // dense wire -> slot table (the generators of the recursion gates find their operands through it); slot_of's wire part IS it
  P2MT_HIP(hipMemcpyAsync(c->d_slot_tab, c->slot_of.data(), n * kNumWires * 4, hipMemcpyHostToDevice, st));
  P2MT_HIP(hipMemcpyAsync(c->d_cs_vals, c->h_cs.data(), c->h_cs.size() * 8, hipMemcpyHostToDevice, st));
  std::copy(k_is, k_is + kNumRouted, c->k_is);
  P2MT_HIP(hipMemcpyAsync(c->d_kis, k_is, sizeof k_is, hipMemcpyHostToDevice, st));
  {
    std::vector<u32> prows;
    for (size_t row = 0; row < n; ++row)
      if (b->gates[row].kind == G_POSEIDON) prows.push_back((u32)row);
    if (!prows.empty()) {  // on the library stream, behind the zero-fill of the allocation above
      P2MT_HIP(hipMemcpyAsync(c->d_prows, prows.data(), prows.size() * 4, hipMemcpyHostToDevice, st));
      P2MT_HIP(hipStreamSynchronize(st));  // `prows` dies here
    }
  }
  {  // (wire index in the [col][row] matrix, slot) of every wire that owns a slot
    std::vector<u32> act;
    for (u32 col = 0; col < kNumWires; ++col)
      for (size_t row = 0; row < n; ++row) {
        const u32 sl = c->slot_of[row * kNumWires + col];
        if (sl != kNoSlot) {
          act.push_back((u32)(col * n + row));
          act.push_back(sl);
        }
      }
    c->n_act = (u32)(act.size() / 2);
    if (c->n_act) P2MT_HIP(hipMemcpyAsync(c->d_wire_slot, act.data(), act.size() * 4, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipStreamSynchronize(st));  // `act` dies here
  }
  if (c->n_pi) P2MT_HIP(hipMemcpyAsync(c->d_pi_slot, pi_slot.data(), c->n_pi * 4, hipMemcpyHostToDevice, st));
  // constants_sigmas commitment and the circuit digest = hash_no_pad(cap || D || degree_bits)
  u64* d_cap = c->d_head + 8;  // borrowed: the proof buffer is not in use yet
  P2MT_TRY(p2mt::commit_batch_dev(c->d_cs_vals, 1, n_cs, c->degree_bits, kRateBits, kCapHeight, c->d_cs_coeffs, c->d_cs_lde,
                                  c->d_cs_leaves, nd ? c->d_cs_dig : nullptr, d_cap));
  {  // D = digest of the empty domain separator, by convention (circuit_types.h kDigestDomainSeparator), then degree_bits
    static_assert(kPowRule == 0, "only fri_proof_of_work's observe-then-squeeze rule is implemented on the device");
    size_t len = 64;
    if (kDigestDomainSeparator == kDomainSepHashPad) {  // hash_pad([]): pad10*1 to the sponge rate
      const u64 padded[8] = {1, 0, 0, 0, 0, 0, 0, 1};
      P2MT_HIP(hipMemcpyAsync(d_cap + 72, padded, sizeof padded, hipMemcpyHostToDevice, st));
      P2MT_TRY(p2mt::launch_hash_rows_dev(d_cap + 72, 1, 8, 0, d_cap + 64));
      len += 4;
    } else if (kDigestDomainSeparator == kDomainSepZeroHash) {
      P2MT_HIP(hipMemsetAsync(d_cap + 64, 0, 32, st));
      len += 4;
    }
    const u64 db = c->degree_bits;
    P2MT_HIP(hipMemcpyAsync(d_cap + len, &db, 8, hipMemcpyHostToDevice, st));
    P2MT_TRY(p2mt::launch_hash_rows_dev(d_cap, 1, len + 1, 0, c->d_head));
  }
  P2MT_HIP(hipMemcpyAsync(c->cs_cap, d_cap, sizeof c->cs_cap, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(c->digest, c->d_head, sizeof c->digest, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  P2MT_TRY(p2mt_challenger_create(&c->ch));
  guard.c = nullptr;
  *out = c;
  return P2MT_OK;
  });
}

extern "C" int p2mt_circuit_get_info(const p2mt_circuit_data* c, p2mt_circuit_info* info) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !info) return p2mt::fail(P2MT_EINVAL, "null pointer");
  info->degree_bits = c->degree_bits;
  info->num_gate_types = c->n_kinds;
  info->num_selectors = c->num_selectors;
  info->num_constants_sigmas = c->n_cs;
  info->num_public_inputs = c->n_pi;
  info->num_partial_products = kNumProds;
  info->proof_len = c->proof_len;
  info->fri_proof_len = c->fri_len;
  for (u32 k = 0; k < kMaxGateTypes; ++k) info->gate_counts[k] = c->counts[k];
  for (u32 g = 0; g < kMaxGateTypes; ++g) {
    info->gate_kinds[g] = g < c->n_kinds ? c->kind[g] : 0;
    info->gate_selector[g] = g < c->n_kinds ? c->sel[g] : 0;
    info->group_start[g] = g < c->n_kinds ? c->gs[g] : 0;
    info->group_end[g] = g < c->n_kinds ? c->ge[g] : 0;
  }
  return P2MT_OK;
  });
}
int p2mt_circuit_common_data(const p2mt_circuit_data* c, p2mt_common_data* out) {
  if (!c || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *out = p2mt_common_data{};
  out->degree_bits = c->degree_bits, out->num_selectors = c->num_selectors, out->n_kinds = c->n_kinds, out->num_public_inputs = c->n_pi;
  for (u32 g = 0; g < kMaxGateTypes; ++g) out->kind[g] = c->kind[g], out->sel[g] = c->sel[g], out->gs[g] = c->gs[g], out->ge[g] = c->ge[g];
  std::copy(c->k_is, c->k_is + kNumRouted, out->k_is);
  out->fri = c->fri;
  out->proof_len = c->proof_len;
  std::copy(c->cs_cap, c->cs_cap + 64, out->cs_cap);
  std::copy(c->digest, c->digest + 4, out->digest);
  return P2MT_OK;
}

extern "C" int p2mt_circuit_public_inputs(const p2mt_circuit_data* c, p2mt_target* out) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || (c->n_pi && !out)) return p2mt::fail(P2MT_EINVAL, "null pointer");
  for (u32 k = 0; k < c->n_pi; ++k) out[k] = c->public_inputs[k];
  return P2MT_OK;
  });
}
extern "C" int p2mt_circuit_constants_sigmas(const p2mt_circuit_data* c, uint64_t* values_out, uint64_t* cap_out, uint64_t* digest_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!c) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (values_out) std::copy(c->h_cs.begin(), c->h_cs.end(), values_out);
  if (cap_out) std::copy(c->cs_cap, c->cs_cap + 64, cap_out);
  if (digest_out) std::copy(c->digest, c->digest + 4, digest_out);
  return P2MT_OK;
  });
}

// ==================================================================================================== PartialWitness
extern "C" int p2mt_pw_create(p2mt_partial_witness** out) {
  return p2mt::abi_guard([&]() -> int {
  if (!out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  *out = new (std::nothrow) p2mt_partial_witness;
  return *out ? P2MT_OK : p2mt::fail(P2MT_ENOMEM, "out of host memory");
  });
}
extern "C" int p2mt_pw_destroy(p2mt_partial_witness* pw) {
  return p2mt::abi_guard([&]() -> int {
  delete pw;
  return P2MT_OK;
  });
}
extern "C" int p2mt_pw_set_target(p2mt_partial_witness* pw, p2mt_target t, uint64_t value) {
  return p2mt::abi_guard([&]() -> int {
  if (!pw) return p2mt::fail(P2MT_EINVAL, "null pointer");
  pw->sets.emplace_back(t, value % gl::P);
  return P2MT_OK;
  });
}
extern "C" int p2mt_pw_clear(p2mt_partial_witness* pw) {
  return p2mt::abi_guard([&]() -> int {
  if (!pw) return p2mt::fail(P2MT_EINVAL, "null pointer");
  pw->sets.clear();
  return P2MT_OK;
  });
}

// ==================================================================================================== prove
extern "C" int p2mt_circuit_generate_witness(p2mt_circuit_data* c, const p2mt_partial_witness* pw, uint64_t* wires_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !pw || !wires_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  P2MT_TRY(fill_witness(c, &pw, 1));
  hipStream_t st = rt().stream;
  int err[4] = {0, 0, 0, 0};
  P2MT_HIP(hipMemcpyAsync(wires_out, c->d_w_vals, (size_t)kNumWires * c->n * 8, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipMemcpyAsync(err, c->d_err, sizeof err, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  err[3] = 0;  // (the opening-point flag belongs to prove)
  return witness_status(c, err);
  });
}

static int prove_once(p2mt_circuit_data* c, const p2mt_partial_witness* const* pws, unsigned B, uint64_t* proofs_out,
                      size_t proof_stride, int* status_out, int* gave_up);

// circuit_data.prove(pw).  The grid-wide witness interpreters assume that all of their (few) workgroups are resident; if the device
// is so oversubscribed by other work that a wait runs out of its budget, the launch drains with an error flag instead of hanging,
// and the proof is redone ONCE with the single-workgroup interpreter, which has no such assumption.
extern "C" int p2mt_circuit_prove(p2mt_circuit_data* c, const p2mt_partial_witness* pw, uint64_t* proof_out, size_t proof_cap) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !pw || !proof_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (proof_cap < c->proof_len) return p2mt::fail(P2MT_EINVAL, "prove: proof buffer too small (p2mt_circuit_get_info)");
  if (p2mt::batch_B() != 1) return p2mt::fail(P2MT_EINVAL, "prove: called inside a batched prove");
  int gave_up = 0;
  int rc = prove_once(c, &pw, 1, proof_out, 0, nullptr, &gave_up);
  if (gave_up && !c->force_single_workgroup) {
    c->force_single_workgroup = true;
    rc = prove_once(c, &pw, 1, proof_out, 0, nullptr, &gave_up);
    c->force_single_workgroup = false;
  }
  return rc;
  });
}

// One pass of the prover pipeline over B proofs (B = 1 outside a batch; inside one every launch carries the proofs in grid z
// and the per-proof pointers below are those of block 0, see runtime.h BatchCtx).  Nothing but the proof-of-work result visits
// the host before the proofs do.
static int prove_once(p2mt_circuit_data* c, const p2mt_partial_witness* const* pws, unsigned B, uint64_t* proofs_out,
                      size_t proof_stride, int* status_out, int* gave_up) {
  *gave_up = 0;
  hipStream_t st = rt().stream;
  const u32 n = c->n, log_n = c->degree_bits, log_big = log_n + kRateBits, big = n << kRateBits, n_cs = c->n_cs;
  u64 *d_digest = c->d_head, *d_pi_hash = c->d_head + 4, *d_proof = c->d_head + 8;
  u64 *d_w_cap = d_proof, *d_z_cap = d_proof + 64, *d_q_cap = d_proof + 128, *d_open_set = d_proof + 192;
  const size_t n_open = n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot;
  u64* d_fri = d_open_set + 2 * n_open;
  u64* d_pi = d_proof + (c->proof_len - c->n_pi);

  // witness, public-input hash, wires commitment
  P2MT_TRY(fill_witness(c, pws, B));
  // (the circuit digest has been sitting at d_head[0..4) since build(): nothing to upload)
  if (c->n_pi) P2MT_TRY(p2mt::launch_hash_rows_dev(d_pi, 1, c->n_pi, 0, d_pi_hash));
  else P2MT_TRY(p2mt::batch_fill(d_pi_hash, 0, 32));
  P2MT_TRY(p2mt::commit_batch_dev(c->d_w_vals, 1, kNumWires, log_n, kRateBits, kCapHeight, c->d_w_coeffs, c->d_w_lde, c->d_w_leaves,
                                  c->n_digests ? c->d_w_dig : nullptr, d_w_cap));
  // challenger: circuit digest, public-input hash, wires cap -> betas, gammas.
  // ONE proof: the transcript is a chain of ~100 dependent permutations (6.9 us each on a lone wavefront: 44 % of a prove) and plonky2
  // keeps it on the host; so does this path (host_poseidon.h, ~1.4 us each): at every phase the cap comes down through the host link
  // and the challenges go back up as kernel arguments.  A batch keeps the device transcript (one lane-parallel launch for all proofs).
  const bool host_tr = B == 1 && host_transcript_on();
  host_poseidon::Challenger hc;
  const size_t n_open_words = 2 * (size_t)(n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot);
  if (host_tr) {
    if (!c->link) P2MT_TRY(p2mt::hostlink_create(&c->link, n_open_words + 256));
    const uint64_t* h;
    P2MT_TRY(p2mt::hostlink_fetch(c->link, d_pi_hash, 4 + 64, &h));  // [public-input hash | wires cap]: contiguous in the proof block
    hc.observe(c->digest, 4);
    hc.observe(h, 4 + 64);
    u64 bg[2 * kNumCh];
    hc.squeeze(bg, 2 * kNumCh);
    P2MT_TRY(p2mt::hostlink_put(bg, 2 * kNumCh, c->d_chal));
  } else {
    P2MT_TRY(p2mt_challenger_restart_duplex_dev(c->ch, d_digest, 8 + 64, c->d_chal, 2 * kNumCh));
  }
  // Z and partial products, committed with Z at the front
  P2MT_TRY(p2mt::partial_products_async_dev(c->d_w_vals, c->d_cs_vals + (size_t)(c->num_selectors + kNumConsts) * n, c->d_kis,
                                            c->d_chal, c->d_chal + kNumCh, kNumCh, kNumRouted, log_n, kQF, c->d_pp_q, c->d_z_vals,
                                            c->d_err + 1));
  P2MT_TRY(p2mt::commit_batch_dev(c->d_z_vals, 1, kNumZs, log_n, kRateBits, kCapHeight, c->d_z_coeffs, c->d_z_lde, c->d_z_leaves,
                                  c->n_digests ? c->d_z_dig : nullptr, d_z_cap));
  if (host_tr) {
    const uint64_t* h;
    P2MT_TRY(p2mt::hostlink_fetch(c->link, d_z_cap, 64, &h));
    hc.observe(h, 64);
    u64 al[kNumCh];
    hc.squeeze(al, kNumCh);
    P2MT_TRY(p2mt::hostlink_put(al, kNumCh, c->d_chal + 2 * kNumCh));
  } else {
    P2MT_TRY(p2mt_challenger_duplex_dev(c->ch, d_z_cap, 64, c->d_chal + 2 * kNumCh, kNumCh));  // alphas
  }
  // quotient polynomials
  QDesc qd{};
  qd.log_n = log_n;
  qd.n_kinds = c->n_kinds;
  qd.num_selectors = c->num_selectors;
  qd.n_cs = n_cs;
  for (u32 g = 0; g < kMaxGateTypes; ++g) qd.kind[g] = c->kind[g], qd.sel[g] = c->sel[g], qd.gs[g] = c->gs[g], qd.ge[g] = c->ge[g];
  {
    const u64 shift_n = h_pow(7, n), w_q = h_root_of_unity(3);
    for (u32 k = 0; k < kQF; ++k) {
      qd.zh[k] = h_add(h_mul(shift_n, h_pow(w_q, k)), gl::P - 1);
      qd.zh_inv[k] = h_pow(qd.zh[k], gl::P - 2);
    }
    qd.n_inv = h_pow(n, gl::P - 2);
    qd.w_big = h_root_of_unity(log_big);
  }
  u32 n_extra = 0;
  for (u32 g = 0; g < c->n_kinds; ++g) n_extra += c->kind[g] > G_POSEIDON;
  if (n_extra) {
    hipLaunchKernelGGL(k_quotient_extra, bgrid(grid_for(big), n_extra), dim3(kBlock), 0, st, qd, (const u64*)c->d_cs_lde,
                       (const u64*)c->d_w_lde, (const u64*)c->d_chal, c->d_q_extra, barg());
    P2MT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_quotient, bgrid((big + 63) / 64), dim3(256), 0, st, qd, (const u64*)c->d_cs_lde, (const u64*)c->d_w_lde,
                     (const u64*)c->d_z_lde, (const u64*)d_pi_hash, (const u64*)c->d_chal, (const u64*)c->d_kis,
                     (const u64*)rt().d_rc, (const u64*)c->d_q_extra, n_extra, c->d_q_vals, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_TRY(p2mt::coset_ifft_dev(c->d_q_vals, log_big, kNumCh, 7, c->d_q_coeffs));
  P2MT_TRY(p2mt::commit_batch_dev(c->d_q_coeffs, 0, kNumQuot, log_n, kRateBits, kCapHeight, nullptr, c->d_q_lde, c->d_q_leaves,
                                  c->n_digests ? c->d_q_dig : nullptr, d_q_cap));
  // zeta stays on the device: the openings, the FRI quotients and plonky2's `ensure!(zeta^n != 1)` (a flag that comes back with
  // the proof) all read it there
  u64* d_zeta = c->d_chal + 3 * kNumCh;
  if (host_tr) {
    const uint64_t* h;
    P2MT_TRY(p2mt::hostlink_fetch(c->link, d_q_cap, 64, &h));
    hc.observe(h, 64);
    u64 z[2];
    hc.squeeze(z, 2);
    P2MT_TRY(p2mt::hostlink_put(z, 2, d_zeta));
  } else {
    P2MT_TRY(p2mt_challenger_duplex_dev(c->ch, d_q_cap, 64, d_zeta, 2));
  }
  // openings at zeta (every polynomial) and g zeta (the Z's), observed; then the FRI proof
  p2mt_fri_oracle oracles[4] = {{c->d_cs_coeffs, c->d_cs_leaves, c->d_cs_dig, n_cs},
                                {c->d_w_coeffs, c->d_w_leaves, c->d_w_dig, kNumWires},
                                {c->d_z_coeffs, c->d_z_leaves, c->d_z_dig, kNumZs},
                                {c->d_q_coeffs, c->d_q_leaves, c->d_q_dig, kNumQuot}};
  std::vector<uint32_t> all_polys, next_polys;
  for (u32 o = 0; o < 4; ++o)
    for (u32 k = 0; k < oracles[o].n_polys; ++k) {
      all_polys.push_back(o);
      all_polys.push_back(k);
    }
  for (u32 k = 0; k < kNumCh; ++k) {
    next_polys.push_back(2);
    next_polys.push_back(k);
  }
  p2mt_fri_batch batches[2] = {{{0, 0}, all_polys.data(), all_polys.size() / 2}, {{0, 0}, next_polys.data(), next_polys.size() / 2}};
  p2mt::FriPointsDev pts{};
  pts.d_point[0] = pts.d_point[1] = d_zeta;
  pts.scale[0] = 1;
  pts.scale[1] = h_root_of_unity(log_n);
  P2MT_TRY(p2mt::fri_openings_points_dev(oracles, 4, batches, 2, pts, log_n, c->d_open));
  if (!host_tr) P2MT_TRY(p2mt_challenger_observe_dev(c->ch, c->d_open, 2 * n_open));
  hipLaunchKernelGGL(k_opening_set, bgrid(grid_for(2 * n_open)), dim3(kBlock), 0, st, (const u64*)c->d_open, d_open_set, n_cs,
                     (const u64*)d_zeta, log_n, c->d_err, barg());
  P2MT_LAUNCH_CHECK();
  // the proofs (with the error flags right behind them) ride back on the FRI prover's own final synchronisation
  if (host_tr) {  // (behind k_opening_set, which is queued already: the openings come down while it runs)
    const uint64_t* h;
    P2MT_TRY(p2mt::hostlink_fetch(c->link, c->d_open, 2 * n_open, &h));
    hc.observe(h, 2 * n_open);
  }
  P2MT_TRY(p2mt::fri_prove_openings_epilogue_dev(oracles, 4, batches, 2, pts, &c->fri, c->ch, d_fri, c->h_pin + 8, d_proof,
                                                 (c->proof_len + 2) * 8, c->pin_pitch * 8, host_tr ? &hc : nullptr,
                                                 host_tr ? c->link : nullptr));
  int rc = P2MT_OK;
  for (unsigned bi = 0; bi < B; ++bi) {
    const u64* h = c->h_pin + (size_t)bi * c->pin_pitch + 8;
    int err[4];
    memcpy(err, h + c->proof_len, sizeof err);
    if (err[2] != 0) *gave_up = 1;
    const int r = witness_status(c, err);
    // plonky2 yields no proof where it panics or returns Err (contradicting witness, zero denominator, zeta in the subgroup): a
    // failed statement leaves zeros, not a well-formed-looking proof of garbage, next to its non-zero status
    if (r == P2MT_OK) std::copy(h, h + c->proof_len, proofs_out + (size_t)bi * proof_stride);
    else std::fill(proofs_out + (size_t)bi * proof_stride, proofs_out + (size_t)bi * proof_stride + c->proof_len, (uint64_t)0);
    if (status_out) status_out[bi] = r;
    if (r != P2MT_OK && rc == P2MT_OK) rc = r;
  }
  return rc;
}

// intermediates of the last p2mt_circuit_prove on this circuit (parity tests)
extern "C" int p2mt_circuit_prove_trace(const p2mt_circuit_data* c, int what, uint64_t* out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const u64* src;
  size_t words;
  switch (what) {
    case 0: src = c->d_w_vals, words = (size_t)kNumWires * c->n; break;
    case 1: src = c->d_z_vals, words = (size_t)kNumZs * c->n; break;
    case 2: src = c->d_q_coeffs, words = (size_t)kNumQuot * c->n; break;
    case 3: src = c->d_chal, words = 8; break;
    case 4: src = c->d_head + 4, words = 4; break;
    default: return p2mt::fail(P2MT_EINVAL, "prove_trace: unknown item");
  }
  P2MT_HIP(hipMemcpyAsync(out, src, words * 8, hipMemcpyDeviceToHost, rt().stream));
  P2MT_HIP(hipStreamSynchronize(rt().stream));
  return P2MT_OK;
  });
}

// ==================================================================================================== verify
// circuit_data.verify(proof) (mmr_plonky2_verifier.rs:150): transcript and Merkle paths on the device (two launches chains and
// two small copies) and so does the field arithmetic (p2mt_verify_dev.hip): launches only, one copy back.  *accepted = 1/0; *reason: 0 ok, 10 malformed
// (length, non-canonical word), 11 vanishing polynomial != Z_H * quotient at zeta (or zeta in the subgroup), 1 proof of work,
// 2 Merkle proof of an oracle row, 4 Merkle proof of a FRI layer, 3 layer value inconsistent, 5 final polynomial.
namespace {
// device layout of one proof under verification: digest | pi_hash | proof | cs_cap | openings in transcript order | challenges |
// flag | Merkle items
struct VLayout {
  size_t n_open, off_open, off_fri, off_pi, off_final, final_len, nq, max_items;
  size_t o_proof, o_cscap, o_fo, o_out, n_out, o_flag, o_res, o_pow, o_items, o_dig, words;
  u32 nred;
};
VLayout verify_layout(const p2mt_circuit_data* c) {
  VLayout L;
  L.nred = c->fri.num_reductions;
  L.n_open = c->n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot;
  L.off_open = 192, L.off_fri = L.off_open + 2 * L.n_open, L.off_pi = c->proof_len - c->n_pi;
  unsigned total_arity = 0;
  for (u32 l = 0; l < L.nred; ++l) total_arity += c->fri.reduction_arity_bits[l];
  L.final_len = (size_t)1 << (c->degree_bits - total_arity), L.nq = c->fri.num_query_rounds;
  L.off_final = L.off_fri + c->fri_len - 1 - 2 * L.final_len;
  L.max_items = L.nq * (4 + L.nred);
  L.o_proof = 8, L.o_cscap = L.o_proof + c->proof_len, L.o_fo = L.o_cscap + 64, L.o_out = L.o_fo + 2 * L.n_open;
  L.n_out = 8 + 2 + 2 * 8 + 1 + L.nq, L.o_flag = L.o_out + L.n_out + 1, L.o_res = L.o_flag + 1, L.o_pow = L.o_res + 1, L.o_items = L.o_pow + 1;
  L.o_dig = L.o_items + (L.max_items * sizeof(VItem) + 7) / 8 + 1;
  L.words = L.o_dig + 4 * L.max_items;
  return L;
}
constexpr int kFlagClear = 0x7F7F7F7F;  // "no failing Merkle path" (the kernel keeps the smallest failing item + 1)

// One pass over B proofs (B = 1: dv is the circuit's own block and no batch context is active; B > 1: dv is block 0 of B blocks
// `stride` bytes apart and the calling thread's batch context is set).  Everything that depends only on the proof -- the whole
// transcript, every Merkle path and all field arithmetic -- runs on the device with the proofs in grid z; the host decides from one small copy.
int verify_pass(p2mt_circuit_data* c, u64* dv, p2mt_challenger* ch, size_t stride, const uint64_t* proofs, size_t proof_stride,
                unsigned B, int* accepted, int* reason) {
  const VLayout L = verify_layout(c);
  const u32 n_cs = c->n_cs, log_n = c->degree_bits, log_big = log_n + kRateBits, nred = L.nred;
  const size_t nq = L.nq;
  hipStream_t st = rt().stream;
  // everything depends on the transcript only through device memory: the row sponges, the vanishing-polynomial check, the FRI
  // arithmetic and the path folds are staged beside / behind it (p2mt_verify_dev.hip), and ONE small copy brings back the verdict
  p2mt::VerifyDevArgs va{};
  va.d.degree_bits = log_n, va.d.num_wires = kNumWires, va.d.num_routed = kNumRouted, va.d.num_constants = kNumConsts;
  va.d.num_selectors = c->num_selectors, va.d.num_challenges = kNumCh, va.d.quotient_degree_factor = kQF, va.d.n_kinds = c->n_kinds;
  for (u32 g = 0; g < kMaxGateTypes; ++g) va.d.kind[g] = c->kind[g], va.d.sel[g] = c->sel[g], va.d.gs[g] = c->gs[g], va.d.ge[g] = c->ge[g];
  va.fri = c->fri;
  va.n_polys[0] = n_cs, va.n_polys[1] = kNumWires, va.n_polys[2] = kNumZs, va.n_polys[3] = kNumQuot;
  va.w_big = h_root_of_unity(log_big), va.w_n = h_root_of_unity(log_n), va.w16 = h_root_of_unity(4);
  va.o_proof = (u32)L.o_proof, va.o_fo = (u32)L.o_fo, va.o_out = (u32)L.o_out, va.o_cscap = (u32)L.o_cscap;
  va.off_open = (u32)L.off_open, va.off_fri = (u32)L.off_fri, va.off_final = (u32)L.off_final, va.final_len = (u32)L.final_len;
  {
    size_t qw = 0;
    for (u32 tr = 0; tr < 4; ++tr) qw += va.n_polys[tr] + 4 * (size_t)(log_big - kCapHeight);
    unsigned log_sz = log_big;
    for (u32 l = 0; l < nred; ++l) {
      const unsigned ab = c->fri.reduction_arity_bits[l];
      qw += ((size_t)2 << ab) + 4 * (size_t)(log_sz - ab - kCapHeight);
      log_sz -= ab;
    }
    va.query_words = (u32)qw;
  }
  if (!c->vstreams) P2MT_TRY(p2mt::verify_streams_create(&c->vstreams));
  int* d_flag = reinterpret_cast<int*>(dv + L.o_flag);      // [0] Merkle: smallest failing item + 1
  int* d_res = reinterpret_cast<int*>(dv + L.o_res);        // [0] openings ok, [1] FRI: 8 * query + reason
  std::vector<char> live(B, 1);
  for (unsigned b = 0; b < B; ++b) {
    accepted[b] = 0;
    reason[b] = 10;
    const uint64_t* pr = proofs + (size_t)b * proof_stride;
    for (size_t k = 0; k < c->proof_len; ++k)
      if (pr[k] >= gl::P) {
        live[b] = 0;
        break;
      }
  }
  // the circuit's own words (digest, constants_sigmas cap) go up once per block, not once per verification; nothing overwrites them.
  // (a batch block is filled for the largest batch it was sized for: B only shrinks within one allocation)
  const int slot = B > 1 ? 1 : 0;
  if (c->vconst_in[slot] != dv || B > c->vconst_B[slot]) {
    P2MT_HIP(hipMemcpyAsync(dv, c->digest, 32, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipMemcpyAsync(dv + L.o_cscap, c->cs_cap, sizeof c->cs_cap, hipMemcpyHostToDevice, st));
    if (B > 1) {
      P2MT_TRY(p2mt::batch_broadcast(dv, 32));
      P2MT_TRY(p2mt::batch_broadcast(dv + L.o_cscap, sizeof c->cs_cap));
    }
    c->vconst_in[slot] = dv;
    c->vconst_B[slot] = B;
  }
  if (B > 1) {
    P2MT_HIP(hipMemcpy2DAsync(dv + L.o_proof, stride, proofs, proof_stride * 8, c->proof_len * 8, B, hipMemcpyHostToDevice, st));
  } else {
    P2MT_HIP(hipMemcpyAsync(dv + L.o_proof, proofs, c->proof_len * 8, hipMemcpyHostToDevice, st));
  }
  P2MT_TRY(p2mt::batch_fill(d_flag, 0x7F, 16));  // flag word and result word: "nothing failed / nothing reported yet"
  const bool host_path = B == 1 && live[0] && host_transcript_on();
  P2MT_TRY(p2mt::verify_dev_begin(c->vstreams, dv, dv + L.o_dig, va, host_path));
  // from here on kernels are in flight on the side streams: whatever fails below, they are joined before this function returns, so
  // that the next pass on this block does not refill the flag words or the proof under them
  struct JoinSideStreams {
    void* vs;
    bool armed = true;
    ~JoinSideStreams() {
      if (armed) p2mt::verify_streams_join(vs);
    }
  } join{c->vstreams};
  if (host_path) {
    // ONE verification: the transcript is a chain of ~100 dependent permutations of words the host already holds -- 6.9 us each on
    // a lone wavefront, ~1.5 us on a host core (host_poseidon.h).  The host derives every challenge while the proof goes up and the
    // row sponges run, sends them up in one small copy, and the three checks that need them run side by side.
    namespace hp = host_poseidon;
    const size_t n_fo = 2 * L.n_open, vwords = 4 + n_fo + L.n_out;
    if (c->h_vpin_words < vwords) {
      if (c->h_vpin) (void)hipHostFree(c->h_vpin);
      c->h_vpin = nullptr, c->h_vpin_words = 0;
      if (hipHostMalloc((void**)&c->h_vpin, vwords * 8, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return p2mt::fail(P2MT_ENOMEM, "hipHostMalloc(verify transcript) failed");
      }
      c->h_vpin_words = vwords;
    }
    u64* h_pi = c->h_vpin;
    u64* h_fo = c->h_vpin + 4;
    u64* h_out = h_fo + n_fo;
    const uint64_t* pr = proofs;
    {
      u64 pi[4] = {0, 0, 0, 0};
      if (c->n_pi) hp::hash_no_pad(pr + L.off_pi, c->n_pi, pi);
      memcpy(h_pi, pi, 32);
    }
    {  // FriOpenings order from the proof's OpeningSet order (k_opening_unset)
      const u32 a = 2 * (n_cs + kNumWires + kNumCh), tail = 2 * (kNumCh * kNumProds + kNumQuot), total = a + 2 * kNumCh + tail;
      const uint64_t* set_order = pr + L.off_open;
      for (u32 t = 0; t < total; ++t) {
        const u32 dst = t < a ? t : (t < a + 2 * kNumCh ? a + tail + (t - a) : t - 2 * kNumCh);
        h_fo[dst] = set_order[t];
      }
    }
    hp::Challenger hc;
    hc.observe(c->digest, 4);
    hc.observe(h_pi, 4);
    hc.observe(pr, 64);                          // wires cap
    hc.squeeze(h_out, 2 * kNumCh);               // betas, gammas
    hc.observe(pr + 64, 64);                     // Z / partial-products cap
    hc.squeeze(h_out + 2 * kNumCh, kNumCh);      // alphas
    hc.observe(pr + 128, 64);                    // quotient cap
    hc.squeeze(h_out + 3 * kNumCh, 2);           // zeta
    hc.observe(h_fo, n_fo);
    hc.squeeze(h_out + 8, 2);                    // FRI alpha
    for (u32 l = 0; l < nred; ++l) {
      hc.observe(pr + L.off_fri + 64 * l, 64);
      hc.squeeze(h_out + 10 + 2 * l, 2);         // FRI betas
    }
    hc.observe(pr + L.off_final, 2 * L.final_len + 1);
    hc.squeeze(h_out + 26, 1);                   // PoW response
    hc.squeeze(h_out + 27, nq);                  // query indices
    P2MT_HIP(hipMemcpyAsync(dv + 4, h_pi, 32, hipMemcpyHostToDevice, st));
    P2MT_HIP(hipMemcpyAsync(dv + L.o_fo, h_fo, (n_fo + L.n_out) * 8, hipMemcpyHostToDevice, st));
    P2MT_TRY(p2mt::verify_dev_with_challenges(c->vstreams, dv, dv + L.o_dig, d_flag, d_res, c->d_kis, va));
  } else {
  if (c->n_pi) P2MT_TRY(p2mt::launch_hash_rows_dev(dv + L.o_proof + L.off_pi, 1, c->n_pi, 0, dv + 4));
  else P2MT_TRY(p2mt::batch_fill(dv + 4, 0, 32));
  // the whole transcript depends only on the proof: enqueue it in one go
  u64* d_out = dv + L.o_out;
  P2MT_TRY(p2mt_challenger_restart_duplex_dev(ch, dv, 8 + 64, d_out, 2 * kNumCh));                          // betas, gammas
  P2MT_TRY(p2mt_challenger_duplex_dev(ch, dv + L.o_proof + 64, 64, d_out + 2 * kNumCh, kNumCh));           // alphas
  P2MT_TRY(p2mt_challenger_duplex_dev(ch, dv + L.o_proof + 128, 64, d_out + 3 * kNumCh, 2));                // zeta
  P2MT_TRY(p2mt::verify_dev_after_zeta(c->vstreams, dv, d_res, c->d_kis, va));
  hipLaunchKernelGGL(k_opening_unset, bgrid(grid_for(2 * L.n_open)), dim3(kBlock), 0, st, (const u64*)(dv + L.o_proof + L.off_open),
                     dv + L.o_fo, n_cs, barg());
  P2MT_LAUNCH_CHECK();
  P2MT_TRY(p2mt_challenger_duplex_dev(ch, dv + L.o_fo, 2 * L.n_open, d_out + 8, 2));                       // FRI alpha
  for (u32 l = 0; l < nred; ++l)
    P2MT_TRY(p2mt_challenger_duplex_dev(ch, dv + L.o_proof + L.off_fri + 64 * l, 64, d_out + 10 + 2 * l, 2));  // FRI betas
  P2MT_TRY(p2mt_challenger_duplex_dev(ch, dv + L.o_proof + L.off_final, 2 * L.final_len + 1, d_out + 26, 1));    // PoW response
  P2MT_TRY(p2mt_challenger_get_challenges_dev(ch, nq, d_out + 27));                                           // query indices
  P2MT_TRY(p2mt::verify_dev_finish(c->vstreams, dv, dv + L.o_items, dv + L.o_dig, d_flag, d_res, va));
  }
  // results, gathered per proof in ONE copy: {Merkle flag, -} {openings ok, FRI} {proof-of-work response} -- three consecutive words
  // (k_verify_fri leaves a copy of the response next to its own result)
  struct Verdict {
    int flag, pad, ok, fri;
    u64 pow;
  };
  std::vector<Verdict> vr(B);
  static_assert(sizeof(Verdict) == 24, "three consecutive device words");
  if (B > 1) P2MT_HIP(hipMemcpy2DAsync(vr.data(), 24, dv + L.o_flag, stride, 24, B, hipMemcpyDeviceToHost, st));
  else P2MT_HIP(hipMemcpyAsync(vr.data(), dv + L.o_flag, 24, hipMemcpyDeviceToHost, st));
  P2MT_HIP(hipStreamSynchronize(st));
  join.armed = false;  // the library stream waited for both side streams before the verdict copy
  for (unsigned b = 0; b < B; ++b) {
    if (!live[b]) continue;  // reason 10
    if (vr[b].ok != 1) {
      reason[b] = 11;
    } else if (c->fri.proof_of_work_bits && (vr[b].pow >> (64 - c->fri.proof_of_work_bits)) != 0) {
      reason[b] = 1;
    } else if (vr[b].flag != kFlagClear) {
      reason[b] = ((size_t)(vr[b].flag - 1) % (4 + nred)) >= 4 ? 4 : 2;
    } else if (vr[b].fri != kFlagClear) {
      reason[b] = vr[b].fri & 7;
    } else {
      reason[b] = 0;
      accepted[b] = 1;
    }
  }
  return P2MT_OK;
}
}  // namespace

extern "C" int p2mt_debug_host_transcript(int on) {
  return p2mt::abi_guard([&]() -> int {
  host_transcript_flag() = on ? 1 : 0;
  return P2MT_OK;
  });
}

extern "C" int p2mt_debug_host_chain(int on) {
  return p2mt::abi_guard([&]() -> int {
  host_chain_flag() = on ? 1 : 0;
  return P2MT_OK;
  });
}
extern "C" int p2mt_circuit_schedule_info(const p2mt_circuit_data* c, uint32_t* n_levels, uint32_t* host_rows) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !n_levels || !host_rows) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (!c->sched_valid) return p2mt::fail(P2MT_EINVAL, "no schedule yet: prove once first");
  *n_levels = c->n_levels;
  *host_rows = c->host_rows;
  return P2MT_OK;
  });
}

extern "C" int p2mt_circuit_verify(p2mt_circuit_data* c, const uint64_t* proof, size_t proof_len, int* accepted, int* reason) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !proof || !accepted) return p2mt::fail(P2MT_EINVAL, "null pointer");
  int dummy = 0;
  if (!reason) reason = &dummy;
  *accepted = 0;
  *reason = 10;
  if (proof_len != c->proof_len) return P2MT_OK;
  if (p2mt::batch_B() != 1) return p2mt::fail(P2MT_EINVAL, "verify: called inside a batched pass");
  if (!c->d_verify) {
    if (hipMalloc((void**)&c->d_verify, verify_layout(c).words * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(verify) failed");
    P2MT_TRY(p2mt_challenger_create(&c->vch));
  }
  return verify_pass(c, c->d_verify, c->vch, 0, proof, proof_len, 1, accepted, reason);
  });
}

// circuit_data.verify for n proofs of this circuit, proofs[i] at proofs + i * proof_stride words: passes of up to 256 proofs with
// the proof index in a grid dimension of every launch (one transcript replay, one row-sponge, one openings-check, one FRI and one
// Merkle-path launch per pass: transcript and field arithmetic on the device).  accepted[i] / reason[i] as p2mt_circuit_verify.
extern "C" int p2mt_circuit_verify_batch(p2mt_circuit_data* c, const uint64_t* proofs, size_t n, size_t proof_stride, int* accepted,
                                         int* reason) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !proofs || !accepted || !reason) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (proof_stride < c->proof_len) return p2mt::fail(P2MT_EINVAL, "verify_batch: proof_stride < proof_len");
  if (p2mt::batch_B() != 1) return p2mt::fail(P2MT_EINVAL, "verify_batch: called inside a batched pass");
  if (n == 0) return P2MT_OK;
  constexpr size_t kMaxPass = 256;
  const VLayout L = verify_layout(c);
  const size_t want = std::min(n, kMaxPass);
  const size_t stride = ((L.words * 8 + p2mt::kChallengerStateBytes + 8) + 255) & ~(size_t)255;
  if (c->vbatch_cap < want) {
    // (the upload-once cache of the circuit's own words is keyed on the block pointer: forget both slots -- a one-proof pass runs on
    // this block under slot 0, and hipMalloc may hand the same address out again)
    c->vconst_in[0] = c->vconst_in[1] = nullptr;
    c->vconst_B[0] = c->vconst_B[1] = 0;
    if (c->d_vbatch) {
      (void)hipStreamSynchronize(rt().stream);
      (void)hipFree(c->d_vbatch);
      c->d_vbatch = nullptr;
      c->vbatch_cap = 0;
    }
    if (hipMalloc((void**)&c->d_vbatch, stride * want) != hipSuccess) {
      (void)hipGetLastError();
      c->d_vbatch = nullptr;
      return p2mt::fail(P2MT_ENOMEM, "hipMalloc(verify batch) failed");
    }
    c->vbatch_cap = want, c->vbatch_stride = stride;
    c->vconst_in[1] = nullptr;
    if (c->vbch) p2mt::challenger_unwrap(c->vbch);
    c->vbch = nullptr;
    P2MT_TRY(p2mt::challenger_wrap(c->d_vbatch + ((L.words * 8 + 7) & ~(size_t)7), &c->vbch));
  }
  struct Scope {
    ~Scope() { p2mt::batch() = p2mt::BatchCtx{}; }
  } scope;
  for (size_t at = 0; at < n; at += kMaxPass) {
    const unsigned cnt = (unsigned)std::min(kMaxPass, n - at);
    p2mt::BatchCtx& ctx = p2mt::batch();
    ctx = p2mt::BatchCtx{};
    if (cnt > 1) {
      ctx.B = cnt;
      ctx.arg = p2mt::BatchArg{(uint64_t)c->d_vbatch, (uint64_t)c->vbatch_stride, (uint64_t)c->vbatch_stride};
    }
    P2MT_TRY(verify_pass(c, reinterpret_cast<u64*>(c->d_vbatch), c->vbch, c->vbatch_stride, proofs + at * proof_stride, proof_stride, cnt,
                         accepted + at, reason + at));
  }
  return P2MT_OK;
  });
}

// ==================================================================================================== many proofs
// n independent proves spread over n_handles worker threads, worker t driving circuits[t] on its own stream (the pattern
// bench.py measures: ~2.8 k proofs/s for the 64-row circuit with 32 handles).  All handles must be builds of the same
// circuit (same targets); witnesses[i] -> proofs_out + i * proof_stride.  status_out[i] (may be NULL) gets each prove's
// status; the return value is the first non-zero one.
extern "C" int p2mt_circuit_prove_many(p2mt_circuit_data* const* circuits, size_t n_handles,
                                       const p2mt_partial_witness* const* witnesses, size_t n, uint64_t* proofs_out,
                                       size_t proof_stride, int* status_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!circuits || !witnesses || !proofs_out || n_handles == 0) return p2mt::fail(P2MT_EINVAL, "null pointer");
  for (size_t t = 0; t < n_handles; ++t) {
    if (!circuits[t]) return p2mt::fail(P2MT_EINVAL, "null circuit handle");
    if (circuits[t]->proof_len != circuits[0]->proof_len || circuits[t]->n_virtual != circuits[0]->n_virtual || proof_stride < circuits[t]->proof_len)
      return p2mt::fail(P2MT_EINVAL, "prove_many: handles must be builds of one circuit and proof_stride >= proof_len");
    for (size_t u = 0; u < t; ++u)
      if (circuits[u] == circuits[t]) return p2mt::fail(P2MT_EINVAL, "prove_many: a handle may appear only once (handles are single-threaded)");
  }
  std::atomic<size_t> next{0};
  std::atomic<int> first_err{P2MT_OK};
  auto worker = [&](size_t t) {
    hipStream_t s = nullptr;
    if (hipSetDevice(rt().device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
      int exp = P2MT_OK;
      first_err.compare_exchange_strong(exp, P2MT_EHIP);
      return;
    }
    rt().stream = s;  // per thread
    for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      const int rc = witnesses[i] ? p2mt_circuit_prove(circuits[t], witnesses[i], proofs_out + i * proof_stride, proof_stride)
                                  : P2MT_EINVAL;
      if (status_out) status_out[i] = rc;
      if (rc != P2MT_OK) {
        int exp = P2MT_OK;
        first_err.compare_exchange_strong(exp, rc);
      }
    }
    p2mt::scratch_release_thread();  // synchronises the stream first
    rt().stream = nullptr;
    (void)hipStreamDestroy(s);
  };
  std::vector<std::thread> pool;
  const size_t n_workers = std::min(n_handles, n);
  bool pool_failed = false;
  try {
    pool.reserve(n_workers);
    for (size_t t = 0; t < n_workers; ++t) pool.emplace_back(worker, t);
  } catch (...) {  // std::system_error: the threads that did start drain the whole queue
    pool_failed = pool.empty();
  }
  for (auto& th : pool) th.join();
  if (pool_failed) return p2mt::fail(P2MT_ENOMEM, "prove_many: could not start a worker thread");
  const int rc = first_err.load();
  return rc == P2MT_OK ? P2MT_OK : p2mt::fail(rc, "prove_many: at least one prove failed (see status_out)");
  });
}

// ==================================================================================================== batched prover
// B proofs of one circuit per pipeline pass: every launch of prove_once carries the proofs in grid dimension z, so a batch costs
// the dispatch packets of ONE proof (the concurrent-prover path is bound by the device's packet rate).  Each proof
// owns a block of device memory -- the per-proof buffers of p2mt_circuit_data, a scratch arena and a challenger state --, the
// circuit's constants / schedule / sigma commitment stay shared.  Proofs are bit-identical to p2mt_circuit_prove's.
namespace {
struct PerProof {
  u64 *w_vals, *w_coeffs, *w_lde, *w_leaves, *w_dig, *z_vals, *z_coeffs, *z_lde, *z_leaves, *z_dig, *pp_q;
  u64 *q_vals, *q_coeffs, *q_lde, *q_leaves, *q_dig, *head, *open, *chal, *init, *q_extra, *vals;
  u32* set;
  int* err;
  p2mt_challenger* ch;
  u64* h_pin;
  size_t pin_pitch;
};
void exchange(p2mt_circuit_data* c, PerProof& p) {
  std::swap(c->d_w_vals, p.w_vals), std::swap(c->d_w_coeffs, p.w_coeffs), std::swap(c->d_w_lde, p.w_lde);
  std::swap(c->d_w_leaves, p.w_leaves), std::swap(c->d_w_dig, p.w_dig);
  std::swap(c->d_z_vals, p.z_vals), std::swap(c->d_z_coeffs, p.z_coeffs), std::swap(c->d_z_lde, p.z_lde);
  std::swap(c->d_z_leaves, p.z_leaves), std::swap(c->d_z_dig, p.z_dig), std::swap(c->d_pp_q, p.pp_q);
  std::swap(c->d_q_vals, p.q_vals), std::swap(c->d_q_coeffs, p.q_coeffs), std::swap(c->d_q_lde, p.q_lde);
  std::swap(c->d_q_leaves, p.q_leaves), std::swap(c->d_q_dig, p.q_dig);
  std::swap(c->d_head, p.head), std::swap(c->d_open, p.open), std::swap(c->d_chal, p.chal), std::swap(c->d_init, p.init);
  std::swap(c->d_vals, p.vals), std::swap(c->d_set, p.set);
  std::swap(c->d_q_extra, p.q_extra), std::swap(c->d_err, p.err), std::swap(c->ch, p.ch), std::swap(c->h_pin, p.h_pin);
  std::swap(c->pin_pitch, p.pin_pitch);
}
}  // namespace

struct p2mt_batch_prover {
  p2mt_circuit_data* c = nullptr;
  unsigned B = 0;
  bool ready = false;
  size_t stride = 0;  // bytes per proof block
  char* d_block = nullptr;
  u64* h_pin = nullptr;
  PerProof pp{};  // block 0's pointers (exchanged with the circuit's own for the duration of a pass)
  size_t arena_off = 0, slot_off[p2mt::kScratchCount] = {}, slot_cap[p2mt::kScratchCount] = {};
};

extern "C" int p2mt_batch_prover_create(p2mt_circuit_data* c, size_t batch, p2mt_batch_prover** out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!c || !out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (batch == 0 || batch > 4096) return p2mt::fail(P2MT_EINVAL, "batch_prover: batch must be in 1..4096");
  p2mt_batch_prover* b = new (std::nothrow) p2mt_batch_prover;
  if (!b) return p2mt::fail(P2MT_ENOMEM, "out of host memory");
  b->c = c;
  b->B = (unsigned)batch;
  *out = b;
  return P2MT_OK;
  });
}

extern "C" int p2mt_batch_prover_destroy(p2mt_batch_prover* b) {
  return p2mt::abi_guard([&]() -> int {
  if (!b) return P2MT_OK;
  if (b->d_block) {
    (void)hipStreamSynchronize(rt().stream);
    (void)hipFree(b->d_block);
  }
  if (b->h_pin) (void)hipHostFree(b->h_pin);
  if (b->pp.ch) p2mt::challenger_unwrap(b->pp.ch);
  delete b;
  return P2MT_OK;
  });
}

// first use: size the per-proof scratch arena from one tracked single-proof run, then allocate the blocks
static int batch_prover_prepare_impl(p2mt_batch_prover* b, const p2mt_partial_witness* pw);
// Any non-OK exit of the preparation gives everything back (device blocks -- several GB for the outer recursion circuit --, the
// pinned staging and the challenger wrapper), so the next prove simply starts over instead of overwriting live pointers.
static int batch_prover_prepare(p2mt_batch_prover* b, const p2mt_partial_witness* pw) {
  const int rc = batch_prover_prepare_impl(b, pw);
  if (rc != P2MT_OK && !b->ready) {
    (void)hipStreamSynchronize(rt().stream);
    if (b->d_block) (void)hipFree(b->d_block);
    if (b->h_pin) (void)hipHostFree(b->h_pin);
    b->d_block = nullptr;
    b->h_pin = nullptr;
    b->pp.h_pin = nullptr;
    if (b->pp.ch) p2mt::challenger_unwrap(b->pp.ch);
    b->pp.ch = nullptr;
  }
  return rc;
}
static int batch_prover_prepare_impl(p2mt_batch_prover* b, const p2mt_partial_witness* pw) {
  p2mt_circuit_data* c = b->c;
  std::vector<u64> tmp(c->proof_len);
  p2mt::scratch_track_reset();
  P2MT_TRY(p2mt_circuit_prove(c, pw, tmp.data(), tmp.size()));
  const size_t n = c->n, big = n << kRateBits, nd = c->n_digests;
  const size_t n_open = c->n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot;
  size_t words = 0;
  auto carve = [&](size_t w) {
    const size_t at = words;
    words += (w + 3) & ~(size_t)3;
    return at;
  };
  struct Bt {
    size_t vals, coeffs, lde, leaves, dig;
  };
  auto carve_batch = [&](size_t polys) { return Bt{carve(polys * n), carve(polys * n), carve(polys * big), carve(polys * big), carve(4 * (nd ? nd : 1))}; };
  const Bt bw = carve_batch(kNumWires), bz = carve_batch(kNumZs), bq = carve_batch(kNumQuot);
  const size_t o_qvals = carve(kNumCh * big), o_ppq = carve((size_t)kNumCh * kNumChunks * n);
  const size_t o_head = carve(8 + c->proof_len + 2), o_open = carve(2 * n_open), o_chal = carve(8), o_init = carve(2 * c->init_cap);
  const size_t o_q_extra = c->has_recursion_gates ? carve((size_t)(G_KINDS - G_POSEIDON - 1) * kNumCh * big) : 0;
  const size_t o_ch = carve(p2mt::kChallengerStateBytes / 8);
  const size_t o_vals = c->lds_bytes ? 0 : carve(c->n_slots), o_set = c->lds_bytes ? 0 : carve((c->n_slots + 1) / 2);
  b->arena_off = words * 8;
  size_t arena = 0;
  for (int k = 0; k < p2mt::kScratchCount; ++k) {
    b->slot_off[k] = arena;
    b->slot_cap[k] = (p2mt::scratch_track_max(k) + 255) & ~(size_t)255;
    arena += b->slot_cap[k];
  }
  b->stride = (b->arena_off + arena + 255) & ~(size_t)255;
  {  // the batched paths move per-proof fields with hipMemcpy2DAsync, block stride = pitch: HIP caps the pitch (~2 GB)
    int max_pitch = 0;
    P2MT_HIP(hipDeviceGetAttribute(&max_pitch, hipDeviceAttributeMaxPitch, rt().device));
    if (max_pitch > 0 && b->stride > (size_t)max_pitch)
      return p2mt::fail(P2MT_EINVAL, "batch prover: one proof's device block exceeds the device's maximum copy pitch (circuit too large "
                                     "for the batched prover; use p2mt_circuit_prove)");
  }
  if (hipMalloc((void**)&b->d_block, b->stride * b->B) != hipSuccess) {
    (void)hipGetLastError();
    b->d_block = nullptr;
    return p2mt::fail(P2MT_ENOMEM, "hipMalloc(batch prover blocks) failed");
  }
  hipStream_t st = rt().stream;
  P2MT_HIP(hipMemsetAsync(b->d_block, 0, b->stride * b->B, st));
  const size_t pitch = (c->pin_pairs_off + 2 * c->init_cap + 3) & ~(size_t)3;
  if (hipHostMalloc((void**)&b->h_pin, pitch * 8 * b->B, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    b->h_pin = nullptr;
    (void)hipStreamSynchronize(st);
    (void)hipFree(b->d_block);  // a later call starts over
    b->d_block = nullptr;
    return p2mt::fail(P2MT_ENOMEM, "hipHostMalloc(batch prover staging) failed");
  }
  u64* base = reinterpret_cast<u64*>(b->d_block);
  PerProof& p = b->pp;
  p.w_vals = base + bw.vals, p.w_coeffs = base + bw.coeffs, p.w_lde = base + bw.lde, p.w_leaves = base + bw.leaves, p.w_dig = base + bw.dig;
  p.z_vals = base + bz.vals, p.z_coeffs = base + bz.coeffs, p.z_lde = base + bz.lde, p.z_leaves = base + bz.leaves, p.z_dig = base + bz.dig;
  p.q_coeffs = base + bq.coeffs, p.q_lde = base + bq.lde, p.q_leaves = base + bq.leaves, p.q_dig = base + bq.dig;
  p.q_vals = base + o_qvals, p.pp_q = base + o_ppq, p.head = base + o_head, p.open = base + o_open, p.chal = base + o_chal;
  p.init = base + o_init, p.q_extra = c->has_recursion_gates ? base + o_q_extra : nullptr;
  p.err = reinterpret_cast<int*>(base + o_head + 8 + c->proof_len);
  p.vals = c->lds_bytes ? nullptr : base + o_vals;
  p.set = c->lds_bytes ? nullptr : reinterpret_cast<u32*>(base + o_set);
  if (p.ch) p2mt::challenger_unwrap(p.ch);  // (left over from an attempt that ran out of memory)
  p.ch = nullptr;
  P2MT_TRY(p2mt::challenger_wrap(base + o_ch, &p.ch));
  p.h_pin = b->h_pin;
  p.pin_pitch = pitch;
  // the circuit digest sits in front of every proof buffer (build() put it into the circuit's own)
  for (unsigned i = 0; i < b->B; ++i)
    P2MT_HIP(hipMemcpyAsync(b->d_block + (size_t)i * b->stride + o_head * 8, c->d_head, 32, hipMemcpyDeviceToDevice, st));
  P2MT_HIP(hipStreamSynchronize(st));
  b->ready = true;
  return P2MT_OK;
}

// witnesses[i] -> proofs_out + i * proof_stride, i < n, in passes of up to `batch` proofs; status_out[i] (may be NULL) gets each
// proof's status, the return value is the first non-zero one.  Every witness must set the same targets in the same order.
extern "C" int p2mt_batch_prover_prove(p2mt_batch_prover* b, const p2mt_partial_witness* const* witnesses, size_t n,
                                       uint64_t* proofs_out, size_t proof_stride, int* status_out) {
  return p2mt::abi_guard([&]() -> int {
  P2MT_TRY(p2mt::ensure_init());
  if (!b || !witnesses || !proofs_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (proof_stride < b->c->proof_len) return p2mt::fail(P2MT_EINVAL, "prove_batch: proof_stride < proof_len (p2mt_circuit_get_info)");
  for (size_t i = 0; i < n; ++i)
    if (!witnesses[i]) return p2mt::fail(P2MT_EINVAL, "prove_batch: null witness");
  if (n == 0) return P2MT_OK;
  if (p2mt::batch_B() != 1) return p2mt::fail(P2MT_EINVAL, "prove_batch: called inside a batched prove");
  if (!b->ready) P2MT_TRY(batch_prover_prepare(b, witnesses[0]));
  p2mt_circuit_data* c = b->c;
  struct Scope {  // the circuit drives block 0 for the duration of the call; the thread's launches carry the batch
    p2mt_batch_prover* b;
    explicit Scope(p2mt_batch_prover* bb) : b(bb) { exchange(b->c, b->pp); }
    ~Scope() {
      exchange(b->c, b->pp);
      p2mt::batch() = p2mt::BatchCtx{};
    }
  } scope(b);
  int rc = P2MT_OK;
  for (size_t at = 0; at < n; at += b->B) {
    const unsigned cnt = (unsigned)std::min<size_t>(b->B, n - at);
    p2mt::BatchCtx& ctx = p2mt::batch();
    ctx.B = cnt;
    ctx.arg = p2mt::BatchArg{(uint64_t)b->d_block, (uint64_t)b->stride, (uint64_t)b->stride};
    ctx.arena = b->d_block + b->arena_off;
    for (int k = 0; k < p2mt::kScratchCount; ++k) ctx.slot_off[k] = b->slot_off[k], ctx.slot_cap[k] = b->slot_cap[k];
    int gave_up = 0;
    constexpr int kUnset = -0x7fffffff;  // a pass that fails before its proofs come back leaves no per-proof status
    std::vector<int> st(cnt, kUnset);
    const int r = prove_once(c, witnesses + at, cnt, proofs_out + at * proof_stride, proof_stride, st.data(), &gave_up);
    if (r != P2MT_OK && rc == P2MT_OK) rc = r;
    for (unsigned i = 0; i < cnt; ++i) {
      if (st[i] == kUnset) st[i] = r != P2MT_OK ? r : P2MT_EHIP;
      if (status_out) status_out[at + i] = st[i];
    }
  }
  return rc;
  });
}

extern "C" size_t p2mt_batch_prover_batch(const p2mt_batch_prover* b) { return b ? b->B : 0; }

// ==================================================================================================== proof bytes
// ProofWithPublicInputs::to_bytes / from_bytes in plonky2's Buffer order (util/serialization.rs @3b21b87d, absent from the
// reference tree -- SURVEY.md App. B.5; recalled, parity unpinned): every field element 8 bytes little endian, an extension
// element as its two coefficients; caps, openings, evaluations and the final polynomial carry no length (CommonCircuitData
// fixes them); a MerkleProof is ONE length byte (number of siblings) followed by the sibling hashes.  The word form of
// p2mt_circuit_prove has exactly that order without the length bytes, so the byte form is the words plus one byte in front of
// each Merkle path: 28 x (4 oracle paths + one per FRI layer).
namespace {
struct ProofShape {
  size_t head_words;                 // 3 caps + openings
  size_t caps_words;                 // commit-phase caps
  std::vector<std::pair<size_t, size_t>> per_query;  // (row / evals words, sibling count) per opened tree, in order
  size_t tail_words;                 // final polynomial + pow witness + public inputs
  size_t n_queries;
};
ProofShape proof_shape(const p2mt_circuit_data* c) {
  ProofShape sh;
  const size_t n_open = c->n_cs + kNumWires + 2 * kNumCh + kNumCh * kNumProds + kNumQuot;
  const p2mt_fri_params& f = c->fri;
  const unsigned log_big = f.degree_bits + f.rate_bits;
  sh.head_words = 3 * 64 + 2 * n_open;
  sh.caps_words = (size_t)f.num_reductions * ((size_t)4 << f.cap_height);
  const size_t polys[4] = {c->n_cs, kNumWires, kNumZs, kNumQuot};
  for (size_t o = 0; o < 4; ++o) sh.per_query.emplace_back(polys[o], (size_t)(log_big - f.cap_height));
  unsigned log_sz = log_big, arity_total = 0;
  for (uint32_t l = 0; l < f.num_reductions; ++l) {
    const unsigned ab = f.reduction_arity_bits[l];
    sh.per_query.emplace_back((size_t)2 << ab, (size_t)(log_sz - ab - f.cap_height));
    log_sz -= ab;
    arity_total += ab;
  }
  sh.n_queries = f.num_query_rounds;
  sh.tail_words = ((size_t)2 << (f.degree_bits - arity_total)) + 1 + c->n_pi;
  return sh;
}
inline void put_le(uint8_t* dst, u64 w) {
  for (int k = 0; k < 8; ++k) dst[k] = (uint8_t)(w >> (8 * k));
}
inline u64 get_le(const uint8_t* src) {
  u64 w = 0;
  for (int k = 0; k < 8; ++k) w |= (u64)src[k] << (8 * k);
  return w;
}
}  // namespace

extern "C" size_t p2mt_proof_bytes_len(const p2mt_circuit_data* c) {
  if (!c) return 0;
  try {  // (size_t result: 0 on failure; proof_shape builds a vector)
    const ProofShape sh = proof_shape(c);
    return c->proof_len * 8 + sh.n_queries * sh.per_query.size();
  } catch (...) {
    return 0;
  }
}

extern "C" int p2mt_proof_to_bytes(const p2mt_circuit_data* c, const uint64_t* proof, size_t proof_len, uint8_t* bytes_out,
                                   size_t bytes_cap) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !proof || !bytes_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (proof_len != c->proof_len) return p2mt::fail(P2MT_EINVAL, "proof_to_bytes: wrong proof length for this circuit");
  if (bytes_cap < p2mt_proof_bytes_len(c)) return p2mt::fail(P2MT_EINVAL, "proof_to_bytes: output buffer too small (p2mt_proof_bytes_len)");
  const ProofShape sh = proof_shape(c);
  const uint64_t* w = proof;
  uint8_t* b = bytes_out;
  auto words = [&](size_t n) {
    for (size_t i = 0; i < n; ++i, b += 8) put_le(b, *w++);
  };
  words(sh.head_words + sh.caps_words);
  for (size_t q = 0; q < sh.n_queries; ++q)
    for (const auto& t : sh.per_query) {
      words(t.first);
      *b++ = (uint8_t)t.second;  // MerkleProof: siblings.len() as one byte
      words(4 * t.second);
    }
  words(sh.tail_words);
  if ((size_t)(w - proof) != proof_len) return p2mt::fail(P2MT_EHIP, "proof_to_bytes: internal layout mismatch");
  return P2MT_OK;
  });
}

extern "C" int p2mt_proof_from_bytes(const p2mt_circuit_data* c, const uint8_t* bytes, size_t n_bytes, uint64_t* proof_out,
                                     size_t proof_cap) {
  return p2mt::abi_guard([&]() -> int {
  if (!c || !bytes || !proof_out) return p2mt::fail(P2MT_EINVAL, "null pointer");
  if (proof_cap < c->proof_len) return p2mt::fail(P2MT_EINVAL, "proof_from_bytes: proof buffer too small");
  if (n_bytes != p2mt_proof_bytes_len(c)) return p2mt::fail(P2MT_EINVAL, "proof_from_bytes: wrong length for this circuit");
  const ProofShape sh = proof_shape(c);
  const uint8_t* b = bytes;
  uint64_t* w = proof_out;
  auto words = [&](size_t n) {
    for (size_t i = 0; i < n; ++i, b += 8) *w++ = get_le(b);
  };
  words(sh.head_words + sh.caps_words);
  for (size_t q = 0; q < sh.n_queries; ++q)
    for (const auto& t : sh.per_query) {
      words(t.first);
      if (*b++ != (uint8_t)t.second) return p2mt::fail(P2MT_EINVAL, "proof_from_bytes: Merkle path length does not match the circuit");
      words(4 * t.second);
    }
  words(sh.tail_words);
  for (size_t i = 0; i < c->proof_len; ++i)  // plonky2's read_field rejects non-canonical encodings
    if (proof_out[i] >= gl::P) return p2mt::fail(P2MT_EINVAL, "proof_from_bytes: non-canonical field element");
  return P2MT_OK;
  });
}

// debug: timeline of the dataflow witness interpreter.  enable = 1 allocates the trace buffer (one tick per generator); after a
// prove / generate_witness, out[3 * i .. 3 * i + 3) = (kind, level, completion tick at 100 MHz) of generator i in schedule order.
extern "C" int p2mt_debug_witness_trace(p2mt_circuit_data* c, int enable, uint64_t* out, size_t cap, size_t* n_out) {
  return p2mt::abi_guard([&]() -> int {
  if (!c) return p2mt::fail(P2MT_EINVAL, "null pointer");
  const size_t n_ops = c->ops_cap;
  if (enable && !c->d_trace) {
    if (hipMalloc((void**)&c->d_trace, (n_ops + 1) * 8) != hipSuccess) return p2mt::fail(P2MT_ENOMEM, "hipMalloc(trace) failed");
    P2MT_HIP(hipMemset(c->d_trace, 0, (n_ops + 1) * 8));
  }
  if (!out) return P2MT_OK;
  if (!c->d_trace || cap < 3 * n_ops) return p2mt::fail(P2MT_EINVAL, "witness_trace: not enabled or buffer too small");
  std::vector<u64> ticks(n_ops);
  std::vector<WOp> ops(n_ops);
  std::vector<u32> lvl(2 * c->n_levels + 1);
  P2MT_HIP(hipMemcpy(ticks.data(), c->d_trace, n_ops * 8, hipMemcpyDeviceToHost));
  P2MT_HIP(hipMemcpy(ops.data(), c->d_ops, n_ops * sizeof(WOp), hipMemcpyDeviceToHost));
  P2MT_HIP(hipMemcpy(lvl.data(), c->d_lvl, lvl.size() * 4, hipMemcpyDeviceToHost));
  size_t n = 0;
  for (u32 l = 0; l < c->n_levels; ++l)
    for (u32 o = lvl[2 * l]; o < lvl[2 * l + 2]; ++o, ++n) {
      out[3 * n] = ops[o].kind & 0xFF;
      out[3 * n + 1] = l;
      out[3 * n + 2] = ticks[o];
    }
  if (n_out) *n_out = n;
  return P2MT_OK;
  });
}
