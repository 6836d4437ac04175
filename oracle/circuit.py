"""oracle/circuit.py -- TEST INFRASTRUCTURE.  [parity unpinned]

CPU restatement of the part of plonky2 (git rev 3b21b87d, NOT in /root/reference) that the reference's verifier
circuits drive: CircuitBuilder (plonk/circuit_builder.rs, gadgets/arithmetic.rs, gadgets/hash.rs, hash/poseidon.rs
permute_swapped), build() (selectors, constants, sigma polynomials, constants_sigmas commitment, circuit digest),
generate_partial_witness (iop/generator.rs), prove (plonk/prover.rs) and verify (plonk/verifier.rs), restricted to the
gate set those circuits use under CircuitConfig::standard_recursion_config(): NoopGate, ConstantGate, PublicInputGate,
ArithmeticGate, PoseidonGate.  Reference call sites: /root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91 (circuit),
:148-150 (prove / verify); /root/reference/src/mmr/common.rs:5-58 (gadgets).

Small-case host logic in plain Python (a 64-row circuit); the field-heavy steps (commit, permutation argument,
quotient, FRI, opening check) are the C restatement behind tests/oracle_lib.Oracle, passed in as `oracle`.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Known, deliberate differences from upstream (documented in DESIGN.md):
  * randomize_unused_pi_wires (RandomValueGenerator on the PublicInputGate's unused wires) is not applied: the wires stay
    zero, so witness generation is deterministic;
  * the proof-of-work witness is the smallest valid one (upstream: any, non-deterministic).
"""
import numpy as np

P = 0xFFFFFFFF00000001
NEG_ONE = P - 1
(NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON, BASE_SUM, ARITHMETIC_EXT, MUL_EXT, REDUCING, REDUCING_EXT, RANDOM_ACCESS,
 COSET_INTERPOLATION, POSEIDON_MDS) = range(13)   # == ORACLE_GATE_* in oracle.h
# Gate types the circuits instantiate under standard_recursion_config (D = 2), with the parameters that config implies
# (new_from_config of each gate: 80 routed / 135 wires / 2 constants).  degree, id and num_constants drive plonky2's gate order
# (sorted by (degree, id)) and selector groups.  The last eight are what builder.verify_proof adds
# (/root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:101-104).  [recalled from plonky2 @3b21b87: parity unpinned]
GATE_DEGREE = {NOOP: 0, CONSTANT: 1, PUBLIC_INPUT: 1, ARITHMETIC: 3, POSEIDON: 7, BASE_SUM: 2, ARITHMETIC_EXT: 3, MUL_EXT: 3,
               REDUCING: 2, REDUCING_EXT: 2, RANDOM_ACCESS: 5, COSET_INTERPOLATION: 6, POSEIDON_MDS: 1}
GATE_ID = {NOOP: "NoopGate", CONSTANT: "ConstantGate { num_consts: 2 }", PUBLIC_INPUT: "PublicInputGate",
           ARITHMETIC: "ArithmeticGate { num_ops: 20 }", POSEIDON: "PoseidonGate(PhantomData<GoldilocksField>)<WIDTH=12>",
           BASE_SUM: "BaseSumGate { num_limbs: 63 } + Base: 2", ARITHMETIC_EXT: "ArithmeticExtensionGate { num_ops: 10 }",
           MUL_EXT: "MulExtensionGate { num_ops: 13 }", REDUCING: "ReducingGate { num_coeffs: 43 }",
           REDUCING_EXT: "ReducingExtensionGate { num_coeffs: 32 }",
           RANDOM_ACCESS: "RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: PhantomData<GoldilocksField> }<D=2>",
           COSET_INTERPOLATION: "CosetInterpolationGate { subgroup_bits: 4, degree: 6, barycentric_weights: [..] }<D=2>",
           POSEIDON_MDS: "PoseidonMdsGate(PhantomData<GoldilocksField>)<WIDTH=12>"}
GATE_NUM_CONSTANTS = {NOOP: 0, CONSTANT: 2, PUBLIC_INPUT: 0, ARITHMETIC: 2, POSEIDON: 0, BASE_SUM: 0, ARITHMETIC_EXT: 2, MUL_EXT: 1,
                      REDUCING: 0, REDUCING_EXT: 0, RANDOM_ACCESS: 2, COSET_INTERPOLATION: 0, POSEIDON_MDS: 0}
GATE_NUM_CONSTRAINTS = {NOOP: 0, CONSTANT: 2, PUBLIC_INPUT: 4, ARITHMETIC: 20, POSEIDON: 123, BASE_SUM: 64, ARITHMETIC_EXT: 20,
                        MUL_EXT: 26, REDUCING: 86, REDUCING_EXT: 64, RANDOM_ACCESS: 26, COSET_INTERPOLATION: 12, POSEIDON_MDS: 24}
UNUSED_SELECTOR = 0xFFFFFFFF
BASE_SUM_LIMBS, ARITH_EXT_OPS, MUL_EXT_OPS, REDUCING_COEFFS, REDUCING_EXT_COEFFS = 63, 10, 13, 43, 32
RA_BITS, RA_COPIES, RA_EXTRA = 4, 4, 2
W_EXT = 7  # F[X] / (X^2 - 7)


# ---- quadratic extension on Python ints: (a, b) = a + b X
def e_add(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def e_sub(x, y):
    return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)


def e_mul(x, y):
    return ((x[0] * y[0] + W_EXT * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def e_scale(x, c):
    return (x[0] * c % P, x[1] * c % P)


def e_inv(x):
    # 1 / (a + bX) = (a - bX) / (a^2 - 7 b^2); inverse of 0 is 0 (plonky2's try_inverse().unwrap_or(0) is never reached here)
    n = (x[0] * x[0] - W_EXT * x[1] * x[1]) % P
    ni = pow(n, P - 2, P)
    return (x[0] * ni % P, (P - x[1]) * ni % P)


def root_of_unity(log_n):
    g = pow(7, (P - 1) >> 32, P)
    for _ in range(log_n, 32):
        g = g * g % P
    return g


def _coset_interp_tables():
    g = root_of_unity(4)
    dom = [pow(g, i, P) for i in range(16)]
    wts = []
    for i in range(16):
        p = 1
        for j in range(16):
            if j != i:
                p = p * (dom[i] - dom[j]) % P
        wts.append(pow(p, P - 2, P))
    return dom, wts


COSET_DOMAIN, COSET_WEIGHTS = _coset_interp_tables()



# ---------------------------------------------------------------------------------------------------------------------------------
# Conventions of plonky2 @3b21b87d that were restated from recall AND have a plausible alternative.  ONE switch per convention; the
# library's twin is csrc/circuit_types.h (kDigestDomainSeparator).  Flipping one is a one-line change here plus one there, then
# `python tools/gen_prove_golden.py && python tools/gen_crosscheck_vectors.py` and the whole suite.  tools/plonky2_crosscheck
# carries the circuit digest (and a full proof) under every alternative, so one cargo run tells which one real plonky2 uses.
#   digest_domain_separator -- circuit_builder.rs build(): circuit_digest = hash_no_pad(cap || D || degree_bits) with
#     "hash_pad"   D = hash_pad(domain_separator = [])   (pad10*1 to the sponge rate: hash_no_pad([1,0,0,0,0,0,0,1]))  [default]
#     "zero_hash"  D = hash_no_pad([]) = [0, 0, 0, 0]     (what rounds 1-2 of this repository assumed)
#     "none"       no D term (a revision older than the domain separator)
CONVENTIONS = {"digest_domain_separator": "hash_pad"}
DIGEST_DOMAIN_SEPARATORS = ("hash_pad", "zero_hash", "none")


def hash_pad(o, elements):
    """Hasher::hash_pad (plonk/config.rs): push 1, zeros until one slot short of a multiple of the rate, push 1, hash_no_pad"""
    padded = [int(x) for x in elements] + [1]
    while (len(padded) + 1) % 8:
        padded.append(0)
    padded.append(1)
    return o.hash_no_pad(np.array(padded, np.uint64))


def circuit_digest(o, cs_cap, degree_bits, mode=None):
    mode = mode or CONVENTIONS["digest_domain_separator"]
    parts = [np.asarray(cs_cap, np.uint64).reshape(-1)]
    if mode == "hash_pad":
        parts.append(hash_pad(o, []))
    elif mode == "zero_hash":
        parts.append(np.zeros(4, np.uint64))
    else:
        assert mode == "none", mode
    parts.append(np.array([degree_bits], np.uint64))
    return o.hash_no_pad(np.concatenate(parts))


class Config:
    """CircuitConfig::standard_recursion_config() (mmr_plonky2_verifier.rs:30)."""
    num_wires = 135
    num_routed_wires = 80
    num_constants = 2
    num_challenges = 2
    max_quotient_degree_factor = 8
    rate_bits = 3
    cap_height = 4


def wire(row, col):
    return ("w", row, col)


class CircuitBuilder:
    def __init__(self, oracle):
        self.o = oracle
        self.cfg = Config
        self.n_virtual = 0
        self.gate_instances = []      # [kind, [constants]]
        self.copy_constraints = []
        self.generators = []
        self.constants_to_targets = {}
        self.targets_to_constants = {}
        self.base_arithmetic_results = {}
        self.arithmetic_results = {}  # ExtensionArithmeticOperation -> ExtensionTarget
        self.current_slots = {}       # (gate kind, params) -> (row, next op)      [find_slot]
        self.constant_generators = [] # (row, constant index, wire): gates with spare constant wires (RandomAccessGate) come first
        self.public_inputs = []
        self.context = ""             # with_context!: label of the phase adding gates (debugging aid: row_context[row])
        self.row_context = []
        self.op_context = {}

    # ---- targets
    def add_virtual_target(self):
        self.n_virtual += 1
        return ("v", self.n_virtual - 1)

    def add_virtual_targets(self, n):
        return [self.add_virtual_target() for _ in range(n)]

    def add_virtual_hash(self):
        return self.add_virtual_targets(4)

    def add_virtual_extension_target(self):
        return tuple(self.add_virtual_targets(2))

    def add_virtual_extension_targets(self, n):
        return [self.add_virtual_extension_target() for _ in range(n)]

    def add_virtual_bool_target_unsafe(self):
        return self.add_virtual_target()

    def add_virtual_bool_target_safe(self):
        b = self.add_virtual_target()
        self.assert_bool(b)
        return b

    def constant(self, c):
        c %= P
        if c in self.constants_to_targets:
            return self.constants_to_targets[c]
        t = self.add_virtual_target()
        self.constants_to_targets[c] = t
        self.targets_to_constants[t] = c
        return t

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def two(self):
        return self.constant(2)

    def neg_one(self):
        return self.constant(NEG_ONE)

    def _false(self):
        return self.zero()

    def connect(self, x, y):
        for t in (x, y):
            assert t[0] == "v" or t[2] < self.cfg.num_routed_wires, "Tried to route a wire that isn't routable"
        self.copy_constraints.append((x, y))

    def assert_zero(self, x):
        self.connect(x, self.zero())

    def register_public_inputs(self, targets):
        self.public_inputs.extend(targets)

    def num_gates(self):
        return len(self.gate_instances)

    def add_gate(self, kind, constants=()):
        constants = list(constants) + [0] * (GATE_NUM_CONSTANTS[kind] - len(constants))
        row = len(self.gate_instances)
        self.gate_instances.append([kind, constants])
        self.row_context.append(self.context)
        # Gate::extra_constant_wires: spare routed wires that can carry a constant (consumed by build() before ConstantGates)
        if kind == RANDOM_ACCESS:
            self.constant_generators += [(row, i, 72 + i) for i in range(RA_EXTRA)]
        elif kind == CONSTANT:
            self.constant_generators += [(row, i, i) for i in range(self.cfg.num_constants)]
        # Gate::generators for the gates whose generators are per row (per-operation ones are added where the slot is taken:
        # build() removes the generators of unused slots, circuit_builder.rs "Remove unused generators, if any")
        if kind == BASE_SUM:
            self.generators.append(("base_split", row))
        elif kind == REDUCING:
            self.generators.append(("reducing", row))
        elif kind == REDUCING_EXT:
            self.generators.append(("reducing_ext", row))
        elif kind == COSET_INTERPOLATION:
            self.generators.append(("interpolation", row))
        elif kind == POSEIDON_MDS:
            self.generators.append(("poseidon_mds", row))
        return row

    def find_slot(self, kind, params, constants, num_ops):
        """circuit_builder.rs find_slot: (row, op index) of the next free operation of a multi-operation gate"""
        key = (kind, tuple(params))
        if key in self.current_slots:
            row, i = self.current_slots[key]
        else:
            row, i = self.add_gate(kind, constants), 0
        if i == num_ops - 1:
            self.current_slots.pop(key, None)
        else:
            self.current_slots[key] = (row, i + 1)
        return row, i

    # ---- gadgets/arithmetic.rs
    def arithmetic(self, const_0, const_1, m0, m1, addend):
        const_0 %= P
        const_1 %= P
        special = self._arithmetic_special_cases(const_0, const_1, m0, m1, addend)
        if special is not None:
            return special
        op = (const_0, const_1, m0, m1, addend)
        if op in self.base_arithmetic_results:
            return self.base_arithmetic_results[op]
        row, i = self.find_slot(ARITHMETIC, (const_0, const_1), [const_0, const_1], self.cfg.num_routed_wires // 4)
        self.connect(m0, wire(row, 4 * i))
        self.connect(m1, wire(row, 4 * i + 1))
        self.connect(addend, wire(row, 4 * i + 2))
        self.generators.append(("arith", row, i, const_0, const_1))
        res = wire(row, 4 * i + 3)
        self.base_arithmetic_results[op] = res
        return res

    def _arithmetic_special_cases(self, const_0, const_1, m0, m1, addend):
        zero = self.zero()
        m0c, m1c, ac = (self.targets_to_constants.get(t) for t in (m0, m1, addend))
        first_zero = const_0 == 0 or m0 == zero or m1 == zero
        second_zero = const_1 == 0 or addend == zero
        first_const = 0 if first_zero else (m0c * m1c * const_0 % P if m0c is not None and m1c is not None else None)
        second_const = 0 if second_zero else (ac * const_1 % P if ac is not None else None)
        if first_const is not None and second_const is not None:
            return self.constant((first_const + second_const) % P)
        if first_zero and const_1 == 1:
            return addend
        if second_zero:
            if m0c is not None and m0c * const_0 % P == 1:
                return m1
            if m1c is not None and m1c * const_0 % P == 1:
                return m0
        return None

    def add(self, x, y):
        return self.arithmetic(1, 1, x, self.one(), y)

    def sub(self, x, y):
        return self.arithmetic(1, NEG_ONE, x, self.one(), y)

    def mul(self, x, y):
        return self.arithmetic(1, 0, x, y, x)

    def square(self, x):
        return self.mul(x, x)

    def mul_add(self, x, y, z):
        return self.arithmetic(1, 1, x, y, z)

    def mul_sub(self, x, y, z):
        return self.arithmetic(1, NEG_ONE, x, y, z)

    def mul_const_add(self, c, x, y):
        return self.arithmetic(c, 1, self.one(), x, y)

    def mul_const(self, c, x):
        return self.mul_const_add(c, x, self.zero())

    def not_(self, b):
        return self.sub(self.one(), b)

    def or_(self, b1, b2):
        res_minus_b2 = self.arithmetic(NEG_ONE, 1, b1, b2, b1)
        return self.add(res_minus_b2, b2)

    def assert_bool(self, b):
        z = self.mul_sub(b, b, b)
        self.connect(z, self.zero())

    def is_equal(self, x, y):
        zero = self.zero()
        equal = self.add_virtual_bool_target_unsafe()
        not_equal = self.not_(equal)
        inv = self.add_virtual_target()
        self.generators.append(("equality", x, y, equal, inv))
        diff = self.sub(x, y)
        not_equal_check = self.mul(diff, inv)
        diff_normalized = self.mul(diff, equal)
        self.connect(diff_normalized, zero)
        self.connect(not_equal, not_equal_check)
        return equal

    def exp_power_of_2(self, base, power_log):
        assert power_log <= self.cfg.num_routed_wires // 4   # else plonky2 switches to ExponentiationGate (never here)
        for _ in range(power_log):
            base = self.square(base)
        return base

    def exp_from_bits_const_base(self, base, exponent_bits):
        """base^(sum bits_i 2^i) for a constant base: product = (base^(2^i) - 1) * product * bit + product per bit"""
        self.constant(base)                                   # `let base_t = self.constant(base)`: registered even when unused
        bits = list(exponent_bits)
        assert len(bits) <= self.cfg.num_routed_wires // 4    # else ExponentiationGate (never for these circuits)
        product = self.one()
        for i, bit in enumerate(bits):
            product = self.arithmetic((pow(base, 1 << i, P) - 1) % P, 1, product, bit, product)
        return product

    def le_sum(self, bits):
        bits = list(bits)
        if not bits:
            return self.zero()
        assert len(bits) - 1 <= self.cfg.num_routed_wires // 4   # else a BaseSumGate (never for these circuits)
        two = self.two()
        rev = bits[::-1]
        acc = rev[0]
        for b in rev[1:]:
            acc = self.mul_add(two, acc, b)
        return acc

    # ---- gadgets/split_base.rs, split_join.rs, range_check.rs
    def split_le(self, integer, num_bits):
        """-> num_bits BoolTargets (little-endian) of `integer`, through ceil(num_bits / 63) BaseSumGate<2> rows"""
        if num_bits == 0:
            return []
        k = -(-num_bits // BASE_SUM_LIMBS)
        rows = [self.add_gate(BASE_SUM) for _ in range(k)]
        bits = [wire(r, 1 + j) for r in rows for j in range(BASE_SUM_LIMBS)]
        for b in bits[num_bits:]:
            self.assert_zero(b)
        bits = bits[:num_bits]
        base = pow(2, BASE_SUM_LIMBS, P)
        acc = self.zero()
        for r in reversed(rows):
            acc = self.mul_const_add(base, acc, wire(r, 0))
        self.connect(acc, integer)
        self.generators.append(("wire_split", integer, tuple(rows)))
        return bits

    def low_bits(self, x, n_low_bits, num_bits):
        return self.split_le(x, num_bits)[:n_low_bits]

    def range_check(self, x, n_log):
        self.split_le(x, n_log)

    def assert_leading_zeros(self, x, leading_zeros):
        self.range_check(x, 64 - leading_zeros)

    # ---- gadgets/arithmetic_extension.rs
    def constant_extension(self, c):
        return (self.constant(c[0]), self.constant(c[1]))

    def zero_extension(self):
        return self.constant_extension((0, 0))

    def one_extension(self):
        return self.constant_extension((1, 0))

    def convert_to_ext(self, t):
        return (t, self.zero())

    def connect_extension(self, a, b):
        for x, y in zip(a, b):
            self.connect(x, y)

    def _const_ext(self, et):
        cs = [self.targets_to_constants.get(t) for t in et]
        return None if None in cs else tuple(cs)

    def arithmetic_extension(self, const_0, const_1, m0, m1, addend):
        const_0 %= P
        const_1 %= P
        m0, m1, addend = tuple(m0), tuple(m1), tuple(addend)
        special = self._arithmetic_extension_special_cases(const_0, const_1, m0, m1, addend)
        if special is not None:
            return special
        op = (const_0, const_1, m0, m1, addend)
        if op in self.arithmetic_results:
            return self.arithmetic_results[op]
        if self._const_ext(addend) == (0, 0):      # "If the addend is zero, we use a multiplication gate."
            row, i = self.find_slot(MUL_EXT, (const_0,), [const_0], MUL_EXT_OPS)
            self.connect_extension(m0, (wire(row, 6 * i), wire(row, 6 * i + 1)))
            self.connect_extension(m1, (wire(row, 6 * i + 2), wire(row, 6 * i + 3)))
            self.generators.append(("mul_ext", row, i, const_0))
            self.op_context[(MUL_EXT, row, i)] = self.context
            res = (wire(row, 6 * i + 4), wire(row, 6 * i + 5))
        else:
            row, i = self.find_slot(ARITHMETIC_EXT, (const_0, const_1), [const_0, const_1], ARITH_EXT_OPS)
            self.connect_extension(m0, (wire(row, 8 * i), wire(row, 8 * i + 1)))
            self.connect_extension(m1, (wire(row, 8 * i + 2), wire(row, 8 * i + 3)))
            self.connect_extension(addend, (wire(row, 8 * i + 4), wire(row, 8 * i + 5)))
            self.generators.append(("arith_ext", row, i, const_0, const_1))
            self.op_context[(ARITHMETIC_EXT, row, i)] = self.context
            res = (wire(row, 8 * i + 6), wire(row, 8 * i + 7))
        self.arithmetic_results[op] = res
        return res

    def _arithmetic_extension_special_cases(self, const_0, const_1, m0, m1, addend):
        zero = self.zero_extension()
        m0c, m1c, ac = self._const_ext(m0), self._const_ext(m1), self._const_ext(addend)
        first_zero = const_0 == 0 or m0 == zero or m1 == zero
        second_zero = const_1 == 0 or addend == zero
        first_const = (0, 0) if first_zero else (e_scale(e_mul(m0c, m1c), const_0) if m0c is not None and m1c is not None else None)
        second_const = (0, 0) if second_zero else (e_scale(ac, const_1) if ac is not None else None)
        if first_const is not None and second_const is not None:
            return self.constant_extension(e_add(first_const, second_const))
        if first_zero and const_1 == 1:
            return addend
        if second_zero:
            if m0c is not None and e_scale(m0c, const_0) == (1, 0):
                return m1
            if m1c is not None and e_scale(m1c, const_0) == (1, 0):
                return m0
        return None

    def add_extension(self, a, b):
        return self.arithmetic_extension(1, 1, self.one_extension(), a, b)

    def sub_extension(self, a, b):
        return self.arithmetic_extension(1, NEG_ONE, self.one_extension(), a, b)

    def mul_extension_with_const(self, const_0, m0, m1):
        return self.arithmetic_extension(const_0, 0, m0, m1, self.zero_extension())

    def mul_extension(self, a, b):
        return self.mul_extension_with_const(1, a, b)

    def square_extension(self, x):
        return self.mul_extension(x, x)

    def mul_many_extension(self, terms):
        acc = self.one_extension()
        for t in terms:
            acc = self.mul_extension(acc, t)
        return acc

    def cube_extension(self, x):
        return self.mul_many_extension([x, x, x])

    def mul_add_extension(self, a, b, c):
        return self.arithmetic_extension(1, 1, a, b, c)

    def mul_sub_extension(self, a, b, c):
        return self.arithmetic_extension(1, NEG_ONE, a, b, c)

    def scalar_mul_ext(self, a, b):
        return self.mul_extension(self.convert_to_ext(a), b)

    def mul_const_extension(self, c, x):
        return self.mul_extension(self.constant_extension((c % P, 0)), x)

    def exp_power_of_2_extension(self, base, power_log):
        for _ in range(power_log):
            base = self.square_extension(base)
        return base

    def exp_u64_extension(self, base, exponent):
        if exponent == 0:
            return self.one_extension()
        if exponent == 1:
            return base
        if exponent == 2:
            return self.square_extension(base)
        if exponent == 3:
            return self.cube_extension(base)
        current, product = base, self.one_extension()
        for j in range(exponent.bit_length()):
            if j != 0:
                current = self.square_extension(current)
            if (exponent >> j) & 1:
                product = self.mul_extension(product, current)
        return product

    def div_add_extension(self, x, y, z):
        """x / y + z: the inverse of y is a generated witness (QuotientGeneratorExtension) pinned by y * inv == 1"""
        inv = self.add_virtual_extension_target()
        one = self.one_extension()
        self.generators.append(("quotient_ext", one, tuple(y), inv))
        y_inv = self.mul_extension(y, inv)
        self.connect_extension(y_inv, one)
        return self.mul_add_extension(x, inv, z)

    def div_extension(self, x, y):
        return self.div_add_extension(x, y, self.zero_extension())

    # ---- util/reducing.rs ReducingFactorTarget: sum_i terms[i] * base^i
    def reduce_arithmetic(self, base, terms):
        acc = self.zero_extension()
        for t in reversed(terms):
            acc = self.mul_add_extension(base, acc, t)
        return acc

    def reduce_base(self, base, terms):
        """terms: base-field Targets"""
        terms = list(terms)
        if len(terms) <= ARITH_EXT_OPS + 1:    # "For small reductions, use an arithmetic gate."
            return self.reduce_arithmetic(base, [self.convert_to_ext(t) for t in terms])
        zero = self.zero()
        acc = self.zero_extension()
        rev = terms + [zero] * (-len(terms) % REDUCING_COEFFS)
        rev.reverse()
        for off in range(0, len(rev), REDUCING_COEFFS):
            row = self.add_gate(REDUCING)
            self.connect_extension(base, (wire(row, 2), wire(row, 3)))
            self.connect_extension(acc, (wire(row, 4), wire(row, 5)))
            for j, t in enumerate(rev[off:off + REDUCING_COEFFS]):
                self.connect(t, wire(row, 6 + j))
            acc = (wire(row, 0), wire(row, 1))
        return acc

    def reduce_ext(self, base, terms):
        """terms: ExtensionTargets"""
        terms = [tuple(t) for t in terms]
        if len(terms) <= ARITH_EXT_OPS + 1:
            return self.reduce_arithmetic(base, terms)
        zero_ext = self.zero_extension()
        acc = zero_ext
        rev = terms + [zero_ext] * (-len(terms) % REDUCING_EXT_COEFFS)
        rev.reverse()
        for off in range(0, len(rev), REDUCING_EXT_COEFFS):
            row = self.add_gate(REDUCING_EXT)
            self.connect_extension(base, (wire(row, 2), wire(row, 3)))
            self.connect_extension(acc, (wire(row, 4), wire(row, 5)))
            for j, t in enumerate(rev[off:off + REDUCING_EXT_COEFFS]):
                self.connect_extension(t, (wire(row, 6 + 2 * j), wire(row, 7 + 2 * j)))
            acc = (wire(row, 0), wire(row, 1))
        return acc

    def reducing_shift(self, base, count, x):
        """ReducingFactorTarget::shift: base^count * x"""
        zero_ext = self.zero_extension()
        exp = zero_ext if tuple(x) == zero_ext else self.exp_u64_extension(base, count)
        return self.mul_extension(exp, x)

    # ---- gadgets/random_access.rs
    def random_access(self, access_index, v):
        v = list(v)
        if len(v) == 1:
            return v[0]
        assert len(v) == 1 << RA_BITS   # the only vector size these circuits use (cap of 16, arity 16)
        claimed = self.add_virtual_target()
        row, copy = self.find_slot(RANDOM_ACCESS, (), [], RA_COPIES)
        base = 18 * copy
        for i, val in enumerate(v):
            self.connect(val, wire(row, base + 2 + i))
        self.connect(access_index, wire(row, base))
        self.connect(claimed, wire(row, base + 1))
        self.generators.append(("random_access", row, copy))
        return claimed

    def random_access_extension(self, access_index, v):
        return tuple(self.random_access(access_index, [et[i] for et in v]) for i in range(2))

    # ---- gadgets/interpolation.rs
    def interpolate_coset(self, coset_shift, values, evaluation_point):
        assert len(values) == 16
        row = self.add_gate(COSET_INTERPOLATION)
        self.connect(coset_shift, wire(row, 0))
        for i, v in enumerate(values):
            self.connect_extension(v, (wire(row, 1 + 2 * i), wire(row, 2 + 2 * i)))
        self.connect_extension(evaluation_point, (wire(row, 33), wire(row, 34)))
        return (wire(row, 35), wire(row, 36))

    # ---- gadgets/hash.rs, hash/hashing.rs, hash/poseidon.rs (AlgebraicHasher::permute_swapped)
    def permute_swapped(self, state, swap):
        row = self.add_gate(POSEIDON)
        self.connect(swap, wire(row, 24))
        for i in range(12):
            self.connect(state[i], wire(row, i))
        self.generators.append(("poseidon", row))
        return [wire(row, 12 + i) for i in range(12)]

    def permute(self, state):
        return self.permute_swapped(state, self._false())

    def hash_n_to_hash_no_pad(self, inputs):
        zero = self.zero()
        state = [zero] * 12
        for off in range(0, len(inputs), 8):
            chunk = inputs[off:off + 8]
            state[:len(chunk)] = chunk
            state = self.permute(state)
        return state[:4]

    def hash_or_noop(self, inputs):
        zero = self.zero()
        if len(inputs) <= 4:
            return list(inputs) + [zero] * (4 - len(inputs))
        return self.hash_n_to_hash_no_pad(inputs)

    # ---- build
    def build(self):
        cfg = self.cfg
        pi_hash_t = self.hash_n_to_hash_no_pad(list(self.public_inputs))
        pi_gate = self.add_gate(PUBLIC_INPUT)
        for i, h in enumerate(pi_hash_t):
            self.connect(h, wire(pi_gate, i))
        # constants: one constant generator per distinct constant, in increasing canonical order; generators of gates with spare
        # constant wires first (in row order), ConstantGates added as needed
        consts = sorted(self.constants_to_targets.items())
        while len(consts) > len(self.constant_generators):
            self.add_gate(CONSTANT)
        for (c, t), (row, ci, wi) in zip(consts, self.constant_generators):
            self.gate_instances[row][1][ci] = c
            self.connect(wire(row, wi), t)
            self.generators.append(("const", row, wi, c))
        while len(self.gate_instances) & (len(self.gate_instances) - 1) or len(self.gate_instances) < 2:
            self.add_gate(NOOP)
        return CircuitData(self)


class _Forest:
    def __init__(self, n):
        self.parent = list(range(n))

    def find(self, x):
        while self.parent[x] != x:
            self.parent[x] = self.parent[self.parent[x]]
            x = self.parent[x]
        return x

    def merge(self, a, b):
        a, b = self.find(a), self.find(b)
        if a != b:
            self.parent[b] = a


class CircuitData:
    def __init__(self, b):
        o, cfg = b.o, b.cfg
        self.o, self.cfg = o, cfg
        self.gate_instances = b.gate_instances
        self.generators = b.generators
        self.row_context = b.row_context
        self.op_context = b.op_context
        self.public_inputs = list(b.public_inputs)
        self.n_virtual = b.n_virtual
        n = self.degree = len(b.gate_instances)
        self.degree_bits = n.bit_length() - 1
        # gate types sorted by (degree, id); selector groups (gates/selectors.rs selector_polynomials)
        kinds = sorted({g[0] for g in b.gate_instances}, key=lambda k: (GATE_DEGREE[k], GATE_ID[k]))
        self.gates = kinds
        max_degree = cfg.max_quotient_degree_factor + 1
        if GATE_DEGREE[kinds[-1]] + len(kinds) - 1 <= max_degree:
            groups = [(0, len(kinds))]
        else:
            groups, start = [], 0
            while start < len(kinds):
                size = 0
                while start + size < len(kinds) and size + GATE_DEGREE[kinds[start + size]] < max_degree:
                    size += 1
                groups.append((start, start + size))
                start += size
        self.groups = groups
        group_of = [next(j for j, (s, e) in enumerate(groups) if s <= i < e) for i in range(len(kinds))]
        self.selector_indices = group_of
        sel = np.zeros((len(groups), n), np.uint64)
        for j, (kind, _) in enumerate(b.gate_instances):
            i = kinds.index(kind)
            for g in range(len(groups)):
                sel[g, j] = i if g == group_of[i] else UNUSED_SELECTOR
        consts = np.zeros((cfg.num_constants, n), np.uint64)
        for j, (_, cs) in enumerate(b.gate_instances):
            for i, c in enumerate(cs):
                consts[i, j] = c
        self.num_selectors = len(groups)
        # copy constraints -> sigma (plonk/permutation_argument.rs)
        self.k_is = np.array([pow(7, j, P) for j in range(cfg.num_routed_wires)], np.uint64)
        self.forest = f = _Forest(n * cfg.num_wires + b.n_virtual)
        for x, y in b.copy_constraints:
            f.merge(self._tidx(x), self._tidx(y))
        partition = {}
        for row in range(n):
            for col in range(cfg.num_routed_wires):
                partition.setdefault(f.find(row * cfg.num_wires + col), []).append((row, col))
        neighbors = {}
        for subset in partition.values():
            for k, w in enumerate(subset):
                neighbors[w] = subset[(k + 1) % len(subset)]
        g = o.root_of_unity(self.degree_bits)
        subgroup = [1]
        for _ in range(n - 1):
            subgroup.append(subgroup[-1] * g % P)
        sigmas = np.zeros((cfg.num_routed_wires, n), np.uint64)
        for col in range(cfg.num_routed_wires):
            for row in range(n):
                nr, nc = neighbors[(row, col)]
                sigmas[col, row] = int(self.k_is[nc]) * subgroup[nr] % P
        self.sigmas = sigmas
        self.constants_sigmas = np.concatenate([sel, consts, sigmas])
        self.cs_coeffs = o.ifft_rows(self.constants_sigmas)
        self.cs_leaves, self.cs_digests, self.cs_cap = o.polynomial_batch_commit(self.constants_sigmas, True,
                                                                                cfg.rate_bits, cfg.cap_height)
        self.circuit_digest = circuit_digest(o, self.cs_cap, self.degree_bits)
        self.fri_params = o.fri_params_standard(self.degree_bits)
        self.desc = o.plonk_desc(self.degree_bits, cfg.num_wires, cfg.num_routed_wires, cfg.num_constants,
                                 self.num_selectors, cfg.num_challenges, cfg.max_quotient_degree_factor, kinds,
                                 [group_of[i] for i in range(len(kinds))], [groups[group_of[i]] for i in range(len(kinds))])
        self.num_partial_products = -(-cfg.num_routed_wires // cfg.max_quotient_degree_factor) - 1

    def _tidx(self, t):
        if t[0] == "w":
            return t[1] * self.cfg.num_wires + t[2]
        return self.degree * self.cfg.num_wires + t[1]

    # ---- iop/generator.rs generate_partial_witness
    def _gen_io(self, gen):
        """-> (dependencies, kind-specific runner) of one generator; targets only (values come from the partition witness)"""
        k = gen[0]
        W = wire
        if k == "const":
            return [], None
        if k == "arith":
            _, row, i = gen[:3]
            return [W(row, 4 * i + j) for j in range(3)], None
        if k == "equality":
            return [gen[1], gen[2]], None
        if k == "poseidon":
            row = gen[1]
            return [W(row, i) for i in range(12)] + [W(row, 24)], None
        if k == "arith_ext":
            _, row, i = gen[:3]
            return [W(row, 8 * i + j) for j in range(6)], None
        if k == "mul_ext":
            _, row, i = gen[:3]
            return [W(row, 6 * i + j) for j in range(4)], None
        if k == "quotient_ext":
            return list(gen[1]) + list(gen[2]), None
        if k == "reducing":
            row = gen[1]
            return [W(row, j) for j in range(2, 6 + REDUCING_COEFFS)], None
        if k == "reducing_ext":
            row = gen[1]
            return [W(row, j) for j in range(2, 6 + 2 * REDUCING_EXT_COEFFS)], None
        if k == "wire_split":
            return [gen[1]], None
        if k == "base_split":
            return [W(gen[1], 0)], None
        if k == "random_access":
            _, row, copy = gen
            return [W(row, 18 * copy)] + [W(row, 18 * copy + 2 + i) for i in range(16)], None
        if k == "interpolation":
            row = gen[1]
            return [W(row, j) for j in range(0, 35)], None
        if k == "poseidon_mds":
            row = gen[1]
            return [W(row, j) for j in range(24)], None
        raise ValueError(k)

    def _run_generator(self, gen, getv, setv):
        k = gen[0]
        W = wire
        if k == "const":
            setv(W(gen[1], gen[2]), gen[3])
        elif k == "arith":
            _, row, i, c0, c1 = gen
            m0, m1, ad = (getv(W(row, 4 * i + j)) for j in range(3))
            setv(W(row, 4 * i + 3), (m0 * m1 % P * c0 + ad * c1) % P)
        elif k == "equality":
            _, x, y, equal, inv = gen
            xv, yv = getv(x), getv(y)
            setv(equal, int(xv == yv))
            setv(inv, pow((xv - yv) % P, P - 2, P) if xv != yv else 0)
        elif k == "poseidon":
            row = gen[1]
            ins = [getv(W(row, i)) for i in range(12)]
            w = self.o.poseidon_gate_witness(np.array(ins, np.uint64), int(getv(W(row, 24))))
            for col in range(12, self.cfg.num_wires):
                if col != 24:
                    setv(W(row, col), int(w[col]))
        elif k == "arith_ext":
            _, row, i, c0, c1 = gen
            v = [getv(W(row, 8 * i + j)) for j in range(6)]
            r = e_add(e_scale(e_mul(v[0:2], v[2:4]), c0), e_scale(v[4:6], c1))
            setv(W(row, 8 * i + 6), r[0])
            setv(W(row, 8 * i + 7), r[1])
        elif k == "mul_ext":
            _, row, i, c0 = gen
            v = [getv(W(row, 6 * i + j)) for j in range(4)]
            r = e_scale(e_mul(v[0:2], v[2:4]), c0)
            setv(W(row, 6 * i + 4), r[0])
            setv(W(row, 6 * i + 5), r[1])
        elif k == "quotient_ext":
            _, num, den, quo = gen
            r = e_mul([getv(t) for t in num], e_inv([getv(t) for t in den]))
            setv(quo[0], r[0])
            setv(quo[1], r[1])
        elif k in ("reducing", "reducing_ext"):
            row = gen[1]
            alpha = (getv(W(row, 2)), getv(W(row, 3)))
            acc = (getv(W(row, 4)), getv(W(row, 5)))
            ext = k == "reducing_ext"
            n = REDUCING_EXT_COEFFS if ext else REDUCING_COEFFS
            start_accs = 6 + (2 * n if ext else n)
            for i in range(n):
                coeff = (getv(W(row, 6 + 2 * i)), getv(W(row, 7 + 2 * i))) if ext else (getv(W(row, 6 + i)), 0)
                acc = e_add(e_mul(acc, alpha), coeff)
                at = 0 if i == n - 1 else start_accs + 2 * i
                setv(W(row, at), acc[0])
                setv(W(row, at + 1), acc[1])
        elif k == "wire_split":
            _, integer, rows = gen
            v = getv(integer)
            for r in rows:
                setv(W(r, 0), v & ((1 << BASE_SUM_LIMBS) - 1))
                v >>= BASE_SUM_LIMBS
            assert v == 0, "Integer too large to fit in the BaseSumGates"
        elif k == "base_split":
            row = gen[1]
            v = getv(W(row, 0))
            assert v >> BASE_SUM_LIMBS == 0, "Integer too large to fit in given number of limbs"
            for j in range(BASE_SUM_LIMBS):
                setv(W(row, 1 + j), (v >> j) & 1)
        elif k == "random_access":
            _, row, copy = gen
            idx = getv(W(row, 18 * copy))
            assert idx < 16, "Access index is larger than the vector size"
            setv(W(row, 18 * copy + 1), getv(W(row, 18 * copy + 2 + idx)))
            for j in range(RA_BITS):
                setv(W(row, 74 + RA_BITS * copy + j), (idx >> j) & 1)
        elif k == "interpolation":
            row = gen[1]
            shift = getv(W(row, 0))
            vals = [(getv(W(row, 1 + 2 * i)), getv(W(row, 2 + 2 * i))) for i in range(16)]
            point = (getv(W(row, 33)), getv(W(row, 34)))
            x = e_scale(point, pow(shift, P - 2, P))        # shifted_evaluation_point = evaluation_point / shift
            setv(W(row, 45), x[0])
            setv(W(row, 46), x[1])

            def partial(lo, hi, ev, pr):
                for i in range(lo, hi):
                    term = e_sub(x, (COSET_DOMAIN[i], 0))
                    ev = e_add(e_mul(ev, term), e_mul(e_scale(vals[i], COSET_WEIGHTS[i]), pr))
                    pr = e_mul(pr, term)
                return ev, pr

            ev, pr = partial(0, 6, (0, 0), (1, 0))
            for i in range(2):
                setv(W(row, 37 + 2 * i), ev[0])
                setv(W(row, 38 + 2 * i), ev[1])
                setv(W(row, 41 + 2 * i), pr[0])
                setv(W(row, 42 + 2 * i), pr[1])
                start = 1 + 5 * (i + 1)
                ev, pr = partial(start, min(start + 5, 16), ev, pr)
            setv(W(row, 35), ev[0])
            setv(W(row, 36), ev[1])
        elif k == "poseidon_mds":
            row = gen[1]
            st = [(getv(W(row, 2 * i)), getv(W(row, 2 * i + 1))) for i in range(12)]
            circ = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
            for r in range(12):
                acc = e_scale(st[r], 8 if r == 0 else 0)
                for i in range(12):
                    acc = e_add(acc, e_scale(st[(i + r) % 12], circ[i]))
                setv(W(row, 24 + 2 * r), acc[0])
                setv(W(row, 25 + 2 * r), acc[1])
        else:
            raise ValueError(k)

    def generate_witness(self, inputs, conflicts=None):
        """inputs: {target: value}.  conflicts (debugging): a list that receives (target, old, new) instead of raising.  -> (wires (num_wires, n), value-of-target function).  A generator runs once all the targets
        it watches are set (generate_partial_witness's watch lists); a target set twice with different values is plonky2's
        "Partition containing ... was set twice with different values" panic."""
        f, vals = self.forest, {}
        watchers, missing, ready = {}, [], []
        find = f.find
        tidx = self._tidx

        def setv(t, v):
            r = find(tidx(t))
            v %= P
            if r in vals:
                if vals[r] != v:
                    if conflicts is None:
                        raise ValueError("Partition containing %r was set twice with different values" % (t,))
                    conflicts.append((t, vals[r], v, getattr(self, "_current_generator", None)))
                return
            vals[r] = v
            for gi in watchers.pop(r, ()):
                missing[gi] -= 1
                if missing[gi] == 0:
                    ready.append(gi)

        def getv(t):
            return vals.get(find(tidx(t)))

        for gi, gen in enumerate(self.generators):
            deps = {find(tidx(t)) for t in self._gen_io(gen)[0]}
            missing.append(len(deps))
            for r in deps:
                watchers.setdefault(r, []).append(gi)
            if not deps:
                ready.append(gi)
        for t, v in inputs.items():
            setv(t, int(v))
        ran = 0
        while ready:
            gi = ready.pop()
            self._current_generator = self.generators[gi]
            self._run_generator(self.generators[gi], getv, setv)
            ran += 1
        if ran != len(self.generators):
            raise ValueError("%d generators weren't run" % (len(self.generators) - ran))
        n, nw = self.degree, self.cfg.num_wires
        wires = np.zeros((nw, n), np.uint64)
        for row in range(n):
            for col in range(nw):
                v = vals.get(find(row * nw + col))
                if v is not None:
                    wires[col, row] = v
        return wires, getv

    def _fri_batches(self, zeta):
        g = self.o.root_of_unity(self.degree_bits)
        gz = np.array([self.o.mul(int(zeta[0]), g), self.o.mul(int(zeta[1]), g)], np.uint64)
        n_cs, nw = self.constants_sigmas.shape[0], self.cfg.num_wires
        n_zs = self.cfg.num_challenges * (1 + self.num_partial_products)
        n_q = self.cfg.num_challenges * self.cfg.max_quotient_degree_factor
        all_polys = [(oi, pi) for oi, k in enumerate((n_cs, nw, n_zs, n_q)) for pi in range(k)]
        return [(zeta, all_polys), (gz, [(2, c) for c in range(self.cfg.num_challenges)])], (n_cs, nw, n_zs, n_q)

    # ---- plonk/prover.rs prove
    def prove(self, inputs, trace=None, wires_hook=None):
        """-> proof words: wires_cap | zs_partial_products_cap | quotient_polys_cap | OpeningSet | FriProof | public_inputs.
        wires_hook (tests only): edits the witness matrix after generation, to show that a witness violating the
        constraints does not yield an accepted proof."""
        o, cfg = self.o, self.cfg
        nch = cfg.num_challenges
        wires, getv = self.generate_witness(inputs)
        if wires_hook is not None:
            wires_hook(wires)
        public_inputs = np.array([getv(t) for t in self.public_inputs], np.uint64)
        pi_hash = o.hash_no_pad(public_inputs)
        w_leaves, w_dig, w_cap = o.polynomial_batch_commit(wires, True, cfg.rate_bits, cfg.cap_height)
        ch = o.challenger()
        ch.observe(self.circuit_digest)
        ch.observe(pi_hash)
        ch.observe(w_cap.reshape(-1))
        betas = ch.get_n_challenges(nch)
        gammas = ch.get_n_challenges(nch)
        zs, pps = o.permutation_partial_products(wires[:cfg.num_routed_wires], self.sigmas, self.k_is, betas, gammas,
                                                 cfg.max_quotient_degree_factor)
        zs_pp = np.concatenate([zs, pps.reshape(-1, self.degree)])
        z_leaves, z_dig, z_cap = o.polynomial_batch_commit(zs_pp, True, cfg.rate_bits, cfg.cap_height)
        ch.observe(z_cap.reshape(-1))
        alphas = ch.get_n_challenges(nch)
        quotient = o.plonk_quotient_polys(self.desc, self.k_is, self.cs_leaves, w_leaves, z_leaves, pi_hash, betas, gammas,
                                          alphas)
        chunks = quotient.reshape(nch * cfg.max_quotient_degree_factor, self.degree)
        q_leaves, q_dig, q_cap = o.polynomial_batch_commit(chunks, False, cfg.rate_bits, cfg.cap_height)
        ch.observe(q_cap.reshape(-1))
        zeta = ch.get_n_challenges(2)
        batches, _ = self._fri_batches(zeta)
        coeffs = [self.cs_coeffs, o.ifft_rows(wires), o.ifft_rows(zs_pp), chunks]
        ev = [np.concatenate([o.eval_polys_ext(c, pt) for c in coeffs]) if bi == 0 else
              o.eval_polys_ext(coeffs[2][:nch], pt) for bi, (pt, _) in enumerate(batches)]
        for e in ev:
            ch.observe(e.reshape(-1))
        oracles = [(coeffs[0], self.cs_leaves, self.cs_digests), (coeffs[1], w_leaves, w_dig), (coeffs[2], z_leaves, z_dig),
                   (coeffs[3], q_leaves, q_dig)]
        fri = o.fri_prove(oracles, batches, self.fri_params, ch)
        n_cs = self.constants_sigmas.shape[0]
        at = np.cumsum([0, n_cs, cfg.num_wires, nch, self.num_partial_products * nch, nch * cfg.max_quotient_degree_factor])
        z = ev[0]
        opening_set = np.concatenate([z[at[0]:at[1]], z[at[1]:at[2]], z[at[2]:at[3]], ev[1], z[at[3]:at[4]], z[at[4]:at[5]]])
        if trace is not None:
            trace.update(wires=wires, public_inputs=public_inputs, pi_hash=pi_hash, betas=betas, gammas=gammas, alphas=alphas,
                         zs_pp=zs_pp, quotient_chunks=chunks, zeta=zeta, caps=(w_cap, z_cap, q_cap), fri=fri)
        return np.concatenate([w_cap.reshape(-1), z_cap.reshape(-1), q_cap.reshape(-1), opening_set.reshape(-1), fri,
                               public_inputs])

    def proof_len(self):
        cfg, nch = self.cfg, self.cfg.num_challenges
        _, counts = self._fri_batches(np.zeros(2, np.uint64))
        n_open = sum(counts) + nch
        return (3 * (4 << cfg.cap_height) + 2 * n_open + self.o.fri_proof_len(self.fri_params, list(counts))
                + len(self.public_inputs))

    # ---- plonk/verifier.rs verify
    def verify(self, proof):
        """-> (accepted, reason): reason 0 ok, 10 malformed, 11 opening check failed, 12 zeta in the subgroup, else the FRI reason"""
        o, cfg = self.o, self.cfg
        nch = cfg.num_challenges
        proof = np.ascontiguousarray(np.asarray(proof, np.uint64))
        if proof.size != self.proof_len() or bool((proof >= np.uint64(P)).any()):
            return False, 10
        capw = 4 << cfg.cap_height
        w_cap, z_cap, q_cap = (proof[i * capw:(i + 1) * capw] for i in range(3))
        batches0, counts = self._fri_batches(np.zeros(2, np.uint64))
        n_cs, nw, n_zs, n_q = counts
        off = 3 * capw
        sizes = [n_cs, nw, nch, nch, self.num_partial_products * nch, n_q]
        parts = []
        for s in sizes:
            parts.append(proof[off:off + 2 * s].reshape(s, 2))
            off += 2 * s
        cs_open, w_open, zs_open, zs_next_open, pp_open, q_open = parts
        fri_len = o.fri_proof_len(self.fri_params, list(counts))
        fri = proof[off:off + fri_len]
        public_inputs = proof[off + fri_len:]
        pi_hash = o.hash_no_pad(public_inputs)
        ch = o.challenger()
        ch.observe(self.circuit_digest)
        ch.observe(pi_hash)
        ch.observe(w_cap)
        betas = ch.get_n_challenges(nch)
        gammas = ch.get_n_challenges(nch)
        ch.observe(z_cap)
        alphas = ch.get_n_challenges(nch)
        ch.observe(q_cap)
        zeta = ch.get_n_challenges(2)
        ns = self.num_selectors + cfg.num_constants
        if not o.plonk_check_openings(self.desc, self.k_is, zeta, cs_open[:ns], cs_open[ns:], w_open, zs_open, zs_next_open,
                                      pp_open, q_open, pi_hash, betas, gammas, alphas):
            return False, 11
        batches, _ = self._fri_batches(zeta)
        openings = [np.concatenate([cs_open, w_open, zs_open, pp_open, q_open]), zs_next_open]
        for e in openings:
            ch.observe(e.reshape(-1))
        caps = np.concatenate([self.cs_cap.reshape(-1), w_cap, z_cap, q_cap])
        ok, reason = o.fri_verify(list(counts), caps, batches, openings, self.fri_params, ch, fri)
        return ok, reason


# ---- /root/reference/src/mmr/common.rs:5-58
def equal(builder, first, second):
    e = [builder.is_equal(first[i], second[i]) for i in range(4)]
    return builder.or_(builder.or_(e[0], e[1]), builder.or_(e[2], e[3]))


def or_list(builder, ins):
    assert len(ins) > 0
    if len(ins) == 1:
        return ins[0]
    if len(ins) == 2:
        return builder.or_(ins[0], ins[1])
    pairs = [builder.or_(ins[i], ins[i + 1]) if i + 1 < len(ins) else ins[i] for i in range(0, len(ins), 2)]
    return or_list(builder, pairs)


def pick_hash(builder, option1, option2, pick_left):
    opposite = builder.not_(pick_left)
    t = [builder.mul(option2[i], opposite) for i in range(4)]
    return [builder.mul_add(option1[i], pick_left, t[i]) for i in range(4)]


# ---- /root/reference/src/mmr/mmr_plonky2_verifier.rs:13-91
def verify_mmr_proof_circuit(oracle, nr_merkle_proof_elms, nr_peaks):
    """-> (circuit_data, leaf target, [(hash target, bool target)], [peak hash targets])"""
    builder = CircuitBuilder(oracle)
    proof_targets, peak_targets = [], []
    leaf_to_prove = builder.add_virtual_target()
    next_hash = builder.hash_or_noop([leaf_to_prove])
    for _ in range(nr_merkle_proof_elms):
        elm = builder.add_virtual_hash()
        on_left = builder.add_virtual_bool_target_safe()
        proof_targets.append((elm, on_left))
        option1 = builder.hash_or_noop(elm + next_hash)
        option2 = builder.hash_or_noop(next_hash + elm)
        next_hash = pick_hash(builder, option1, option2, on_left)
    peaks, equals = [], []
    for _ in range(nr_peaks):
        peak = builder.add_virtual_hash()
        peaks.append(peak)
        peak_targets.append(peak)
        equals.append(equal(builder, peak, next_hash))
    hash_in_peaks = or_list(builder, equals)
    builder.connect(builder.one(), hash_in_peaks)
    if len(peaks) > 1:
        root = builder.hash_n_to_hash_no_pad([e for p in peaks for e in p])
        builder.register_public_inputs(root)
    else:
        builder.register_public_inputs(peaks[0])
    return builder.build(), leaf_to_prove, proof_targets, peak_targets


# ---- /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:20-75 (the inner circuit of config 4)
def verify_inner_merkle_proof_circuit(oracle, nr_merkle_proof_elms, nr_peaks):
    """-> (circuit_data, leaf target, [(hash target, bool target)]); public inputs = the peaks (4 per peak)"""
    builder = CircuitBuilder(oracle)
    proof_targets = []
    leaf_to_prove = builder.add_virtual_target()
    next_hash = builder.hash_or_noop([leaf_to_prove])
    for _ in range(nr_merkle_proof_elms):
        elm = builder.add_virtual_hash()
        on_left = builder.add_virtual_bool_target_safe()
        proof_targets.append((elm, on_left))
        option1 = builder.hash_or_noop(elm + next_hash)
        option2 = builder.hash_or_noop(next_hash + elm)
        next_hash = pick_hash(builder, option1, option2, on_left)
    equals = []
    for _ in range(nr_peaks):
        peak = builder.add_virtual_hash()
        builder.register_public_inputs(peak)
        equals.append(equal(builder, peak, next_hash))
    hash_in_peaks = or_list(builder, equals)
    builder.connect(builder.one(), hash_in_peaks)
    return builder.build(), leaf_to_prove, proof_targets
