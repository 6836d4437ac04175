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
NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON = range(5)
GATE_DEGREE = {NOOP: 0, CONSTANT: 1, PUBLIC_INPUT: 1, ARITHMETIC: 3, POSEIDON: 7}
GATE_ID = {NOOP: "NoopGate", CONSTANT: "ConstantGate { num_consts: 2 }", PUBLIC_INPUT: "PublicInputGate",
           ARITHMETIC: "ArithmeticGate { num_ops: 20 }", POSEIDON: "PoseidonGate(PhantomData<GoldilocksField>)<WIDTH=12>"}
GATE_NUM_CONSTANTS = {NOOP: 0, CONSTANT: 2, PUBLIC_INPUT: 0, ARITHMETIC: 2, POSEIDON: 0}
UNUSED_SELECTOR = 0xFFFFFFFF


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
        self.current_slots = {}       # (c0, c1) -> (row, next op)
        self.public_inputs = []

    # ---- targets
    def add_virtual_target(self):
        self.n_virtual += 1
        return ("v", self.n_virtual - 1)

    def add_virtual_targets(self, n):
        return [self.add_virtual_target() for _ in range(n)]

    def add_virtual_hash(self):
        return self.add_virtual_targets(4)

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

    def connect(self, x, y):
        for t in (x, y):
            assert t[0] == "v" or t[2] < self.cfg.num_routed_wires, "Tried to route a wire that isn't routable"
        self.copy_constraints.append((x, y))

    def register_public_inputs(self, targets):
        self.public_inputs.extend(targets)

    def add_gate(self, kind, constants=()):
        constants = list(constants) + [0] * (GATE_NUM_CONSTANTS[kind] - len(constants))
        self.gate_instances.append([kind, constants])
        return len(self.gate_instances) - 1

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
        key = (const_0, const_1)
        if key in self.current_slots:
            row, i = self.current_slots[key]
        else:
            row, i = self.add_gate(ARITHMETIC, [const_0, const_1]), 0
        if i == self.cfg.num_routed_wires // 4 - 1:
            self.current_slots.pop(key, None)
        else:
            self.current_slots[key] = (row, i + 1)
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

    def mul_add(self, x, y, z):
        return self.arithmetic(1, 1, x, y, z)

    def mul_sub(self, x, y, z):
        return self.arithmetic(1, NEG_ONE, x, y, z)

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

    # ---- gadgets/hash.rs, hash/hashing.rs, hash/poseidon.rs (AlgebraicHasher::permute_swapped)
    def permute(self, state):
        row = self.add_gate(POSEIDON)
        self.connect(self.zero(), wire(row, 24))  # swap = _false()
        for i in range(12):
            self.connect(state[i], wire(row, i))
        self.generators.append(("poseidon", row))
        return [wire(row, 12 + i) for i in range(12)]

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
        # constants: one ConstantGate slot per distinct constant, in increasing canonical order
        consts = sorted(self.constants_to_targets.items())
        slots = []
        while len(slots) < len(consts):
            row = self.add_gate(CONSTANT)
            slots += [(row, i) for i in range(cfg.num_constants)]
        for (c, t), (row, i) in zip(consts, slots):
            self.gate_instances[row][1][i] = c
            self.connect(wire(row, i), t)
            self.generators.append(("const", row, i, c))
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
        # circuit_digest = hash_no_pad(cap || hash_no_pad(domain_separator = []) || degree_bits)
        self.circuit_digest = o.hash_no_pad(np.concatenate([self.cs_cap.reshape(-1), np.zeros(4, np.uint64),
                                                            np.array([self.degree_bits], np.uint64)]))
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
    def generate_witness(self, inputs):
        """inputs: {target: value}.  -> (wires (num_wires, n), value-of-target function)"""
        f, vals = self.forest, {}

        def setv(t, v):
            r = f.find(self._tidx(t))
            v %= P
            if r in vals and vals[r] != v:
                raise ValueError("Partition containing %r was set twice with different values" % (t,))
            vals[r] = v

        def getv(t):
            return vals.get(f.find(self._tidx(t)))

        for t, v in inputs.items():
            setv(t, int(v))
        pending = list(self.generators)
        while pending:
            rest = []
            for gen in pending:
                if gen[0] == "const":
                    setv(wire(gen[1], gen[2]), gen[3])
                elif gen[0] == "arith":
                    _, row, i, c0, c1 = gen
                    m0, m1, ad = (getv(wire(row, 4 * i + k)) for k in range(3))
                    if None in (m0, m1, ad):
                        rest.append(gen)
                        continue
                    setv(wire(row, 4 * i + 3), (m0 * m1 % P * c0 + ad * c1) % P)
                elif gen[0] == "equality":
                    _, x, y, equal, inv = gen
                    xv, yv = getv(x), getv(y)
                    if xv is None or yv is None:
                        rest.append(gen)
                        continue
                    setv(equal, int(xv == yv))
                    setv(inv, pow((xv - yv) % P, P - 2, P) if xv != yv else 0)
                elif gen[0] == "poseidon":
                    row = gen[1]
                    ins = [getv(wire(row, i)) for i in range(12)]
                    swap = getv(wire(row, 24))
                    if None in ins or swap is None:
                        rest.append(gen)
                        continue
                    w = self.o.poseidon_gate_witness(np.array(ins, np.uint64), int(swap))
                    for col in range(12, self.cfg.num_wires):
                        setv(wire(row, col), int(w[col]))
            if len(rest) == len(pending):
                raise ValueError("%d generators weren't run" % len(rest))
            pending = rest
        n, nw = self.degree, self.cfg.num_wires
        wires = np.zeros((nw, n), np.uint64)
        for row in range(n):
            for col in range(nw):
                v = vals.get(f.find(row * nw + col))
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
