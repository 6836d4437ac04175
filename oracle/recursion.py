"""oracle/recursion.py -- TEST INFRASTRUCTURE.  [parity unpinned]

CPU restatement of plonky2's in-circuit verifier (git rev 3b21b87d, NOT in /root/reference): what
`builder.add_virtual_proof_with_pis`, `builder.add_virtual_verifier_data`, `builder.verify_proof::<PoseidonGoldilocksConfig>`,
`pw.set_proof_with_pis_target` and `pw.set_verifier_data_target` do at their call sites in
/root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:95-104 and :201-202, restated from the published algorithm of
plonk/circuit_builder.rs, recursion/recursive_verifier.rs, plonk/get_challenges.rs (get_challenges for targets),
iop/challenger.rs (RecursiveChallenger), plonk/vanishing_poly.rs (eval_vanishing_poly_circuit, evaluate_gate_constraints_circuit),
plonk/plonk_common.rs (eval_l_0_circuit, check_partial_products_circuit), gates/*.rs eval_unfiltered_circuit,
hash/poseidon.rs (constant_layer_circuit / sbox_monomial_circuit / mds_layer_circuit through the PoseidonMdsGate),
fri/recursive_verifier.rs (verify_fri_proof, fri_verifier_query_round, fri_combine_initial, compute_evaluation,
PrecomputedReducedOpeningsTarget), hash/merkle_proofs.rs (verify_merkle_proof_to_cap_with_cap_index), util/reducing.rs, and the
outer circuit of the reference itself (`complete_verification_circuit_with_inner_proof`, :84-140, quirk Q4 included).
Host logic in plain Python on top of oracle/circuit.py's CircuitBuilder.  Only tests/, smoke() and bench.py's cpu_baseline leg
may import this module.

A ProofWithPublicInputsTarget is kept FLAT: one target per proof word, in the word order of CircuitData.prove
(wires_cap | zs_partial_products_cap | quotient_polys_cap | OpeningSet | FriProof | public_inputs), so that
set_proof_with_pis_target is `target[i] <- word[i]`.
"""
from . import circuit as OC
from .circuit import (P, NEG_ONE, NOOP, CONSTANT, PUBLIC_INPUT, ARITHMETIC, POSEIDON, GATE_NUM_CONSTRAINTS, UNUSED_SELECTOR,
                      root_of_unity)

POSEIDON_RC = None  # 360 round constants, taken from the oracle library on first use


class CommonData:
    """CommonCircuitData of an inner circuit: everything verify_proof reads from `inner_circuit_data.common`"""

    def __init__(self, cd):
        cfg = cd.cfg
        self.cfg = cfg
        self.degree_bits = cd.degree_bits
        self.gates = list(cd.gates)                      # kinds in plonky2's sorted order
        self.selector_indices = list(cd.selector_indices)
        self.groups = list(cd.groups)
        self.num_selectors = cd.num_selectors
        self.num_constants = cd.num_selectors + cfg.num_constants
        self.num_public_inputs = len(cd.public_inputs)
        self.k_is = [int(x) for x in cd.k_is]
        self.quotient_degree_factor = cfg.max_quotient_degree_factor
        self.num_partial_products = cd.num_partial_products
        self.num_gate_constraints = max(GATE_NUM_CONSTRAINTS[g] for g in self.gates)
        fp = cd.fri_params
        self.rate_bits, self.cap_height = fp.rate_bits, fp.cap_height
        self.proof_of_work_bits, self.num_query_rounds = fp.proof_of_work_bits, fp.num_query_rounds
        self.reduction_arity_bits = [fp.reduction_arity_bits[i] for i in range(fp.num_reductions)]
        self.lde_bits = self.degree_bits + self.rate_bits
        self.final_poly_len = 1 << (self.degree_bits - sum(self.reduction_arity_bits))
        nch = cfg.num_challenges
        self.oracle_widths = [self.num_constants + cfg.num_routed_wires, cfg.num_wires, nch * (1 + self.num_partial_products),
                              nch * self.quotient_degree_factor]

    def proof_len(self):
        return self.layout()["end"]

    def layout(self):
        """word offsets of every part of a proof (the same order as CircuitData.prove writes)"""
        cfg, nch = self.cfg, self.cfg.num_challenges
        capw = 4 << self.cap_height
        L = {"wires_cap": 0, "zs_cap": capw, "quotient_cap": 2 * capw}
        off = 3 * capw
        for name, cnt in (("constants", self.num_constants), ("sigmas", cfg.num_routed_wires), ("wires", cfg.num_wires),
                          ("zs", nch), ("zs_next", nch), ("pps", nch * self.num_partial_products),
                          ("quotient", nch * self.quotient_degree_factor)):
            L[name] = (off, cnt)
            off += 2 * cnt
        L["commit_caps"] = off
        off += capw * len(self.reduction_arity_bits)
        L["queries"] = []
        for _ in range(self.num_query_rounds):
            q = {"initial": [], "steps": []}
            plen = self.lde_bits - self.cap_height
            for w in self.oracle_widths:
                q["initial"].append((off, w, off + w, plen))       # leaves at off, siblings at off + w
                off += w + 4 * plen
            for ab in self.reduction_arity_bits:
                plen -= ab
                q["steps"].append((off, 1 << ab, off + 2 * (1 << ab), plen))
                off += 2 * (1 << ab) + 4 * plen
            L["queries"].append(q)
        L["final_poly"] = (off, self.final_poly_len)
        off += 2 * self.final_poly_len
        L["pow_witness"] = off
        off += 1
        L["public_inputs"] = (off, self.num_public_inputs)
        L["end"] = off + self.num_public_inputs
        return L


class ProofTarget:
    """ProofWithPublicInputsTarget, flat (see the module docstring) with views of its parts"""

    def __init__(self, flat, common):
        self.flat, self.common = flat, common
        L = self.L = common.layout()
        capw = 4 << common.cap_height
        hashes = lambda off, n: [flat[off + 4 * i:off + 4 * i + 4] for i in range(n)]
        exts = lambda oc: [(flat[oc[0] + 2 * i], flat[oc[0] + 2 * i + 1]) for i in range(oc[1])]
        ncap = 1 << common.cap_height
        self.wires_cap, self.zs_cap, self.quotient_cap = (hashes(L[k], ncap) for k in ("wires_cap", "zs_cap", "quotient_cap"))
        self.constants, self.sigmas, self.wires, self.zs, self.zs_next, self.pps, self.quotient = (
            exts(L[k]) for k in ("constants", "sigmas", "wires", "zs", "zs_next", "pps", "quotient"))
        self.commit_caps = [hashes(L["commit_caps"] + capw * i, ncap) for i in range(len(common.reduction_arity_bits))]
        self.queries = []
        for q in L["queries"]:
            initial = [(flat[lo:lo + w], hashes(so, plen)) for lo, w, so, plen in q["initial"]]
            steps = [([(flat[eo + 2 * i], flat[eo + 2 * i + 1]) for i in range(ar)], hashes(so, plen)) for eo, ar, so, plen in q["steps"]]
            self.queries.append((initial, steps))
        self.final_poly = exts(L["final_poly"])
        self.pow_witness = flat[L["pow_witness"]]
        po, pn = L["public_inputs"]
        self.public_inputs = flat[po:po + pn]


def add_virtual_proof_with_pis(b, common):
    return ProofTarget(b.add_virtual_targets(common.proof_len()), common)


class VerifierCircuitTarget:
    def __init__(self, b, cap_height):
        self.constants_sigmas_cap = [b.add_virtual_hash() for _ in range(1 << cap_height)]
        self.circuit_digest = b.add_virtual_hash()


def add_virtual_verifier_data(b, cap_height):
    return VerifierCircuitTarget(b, cap_height)


def set_proof_with_pis_target(set_target, proof_target, proof_words):
    assert len(proof_words) == len(proof_target.flat)
    for t, v in zip(proof_target.flat, proof_words):
        set_target(t, int(v))


def set_verifier_data_target(set_target, vd_target, inner_cd):
    for ht, h in zip(vd_target.constants_sigmas_cap, inner_cd.cs_cap.reshape(-1, 4)):
        for t, v in zip(ht, h):
            set_target(t, int(v))
    for t, v in zip(vd_target.circuit_digest, inner_cd.circuit_digest):
        set_target(t, int(v))


# ---- iop/challenger.rs RecursiveChallenger
class RecursiveChallenger:
    def __init__(self, b):
        self.b = b
        self.state = [b.zero()] * 12
        self.inp, self.out = [], []

    def observe_element(self, t):
        self.out = []   # any buffered outputs are now invalid
        self.inp.append(t)

    def observe_elements(self, ts):
        for t in ts:
            self.observe_element(t)

    def observe_hash(self, h):
        self.observe_elements(h)

    def observe_cap(self, cap):
        for h in cap:
            self.observe_hash(h)

    def observe_extension_elements(self, ets):
        for et in ets:
            self.observe_elements(et)

    def _absorb_buffered_inputs(self):
        if not self.inp:
            return
        for off in range(0, len(self.inp), 8):
            chunk = self.inp[off:off + 8]
            self.state[:len(chunk)] = chunk      # overwrite mode
            self.state = self.b.permute(self.state)
        self.out = list(self.state[:8])
        self.inp = []

    def get_challenge(self):
        self._absorb_buffered_inputs()
        if not self.out:
            self.state = self.b.permute(self.state)
            self.out = list(self.state[:8])
        return self.out.pop()

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self):
        return tuple(self.get_n_challenges(2))


# ---- gates/*.rs eval_unfiltered_circuit of the gate types an inner MMR-verifier circuit contains
def _poseidon_rc(b):
    global POSEIDON_RC
    if POSEIDON_RC is None:
        POSEIDON_RC = [int(x) for x in b.o.poseidon_round_constants()]
    return POSEIDON_RC


def _mds_layer_circuit(b, state):
    """Poseidon::mds_layer_circuit: PoseidonMdsGate (48 wires <= 80 routed)"""
    row = b.add_gate(OC.POSEIDON_MDS)
    for i in range(12):
        b.connect_extension(state[i], (OC.wire(row, 2 * i), OC.wire(row, 2 * i + 1)))
    return [(OC.wire(row, 24 + 2 * i), OC.wire(row, 25 + 2 * i)) for i in range(12)]


def _constant_layer_circuit(b, state, round_ctr):
    rc = _poseidon_rc(b)
    for i in range(12):
        c = b.constant_extension((rc[i + 12 * round_ctr], 0))
        state[i] = b.add_extension(state[i], c)


def _poseidon_gate_eval_circuit(b, w):
    """PoseidonGate::eval_unfiltered_circuit (use_mds_gate = true): 123 constraints over the opened wires"""
    cons = []
    swap = w[24]
    cons.append(b.mul_sub_extension(swap, swap, swap))
    for i in range(4):
        diff = b.sub_extension(w[i + 4], w[i])
        cons.append(b.mul_sub_extension(swap, diff, w[25 + i]))
    state = [b.zero_extension()] * 12
    for i in range(4):
        state[i] = b.add_extension(w[i], w[25 + i])
        state[i + 4] = b.sub_extension(w[i + 4], w[25 + i])
    for i in range(8, 12):
        state[i] = w[i]
    round_ctr = 0
    for r in range(4):
        _constant_layer_circuit(b, state, round_ctr)
        if r != 0:
            for i in range(12):
                sbox_in = w[29 + 12 * (r - 1) + i]
                cons.append(b.sub_extension(state[i], sbox_in))
                state[i] = sbox_in
        state = [b.exp_u64_extension(x, 7) for x in state]
        state = _mds_layer_circuit(b, state)
        round_ctr += 1
    for r in range(22):
        _constant_layer_circuit(b, state, round_ctr)
        sbox_in = w[65 + r]
        cons.append(b.sub_extension(state[0], sbox_in))
        state[0] = b.exp_u64_extension(sbox_in, 7)
        state = _mds_layer_circuit(b, state)
        round_ctr += 1
    for r in range(4):
        _constant_layer_circuit(b, state, round_ctr)
        for i in range(12):
            sbox_in = w[87 + 12 * r + i]
            cons.append(b.sub_extension(state[i], sbox_in))
            state[i] = sbox_in
        state = [b.exp_u64_extension(x, 7) for x in state]
        state = _mds_layer_circuit(b, state)
        round_ctr += 1
    for i in range(12):
        cons.append(b.sub_extension(state[i], w[12 + i]))
    return cons


def _gate_eval_unfiltered_circuit(b, kind, gc, w, pi_hash, cfg):
    if kind == NOOP:
        return []
    if kind == CONSTANT:
        return [b.sub_extension(gc[i], w[i]) for i in range(cfg.num_constants)]
    if kind == PUBLIC_INPUT:
        return [b.sub_extension(w[i], b.convert_to_ext(pi_hash[i])) for i in range(4)]
    if kind == ARITHMETIC:
        out = []
        for i in range(cfg.num_routed_wires // 4):
            m0, m1, ad, o = w[4 * i:4 * i + 4]
            scaled_mul = b.mul_many_extension([gc[0], m0, m1])
            computed = b.mul_add_extension(gc[1], ad, scaled_mul)
            out.append(b.sub_extension(o, computed))
        return out
    if kind == POSEIDON:
        return _poseidon_gate_eval_circuit(b, w)
    raise NotImplementedError("in-circuit evaluation of gate kind %d (the reference's inner circuits never contain it)" % kind)


def _evaluate_gate_constraints_circuit(b, common, local_constants, local_wires, pi_hash):
    """plonk/vanishing_poly.rs evaluate_gate_constraints_circuit + Gate::eval_filtered_circuit + compute_filter_circuit"""
    acc = [b.zero_extension()] * common.num_gate_constraints
    for i, kind in enumerate(common.gates):
        sel = common.selector_indices[i]
        gs, ge = common.groups[sel]
        s = local_constants[sel]
        terms = []
        for j in list(range(gs, ge)) + ([UNUSED_SELECTOR] if common.num_selectors > 1 else []):
            if j == i:
                continue
            terms.append(b.sub_extension(b.constant_extension((j, 0)), s))
        filt = b.mul_many_extension(terms)
        mine = _gate_eval_unfiltered_circuit(b, kind, local_constants[common.num_selectors:], local_wires, pi_hash, common.cfg)
        for j, c in enumerate(mine):
            acc[j] = b.mul_add_extension(filt, c, acc[j])
    return acc


def _eval_l_0_circuit(b, n, x, x_pow_n):
    one = b.one_extension()
    neg_one = b.convert_to_ext(b.neg_one())
    eval_zero_poly = b.sub_extension(x_pow_n, one)
    denominator = b.arithmetic_extension(n % P, n % P, x, one, neg_one)
    return b.div_extension(eval_zero_poly, denominator)


def _eval_vanishing_poly_circuit(b, common, x, x_pow_deg, pt, pi_hash, betas, gammas, alphas):
    cfg = common.cfg
    max_degree, num_prods = common.quotient_degree_factor, common.num_partial_products
    constraint_terms = _evaluate_gate_constraints_circuit(b, common, pt.constants, pt.wires, pi_hash)
    vanishing_z_1_terms, vanishing_pp_terms = [], []
    l_0_x = _eval_l_0_circuit(b, 1 << common.degree_bits, x, x_pow_deg)
    s_ids = [b.scalar_mul_ext(b.constant(common.k_is[j]), x) for j in range(cfg.num_routed_wires)]
    for i in range(cfg.num_challenges):
        z_x, z_gx = pt.zs[i], pt.zs_next[i]
        vanishing_z_1_terms.append(b.mul_sub_extension(l_0_x, z_x, l_0_x))
        nums, dens = [], []
        for j in range(cfg.num_routed_wires):
            beta_ext, gamma_ext = b.convert_to_ext(betas[i]), b.convert_to_ext(gammas[i])
            wire_value_plus_gamma = b.add_extension(pt.wires[j], gamma_ext)
            nums.append(b.mul_add_extension(beta_ext, s_ids[j], wire_value_plus_gamma))
            dens.append(b.mul_add_extension(beta_ext, pt.sigmas[j], wire_value_plus_gamma))
        accs = [z_x] + pt.pps[i * num_prods:(i + 1) * num_prods] + [z_gx]
        for q in range(len(accs) - 1):      # check_partial_products_circuit
            nume = b.mul_many_extension(nums[q * max_degree:(q + 1) * max_degree])
            deno = b.mul_many_extension(dens[q * max_degree:(q + 1) * max_degree])
            next_acc_deno = b.mul_extension(accs[q + 1], deno)
            vanishing_pp_terms.append(b.mul_sub_extension(accs[q], nume, next_acc_deno))
    terms = vanishing_z_1_terms + vanishing_pp_terms + constraint_terms
    return [b.reduce_ext(b.convert_to_ext(alpha), terms) for alpha in alphas]


# ---- hash/merkle_proofs.rs
def _verify_merkle_proof_to_cap_with_cap_index(b, leaf_data, leaf_index_bits, cap_index, cap, siblings):
    zero = b.zero()
    state = b.hash_or_noop(list(leaf_data))
    for bit, sib in zip(leaf_index_bits, siblings):
        perm_inputs = list(state) + list(sib) + [zero] * 4
        state = b.permute_swapped(perm_inputs, bit)[:4]
    for i in range(4):
        result = b.random_access(cap_index, [h[i] for h in cap])
        b.connect(result, state[i])


# ---- fri/recursive_verifier.rs
def _fri_combine_initial(b, common, zeta, zeta_next, initial, alpha, subgroup_x, reduced_openings):
    cfg, nch = common.cfg, common.cfg.num_challenges
    subgroup_x = b.convert_to_ext(subgroup_x)
    all_evals = [t for leaves, _ in initial for t in leaves]               # fri_all_polys: every polynomial of the 4 oracles
    next_evals = list(initial[2][0][:nch])                                  # fri_next_batch_polys: the Z's
    total = b.zero_extension()
    count = 0
    for point, evals, reduced in ((zeta, all_evals, reduced_openings[0]), (zeta_next, next_evals, reduced_openings[1])):
        reduced_evals = b.reduce_base(alpha, evals)
        count += len(evals)
        numerator = b.sub_extension(reduced_evals, reduced)
        denominator = b.sub_extension(subgroup_x, point)
        total = b.reducing_shift(alpha, count, total)
        count = 0
        total = b.div_add_extension(numerator, denominator, total)
    # "Multiply the final polynomial by X, so that final_poly has the maximum degree for which the LDT will pass" (plonky2 #436)
    return b.mul_extension(total, subgroup_x)


def _compute_evaluation(b, common, x, x_index_within_coset_bits, arity_bits, evals, beta):
    arity = 1 << arity_bits
    assert arity == 16, "CosetInterpolationGate { subgroup_bits: 4 } is the only interpolation gate these circuits use"
    g = root_of_unity(arity_bits)
    g_inv = pow(g, arity - 1, P)
    ev = list(evals)
    ev = [ev[int(format(i, "0%db" % arity_bits)[::-1], 2)] for i in range(arity)]      # reverse_index_bits_in_place
    start = b.exp_from_bits_const_base(g_inv, x_index_within_coset_bits[::-1])
    coset_start = b.mul(start, x)
    return b.interpolate_coset(coset_start, ev, beta)


def _fri_verifier_query_round(b, common, zeta, zeta_next, fri_alpha, fri_betas, reduced_openings, caps, pt, x_index, query):
    n_log = common.lde_bits
    initial, steps = query
    x_index_bits = b.low_bits(x_index, n_log, 64)
    cap_index = b.le_sum(x_index_bits[len(x_index_bits) - common.cap_height:])
    for (leaves, siblings), cap in zip(initial, caps):                       # fri_verify_initial_proof
        _verify_merkle_proof_to_cap_with_cap_index(b, leaves, x_index_bits, cap_index, cap, siblings)
    g = b.constant(7)                                                       # F::coset_shift()
    phi = b.exp_from_bits_const_base(root_of_unity(n_log), x_index_bits[::-1])
    subgroup_x = b.mul(g, phi)
    old_eval = _fri_combine_initial(b, common, zeta, zeta_next, initial, fri_alpha, subgroup_x, reduced_openings)
    for i, arity_bits in enumerate(common.reduction_arity_bits):
        evals, siblings = steps[i]
        coset_index_bits = x_index_bits[arity_bits:]
        x_index_within_coset_bits = x_index_bits[:arity_bits]
        x_index_within_coset = b.le_sum(x_index_within_coset_bits)
        new_eval = b.random_access_extension(x_index_within_coset, evals)
        b.connect_extension(new_eval, old_eval)
        old_eval = _compute_evaluation(b, common, subgroup_x, x_index_within_coset_bits, arity_bits, evals, fri_betas[i])
        _verify_merkle_proof_to_cap_with_cap_index(b, [t for e in evals for t in e], coset_index_bits, cap_index,
                                                   pt.commit_caps[i], siblings)
        subgroup_x = b.exp_power_of_2(subgroup_x, arity_bits)
        x_index_bits = coset_index_bits
    final_eval = b.reduce_ext(b.convert_to_ext(subgroup_x), pt.final_poly)   # final_poly.eval_scalar
    b.connect_extension(final_eval, old_eval)


def verify_proof(b, pt, vd, common):
    """builder.verify_proof::<PoseidonGoldilocksConfig>(&proof_with_pis, &inner_verifier_data, &inner_common_data)"""
    cfg, nch = common.cfg, common.cfg.num_challenges
    assert len(pt.public_inputs) == common.num_public_inputs
    b.context = "public inputs hash"
    public_inputs_hash = b.hash_n_to_hash_no_pad(list(pt.public_inputs))
    # ---- get_challenges
    b.context = "get_challenges"
    ch = RecursiveChallenger(b)
    ch.observe_hash(vd.circuit_digest)
    ch.observe_hash(public_inputs_hash)
    ch.observe_cap(pt.wires_cap)
    betas = ch.get_n_challenges(nch)
    gammas = ch.get_n_challenges(nch)
    ch.observe_cap(pt.zs_cap)
    alphas = ch.get_n_challenges(nch)
    ch.observe_cap(pt.quotient_cap)
    zeta = ch.get_extension_challenge()
    zeta_batch = pt.constants + pt.sigmas + pt.wires + pt.zs + pt.pps + pt.quotient   # OpeningSetTarget::to_fri_openings
    ch.observe_extension_elements(zeta_batch)
    ch.observe_extension_elements(pt.zs_next)
    fri_alpha = ch.get_extension_challenge()
    fri_betas = []
    for cap in pt.commit_caps:
        ch.observe_cap(cap)
        fri_betas.append(ch.get_extension_challenge())
    ch.observe_extension_elements(pt.final_poly)
    ch.observe_element(pt.pow_witness)
    fri_pow_response = ch.get_challenge()
    fri_query_indices = [ch.get_challenge() for _ in range(common.num_query_rounds)]
    # ---- verify_proof_with_challenges
    b.context = "evaluate the vanishing polynomial at zeta"
    one = b.one_extension()
    zeta_pow_deg = b.exp_power_of_2_extension(zeta, common.degree_bits)
    vanishing = _eval_vanishing_poly_circuit(b, common, zeta, zeta_pow_deg, pt, public_inputs_hash, betas, gammas, alphas)
    b.context = "check vanishing and quotient polynomials"
    z_h_zeta = b.sub_extension(zeta_pow_deg, one)
    qf = common.quotient_degree_factor
    for i in range(nch):
        recombined = b.reduce_ext(zeta_pow_deg, pt.quotient[i * qf:(i + 1) * qf])
        computed = b.mul_extension(z_h_zeta, recombined)
        b.connect_extension(vanishing[i], computed)
    caps = [vd.constants_sigmas_cap, pt.wires_cap, pt.zs_cap, pt.quotient_cap]
    zeta_next = b.mul_const_extension(root_of_unity(common.degree_bits), zeta)        # get_fri_instance_target
    # ---- verify_fri_proof
    b.context = "verify FRI proof: PoW, precomputed reduced openings"
    b.assert_leading_zeros(fri_pow_response, common.proof_of_work_bits)               # fri_verify_proof_of_work
    reduced_openings = [b.reduce_ext(fri_alpha, zeta_batch), b.reduce_ext(fri_alpha, pt.zs_next)]
    for qi, (x_index, query) in enumerate(zip(fri_query_indices, pt.queries)):
        b.context = "FRI query round %d" % qi
        _fri_verifier_query_round(b, common, zeta, zeta_next, fri_alpha, fri_betas, reduced_openings, caps, pt, x_index, query)


# ---- /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:84-140
def complete_verification_circuit_with_inner_proof(oracle, inner_common, nr_peaks):
    """-> (circuit_data, proof target, verifier-data target, [peak hash targets])"""
    b = OC.CircuitBuilder(oracle)
    prev_proof_target = add_virtual_proof_with_pis(b, inner_common)
    prev_proof_verifier_data = add_virtual_verifier_data(b, inner_common.cap_height)
    verify_proof(b, prev_proof_target, prev_proof_verifier_data, inner_common)
    b.context = "peaks and root"
    targets, peaks, equals = [], [], []
    prev_hash = prev_proof_target.public_inputs[0:4]          # quirk Q4: the FIRST PEAK of the inner proof's public inputs
    for _ in range(nr_peaks):
        peak = b.add_virtual_hash()
        peaks.append(peak)
        targets.append(peak)
        equals.append(OC.equal(b, peak, prev_hash))
    hash_in_peaks = OC.or_list(b, equals)
    b.connect(b.one(), hash_in_peaks)
    if len(peaks) > 1:
        root = b.hash_n_to_hash_no_pad([e for p in peaks for e in p])
        b.register_public_inputs(root)
    else:
        b.register_public_inputs(peaks[0])
    b.context = "build"
    return b.build(), prev_proof_target, prev_proof_verifier_data, targets
