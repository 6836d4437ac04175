"""CPU-only checks of the circuit / prover restatement (oracle/circuit.py + oracle/plonk.c; parity unpinned -- the
reference's prover tests only call verify, mmr_plonky2_verifier.rs:147-150): circuit shapes, prove -> verify, tampering,
unsatisfied witnesses."""
from collections import Counter

import numpy as np
import pytest

from circuit_cases import P, assign, mmr_case, synthetic_case
from oracle import circuit as OC


def build_and_assign(oracle, case):
    cd, leaf_t, proof_ts, peak_ts = OC.verify_mmr_proof_circuit(oracle, len(case[1]), len(case[3]))
    pw = {}
    assign(leaf_t, proof_ts, peak_ts, cd.public_inputs, case, pw.__setitem__)
    return cd, pw


def test_config3_circuit_shape(oracle):
    """20 path elements + 1 peak (a leaf of a 2^20 MMR): 41 PoseidonGates, 14 ArithmeticGates, 1 ConstantGate, 1
    PublicInputGate -> 57 rows padded to 64 (d = 6); two selector groups; 84 constants_sigmas."""
    cd, _, _, _ = OC.verify_mmr_proof_circuit(oracle, 20, 1)
    kinds = Counter(g[0] for g in cd.gate_instances)
    assert cd.degree == 64 and cd.degree_bits == 6
    assert kinds == {OC.POSEIDON: 41, OC.ARITHMETIC: 14, OC.CONSTANT: 1, OC.PUBLIC_INPUT: 1, OC.NOOP: 7}
    assert cd.gates == [OC.NOOP, OC.CONSTANT, OC.PUBLIC_INPUT, OC.ARITHMETIC, OC.POSEIDON]
    assert cd.groups == [(0, 4), (4, 5)] and cd.constants_sigmas.shape == (84, 64)
    assert len(cd.public_inputs) == 4
    # sigma is a permutation of the 80 x 64 routed positions
    g = oracle.root_of_unity(6)
    table = {int(cd.k_is[j]) * pow(g, i, P) % P for j in range(80) for i in range(64)}
    assert len(table) == 80 * 64 and {int(x) for x in cd.sigmas.reshape(-1)} == table


@pytest.mark.parametrize("n_leaves,idx", [(3, 1), (11, 6), (1 << 10, 777)])
def test_prove_verify_tamper_real_mmr(oracle, n_leaves, idx):
    case = mmr_case(oracle, n_leaves, idx)
    cd, pw = build_and_assign(oracle, case)
    trace = {}
    proof = cd.prove(pw, trace)
    assert proof.size == cd.proof_len()
    assert list(proof[-4:]) == [int(x) for x in case[4]]  # public inputs = the bagged root
    assert cd.verify(proof) == (True, 0)
    rng = np.random.default_rng(n_leaves)
    for pos in rng.integers(0, proof.size, size=40):
        bad = proof.copy()
        bad[pos] ^= np.uint64(1)
        ok, reason = cd.verify(bad)
        assert not ok and reason != 0
    # every wire of the witness satisfies every gate: the quotient chunks really are the quotient (degree < 8n is
    # implied by the verifier accepting at a random zeta); Z starts at 1
    assert (trace["zs_pp"][:2, 0] == 1).all()


def test_config3_shape_prove_verify(oracle):
    case = synthetic_case(oracle, 20, 7)
    cd, pw = build_and_assign(oracle, case)
    proof = cd.prove(pw)
    assert cd.verify(proof) == (True, 0)
    other = list(proof)
    other[-1] = (int(other[-1]) + 1) % P  # a different public input (root) with the same proof
    assert cd.verify(np.array(other, np.uint64))[0] is False


def test_inconsistent_witness_is_rejected_at_generation(oracle):
    """plonky2 panics ("set twice with different values") when the assignment contradicts the circuit: wrong side bit."""
    case = list(mmr_case(oracle, 11, 6))
    case[2] = case[2].copy()
    case[2][0] ^= 1
    cd, pw = build_and_assign(oracle, tuple(case))
    with pytest.raises(ValueError):
        cd.prove(pw)


@pytest.mark.parametrize("kind,col", [(OC.ARITHMETIC, 3), (OC.POSEIDON, 12), (OC.POSEIDON, 5), (OC.POSEIDON, 100),
                                      (OC.POSEIDON, 70), (OC.CONSTANT, 1), (OC.PUBLIC_INPUT, 2)])
def test_unsatisfied_constraints_do_not_verify(oracle, kind, col):
    """Change one wire after witness generation (an arithmetic output, a Poseidon output / input, non-routed S-box wires,
    a constant wire, a public-input-hash wire): the proof produced from it must be rejected."""
    case = mmr_case(oracle, 3, 1)
    cd, pw = build_and_assign(oracle, case)
    row = next(i for i, g in enumerate(cd.gate_instances) if g[0] == kind)

    def hook(wires):
        wires[col, row] = (int(wires[col, row]) + 1) % P

    try:
        proof = cd.prove(pw, wires_hook=hook)
    except ValueError:
        return  # zero denominator in the permutation argument: also a failure to prove
    assert cd.verify(proof)[0] is False


@pytest.mark.parametrize("n_leaves,idx", [(11, 6), (1 << 10, 5)])
def test_inner_circuit_of_the_recursion(oracle, n_leaves, idx):
    """mmr_plonky2_verifier_1_recursion.rs:20-75 + its driver :152-192: peaks are the public inputs, no root hash."""
    leaf, siblings, lefts, peaks, _ = mmr_case(oracle, n_leaves, idx)
    cd, leaf_t, proof_ts = OC.verify_inner_merkle_proof_circuit(oracle, len(siblings), len(peaks))
    assert len(cd.public_inputs) == 4 * len(peaks)
    pw = {leaf_t: leaf}
    for (ht, bt), sib, left in zip(proof_ts, siblings, lefts):
        for k in range(4):
            pw[ht[k]] = int(sib[k])
        pw[bt] = int(left)
    for k, t in enumerate(cd.public_inputs):
        pw[t] = int(peaks.reshape(-1)[k])
    proof = cd.prove(pw)
    assert cd.verify(proof) == (True, 0)
    assert np.array_equal(proof[-4 * len(peaks):], peaks.reshape(-1))


def test_golden_prove_vectors(oracle, golden):
    """The committed vectors (tools/gen_prove_golden.py) still come out of the restatement: statement -> circuit digest
    and proof words."""
    import hashlib
    for c in golden["prove_vectors"]["cases"]:
        case = (c["leaf"], np.array(c["siblings"], np.uint64).reshape(-1, 4), np.array(c["lefts"], np.uint8),
                np.array(c["peaks"], np.uint64).reshape(-1, 4), np.array(c["root"], np.uint64))
        cd, pw = build_and_assign(oracle, case)
        assert cd.degree_bits == c["degree_bits"] and [int(x) for x in cd.circuit_digest] == c["circuit_digest"]
        proof = cd.prove(pw)
        assert proof.size == c["proof_len"]
        assert hashlib.sha256(proof.astype("<u8").tobytes()).hexdigest() == c["proof_sha256"]
        assert [int(x) for x in proof[:8]] == c["proof_first_words"]
        assert [int(x) for x in proof[-len(c["public_inputs"]):]] == c["public_inputs"]


def test_copy_constraints_alone_are_enforced(oracle):
    """Change an ArithmeticGate operand AND its output consistently (the gate constraint still holds) so that only the copy
    constraints -- the permutation argument (sigma, Z, partial products) -- can catch it: the proof must not verify."""
    case = mmr_case(oracle, 3, 1)
    cd, pw = build_and_assign(oracle, case)
    row = next(i for i, g in enumerate(cd.gate_instances) if g[0] == OC.ARITHMETIC)
    c0, c1 = cd.gate_instances[row][1]

    def hook(wires):
        m0, m1, ad = (int(wires[k, row]) for k in range(3))
        m0 = (m0 + 1) % P
        wires[0, row] = m0
        wires[3, row] = (m0 * m1 % P * c0 + ad * c1) % P

    try:
        proof = cd.prove(pw, wires_hook=hook)
    except ValueError:
        return
    ok, reason = cd.verify(proof)
    assert not ok and reason == 11
