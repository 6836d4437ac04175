"""The oracle's restatement of plonky2's in-circuit verifier and of the reference's outer recursion circuit
(/root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:84-140, test driver :152-221), CPU only:
inner circuit -> inner proof -> outer circuit (builder.verify_proof) -> outer witness -> outer proof -> verify.
[parity unpinned: plonky2 is absent; what is checked here is that the construction is self-consistent -- every row of the
outer witness satisfies its own gate's constraints, the outer proof is accepted, an invalid inner proof cannot be witnessed.]"""
import numpy as np
import pytest

from circuit_cases import mmr_case
from oracle import circuit as OC, recursion as R

P = 0xFFFFFFFF00000001


def inner_prove(oracle, case):
    leaf, sib, lefts, peaks, root = case
    icd, leaf_t, proof_ts = OC.verify_inner_merkle_proof_circuit(oracle, len(sib), len(peaks))
    pw = {leaf_t: leaf}
    for (ht, bt), s, l in zip(proof_ts, sib, lefts):
        for k in range(4):
            pw[ht[k]] = int(s[k])
        pw[bt] = int(l)
    for i, pk in enumerate(peaks):
        for k in range(4):
            pw[icd.public_inputs[4 * i + k]] = int(pk[k])
    return icd, icd.prove(pw)


def outer_witness(oracle, icd, inner_proof, case):
    leaf, sib, lefts, peaks, root = case
    common = R.CommonData(icd)
    ocd, ptgt, vdt, peak_ts = R.complete_verification_circuit_with_inner_proof(oracle, common, len(peaks))
    pw2 = {}
    R.set_proof_with_pis_target(pw2.__setitem__, ptgt, inner_proof)
    R.set_verifier_data_target(pw2.__setitem__, vdt, icd)
    for pt, pk in zip(peak_ts, peaks):
        for k in range(4):
            pw2[pt[k]] = int(pk[k])
    for k in range(4):
        pw2[ocd.public_inputs[k]] = int(root[k])
    return ocd, pw2


@pytest.fixture(scope="module")
def seven_leaves(oracle):
    """test_mmr_verifier_7leaves_multiple's shape (:229-235): 3 peaks; leaf 5 sits in the SECOND mountain, so quirk Q4 (the
    outer circuit compares the peaks with the inner proof's FIRST public-input peak) is what makes the check pass"""
    case = mmr_case(oracle, 7, 5)
    icd, inner_proof = inner_prove(oracle, case)
    assert icd.verify(inner_proof) == (True, 0)
    ocd, pw2 = outer_witness(oracle, icd, inner_proof, case)
    return case, icd, inner_proof, ocd, pw2


def test_outer_circuit_shape(seven_leaves):
    case, icd, inner_proof, ocd, pw2 = seven_leaves
    assert R.CommonData(icd).proof_len() == inner_proof.size
    assert ocd.degree_bits == 12 and ocd.num_selectors == 3      # "outer d = 11 or 12"
    # plonky2's gate order (degree, id); no reduction layer for this tiny inner circuit, so no CosetInterpolationGate
    assert ocd.gates == [OC.NOOP, OC.CONSTANT, OC.POSEIDON_MDS, OC.PUBLIC_INPUT, OC.BASE_SUM, OC.REDUCING_EXT, OC.REDUCING,
                         OC.ARITHMETIC_EXT, OC.ARITHMETIC, OC.MUL_EXT, OC.RANDOM_ACCESS, OC.POSEIDON]
    assert ocd.groups == [(0, 7), (7, 11), (11, 12)]
    kinds = {}
    for k, _ in ocd.gate_instances:
        kinds[k] = kinds.get(k, 0) + 1
    assert kinds[OC.POSEIDON_MDS] == 30 and kinds[OC.PUBLIC_INPUT] == 1      # one in-circuit PoseidonGate evaluation: 30 MDS layers


def test_outer_witness_satisfies_every_gate(oracle, seven_leaves):
    case, icd, inner_proof, ocd, pw2 = seven_leaves
    wires, getv = ocd.generate_witness(pw2)
    pis = np.array([getv(t) for t in ocd.public_inputs], np.uint64)
    assert np.array_equal(pis, case[4])
    pih = oracle.hash_no_pad(pis)
    for row, (kind, cs) in enumerate(ocd.gate_instances):
        c = oracle.gate_constraints_row(kind, wires[:, row].copy(), np.array(list(cs) + [0, 0], np.uint64)[:2], pih)
        assert not c.any(), (row, kind, ocd.row_context[row])


def test_outer_prove_and_verify(oracle, seven_leaves):
    case, icd, inner_proof, ocd, pw2 = seven_leaves
    proof = ocd.prove(pw2)
    assert proof.size == ocd.proof_len()
    assert np.array_equal(proof[-4:], case[4])                   # the public input is the MMR root
    assert ocd.verify(proof) == (True, 0)
    bad = proof.copy()
    bad[200] = (int(bad[200]) + 1) % P                            # an opening
    assert ocd.verify(bad)[0] is False
    bad = proof.copy()
    bad[-1] = (int(bad[-1]) + 1) % P                              # a different root
    assert ocd.verify(bad)[0] is False


@pytest.mark.parametrize("word", [5, 70, 250, 700, 1500, 9000, -20, -2])
def test_invalid_inner_proof_cannot_be_witnessed(oracle, seven_leaves, word):
    """Any single-word change of the inner proof (a cap, an opening, a FRI value, the PoW witness, a public input) makes some
    in-circuit check fail: witness generation hits plonky2's "set twice with different values" (or a range check)"""
    case, icd, inner_proof, ocd, pw2 = seven_leaves
    bad = inner_proof.copy()
    bad[word] = (int(bad[word]) + 1) % P
    assert icd.verify(bad)[0] is False
    common = R.CommonData(icd)
    ocd2, pw_bad = ocd, dict(pw2)
    ptgt = R.ProofTarget([("v", i) for i in range(common.proof_len())], common)   # the first virtual targets are the proof's
    R.set_proof_with_pis_target(pw_bad.__setitem__, ptgt, bad)
    with pytest.raises((ValueError, AssertionError)):
        ocd2.generate_witness(pw_bad)
