"""mmr_plonky2_verifier_1_recursion on the GPU (BASELINE config 4): inner circuit -> inner proof -> outer circuit
(plonky2's in-circuit verifier built by the library) -> outer witness on the device -> outer proof -> verify.
Everything is compared stage by stage with the oracle's restatement (oracle/recursion.py): circuit (gate rows, selector groups,
constants_sigmas values, cap, digest), witness matrix, challenges, Z / partial products, quotient chunks, proof words.
The reference's five recursion tests (/root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:223-257) are re-expressed below.
[parity unpinned: plonky2 is absent; both sides restate it from its published algorithm]"""
import numpy as np
import pytest

import __graft_entry__ as ge
from circuit_cases import mmr_case, synthetic_case
from oracle import circuit as OC, recursion as R
from test_circuit_gpu import check_build, check_prove

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def inner_both(pkg, oracle, case):
    """verify_inner_merkle_proof_circuit + witness (:168-190) on both sides -> (gpu circuit, gpu witness, oracle circuit, oracle witness)"""
    leaf, sib, lefts, peaks, root = case
    gcd, gleaf, gproof_ts = pkg.verify_inner_merkle_proof_circuit(len(sib), len(peaks))
    ocd, oleaf, oproof_ts = OC.verify_inner_merkle_proof_circuit(oracle, len(sib), len(peaks))
    pw, opw = pkg.PartialWitness(), {}
    for set_target, leaf_t, proof_ts, pis in ((pw.set_target, gleaf, gproof_ts, gcd.prover_only.public_inputs),
                                              (opw.__setitem__, oleaf, oproof_ts, ocd.public_inputs)):
        set_target(leaf_t, leaf)
        for (ht, bt), s, l in zip(proof_ts, sib, lefts):
            for k in range(4):
                set_target(ht[k], int(s[k]))
            set_target(bt, int(l))
        for i, pk in enumerate(peaks):
            for k in range(4):
                set_target(pis[4 * i + k], int(pk[k]))
    return gcd, pw, ocd, opw


def outer_both(pkg, oracle, case, gcd_inner, ocd_inner, inner_proof):
    """complete_verification_circuit_with_inner_proof + witness (:195-216) on both sides"""
    leaf, sib, lefts, peaks, root = case
    gcd, gpt, gvd, gpeak_ts = pkg.complete_verification_circuit_with_inner_proof(gcd_inner.common, len(peaks))
    pw = pkg.PartialWitness()
    pw.set_proof_with_pis_target(gpt, inner_proof)
    pw.set_verifier_data_target(gvd, gcd_inner.verifier_only)
    for pt, pk in zip(gpeak_ts, peaks):
        pw.set_hash_target(pt, [int(x) for x in pk])
    for k, t in enumerate(gcd.prover_only.public_inputs):
        pw.set_target(t, int(root[k]))
    ocd, opt, ovd, opeak_ts = R.complete_verification_circuit_with_inner_proof(oracle, R.CommonData(ocd_inner), len(peaks))
    opw = {}
    R.set_proof_with_pis_target(opw.__setitem__, opt, inner_proof)
    R.set_verifier_data_target(opw.__setitem__, ovd, ocd_inner)
    for pt, pk in zip(opeak_ts, peaks):
        for k in range(4):
            opw[pt[k]] = int(pk[k])
    for k in range(4):
        opw[ocd.public_inputs[k]] = int(root[k])
    return gcd, pw, ocd, opw


def run_recursion(pkg, oracle, case, full=True):
    gi, pwi, oi, opwi = inner_both(pkg, oracle, case)
    check_build(gi, oi)
    inner_proof = check_prove(gi, pwi, oi, opwi) if full else gi.prove(pwi)
    assert gi.verify(inner_proof)
    go, pwo, oo, opwo = outer_both(pkg, oracle, case, gi, oi, inner_proof)
    check_build(go, oo)
    if full:
        final_proof = check_prove(go, pwo, oo, opwo)          # witness, challenges, Z, quotient, proof words == oracle; oracle verifies
    else:
        assert np.array_equal(go.generate_witness(pwo), oo.generate_witness(opwo)[0])
        final_proof = go.prove(pwo)
        assert oo.verify(final_proof) == (True, 0)
    assert go.verify(final_proof)                              # main_circuit_data.verify(final_proof) (:220)
    assert np.array_equal(final_proof[-4:], case[4])           # the public input is the MMR root
    return go, final_proof


def test_mmr_verifier_2leaves(pkg, oracle):
    """:223-227"""
    run_recursion(pkg, oracle, mmr_case(oracle, 2, 0))


@pytest.mark.parametrize("leaf", [0, 3, 5, 6])
def test_mmr_verifier_7leaves_multiple(pkg, oracle, leaf):
    """:229-236 (3 peaks; leaves 4..6 sit outside the first mountain: quirk Q4 makes the outer peak check pass regardless)"""
    run_recursion(pkg, oracle, mmr_case(oracle, 7, leaf), full=(leaf == 5))


@pytest.mark.parametrize("leaf", [0, 7])
def test_mmr_verifier_8leaves_multiple(pkg, oracle, leaf):
    """:238-245"""
    run_recursion(pkg, oracle, mmr_case(oracle, 8, leaf), full=False)


def test_mmr_verifier_31leaves(pkg, oracle):
    """:247-251"""
    run_recursion(pkg, oracle, mmr_case(oracle, 31, 8), full=False)


def test_mmr_verifier_1031leaves(pkg, oracle):
    """:253-257: 4 peaks, 10 path elements -> the inner circuit has a FRI reduction layer, the outer one CosetInterpolationGates"""
    go, _ = run_recursion(pkg, oracle, mmr_case(oracle, 1031, 100))
    assert go.info.gate_counts[11] == 28 and go.degree_bits == 12


def test_config4_leaf_of_a_2pow20_mmr(pkg, oracle):
    """BASELINE config 4: inner + outer prove for a leaf of a 2^20-leaf MMR (20 path elements, 1 peak; the MMR and its membership
    proof come from the device-resident MMR of config 2)"""
    leaves = pkg.synthetic.splitmix_leaves(1 << 20, 0x5EED0000 + 3)
    mmr = pkg.MMR.from_leaves(leaves)
    root = mmr.bagging_the_peaks()
    pr = mmr.get_proof_normal_index(777777)
    case = (int(leaves[777777]), pr.siblings, pr.lefts, pr.peaks, root)
    go, final_proof = run_recursion(pkg, oracle, case)
    assert go.degree_bits == 12
    bad = final_proof.copy()
    bad[300] = (int(bad[300]) + 1) % P
    assert go.verify(bad, with_reason=True)[0] is False


def test_outer_witness_rejects_a_tampered_inner_proof(pkg, oracle):
    """plonky2 panics in generate_partial_witness ("set twice with different values") when the inner proof does not verify"""
    case = mmr_case(oracle, 8, 3)
    gi, pwi, oi, opwi = inner_both(pkg, oracle, case)
    inner_proof = gi.prove(pwi)
    bad = inner_proof.copy()
    bad[250] = (int(bad[250]) + 1) % P
    go, pwo, oo, opwo = outer_both(pkg, oracle, case, gi, oi, bad)
    with pytest.raises(pkg.P2mtPanic):
        go.prove(pwo)


def test_host_evaluated_poseidon_chain_same_proof(pkg, oracle):
    """Single proves evaluate the long PoseidonGate chain of the outer circuit (the inner proof's transcript) on the host before the
    launch (p2mt_circuit.hip select_host_chain): the proof is the same words with it, without it, and from the batched prover (which
    never uses it and re-schedules); the schedule loses the chain's levels."""
    import ctypes as C
    lib = pkg.lib()

    def info(cd):
        lv, rows = C.c_uint32(), C.c_uint32()
        pkg._native.check(lib.p2mt_circuit_schedule_info(cd._h, C.byref(lv), C.byref(rows)))
        return lv.value, rows.value

    case = mmr_case(oracle, 8, 3)
    gi, pwi, oi, opwi = inner_both(pkg, oracle, case)
    inner_proof = gi.prove(pwi)
    assert info(gi)[1] == 0  # 3 path elements: a chain of 4 rows is below the threshold
    go, pwo, oo, opwo = outer_both(pkg, oracle, case, gi, oi, inner_proof)
    try:
        a = go.prove(pwo)
        lv_on, rows_on = info(go)
        pkg._native.check(lib.p2mt_debug_host_chain(0))
        b = go.prove(pwo)
        lv_off, rows_off = info(go)
        batch = pkg.BatchProver(go, 2).prove([pwo, pwo])
        pkg._native.check(lib.p2mt_debug_host_chain(1))
        c = go.prove(pwo)
    finally:
        pkg._native.check(lib.p2mt_debug_host_chain(1))
    assert rows_off == 0 and 90 <= rows_on <= 192, (rows_on, rows_off)
    assert lv_on + 40 < lv_off, (lv_on, lv_off)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert all(np.array_equal(a, x) for x in batch)
    assert oo.verify(a) == (True, 0)


@pytest.mark.parametrize("mode", ["1", "0"])
def test_outer_witness_interpreters_agree(mode):
    """The three interpreters of the generator schedule (dataflow = default, level-synchronous over the grid, one workgroup)
    produce the same outer proof (fresh process: the knob is read once)."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path[:0] = [%r, %r]\n"
        "import numpy as np, hashlib, __graft_entry__ as ge\n"
        "from oracle_lib import Oracle\n"
        "from circuit_cases import mmr_case\n"
        "import test_recursion_gpu as T\n"
        "pkg = ge.load_package(); pkg.init(0); o = Oracle()\n"
        "case = mmr_case(o, 8, 3)\n"
        "gi, pwi, oi, opwi = T.inner_both(pkg, o, case)\n"
        "ip = gi.prove(pwi)\n"
        "go, pwo, oo, opwo = T.outer_both(pkg, o, case, gi, oi, ip)\n"
        "print('SHA', hashlib.sha256(go.prove(pwo).tobytes()).hexdigest())\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for env in ({}, {"P2MT_WITNESS_GRID": mode}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        shas.append([l for l in r.stdout.splitlines() if l.startswith("SHA")][-1])
    assert shas[0] == shas[1]


@pytest.fixture(scope="module")
def outer_case(pkg, oracle):
    case = mmr_case(oracle, 1031, 100)          # inner circuit with a FRI reduction layer: every recursion gate type occurs
    gi, pwi, oi, opwi = inner_both(pkg, oracle, case)
    inner_proof = gi.prove(pwi)
    go, pwo, oo, opwo = outer_both(pkg, oracle, case, gi, oi, inner_proof)
    return go, pwo, oo, opwo


# (gate kind, wire column to corrupt): one wire of every gate type the in-circuit verifier adds
@pytest.mark.parametrize("kind,col", [(OC.BASE_SUM, 5), (OC.ARITHMETIC_EXT, 6), (OC.MUL_EXT, 4), (OC.REDUCING, 60), (OC.REDUCING_EXT, 80),
                                      (OC.RANDOM_ACCESS, 75), (OC.COSET_INTERPOLATION, 38), (OC.POSEIDON_MDS, 30)])
def test_recursion_gate_constraints_are_enforced(pkg, oracle, outer_case, kind, col):
    """A witness with one wire of a recursion gate changed after generation (so that exactly that gate's constraints break) must
    not yield a proof either verifier accepts: the constraints of all eight new gate types are live in the quotient and in the
    verifiers (the proof is made by the oracle's prover, which -- like plonky2's -- does not check the witness)."""
    go, pwo, oo, opwo = outer_case
    row = next(i for i, g in enumerate(oo.gate_instances) if g[0] == kind)

    def hook(wires):
        wires[col, row] = (int(wires[col, row]) + 1) % P

    bad = oo.prove(opwo, wires_hook=hook)
    assert oo.verify(bad)[0] is False
    acc, reason = go.verify(bad, with_reason=True)
    assert acc is False and reason == 11, (acc, reason)       # vanishing polynomial != Z_H * quotient at zeta


def test_concurrent_outer_provers(pkg, oracle):
    """Four provers of the outer circuit at once (one handle and stream per thread inside the library): the dataflow witness
    interpreter's wavefronts of different proofs share the device; every proof equals the single-prover one."""
    case = mmr_case(oracle, 8, 3)
    gi, pwi, oi, opwi = inner_both(pkg, oracle, case)
    inner_proof = gi.prove(pwi)
    handles, witnesses = [], []
    for _ in range(4):
        leaf, sib, lefts, peaks, root = case
        go, gpt, gvd, gpeak_ts = pkg.complete_verification_circuit_with_inner_proof(gi.common, len(peaks))
        handles.append(go)
    for k in range(8):
        go = handles[0]  # targets are the same in every build of the circuit
        pw = pkg.PartialWitness()
        pw.set_proof_with_pis_target(gpt, inner_proof)
        pw.set_verifier_data_target(gvd, gi.verifier_only)
        for pt, pk in zip(gpeak_ts, peaks):
            pw.set_hash_target(pt, [int(x) for x in pk])
        for j, t in enumerate(go.prover_only.public_inputs):
            pw.set_target(t, int(root[j]))
        witnesses.append(pw)
    want = handles[0].prove(witnesses[0])
    got = pkg.prove_many(handles, witnesses)
    for k in range(8):
        assert np.array_equal(got[k], want), k
    assert handles[1].verify(got[5])
