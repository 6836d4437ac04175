"""The reference's circuit tests re-expressed IN FULL (round 2 sampled them): every leaf of every MMR size the reference proves.

  /root/reference/src/mmr/mmr_plonky2_verifier.rs:153-209 -- test_mmr_verifier_{3,7,31,70}leaves, _multiple_sizes_1 (6..15 leaves),
      _multiple_sizes_2 (0..39 leaves): for every size, for every leaf: MMR -> get_proof -> verify -> verify_mmr_proof_circuit
      (proof length, number of peaks) -> witness -> prove -> circuit_data.verify(proof).
  /root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:229-245 -- every leaf of the 7- and of the 8-leaf MMR through inner
      prove -> outer circuit -> outer prove -> verify.

The batched prover makes this seconds of GPU time: leaves with the same circuit shape (path length, number of peaks) are proved in
one pass.  Checks per proof: p2mt_circuit_verify_batch accepts it, the oracle's verifier restatement accepts it, its public inputs
are the MMR root; per circuit shape: built circuit == the oracle's (gate rows, constants_sigmas, cap, digest) and one proof equals the
oracle prover's word for word.  [parity unpinned: plonky2 is absent; both sides restate it from its published algorithm]"""
import collections

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import circuit as OC, recursion as R
from test_circuit_gpu import check_build

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
SIZES = list(range(1, 40)) + [70]   # 0..39 (size 0 proves nothing), 3, 7, 31, 6..15 are inside; 70


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def _mmr_proofs(pkg, n_leaves):
    """MMR of n random leaves on the GPU; every leaf's proof through the batched proof service; root.  Every proof verifies
    (MMR_proof::verify, :115)."""
    rng = np.random.default_rng(1000 + n_leaves)
    leaves = rng.integers(0, P, size=n_leaves, dtype=np.uint64)
    mmr = pkg.MMR()
    for l in leaves:  # the reference's loop (:108-111), write-combined by the library
        mmr.add_leaf(int(l))
    root = mmr.bagging_the_peaks()
    peaks = mmr.get_peaks()
    idx = np.array([pkg.get_mmr_index(i) for i in range(n_leaves)], np.uint64)
    sib, lefts, ns = mmr.get_proof_batch(idx)
    status = pkg.verify_proof_batch(sib, lefts, ns, peaks, leaves, root)
    assert (status == 1).all()
    return leaves, root, peaks, [(sib[i, :ns[i]].copy(), lefts[i, :ns[i]].copy()) for i in range(n_leaves)]


def test_mmr_verifier_every_leaf_of_every_size(pkg, oracle):
    """mmr_plonky2_verifier.rs:153-209 in full: 850 proofs."""
    assign = pkg.synthetic.assign_mmr_proof
    # group all (size, leaf) statements by circuit shape
    by_shape = collections.defaultdict(list)
    for n in SIZES:
        leaves, root, peaks, proofs = _mmr_proofs(pkg, n)
        for i, (sib, lefts) in enumerate(proofs):
            by_shape[(len(sib), len(peaks))].append((int(leaves[i]), sib, lefts, peaks, root))
    assert sum(len(v) for v in by_shape.values()) == sum(SIZES) == 850
    for (n_sib, n_peaks), cases in sorted(by_shape.items()):
        cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(n_sib, n_peaks)
        ocd, oleaf, oproof_ts, opeak_ts = OC.verify_mmr_proof_circuit(oracle, n_sib, n_peaks)
        check_build(cd, ocd)
        pws = []
        for case in cases:
            pw = pkg.PartialWitness()
            assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
            pws.append(pw)
        proofs = pkg.BatchProver(cd, min(len(pws), 64)).prove(pws)
        accepted, reasons = cd.verify_batch(proofs)
        assert all(accepted), ((n_sib, n_peaks), reasons)
        for proof, case in zip(proofs, cases):
            assert ocd.verify(proof) == (True, 0)
            assert np.array_equal(proof[-4:], case[4])  # public inputs = the MMR root
        # one statement per shape word for word against the oracle's prover (and against the one-at-a-time prover)
        opw = {}
        assign(oleaf, oproof_ts, opeak_ts, ocd.public_inputs, cases[0], opw.__setitem__)
        assert np.array_equal(proofs[0], ocd.prove(opw)), (n_sib, n_peaks)
        assert np.array_equal(proofs[-1], cd.prove(pws[-1]))


def _inner_witness(pkg, gcd, gleaf, gproof_ts, case):
    leaf, sib, lefts, peaks, root = case
    pw = pkg.PartialWitness()
    pw.set_target(gleaf, leaf)
    for (ht, bt), s, l in zip(gproof_ts, sib, lefts):
        pw.set_hash_target(ht, [int(x) for x in s])
        pw.set_target(bt, int(l))
    for i, pk in enumerate(peaks):
        for k in range(4):
            pw.set_target(gcd.prover_only.public_inputs[4 * i + k], int(pk[k]))
    return pw


@pytest.mark.parametrize("n_leaves", [7, 8])
def test_recursion_every_leaf(pkg, oracle, n_leaves):
    """mmr_plonky2_verifier_1_recursion.rs:229-245 in full: every leaf of the 7-leaf (3 peaks; leaves 4..6 lie outside the first
    mountain, quirk Q4) and of the 8-leaf MMR: inner proofs in one batch per shape, outer proofs in one batch per shape."""
    leaves, root, peaks, proofs = _mmr_proofs(pkg, n_leaves)
    by_shape = collections.defaultdict(list)
    for i, (sib, lefts) in enumerate(proofs):
        by_shape[len(sib)].append((int(leaves[i]), sib, lefts, peaks, root))
    done = 0
    for n_sib, cases in sorted(by_shape.items()):
        gi, gleaf, gproof_ts = pkg.verify_inner_merkle_proof_circuit(n_sib, len(peaks))
        oi, oleaf, oproof_ts = OC.verify_inner_merkle_proof_circuit(oracle, n_sib, len(peaks))
        check_build(gi, oi)
        ipws = [_inner_witness(pkg, gi, gleaf, gproof_ts, c) for c in cases]
        inner_proofs = pkg.BatchProver(gi, len(ipws)).prove(ipws)
        acc, _ = gi.verify_batch(inner_proofs)
        assert all(acc)
        go, gpt, gvd, gpeak_ts = pkg.complete_verification_circuit_with_inner_proof(gi.common, len(peaks))
        oo, opt, ovd, opeak_ts = R.complete_verification_circuit_with_inner_proof(oracle, R.CommonData(oi), len(peaks))
        check_build(go, oo)
        opws = []
        for ip in inner_proofs:
            assert oi.verify(ip) == (True, 0)
            pw = pkg.PartialWitness()
            pw.set_proof_with_pis_target(gpt, ip)
            pw.set_verifier_data_target(gvd, gi.verifier_only)
            for pt, pk in zip(gpeak_ts, peaks):
                pw.set_hash_target(pt, [int(x) for x in pk])
            for k, t in enumerate(go.prover_only.public_inputs):
                pw.set_target(t, int(root[k]))
            opws.append(pw)
        outer_proofs = pkg.BatchProver(go, len(opws)).prove(opws)
        acc, reasons = go.verify_batch(outer_proofs)
        assert all(acc), reasons
        for op in outer_proofs:
            assert oo.verify(op) == (True, 0)         # main_circuit_data.verify(final_proof) (:220), oracle side
            assert np.array_equal(op[-4:], root)
        # the outer WITNESS of one statement per shape against the oracle's generator (the word-for-word outer prove of these
        # shapes is tests/test_recursion_gpu.py's)
        w = {}
        R.set_proof_with_pis_target(w.__setitem__, opt, inner_proofs[0])
        R.set_verifier_data_target(w.__setitem__, ovd, oi)
        for pt, pk in zip(opeak_ts, peaks):
            for k in range(4):
                w[pt[k]] = int(pk[k])
        for k in range(4):
            w[oo.public_inputs[k]] = int(root[k])
        assert np.array_equal(go.generate_witness(opws[0]), oo.generate_witness(w)[0])
        done += len(cases)
    assert done == n_leaves


@pytest.mark.parametrize("n_sib,n_peaks", [(31, 1), (31, 2)])
def test_recursion_over_a_2pow7_row_inner_circuit(pkg, oracle, n_sib, n_peaks):
    """verify_inner_merkle_proof_circuit is generic in nr_merkle_proof_elms (mmr_plonky2_verifier_1_recursion.rs:20-75): ~31 path
    elements -- MMRs near the reference's own 2^31-leaf limit (quirk Q6) -- pad the inner circuit to 2^7 rows, one more FRI query
    step bit and a longer Merkle path per query than the 2^6-row case of config 4.  The outer circuit over it (still 2^12 rows):
    built circuit == oracle's, outer witness == oracle's, the outer proof is accepted by both verifiers, a tampered inner proof
    cannot be witnessed."""
    from circuit_cases import synthetic_case
    leaf, sib, lefts, peaks1, root1 = synthetic_case(oracle, n_sib, 4242 + n_peaks)
    # n_peaks > 1: the recomputed mountain root is the LAST peak, the others are arbitrary digests; root = bagging of all of them
    rng = np.random.default_rng(99)
    peaks = np.concatenate([rng.integers(0, P, size=(n_peaks - 1, 4), dtype=np.uint64), peaks1]) if n_peaks > 1 else peaks1
    root = root1 if n_peaks == 1 else oracle.hash_no_pad(peaks.reshape(-1))
    case = (leaf, sib, lefts, peaks, root)
    gi, gleaf, gproof_ts = pkg.verify_inner_merkle_proof_circuit(n_sib, n_peaks)
    oi, oleaf, oproof_ts = OC.verify_inner_merkle_proof_circuit(oracle, n_sib, n_peaks)
    assert gi.info.degree_bits == 7
    check_build(gi, oi)
    ipw = _inner_witness(pkg, gi, gleaf, gproof_ts, case)
    inner_proof = gi.prove(ipw)
    assert gi.verify(inner_proof) and oi.verify(inner_proof) == (True, 0)
    go, gpt, gvd, gpeak_ts = pkg.complete_verification_circuit_with_inner_proof(gi.common, n_peaks)
    oo, opt, ovd, opeak_ts = R.complete_verification_circuit_with_inner_proof(oracle, R.CommonData(oi), n_peaks)
    assert go.info.degree_bits == 12
    check_build(go, oo)

    def outer_witness(ip):
        pw = pkg.PartialWitness()
        pw.set_proof_with_pis_target(gpt, ip)
        pw.set_verifier_data_target(gvd, gi.verifier_only)
        for pt, pk in zip(gpeak_ts, peaks):
            pw.set_hash_target(pt, [int(x) for x in pk])
        for k, t in enumerate(go.prover_only.public_inputs):
            pw.set_target(t, int(root[k]))
        return pw

    opw = outer_witness(inner_proof)
    w = {}
    R.set_proof_with_pis_target(w.__setitem__, opt, inner_proof)
    R.set_verifier_data_target(w.__setitem__, ovd, oi)
    for pt, pk in zip(opeak_ts, peaks):
        for k in range(4):
            w[pt[k]] = int(pk[k])
    for k in range(4):
        w[oo.public_inputs[k]] = int(root[k])
    assert np.array_equal(go.generate_witness(opw), oo.generate_witness(w)[0])
    final_proof = go.prove(opw)
    assert go.verify(final_proof) and oo.verify(final_proof) == (True, 0)
    assert np.array_equal(final_proof[-4:], root)
    bad = inner_proof.copy()
    bad[300] ^= 1
    with pytest.raises(pkg.P2mtError):
        go.prove(outer_witness(bad))
