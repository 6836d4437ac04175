"""The batched prover (p2mt_batch_prover_*): B proofs of one mmr_plonky2_verifier circuit per pass of the prover pipeline, the proof
index riding in a grid dimension of every launch.  The bar is the same as for prove itself: every proof's words equal the
sequential `circuit_data.prove(pw)` bit for bit (which test_circuit_gpu.py pins against the CPU restatement), and verify accepts."""
import numpy as np
import pytest

import __graft_entry__ as ge
from circuit_cases import P, assign, synthetic_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def circuit_and_witnesses(pkg, oracle, n_sib, seeds):
    cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(n_sib, 1)
    cases = [synthetic_case(oracle, n_sib, s) for s in seeds]
    pws = []
    for case in cases:
        pw = pkg.PartialWitness()
        assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
        pws.append(pw)
    return cd, (leaf_t, proof_ts, peak_ts), cases, pws


@pytest.mark.parametrize("batch", [1, 4, 32])
def test_batch_equals_sequential(pkg, oracle, batch):
    """24 different statements (config 3's shape: 20 path elements, one peak) in passes of `batch` (the last pass is ragged for 32):
    same words as 24 sequential proves (each with the smallest proof-of-work witness, which the batch's early-exit grind must keep)."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, range(500, 524))
    want = [cd.prove(pw) for pw in pws]
    bp = pkg.BatchProver(cd, batch)
    assert bp.batch == batch
    got = bp.prove(pws)
    assert got.shape == (len(pws), cd.info.proof_len)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert all(cd.verify(g) for g in got[:4])
    # the circuit handle is its own again afterwards: a plain prove, then another batch
    assert np.array_equal(cd.prove(pws[3]), want[3])
    again = bp.prove(pws[5:11])
    for g, w in zip(again, want[5:11]):
        assert np.array_equal(g, w)


def test_batch_grind_shared_first_round_and_forced_redo(pkg, oracle):
    """The batch's proof-of-work queue shares eleven of the twelve first-round S-boxes between the candidates of a proof and computes
    only word 7 of the last layer (k_fri_pow_queue<2, 5>): the witnesses -- hence every proof word -- must stay those of the sequential
    prove, also with every wave forced onto the exact redo path (the reference permutation from the full input)."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, range(560, 566))
    want = [cd.prove(pw) for pw in pws]
    bp = pkg.BatchProver(cd, 4)
    lib = pkg.lib()
    got = bp.prove(pws)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    try:
        lib.p2mt_debug_force_fallback(1)
        got = bp.prove(pws[:4])
    finally:
        lib.p2mt_debug_force_fallback(0)
    for g, w in zip(got, want[:4]):
        assert np.array_equal(g, w)


def test_batch_layouts_agree(pkg, oracle):
    """Throughput mode switches the leaf sponges / Merkle levels to the lane-per-hash layouts; the words do not change."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, range(540, 548))
    want = [cd.prove(pw) for pw in pws]
    bp = pkg.BatchProver(cd, 8)
    lib = pkg.lib()
    try:
        pkg._native.check(lib.p2mt_set_throughput_mode(1))
        got = bp.prove(pws)
    finally:
        pkg._native.check(lib.p2mt_set_throughput_mode(0))
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def test_batch_larger_circuit(pkg, oracle):
    """200 path elements: 512 rows (degree_bits 9), the largest circuit whose witness table still fits LDS."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 200, range(560, 565))
    assert cd.info.degree_bits == 9
    want = [cd.prove(pw) for pw in pws]
    got = pkg.BatchProver(cd, 3).prove(pws)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert cd.verify(got[4])


def test_batch_reports_a_contradicting_witness(pkg, oracle):
    """A witness with a wrong side bit (the recomputed peak differs): plonky2 panics in generate_partial_witness; here that proof's
    status says so and the other proofs of the pass are unaffected."""
    cd, (leaf_t, proof_ts, peak_ts), cases, pws = circuit_and_witnesses(pkg, oracle, 20, range(570, 576))
    want = [cd.prove(pw) for pw in pws]
    leaf, sib, lefts, peaks, root = cases[2]
    flipped = lefts.copy()
    flipped[0] ^= 1
    bad = pkg.PartialWitness()
    assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, (leaf, sib, flipped, peaks, root), bad.set_target)
    with pytest.raises(pkg.P2mtPanic):
        cd.prove(bad)
    mixed = list(pws)
    mixed[2] = bad
    got, rc, status = pkg.BatchProver(cd, 6).prove(mixed, status=True)
    assert rc != 0 and status[2] != 0 and [s for i, s in enumerate(status) if i != 2] == [0] * 5
    for i, (g, w) in enumerate(zip(got, want)):
        if i != 2:
            assert np.array_equal(g, w)


def test_batch_rejects_what_it_cannot_batch(pkg, oracle):
    cd, (leaf_t, proof_ts, peak_ts), cases, pws = circuit_and_witnesses(pkg, oracle, 20, range(580, 583))
    bp = pkg.BatchProver(cd, 4)
    # witnesses that set their targets in another order
    other = pkg.PartialWitness()
    seq = []
    assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, cases[1], lambda t, v: seq.append((t, v)))
    for t, v in reversed(seq):
        other.set_target(t, v)
    with pytest.raises(pkg.P2mtPanic, match="same targets"):
        bp.prove([pws[0], other])
    assert np.array_equal(bp.prove(pws)[1], cd.prove(pws[1]))  # still usable
    with pytest.raises(pkg.P2mtPanic):
        pkg.BatchProver(cd, 0)


def test_batch_circuit_with_a_global_witness_table(pkg, oracle):
    """1500 path elements: 2^12 rows, the value table lives in global memory -- inside a batch every proof gets the one-workgroup
    interpreter (the grid-wide ones want the chip to themselves) and the streaming 2^12 LDE / 2^15 NTT kernels carry the batch."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 1500, range(590, 593))
    assert cd.info.degree_bits == 12
    want = [cd.prove(pw) for pw in pws]
    got = pkg.BatchProver(cd, 2).prove(pws)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert cd.verify(got[2])


def test_batch_of_recursion_proofs(pkg, oracle):
    """mmr_plonky2_verifier_1_recursion in batches: three statements -> three inner proofs in one pass -> three outer witnesses ->
    three outer proofs in passes of two; all equal to the one-at-a-time proofs (which test_recursion_gpu.py pins to the oracle)."""
    from circuit_cases import mmr_case
    cases = [mmr_case(oracle, 7, leaf) for leaf in (0, 1, 3)]  # 7 leaves: 3 peaks, leaves of the first mountain (2 path elements)
    n_sib, n_peaks = len(cases[0][1]), len(cases[0][3])
    cases = [c for c in cases if len(c[1]) == n_sib]
    assert len(cases) >= 2
    inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(n_sib, n_peaks)
    outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, n_peaks)
    pws = []
    for leaf, sib, lefts, peaks, root in cases:
        pw = pkg.PartialWitness()
        pw.set_target(leaf_t, leaf)
        for (ht, bt), s, l in zip(proof_ts, sib, lefts):
            pw.set_hash_target(ht, [int(x) for x in s])
            pw.set_target(bt, int(l))
        for i, pk in enumerate(peaks):
            for k in range(4):
                pw.set_target(inner.prover_only.public_inputs[4 * i + k], int(pk[k]))
        pws.append(pw)
    inner_want = [inner.prove(pw) for pw in pws]
    inner_got = pkg.BatchProver(inner, 4).prove(pws)
    for g, w in zip(inner_got, inner_want):
        assert np.array_equal(g, w)
    opws = []
    for (leaf, sib, lefts, peaks, root), ip in zip(cases, inner_got):
        pw = pkg.PartialWitness()
        pw.set_proof_with_pis_target(pt, ip)
        pw.set_verifier_data_target(vd, inner.verifier_only)
        for t, pk in zip(peak_ts, peaks):
            pw.set_hash_target(t, [int(x) for x in pk])
        for k, t in enumerate(outer.prover_only.public_inputs):
            pw.set_target(t, int(root[k]))
        opws.append(pw)
    outer_want = [outer.prove(pw) for pw in opws]
    outer_got = pkg.BatchProver(outer, 2).prove(opws)
    for g, w in zip(outer_got, outer_want):
        assert np.array_equal(g, w)
    assert outer.verify(outer_got[-1])


def test_batch_provers_on_two_threads(pkg, oracle):
    """Two host threads, each with its own stream, circuit handle and batch prover (the shape bench.py's throughput leg uses): the
    per-thread batch context and scratch keep the passes apart; all proofs equal the sequential ones."""
    import threading
    seeds = [range(600, 612), range(620, 632)]
    want, got, errs = [None, None], [None, None], []
    for i, sd in enumerate(seeds):
        cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, sd)
        want[i] = [cd.prove(pw) for pw in pws]

    def worker(i):
        try:
            pkg._native.check(pkg.lib().p2mt_thread_stream_create())
            cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, seeds[i])
            bp = pkg.BatchProver(cd, 8)
            for _ in range(3):
                got[i] = bp.prove(pws)
            del bp, cd, pws
            pkg._native.check(pkg.lib().p2mt_thread_stream_destroy())
        except Exception as e:  # surfaced below
            errs.append(repr(e))

    ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for i in range(2):
        for g, w in zip(got[i], want[i]):
            assert np.array_equal(g, w)


def test_verify_batch_matches_single_verify(pkg, oracle):
    """p2mt_circuit_verify_batch: accepted / reason per proof equal circuit_data.verify's, for good proofs and for every way of
    breaking one (non-canonical word, opening, proof-of-work witness, oracle Merkle path, layer Merkle path, final polynomial);
    more proofs than one pass holds (256) go through several passes."""
    cd, _, _, pws = circuit_and_witnesses(pkg, oracle, 20, range(640, 652))
    proofs = pkg.BatchProver(cd, 12).prove(pws)
    acc, why = cd.verify_batch(proofs)
    assert acc == [True] * 12 and why == [0] * 12
    info, plen = cd.info, cd.info.proof_len
    n_cs = info.num_selectors + 2 + 80
    n_open = n_cs + 135 + 2 * 2 + 2 * 9 + 16
    off_fri = 192 + 2 * n_open
    first_row = off_fri + 64          # one reduction at degree 2^6: the first query's constants_sigmas row
    bad = proofs.copy()
    bad[1, 5] = 0xFFFFFFFFFFFFFFFF                            # not canonical
    bad[2, 192 + 7] ^= 1                                      # an opening
    bad[3, plen - info.num_public_inputs - 1] ^= 1            # the proof-of-work witness
    bad[4, first_row + n_cs + 2] ^= 1                         # a sibling of an oracle row's Merkle path
    bad[5, plen - info.num_public_inputs - 3] ^= 1            # a coefficient of the final polynomial
    bad[6, first_row + 3] ^= 1                                # an opened oracle value
    acc, why = cd.verify_batch(bad)
    for i in range(12):
        a, r = cd.verify(bad[i], with_reason=True)
        assert (acc[i], why[i]) == (a, r), (i, acc[i], why[i], a, r)
    assert acc[0] and not any(acc[1:7]) and all(acc[7:])
    assert why[1] == 10 and why[2] == 11 and why[4] == 2
    # several passes, a single proof, an empty batch
    many = np.tile(proofs, (25, 1))[:290]
    many[277] = bad[2]
    acc, why = cd.verify_batch(many)
    assert acc == [i != 277 for i in range(290)] and why[277] == 11
    assert cd.verify_batch(proofs[:1]) == ([True], [0])
    assert cd.verify_batch(np.zeros((0, plen), np.uint64)) == ([], [])
