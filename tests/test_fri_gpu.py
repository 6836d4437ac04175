"""GPU parity for the opening proof (Challenger, opening evaluation, FRI prover) against oracle/fri.c, through the
C ABI.  PARITY UNPINNED with respect to plonky2 itself (no reference vector exists); against the oracle
the proof words are compared bit for bit, and the oracle's verifier restatement must accept the GPU's proof."""
import numpy as np
import pytest

import __graft_entry__ as ge
from fri_cases import P, make_instance, openings_of, oracle_commit, rand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def state_words(och):
    st = och.st
    return np.array(list(st.state) + list(st.inp) + list(st.out) + [st.n_in, st.n_out], np.uint64)


def assert_same_challenger(gch, och):
    g, o = gch.state(), state_words(och)
    n_in, n_out = int(o[28]), int(o[29])
    assert (int(g[28]), int(g[29])) == (n_in, n_out)
    assert np.array_equal(g[:12], o[:12])
    assert np.array_equal(g[12:12 + n_in], o[12:12 + n_in])
    assert np.array_equal(g[20:20 + n_out], o[20:20 + n_out])


def test_challenger_random_transcripts(pkg, oracle):
    rng = np.random.default_rng(7)
    gch, och = pkg.Challenger(), oracle.challenger()
    for step in range(60):
        if rng.integers(0, 2):
            k = int(rng.integers(1, 40))
            e = rand(k, 1000 + step)
            if step % 7 == 0:
                e[0] = np.uint64(P + 5)  # non-canonical input is canonicalised
            gch.observe_elements(e)
            och.observe(e)
        else:
            k = int(rng.integers(1, 20))
            assert np.array_equal(gch.get_n_challenges(k), och.get_n_challenges(k))
        if step % 10 == 0:
            assert_same_challenger(gch, och)
    assert_same_challenger(gch, och)


def test_challenger_clone_and_state_roundtrip(pkg, oracle):
    a = pkg.Challenger()
    a.observe_elements(np.arange(13, dtype=np.uint64))
    b = a.clone()
    x = a.get_n_challenges(5)
    assert np.array_equal(b.get_n_challenges(5), x)
    c = pkg.Challenger()
    c.set_state(a.state())
    assert np.array_equal(c.get_n_challenges(9), a.get_n_challenges(9))
    och = oracle.challenger()
    och.observe(np.arange(13, dtype=np.uint64))
    assert np.array_equal(och.get_n_challenges(5), x)
    with pytest.raises(pkg.P2mtPanic):
        bad = a.state()
        bad[28] = 8
        c.set_state(bad)


@pytest.mark.parametrize("log_n", [0, 1, 3, 6, 8, 9, 12, 14])
def test_eval_polys_ext_vs_oracle(pkg, oracle, log_n):
    c = rand((7, 1 << log_n), 300 + log_n)
    z = rand(2, 400 + log_n)
    assert np.array_equal(pkg.eval_polys_ext(c, z), oracle.eval_polys_ext(c, z))
    zb = np.array([z[0], 0], np.uint64)  # base-field point
    assert np.array_equal(pkg.eval_polys_ext(c, zb), oracle.eval_polys_ext(c, zb))


def run_case(pkg, oracle, degree_bits, n_polys, over, seed, pow_bits=6):
    oparams = oracle.fri_params_standard(degree_bits, proof_of_work_bits=pow_bits, **over)
    gparams = pkg.FriParams.standard(degree_bits, proof_of_work_bits=pow_bits, **over)
    assert bytes(oparams) == bytes(gparams)
    coeffs, batches = make_instance(oracle, degree_bits, n_polys, seed)
    ooracles, caps = oracle_commit(oracle, coeffs, oparams)
    goracles = [pkg.PolynomialBatch.from_coeffs(c, oparams.rate_bits, oparams.cap_height) for c in coeffs]
    for go, (c, leaves, dig) in zip(goracles, ooracles):
        assert np.array_equal(go.merkle_tree.leaves, leaves) and np.array_equal(go.merkle_tree.digests, dig)
    openings = openings_of(oracle, coeffs, batches)
    for g_open, o_open in zip(pkg.fri.openings(batches, goracles), openings):
        assert np.array_equal(g_open, o_open)
    och, gch = oracle.challenger(), pkg.Challenger()
    for ch_observe in (och.observe, gch.observe_elements):
        ch_observe(caps.reshape(-1))
        for o in openings:
            ch_observe(o.reshape(-1))
    overify = och.clone()
    want = oracle.fri_prove(ooracles, batches, oparams, och)
    got = pkg.prove_openings(batches, goracles, gch, gparams)
    assert got.size == want.size == pkg.fri.fri_proof_len(gparams, n_polys)
    if not np.array_equal(got, want):
        bad = np.flatnonzero(got != want)
        raise AssertionError("proof differs at %d words, first at %d of %d" % (bad.size, bad[0], got.size))
    assert_same_challenger(gch, och)
    ok, reason = oracle.fri_verify(n_polys, caps, batches, openings, oparams, overify, got)
    assert ok and reason == 0
    return got


@pytest.mark.parametrize("degree_bits,n_polys,over", [
    (6, [5, 9, 4, 3], {}),
    (5, [2], {}),
    (4, [3], {"reduction_arity_bits": [1]}),
    (8, [3, 2], {"reduction_arity_bits": [3, 1]}),
    (9, [4, 3], {"cap_height": 2, "num_query_rounds": 5}),
    (7, [2, 2], {"reduction_arity_bits": [4], "cap_height": 6}),
    (10, [40, 3], {"reduction_arity_bits": [2, 2, 2, 2]}),
    (12, [33, 7, 2], {}),
])
def test_fri_proof_vs_oracle(pkg, oracle, degree_bits, n_polys, over):
    run_case(pkg, oracle, degree_bits, n_polys, over, 500 + degree_bits)


def test_fri_config3_shape_full_pow(pkg, oracle):
    """Config 3: d = 6, oracles constants_sigmas/wires/zs_pp/quotient, standard 16-bit proof of work."""
    proof = run_case(pkg, oracle, 6, [84, 135, 20, 16], {}, 606, pow_bits=16)
    assert int(proof[-1]) < (1 << 30)


def test_fri_config4_shape(pkg, oracle):
    """Config 4's outer circuit (d = 12): 135 / 20 / 16 polynomials plus constants_sigmas, arities [4, 4]."""
    run_case(pkg, oracle, 12, [84, 135, 20, 16], {}, 1212, pow_bits=10)


def test_fri_exact_variants_and_forced_fallback(pkg, oracle):
    """The proof-of-work grind runs on the selected Poseidon variant; every variant must find the same witness."""
    lib = pkg.lib()
    try:
        for mds, partial, force in [(0, 0, 0), (1, 1, 0), (2, 0, 1)]:
            pkg.set_variant(mds, partial)
            lib.p2mt_debug_force_fallback(force)
            run_case(pkg, oracle, 6, [3, 2], {}, 707, pow_bits=8)
    finally:
        pkg.set_variant(2, 0)
        lib.p2mt_debug_force_fallback(0)


def test_fri_rejects_bad_arguments(pkg, oracle):
    params = pkg.FriParams.standard(6, proof_of_work_bits=4)
    coeffs, batches = make_instance(oracle, 6, [2], 808)
    go = [pkg.PolynomialBatch.from_coeffs(coeffs[0])]
    with pytest.raises(pkg.P2mtPanic):
        pkg.prove_openings([(batches[0][0], [(0, 5)])], go, pkg.Challenger(), params)  # polynomial index out of range
    with pytest.raises(pkg.P2mtPanic):
        pkg.prove_openings(batches, go, pkg.Challenger(), pkg.FriParams.standard(6, reduction_arity_bits=[5]))
    with pytest.raises(pkg.P2mtPanic):
        pkg.prove_openings(batches, go, pkg.Challenger(), pkg.FriParams.standard(13))


def test_gpu_matches_committed_fri_vectors(pkg, oracle, golden):
    """The committed vectors (tests/golden/fri_vectors.json) through the C ABI; inputs are rebuilt from the seeds."""
    g = golden["fri_vectors"]
    ch = pkg.Challenger()
    ch.observe_elements(g["challenger"]["observe_1"])
    assert [int(x) for x in ch.get_n_challenges(3)] == g["challenger"]["challenges_1"]
    ch.observe_elements(g["challenger"]["observe_2"])
    assert ch.get_challenge() == g["challenger"]["challenge_2"]
    e = g["eval_ext"]
    assert [int(x) for x in pkg.eval_polys_ext(np.array(e["coeffs"], np.uint64)[None], e["point"]).reshape(-1)] == e["value"]
    for case in g["fri"]:
        params = pkg.FriParams.standard(case["degree_bits"], **case["override"])
        coeffs, batches = make_instance(oracle, case["degree_bits"], case["n_polys"], case["seed"])
        goracles = [pkg.PolynomialBatch.from_coeffs(c, params.rate_bits, params.cap_height) for c in coeffs]
        caps = np.concatenate([o.merkle_tree.cap for o in goracles])
        assert [int(x) for x in caps.reshape(-1)] == case["caps"]
        ch = pkg.Challenger()
        ch.observe_cap(caps)
        for op, want in zip(pkg.fri.openings(batches, goracles), case["openings"]):
            assert [int(x) for x in op.reshape(-1)] == want
            ch.observe_extension_elements(op)
        proof = pkg.prove_openings(batches, goracles, ch, params)
        assert [int(x) for x in proof] == case["proof"]
        assert ch.get_challenge() == case["next_challenge"]
