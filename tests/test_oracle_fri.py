"""CPU-only checks of the FRI restatement (oracle/fri.c; parity unpinned -- the reference's tests only call verify):
the prover's proof is accepted by the verifier restatement, any single-word tampering is rejected, the challenger's
duplex rules hold, and the opening evaluation matches a direct Horner evaluation in Python integers."""
import numpy as np
import pytest

from fri_cases import P, make_instance, openings_of, oracle_commit, rand


def test_challenger_duplex_rules(oracle):
    """Overwrite-mode absorb at rate 8, outputs popped from the back, observing invalidates buffered outputs."""
    ch = oracle.challenger()
    ch.observe(np.arange(1, 4, dtype=np.uint64))
    st = np.zeros(12, np.uint64)
    st[:3] = [1, 2, 3]
    out = oracle.permute(st)
    assert ch.get_challenge() == int(out[7]) and ch.get_challenge() == int(out[6])
    ch.observe([5])  # buffered outputs are dropped; the next challenge re-permutes with state[0] overwritten
    st2 = out.copy()
    st2[0] = 5
    out2 = oracle.permute(st2)
    assert ch.get_challenge() == int(out2[7])
    # exactly 8 observed elements duplex at once and leave 8 fresh outputs
    ch2 = oracle.challenger()
    ch2.observe(np.arange(8, dtype=np.uint64))
    o = oracle.permute(np.concatenate([np.arange(8, dtype=np.uint64), np.zeros(4, np.uint64)]))
    assert [ch2.get_challenge() for _ in range(8)] == [int(x) for x in o[7::-1]]
    o2 = oracle.permute(o)  # ninth challenge: squeeze again
    assert ch2.get_challenge() == int(o2[7])


def test_eval_polys_ext_matches_python_ints(oracle):
    c = rand((3, 16), 5)
    z = rand(2, 6)

    def emul(x, y):
        return ((x[0] * y[0] + 7 * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)

    got = oracle.eval_polys_ext(c, z)
    for j in range(3):
        acc = (0, 0)
        for i in reversed(range(16)):
            acc = emul(acc, (int(z[0]), int(z[1])))
            acc = ((acc[0] + int(c[j, i])) % P, acc[1])
        assert (int(got[j, 0]), int(got[j, 1])) == acc


def test_ext_inverse(oracle):
    x = rand(2, 9)
    assert list(oracle.ext_mul(x, oracle.ext_inv(x))) == [1, 0]


@pytest.mark.parametrize("degree_bits,n_polys,over", [
    (6, [5, 9, 4, 3], {}),                                  # config 3 shape: 64 rows -> arities [4], final 4
    (5, [2], {}),                                           # no reduction at all: final poly only
    (8, [3, 2], {"reduction_arity_bits": [3, 1]}),          # mixed arities
    (9, [4, 3], {"cap_height": 2, "num_query_rounds": 5}),
    (7, [2, 2], {"reduction_arity_bits": [4], "cap_height": 6}),  # layer tree == cap (no siblings)
])
def test_prove_then_verify_and_tamper(oracle, degree_bits, n_polys, over):
    params = oracle.fri_params_standard(degree_bits, proof_of_work_bits=6, **over)
    coeffs, batches = make_instance(oracle, degree_bits, n_polys, 40 + degree_bits)
    oracles, caps = oracle_commit(oracle, coeffs, params)
    ch = oracle.challenger()
    ch.observe(caps.reshape(-1))
    openings = openings_of(oracle, coeffs, batches)
    for o in openings:
        ch.observe(o.reshape(-1))
    proof = oracle.fri_prove(oracles, batches, params, ch.clone())
    assert proof.size == oracle.fri_proof_len(params, n_polys)
    ok, reason = oracle.fri_verify(n_polys, caps, batches, openings, params, ch.clone(), proof)
    assert ok and reason == 0
    rng = np.random.default_rng(degree_bits)
    for pos in rng.integers(0, proof.size, size=60):
        bad = proof.copy()
        bad[pos] ^= np.uint64(1)
        ok2, reason2 = oracle.fri_verify(n_polys, caps, batches, openings, params, ch.clone(), bad)
        assert not ok2 and reason2 in (1, 2, 3, 4, 5)
    wrong = [o.copy() for o in openings]
    wrong[0][0, 0] ^= np.uint64(1)
    assert not oracle.fri_verify(n_polys, caps, batches, wrong, params, ch.clone(), proof)[0]
    # a different transcript (other challenger state) must not accept the same proof
    ch3 = ch.clone()
    ch3.observe([1])
    assert not oracle.fri_verify(n_polys, caps, batches, openings, params, ch3, proof)[0]


def test_standard_params_shapes(oracle):
    """d = 6 -> arities [4] (final poly 4 coefficients); d = 12 -> [4, 4] (final 16)."""
    p6, p12 = oracle.fri_params_standard(6), oracle.fri_params_standard(12)
    assert [p6.reduction_arity_bits[i] for i in range(p6.num_reductions)] == [4]
    assert [p12.reduction_arity_bits[i] for i in range(p12.num_reductions)] == [4, 4]
    assert (p6.rate_bits, p6.cap_height, p6.proof_of_work_bits, p6.num_query_rounds) == (3, 4, 16, 28)


def test_not_low_degree_is_rejected(oracle):
    """Leaves that are not the LDE of the claimed coefficients: the initial-tree openings disagree with the
    composition the prover folded, so the consistency check fails."""
    params = oracle.fri_params_standard(6, proof_of_work_bits=4)
    coeffs, batches = make_instance(oracle, 6, [3, 2], 77)
    oracles, caps = oracle_commit(oracle, coeffs, params)
    forged = [(rand(coeffs[0].shape, 78), oracles[0][1], oracles[0][2]), oracles[1]]  # other coefficients, same tree
    ch = oracle.challenger()
    ch.observe(caps.reshape(-1))
    openings = openings_of(oracle, [forged[0][0], coeffs[1]], batches)
    proof = oracle.fri_prove(forged, batches, params, ch.clone())
    ok, reason = oracle.fri_verify([3, 2], caps, batches, openings, params, ch.clone(), proof)
    assert not ok and reason == 3


def test_oracle_matches_committed_fri_vectors(oracle, golden):
    """tests/golden/fri_vectors.json (tools/gen_fri_golden.py): regression pin of the restatement."""
    g = golden["fri_vectors"]
    ch = oracle.challenger()
    ch.observe(g["challenger"]["observe_1"])
    assert [int(x) for x in ch.get_n_challenges(3)] == g["challenger"]["challenges_1"]
    ch.observe(g["challenger"]["observe_2"])
    assert ch.get_challenge() == g["challenger"]["challenge_2"]
    e = g["eval_ext"]
    assert [int(x) for x in oracle.eval_polys_ext(np.array(e["coeffs"], np.uint64)[None], e["point"]).reshape(-1)] == e["value"]
    for case in g["fri"]:
        params = oracle.fri_params_standard(case["degree_bits"], **case["override"])
        coeffs, batches = make_instance(oracle, case["degree_bits"], case["n_polys"], case["seed"])
        oracles, caps = oracle_commit(oracle, coeffs, params)
        assert [int(x) for x in caps.reshape(-1)] == case["caps"]
        ch = oracle.challenger()
        ch.observe(caps.reshape(-1))
        for op, want in zip(openings_of(oracle, coeffs, batches), case["openings"]):
            assert [int(x) for x in op.reshape(-1)] == want
            ch.observe(op.reshape(-1))
        proof = oracle.fri_prove(oracles, batches, params, ch)
        assert [int(x) for x in proof] == case["proof"]
        assert ch.get_challenge() == case["next_challenge"]
