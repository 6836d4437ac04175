"""GPU parity for the commit step (NTT / IFFT / coset LDE / wide-leaf sponge / Merkle cap /
PolynomialBatch) against the oracle.  The oracle's conventions here are PARITY UNPINNED (no reference vector
exists); values are exact field elements, so agreement is bit-exact."""
import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def rand(shape, seed):
    return np.random.default_rng(seed).integers(0, P, size=shape, dtype=np.uint64)


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 6, 9, 11, 12, 13, 14, 15, 16, 17, 18, 19])
def test_fft_ifft_vs_oracle(pkg, oracle, log_n):
    """(2^13 points and up: the two-pass four-step path -- 1024-point column pass + the row pass for 2^(log_n - 10)-point rows)"""
    n_polys = 5 if log_n < 13 else 2
    a = rand((n_polys, 1 << log_n), 100 + log_n)
    a[0, :] = np.arange(1, (1 << log_n) + 1, dtype=np.uint64)
    f = pkg.fft(a)
    for j in range(n_polys):
        assert np.array_equal(f[j], oracle.fft(a[j]))
    g = pkg.ifft(a)
    for j in range(n_polys):
        assert np.array_equal(g[j], oracle.ifft(a[j]))
    assert np.array_equal(pkg.ifft(f), a)  # round trip


def test_fft_survey_kat(pkg):
    f = pkg.fft(np.arange(1, 9, dtype=np.uint64)[None])[0]
    assert [int(x) for x in f] == [36, 18445622567621360637, 18445618169507741693, 1130298020461564,
                                   18446744069414584317, 18445613771394122749, 1125899906842620, 1121501793223676]


def test_fft_noncanonical_inputs(pkg, oracle):
    a = np.array([[P, P + 1, 0xFFFFFFFFFFFFFFFF, 0, 1, P - 1, 7, 1 << 63]], dtype=np.uint64)
    assert np.array_equal(pkg.fft(a)[0], oracle.fft(a[0]))


@pytest.mark.parametrize("log_n,rate_bits", [(0, 3), (1, 3), (3, 3), (6, 3), (9, 3), (12, 3), (5, 1), (4, 0), (6, 4)])
def test_coset_lde_vs_oracle(pkg, oracle, log_n, rate_bits):
    c = rand((4, 1 << log_n), 200 + log_n)
    c[0, :] = np.arange(1, (1 << log_n) + 1, dtype=np.uint64)
    out = pkg.coset_lde(c, rate_bits)
    for j in range(4):
        assert np.array_equal(out[j], oracle.coset_lde(c[j], rate_bits))


@pytest.mark.parametrize("log_n,rate_bits", [(13, 3), (14, 1), (15, 2)])
def test_coset_lde_above_one_workgroup(pkg, oracle, log_n, rate_bits):
    """transforms that do not fit a workgroup's LDS (the general path: per-coset scaling into leaf-order rows + one in-place DIF)"""
    c = rand((3, 1 << log_n), 500 + log_n)
    c[0, :] = np.arange(1, (1 << log_n) + 1, dtype=np.uint64)
    out = pkg.coset_lde(c, rate_bits)
    for j in range(3):
        assert np.array_equal(out[j], oracle.coset_lde(c[j], rate_bits))
    pb = pkg.PolynomialBatch.from_coeffs(c[:2, :1 << 13] if log_n > 13 else c[:2], rate_bits, 4, want_leaves=False)
    _, _, cap = oracle.polynomial_batch_commit(c[:2, :1 << 13] if log_n > 13 else c[:2], False, rate_bits, 4)
    assert np.array_equal(pb.merkle_tree.cap, cap)


def test_coset_lde_2pow12_fast_path_and_forced_redo(pkg, oracle):
    """k_coset_lde12_v2 (shift-only radix-16 passes, flag-form arithmetic): 19 polynomials (two full groups of 8 in the XCD-aware
    block order + a partial group in plain order), non-canonical and extreme coefficients among them, against the oracle, every word;
    then the same with the flagged-workgroup exact radix-2 redo forced for every workgroup."""
    c = rand((19, 4096), 77)
    c[1, :] = np.arange(1, 4097, dtype=np.uint64)
    ext = np.array([P, P + 1, 0xFFFFFFFFFFFFFFFF, 0, 1, P - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1 << 63, 0xFFFFFFFEFFFFFFFF], dtype=np.uint64)
    c[2, :] = np.resize(ext, 4096)
    want = [oracle.coset_lde(c[j], 3) for j in range(19)]
    for force in (0, 1):
        pkg.lib().p2mt_debug_force_fallback(force)
        try:
            out = pkg.coset_lde(c, 3)
        finally:
            pkg.lib().p2mt_debug_force_fallback(0)
        for j in range(19):
            assert np.array_equal(out[j], want[j]), (force, j)


def test_large_fft_properties(pkg):
    """2^20-point transforms (the four-step path): round trip, linearity and a direct evaluation."""
    log_n = 20
    a, b = rand((1, 1 << log_n), 31), rand((1, 1 << log_n), 32)
    fa, fb = pkg.fft(a), pkg.fft(b)
    assert np.array_equal(pkg.ifft(fa), a)
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    fs = pkg.fft(s)
    assert np.array_equal(fs, ((fa.astype(object) + fb.astype(object)) % P).astype(np.uint64))
    assert int(fa[0, 0]) == int(sum(int(x) for x in a[0]) % P)  # f(1) = sum of coefficients


def test_four_step_column_pass_forced_redo_small_rows(pkg, oracle):
    """the column pass's exact redo with 8- and 16-column tiles (rows of 8 and of 64 points)"""
    for log_n in (13, 16):
        a = rand((2, 1 << log_n), 900 + log_n)
        pkg.lib().p2mt_debug_force_fallback(1)
        try:
            f, g = pkg.fft(a), pkg.ifft(a)
        finally:
            pkg.lib().p2mt_debug_force_fallback(0)
        for j in range(2):
            assert np.array_equal(f[j], oracle.fft(a[j])) and np.array_equal(g[j], oracle.ifft(a[j])), (log_n, j)


def test_four_step_fft_2pow20_vs_oracle(pkg, oracle):
    """The two-pass (1024 x 1024) path of p2mt_ntt_batch_dev against the oracle's fft / ifft, every word: random rows, a row of
    non-canonical and extreme values (the loose-u64 arithmetic must reduce them as the reference's field type does), and the same
    again with the flagged-tile exact radix-2 redo forced for every tile."""
    log_n = 20
    a = rand((3, 1 << log_n), 41)
    a[1, :] = np.arange(1, (1 << log_n) + 1, dtype=np.uint64)
    ext = np.array([P, P + 1, 0xFFFFFFFFFFFFFFFF, 0, 1, P - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1 << 63, 0xFFFFFFFEFFFFFFFF],
                   dtype=np.uint64)
    a[2, :] = np.resize(ext, 1 << log_n)
    want_f = [oracle.fft(a[j]) for j in range(3)]
    want_i = [oracle.ifft(a[j]) for j in range(3)]
    for force in (0, 1):
        pkg.lib().p2mt_debug_force_fallback(force)
        try:
            f, g = pkg.fft(a), pkg.ifft(a)
        finally:
            pkg.lib().p2mt_debug_force_fallback(0)
        for j in range(3):
            assert np.array_equal(f[j], want_f[j]), (force, j)
            assert np.array_equal(g[j], want_i[j]), (force, j)


@pytest.mark.parametrize("width", [1, 3, 4, 5, 8, 9, 16, 20, 135])
@pytest.mark.parametrize("cap_height", [0, 2, 4])
def test_merkle_cap_commit_vs_oracle(pkg, oracle, width, cap_height):
    leaves = rand((64, width), 300 + width)
    t = pkg.MerkleCapTree.new(leaves, cap_height)
    digests, cap = oracle.merkle_cap_commit(leaves, cap_height)
    assert np.array_equal(t.cap, cap)
    assert np.array_equal(t.digests, digests)


def test_merkle_cap_edge_shapes(pkg, oracle):
    leaves = rand((16, 7), 77)
    for cap_height in (4,):  # cap == leaf digests
        t = pkg.MerkleCapTree.new(leaves, cap_height)
        _, cap = oracle.merkle_cap_commit(leaves, cap_height)
        assert np.array_equal(t.cap, cap) and t.digests.shape[0] == 0
    with pytest.raises(pkg.P2mtPanic):
        pkg.MerkleCapTree.new(leaves, 5)
    with pytest.raises(pkg.P2mtPanic):
        pkg.MerkleCapTree.new(rand((12, 3), 1), 2)


def test_commit_all_poseidon_variants(pkg, oracle):
    polys = rand((20, 64), 555)
    _, _, cap = oracle.polynomial_batch_commit(polys, True, 3, 4)
    big = rand((9, 4096), 556)  # 2^15 leaves: quad-lane sponge + quad/wave cap levels
    _, _, cap_big = oracle.polynomial_batch_commit(big, False, 3, 4)
    try:
        for v in [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]:
            pkg.set_variant(*v)
            assert np.array_equal(pkg.PolynomialBatch.from_values(polys).merkle_tree.cap, cap)
        assert np.array_equal(pkg.PolynomialBatch.from_coeffs(big, want_leaves=False).merkle_tree.cap, cap_big)
        pkg.lib().p2mt_debug_force_fallback(1)
        assert np.array_equal(pkg.PolynomialBatch.from_values(polys).merkle_tree.cap, cap)
        assert np.array_equal(pkg.PolynomialBatch.from_coeffs(big, want_leaves=False).merkle_tree.cap, cap_big)
    finally:
        pkg.lib().p2mt_debug_force_fallback(0)
        pkg.set_variant(2, 0)


def test_polynomial_batch_mini_kat(pkg):
    """SURVEY A.5 mini from_coeffs."""
    polys = np.array([[j + 1 + i for i in range(8)] for j in range(3)], dtype=np.uint64)
    pb = pkg.PolynomialBatch.from_coeffs(polys, 3, 2)
    assert [int(x) for x in pb.merkle_tree.leaves[1]] == [18446744069408729445, 18446744069408008845,
                                                          18446744069407288245]
    assert [int(x) for x in pb.merkle_tree.cap[0]] == [6767426713459994308, 4464047079632709065,
                                                       16885200355009179906, 9438656522865686595]
    assert [int(x) for x in pb.merkle_tree.cap[3]] == [13903821440632216401, 12504631715270832321,
                                                       9898450178365270810, 7220911749143292772]


# config 3 (d = 6) and config 4 outer circuit (d = 12) commit shapes: wires 135, Z/partial products 20, quotient 16
@pytest.mark.parametrize("n_polys,log_n,is_values", [(135, 6, True), (20, 6, True), (16, 6, False), (2, 3, True),
                                                     (135, 12, True), (20, 12, True), (16, 12, False)])
def test_polynomial_batch_commit_vs_oracle(pkg, oracle, n_polys, log_n, is_values):
    polys = rand((n_polys, 1 << log_n), 400 + n_polys + log_n)
    pb = (pkg.PolynomialBatch.from_values if is_values else pkg.PolynomialBatch.from_coeffs)(polys)
    leaves, digests, cap = oracle.polynomial_batch_commit(polys, is_values, 3, 4)
    assert np.array_equal(pb.merkle_tree.cap, cap)
    assert np.array_equal(pb.merkle_tree.leaves, leaves)
    assert np.array_equal(pb.merkle_tree.digests, digests)
    # without materialising leaves the cap is identical
    pb2 = (pkg.PolynomialBatch.from_values if is_values else pkg.PolynomialBatch.from_coeffs)(polys, want_leaves=False)
    assert np.array_equal(pb2.merkle_tree.cap, cap)
    # a Merkle path re-hashed with the independent two_to_one batch kernel reaches the right cap entry
    idx = 5 % leaves.shape[0]
    path = pb.merkle_tree.prove(idx)
    cur = pkg.hash_or_noop(leaves[idx])
    i = idx
    for sib in path:
        cur = pkg.two_to_one(cur, sib) if i % 2 == 0 else pkg.two_to_one(sib, cur)
        i >>= 1
    assert np.array_equal(cur, cap[i])
