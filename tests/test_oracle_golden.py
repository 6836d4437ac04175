"""Pin the oracle (oracle/*.c) against every known-answer vector the reference holds for the hot path
 and against SURVEY Appendix A check values. CPU-only."""
import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def u64(x):
    return np.asarray(x, dtype=np.uint64)


def test_field_order(golden):
    assert golden["reference_vectors"]["field_order"]["p"] == P


def test_round_constants_regenerate():
    """tools/gen_poseidon_constants.py --check: committed headers == ChaCha8 regeneration (sha256 pinned)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_poseidon_constants.py"), "--check"])


def test_permutation_kats(oracle):
    # SURVEY.md A.2 (first two equal upstream plonky2's published test vectors)
    z = oracle.permute([0] * 12)
    assert [int(x) for x in z] == [
        0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca, 0xd7709673896996dc,
        0x46a84e87642f44ed, 0xd032648251ee0b3c, 0x1c687363b207df62, 0xdf8565563e8045fe, 0x40f5b37ff4254dae,
        0xd070f637b431067c, 0x1792b1c4342109d7]
    s = oracle.permute(list(range(12)))
    assert [int(x) for x in s] == [
        0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2, 0x414754bfebd051f0,
        0xcb1f8980294a023f, 0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056, 0xf045d5eafdc6021f, 0xe4150f77caaa3be5,
        0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]
    m = oracle.permute([P - 1] * 12)
    assert [int(x) for x in m[:4]] == [0xbe0085cfc57a8357, 0xd95af71847d05c09, 0xcf55a13d33c1c953, 0x95803a74f4530e82]


def test_oracle_matches_python_spec(oracle):
    """oracle/poseidon.c vs the independent big-int python spec (tools/poseidon_spec.py), random + edge states."""
    import os, sys, random
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import poseidon_spec as ps
    rc = ps.round_constants()
    rnd = random.Random(11)
    cases = [[rnd.randrange(P) for _ in range(12)] for _ in range(6)]
    cases += [[P - 1] * 12, [0] * 12, [1 << 63] * 12, [0xFFFFFFFF] * 12, [0xFFFFFFFF00000000] * 12]
    for st in cases:
        assert [int(x) for x in oracle.permute(st)] == ps.poseidon_naive(st, rc)


def test_field_ops_vs_bigint(oracle):
    import random
    rnd = random.Random(5)
    edge = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63]
    vals = edge + [rnd.randrange(P) for _ in range(50)]
    for a in vals:
        for b in vals[:12]:
            assert oracle.mul(a, b) == a * b % P
            assert oracle.add(a, b) == (a + b) % P
            assert oracle.sub(a, b) == (a - b) % P
    assert oracle.mul(0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF) == (0xFFFFFFFFFFFFFFFF ** 2) % P  # non-canonical in
    assert oracle.root_of_unity(32) == 1753635133440165772
    assert oracle.pow(oracle.root_of_unity(32), 1 << 31) == P - 1
    assert oracle.root_of_unity(3) == P - (1 << 24)


@pytest.mark.parametrize("name", ["tree4", "tree16"])
def test_reference_tree_vectors(oracle, golden, name):
    """simple_merkle_tree.rs:136-140 and :181-190: every level and the root."""
    g = golden["reference_vectors"][name]
    k, levels, root = oracle.merkle_build(g["leaves"])
    exp = np.concatenate([u64(l) for l in g["levels"]])
    assert k == len(g["levels"])
    assert np.array_equal(levels, exp)
    assert np.array_equal(root, u64(g["root"]))


def test_reference_asserted_proof(oracle, golden):
    """simple_merkle_tree.rs:209-211 (the only assert!-ed hash values in the reference)."""
    g = golden["reference_vectors"]["tree4"]
    k, levels, root = oracle.merkle_build(g["leaves"])
    proof = oracle.merkle_get_proof(levels, 4, 0)
    assert np.array_equal(proof, u64(g["proof_leaf0"]))
    # test_verify_small_merkle_proof :214-233
    for i in (0, 3):
        assert oracle.verify_merkle_proof(g["leaves"][i], i, root, oracle.merkle_get_proof(levels, 4, i))


def test_verify_merkle_proof_16(oracle, golden):
    """simple_merkle_tree.rs:236-308 incl. the four negative cases :298-306."""
    g = golden["reference_vectors"]["tree16"]
    leaves = g["leaves"]
    k, levels, root = oracle.merkle_build(leaves)
    proofs = [oracle.merkle_get_proof(levels, 16, i) for i in range(16)]
    for i in range(16):
        assert oracle.verify_merkle_proof(leaves[i], i, root, proofs[i])
    assert not oracle.verify_merkle_proof(leaves[1], 0, root, proofs[0])  # wrong leaf
    assert not oracle.verify_merkle_proof(leaves[0], 1, root, proofs[0])  # wrong index
    assert not oracle.verify_merkle_proof(leaves[0], 0, root, proofs[1])  # wrong proof
    assert not oracle.verify_merkle_proof(leaves[0], 0, levels[0], proofs[0])  # wrong root
    # get_in_between_hashes :76-86
    ib = oracle.merkle_get_in_between_hashes(levels, root, 16, 5)
    assert ib.shape == (4, 4) and np.array_equal(ib[-1], root)
    assert np.array_equal(ib[0], u64(g["levels"][1][2]))


def test_merkle_build_panics(oracle):
    for bad in ([1], [1, 2, 3], []):  # Q6: 1 leaf underflows, non powers of two panic in log2_strict
        with pytest.raises(ValueError):
            oracle.merkle_build(bad)


def test_heights_bitmap_table(oracle, golden):
    for size, bitmap in golden["reference_vectors"]["heights_bitmap"]["pairs"]:
        assert oracle.heights_bitmap(size) == (bitmap, 0)
    assert oracle.heights_bitmap(0) == (0, 0)


def test_mmr_index_table(oracle, golden):
    for n, idx in golden["reference_vectors"]["mmr_index"]["pairs"]:
        assert oracle.get_mmr_index(n) == idx
    for n in (0, 1, 5, 777, 1 << 20, (1 << 20) + 12345):
        assert oracle.get_mmr_index(n) == 2 * n - bin(n).count("1")


def test_hash_modes(oracle):
    # SURVEY A.3
    assert np.array_equal(oracle.hash_no_pad(range(1, 9)), oracle.two_to_one([1, 2, 3, 4], [5, 6, 7, 8]))
    assert [int(x) for x in oracle.hash_no_pad(range(1, 9))] == [
        15064728126975588673, 10314245681893968020, 11300930272442645327, 2830815762300183090]
    assert [int(x) for x in oracle.hash_no_pad(range(1, 13))] == [
        1338892677694428047, 3607799255695052410, 2153232312043816145, 16174734614637570317]
    assert [int(x) for x in oracle.hash_or_noop([5])] == [5, 0, 0, 0]
    assert [int(x) for x in oracle.hash_or_noop([1, 2, 3, 4])] == [1, 2, 3, 4]
    assert np.array_equal(oracle.hash_or_noop([1, 2, 3, 4, 5]), oracle.hash_no_pad([1, 2, 3, 4, 5]))
    assert [int(x) for x in oracle.hash_no_pad([])] == [0, 0, 0, 0]
    # A.5 wide leaves
    assert [int(x) for x in oracle.hash_no_pad(range(135))] == [
        4848071992462728551, 7985168359107384293, 2979147297992328185, 11181256925898874940]
    assert [int(x) for x in oracle.hash_no_pad(range(20))] == [
        18012712284349310111, 2112131471180434530, 10118411046939476455, 16148679918091913951]
    assert [int(x) for x in oracle.hash_no_pad(range(16))] == [
        3047308842360922440, 10591378326149447922, 5991327740561014578, 5671799819667753500]


def test_mmr_small_goldens(oracle):
    """SURVEY A.4 (values from the verified restatement)."""
    k, levels, root = oracle.merkle_build(list(range(1024)))
    assert [int(x) for x in root] == [14342627526773219473, 1605964016051269283, 13081912992221981033,
                                      8024676129753453574]
    m = oracle.mmr(range(1, 8))
    assert len(m) == 11
    assert [[int(x) for x in p] for p in m.get_peaks()] == [
        [13574310676501394007, 16095010539665613421, 5677891464623049409, 6468220088955311139],
        [9783427051098178031, 17019276859948411944, 8215786202244449292, 10012800663576269083], [7, 0, 0, 0]]
    assert [int(x) for x in m.bagging_the_peaks()] == [9415449691735571594, 4029994303924475785,
                                                       1480575162239463180, 1589836677903401482]
    pr = m.get_proof(7)
    assert pr["lefts"].tolist() == [0] and pr["siblings"].tolist() == [[6, 0, 0, 0]]
    m8 = oracle.mmr(range(1, 9))
    assert len(m8) == 15
    assert [int(x) for x in m8.bagging_the_peaks()] == [13933167704481838627, 17422631469142100487,
                                                        16127254178629937220, 10088058593358781755]
    assert np.array_equal(m8.elements[-1], m8.bagging_the_peaks())
    m1k = oracle.mmr(range(1000))
    assert len(m1k) == 1994 and len(m1k.get_peaks()) == 6
    assert [int(x) for x in m1k.bagging_the_peaks()] == [13493064156419223771, 13520521149426597726,
                                                         15784220164657724348, 18117589472856137893]
    pr = m1k.get_proof_normal_index(777)
    assert pr["lefts"].tolist() == [1, 0, 0, 1, 0, 0, 0]
    assert pr["siblings"][0].tolist() == [776, 0, 0, 0]
    assert [int(x) for x in pr["siblings"][-1]] == [12102538752217177721, 384056655516491996,
                                                    15456118281923553820, 18216154410072501352]
    assert oracle.mmr_proof_verify(pr["siblings"], pr["lefts"], pr["peaks"], 777, m1k.bagging_the_peaks())


def test_mmr_equals_simple_tree_at_pow2(oracle):
    """A 2^k-leaf MMR has one peak equal to the simple tree's root (Q2) and post-order geometry (A.4)."""
    leaves = list(range(100, 164))
    k, levels, root = oracle.merkle_build(leaves)
    m = oracle.mmr(leaves)
    assert len(m) == 127
    assert np.array_equal(m.bagging_the_peaks(), root)
    el = m.elements
    # node of height h whose last leaf is L sits at 2L - popcount(L) + h
    off = 0
    for h in range(k):
        cnt = 64 >> h
        for j in range(cnt):
            last_leaf = (j + 1) * (1 << h) - 1
            pos = 2 * last_leaf - bin(last_leaf).count("1") + h
            assert np.array_equal(el[pos], levels[off + j])
        off += cnt


@pytest.mark.parametrize("n", list(range(1, 41)) + [100, 255, 256, 257])
def test_mmr_all_proofs_verify(oracle, n):
    """reference test helper mmr_plonky2_verifier.rs:102-117 (native part): every leaf's proof verifies."""
    leaves = [(i * 0x9E3779B97F4A7C15 + 12345) % P for i in range(n)]
    m = oracle.mmr(leaves)
    assert len(m) == 2 * n - bin(n).count("1")
    root = m.bagging_the_peaks()
    assert len(m.get_peaks()) == bin(n).count("1")
    for i in (range(n) if n <= 40 else (0, 1, n // 2, n - 1)):
        pr = m.get_proof_normal_index(i)
        assert pr["mmr_size"] == len(m)
        assert oracle.mmr_proof_verify(pr["siblings"], pr["lefts"], pr["peaks"], leaves[i], root)
        # wrong root => false; wrong leaf => the reference PANICS (assert at :245), Q5
        bad_root = root.copy(); bad_root[0] ^= np.uint64(1)
        assert not oracle.mmr_proof_verify(pr["siblings"], pr["lefts"], pr["peaks"], leaves[i], bad_root)
        if len(pr["siblings"]):
            with pytest.raises(AssertionError):
                oracle.mmr_proof_verify(pr["siblings"], pr["lefts"], pr["peaks"], (leaves[i] + 1) % P, root)


def test_mmr_empty_panics(oracle):
    m = oracle.mmr()
    with pytest.raises(OverflowError):
        m.get_peaks()  # Q6


def test_fft_conventions(oracle):
    """SURVEY A.5 [parity unpinned: conventions from recall]; identities checked independently."""
    a = list(range(1, 9))
    f = oracle.fft(a)
    assert [int(x) for x in f] == [36, 18445622567621360637, 18445618169507741693, 1130298020461564,
                                   18446744069414584317, 18445613771394122749, 1125899906842620, 1121501793223676]
    assert oracle.ifft(f).tolist() == a
    w = oracle.root_of_unity(3)
    for i in range(8):  # direct evaluation f(w^i)
        x = pow(w, i, P)
        assert int(f[i]) == sum(c * pow(x, j, P) for j, c in enumerate(a)) % P
    lde = oracle.coset_lde(a, 3)
    assert [int(x) for x in lde[:3]] == [7526268, 17426854749847130487, 15994200817435482274]
    assert int(lde[-1]) == 7380778535019697251
    w64 = oracle.root_of_unity(6)
    for i in (0, 1, 17, 63):
        x = 7 * pow(w64, i, P) % P
        assert int(lde[i]) == sum(c * pow(x, j, P) for j, c in enumerate(a)) % P


def test_polynomial_batch_commit_mini(oracle):
    """SURVEY A.5 mini PolynomialBatch::from_coeffs check values."""
    polys = np.array([[j + 1 + i for i in range(8)] for j in range(3)], dtype=np.uint64)
    leaves, digests, cap = oracle.polynomial_batch_commit(polys, False, 3, 2)
    assert [int(x) for x in leaves[1]] == [18446744069408729445, 18446744069408008845, 18446744069407288245]
    assert [int(x) for x in cap[0]] == [6767426713459994308, 4464047079632709065, 16885200355009179906,
                                        9438656522865686595]
    assert [int(x) for x in cap[3]] == [13903821440632216401, 12504631715270832321, 9898450178365270810,
                                        7220911749143292772]
    # from_values(fft(coeffs)) == from_coeffs(coeffs)
    vals = np.stack([oracle.fft(p) for p in polys])
    l2, d2, c2 = oracle.polynomial_batch_commit(vals, True, 3, 2)
    assert np.array_equal(l2, leaves) and np.array_equal(c2, cap) and np.array_equal(d2, digests)


def test_poseidon_gate_witness_consistency(oracle):
    """[parity unpinned layout] internal consistency of the PoseidonGate witness row against the pinned permutation:
    outputs == permute(swapped inputs); every recorded S-box input reproduces the next recorded value."""
    import random
    rnd = random.Random(2)
    for swap in (0, 1):
        st = [rnd.randrange(P) for _ in range(12)]
        w = [int(x) for x in oracle.poseidon_gate_witness(st, swap)]
        assert w[:12] == st and w[24] == swap
        sw = st[4:8] + st[:4] + st[8:] if swap else st
        assert w[25:29] == [((st[i + 4] - st[i]) % P) * swap for i in range(4)]
        assert w[12:24] == [int(x) for x in oracle.permute(sw)]
    # hash semantics the circuits rely on: two_to_one(l, r) = outputs[0..4] of the row with inputs [l, r, 0...] and
    # swap = 0; with swap = 1 it is two_to_one(r, l) (pick_hash's two options, mmr_plonky2_verifier.rs:46-54)
    l, r = [rnd.randrange(P) for _ in range(4)], [rnd.randrange(P) for _ in range(4)]
    w0 = oracle.poseidon_gate_witness(l + r + [0] * 4, 0)
    w1 = oracle.poseidon_gate_witness(l + r + [0] * 4, 1)
    assert np.array_equal(w0[12:16], oracle.two_to_one(l, r)) and np.array_equal(w1[12:16], oracle.two_to_one(r, l))


def test_parallel_cpu_baseline_equals_add_leaf_loop(oracle):
    """BASELINE.md B2 (all-core level-parallel build) produces the same array as the faithful add_leaf loop (B1)."""
    leaves = [(i * 0x9E3779B97F4A7C15 + 7) % P for i in range(1 << 11)]
    el, threads = oracle.mmr_build_pow2_parallel(leaves)
    assert threads >= 1
    assert np.array_equal(el, oracle.mmr(leaves).elements)


def test_fast_port_equals_spec_form(oracle):
    """oracle/poseidon_fast.c (bench.py's cpu_baseline.port_fast: sparse partial rounds, lazy reduction) is the same function as
    the spec-form restatement: permutation on edge values and random states, the add_leaf loop and the level-order build."""
    rng = np.random.default_rng(23)
    states = [np.zeros(12, np.uint64), np.full(12, P - 1, np.uint64), np.full(12, 2**64 - 1, np.uint64),
              np.arange(12, dtype=np.uint64)]
    states += [rng.integers(0, 2**64, size=12, dtype=np.uint64) for _ in range(200)]
    for s in states:
        assert np.array_equal(oracle.fast_permute(s), oracle.permute(s))
    leaves = rng.integers(0, P, size=(1 << 10) + 37, dtype=np.uint64)
    assert np.array_equal(oracle.fast_mmr_add_leaf_loop(leaves), oracle.mmr(leaves).elements)
    el, _ = oracle.fast_mmr_build_pow2(leaves[:1 << 10], 3)
    assert np.array_equal(el, oracle.mmr(leaves[:1 << 10]).elements)


def test_avx512_port_equals_spec_form(oracle):
    """oracle/poseidon_avx512.c (bench.py's cpu_baseline.port_fast on hosts with AVX-512: eight permutations per zmm lane set) is
    the same function as the spec-form restatement: edge values in every lane position, random states, ragged batch sizes, and
    the level-order MMR build."""
    import oracle_lib
    lib = oracle_lib.build_avx512_port_native()
    if lib is None:
        pytest.skip("no AVX-512 on this host")
    rng = np.random.default_rng(29)
    edge = [np.zeros(12, np.uint64), np.full(12, P - 1, np.uint64), np.full(12, 2**64 - 1, np.uint64), np.full(12, P, np.uint64),
            np.arange(12, dtype=np.uint64), np.full(12, 2**32 - 1, np.uint64), np.full(12, 2**63, np.uint64)]
    states = np.stack(edge + [rng.integers(0, 2**64, size=12, dtype=np.uint64) for _ in range(300)])
    for n in (1, 7, 8, 9, len(states)):
        got = oracle_lib.avx512_permute_batch(lib, states[:n])
        for g, s in zip(got, states[:n]):
            assert np.array_equal(g, oracle.permute(s))
    leaves = rng.integers(0, 2**64, size=1 << 10, dtype=np.uint64)
    for threads in (1, 3):
        el, _ = oracle_lib.avx512_mmr_build_pow2(lib, leaves, threads)
        assert np.array_equal(el, oracle.mmr(leaves % np.uint64(P)).elements)
    el, _ = oracle_lib.avx512_mmr_build_pow2(lib, leaves[:4], 1)  # fewer nodes per level than lanes
    assert np.array_equal(el, oracle.mmr(leaves[:4] % np.uint64(P)).elements)


def test_plonky2_digest_layout_restatement(oracle):
    """oracle/merkle_cap.py (plonky2's fill_subtree order + MerkleTree::prove's index formula) agrees with the level-major cap
    tree of the C oracle: same cap, and every Merkle path read through plonky2's indexing equals the level-major walk."""
    from oracle import merkle_cap as MC
    rng = np.random.default_rng(2)
    for n, w, cap in ((64, 3, 2), (64, 9, 4), (16, 5, 4), (32, 20, 0)):
        leaves = rng.integers(0, P, size=(n, w), dtype=np.uint64)
        d, c = MC.merkle_tree_new(oracle, leaves, cap)
        dig, capo = oracle.merkle_cap_commit(leaves, cap)
        assert np.array_equal(capo, c) and d.shape[0] == 2 * (n - (1 << cap))
        k = n.bit_length() - 1
        for leaf in range(n):
            off, idx, exp = 0, leaf, []
            for j in range(k - cap):
                exp.append(dig[off + (idx ^ 1)])
                off += n >> j
                idx >>= 1
            assert np.array_equal(MC.prove(d, n, cap, leaf), np.array(exp, np.uint64).reshape(-1, 4))
