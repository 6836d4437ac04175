"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the committed golden fixtures.
Bit-exact everywhere (integer work).  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from conftest import splitmix_leaves

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
VARIANTS = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]
DEFAULT_VARIANT = (2, 0)


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def edge_states():
    e = [0, 1, P - 1, P, P + 1, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63]
    rows = [[v] * 12 for v in e]
    rows += [[e[(i + j) % len(e)] for j in range(12)] for i in range(len(e))]
    return np.array(rows, dtype=np.uint64)


@pytest.mark.parametrize("variant", VARIANTS)
def test_permutation_all_variants(pkg, oracle, variant):
    pkg.set_variant(*variant)
    try:
        rng = np.random.default_rng(3)
        states = np.concatenate([edge_states(), rng.integers(0, 1 << 64, size=(1500, 12), dtype=np.uint64),
                                 np.arange(12, dtype=np.uint64)[None]])
        got = pkg.poseidon_permute_batch(states)
        exp = oracle.permute_batch(states)
        assert np.array_equal(got, exp)
        assert [int(x) for x in got[-1][:2]] == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47]  # SURVEY A.2 KAT
    finally:
        pkg.set_variant(*DEFAULT_VARIANT)


@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_golden_trees(pkg, golden, variant):
    """simple_merkle_tree.rs:136-140, :181-190, :210-211 on the GPU."""
    pkg.set_variant(*variant)
    try:
        for name in ("tree4", "tree16"):
            g = golden["reference_vectors"][name]
            t = pkg.MerkleTree.build(g["leaves"])
            assert t.count_levels == len(g["levels"])
            for lvl, exp in zip(t.tree, g["levels"]):
                assert np.array_equal(lvl, np.asarray(exp, dtype=np.uint64))
            assert np.array_equal(t.root, np.asarray(g["root"], dtype=np.uint64))
        g = golden["reference_vectors"]["tree4"]
        t = pkg.MerkleTree.build(g["leaves"])
        assert np.array_equal(t.get_merkle_proof(0), np.asarray(g["proof_leaf0"], dtype=np.uint64))
    finally:
        pkg.set_variant(*DEFAULT_VARIANT)


def test_fast_path_fallback_is_exact(pkg, oracle):
    """The shipped (mds=2) path has a rare-event fallback; forcing every wave through it must not change a bit,
    and the primitives' rare cases are hit with crafted operands (products whose reduction borrows/wraps)."""
    pkg.set_variant(2, 0)
    lib = pkg.lib()
    rng = np.random.default_rng(8)
    states = np.concatenate([edge_states(), rng.integers(0, 1 << 64, size=(3000, 12), dtype=np.uint64)])
    # 2^17 + 4099 leaves: exercises the fused tiles, the lane-, quad- and wave-per-node level kernels
    leaves = splitmix_leaves((1 << 17) + 4099, 0x5EED00AA)
    tree_leaves = splitmix_leaves(1 << 15, 0x5EED00AB)
    try:
        fast = pkg.poseidon_permute_batch(states)
        m_fast = pkg.MMR.from_leaves(leaves).elements
        t_fast = pkg.MerkleTree.build(tree_leaves)
        lib.p2mt_debug_force_fallback(1)
        slow = pkg.poseidon_permute_batch(states)
        m_slow = pkg.MMR.from_leaves(leaves).elements
        t_slow = pkg.MerkleTree.build(tree_leaves)
    finally:
        lib.p2mt_debug_force_fallback(0)
    assert np.array_equal(fast, slow) and np.array_equal(fast, oracle.permute_batch(states))
    assert np.array_equal(m_fast, m_slow) and np.array_equal(m_fast, oracle.mmr(leaves).elements)
    assert np.array_equal(t_fast._flat, t_slow._flat) and np.array_equal(t_fast.root, t_slow.root)
    assert np.array_equal(t_fast.root, oracle.merkle_build(tree_leaves)[2])


def test_fast_path_many_random_states(pkg, oracle):
    """2^17 random permutations: at ~2^-22 per MDS row the sticky fallback fires for a few waves here."""
    pkg.set_variant(2, 0)
    rng = np.random.default_rng(21)
    states = rng.integers(0, 1 << 64, size=(1 << 17, 12), dtype=np.uint64)
    got = pkg.poseidon_permute_batch(states)
    pkg.set_variant(0, 0)
    try:
        ref = pkg.poseidon_permute_batch(states)
    finally:
        pkg.set_variant(*DEFAULT_VARIANT)
    assert np.array_equal(got, ref)
    sel = rng.integers(0, 1 << 17, size=300)
    assert np.array_equal(got[sel], oracle.permute_batch(states[sel]))


def test_two_to_one_and_hash_modes(pkg, oracle):
    rng = np.random.default_rng(4)
    pairs = rng.integers(0, 1 << 64, size=(300, 8), dtype=np.uint64)
    got = pkg.two_to_one_batch(pairs)
    for i in range(0, 300, 7):
        assert np.array_equal(got[i], oracle.two_to_one(pairs[i, :4], pairs[i, 4:]))
    big = rng.integers(0, 1 << 64, size=(4096 + 77, 8), dtype=np.uint64)  # >= 4096 pairs: the matrix-pipe kernel, ragged last wave
    got = pkg.two_to_one_batch(big)
    for i in list(range(0, 4096, 97)) + list(range(4096, 4096 + 77)):
        assert np.array_equal(got[i], oracle.two_to_one(big[i, :4], big[i, 4:]))
    assert [int(x) for x in pkg.two_to_one([2890852870, 0, 0, 0], [156728478, 0, 0, 0])] == [
        6678006133445961348, 15827935749738443865, 6295652393730592048, 1546515167911236130]  # reference :138
    for length in (1, 2, 4, 5, 7, 8, 9, 12, 16, 17, 20, 64, 135):
        rows = rng.integers(0, 1 << 64, size=(70, length), dtype=np.uint64)
        rows[0] = np.arange(length, dtype=np.uint64)
        a, b = pkg.hash_or_noop_batch(rows), pkg.hash_no_pad_batch(rows)
        for i in range(0, 70, 9):
            assert np.array_equal(a[i], oracle.hash_or_noop(rows[i]))
            assert np.array_equal(b[i], oracle.hash_no_pad(rows[i]))
    assert [int(x) for x in pkg.hash_no_pad(range(135))] == [4848071992462728551, 7985168359107384293,
                                                             2979147297992328185, 11181256925898874940]  # A.5


@pytest.mark.parametrize("log_n", [1, 2, 5, 10, 13])
def test_merkle_tree_vs_oracle(pkg, oracle, log_n):
    """config 1 shape (2^10) and neighbours: all levels, root, proofs, verify incl. negatives."""
    n = 1 << log_n
    leaves = splitmix_leaves(n, 0x5EED0001)
    t = pkg.MerkleTree.build(leaves)
    k, levels, root = oracle.merkle_build(leaves)
    assert t.count_levels == k
    assert np.array_equal(t._flat[:2 * n - 2], levels[:2 * n - 2])
    assert np.array_equal(t.root, root)
    idxs = sorted(set([0, 1, n // 2, n - 1]))
    for i in idxs:
        pr = t.get_merkle_proof(i)
        assert np.array_equal(pr, oracle.merkle_get_proof(levels, n, i))
        assert pkg.verify_merkle_proof(leaves[i], i, t.root, pr)
        assert not pkg.verify_merkle_proof((int(leaves[i]) + 1) % P, i, t.root, pr)  # wrong leaf
        assert not pkg.verify_merkle_proof(leaves[i], i ^ 1, t.root, pr)             # wrong index
        assert not pkg.verify_merkle_proof(leaves[i], i, t.tree[0][0], pr)           # wrong root


@pytest.mark.parametrize("log_n,levels", [(18, 2), (21, 2), (23, 3)])
def test_merkle_tree_large_subtree_stage(pkg, oracle, log_n, levels):
    """Large MerkleTree::build: stage 1 as per-lane subtrees of 2^levels leaves (k_merkle_subtree: leaf digests and levels 1..levels in
    one barrier-free launch, level-major slots, matrix-pipe MDS), the level kernels above it.  Every level and the root against the oracle,
    proofs for the corner leaves."""
    n = 1 << log_n
    leaves = splitmix_leaves(n, 0x5EED0600 + log_n)
    t = pkg.MerkleTree.build(leaves)
    k, lv, root = oracle.merkle_build(leaves)
    assert t.count_levels == k
    assert np.array_equal(t._flat[:2 * n - 2], lv[:2 * n - 2])
    assert np.array_equal(t.root, root)
    for i in (0, 1, n // 2 + 5, n - 1):
        pr = t.get_merkle_proof(i)
        assert np.array_equal(pr, oracle.merkle_get_proof(lv, n, i))
        assert pkg.verify_merkle_proof(leaves[i], i, t.root, pr)


def test_merkle_tree_1024_golden(pkg):
    t = pkg.MerkleTree.build(np.arange(1024, dtype=np.uint64))
    assert [int(x) for x in t.root] == [14342627526773219473, 1605964016051269283, 13081912992221981033,
                                        8024676129753453574]


def test_merkle_build_panics(pkg):
    for bad in ([1], [1, 2, 3], [], list(range(12))):
        with pytest.raises(pkg.P2mtPanic):
            pkg.MerkleTree.build(bad)


@pytest.mark.parametrize("n", list(range(1, 34)) + [63, 64, 65, 100, 255, 256, 257, 1000, 4099])
def test_mmr_elements_vs_oracle(pkg, oracle, n):
    leaves = splitmix_leaves(n, 0x5EED0002 + n)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    assert len(m) == len(om) == 2 * n - bin(n).count("1")
    assert np.array_equal(m.elements, om.elements)
    assert np.array_equal(m.get_peaks(), om.get_peaks())
    assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())


def test_mmr_small_goldens(pkg):
    m = pkg.MMR.from_leaves(np.arange(1000, dtype=np.uint64))
    assert len(m) == 1994
    assert [int(x) for x in m.bagging_the_peaks()] == [13493064156419223771, 13520521149426597726,
                                                       15784220164657724348, 18117589472856137893]
    pr = m.get_proof_normal_index(777)
    assert pr.lefts.tolist() == [1, 0, 0, 1, 0, 0, 0]
    assert pr.siblings[0].tolist() == [776, 0, 0, 0]
    assert [int(x) for x in pr.siblings[-1]] == [12102538752217177721, 384056655516491996, 15456118281923553820,
                                                 18216154410072501352]
    m7 = pkg.MMR.from_leaves(range(1, 8))
    assert [int(x) for x in m7.bagging_the_peaks()] == [9415449691735571594, 4029994303924475785,
                                                        1480575162239463180, 1589836677903401482]


def test_mmr_incremental_equals_bulk(pkg, oracle):
    """add_leaf one by one, ragged extends and one bulk extend give the same elements (== the oracle's)."""
    leaves = splitmix_leaves(777, 0x5EED0003)
    om = oracle.mmr(leaves)
    a = pkg.MMR.new()
    for v in leaves[:70]:
        a.add_leaf(int(v))
    assert np.array_equal(a.elements, oracle.mmr(leaves[:70]).elements)
    b = pkg.MMR.new()
    cuts = [0, 1, 2, 5, 64, 65, 130, 500, 501, 777]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        b.extend(leaves[lo:hi])
    assert np.array_equal(b.elements, om.elements)
    b.extend([])  # empty extend is a no-op
    assert len(b) == len(om)
    b.reset()
    assert len(b) == 0
    b.extend(leaves)
    assert np.array_equal(b.elements, om.elements)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 16, 31, 70, 1031])
def test_mmr_proofs_vs_oracle(pkg, oracle, n):
    """get_proof / verify for the leaf counts the reference's own tests use (mmr_plonky2_verifier.rs:153-209,
    _1_recursion.rs:223-257), incl. the panic-on-wrong-leaf quirk (Q5)."""
    leaves = splitmix_leaves(n, 0x5EED0004 + n)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    root = m.bagging_the_peaks()
    idxs = range(n) if n <= 70 else [0, 1, 100, 512, 1023, 1024, 1030]
    for i in idxs:
        pr = m.get_proof_normal_index(i)
        opr = om.get_proof_normal_index(i)
        assert pr.mmr_size == opr["mmr_size"]
        assert np.array_equal(pr.siblings, opr["siblings"])
        assert np.array_equal(pr.lefts, opr["lefts"])
        assert np.array_equal(pr.peaks, opr["peaks"])
        assert pr.verify(int(leaves[i]), root)
        bad_root = root.copy()
        bad_root[3] ^= np.uint64(1)
        assert not pr.verify(int(leaves[i]), bad_root)
        if len(pr.lefts):
            with pytest.raises(pkg.P2mtPanic) as e:
                pr.verify((int(leaves[i]) + 1) % P, root)
            assert e.value.code == -5
    # proofs for non-leaf / arbitrary mmr indices follow the reference's walk too
    for idx in range(len(om)):
        if n <= 31:
            pr, opr = m.get_proof(idx), om.get_proof(idx)
            assert np.array_equal(pr.siblings, opr["siblings"]) and np.array_equal(pr.lefts, opr["lefts"])
    with pytest.raises(pkg.P2mtPanic):
        m.get_proof(len(om))


def test_mmr_tiled_build_ragged(pkg, oracle):
    """Sizes and extend cuts that straddle the fused-tile boundaries (2^11 leaves, 2^16 leaves)."""
    n = (1 << 17) + 4099
    leaves = splitmix_leaves(n, 0x5EED0077)
    om = oracle.mmr(leaves)
    a = pkg.MMR.from_leaves(leaves)
    assert np.array_equal(a.elements, om.elements)
    b = pkg.MMR.new()
    cuts = [0, 3, 2047, 2049, 6144, 65535, 65537, 70000, (1 << 17) - 1, (1 << 17) + 1, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        b.extend(leaves[lo:hi])
    assert np.array_equal(b.elements, om.elements)
    assert np.array_equal(b.bagging_the_peaks(), om.bagging_the_peaks())


def test_stage1_sparse_partial_rounds_variant(pkg, oracle):
    """Variant (2, 1): the stage-1 MMR kernel with the fast path's SPARSE partial rounds (plonky2's mds_partial_layer_fast in
    flag-form arithmetic; an A/B that lost by 2.3 %, profiles/r02_sparse_flag_form_ab.txt, kept selectable) is the same function:
    all nodes equal the oracle's, with and without the forced exact redo."""
    n = (1 << 16) + 4099
    leaves = splitmix_leaves(n, 0x5EED0177)
    om = oracle.mmr(leaves)
    try:
        pkg.set_variant(2, 1)
        a = pkg.MMR.from_leaves(leaves)
        assert np.array_equal(a.elements, om.elements)
        pkg.lib().p2mt_debug_force_fallback(1)
        b = pkg.MMR.from_leaves(leaves[:1 << 12])
        assert np.array_equal(b.elements, oracle.mmr(leaves[:1 << 12]).elements)
    finally:
        pkg.lib().p2mt_debug_force_fallback(0)
        pkg.set_variant(*DEFAULT_VARIANT)


def test_mfma_mds_variant(pkg, oracle):
    """Variant (2, 2): the stage-1 MMR kernel with every MDS layer on the matrix pipe (v_mfma_i32_4x4x4_16b_i8 over 8-bit limbs, one
    hash per lane, no cross-lane movement; an A/B that came out even, profiles/r03_mds_mfma_ab.txt, kept selectable) is the same
    function: all nodes equal the oracle's on a ragged size (partial waves: the MFMA ignores EXEC) and on limb patterns that
    stress the signed-byte offset (all-0x00, all-0x7F/0x80/0xFF bytes, p - 1), with and without the forced exact redo."""
    n = (1 << 16) + 4099
    leaves = splitmix_leaves(n, 0x5EED0322)
    pat = [0, 1, P - 1, 0x7F7F7F7F7F7F7F7F, 0x8080808080808080, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF, 0x80FF7F0001FE807F,
           0xFEFEFEFEFEFEFEFE, 0x0101010101010101]
    leaves[:1024] = np.array([pat[i % len(pat)] for i in range(1024)], dtype=np.uint64)
    om = oracle.mmr(leaves)
    try:
        pkg.set_variant(2, 2)
        a = pkg.MMR.from_leaves(leaves)
        assert np.array_equal(a.elements, om.elements)
        pkg.lib().p2mt_debug_force_fallback(1)
        b = pkg.MMR.from_leaves(leaves[:1 << 12])
        assert np.array_equal(b.elements, oracle.mmr(leaves[:1 << 12]).elements)
        with pytest.raises(Exception):
            pkg.set_variant(0, 2)  # the matrix-pipe form exists for the fast path only
    finally:
        pkg.lib().p2mt_debug_force_fallback(0)
        pkg.set_variant(*DEFAULT_VARIANT)


@pytest.mark.parametrize("variant", [(2, 0), (2, 5), (2, 6), (2, 7), (2, 8)])
def test_mfma32_default_and_valu_forms(pkg, oracle, variant):
    """The default of the fast path puts every 12-row dense MDS layer on ONE v_mfma_i32_32x32x32_i8 per 8-bit limb (block-structured A
    operand, one hash per lane, no cross-lane movement: poseidon_fast::mds_layer_mfma32) and multiplies in four mads with the
    carry folded into the reduction, and runs the 22 partial rounds as five groups of four and one of three with one MDS application
    each; variant (2, 5) keeps the MDS on the VALU, (2, 6) additionally the previous multiply, (2, 7) is the default with the partial
    rounds in groups of three, (2, 8) the default with the flag-form folds in its MDS layers.  All five are the same function: every node equals the oracle's on a ragged size (partial waves: an MFMA ignores EXEC, so the
    kernels keep every lane in the permutation and only predicate the stores), on limb patterns that stress the signed-byte
    offset carried in the spare K slots (all-0x00 / 0x7F / 0x80 / 0xFF bytes, p - 1), and with the exact redo forced."""
    n = (1 << 16) + 4099
    leaves = splitmix_leaves(n, 0x5EED0532)
    pat = [0, 1, P - 1, 0x7F7F7F7F7F7F7F7F, 0x8080808080808080, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF, 0x80FF7F0001FE807F,
           0xFEFEFEFEFEFEFEFE, 0x0101010101010101, 0xFFFFFFFFFFFFFFFF]
    leaves[:1024] = np.array([pat[i % len(pat)] for i in range(1024)], dtype=np.uint64)
    om = oracle.mmr(leaves)
    try:
        pkg.set_variant(*variant)
        a = pkg.MMR.from_leaves(leaves)
        assert np.array_equal(a.elements, om.elements)
        assert np.array_equal(a.bagging_the_peaks(), om.bagging_the_peaks())
        for small in (1, 2, 3, 17, 63, 64, 65, 255, 1025):  # fewer nodes than lanes: level kernels with idle lanes
            c = pkg.MMR.from_leaves(leaves[:small])
            assert np.array_equal(c.elements, oracle.mmr(leaves[:small]).elements)
        pkg.lib().p2mt_debug_force_fallback(1)
        b = pkg.MMR.from_leaves(leaves[:1 << 12])
        assert np.array_equal(b.elements, oracle.mmr(leaves[:1 << 12]).elements)
    finally:
        pkg.lib().p2mt_debug_force_fallback(0)
        pkg.set_variant(*DEFAULT_VARIANT)


@pytest.mark.parametrize("log_n,levels", [(18, 2), (20, 2), (22, 2), (23, 3)])
def test_adaptive_subtree_size(pkg, oracle, log_n, levels):
    """The per-lane subtrees of the stage-1 launch shrink with the build (2^4 leaves per lane from 2^24 leaves up, 2^3 from 2^23,
    2^2 below: the smallest subtree that leaves 2^20 lanes -- round 4's rule; test_full_size_bit_exact_2pow24 is the 2^4 class).
    Every size class, full node array against the oracle's level-order build, plus a ragged size in the same class."""
    import hashlib
    import torch
    n = 1 << log_n
    assert pkg.lib().p2mt_mmr_stage1_levels(n) == levels
    assert pkg.lib().p2mt_mmr_stage1_levels(1 << 24) == 4 and pkg.lib().p2mt_mmr_stage1_levels(1 << 21) == 2
    leaves = splitmix_leaves(n + 77, 0x5EED0400 + log_n)
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    m = pkg.MMR()
    m.extend_dev(d, n)
    el, _ = oracle.mmr_build_pow2_parallel(leaves[:n], 8)
    assert hashlib.sha256(m.elements.tobytes()).hexdigest() == hashlib.sha256(el.tobytes()).hexdigest()
    m.extend_dev(d[n:], 77)                       # a ragged continuation on top (incremental extend, tiny subtrees)
    r = pkg.MMR()
    r.extend_dev(d, n + 77)                       # and the same leaves in one ragged build
    assert np.array_equal(m.elements, r.elements)
    if log_n <= 20:
        assert np.array_equal(r.elements, oracle.mmr(leaves).elements)


def test_mmr_checkpoint_roundtrip(pkg, oracle, tmp_path):
    """save -> load -> extend continues exactly where the saved MMR stopped; corrupted files are rejected."""
    leaves = splitmix_leaves(3000, 0x5EED0099)
    a = pkg.MMR.from_leaves(leaves[:1777])
    path = str(tmp_path / "mmr.ckpt")
    a.save(path)
    raw = open(path, "rb").read()
    assert raw[:8] == b"P2MTMMR1" and len(raw) == 32 + len(a) * 32
    assert np.array_equal(np.frombuffer(raw[32:], dtype="<u8").reshape(-1, 4), oracle.mmr(leaves[:1777]).elements)
    b = pkg.MMR.load(path)
    assert b.num_leaves == 1777 and np.array_equal(b.elements, a.elements)
    b.extend(leaves[1777:])
    assert np.array_equal(b.elements, oracle.mmr(leaves).elements)
    bad = bytearray(raw)
    bad[100] ^= 1
    open(path, "wb").write(bytes(bad))
    with pytest.raises(pkg.P2mtPanic):
        pkg.MMR.load(path)
    open(path, "wb").write(raw[:-8])
    with pytest.raises(pkg.P2mtPanic):
        pkg.MMR.load(path)
    e = pkg.MMR.new()
    e.save(path)
    assert len(pkg.MMR.load(path)) == 0


def test_mmr_empty_panics(pkg):
    m = pkg.MMR.new()
    assert len(m) == 0
    with pytest.raises(pkg.P2mtPanic):
        m.get_peaks()
    with pytest.raises(pkg.P2mtPanic):
        m.bagging_the_peaks()


def test_config2_mmr_2pow20(pkg, oracle):
    """BASELINE config 2: 2^20-leaf MMR build + get_proof for i in {0, 1, 777777, 2^20-1} + verify."""
    n = 1 << 20
    leaves = splitmix_leaves(n, 0x5EED0000 + 2)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    assert len(m) == 2 * n - 1
    assert np.array_equal(m.elements, om.elements)
    root = m.bagging_the_peaks()
    assert np.array_equal(root, om.bagging_the_peaks())
    assert np.array_equal(root, m.copy_elements(len(m) - 1, 1)[0])  # single peak => root == last element (Q2)
    for i in (0, 1, 777777, n - 1):
        pr = m.get_proof_normal_index(i)
        opr = om.get_proof_normal_index(i)
        assert len(pr.lefts) == 20 and len(pr.peaks) == 1
        assert np.array_equal(pr.siblings, opr["siblings"]) and np.array_equal(pr.lefts, opr["lefts"])
        assert pr.lefts.tolist() == [(i >> b) & 1 for b in range(20)]  # leaf-index bits, LSB first (A.4)
        assert pr.verify(int(leaves[i]), root)


def test_batched_proofs_and_verify(pkg, oracle):
    n = 5000
    leaves = splitmix_leaves(n, 0x5EED0005)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    idx = np.array([0, 1, 2, 4095, 4096, 4999, 777], dtype=np.uint64)
    mmr_idx = np.array([pkg.get_mmr_index(int(i)) for i in idx], dtype=np.uint64)
    sib, lefts, ns = m.get_proof_batch(mmr_idx, max_siblings=16)
    for t, i in enumerate(idx):
        opr = om.get_proof_normal_index(int(i))
        assert ns[t] == len(opr["lefts"])
        assert np.array_equal(sib[t, :ns[t]], opr["siblings"]) and np.array_equal(lefts[t, :ns[t]], opr["lefts"])
    peaks, root = m.get_peaks(), m.bagging_the_peaks()
    st = pkg.verify_proof_batch(sib, lefts, ns, peaks, leaves[idx.astype(np.int64)], root)
    assert st.tolist() == [1] * len(idx)
    wrong = leaves[idx.astype(np.int64)].copy()
    wrong[2] = (int(wrong[2]) + 1) % P
    st = pkg.verify_proof_batch(sib, lefts, ns, peaks, wrong, root)
    assert st.tolist() == [1, 1, -5, 1, 1, 1, 1]
    bad_root = root.copy(); bad_root[0] ^= np.uint64(2)
    assert pkg.verify_proof_batch(sib, lefts, ns, peaks, leaves[idx.astype(np.int64)], bad_root).tolist() == [0] * 7


def test_full_size_properties_2pow24(pkg):
    """BASELINE headline size: properties that need no oracle run (it would take minutes on one core)."""
    n = 1 << 24
    leaves = splitmix_leaves(n, 0x5EED0000 + 24)
    m = pkg.MMR.new()
    m.reserve(n)
    m.extend(leaves)
    assert len(m) == 2 * n - 1
    root = m.bagging_the_peaks()
    assert np.array_equal(root, m.copy_elements(len(m) - 1, 1)[0])
    # (1) 512 random internal nodes equal two_to_one(children) recomputed by the independent batch kernel
    rng = np.random.default_rng(9)
    hs = rng.integers(1, 24, size=512)
    js = np.array([rng.integers(0, n >> h) for h in hs])
    last = ((js + 1) << hs) - 1
    pos = np.array([2 * int(L) - bin(int(L)).count("1") + int(h) for L, h in zip(last, hs)])
    parents = np.stack([m.copy_elements(int(p), 1)[0] for p in pos])
    lefts = np.stack([m.copy_elements(int(p) - (1 << int(h)), 1)[0] for p, h in zip(pos, hs)])
    rights = np.stack([m.copy_elements(int(p) - 1, 1)[0] for p in pos])
    assert np.array_equal(pkg.two_to_one_batch(np.concatenate([lefts, rights], axis=1)), parents)
    # (2) leaf digests are the no-op pad at 2i - popcount(i)
    for i in (0, 1, 12345678, n - 1):
        assert m.copy_elements(2 * i - bin(i).count("1"), 1)[0].tolist() == [int(leaves[i]), 0, 0, 0]
    # (3) 256 random proofs verify against the root (batched path), and a corrupted leaf is rejected
    idx = rng.integers(0, n, size=256)
    mmr_idx = np.array([pkg.get_mmr_index(int(i)) for i in idx], dtype=np.uint64)
    sib, lf, ns = m.get_proof_batch(mmr_idx, max_siblings=24)
    assert (ns == 24).all()
    st = pkg.verify_proof_batch(sib, lf, ns, root[None], leaves[idx], root)
    assert (st == 1).all()
    # (4) the first 2^16 leaves' subtree equals an independently built 2^16 MMR (prefix property of post-order)
    sub = pkg.MMR.from_leaves(leaves[:1 << 16])
    assert np.array_equal(sub.elements, m.copy_elements(0, len(sub)))
    # (5) determinism: rebuilding gives identical bytes
    m2 = pkg.MMR.from_leaves(leaves)
    assert np.array_equal(m2.bagging_the_peaks(), root)


def test_config5_shard_size_2pow26(pkg):
    """A 2^26-leaf build on one GPU (4.3 GB of nodes; config 5 shards 2^23 per GPU, this is 8x that): 64-bit
    positions, grid sizes and the staging policy at a size no oracle run can cover -- checked through properties."""
    n = 1 << 26
    leaves = splitmix_leaves(n, 0x5EED0000 + 26)
    m = pkg.MMR.new()
    m.reserve(n)
    m.extend(leaves)
    assert len(m) == 2 * n - 1
    root = m.bagging_the_peaks()
    assert np.array_equal(root, m.copy_elements(len(m) - 1, 1)[0])
    # the two halves are themselves 2^25-leaf MMRs whose roots are the children of the root
    half = pkg.MMR.from_leaves(leaves[n // 2:])
    right = half.bagging_the_peaks()
    assert np.array_equal(right, m.copy_elements(len(m) - 2, 1)[0])
    left = m.copy_elements(len(m) - 1 - (1 << 26), 1)[0]
    assert np.array_equal(pkg.two_to_one(left, right), root)
    # proofs from both ends and the middle verify (26 siblings, bits of the index)
    idx = np.array([0, 1, n // 2 - 1, n // 2, n - 2, n - 1, 12345678], dtype=np.uint64)
    mmr_idx = np.array([2 * int(i) - bin(int(i)).count("1") for i in idx], dtype=np.uint64)  # get_mmr_index panics >= 2^30 only
    sib, lf, ns = m.get_proof_batch(mmr_idx, max_siblings=26)
    assert (ns == 26).all()
    for t, i in enumerate(idx):
        assert lf[t].tolist() == [(int(i) >> b) & 1 for b in range(26)]
    st = pkg.verify_proof_batch(sib, lf, ns, root[None], leaves[idx.astype(np.int64)], root)
    assert (st == 1).all()


def test_poseidon_gate_witness_rows(pkg, oracle):
    """Witness fill for PoseidonGate rows (wire-major) vs the oracle [wire layout parity unpinned]."""
    rng = np.random.default_rng(17)
    n = 700
    x = np.concatenate([edge_states(), rng.integers(0, 1 << 64, size=(n - len(edge_states()), 12), dtype=np.uint64)])
    sw = rng.integers(0, 2, size=n).astype(np.uint8)
    w = pkg.poseidon_gate_witness_batch(x, sw)
    assert w.shape == (135, n)
    for i in list(range(0, n, 23)) + [n - 1]:
        assert np.array_equal(w[:, i], oracle.poseidon_gate_witness(x[i], sw[i])), i
    # outputs column block == the batch permutation of the swapped inputs (independent kernel)
    xs = x.copy() % np.uint64(P)
    swapped = xs.copy()
    m = sw.astype(bool)
    swapped[m, 0:4], swapped[m, 4:8] = xs[m, 4:8], xs[m, 0:4]
    assert np.array_equal(w[12:24].T, pkg.poseidon_permute_batch(swapped))


def test_field_primitives_rare_paths(pkg):
    """Rule: a rare data-dependent path needs its own test.  The carry/borrow fix-ups of the device field
    primitives fire with probability 2^-22 .. 2^-64 on random data, so they are driven here with crafted operands
    (all-ones words, values within 2^32 of 2^64, p-1, ...) and checked against Python big integers."""
    import ctypes as C
    lib, N = pkg.lib(), pkg._native
    M64 = (1 << 64) - 1
    edge = [0, 1, 2, P - 1, P, P + 1, M64, M64 - 1, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF,
            0x00000000FFFFFFFE, 1 << 63, (1 << 63) - 1, 0xFFFFFFFF00000002, 0x8000000080000000, 0x3FF, 0xFFFFFC0000000000]
    rng = np.random.default_rng(99)
    a = np.array([x for x in edge for _ in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4000, dtype=np.uint64)], dtype=np.uint64)
    b = np.array([y for _ in edge for y in edge] + [int(v) for v in rng.integers(0, 1 << 64, 4000, dtype=np.uint64)], dtype=np.uint64)

    def run(op):
        out, flag = np.zeros(a.size, np.uint64), np.zeros(a.size, np.uint8)
        N.check(lib.p2mt_debug_field_op(op, N.ptr(a), N.ptr(b), a.size, N.ptr(out), N.ptr(flag)))
        return [int(x) for x in out], flag
    A, B = [int(x) for x in a], [int(x) for x in b]
    out, _ = run(0)
    assert out == [(x + (y << 64)) % P for x, y in zip(A, B)]                      # exact 128 -> 64 reduce
    out, _ = run(1)
    assert out == [(x + ((y & 0x3FF) << 64)) % P for x, y in zip(A, B)]            # exact 96 -> 64 fold
    out, _ = run(3)
    assert out == [x * y % P for x, y in zip(A, B)]                                # exact multiply
    out, flag = run(2)                                                             # flag form: right unless flagged
    exp = [(x + (y << 64)) % P for x, y in zip(A, B)]
    assert all(o == e for o, e, f in zip(out, exp, flag) if not f)
    assert flag.sum() > 0, "crafted operands must drive the flagged path"
    out, _ = run(6)
    assert out == [x * y % P for x, y in zip(A, B)]                                # gl::mul (four mads, carry folded into the reduce)
    out, _ = run(7)
    rotl = lambda x: ((x << 17) | (x >> 47)) & M64
    assert out == [(x * y + (rotl(x) ^ y)) % P for x, y in zip(A, B)]              # gl::mul_add
    out, flag = run(8)                                                             # flag-form multiply of the permutation
    assert all(o == x * y % P for o, x, y, f in zip(out, A, B, flag) if not f)
    assert flag.sum() < len(A) // 4
    out, _ = run(9)
    assert out == [(x - y) % P for x, y in zip(A, B)]                              # a - b for any operands (both wraps taken back)
    out, flag = run(10)                                                            # ... the second wrap left to the flag
    assert all(o == (x - y) % P for o, x, y, f in zip(out, A, B, flag) if not f)
    assert 0 < flag.sum() < len(A) // 8
    for op, fn in ((4, lambda x, y: (x + y) % P), (5, lambda x, y: (x - y) % P)):  # loose add / sub of the LDE kernel
        out, flag = run(op)
        assert all(o == fn(x, y) for o, x, y, f in zip(out, A, B, flag) if not f)
        assert flag.sum() > 0


def test_partial_round_groups_and_their_overflow_flag(pkg):
    """The 22 partial rounds run as groups of four (and one of three) with ONE MDS application each (poseidon_fast::partial_rounds_g):
    M^4 has 29-bit entries and row sums of 1.01 - 1.04 x 2^32, so the last link of a row's mad chain can carry out of 64 bits when
    all twelve 32-bit halves of a state are within 4 % of 2^32 -- no hash input reaches that, so the states go in directly
    (p2mt_debug_partial_group).  Against a plain Python restatement of the same rounds: every lane that does not raise the flag is
    exact; random states never raise it; states of all-ones words always do (in the four-round groups), and so do some of the
    near-all-ones ones -- the kernels answer the flag with the exact permutation (test_fast_path_fallback_is_exact)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import poseidon_spec as ps
    lib, N = pkg.lib(), pkg._native
    rc, mds = ps.round_constants(), ps.mds_matrix()
    M64 = (1 << 64) - 1
    rng = np.random.default_rng(2024)
    n_rand = 600
    states = [[int(v) for v in row] for row in rng.integers(0, 1 << 64, (n_rand, 12), dtype=np.uint64)]
    states += [[M64] * 12, [P - 1] * 12, [0] * 12, [0xFFFFFFFF] * 12, [0xFFFFFFFF00000000] * 12]
    for _ in range(400):  # eleven words of all ones, word 0 anything; and states a few per cent below all ones in every half
        states.append([int(rng.integers(0, 1 << 64, dtype=np.uint64))] + [M64] * 11)
        lo = rng.integers(int(0.9 * 2**32), 1 << 32, 12, dtype=np.uint64)
        hi = rng.integers(int(0.9 * 2**32), 1 << 32, 12, dtype=np.uint64)
        states.append([int((h << np.uint64(32)) | l) for h, l in zip(hi, lo)])
    a = np.array(states, dtype=np.uint64)

    def reference(state, g):
        G, lead, r0 = (3, True, 23) if g == 5 else (4, g != 0, 3 + 4 * g)
        s = [x % P for x in state]
        if lead:
            s[0] = ps.sbox(s[0])
        for t in range(1, G + 1):
            s = [(v + rc[12 * (r0 + t) + i]) % P for i, v in enumerate(ps.mat_vec(mds, s))]
            if t < G:
                s[0] = ps.sbox(s[0])
        return s

    for g in range(6):
        out = np.zeros_like(a)
        flag = np.zeros(len(states), np.uint8)
        N.check(lib.p2mt_debug_partial_group(g, N.ptr(a), len(states), N.ptr(out), N.ptr(flag)))
        assert flag[:n_rand].sum() == 0, "random states must not raise the flag"
        for i, st in enumerate(states):
            if not flag[i]:
                assert [int(x) for x in out[i]] == reference(st, g), (g, i)
        if g < 5:
            assert flag[n_rand] == 1, "all-ones words: every chain of a four-round group passes 2^64"
            assert flag[n_rand + 5:].sum() > 0
        else:
            assert flag.sum() == 0 or flag.sum() < 8  # groups of three cannot overflow; only the 2^-32 borrows could flag


def test_transform_arithmetic_primitives(pkg):
    """The NTT / LDE kernels' field arithmetic (csrc/ntt_arith.hip.h): add, sub, multiply, the fused butterfly and the
    multiplication by every power of two 2^E, E in [0, 192), on crafted edge operands + random ones, against Python integers.
    A flagged lane (the astronomically rare second wrap the kernels answer with an exact redo) may hold anything; every other lane is
    exact, the crafted operands do raise the flag where the design says they can, and random operands never do."""
    lib, N = pkg.lib(), pkg._native
    M64 = (1 << 64) - 1
    edge = [0, 1, 2, P - 1, P, P + 1, M64, M64 - 1, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFEFFFFFFFF,
            0x00000000FFFFFFFE, 1 << 63, (1 << 63) - 1, 0xFFFFFFFF00000002, 0x8000000080000000, 0x3FF, 0xFFFFFC0000000000]
    rng = np.random.default_rng(123)
    n_edge = len(edge) ** 2
    a = np.array([x for x in edge for _ in edge] + [int(v) for v in rng.integers(0, 1 << 64, 3000, dtype=np.uint64)], dtype=np.uint64)
    b = np.array([y for _ in edge for y in edge] + [int(v) for v in rng.integers(0, 1 << 64, 3000, dtype=np.uint64)], dtype=np.uint64)
    A, B = [int(x) for x in a], [int(x) for x in b]

    def run(op):
        out, flag = np.zeros(a.size, np.uint64), np.zeros(a.size, np.uint8)
        N.check(lib.p2mt_debug_field_op(op, N.ptr(a), N.ptr(b), a.size, N.ptr(out), N.ptr(flag)))
        return [int(x) for x in out], flag

    flagged_somewhere = 0
    for op, fn in ((11, lambda x, y: x + y), (12, lambda x, y: x - y), (13, lambda x, y: x * y), (14, lambda x, y: x + y),
                   (15, lambda x, y: x - y)):
        out, flag = run(op)
        assert all(o == fn(x, y) % P for o, x, y, f in zip(out, A, B, flag) if not f), op
        assert flag[n_edge:].sum() == 0, op  # random operands never take the rare path
        flagged_somewhere += int(flag.sum())
    assert flagged_somewhere > 0
    for e in range(192):
        out, flag = run(16 + e)
        k = pow(2, e, P)
        assert all(o == x * k % P for o, x, f in zip(out, A, flag) if not f), e
        assert flag[n_edge:].sum() == 0, e


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_mmr_random_extend_sequences(pkg, oracle, seed):
    """Random sequences of extends (sizes 0, 1, small, around the 2^10-leaf tile and 2^12/2^16 policy boundaries)
    must reproduce the oracle's `for leaf { add_leaf }` array at every checkpoint."""
    rng = np.random.default_rng(seed)
    total = int(rng.integers(30000, 90000))
    leaves = splitmix_leaves(total, 0x5EED1000 + seed)
    cuts, pos = [0], 0
    choices = [0, 1, 2, 3, 63, 64, 65, 1023, 1024, 1025, 2047, 2048, 4095, 4097, 8191, 16385]
    while pos < total:
        step = int(rng.choice(choices)) if rng.random() < 0.7 else int(rng.integers(1, 20000))
        pos = min(total, pos + step)
        cuts.append(pos)
    m = pkg.MMR.new()
    om = oracle.mmr()
    check_at = set(rng.choice(len(cuts) - 1, size=min(6, len(cuts) - 1), replace=False).tolist()) | {len(cuts) - 2}
    for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        m.extend(leaves[lo:hi])
        om.add_leaves(leaves[lo:hi])
        if i in check_at:
            assert len(m) == len(om)
            assert np.array_equal(m.elements, om.elements), (seed, i, lo, hi)
            if hi > 0:
                assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())


def test_library_stream_selection(pkg, oracle):
    """p2mt_set_stream: the library enqueues on the caller's HIP stream (a torch stream here) and results do not
    depend on which stream is used."""
    import torch
    lib, N = pkg.lib(), pkg._native
    leaves = splitmix_leaves(70000, 0x5EED2000)
    ref = oracle.mmr(leaves)
    stream = torch.cuda.Stream()
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    torch.cuda.synchronize()
    try:
        N.check(lib.p2mt_set_stream(stream.cuda_stream))
        m = pkg.MMR.new()
        m.extend_dev(d, leaves.size)
        root = m.bagging_the_peaks()        # synchronises the library stream
        el = m.elements
    finally:
        N.check(lib.p2mt_set_stream(None))
    assert np.array_equal(root, ref.bagging_the_peaks()) and np.array_equal(el, ref.elements)


def test_add_leaf_loop_is_write_combined(pkg, oracle):
    """`for leaf { mmr.add_leaf(leaf) }` (how every reference caller builds an MMR, mmr_plonky2_verifier.rs:109-112):
    same observable state as the oracle at every observation point, and fast enough to be usable."""
    import time
    leaves = splitmix_leaves(50000, 0x5EED3000)
    m = pkg.MMR.new()
    om = oracle.mmr()
    t0 = time.perf_counter()
    for i, v in enumerate(leaves):
        m.add_leaf(int(v))
        if i in (0, 1, 2, 6, 999, 30000):           # observing flushes the queue
            om.add_leaves(leaves[om_len_leaves(om):i + 1])
            assert len(m) == len(om) and m.num_leaves == i + 1
            assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())
    dt = time.perf_counter() - t0
    om.add_leaves(leaves[om_len_leaves(om):])
    assert np.array_equal(m.elements, om.elements)
    pr = m.get_proof_normal_index(49999)
    assert pr.verify(int(leaves[49999]), m.bagging_the_peaks())
    m.add_leaf(5)
    m.extend([6, 7])                                   # extend after queued add_leaf keeps the order
    om.add_leaves(np.array([5, 6, 7], dtype=np.uint64))
    assert np.array_equal(m.elements, om.elements)
    assert dt < 20.0, "add_leaf loop took %.1f s" % dt


def om_len_leaves(om):
    """number of leaves in an oracle MMR of len L = 2N - popcount(N)"""
    L = len(om)
    n = (L + 1) // 2
    while 2 * n - bin(n).count("1") < L:
        n += 1
    return n


def test_device_resident_proof_service(pkg, oracle):
    """p2mt_mmr_proof_batch_dev / p2mt_mmr_proof_verify_batch_dev: proofs produced and verified without leaving HBM."""
    import torch
    lib, N = pkg.lib(), pkg._native
    n = 30000
    leaves = splitmix_leaves(n, 0x5EED4000)
    m = pkg.MMR.from_leaves(leaves)
    om = oracle.mmr(leaves)
    idx = np.array([0, 1, 16383, 16384, 29999, 777], dtype=np.int64)
    mmr_idx = np.array([pkg.get_mmr_index(int(i)) for i in idx], dtype=np.int64)
    d_idx = torch.from_numpy(mmr_idx).cuda()
    cnt, ms = len(idx), 16
    d_sib = torch.zeros(cnt * ms * 4, dtype=torch.int64, device="cuda")
    d_lf = torch.zeros(cnt * ms, dtype=torch.uint8, device="cuda")
    d_ns = torch.zeros(cnt, dtype=torch.int32, device="cuda")
    N.check(lib.p2mt_mmr_proof_batch_dev(m._h, N.ptr(d_idx), cnt, ms, N.ptr(d_sib), N.ptr(d_lf), N.ptr(d_ns)))
    peaks, root = m.get_peaks(), m.bagging_the_peaks()
    d_peaks = torch.from_numpy(peaks.view(np.int64).copy()).cuda()
    d_root = torch.from_numpy(root.view(np.int64).copy()).cuda()
    d_leaves = torch.from_numpy(leaves[idx].view(np.int64).copy()).cuda()
    d_st = torch.zeros(cnt, dtype=torch.int8, device="cuda")
    N.check(lib.p2mt_mmr_proof_verify_batch_dev(N.ptr(d_sib), N.ptr(d_lf), N.ptr(d_ns), ms, N.ptr(d_peaks), len(peaks),
                                                N.ptr(d_leaves), N.ptr(d_root), cnt, N.ptr(d_st)))
    N.check(lib.p2mt_sync())
    torch.cuda.synchronize()
    assert d_st.cpu().tolist() == [1] * cnt
    sib = d_sib.cpu().numpy().view(np.uint64).reshape(cnt, ms, 4)
    ns = d_ns.cpu().numpy()
    for t, i in enumerate(idx):
        ref = om.get_proof_normal_index(int(i))
        assert ns[t] == len(ref["lefts"]) and np.array_equal(sib[t, :ns[t]], ref["siblings"])


@pytest.mark.parametrize("env", [
    {"P2MT_SUBTREE": "0"},                              # fused LDS tiles (k_mmr_tile) as stage 1
    {"P2MT_SUBTREE": "0", "P2MT_TILE_LOG": "9"},
    {"P2MT_SUBTREE": "0", "P2MT_TILE_LOG": "11"},
    {"P2MT_SUBTREE": "5"},                              # 32-leaf per-lane subtrees
    {"P2MT_SUBTREE": "4"},                              # 16-leaf subtrees pinned (the default adapts the size to the build)
    {"P2MT_SUBTREE": "3"},
    {"P2MT_SUBTREE": "2", "P2MT_SUBTREE_BLOCK": "64"},  # (small subtrees exist for 256-lane workgroups: the block knob yields)
    {"P2MT_SUBTREE_OCC": "3"},                          # stage 1 at the allocator's own three waves per SIMD
    {"P2MT_SUBTREE_BLOCK": "64"},
    {"P2MT_SUBTREE_BLOCK": "128"},
    {"P2MT_GRIND_QUEUE": "0"},                          # single-proof proof-of-work on the four-lane kernel instead of the queue kernel
    {"P2MT_QUAD": "0", "P2MT_LDE12": "0"},              # no four-lane kernels, radix-2 LDE at 2^12
    {"P2MT_THROUGHPUT": "1"},                           # lane-efficient layouts instead of the latency-optimised ones
    {"P2MT_WITNESS_LDS": "0"},                          # witness value table in global memory (dataflow interpreter)
    {"P2MT_WITNESS_LDS": "0", "P2MT_WITNESS_GRID": "1"},  # ... level-synchronous over the grid
    {"P2MT_WITNESS_LDS": "0", "P2MT_WITNESS_GRID": "0"},  # ... one workgroup
])
def test_env_knobs(env):
    """Every P2MT_* runtime knob selects kernels that stay bit-exact (each in a fresh process: read once at init)."""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "knob_check.py")], env=e,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "knobs ok" in r.stdout
