"""CPU-only: the C-ABI library loads, exports every symbol include/p2mt.h declares, pure-host index maths
matches the reference's tables, and compute entry points fail loudly without a GPU (no CPU fallback)."""
import os
import re
import sys

import numpy as np
import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    return ge.load_package()


def test_header_symbols_exported(pkg):
    hdr = open(os.path.join(ROOT, "include", "p2mt.h")).read()
    declared = set(re.findall(r"\b(p2mt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"p2mt_status"}
    assert len(declared) >= 45
    lib = pkg.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    # the ctypes table covers the whole header (so every test calls through the declared ABI)
    assert declared == set(pkg._native.SIGNATURES.keys())


def test_index_tables(pkg, golden):
    """merkle_mountain_ranges.rs:280-301 and :307-327 through the product's ABI."""
    for size, bitmap in golden["reference_vectors"]["heights_bitmap"]["pairs"]:
        assert pkg.get_heights_bitmap_for_mmr_size(size) == (bitmap, 0)
    assert pkg.get_heights_bitmap_for_mmr_size(0) == (0, 0)
    for n, idx in golden["reference_vectors"]["mmr_index"]["pairs"]:
        assert pkg.get_mmr_index(n) == idx
    with pytest.raises(pkg.P2mtPanic):
        pkg.get_mmr_index(1 << 30)  # i32 overflow in the reference (:264)


def test_index_maths_matches_oracle(pkg, oracle):
    rng = np.random.default_rng(0)
    for x in list(range(0, 300)) + [int(v) for v in rng.integers(0, 1 << 40, size=200)]:
        assert pkg.get_heights_bitmap_for_mmr_size(x) == oracle.heights_bitmap(x)
    for n in list(range(0, 300)) + [int(v) for v in rng.integers(0, 1 << 29, size=200)]:
        assert pkg.get_mmr_index(n) == oracle.get_mmr_index(n)
    lib = pkg.lib()
    for L in (0, 1, 7, 8, 1023, 12345678):
        for h in (0, 1, 5):
            assert lib.p2mt_mmr_node_pos(L, h) == 2 * L - bin(L).count("1") + h
    assert lib.p2mt_mmr_shard_first_pos(1 << 23, 3) == 2 * (3 << 23) - 2


def test_host_side_tree_indexing(pkg, oracle, golden):
    """get_merkle_proof / get_in_between_hashes are pure index arithmetic on a level-major tree: check them on
    the reference's 16-leaf vector (levels taken from the golden fixture, no hashing involved)."""
    g = golden["reference_vectors"]["tree16"]
    levels = np.concatenate([np.asarray(l, dtype=np.uint64) for l in g["levels"]])
    root = np.asarray(g["root"], dtype=np.uint64)
    tree = pkg.MerkleTree(4, levels, root, 16)
    for i in range(16):
        assert np.array_equal(tree.get_merkle_proof(i), oracle.merkle_get_proof(levels, 16, i))
        assert np.array_equal(tree.get_in_between_hashes(i), oracle.merkle_get_in_between_hashes(levels, root, 16, i))
    with pytest.raises(pkg.P2mtPanic):
        tree.get_merkle_proof(16)  # assert!(leaf_index < len) :56


def test_no_cpu_fallback(pkg):
    if pkg.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(pkg.P2mtError) as e:
        pkg.MerkleTree.build([1, 2, 3, 4])
    assert e.value.code == -3
    with pytest.raises(pkg.P2mtError):
        pkg.two_to_one([1, 2, 3, 4], [5, 6, 7, 8])
    with pytest.raises(pkg.P2mtError):
        pkg.MMR.new()


def test_product_does_not_touch_oracle():
    """The product path must never import/link anything under oracle/ (plonky2's own `fri/oracle.rs` file name
    may be cited in comments)."""
    pat = re.compile(r"oracle_lib|liboracle|oracle/|oracle\.h|import oracle|from oracle|oracle_[a-z]")
    pkg_dir = os.path.join(ROOT, "plonky2-merkle-trees_amd")
    scanned = 0
    for base in (pkg_dir, os.path.join(ROOT, "include")):
        for dirpath, _, files in os.walk(base):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f)).read().replace("fri/oracle.rs", "")
                    assert not pat.search(text), (dirpath, f, pat.search(text).group(0))
                    scanned += 1
    assert scanned >= 10


def _construct(builder, gadgets, n_sib, n_peaks):
    """verify_mmr_proof_circuit (mmr_plonky2_verifier.rs:13-87) up to, not including, build()."""
    leaf = builder.add_virtual_target()
    next_hash = builder.hash_or_noop([leaf])
    for _ in range(n_sib):
        elm = builder.add_virtual_hash()
        on_left = builder.add_virtual_bool_target_safe()
        option1 = builder.hash_or_noop(list(elm) + list(next_hash))
        option2 = builder.hash_or_noop(list(next_hash) + list(elm))
        next_hash = gadgets.pick_hash(builder, option1, option2, on_left)
    peaks, equals = [], []
    for _ in range(n_peaks):
        peak = builder.add_virtual_hash()
        peaks.append(peak)
        equals.append(gadgets.equal(builder, peak, next_hash))
    builder.connect(builder.one(), gadgets.or_list(builder, equals))
    if n_peaks > 1:
        builder.register_public_inputs(builder.hash_n_to_hash_no_pad([e for p in peaks for e in p]))
    else:
        builder.register_public_inputs(peaks[0])


@pytest.mark.parametrize("n_sib,n_peaks", [(0, 1), (3, 2), (20, 1), (31, 5), (63, 3)])
def test_circuit_builder_host_logic_matches_oracle(pkg, oracle, n_sib, n_peaks):
    """The builder is host code (constant folding, operation cache, ArithmeticGate slot packing): without a GPU the gate rows
    it lays out before build() must already equal the oracle builder's; build() itself needs the device (status -3)."""
    from oracle import circuit as OC
    from plonky2_merkle_trees_amd import mmr_plonky2_verifier as G
    b = pkg.CircuitBuilder()
    _construct(b, G, n_sib, n_peaks)
    ob = OC.CircuitBuilder(oracle)
    _construct(ob, OC, n_sib, n_peaks)
    assert b.num_gates() == len(ob.gate_instances)
    poseidon_rows = 2 * n_sib + ((4 * n_peaks + 7) // 8 if n_peaks > 1 else 0)  # two hashes per path element + the bagging
    arithmetic_rows = sum(1 for g in ob.gate_instances if g[0] == OC.ARITHMETIC)
    assert b.num_gates() == poseidon_rows + arithmetic_rows
    if pkg.device_count() == 0:
        with pytest.raises(pkg.P2mtError) as e:
            b.build()
        assert e.value.code == -3


def test_rust_ffi_matches_header():
    """shim/src/ffi.rs (the Rust extern block a maintainer of the reference links against; source only, no Rust toolchain here) is
    generated from include/p2mt.h: it must be current and declare every exported function exactly once."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"])
    assert r.returncode == 0, "shim/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"
    hdr = open(os.path.join(ROOT, "include", "p2mt.h")).read()
    declared = set(re.findall(r"\b(p2mt_[a-z0-9_]+)\s*\(", hdr)) - {"p2mt_status"}
    ffi = open(os.path.join(ROOT, "shim", "src", "ffi.rs")).read()
    have = re.findall(r"pub fn (p2mt_\w+)\(", ffi)
    assert sorted(have) == sorted(declared)
    # the shim's modules call only functions the extern block declares
    for rel in ("lib.rs", "plonk.rs", "simple_merkle_tree/simple_merkle_tree.rs", "mmr/merkle_mountain_ranges.rs", "mmr/common.rs",
                "mmr/mmr_plonky2_verifier.rs", "mmr/mmr_plonky2_verifier_1_recursion.rs"):
        src = open(os.path.join(ROOT, "shim", "src", rel)).read()
        for name in set(re.findall(r"ffi::(p2mt_\w+)\(", src)):
            assert name in declared, (rel, name)
    # the reference's public signatures are kept (file:line in the shim's docs)
    mmr = open(os.path.join(ROOT, "shim", "src", "mmr", "merkle_mountain_ranges.rs")).read()
    for sig in ("pub fn new() -> Self", "pub fn add_leaf(&mut self, leaf: GoldilocksField)",
                "pub fn get_heights_bitmap_for_mmr_size(mmr_size: usize) -> (u64, usize)",
                "pub fn get_mmr_index(leaf_normal_index: usize) -> usize",
                "pub fn verify(self, leaf: GoldilocksField, root: HashOut<GoldilocksField>) -> bool",
                "pub fn get_subtree_proof_elm(mmr: MMR, mmr_index: usize) -> Vec<(HashOut<GoldilocksField>, bool)>"):
        assert sig in mmr, sig
    mt = open(os.path.join(ROOT, "shim", "src", "simple_merkle_tree", "simple_merkle_tree.rs")).read()
    for sig in ("pub fn build(leaves: Vec<GoldilocksField>) -> Self",
                "pub fn get_merkle_proof(self, leaf_index: usize) -> Vec<HashOut<GoldilocksField>>",
                "pub fn get_in_between_hashes(self, leaf_index: usize) -> Vec<HashOut<GoldilocksField>>"):
        assert sig in mt, sig


def test_rust_shim_circuit_surface():
    """The circuit side of the shim (source only): the reference's module list (/root/reference/src/mmr/mod.rs), its constructor names
    with their tuple returns, and the Plonky2 surface they are written against -- with arities checked against the extern block, so a
    header change that the shim does not follow fails here rather than at a maintainer's first `cargo build`."""
    src = lambda rel: open(os.path.join(ROOT, "shim", "src", rel)).read()
    lib = src("lib.rs")
    for mod in ("common", "merkle_mountain_ranges", "mmr_plonky2_verifier", "mmr_plonky2_verifier_1_recursion"):
        assert "pub mod %s;" % mod in lib, mod
    common = src("mmr/common.rs")
    for sig in ("pub fn equal(builder: &mut CircuitBuilder, first: HashOutTarget, second: HashOutTarget) -> BoolTarget",
                "pub fn or_list(builder: &mut CircuitBuilder, ins: Vec<BoolTarget>) -> BoolTarget",
                "pub fn pick_hash(builder: &mut CircuitBuilder, option1: HashOutTarget, option2: HashOutTarget, pick_left: BoolTarget) -> HashOutTarget"):
        assert sig in common, sig
    # pick_hash: the reference's call order (not, four mul on option2, four mul_add on option1) is what fixes the gate layout
    body = common[common.index("pub fn pick_hash"):]
    assert body.index("builder.not(") < body.index("builder.mul(") < body.index("builder.mul_add(")
    assert "pub fn verify_mmr_proof_circuit(nr_merkle_proof_elms: usize, nr_peaks: usize) -> (CircuitData, Target, Vec<(HashOutTarget, BoolTarget)>, Vec<HashOutTarget>)" \
        in src("mmr/mmr_plonky2_verifier.rs")
    rec = src("mmr/mmr_plonky2_verifier_1_recursion.rs")
    assert "pub fn verify_inner_merkle_proof_circuit(nr_merkle_proof_elms: usize, nr_peaks: usize) -> (CircuitData, Target, Vec<(HashOutTarget, BoolTarget)>)" in rec
    assert "-> (CircuitData, ProofWithPublicInputsTarget, VerifierCircuitTarget, Vec<HashOutTarget>)" in rec
    plonk = src("plonk.rs")
    for name in ("pub fn prove(&self, pw: PartialWitness)", "pub fn verify(&self, proof: ProofWithPublicInputs)", "pub fn set_target(",
                 "pub fn set_hash_target(", "pub fn set_bool_target(", "pub fn set_proof_with_pis_target(", "pub fn set_verifier_data_target(",
                 "pub fn add_virtual_proof_with_pis(", "pub fn add_virtual_verifier_data(", "pub fn verify_proof(", "pub fn build(self) -> CircuitData"):
        assert name in plonk, name
    # every ffi call in plonk.rs passes as many arguments as the extern block declares
    ffi = src("ffi.rs")
    arity = {m.group(1): (0 if not m.group(2).strip() else m.group(2).count(",") + 1) for m in re.finditer(r"pub fn (p2mt_\w+)\(([^)]*)\)", ffi)}
    for m in re.finditer(r"ffi::(p2mt_\w+)\(", plonk):
        name, i, depth, commas, any_arg = m.group(1), m.end(), 1, 0, False
        while depth:
            ch = plonk[i]
            if ch in "([": depth += 1
            elif ch in ")]": depth -= 1
            elif ch == "," and depth == 1: commas += 1
            elif depth == 1 and not ch.isspace(): any_arg = True
            i += 1
        assert (commas + 1 if any_arg else 0) == arity[name], (name, commas + 1, arity[name])
