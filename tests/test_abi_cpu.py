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
    for rel in ("src/lib.rs", "src/simple_merkle_tree/simple_merkle_tree.rs", "src/mmr/merkle_mountain_ranges.rs", "plonky2/src/lib.rs",
                "p2mt-sys/src/lib.rs"):
        src = open(os.path.join(ROOT, "shim", rel)).read()
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
    """The circuit side of the shim (source only) is a crate NAMED plonky2 (shim/plonky2) with the slice of plonky2's surface the
    reference's circuit files use -- module paths, generics, method names -- so that /root/reference/src/mmr/{common,
    mmr_plonky2_verifier, mmr_plonky2_verifier_1_recursion}.rs compile against it UNCHANGED; the repository holds no
    copy of those files any more.  Checked here: every path the reference imports resolves to an item of that crate, the generic
    signatures the reference's annotations need exist, and every ffi call passes as many arguments as the extern block declares."""
    shim = os.path.join(ROOT, "shim")
    for gone in ("src/mmr/common.rs", "src/mmr/mmr_plonky2_verifier.rs", "src/mmr/mmr_plonky2_verifier_1_recursion.rs", "src/plonk.rs"):
        assert not os.path.exists(os.path.join(shim, gone)), gone + ": the reference's circuit files are not re-hosted"
    plonk = open(os.path.join(shim, "plonky2", "src", "lib.rs")).read()
    cargo = open(os.path.join(shim, "plonky2", "Cargo.toml")).read()
    assert 'name = "plonky2"' in cargo and "p2mt-sys" in cargo
    # module tree: plonky2::{field, hash::{hash_types, poseidon}, iop::{target, witness}, plonk::{config, proof, circuit_data, circuit_builder}}
    for mod in ("pub use plonky2_field as field;", "pub mod iop {", "pub mod target {", "pub mod witness {", "pub mod hash {", "pub mod hash_types {",
                "pub mod poseidon {", "pub mod plonk {", "pub mod config {", "pub mod proof {", "pub mod circuit_data {", "pub mod circuit_builder {"):
        assert mod in plonk, mod
    # the items the reference names (mmr_plonky2_verifier.rs:2, :18-20, :30-34, :89, :121-150; ..._1_recursion.rs:2, :84-104, :183-220;
    # common.rs:1; merkle_mountain_ranges.rs:3), with the generics its type annotations spell out
    for item in ("pub struct PoseidonHash;", "pub struct HashOutTarget {", "pub struct HashOut<F> {", "pub struct PoseidonGoldilocksConfig;",
                 "pub trait GenericConfig<const D: usize>", "type F: Gl64;", "pub trait Hasher<F: Gl64>", "pub trait AlgebraicHasher<F: Gl64>",
                 "pub struct CircuitData<F: Gl64, C: GenericConfig<D, F = F>, const D: usize>", "pub struct CircuitConfig {",
                 "pub fn standard_recursion_config() -> Self", "pub struct CommonCircuitData<F: Gl64, const D: usize>", "pub config: CircuitConfig,",
                 "pub fri_config: FriConfig,", "pub cap_height: usize,", "pub struct VerifierCircuitTarget {",
                 "pub struct VerifierOnlyCircuitData<C: GenericConfig<D>, const D: usize>", "pub public_inputs: Vec<Target>,",
                 "pub struct CircuitBuilder<F: Gl64, const D: usize>", "pub fn new(config: CircuitConfig) -> Self",
                 "pub fn hash_or_noop<H: AlgebraicHasher<F>>(&mut self, inputs: Vec<Target>) -> HashOutTarget",
                 "pub fn hash_n_to_hash_no_pad<H: AlgebraicHasher<F>>(&mut self, inputs: Vec<Target>) -> HashOutTarget",
                 "pub fn build<C: GenericConfig<D, F = F>>(self) -> CircuitData<F, C, D>", "pub fn verify_proof<C: GenericConfig<D, F = F>>(",
                 "pub fn add_virtual_proof_with_pis(&mut self, common_data: &CommonCircuitData<F, D>) -> ProofWithPublicInputsTarget<D>",
                 "pub fn add_virtual_verifier_data(&mut self, cap_height: usize) -> VerifierCircuitTarget",
                 "pub struct ProofWithPublicInputs<F: Gl64, C: GenericConfig<D, F = F>, const D: usize>", "pub struct ProofWithPublicInputsTarget<const D: usize>",
                 "pub struct PartialWitness<F: Gl64>", "pub trait WitnessWrite<F: Gl64>", "fn set_target(&mut self, target: Target, value: F);",
                 "fn set_bool_target(&mut self, target: BoolTarget, value: bool)", "fn set_hash_target(&mut self, ht: HashOutTarget, value: HashOut<F>)",
                 "fn set_proof_with_pis_target<C: GenericConfig<D, F = F>, const D: usize>(", "fn set_verifier_data_target<C: GenericConfig<D, F = F>, const D: usize>(",
                 "pub fn prove(&self, inputs: PartialWitness<F>) -> anyhow::Result<ProofWithPublicInputs<F, C, D>>",
                 "pub fn verify(&self, proof_with_pis: ProofWithPublicInputs<F, C, D>) -> anyhow::Result<()>", "pub fn from_vec(elements: Vec<Target>) -> Self",
                 "fn two_to_one(left: HashOut<F>, right: HashOut<F>) -> HashOut<F>", "fn hash_or_noop(inputs: &[F]) -> HashOut<F>"):
        assert item in plonk, item
    for fn in ("add_virtual_target", "add_virtual_bool_target_safe", "add_virtual_hash", "one", "connect", "mul", "mul_add", "not", "or", "is_equal",
               "register_public_inputs"):
        assert re.search(r"pub fn %s\(" % fn, plonk), fn
    # every `use plonky2::...` path of the reference's circuit files (as committed strings here: the files themselves are not in this
    # repository) names a module and an item that exist above
    ref_imports = ["hash::poseidon::PoseidonHash", "hash::hash_types::HashOutTarget", "plonk::config::PoseidonGoldilocksConfig",
                   "plonk::config::GenericConfig", "plonk::circuit_data::CircuitData", "plonk::circuit_data::CircuitConfig",
                   "plonk::circuit_data::CommonCircuitData", "plonk::circuit_data::VerifierCircuitTarget", "plonk::circuit_builder::CircuitBuilder",
                   "plonk::proof::ProofWithPublicInputsTarget", "plonk::proof::ProofWithPublicInputs", "iop::target::BoolTarget", "iop::target::Target",
                   "iop::witness::WitnessWrite", "iop::witness::PartialWitness", "hash::hash_types::HashOut", "plonk::config::Hasher"]
    for path in ref_imports:
        mods, item = path.split("::")[:-1], path.split("::")[-1]
        at = 0
        for m in mods:
            at = plonk.index("pub mod %s {" % m, at)
        assert re.search(r"pub (struct|trait) %s\b" % item, plonk[at:]), path
    # every ffi call in the crate passes as many arguments as the extern block declares
    ffi = open(os.path.join(shim, "src", "ffi.rs")).read()
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
