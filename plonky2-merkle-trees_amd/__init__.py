"""plonky2-merkle-trees_amd: MI355X-native Poseidon/Goldilocks Merkle tree, MMR and Plonky2 commit kernels.

The product is the HIP library `libp2mt_hip.so` behind the C ABI of include/p2mt.h; this package is the
host-side mirror of the reference's public API over that ABI (ctypes), used by the tests and bench.py.
Import name: `plonky2_merkle_trees_amd` (registered by __graft_entry__.load_package(), since the directory
name carries a hyphen).
"""
from . import _native
from ._native import P2mtError, P2mtPanic, lib
from .hashing import (poseidon_gate_witness_batch, hash_no_pad, hash_no_pad_batch, hash_or_noop, hash_or_noop_batch, poseidon_permute_batch,
                      two_to_one, two_to_one_batch)
from .merkle_tree import MerkleTree, verify_merkle_proof, verify_merkle_proof_batch
from . import circuit, commit, distributed, fri, mmr_plonky2_verifier, mmr_plonky2_verifier_1_recursion, plonk, synthetic
from .circuit import BatchProver, CircuitBuilder, CircuitData, PartialWitness, prove_many
from .mmr_plonky2_verifier import verify_mmr_proof_circuit
from .mmr_plonky2_verifier_1_recursion import complete_verification_circuit_with_inner_proof, verify_inner_merkle_proof_circuit
from .commit import MerkleCapTree, PolynomialBatch, coset_lde, fft, ifft
from .distributed import ShardedMMR
from .fri import Challenger, FriParams, eval_polys_ext, prove_openings
from .plonk import all_wires_permutation_partial_products
from .mmr import (PinnedBuffer, MMR, MMR_proof, get_heights_bitmap_for_mmr_size, get_mmr_index, verify_proof_batch)

GOLDILOCKS_FIELD_ORDER = 18446744069414584321  # src/mmr/common.rs:3


def init(device=0):
    _native.check(lib().p2mt_init(device))


def set_variant(mds, partial):
    _native.check(lib().p2mt_set_variant(mds, partial))


def device_count():
    return lib().p2mt_device_count()


def stage1_info(n_leaves=1 << 24):
    """The dominant launch of a build of n_leaves under the current variant / environment knobs (for bench.py's roofline label):
    per-lane subtrees whose size adapts to the build (p2mt_mmr_stage1_levels), or fused tiles."""
    import ctypes as C
    sub, tile, blk = C.c_int(), C.c_int(), C.c_int()
    _native.check(lib().p2mt_get_build_config(C.byref(sub), C.byref(tile), C.byref(blk)))
    if sub.value:
        lv = int(lib().p2mt_mmr_stage1_levels(n_leaves))
        if lv < 0:
            _native.check(lv)  # a status code, not a level count
        return {"key": "subtree%d" % lv, "levels": lv,
                "kernel": "k_mmr_subtree (stage 1: each lane builds levels 1..%d of its own 2^%d leaves)" % (lv, lv)}
    lv = tile.value - 6
    return {"key": "tile%d" % tile.value, "levels": lv,
            "kernel": "k_mmr_tile (stage 1: levels 1..%d of every 2^%d-leaf tile)" % (lv, tile.value)}
