"""oracle/merkle_cap.py -- TEST INFRASTRUCTURE.  [parity unpinned: plonky2 source absent]

Restatement of plonky2 (git rev 3b21b87d, NOT in /root/reference) hash/merkle_tree.rs MerkleTree::new -> fill_digests_buf ->
fill_subtree and MerkleTree::prove, i.e. the order in which plonky2 itself stores `MerkleTree.digests`:
per cap subtree, recursively, "left recursive output || left child digest || right child digest || right recursive output".
Reference call sites that reach it: every PolynomialBatch commitment inside CircuitData::prove
(/root/reference/src/mmr/mmr_plonky2_verifier.rs:148, mmr_plonky2_verifier_1_recursion.rs:192,218).
Small cases only (plain Python recursion); hashing through tests/oracle_lib.Oracle."""
import numpy as np


def fill_subtree(o, buf, lo, hi, leaves):
    """fill_subtree(digests_buf[lo:hi], leaves) -> digest of the subtree's root; writes the 2*(len(leaves)-1) digests"""
    assert len(leaves) == (hi - lo) // 2 + 1
    if hi == lo:
        return o.hash_or_noop(leaves[0])
    mid = lo + (hi - lo) // 2
    half = len(leaves) // 2
    left = fill_subtree(o, buf, lo, mid - 1, leaves[:half])      # left_digests_buf = left half minus its last slot
    right = fill_subtree(o, buf, mid + 1, hi, leaves[half:])     # right_digests_buf = right half minus its first slot
    buf[mid - 1] = left
    buf[mid] = right
    return o.two_to_one(left, right)


def merkle_tree_new(o, leaves, cap_height):
    """MerkleTree::new(leaves, cap_height) -> (digests in plonky2's order (2*(n - 2^cap_height), 4), cap (2^cap_height, 4))"""
    leaves = np.asarray(leaves, np.uint64)
    n = leaves.shape[0]
    n_cap = 1 << cap_height
    assert n & (n - 1) == 0 and n_cap <= n
    num_digests = 2 * (n - n_cap)
    digests = np.zeros((num_digests, 4), np.uint64)
    cap = np.zeros((n_cap, 4), np.uint64)
    sub_d, sub_l = num_digests >> cap_height, n >> cap_height
    for s in range(n_cap):
        cap[s] = fill_subtree(o, digests, s * sub_d, (s + 1) * sub_d, leaves[s * sub_l:(s + 1) * sub_l])
    return digests, cap


def prove(digests, n_leaves, cap_height, leaf_index):
    """MerkleTree::prove: sibling digests bottom-up, indexing `digests` the way plonky2 does"""
    num_layers = (n_leaves.bit_length() - 1) - cap_height
    subtree_digest_size = (1 << (num_layers + 1)) - 2
    subtree_idx = leaf_index >> num_layers
    base = subtree_idx * subtree_digest_size
    pair_index = leaf_index & ((1 << num_layers) - 1)
    out = []
    for i in range(num_layers):
        parity = pair_index & 1
        pair_index >>= 1
        siblings_index = (pair_index << (i + 1)) + (1 << i) - 1
        out.append(digests[base + 2 * siblings_index + (1 - parity)])
    return np.array(out, np.uint64).reshape(-1, 4)
