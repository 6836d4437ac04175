"""Host-side mirror of /root/reference/src/simple_merkle_tree/simple_merkle_tree.rs over the C ABI.

Same names, argument meaning and error behaviour as the reference's public API:
  MerkleTree::build (:28-51), get_merkle_proof (:55-74), get_in_between_hashes (:76-86),
  verify_merkle_proof (:91-109).  Where the reference panics, P2mtPanic is raised.
All hashing happens in the HIP library (include/p2mt.h); this file only moves buffers.
"""
import numpy as np

from . import _native as N


class MerkleTree:
    """struct MerkleTree { count_levels, tree: Vec<Vec<HashOut>>, root } (simple_merkle_tree.rs:11-16)."""

    def __init__(self, count_levels, levels_flat, root, n_leaves):
        self.count_levels = count_levels
        self._flat = levels_flat  # level-major (2n-2, 4)
        self.root = root
        self._n = n_leaves

    @property
    def tree(self):
        out, off = [], 0
        for i in range(self.count_levels):
            cnt = self._n >> i
            out.append(self._flat[off:off + cnt])
            off += cnt
        return out

    @staticmethod
    def build(leaves):
        leaves = N.as_u64(leaves).reshape(-1)
        n = leaves.size
        levels = np.zeros((max(2 * n - 2, 1), 4), np.uint64)
        root = np.zeros(4, np.uint64)
        N.check(N.lib().p2mt_merkle_build_pow2(N.ptr(leaves), n, N.ptr(levels), N.ptr(root)))
        return MerkleTree(n.bit_length() - 1, levels, root, n)

    def get_merkle_proof(self, leaf_index):
        out = np.zeros((self.count_levels, 4), np.uint64)
        N.check(N.lib().p2mt_merkle_get_proof(N.ptr(self._flat), self._n, leaf_index, N.ptr(out)))
        return out

    def get_in_between_hashes(self, leaf_index):
        out = np.zeros((self.count_levels, 4), np.uint64)
        N.check(N.lib().p2mt_merkle_get_in_between_hashes(N.ptr(self._flat), N.ptr(self.root), self._n, leaf_index,
                                                          N.ptr(out)))
        return out


def verify_merkle_proof_batch(leaves, leaf_indices, roots, hashes):
    """m independent verify_merkle_proof calls in one launch; hashes: (m, n_hashes, 4)."""
    leaves = N.as_u64(leaves).reshape(-1)
    m = leaves.size
    idx = N.as_u64(leaf_indices).reshape(-1)
    roots = N.as_u64(roots).reshape(m, 4)
    hashes = N.as_u64(hashes).reshape(m, -1, 4)
    res = np.zeros(m, np.uint8)
    N.check(N.lib().p2mt_verify_merkle_proof_batch(N.ptr(leaves), N.ptr(idx), N.ptr(roots), N.ptr(hashes),
                                                   hashes.shape[1], m, N.ptr(res)))
    return res.astype(bool)


def verify_merkle_proof(leaf, leaf_index, root, hashes):
    """simple_merkle_tree.rs:91-109"""
    hashes = N.as_u64(hashes).reshape(-1, 4)
    return bool(verify_merkle_proof_batch([leaf], [leaf_index], N.as_u64(root).reshape(1, 4), hashes[None])[0])
