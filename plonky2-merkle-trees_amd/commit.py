"""Host-side mirror of the commit step that CircuitData::prove drives (mmr_plonky2_verifier.rs:148,
mmr_plonky2_verifier_1_recursion.rs:192,218): plonky2's PolynomialBatch::from_values / from_coeffs,
fft / ifft / coset LDE and MerkleTree::new(leaves, cap_height), over the C ABI (include/p2mt.h)."""
import numpy as np

from . import _native as N

# CircuitConfig::standard_recursion_config() (mmr_plonky2_verifier.rs:30): FRI rate_bits 3, cap_height 4
RATE_BITS = 3
CAP_HEIGHT = 4
COSET_SHIFT = 7  # plonky2_field: MULTIPLICATIVE_GROUP_GENERATOR


def _log2(n):
    if n <= 0 or n & (n - 1):
        raise N.P2mtPanic(N.P2MT_EINVAL, "log2_strict: %d is not a power of two" % n)
    return n.bit_length() - 1


def fft(polys):
    """fft_with_options: rows of coefficients -> values at w^i (natural order)."""
    a = N.as_u64(polys).copy()
    a2 = a.reshape(-1, a.shape[-1])
    N.check(N.lib().p2mt_ntt_batch(N.ptr(a2), _log2(a2.shape[1]), a2.shape[0], 0))
    return a


def ifft(polys):
    a = N.as_u64(polys).copy()
    a2 = a.reshape(-1, a.shape[-1])
    N.check(N.lib().p2mt_ntt_batch(N.ptr(a2), _log2(a2.shape[1]), a2.shape[0], 1))
    return a


def coset_lde(coeffs, rate_bits=RATE_BITS, shift=COSET_SHIFT):
    c = N.as_u64(coeffs)
    c2 = c.reshape(-1, c.shape[-1])
    out = np.zeros((c2.shape[0], c2.shape[1] << rate_bits), np.uint64)
    N.check(N.lib().p2mt_coset_lde_batch(N.ptr(c2), _log2(c2.shape[1]), rate_bits, shift, c2.shape[0], N.ptr(out)))
    return out.reshape(c.shape[:-1] + (out.shape[1],))


def _n_digests(n, cap_height):
    k = _log2(n)
    return sum(n >> j for j in range(max(k - cap_height, 0)))


class MerkleCapTree:
    """plonky2 MerkleTree { leaves, digests, cap }; digests are level-major here (level 0 = leaf digests)."""

    def __init__(self, leaves, digests, cap, cap_height):
        self.leaves, self.digests, self.cap, self.cap_height = leaves, digests, cap, cap_height

    @staticmethod
    def new(leaves, cap_height=CAP_HEIGHT):
        leaves = N.as_u64(leaves)
        n, w = leaves.shape
        k = _log2(n)
        if cap_height > k:
            raise N.P2mtPanic(N.P2MT_EINVAL, "cap_height exceeds tree height")
        nd = _n_digests(n, cap_height)
        digests = np.zeros((max(nd, 1), 4), np.uint64)
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        N.check(N.lib().p2mt_merkle_cap_commit(N.ptr(leaves), n, w, cap_height, N.ptr(digests), N.ptr(cap)))
        return MerkleCapTree(leaves, digests[:nd], cap, cap_height)

    def plonky2_digests(self):
        """`MerkleTree.digests` in plonky2's own order (hash/merkle_tree.rs fill_subtree: per cap subtree, recursively, left
        subtree || left child || right child || right subtree), converted on the device from the level-major array."""
        nd = self.digests.shape[0]
        out = np.zeros((max(nd, 1), 4), np.uint64)
        lm = N.as_u64(self.digests) if nd else out
        N.check(N.lib().p2mt_merkle_digests_to_plonky2_layout(N.ptr(lm), self.leaves.shape[0], self.cap_height, N.ptr(out)))
        return out[:nd]

    def prove(self, leaf_index):
        """Merkle path of a leaf up to (excluding) the cap: sibling digests bottom-up."""
        n = self.leaves.shape[0]
        k = _log2(n)
        out, off, idx = [], 0, leaf_index
        for j in range(k - self.cap_height):
            out.append(self.digests[off + (idx ^ 1)])
            off += n >> j
            idx >>= 1
        return np.array(out, dtype=np.uint64).reshape(-1, 4)


class PolynomialBatch:
    """plonky2 fri/oracle.rs PolynomialBatch (blinding = false): LDE of every polynomial on the coset 7*<w_N>,
    leaf i (bit-reversed order) = all polynomials at one point, Merkle tree with a cap."""

    def __init__(self, tree, n_polys, degree_log, rate_bits, polynomials=None):
        self.merkle_tree, self.n_polys, self.degree_log, self.rate_bits = tree, n_polys, degree_log, rate_bits
        self.polynomials = polynomials  # coefficients [n_polys][n], what prove_openings composes (fri.py)

    @staticmethod
    def _commit(polys, is_values, rate_bits, cap_height, want_leaves):
        polys = N.as_u64(polys)
        n_polys, n = polys.shape
        log_n = _log2(n)
        big = n << rate_bits
        if cap_height > log_n + rate_bits:
            raise N.P2mtPanic(N.P2MT_EINVAL, "cap_height exceeds tree height")
        leaves = np.zeros((big, n_polys), np.uint64) if want_leaves else None
        nd = _n_digests(big, cap_height)
        digests = np.zeros((max(nd, 1), 4), np.uint64)
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        N.check(N.lib().p2mt_polynomial_batch_commit(N.ptr(polys), int(is_values), n_polys, log_n, rate_bits,
                                                     cap_height, N.ptr(leaves), N.ptr(digests), N.ptr(cap)))
        return PolynomialBatch(MerkleCapTree(leaves, digests[:nd], cap, cap_height), n_polys, log_n, rate_bits,
                               ifft(polys) if is_values else polys)

    @staticmethod
    def from_values(values, rate_bits=RATE_BITS, cap_height=CAP_HEIGHT, want_leaves=True):
        return PolynomialBatch._commit(values, True, rate_bits, cap_height, want_leaves)

    @staticmethod
    def from_coeffs(coeffs, rate_bits=RATE_BITS, cap_height=CAP_HEIGHT, want_leaves=True):
        return PolynomialBatch._commit(coeffs, False, rate_bits, cap_height, want_leaves)
