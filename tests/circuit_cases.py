"""Shared inputs for the circuit / prover tests: MMR membership proofs shaped like the reference's own test driver
(/root/reference/src/mmr/mmr_plonky2_verifier.rs:102-151) and the witness assignment it performs."""
import numpy as np

P = 0xFFFFFFFF00000001


def mmr_case(oracle, n_leaves, leaf_index, seed=None):
    """A real MMR proof from the oracle's MMR: -> (leaf, siblings (k,4), lefts (k,), peaks (m,4), root (4,))"""
    rng = np.random.default_rng(n_leaves if seed is None else seed)
    leaves = rng.integers(0, P, size=n_leaves, dtype=np.uint64)
    m = oracle.mmr(leaves)
    pr = m.get_proof_normal_index(leaf_index)
    return int(leaves[leaf_index]), pr["siblings"], pr["lefts"], pr["peaks"], m.bagging_the_peaks()


def synthetic_case(oracle, n_siblings, seed):
    """A membership proof with n_siblings path elements and one peak without building the 2^n_siblings-leaf MMR: random
    siblings and sides, the peak (= root) is whatever the path folds to -- config 3's shape is n_siblings = 20."""
    rng = np.random.default_rng(seed)
    leaf = int(rng.integers(0, P, dtype=np.uint64))
    siblings = rng.integers(0, P, size=(n_siblings, 4), dtype=np.uint64)
    lefts = rng.integers(0, 2, size=n_siblings).astype(np.uint8)
    cur = np.array([leaf, 0, 0, 0], np.uint64)
    for s, l in zip(siblings, lefts):
        cur = oracle.two_to_one(s, cur) if l else oracle.two_to_one(cur, s)
    return leaf, siblings, lefts, cur.reshape(1, 4), cur.copy()


def assign(leaf_t, proof_ts, peak_ts, public_input_ts, case, set_target):
    """The witness assignment of mmr_plonky2_verifier.rs:122-146 (lives in the package: synthetic.assign_mmr_proof)."""
    import __graft_entry__ as ge
    return ge.load_package().synthetic.assign_mmr_proof(leaf_t, proof_ts, peak_ts, public_input_ts, case, set_target)
