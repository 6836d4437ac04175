"""Synthetic inputs of the benchmarks and tests: seeded leaves and the witness
assignment the reference's own test driver performs.  Pure host code (numpy); no hashing happens here."""
import numpy as np

GOLDILOCKS_FIELD_ORDER = 0xFFFFFFFF00000001


def splitmix_leaves(n, seed):
    """n uniform Goldilocks elements: SplitMix64(seed) with rejection of values >= p
    (mirrors rng.gen_range(0..GOLDILOCKS_FIELD_ORDER), /root/reference/src/mmr/merkle_mountain_ranges.rs:336)."""
    out = np.empty(n, dtype=np.uint64)
    filled = 0
    state = np.uint64(seed)
    with np.errstate(over="ignore"):
        while filled < n:
            m = max(1024, int((n - filled) * 1.01))
            idx = np.arange(1, m + 1, dtype=np.uint64)
            z = state + idx * np.uint64(0x9E3779B97F4A7C15)
            state = z[-1]
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            z = z[z < np.uint64(GOLDILOCKS_FIELD_ORDER)]
            take = min(z.size, n - filled)
            out[filled:filled + take] = z[:take]
            filled += take
    return out


def bench_leaves(log_leaves, rank=0):
    """The bench input of rank `rank`: 2^log_leaves leaves, seed 0x5EED0000 + 24 + 1000 * rank (bench.py, tests)."""
    return splitmix_leaves(1 << log_leaves, 0x5EED0000 + 24 + 1000 * rank)


def assign_mmr_proof(leaf_t, proof_ts, peak_ts, public_input_ts, case, set_target):
    """The witness assignment of /root/reference/src/mmr/mmr_plonky2_verifier.rs:122-146 through set_target(target, value).
    case = (leaf, siblings (k,4), lefts (k,), peaks (m,4), root (4,))."""
    leaf, siblings, lefts, peaks, root = case
    set_target(leaf_t, int(leaf))
    for (ht, bt), sib, left in zip(proof_ts, siblings, lefts):
        for k in range(4):
            set_target(ht[k], int(sib[k]))
        set_target(bt, int(left))
    for pt, pk in zip(peak_ts, peaks):
        for k in range(4):
            set_target(pt[k], int(pk[k]))
    for k, t in enumerate(public_input_ts):
        set_target(t, int(root[k]))
