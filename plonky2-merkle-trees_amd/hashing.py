"""Batch-shaped mirror of plonky2's `Hasher` for PoseidonHash (two_to_one / hash_or_noop / hash_no_pad),
as used at simple_merkle_tree.rs:23,33,45 and merkle_mountain_ranges.rs:91,96,111,125."""
import numpy as np

from . import _native as N


def poseidon_permute_batch(states):
    s = N.as_u64(states).reshape(-1, 12)
    out = np.zeros_like(s)
    N.check(N.lib().p2mt_poseidon_permute_batch(N.ptr(s), N.ptr(out), s.shape[0]))
    return out


def two_to_one_batch(pairs):
    p = N.as_u64(pairs).reshape(-1, 8)
    out = np.zeros((p.shape[0], 4), np.uint64)
    N.check(N.lib().p2mt_two_to_one_batch(N.ptr(p), N.ptr(out), p.shape[0]))
    return out


def two_to_one(left, right):
    return two_to_one_batch(np.concatenate([N.as_u64(left).reshape(4), N.as_u64(right).reshape(4)]))[0]


def hash_or_noop_batch(rows):
    r = N.as_u64(rows)
    r = r.reshape(r.shape[0], -1)
    out = np.zeros((r.shape[0], 4), np.uint64)
    N.check(N.lib().p2mt_hash_or_noop_batch(N.ptr(r), r.shape[0], r.shape[1], N.ptr(out)))
    return out


def hash_no_pad_batch(rows):
    r = N.as_u64(rows)
    r = r.reshape(r.shape[0], -1)
    out = np.zeros((r.shape[0], 4), np.uint64)
    N.check(N.lib().p2mt_hash_no_pad_batch(N.ptr(r), r.shape[0], r.shape[1], N.ptr(out)))
    return out


def hash_or_noop(x):
    return hash_or_noop_batch(N.as_u64(x).reshape(1, -1))[0]


def hash_no_pad(x):
    return hash_no_pad_batch(N.as_u64(x).reshape(1, -1))[0]
