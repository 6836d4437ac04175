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


POSEIDON_GATE_WIRES = 135


def poseidon_gate_witness_batch(inputs, swaps):
    """Wire values of n PoseidonGate rows, wire-major (135, n): the PoseidonGenerator's output for every hash the
    MMR-verifier circuits add (mmr_plonky2_verifier.rs:46-54,81). Layout from recall of plonky2 (parity unpinned)."""
    x = N.as_u64(inputs).reshape(-1, 12)
    sw = np.ascontiguousarray(np.asarray(swaps, dtype=np.uint8)).reshape(-1)
    assert sw.size == x.shape[0]
    out = np.zeros((POSEIDON_GATE_WIRES, x.shape[0]), np.uint64)
    N.check(N.lib().p2mt_poseidon_gate_witness_batch(N.ptr(x), N.ptr(sw), x.shape[0], N.ptr(out)))
    return out
