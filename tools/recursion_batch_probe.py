#!/usr/bin/env python3
"""Throughput of mmr_plonky2_verifier_1_recursion through the batched prover on one GPU: per pass B inner proofs
(p2mt_batch_prover on the inner circuit), B outer witnesses (set_proof_with_pis_target + verifier data + peaks + root on the host),
B outer proofs (p2mt_batch_prover on the 2^12-row outer circuit).  One host thread.

usage: recursion_batch_probe.py <batch> [seconds] [threads]     Prints one JSON line (recursion proofs/s and the one-at-a-time time).
With threads > 1 every thread owns its stream, circuits and provers (the one-workgroup witness interpreter of one thread's outer pass
overlaps the other thread's hashing)."""
import json
import os
import sys
import time

import threading

B = int(sys.argv[1])
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
Nn.check(lib.p2mt_set_throughput_mode(1))
P = pkg.GOLDILOCKS_FIELD_ORDER


def make_case(seed, n_sib=20):
    rng = np.random.default_rng(seed)
    leaf = int(rng.integers(0, P, dtype=np.uint64))
    siblings = rng.integers(0, P, size=(n_sib, 4), dtype=np.uint64)
    lefts = rng.integers(0, 2, size=n_sib).astype(np.uint8)
    cur = np.array([leaf, 0, 0, 0], np.uint64)
    for s, l in zip(siblings, lefts):
        cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
    return leaf, siblings, lefts, cur.reshape(1, 4), cur.copy()


def make_worker(tid):
    """circuits, witnesses and provers of one thread -> (one_pass, single)"""
    cases = [make_case(2000 + 1000 * tid + i) for i in range(B)]
    inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(20, 1)
    outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, 1)
    ipws = []
    for leaf, sib, lefts, peaks, root in cases:
        pw = pkg.PartialWitness()
        pw.set_target(leaf_t, leaf)
        for (ht, bt), s, l in zip(proof_ts, sib, lefts):
            pw.set_hash_target(ht, [int(x) for x in s])
            pw.set_target(bt, int(l))
        for k in range(4):
            pw.set_target(inner.prover_only.public_inputs[k], int(peaks[0][k]))
        ipws.append(pw)
    opws = [pkg.PartialWitness() for _ in cases]
    bi, bo = pkg.BatchProver(inner, B), pkg.BatchProver(outer, B)

    def one_pass():
        inner_proofs = bi.prove(ipws)
        for pw, ip, (leaf, sib, lefts, peaks, root) in zip(opws, inner_proofs, cases):
            pw.clear()
            pw.set_proof_with_pis_target(pt, ip)
            pw.set_verifier_data_target(vd, inner.verifier_only)
            pw.set_hash_target(peak_ts[0], [int(x) for x in peaks[0]])
            for k, t in enumerate(outer.prover_only.public_inputs):
                pw.set_target(t, int(root[k]))
        return bo.prove(opws)

    def single():
        t0 = time.perf_counter()
        inner.prove(ipws[0])
        p = outer.prove(opws[0])
        return p, (time.perf_counter() - t0) * 1e3, outer

    return one_pass, single


counts, errs, single_ms = [0] * T, [], [0.0]
start, stop = threading.Barrier(T + 1), threading.Event()


def worker(tid):
    try:
        if T > 1:
            Nn.check(lib.p2mt_thread_stream_create())
        one_pass, single = make_worker(tid)
        proofs = one_pass()
        sp, ms, outer = single()  # one at a time, for the comparison and as the parity check of this run
        assert np.array_equal(sp, proofs[0]) and outer.verify(proofs[B - 1])
        if tid == 0:
            single_ms[0] = ms
        one_pass()
        start.wait()
        while not stop.is_set():
            one_pass()
            counts[tid] += B
    except Exception as e:
        errs.append(repr(e))
        stop.set()
        try:
            start.abort()
        except Exception:
            pass


ths = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in ths:
    t.start()
try:
    start.wait()
except threading.BrokenBarrierError:
    pass
t0 = time.perf_counter()
time.sleep(seconds)
stop.set()
for t in ths:
    t.join()
dt = time.perf_counter() - t0
if errs:
    print(json.dumps({"error": errs[:3]}))
    sys.exit(1)
n = sum(counts)
print(json.dumps({"recursion_proofs_per_s": n / dt, "batch": B, "threads": T, "ms_per_pass": dt * 1e3 * B * T / max(n, 1),
                  "ms_per_proof_amortised": dt * 1e3 / max(n, 1), "ms_one_at_a_time": single_ms[0], "proofs": n, "seconds": dt}))
