#!/usr/bin/env python3
"""Outer (recursion) and inner proves for a few different leaves: run under rocprofv3 --kernel-trace to see how the duration of the
proof-of-work grind (k_fri_pow_queue) varies with the transcript state.  usage: grind_probe.py [n_leaves]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
P = pkg.GOLDILOCKS_FIELD_ORDER
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(20, 1)
outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, 1)
for seed in range(n):
    rng = np.random.default_rng(100 + seed)
    leaf = int(rng.integers(0, P, dtype=np.uint64))
    sib = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
    lefts = rng.integers(0, 2, size=20).astype(np.uint8)
    cur = np.array([leaf, 0, 0, 0], np.uint64)
    for s, l in zip(sib, lefts):
        cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
    pw = pkg.PartialWitness()
    pw.set_target(leaf_t, leaf)
    for (ht, bt), s, l in zip(proof_ts, sib, lefts):
        pw.set_hash_target(ht, [int(x) for x in s])
        pw.set_target(bt, int(l))
    for k, t in enumerate(inner.prover_only.public_inputs):
        pw.set_target(t, int(cur[k]))
    ip = inner.prove(pw)
    pwo = pkg.PartialWitness()
    pwo.set_proof_with_pis_target(pt, ip)
    pwo.set_verifier_data_target(vd, inner.verifier_only)
    pwo.set_hash_target(peak_ts[0], [int(x) for x in cur])
    for k, t in enumerate(outer.prover_only.public_inputs):
        pwo.set_target(t, int(cur[k]))
    op = outer.prove(pwo)
    print(seed, "ok", outer.verify(op), flush=True)
