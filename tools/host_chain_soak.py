#!/usr/bin/env python3
"""Soak: N random statements; inner and outer proofs with the PoseidonGate chains on the host (default) against every generator on
the device (p2mt_debug_host_chain(0)): the proof words must be identical, and the outer proof must verify.  usage: host_chain_soak.py [N]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
P = pkg.GOLDILOCKS_FIELD_ORDER
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(20, 1)
outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, 1)
bad = 0
for seed in range(n):
    rng = np.random.default_rng(4000 + seed)
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
    for k in range(4):
        pw.set_target(inner.prover_only.public_inputs[k], int(cur[k]))
    res = {}
    for mode in (1, 0):
        Nn.check(lib.p2mt_debug_host_chain(mode))
        ip = inner.prove(pw)
        opw = pkg.PartialWitness()
        opw.set_proof_with_pis_target(pt, ip)
        opw.set_verifier_data_target(vd, inner.verifier_only)
        opw.set_hash_target(peak_ts[0], [int(x) for x in cur])
        for k, t in enumerate(outer.prover_only.public_inputs):
            opw.set_target(t, int(cur[k]))
        op = outer.prove(opw)
        res[mode] = (ip, op)
    Nn.check(lib.p2mt_debug_host_chain(1))
    same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    ok = bool(inner.verify(res[1][0])) and bool(outer.verify(res[1][1]))
    bad += (not same) or (not ok)
    print(seed, "identical", same, "verified", ok, flush=True)
print({"statements": n, "failures": bad})
sys.exit(1 if bad else 0)
