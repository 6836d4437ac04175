#!/usr/bin/env python3
"""main_circuit_data.verify(final_proof) of the recursion's outer circuit in a loop (ms per verification); run under rocprofv3 --kernel-trace
for the timeline.  usage: verify_outer_probe.py [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
P = pkg.GOLDILOCKS_FIELD_ORDER
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(11)
leaf = int(rng.integers(0, P, dtype=np.uint64))
sib = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
lefts = rng.integers(0, 2, size=20).astype(np.uint8)
cur = np.array([leaf, 0, 0, 0], np.uint64)
for s, l in zip(sib, lefts):
    cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(20, 1)
outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, 1)
pw = pkg.PartialWitness()
pw.set_target(leaf_t, leaf)
for (ht, bt), s, l in zip(proof_ts, sib, lefts):
    pw.set_hash_target(ht, [int(x) for x in s])
    pw.set_target(bt, int(l))
for k in range(4):
    pw.set_target(inner.prover_only.public_inputs[k], int(cur[k]))
ip = inner.prove(pw)
opw = pkg.PartialWitness()
opw.set_proof_with_pis_target(pt, ip)
opw.set_verifier_data_target(vd, inner.verifier_only)
opw.set_hash_target(peak_ts[0], [int(x) for x in cur])
for k, t in enumerate(outer.prover_only.public_inputs):
    opw.set_target(t, int(cur[k]))
fp = outer.prove(opw)
for _ in range(5):
    assert outer.verify(fp)
t0 = time.perf_counter()
for _ in range(reps):
    ok = outer.verify(fp)
dt = (time.perf_counter() - t0) / reps
print({"reps": reps, "verify_outer_ms": dt * 1e3, "accepted": bool(ok)})
