#!/usr/bin/env python3
"""One mmr_plonky2_verifier prove (config 3) and one recursion prove (config 4) in a loop, transcript on the host (default) against
transcript on the device (p2mt_debug_host_transcript(0)): the proofs must be identical word for word; ms per proof of each."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
P = pkg.GOLDILOCKS_FIELD_ORDER
rng = np.random.default_rng(7)
leaf = int(rng.integers(0, P, dtype=np.uint64))
siblings = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
lefts = rng.integers(0, 2, size=20).astype(np.uint8)
cur = np.array([leaf, 0, 0, 0], np.uint64)
for s, l in zip(siblings, lefts):
    cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
case = (leaf, siblings, lefts, cur.reshape(1, 4), cur.copy())
cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(20, 1)
pw = pkg.PartialWitness()
pkg.synthetic.assign_mmr_proof(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
out = {"reps": reps}
proofs = {}
for name, on in (("prove_ms", 1), ("prove_ms_device_transcript", 0), ("prove_ms_again", 1)):
    Nn.check(lib.p2mt_debug_host_transcript(on))
    for _ in range(5):
        proof = cd.prove(pw)
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = cd.prove(pw)
    out[name] = (time.perf_counter() - t0) * 1e3 / reps
    proofs[name] = proof.copy()
    cd.verify(proof)
out["proofs_identical"] = bool(np.array_equal(proofs["prove_ms"], proofs["prove_ms_device_transcript"]) and
                               np.array_equal(proofs["prove_ms"], proofs["prove_ms_again"]))
print(json.dumps(out))
sys.exit(0 if out["proofs_identical"] else 3)
