#!/usr/bin/env python3
"""circuit_data.verify(proof) of the config-3 circuit in a loop (one proof at a time), for a per-kernel timeline under
rocprofv3 --kernel-trace (tools/rocpd_timeline.py <db> k_verify_items) and the un-profiled ms next to it."""
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
proof = cd.prove(pw)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
acc, reason = C.c_int(0), C.c_int(0)
out = {"reps": reps}
for name, on in (("verify_ms", 1), ("verify_ms_device_transcript", 0), ("verify_ms_again", 1)):
    Nn.check(lib.p2mt_debug_host_transcript(on))
    for _ in range(5):
        Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
    assert acc.value == 1
    t0 = time.perf_counter()
    for _ in range(reps):
        Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
    out[name] = (time.perf_counter() - t0) * 1e3 / reps
    bad = proof.copy()
    bad[200] ^= np.uint64(1)
    Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(bad), bad.size, C.byref(acc), C.byref(reason)))
    out[name + "_tampered"] = [acc.value, reason.value]
Nn.check(lib.p2mt_debug_host_transcript(1))
st = np.arange(12, dtype=np.uint64)
t0 = time.perf_counter()
for _ in range(2000):
    Nn.check(lib.p2mt_host_poseidon_permute(Nn.ptr(st), Nn.ptr(st), 1))
out["host_permutation_us_incl_ctypes"] = (time.perf_counter() - t0) * 1e6 / 2000
big = np.tile(np.arange(12, dtype=np.uint64), 20000)
t0 = time.perf_counter()
Nn.check(lib.p2mt_host_poseidon_permute(Nn.ptr(big), Nn.ptr(big), 20000))
out["host_permutation_us"] = (time.perf_counter() - t0) * 1e6 / 20000
print(json.dumps(out))
