#!/usr/bin/env python3
"""Timeline of the dataflow witness interpreter on the outer recursion circuit (debug hook p2mt_debug_witness_trace): when every
generator finished, by level and kind.  usage: witness_trace.py > gpurun_out/witness_trace.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
P = pkg.GOLDILOCKS_FIELD_ORDER
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
Nn.check(lib.p2mt_debug_witness_trace(outer._h, 1, None, 0, None))
for _ in range(3):
    outer.prove(opw)
cap = 3 * 400000
buf = np.zeros(cap, np.uint64)
n = C.c_size_t(0)
Nn.check(lib.p2mt_debug_witness_trace(outer._h, 1, Nn.ptr(buf), cap, C.byref(n)))
t = buf[:3 * n.value].reshape(-1, 3)
t = t[t[:, 0] != 0xFF]  # padding records of the schedule
kind, level, tick = t[:, 0].astype(int), t[:, 1].astype(int), t[:, 2].astype(np.int64)
t0 = tick[tick > 0].min()
us = (tick - t0) / 100.0
names = ["poseidon", "arith", "equality", "const", "arith_ext", "mul_ext", "quotient_ext", "reducing", "reducing_ext", "wire_split",
         "base_split", "random_access", "interpolation", "poseidon_mds", "reducing_local", "reducing_ext_local", "reducing_combine"]
print("# generators %d, levels %d, span %.1f us (first to last completion)" % (n.value, level.max() + 1, us.max()))
print("# level: n_poseidon n_other | first / last completion (us) | last-completing kind")
prev = 0.0
for l in range(level.max() + 1):
    m = level == l
    if not m.any():
        continue
    last = us[m].max()
    k_last = names[kind[m][us[m].argmax()]]
    per_kind = " ".join("%s:%d@%.0f" % (names[k], int((kind[m] == k).sum()), us[m][kind[m] == k].max()) for k in sorted(set(kind[m])) if k != 0)
    print("%4d: %4d %4d | %8.1f %8.1f | +%6.1f  %s | %s" % (l, int((kind[m] == 0).sum()), int((kind[m] != 0).sum()), us[m].min(), last,
                                                           last - prev, k_last, per_kind))
    prev = max(prev, last)
