#!/usr/bin/env python3
"""Critical path of a circuit's witness generation (CPU, oracle side): the longest dependency chain through the generators, as
(PoseidonGate rows, other generators) -- the floor of any witness interpreter is chain_poseidon x (one wave-permutation) +
chain_other x (one dependent hop).  usage: critical_path.py [inner|outer] [n_siblings]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import oracle_lib  # noqa: E402
from circuit_cases import synthetic_case  # noqa: E402
from oracle import circuit as OC, recursion as R  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "outer"
n_sib = int(sys.argv[2]) if len(sys.argv) > 2 else 20
o = oracle_lib.Oracle()
case = synthetic_case(o, n_sib, 5)
leaf, sib, lefts, peaks, root = case
inner, leaf_t, proof_ts = OC.verify_inner_merkle_proof_circuit(o, n_sib, 1)
opw = {leaf_t: leaf}
for (ht, bt), s, l in zip(proof_ts, sib, lefts):
    for k in range(4):
        opw[ht[k]] = int(s[k])
    opw[bt] = int(l)
for k in range(4):
    opw[inner.public_inputs[k]] = int(peaks[0][k])
cd, pw = inner, opw
if which == "outer":
    ip = inner.prove(opw)
    outer, pt, vd, peak_ts = R.complete_verification_circuit_with_inner_proof(o, R.CommonData(inner), 1)
    pw = {}
    R.set_proof_with_pis_target(pw.__setitem__, pt, ip)
    R.set_verifier_data_target(pw.__setitem__, vd, inner)
    for k in range(4):
        pw[peak_ts[0][k]] = int(peaks[0][k])
        pw[outer.public_inputs[k]] = int(root[k])
    cd = outer

# replay generate_witness with a clock per partition: (poseidon count, other count) of the longest chain that produced it
f, find, tidx = cd.forest, cd.forest.find, cd._tidx
vals, clock = {}, {}
watchers, missing, ready = {}, [], []
now = [(0, 0)]


def weight(c):  # a PoseidonGate row ~ 9.5 us, another generator ~ 2 us through a global-memory table
    return 9.5 * c[0] + 2.0 * c[1]


def setv(t, v):
    r = find(tidx(t))
    v %= OC.P
    if r in vals:
        return
    vals[r] = v
    clock[r] = now[0]
    for gi in watchers.pop(r, ()):
        missing[gi] -= 1
        if missing[gi] == 0:
            ready.append(gi)


def getv(t):
    return vals.get(find(tidx(t)))


deps_of = []
for gi, gen in enumerate(cd.generators):
    deps = {find(tidx(t)) for t in cd._gen_io(gen)[0]}
    deps_of.append(deps)
    missing.append(len(deps))
    for r in deps:
        watchers.setdefault(r, []).append(gi)
    if not deps:
        ready.append(gi)
for t, v in pw.items():
    setv(t, int(v))
best = (0, 0)
kinds = {}
while ready:
    gi = ready.pop()
    gen = cd.generators[gi]
    start = max((clock[r] for r in deps_of[gi]), key=weight, default=(0, 0))
    now[0] = (start[0] + 1, start[1]) if gen[0] == "poseidon" else (start[0], start[1] + (0 if gen[0] == "const" else 1))
    kinds[gen[0]] = kinds.get(gen[0], 0) + 1
    cd._run_generator(gen, getv, setv)
    if weight(now[0]) > weight(best):
        best = now[0]
print("circuit: %s, 2^%d rows, %d generators %s" % (which, cd.degree_bits, len(cd.generators), kinds))
print("critical path: %d PoseidonGate rows + %d other generators  ~ %.0f us at 9.5 us per permutation and 2 us per dependent hop"
      % (best[0], best[1], weight(best)))
