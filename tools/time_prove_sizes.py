import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import __graft_entry__ as ge
from oracle_lib import Oracle
from circuit_cases import synthetic_case, assign
pkg = ge.load_package(); pkg.init(0)
o = Oracle()
for n_sib in (20, 200, 1500):
    case = synthetic_case(o, n_sib, 3)
    t0 = time.perf_counter()
    cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(n_sib, 1)
    tb = time.perf_counter() - t0
    pw = pkg.PartialWitness()
    assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
    for _ in range(3): p = cd.prove(pw)
    t0 = time.perf_counter()
    for _ in range(10): p = cd.prove(pw)
    tp = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): ok = cd.verify(p)
    tv = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(5): w = cd.generate_witness(pw)
    tw = (time.perf_counter() - t0) / 5
    print("n_sib %d degree_bits %d build %.1f ms prove %.3f ms verify %.3f ms witness-only %.3f ms proof words %d" % (n_sib, cd.degree_bits, tb*1e3, tp*1e3, tv*1e3, tw*1e3, p.size))
