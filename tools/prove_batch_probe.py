#!/usr/bin/env python3
"""Batched-prover throughput of mmr_plonky2_verifier prove on one GPU: T host threads, each with its own stream, circuit handle
and p2mt_batch_prover of B proofs per pass, proving B different statements per pass in a loop for a few seconds.

usage: prove_batch_probe.py <hipDeviceSchedule flag: -1 keep default (spin), 4 blocking sync> <threads> <batch> [seconds] [throughput mode 0/1]
Prints one JSON line.  Run as its own process (bench.py --workload prove does): the device flag has to precede the HIP context."""
import ctypes
import json
import os
import sys
import threading
import time

flag, T, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
throughput_mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1
if flag >= 0:
    ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(flag)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
Nn.check(lib.p2mt_set_throughput_mode(throughput_mode))
P = pkg.GOLDILOCKS_FIELD_ORDER
assign = pkg.synthetic.assign_mmr_proof


def make_case(seed):
    # a 20-element membership path with one peak (config 3's shape), folded with the product's own hashing
    rng = np.random.default_rng(seed)
    leaf = int(rng.integers(0, P, dtype=np.uint64))
    siblings = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
    lefts = rng.integers(0, 2, size=20).astype(np.uint8)
    cur = np.array([leaf, 0, 0, 0], np.uint64)
    for s, l in zip(siblings, lefts):
        cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
    return leaf, siblings, lefts, cur.reshape(1, 4), cur.copy()


cases = [make_case(1000 + i) for i in range(min(B, 64))]  # statements repeat beyond 64 per pass (the work does not depend on them)
counts, errs = [0] * T, []
start, stop = threading.Barrier(T + 1), threading.Event()


def worker(i):
    try:
        Nn.check(lib.p2mt_thread_stream_create())
        cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(20, 1)
        pws = []
        for k in range(B):
            pw = pkg.PartialWitness()
            assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, cases[(k + i) % len(cases)], pw.set_target)
            pws.append(pw)
        bp = pkg.BatchProver(cd, B)
        warr = (ctypes.c_void_p * B)(*[w._h for w in pws])
        out = np.zeros((B, cd.info.proof_len), np.uint64)
        for _ in range(2):
            Nn.check(lib.p2mt_batch_prover_prove(bp._h, warr, B, Nn.ptr(out), cd.info.proof_len, None))
        assert np.array_equal(out[0], cd.prove(pws[0])) and cd.verify(out[B - 1])
        start.wait()
        while not stop.is_set():
            Nn.check(lib.p2mt_batch_prover_prove(bp._h, warr, B, Nn.ptr(out), cd.info.proof_len, None))
            counts[i] += B
        del bp, cd, pws
        Nn.check(lib.p2mt_thread_stream_destroy())
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
t0, c0 = time.perf_counter(), time.process_time()
time.sleep(seconds)
stop.set()
for t in ths:
    t.join()
dt, cpu = time.perf_counter() - t0, time.process_time() - c0
if errs:
    print(json.dumps({"error": errs[:3]}))
    sys.exit(1)
total = sum(counts)
print(json.dumps({"proofs_per_s": total / dt, "threads": T, "batch": B, "seconds": dt, "proofs": total,
                  "ms_per_pass": dt * 1e3 * T / max(1, total // B), "host_cores_busy": cpu / dt,
                  "sync": "blocking" if flag == 4 else "spin", "throughput_mode": throughput_mode}))
