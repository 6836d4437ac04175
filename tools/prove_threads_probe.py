#!/usr/bin/env python3
"""Concurrent-prover throughput of mmr_plonky2_verifier prove on one GPU: T host threads, each with its own stream
(p2mt_thread_stream_create), circuit handle and witness, proving the same statement in a loop for a few seconds.

usage: prove_threads_probe.py <hipDeviceSchedule flag: -1 keep default (spin), 4 blocking sync> <threads> [seconds] [throughput mode 0/1]
Prints one JSON line.  Run as its own process (bench.py --workload prove does): the device flag has to be set before the
HIP context exists, and blocking sync costs a single prover ~0.1 ms of wake-up latency, so the latency leg keeps spinning."""
import ctypes
import json
import os
import sys
import threading
import time

flag, T = int(sys.argv[1]), int(sys.argv[2])
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
throughput_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 1
if flag >= 0:
    ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(flag)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402
from circuit_cases import assign  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
lib, Nn = pkg.lib(), pkg._native
Nn.check(lib.p2mt_set_throughput_mode(throughput_mode))
# a 20-element membership path with one peak (config 3's shape), folded with the product's own hashing
rng = np.random.default_rng(3)
P = pkg.GOLDILOCKS_FIELD_ORDER
leaf = int(rng.integers(0, P, dtype=np.uint64))
siblings = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
lefts = rng.integers(0, 2, size=20).astype(np.uint8)
cur = np.array([leaf, 0, 0, 0], np.uint64)
for s, l in zip(siblings, lefts):
    cur = pkg.two_to_one(s, cur) if l else pkg.two_to_one(cur, s)
case = (leaf, siblings, lefts, cur.reshape(1, 4), cur.copy())
counts, errs = [0] * T, []
start, stop = threading.Barrier(T + 1), threading.Event()


def worker(i):
    try:
        Nn.check(lib.p2mt_thread_stream_create())
        cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(20, 1)
        pw = pkg.PartialWitness()
        assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
        proof = np.zeros(cd.info.proof_len, np.uint64)
        for _ in range(3):
            Nn.check(lib.p2mt_circuit_prove(cd._h, pw._h, Nn.ptr(proof), proof.size))
        assert cd.verify(proof)
        start.wait()
        while not stop.is_set():
            Nn.check(lib.p2mt_circuit_prove(cd._h, pw._h, Nn.ptr(proof), proof.size))
            counts[i] += 1
        del cd, pw
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
dt = time.perf_counter() - t0
print(json.dumps({"threads": T, "device_schedule_flag": flag, "throughput_mode": throughput_mode, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                  "proofs": int(sum(counts)), "seconds": dt, "proofs_per_s": sum(counts) / dt,
                  "amortised_ms_per_proof": dt * 1e3 / max(sum(counts), 1),
                  "host_cpu_cores_busy": (time.process_time() - c0) / dt, "errors": errs[:2]}))
