#!/usr/bin/env python3
"""bench.py -- Poseidon hashes/s of a full MMR build (BASELINE.json metric), one process per GPU.

A "step" is one complete build of a 2^LOG-leaf MMR (default 2^24, the size BASELINE.json's metric is quoted
on) from leaves already resident in HBM: reset + extend == 2^24 x MMR::add_leaf
(/root/reference/src/mmr/merkle_mountain_ranges.rs:89-120) + bagging_the_peaks, + the all-gather/top-levels
combine when N > 1.  Unit of work: one `two_to_one` Poseidon permutation; an N-leaf build performs
N - popcount(N) of them.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (weak scaling: every rank builds its own 2^LOG shard)
  python bench.py --workload prove                        (BASELINE's second metric: ms/proof mmr_plonky2_verifier)
  python bench.py --workload commit | fri                 (parts of a prove at the d = 12 shape: commit phase, opening proof)

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as ge  # noqa: E402

ALGO_BYTES_PER_HASH = 72.0   # 8 B leaf in + 2 x 32 B nodes out per two_to_one, N large
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
# Integer-issue roofline: VALU instructions per two_to_one of the shipped kernel, from
# SQ_INSTS_VALU / hashes (profiles/r01_final_*.txt), and the measured gfx950 issue rates
# (profiles/r01_valu_issue_rates_gfx950.txt): ~2.05 wave-instr/CU/ns for v_mad_u64_u32-class ops, ~4.0 for
# add/sub/xor/mov.  The 85 % / 15 % mix of the kernel predicts ~2.3; the best rate any of our kernels sustains in situ is
# 2.33 (verify_batch: 24 chained two_to_one per lane, 2.35 G hashes/s), so 2.4 wave-instr/CU/ns is used as the roof.
VALU_INSTR_PER_HASH = 16190.0
ISSUE_PEAK_WAVE_INSTR_PER_S = 256 * 2.4e9
# HBM bytes of ONE stage-1 launch (tile_log 10, 2^24 leaves) from the PMC passes in
# profiles/r01_final_mmr_build_2p24.txt: FETCH_SIZE 75.0 MB x2 (gfx950 streaming-read correction) + WRITE_SIZE 1346.9 MB.
# Counters cannot be read live from inside the process, so this is the committed measurement; it is reported only
# for the configuration it was measured on.
# k_mmr_subtree: FETCH_SIZE 444.8 MB raw (889.5 MB with the x2 correction, which is calibrated for coalesced streams only;
# the leaf reads here are lane-strided) + WRITE_SIZE 1288.4 MB.  k_mmr_tile (P2MT_SUBTREE=0): 150.1 + 1346.9 MB.
MEASURED_TRAFFIC_BYTES_PER_LAUNCH = {("subtree4", 24): 889.5e6 + 1288.4e6, ("tile10", 24): 150.1e6 + 1346.9e6}


def splitmix_leaves(n, seed):
    from conftest import splitmix_leaves as f
    return f(n, seed)


def cpu_baseline_mmr(target_seconds=12.0, max_log=22):
    """Oracle (C restatement, `for leaf { add_leaf }`, 1 thread) on a bounded sample of the same workload."""
    from oracle_lib import Oracle
    o = Oracle()
    log_n = 16
    leaves = splitmix_leaves(1 << log_n, 0x5EED0000 + 24)
    t0 = time.perf_counter()
    m = o.mmr(leaves)
    dt = time.perf_counter() - t0
    rate = ((1 << log_n) - 1) / dt
    while log_n < max_log and (2 << log_n) / rate < target_seconds:
        log_n += 1
    leaves = splitmix_leaves(1 << log_n, 0x5EED0000 + 24)
    t0 = time.perf_counter()
    m = o.mmr(leaves)
    dt = time.perf_counter() - t0
    hashes = (1 << log_n) - 1
    # B2: "generous" all-core, level-parallel build of the same array (not the reference's algorithm)
    # The host may expose far more hardware threads than its CPU quota lets run (observed: 256 visible, ~16 cores'
    # worth of throughput), so the thread count is calibrated on a small build and the best one is used.
    el = np.empty((2 * leaves.size - 1, 4), np.uint64)
    cal = leaves[:1 << 17]
    best_t, best_rate = 1, 0.0
    for t in sorted({t for t in (8, 16, 32, 64, 128, os.cpu_count() or 1) if t <= (os.cpu_count() or 1)}):
        t0 = time.perf_counter()
        o.mmr_build_pow2_parallel(cal, t, el[:2 * cal.size - 1])
        r = cal.size / (time.perf_counter() - t0)
        if r > best_rate:
            best_t, best_rate = t, r
    t0 = time.perf_counter()
    el, threads = o.mmr_build_pow2_parallel(leaves, best_t, el)
    dt2 = time.perf_counter() - t0
    assert np.array_equal(el[-1], m.bagging_the_peaks())
    return {"value": hashes / dt, "unit": "Poseidon hashes/s", "cores": 1, "kind": "port",
            "sample": "oracle/mmr.c add_leaf loop (B1, faithful: the reference is single-threaded), first 2^%d "
                      "leaves of the bench input, %.1f s" % (log_n, dt),
            "all_cores": {"value": hashes / dt2, "cores": threads, "seconds": dt2,
                          "what": "B2: level-parallel OpenMP build of the same node array (generous, not the "
                                  "reference's algorithm)"},
            "_root": m.bagging_the_peaks(), "_log_n": log_n}


def run_mmr(args, torch, pkg, lib, rank, world, local_rank, dist):
    n = 1 << args.log_leaves
    host_leaves = splitmix_leaves(n, 0x5EED0000 + 24 + 1000 * rank)
    d_leaves = torch.from_numpy(host_leaves.view(np.int64)).cuda()
    shard = pkg.ShardedMMR(pkg, n, rank, world, dist)

    def step():
        return shard.build_dev(d_leaves)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    lib.p2mt_profile_enable(1)
    pkg._native.check(lib.p2mt_timer_start())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        root = step()
    region_ms = C.c_float(0)
    pkg._native.check(lib.p2mt_timer_stop(C.byref(region_ms)))
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = C.c_float(0), C.c_int(0)
    pkg._native.check(lib.p2mt_profile_read(C.byref(kern_ms), C.byref(kern_n)))
    lib.p2mt_profile_enable(0)
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hashes_per_rank = n - bin(n).count("1")
    total_hashes = hashes_per_rank * world + (world - 1)
    ms_per_step = elapsed * 1e3 / args.steps
    value = total_hashes / (ms_per_step * 1e-3)
    if rank != 0:
        return None
    mds, partial = C.c_int(), C.c_int()
    lib.p2mt_get_variant(C.byref(mds), C.byref(partial))
    tile_log = int(os.environ.get("P2MT_TILE_LOG", "10"))
    subtree = int(os.environ.get("P2MT_SUBTREE", "4"))
    # dominant kernel = stage 1: per-lane subtrees (levels 1..4 of every 16-leaf block, default) or, with
    # P2MT_SUBTREE=0, the fused LDS tile kernel (levels 1 .. tile_log-6 of every 2^tile_log-leaf tile)
    fused_levels = subtree if subtree in (4, 5) else tile_log - 6
    stage1 = ("k_mmr_subtree (stage 1: each lane builds levels 1..%d of its own 2^%d leaves)" % (fused_levels, fused_levels)
              if subtree in (4, 5) else
              "k_mmr_tile (stage 1: levels 1..%d of every 2^%d-leaf tile)" % (fused_levels, tile_log))
    # (with the chunked two-stream build there are several stage-1 launches per step; hashes are split evenly)
    launches_per_step = max(kern_n.value, 1) / float(args.steps)
    hashes_in_launch = (n - (n >> fused_levels)) / launches_per_step
    launch_ms = kern_ms.value / max(kern_n.value, 1)
    algo_bytes = hashes_in_launch * ALGO_BYTES_PER_HASH
    achieved_gbs = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    in_kernel_rate = hashes_in_launch / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
    out = {
        "metric": "Poseidon hashes/s (MMR build, 2^%d leaves per GPU)" % args.log_leaves,
        "value": value, "unit": "Poseidon hashes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64 (Goldilocks, integer VALU)", "data": "synthetic",
        "config": {"workload": "mmr::merkle_mountain_ranges build, 2^%d leaves per GPU, device-resident leaves"
                               % args.log_leaves,
                   "leaves_per_gpu": n, "hashes_per_step": total_hashes,
                   "poseidon_variant": {"mds": mds.value, "partial": partial.value},
                   "stage1": "subtree%d" % subtree if subtree in (4, 5) else "tile%d" % tile_log,
                   "sharding": "leaf ranges per rank + all-gather of 32-byte shard roots" if world > 1 else "none"},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS,
                     "traffic": (MEASURED_TRAFFIC_BYTES_PER_LAUNCH.get(("subtree%d" % subtree if subtree in (4, 5)
                                                                        else "tile%d" % tile_log, args.log_leaves)) or 0)
                     / launches_per_step or None,
                     "kernel": stage1,
                     "launch_ms": launch_ms, "launches_timed": kern_n.value,
                     "algorithmic_bytes_per_launch": algo_bytes, "hashes_per_launch": hashes_in_launch,
                     "note": "Poseidon is integer-issue bound (see issue_roofline): ~16k VALU instructions per "
                             "72 algorithmic bytes, so the HBM fraction is ~1-2 % by construction (SURVEY.md 8d)"},
        "issue_roofline": {"bound": "valu-issue", "unit": "wave-instr/s",
                           "achieved": in_kernel_rate * VALU_INSTR_PER_HASH / 64.0,
                           "peak": ISSUE_PEAK_WAVE_INSTR_PER_S,
                           "frac": in_kernel_rate * VALU_INSTR_PER_HASH / 64.0 / ISSUE_PEAK_WAVE_INSTR_PER_S,
                           "valu_instr_per_hash": VALU_INSTR_PER_HASH, "in_kernel_hashes_per_s": in_kernel_rate},
        "device_ms_per_step": region_ms.value / args.steps,
        "root": [int(x) for x in root],
    }
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline_mmr()
        # parity check of the sample: GPU build of the same prefix must give the oracle's root
        sub = pkg.MMR.from_leaves(host_leaves[:1 << cb["_log_n"]])
        assert np.array_equal(sub.bagging_the_peaks(), cb["_root"]), "GPU root != oracle root on the CPU sample"
        cb = {k: v for k, v in cb.items() if not k.startswith("_")}
        cb["gpu_over_cpu"] = value / cb["value"]
        cb["all_cores"]["gpu_over_cpu"] = value / cb["all_cores"]["value"]
        out["cpu_baseline"] = cb
    if world == 1 and not args.no_prove:
        # BASELINE.json's second metric (ms/proof mmr_plonky2_verifier), measured after and outside the timed region above
        try:
            import copy
            pa = copy.copy(args)
            pa.steps, pa.warmup = 30, 5
            pr = run_prove(pa, torch, pkg, lib, cpu_seconds=3.0)
            out["ms_per_proof_mmr_plonky2_verifier"] = {
                "value": pr["value"], "unit": "ms", "config": pr["config"]["workload"],
                "throughput": pr.get("throughput"), "cpu_baseline": pr.get("cpu_baseline"),
                "how": "python bench.py --workload prove"}
        except Exception as e:  # the headline line must survive a failure of the secondary leg
            out["ms_per_proof_mmr_plonky2_verifier"] = {"error": repr(e)}
    return out


def run_commit(args, torch, pkg, lib):
    """Secondary metric (ms/proof, commit phase only): the three PolynomialBatch commits of one prove at the
    outer-circuit shape of config 4 (135 / 20 / 16 polynomials, 2^12 -> 2^15, cap height 4) and at config 3's
    (2^6 -> 2^9).  NOT a full plonky2 prove (no witness generation, quotient evaluation or FRI)."""
    res = {}
    for name, log_n in (("config4_outer_d12", 12), ("config3_d6", 6)):
        n = 1 << log_n
        shapes = [(135, True), (20, True), (16, False)]
        rng = np.random.default_rng(5)
        bufs = []
        for w, is_values in shapes:
            host = rng.integers(0, pkg.GOLDILOCKS_FIELD_ORDER, size=(w, n), dtype=np.uint64)
            bufs.append((torch.from_numpy(host.view(np.int64)).cuda(), host, w, is_values))
        cap = torch.zeros(16 * 4, dtype=torch.int64, device="cuda")
        dig = torch.zeros(((n << 3) * 2) * 4, dtype=torch.int64, device="cuda")

        def one_prove():
            for d_polys, _, w, is_values in bufs:
                pkg._native.check(lib.p2mt_polynomial_batch_commit_dev(
                    pkg._native.ptr(d_polys), int(is_values), w, log_n, 3, 4, None, pkg._native.ptr(dig),
                    pkg._native.ptr(cap)))

        for _ in range(args.warmup):
            one_prove()
        torch.cuda.synchronize()
        lib.p2mt_profile_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_prove()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        kern_ms, kern_n = C.c_float(0), C.c_int(0)
        pkg._native.check(lib.p2mt_profile_read(C.byref(kern_ms), C.byref(kern_n)))
        lib.p2mt_profile_enable(0)
        lde_bytes = sum(w * n * 8 * (1 + 8) for _, _, w, _ in bufs)  # read coeffs once, write the x8 LDE once
        lde_ms = kern_ms.value / args.steps
        entry = {"ms_per_proof_commit_phase": ms, "lde_kernels_ms": lde_ms,
                 "lde_algorithmic_GBps": lde_bytes / (lde_ms * 1e-3) / 1e9 if lde_ms > 0 else None}
        if not args.no_cpu_baseline:
            from oracle_lib import Oracle
            o = Oracle()
            t0 = time.perf_counter()
            caps = [o.polynomial_batch_commit(host, is_values, 3, 4)[2] for _, host, _, is_values in bufs]
            entry["cpu_port_ms_1core"] = (time.perf_counter() - t0) * 1e3
            pb = pkg.PolynomialBatch.from_coeffs(bufs[2][1], want_leaves=False)
            assert np.array_equal(pb.merkle_tree.cap, caps[2]), "GPU cap != oracle cap"
        res[name] = entry
    main = res["config4_outer_d12"]
    return {"metric": "ms/proof, commit phase only (3 x PolynomialBatch: 135/20/16 polys, 2^12 -> 2^15, cap 4)",
            "value": main["ms_per_proof_commit_phase"], "unit": "ms", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main["ms_per_proof_commit_phase"], "higher_is_better": False,
            "scaling": "replicas only", "vs_baseline": None, "dtype": "u64 (Goldilocks)", "data": "synthetic",
            "config": {"workload": "commit phase of mmr_plonky2_verifier_1_recursion outer prove (synthetic wire "
                                   "matrix); NOT a full plonky2 prove"},
            "roofline": {"bound": "hbm", "achieved": main["lde_algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s",
                         "frac": (main["lde_algorithmic_GBps"] or 0) / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_coset_lde (x8 coset LDE, 72 B per coefficient)"},
            "details": res}


def run_fri(args, torch, pkg, lib):
    """Secondary metric (ms/proof, opening proof only): PolynomialBatch::prove_openings -> fri_proof over the four
    committed oracles of one prove (constants_sigmas 84, wires 135, Z/partial products 20, quotient chunks 16) at the
    outer-circuit shape of config 4 (d = 12) and at config 3's (d = 6), standard_recursion_config FRI parameters
    (16-bit proof of work, 28 queries).  Oracles are committed beforehand and stay device-resident.  NOT a full plonky2
    prove (no witness generation / quotient evaluation); the polynomials are synthetic."""
    from plonky2_merkle_trees_amd import fri as F
    Nn = pkg._native
    res = {}
    for name, log_n in (("config4_outer_d12", 12), ("config3_d6", 6)):
        n, big = 1 << log_n, 1 << (log_n + 3)
        widths = [84, 135, 20, 16]
        rng = np.random.default_rng(11)
        params = F.FriParams.standard(log_n)
        hosts, dev = [], []
        oarr = (F._FriOracle * len(widths))()
        nd = sum(big >> j for j in range(log_n + 3 - 4))
        for i, w in enumerate(widths):
            host = rng.integers(0, pkg.GOLDILOCKS_FIELD_ORDER, size=(w, n), dtype=np.uint64)
            d_c = torch.from_numpy(host.view(np.int64)).cuda()
            d_l = torch.zeros(big * w, dtype=torch.int64, device="cuda")
            d_d = torch.zeros(max(nd, 1) * 4, dtype=torch.int64, device="cuda")
            d_cap = torch.zeros(64, dtype=torch.int64, device="cuda")
            Nn.check(lib.p2mt_polynomial_batch_commit_dev(Nn.ptr(d_c), 0, w, log_n, 3, 4, Nn.ptr(d_l), Nn.ptr(d_d),
                                                          Nn.ptr(d_cap)))
            hosts.append(host)
            dev.append((d_c, d_l, d_d, d_cap))
            oarr[i].coeffs = C.cast(d_c.data_ptr(), Nn.u64p)
            oarr[i].leaves = C.cast(d_l.data_ptr(), Nn.u64p)
            oarr[i].digests = C.cast(d_d.data_ptr(), Nn.u64p)
            oarr[i].n_polys = w
        torch.cuda.synchronize()
        zeta = rng.integers(0, pkg.GOLDILOCKS_FIELD_ORDER, size=2, dtype=np.uint64)
        all_polys = np.array([(oi, pi) for oi, w in enumerate(widths) for pi in range(w)], np.uint32)
        nxt = np.array([(2, 0), (2, 1)], np.uint32)
        barr = (F._FriBatch * 2)()
        for b, (pt, pl) in enumerate(((zeta, all_polys), (zeta[::-1].copy(), nxt))):
            barr[b].point[0], barr[b].point[1] = int(pt[0]), int(pt[1])
            barr[b].polys = pl.ctypes.data_as(C.POINTER(C.c_uint32))
            barr[b].n_polys = pl.shape[0]
        total = F.fri_proof_len(params, widths)
        d_proof = torch.zeros(total, dtype=torch.int64, device="cuda")
        ch = F.Challenger()
        ch.observe_elements(np.arange(1, 20, dtype=np.uint64))
        st0 = ch.state()
        d_open = torch.zeros(2 * (sum(widths) + 2), dtype=torch.int64, device="cuda")

        def openings():  # OpeningSet: every polynomial at zeta, the Zs at g*zeta
            Nn.check(lib.p2mt_fri_openings_dev(C.addressof(oarr), len(widths), C.addressof(barr), 2, log_n,
                                               Nn.ptr(d_open)))

        def one():
            ch.set_state(st0)
            openings()
            Nn.check(lib.p2mt_fri_prove_openings_dev(C.addressof(oarr), len(widths), C.addressof(barr), 2,
                                                     C.addressof(params), ch._h, Nn.ptr(d_proof)))

        for _ in range(args.warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        entry = {"ms_per_opening_proof": ms, "proof_words": int(total), "pow_witness": int(d_proof[-1].item())}
        if not args.no_cpu_baseline and log_n <= 12:
            from oracle_lib import Oracle
            o = Oracle()
            op = o.fri_params_standard(log_n)
            ooracles = [(h, d_l.cpu().numpy().view(np.uint64).reshape(big, w), d_d.cpu().numpy().view(np.uint64).reshape(-1, 4))
                        for h, (_, d_l, d_d, _), w in zip(hosts, dev, widths)]
            och = o.challenger()
            och.observe(np.arange(1, 20, dtype=np.uint64))
            batches = [(zeta, all_polys), (zeta[::-1].copy(), nxt)]
            t0 = time.perf_counter()
            want = o.fri_prove(ooracles, batches, op, och)
            entry["cpu_port_ms_1core"] = (time.perf_counter() - t0) * 1e3
            got = d_proof.cpu().numpy().view(np.uint64)
            assert np.array_equal(got, want), "GPU FRI proof != oracle proof"
        res[name] = entry
    main = res["config4_outer_d12"]
    return {"metric": "ms/proof, opening proof only (openings + FRI over 84/135/20/16 polys, 2^12 -> 2^15, 28 queries, PoW 16)",
            "value": main["ms_per_opening_proof"], "unit": "ms", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": main["ms_per_opening_proof"], "higher_is_better": False,
            "scaling": "replicas only", "vs_baseline": None, "dtype": "u64 (Goldilocks + quadratic extension)",
            "data": "synthetic",
            "config": {"workload": "PolynomialBatch::prove_openings of mmr_plonky2_verifier_1_recursion outer prove "
                                   "(synthetic polynomials); NOT a full plonky2 prove"},
            "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                         "note": "a chain of ~40 dependent small launches; see profiles/ for the per-kernel split"},
            "details": res}


def run_prove(args, torch, pkg, lib, cpu_seconds=10.0):
    """BASELINE.json's second metric: ms/proof of mmr_plonky2_verifier -- circuit_data.prove(pw)
    (/root/reference/src/mmr/mmr_plonky2_verifier.rs:148) for one leaf of a 2^20-leaf MMR (config 3: 20 path elements,
    1 peak -> 64-row circuit under standard_recursion_config).  One step = one complete prove from a PartialWitness held
    on the host to the proof words back on the host: witness fill, 3 commitments, permutation argument, quotient
    polynomials, openings, FRI (16-bit proof of work, 28 queries).  The MMR is built on the GPU and the membership proof
    comes from the device-resident MMR (config 2)."""
    Nn = pkg._native
    leaves = splitmix_leaves(1 << 20, 0x5EED0000 + 3)
    mmr = pkg.MMR.from_leaves(leaves)
    root = mmr.bagging_the_peaks()
    idx = 777777
    pr = mmr.get_proof_normal_index(idx)
    assert pr.verify(int(leaves[idx]), root)
    t0 = time.perf_counter()
    cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(len(pr.siblings), len(pr.peaks))
    build_ms = (time.perf_counter() - t0) * 1e3
    case = (int(leaves[idx]), pr.siblings, pr.lefts, pr.peaks, root)
    from circuit_cases import assign
    pw = pkg.PartialWitness()
    assign(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs, case, pw.set_target)
    proof = np.zeros(cd.info.proof_len, np.uint64)

    def one():
        Nn.check(lib.p2mt_circuit_prove(cd._h, pw._h, Nn.ptr(proof), proof.size))

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    out = {"metric": "ms/proof mmr_plonky2_verifier (prove, one leaf of a 2^20-leaf MMR)", "value": ms, "unit": "ms",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False,
           "scaling": "replicas only", "vs_baseline": None, "dtype": "u64 (Goldilocks + quadratic extension)",
           "data": "synthetic",
           "config": {"workload": "mmr_plonky2_verifier circuit_data.prove(pw), standard_recursion_config, 20 path "
                                  "elements + 1 peak, degree 2^%d, host PartialWitness -> host proof words" % cd.degree_bits,
                      "proof_words": int(cd.info.proof_len), "circuit_build_ms": build_ms,
                      "gate_rows": {k: int(v) for k, v in zip(("noop", "constant", "public_input", "arithmetic", "poseidon"),
                                                              cd.info.gate_counts)}},
           "public_inputs": [int(x) for x in proof[-4:]]}
    # Roofline of the chain's unit of work: one wave-permutation (k_challenger & co: ~180 of them are chained per proof).
    # Measured live with HIP events on the library stream: a transcript absorbing 8192 words = 1024 permutations in ONE launch.
    # Bound: VALU issue of a single wavefront -- 30 rounds x (54 v_mad_u64_u32 at 8 cycles + 59 other VALU at 4 cycles) =
    # 20 040 cycles at the 2.4 GHz peak engine clock (instruction mix from the ISA).
    from plonky2_merkle_trees_amd import fri as F
    ch = F.Challenger()
    d_words = torch.arange(1, 8193, dtype=torch.int64, device="cuda")
    Nn.check(lib.p2mt_challenger_observe_dev(ch._h, Nn.ptr(d_words), 8192))
    torch.cuda.synchronize()
    Nn.check(lib.p2mt_timer_start())
    Nn.check(lib.p2mt_challenger_observe_dev(ch._h, Nn.ptr(d_words), 8192))
    t_ms = C.c_float(0)
    Nn.check(lib.p2mt_timer_stop(C.byref(t_ms)))
    perm_us = t_ms.value * 1e3 / 1024
    peak_perms = 2.4e9 / (30 * (54 * 8 + 59 * 4))
    out["roofline"] = {"bound": "valu-issue (one wavefront; the chain is latency-bound, no HBM or MFMA roof applies)",
                       "achieved": 1e6 / perm_us, "peak": peak_perms, "unit": "wave-permutations/s per wavefront",
                       "frac": (1e6 / perm_us) / peak_perms, "traffic": None, "kernel": "k_challenger (permute_wave)",
                       "us_per_permutation": perm_us, "permutations_timed": 1024,
                       "note": "a 64-row proof chains ~180 wave-permutations (transcript 106, witness 21, leaf sponges 22, "
                               "Merkle levels ~20, FRI ~10) plus ~30 small launches; per-kernel split in profiles/"}
    acc, reason = C.c_int(0), C.c_int(0)
    Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
    assert acc.value == 1, "the product's verifier rejects the product's proof (reason %d)" % reason.value
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
    out["verify_ms"] = (time.perf_counter() - t0) * 1e3 / args.steps
    if args.threads > 1:
        # throughput: one prover per host thread (own stream, circuit handle, witness), in a separate process so that it can
        # run with blocking synchronisation (a device flag that must precede the HIP context; it frees the host cores the
        # spinning waits burn and lets 32 provers share the box's 16 cores) while the latency leg above keeps spinning
        import subprocess
        env = dict(os.environ, GPU_MAX_HW_QUEUES=str(min(args.threads, 32)))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prove_threads_probe.py"), "4", str(args.threads), "3"],
                           capture_output=True, text=True, env=env, timeout=300)
        try:
            out["throughput"] = json.loads(r.stdout.strip().splitlines()[-1])
            out["throughput"]["note"] = ("independent provers on one GPU, one per host thread and stream; `value` above stays "
                                         "the single-proof latency")
        except Exception:
            out["throughput"] = {"error": (r.stdout + r.stderr)[-400:]}
    if not args.no_cpu_baseline:
        from oracle_lib import Oracle
        from oracle import circuit as OC
        o = Oracle()
        ocd, oleaf, oproof_ts, opeak_ts = OC.verify_mmr_proof_circuit(o, len(pr.siblings), len(pr.peaks))
        opw = {}
        assign(oleaf, oproof_ts, opeak_ts, ocd.public_inputs, case, opw.__setitem__)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < cpu_seconds and reps < 40):
            want = ocd.prove(opw)
            reps += 1
        cpu_ms = (time.perf_counter() - t0) * 1e3 / reps
        assert np.array_equal(proof, want), "GPU proof != oracle proof"
        assert ocd.verify(proof) == (True, 0), "oracle verifier rejects the GPU proof"
        out["cpu_baseline"] = {"value": cpu_ms, "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "oracle/circuit.py + oracle/*.c prove of the same circuit and witness, %d proofs "
                                         "(C restatement, 1 thread; ~75 %% of it is the 2^16-hash proof-of-work grind)" % reps,
                               "cpu_over_gpu": cpu_ms / ms}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-leaves", type=int, default=24, help="leaves per GPU = 2^this")
    ap.add_argument("--variant", default=None, help="mds,partial (e.g. 2,0) Poseidon kernel variant")
    ap.add_argument("--workload", default="mmr", choices=["mmr", "commit", "fri", "prove"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="mmr workload: skip the secondary ms/proof leg")
    ap.add_argument("--threads", type=int, default=32, help="--workload prove: concurrent provers for the throughput leg (1 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --single-device exist only to exercise the N>1 code "
                         "path on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="test only: every rank uses GPU 0")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    pkg = ge.load_package()
    pkg.init(dev_index)
    if args.variant:
        pkg.set_variant(*[int(x) for x in args.variant.split(",")])
    lib = pkg.lib()
    if args.workload == "commit":
        out = run_commit(args, torch, pkg, lib) if rank == 0 else None
    elif args.workload == "fri":
        out = run_fri(args, torch, pkg, lib) if rank == 0 else None
    elif args.workload == "prove":
        out = run_prove(args, torch, pkg, lib) if rank == 0 else None
    else:
        out = run_mmr(args, torch, pkg, lib, rank, world, local_rank, dist)
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
