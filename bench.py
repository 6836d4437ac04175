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
  python bench.py --workload recursion                    (config 4: ms/proof mmr_plonky2_verifier_1_recursion, inner + outer)
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

import __graft_entry__ as ge  # noqa: E402

ALGO_BYTES_PER_HASH = 72.0   # 8 B leaf in + 2 x 32 B nodes out per two_to_one, N large
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
# Counter-derived facts about the dominant kernel cannot be read live from inside the process (PMC passes are separate
# rocprofv3 runs): they come from the committed summary of those passes, keyed by kernel configuration, and are reported only
# for the configuration they were measured on.  profiles/README.md says how the file is produced.
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_summary.json")


def pmc_summary(key):
    try:
        with open(PMC_SUMMARY) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def splitmix_leaves(n, seed):
    return ge.load_package().synthetic.splitmix_leaves(n, seed)


def _oracle():
    """tests/oracle_lib.py is the ctypes binding of oracle/: imported by the cpu_baseline / parity legs only."""
    tests_dir = os.path.join(ROOT, "tests")
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    import oracle_lib
    return oracle_lib


def cpu_baseline_mmr(host_leaves, gpu_elements_sha256, gpu_root, sample_log=20, fast_log=22):
    """CPU legs on the bench input itself (rank 0's leaves), bounded to ~30 s of CPU work:
      B1   oracle/mmr.c `for leaf { add_leaf }` on the spec-form port, 1 thread, first 2^sample_log leaves (faithful: the
           reference is single-threaded; this is `value`), extrapolation to the full size stated;
      B1'  port_fast: the same loop on the tuned scalar port (oracle/poseidon_fast.c, -O3 -march=native built here), 1 thread,
           first 2^fast_log leaves;
      B2   all_cores: level-order OpenMP build of ALL leaves with the tuned port -- also the full-size parity check: its root
           and the SHA-256 of its node array must equal the GPU's (and the spec-form port's all-core build must agree)."""
    import hashlib
    ol = _oracle()
    o = ol.Oracle()
    n = host_leaves.size
    sample_log, fast_log = min(sample_log, n.bit_length() - 1), min(fast_log, n.bit_length() - 1)
    t0 = time.perf_counter()
    m = o.mmr(host_leaves[:1 << sample_log])
    dt = time.perf_counter() - t0
    hashes = (1 << sample_log) - 1
    b1_root = m.bagging_the_peaks()
    fast = ol.build_fast_port_native()
    t0 = time.perf_counter()
    el_fast = o.fast_mmr_add_leaf_loop(host_leaves[:1 << fast_log], lib=fast)
    dt_fast = time.perf_counter() - t0
    fast_root = el_fast[-1].copy()
    del el_fast
    # thread count: the host may expose far more hardware threads than its CPU quota lets run (observed: 256 visible, ~16
    # cores' worth of throughput), so it is calibrated on a small build
    el = np.empty((2 * n - 1, 4), np.uint64)
    cal = host_leaves[:1 << 17]
    best_t, best_rate = 1, 0.0
    for t in sorted({t for t in (8, 16, 32, 64, 128, os.cpu_count() or 1) if t <= (os.cpu_count() or 1)}):
        t0 = time.perf_counter()
        o.fast_mmr_build_pow2(cal, t, el[:2 * cal.size - 1], lib=fast)
        r = cal.size / (time.perf_counter() - t0)
        if r > best_rate:
            best_t, best_rate = t, r
    t0 = time.perf_counter()
    el, threads = o.fast_mmr_build_pow2(host_leaves, best_t, el, lib=fast)
    dt2 = time.perf_counter() - t0
    cpu_sha = hashlib.sha256(el.tobytes()).hexdigest()
    cpu_root = el[-1].copy()
    t0 = time.perf_counter()
    el, _ = o.mmr_build_pow2_parallel(host_leaves, best_t, el)  # the pinned spec-form restatement, same array
    dt_spec = time.perf_counter() - t0
    spec_sha = hashlib.sha256(el.tobytes()).hexdigest()
    # B1'' / B2'': the AVX-512 port (eight permutations per zmm lane set, oracle/poseidon_avx512.c) where the host has AVX-512.
    # Level order (the add_leaf loop is one hash at a time and cannot feed eight lanes); `el` is already resident (no page faults).
    simd, simd_1, simd_all = ol.build_avx512_port_native(), None, None
    if simd is not None:
        # one core at the FULL headline size (~8 s at 2.2 M hashes/s), then all cores; both node arrays must equal the spec form's
        t0 = time.perf_counter()
        el, _ = ol.avx512_mmr_build_pow2_into(simd, host_leaves, el, 1)
        dt_s1 = time.perf_counter() - t0
        assert hashlib.sha256(el.tobytes()).hexdigest() == spec_sha, "AVX-512 port's 1-thread 2^24 build != the spec-form port's"
        t0 = time.perf_counter()
        el, _ = ol.avx512_mmr_build_pow2_into(simd, host_leaves, el, best_t)
        dt_sa = time.perf_counter() - t0
        assert hashlib.sha256(el.tobytes()).hexdigest() == spec_sha, "AVX-512 port's 2^24 build != the spec-form port's"
        simd_1 = {"value": (n - 1) / dt_s1, "cores": 1, "seconds": dt_s1,
                  "what": "B1'': level-order build of ALL 2^%d leaves (the headline size) on the AVX-512 port (oracle/poseidon_avx512.c, "
                          "eight hashes per permutation call, -O3 -march=native on this host), 1 thread; node array SHA-256 == the "
                          "spec-form port's" % (n.bit_length() - 1)}
        simd_all = {"value": (n - 1) / dt_sa, "cores": threads, "seconds": dt_sa,
                    "what": "B2'': the same build of ALL 2^%d leaves, OpenMP; node array SHA-256 == the spec-form port's"
                            % (n.bit_length() - 1)}
    parity = {"size_log2": n.bit_length() - 1, "root_equal": bool(np.array_equal(cpu_root, gpu_root)),
              "elements_sha256_gpu": gpu_elements_sha256, "elements_sha256_oracle": spec_sha,
              "elements_sha256_equal": spec_sha == gpu_elements_sha256 and cpu_sha == gpu_elements_sha256,
              "oracle": "oracle_mmr_build_pow2_parallel (spec-form port, %d threads, %.1f s) and the tuned port's build" % (threads, dt_spec)}
    scalar_1 = {"value": ((1 << fast_log) - 1) / dt_fast, "cores": 1, "seconds": dt_fast,
                "what": "B1': the same add_leaf loop on the tuned scalar port (oracle/poseidon_fast.c: sparse partial "
                        "rounds, lazy reduction, -O3 -march=native on this host), first 2^%d leaves" % fast_log}
    scalar_all = {"value": (n - 1) / dt2, "cores": threads, "seconds": dt2,
                  "what": "B2: level-order OpenMP build of ALL 2^%d leaves with the tuned scalar port (generous, not the "
                          "reference's algorithm)" % (n.bit_length() - 1)}
    # port_fast / all_cores = the fastest CPU legs this host can run (the AVX-512 port where available); the scalar legs stay
    # beside them
    use_simd = simd_1 is not None and simd_1["value"] > scalar_1["value"]
    return {"value": hashes / dt, "unit": "Poseidon hashes/s", "cores": 1, "kind": "port",
            "sample": "oracle/mmr.c add_leaf loop (B1, faithful: the reference is single-threaded), first 2^%d leaves of the "
                      "bench input, %.1f s; at this rate the full 2^%d-leaf build would take %.0f s"
                      % (sample_log, dt, n.bit_length() - 1, (n - 1) / (hashes / dt)),
            "port_fast": simd_1 if use_simd else scalar_1,
            "all_cores": simd_all if use_simd and simd_all["value"] > scalar_all["value"] else scalar_all,
            "port_fast_scalar": scalar_1, "all_cores_scalar": scalar_all,
            "full_size_parity": parity,
            "_roots": (b1_root, fast_root, sample_log, fast_log)}


def transfer_times(torch, host_leaves, mmr):
    """H2D of the leaves and D2H of the node array, reported separately: the timed region starts with the leaves
    resident in HBM and ends with the 32-byte root on the host; these are what a caller holding host buffers pays on top."""
    n = host_leaves.size
    pinned = torch.from_numpy(host_leaves.view(np.int64)).pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d = pinned.cuda(non_blocking=False)
    torch.cuda.synchronize()
    h2d_pinned = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    d2 = torch.from_numpy(host_leaves.view(np.int64)).cuda()
    torch.cuda.synchronize()
    h2d_pageable = (time.perf_counter() - t0) * 1e3
    del d, d2
    count = len(mmr)
    out = np.empty((count, 4), np.uint64)
    out.fill(0)  # fault the pages in before timing
    t0 = time.perf_counter()
    got = mmr.copy_elements(0, count)
    d2h = (time.perf_counter() - t0) * 1e3
    assert got.shape == out.shape
    # the same bytes into page-locked memory of the library, and the extend that streams what it appends while it hashes
    # (p2mt_mmr_extend_dev_to_host: chunks of 2^20 leaves, copies on the copy engines): a whole build WITH the node array on the host
    import hashlib
    pkg = ge.load_package()
    Nn, lib = pkg._native, pkg._native.lib()
    pin = pkg.mmr.PinnedBuffer(4 * count)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mmr.copy_elements_async(0, count, pin)
    Nn.check(lib.p2mt_sync())
    d2h_pinned = (time.perf_counter() - t0) * 1e3
    want = hashlib.sha256(got.tobytes()).hexdigest()
    assert hashlib.sha256(pin.array.tobytes()).hexdigest() == want
    d3 = torch.from_numpy(host_leaves.view(np.int64)).cuda()
    m2 = pkg.mmr.MMR()
    m2.reserve(n)
    over = []
    for _ in range(3):
        pin.array[:8] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m2.reset()
        m2.extend_dev_to_host(d3, n, pin, chunk_log=min(20, n.bit_length() - 1))
        Nn.check(lib.p2mt_sync())
        over.append((time.perf_counter() - t0) * 1e3)
    assert hashlib.sha256(pin.array.tobytes()).hexdigest() == want
    del m2, d3
    pin.free()
    return {"h2d_ms_leaves_pinned": h2d_pinned, "h2d_ms_leaves_pageable": h2d_pageable, "leaves_bytes": int(n * 8),
            "d2h_ms_elements_pageable": d2h, "d2h_ms_elements_pinned": d2h_pinned,
            "build_plus_elements_on_host_overlapped_ms": float(np.median(over)), "elements_bytes": int(count * 32),
            "note": "not part of `value`: leaves are resident in HBM when the timed region starts and only the root comes back; "
                    "D2H includes allocating and faulting in the destination (what MMR.elements costs a caller)"}


def run_merkle_tree(torch, pkg, lib, log_n=24):
    """simple_merkle_tree::MerkleTree::build (/root/reference/src/simple_merkle_tree/simple_merkle_tree.rs:28-51) at the headline size,
    device-resident leaves -> device-resident level-major tree + root (p2mt_merkle_build_pow2_dev): stage 1 as per-lane subtrees like the
    MMR's, level kernels above.  The same 2^24 - 1 hashes as the MMR build of the same leaves, in the reference's other layout."""
    n = 1 << log_n
    leaves = pkg.synthetic.bench_leaves(log_n, 0)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).cuda()
    d_levels = torch.empty(4 * (2 * n - 2), dtype=torch.int64, device="cuda")
    d_root = torch.empty(4, dtype=torch.int64, device="cuda")

    def build():
        pkg._native.check(lib.p2mt_merkle_build_pow2_dev(d_leaves.data_ptr(), n, d_levels.data_ptr(), d_root.data_ptr()))
        pkg._native.check(lib.p2mt_sync())
        return d_root.cpu().numpy().view(np.uint64)

    for _ in range(3):
        root = build()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        root = build()
        ts.append((time.perf_counter() - t0) * 1e3)
    # the tree's top two levels hash the same pairs as the MMR of the same leaves: its root is the MMR's single peak
    m = pkg.MMR()
    m.reserve(n)
    m.extend_dev(d_leaves, n)
    peaks = m.get_peaks()
    return {"workload": "simple_merkle_tree MerkleTree::build, 2^%d leaves, device-resident leaves and tree" % log_n,
            "build_ms": float(np.median(ts)), "build_ms_min": float(np.min(ts)), "hashes_per_s": (n - 1) / (float(np.median(ts)) * 1e-3),
            "root": [int(x) for x in root], "root_equals_mmr_peak": bool(np.array_equal(root, np.asarray(peaks).reshape(-1, 4)[0]))}


def run_config2(torch, pkg, lib, cpu_baseline=True):
    """BASELINE.md B3 / BASELINE.json config 2: mmr::merkle_mountain_ranges build + get_proof, 2^20 leaves, 1 GPU -- build from
    device-resident leaves, get_proof + MMR_proof::verify for the four fixed leaves {0, 1, 777 777, 2^20 - 1}
    (/root/reference/src/mmr/merkle_mountain_ranges.rs:209-252), the same through the oracle on one host core, and the batched
    proof service (p2mt_mmr_proof_batch / _verify_batch) on 2^20 proofs."""
    n = 1 << 20
    leaves = splitmix_leaves(n, 0x5EED0000 + 2)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).cuda()
    m = pkg.MMR()
    m.reserve(n)
    idxs = [0, 1, 777777, n - 1]

    def build():
        m.reset()
        m.extend_dev(d_leaves, n)
        return m.bagging_the_peaks()

    for _ in range(3):
        root = build()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        root = build()
        ts.append((time.perf_counter() - t0) * 1e3)
    build_ms = float(np.median(ts))
    ts, proofs = [], None
    for _ in range(10):
        t0 = time.perf_counter()
        proofs = [m.get_proof_normal_index(i) for i in idxs]
        oks = [pr.verify(int(leaves[i]), root) for pr, i in zip(proofs, idxs)]
        ts.append((time.perf_counter() - t0) * 1e3)
    assert all(oks)
    four_ms = float(np.median(ts))
    # batched proof service: every leaf's proof in one call, verified in one call (host buffers in and out)
    sel = np.arange(0, n, 4)  # every fourth leaf: 2^18 proofs (168 MB of siblings through host buffers)
    all_idx = np.array([2 * int(i) - bin(int(i)).count("1") for i in sel], np.uint64)
    t0 = time.perf_counter()
    sib, lefts, ns = m.get_proof_batch(all_idx, max_siblings=20)
    t1 = time.perf_counter()
    status = pkg.verify_proof_batch(sib, lefts, ns, m.get_peaks(), leaves[sel], root)
    t2 = time.perf_counter()
    assert (status == 1).all() and (ns == 20).all()
    n_service = sel.size
    out = {"workload": "config 2 (BASELINE.md B3): 2^20-leaf MMR build + get_proof + verify for leaves {0, 1, 777777, 2^20-1}",
           "build_ms": build_ms, "build_hashes_per_s": (n - 1) / (build_ms * 1e-3),
           "get_proof_and_verify_4_leaves_ms": four_ms,
           "proof_service": {"proofs": int(n_service), "get_proof_batch_proofs_per_s": n_service / (t1 - t0),
                             "verify_batch_proofs_per_s": n_service / (t2 - t1),
                             "note": "p2mt_mmr_proof_batch / p2mt_mmr_proof_verify_batch, host buffers in and out (PCIe-inclusive)"},
           "root": [int(x) for x in root]}
    if cpu_baseline:
        o = _oracle().Oracle()
        t0 = time.perf_counter()
        om = o.mmr(leaves)  # the reference's add_leaf loop, spec-form port, 1 thread
        cpu_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i, pr in zip(idxs, proofs):
            opr = om.get_proof_normal_index(i)
            assert np.array_equal(opr["siblings"], pr.siblings) and np.array_equal(opr["lefts"], pr.lefts)
            assert o.mmr_proof_verify(opr["siblings"], opr["lefts"], opr["peaks"], leaves[i], om.bagging_the_peaks())
        cpu_four = time.perf_counter() - t0
        assert np.array_equal(om.bagging_the_peaks(), root), "config 2: GPU root != oracle root"
        out["cpu_baseline"] = {"kind": "port", "cores": 1, "build_ms": cpu_build * 1e3, "get_proof_and_verify_4_leaves_ms": cpu_four * 1e3,
                               "sample": "oracle/mmr.c: add_leaf loop over all 2^20 leaves, then get_proof + verify of the same four "
                                         "leaves; proofs and root equal the GPU's",
                               "gpu_over_cpu_build": cpu_build * 1e3 / build_ms}
    return out


def run_mmr(args, torch, pkg, lib, rank, world, local_rank, dist):
    # weak: every rank builds its own 2^log_leaves shard (per-GPU work fixed); strong: 2^log_leaves leaves in total, split
    # into world leaf ranges (BASELINE.json's "2^24 leaves at 1/2/4/8 GPUs"; config 5 = --scaling strong --log-leaves 26 --gpus 8
    # or, equivalently, --scaling weak --log-leaves 23 --gpus 8)
    g = world.bit_length() - 1
    if world & (world - 1):
        raise SystemExit("--gpus must be a power of two (perfect subtrees per rank)")
    local_log = args.log_leaves - g if args.scaling == "strong" else args.log_leaves
    if local_log < 1:
        raise SystemExit("--log-leaves too small for %d ranks" % world)
    n = 1 << local_log
    host_leaves = pkg.synthetic.bench_leaves(local_log, rank)
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
    step_ms, t_prev = [], t0
    for _ in range(args.steps):
        root = step()  # ends with the root read-back, i.e. synchronised: per-step wall times cost nothing extra
        t_now = time.perf_counter()
        step_ms.append((t_now - t_prev) * 1e3)
        t_prev = t_now
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
    info = pkg.stage1_info(n)
    fused_levels = info["levels"]
    # (hashes are split evenly over the stage-1 launches of a step)
    launches_per_step = max(kern_n.value, 1) / float(args.steps)
    hashes_in_launch = (n - (n >> fused_levels)) / launches_per_step
    launch_ms = kern_ms.value / max(kern_n.value, 1)
    algo_bytes = hashes_in_launch * ALGO_BYTES_PER_HASH
    achieved_gbs = algo_bytes / (launch_ms * 1e-3) / 1e9 if launch_ms > 0 else 0.0
    in_kernel_rate = hashes_in_launch / (launch_ms * 1e-3) if launch_ms > 0 else 0.0
    pmc = pmc_summary("%s@2^%d" % (info["key"], local_log)) or {}
    total_log = local_log + g
    out = {
        "metric": "Poseidon hashes/s (MMR build, 2^%d leaves%s)" % (args.log_leaves, " per GPU" if args.scaling == "weak" else " in total"),
        "value": value, "unit": "Poseidon hashes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "u64 (Goldilocks, integer VALU)", "data": "synthetic",
        "config": {"workload": "mmr::merkle_mountain_ranges build of one 2^%d-leaf MMR: 2^%d leaves per GPU x %d GPU(s) (%s "
                               "scaling), device-resident leaves" % (total_log, local_log, world, args.scaling),
                   "leaves_per_gpu": n, "total_leaves": n * world, "hashes_per_step": total_hashes,
                   "poseidon_variant": {"mds": mds.value, "partial": partial.value},
                   "stage1": info["key"],
                   "sharding": "leaf ranges per rank + all-gather of 32-byte shard roots" if world > 1 else "none",
                   "exchange": ("device-resident: p2mt_mmr_root_dev -> all_gather_into_tensor (RCCL) -> one combine launch -> "
                                "one read-back" if shard._device_exchange() else
                                ("host round trip (gloo rehearsal backend)" if dist is not None else "none (one rank)"))},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS,
                     "traffic": pmc.get("hbm_bytes_per_launch"),
                     "kernel": info["kernel"],
                     "launch_ms": launch_ms, "launches_timed": kern_n.value,
                     "algorithmic_bytes_per_launch": algo_bytes, "hashes_per_launch": hashes_in_launch,
                     "note": "Poseidon is integer-issue bound (see `valu`): ~10.1k VALU + 44 matrix-pipe instructions per 72 "
                             "algorithmic bytes, so the HBM fraction is ~3 % by construction (SURVEY.md 8d)"},
        # counter evidence for "compute-bound, the right way": share of SIMD issue cycles spent on VALU instructions and VALU
        # instructions per hash, from the committed PMC passes (null when this configuration was not profiled)
        "valu": {"simd_cycles_per_valu_instr": pmc.get("simd_cycles_per_valu_instr"),
                 "valu_instr_per_hash": pmc.get("valu_instr_per_hash"), "effective_clock_ghz": pmc.get("effective_clock_ghz"),
                 "ubench_simd_cycles_per_valu_instr": pmc.get("ubench_simd_cycles_per_valu_instr"),
                 # the dense MDS layers run on the matrix pipe (one v_mfma_i32_32x32x32_i8 per 8-bit limb): instructions per hash,
                 # pipe cycles per instruction, share of them during which the VALU co-executes, pipe-busy share of the launch
                 "mfma": {"instr_per_hash": pmc.get("mfma_instr_per_hash"),
                          "busy_cycles_per_instr": pmc.get("mfma_busy_cycles_per_instr"),
                          "valu_coexec_frac": pmc.get("mfma_valu_coexec_frac"),
                          "busy_frac_of_simd_cycles": pmc.get("mfma_busy_frac_of_simd_cycles")},
                 "source": pmc.get("source"), "in_kernel_hashes_per_s": in_kernel_rate,
                 "note": "simd_cycles_per_valu_instr = (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) / SQ_INSTS_VALU, both counters from "
                         "the same dispatches: how many SIMD cycles the launch spends per VALU instruction it issues.  The "
                         "micro-benchmarked cost of this kernel's instruction class on the same chip (tools/ubench_valu.hip: "
                         "v_mad_u64_u32 and the other non-trivial VALU ops issue once per ~4.0-4.6 SIMD cycles, add/xor/mov once "
                         "per ~2.3) is beside it: the kernel sits on the VALU issue cadence, so instructions per hash is the only "
                         "lever.  (SQ_ACTIVE_INST_VALU equals SQ_INSTS_VALU on gfx950 -- an instruction count, not a busy counter -- "
                         "which is why round 2's clamped `valu_busy` is gone.)"},
        "device_ms_per_step": region_ms.value / args.steps,
        # SURVEY.md 8d asks for the median of the timed runs: `value` / `ms_per_step` stay total time / steps (the contract of this
        # file); the per-step wall times (each step ends with the root read-back) give the median and the spread beside them
        "ms_per_step_median": float(np.median(step_ms)), "ms_per_step_min": float(np.min(step_ms)),
        "ms_per_step_max": float(np.max(step_ms)),
        "value_at_median": total_hashes / (float(np.median(step_ms)) * 1e-3),
        "root": [int(x) for x in root],
    }
    if world == 1:
        out["transfers"] = transfer_times(torch, host_leaves, shard.local)
    if world == 1 and not args.no_cpu_baseline:
        import hashlib
        gpu_sha = hashlib.sha256(shard.local.elements.tobytes()).hexdigest()
        cb = cpu_baseline_mmr(host_leaves, gpu_sha, root)
        assert cb["full_size_parity"]["root_equal"] and cb["full_size_parity"]["elements_sha256_equal"], \
            "GPU MMR != oracle at the full bench size: %r" % (cb["full_size_parity"],)
        b1_root, fast_root, sample_log, fast_log = cb.pop("_roots")
        # the bounded samples are prefixes of the same input: the GPU build of each prefix must give the port's root
        for r_cpu, lg in ((b1_root, sample_log), (fast_root, fast_log)):
            sub = pkg.MMR.from_leaves(host_leaves[:1 << lg])
            assert np.array_equal(sub.bagging_the_peaks(), r_cpu), "GPU root != oracle root on the 2^%d CPU sample" % lg
        cb["gpu_over_cpu"] = value / cb["value"]
        cb["port_fast"]["gpu_over_cpu"] = value / cb["port_fast"]["value"]
        cb["all_cores"]["gpu_over_cpu"] = value / cb["all_cores"]["value"]
        out["cpu_baseline"] = cb
    if world == 1 and not args.no_prove:
        try:
            out["config2_mmr_2pow20"] = run_config2(torch, pkg, lib, cpu_baseline=not args.no_cpu_baseline)
        except Exception as e:  # the headline line must survive a failure of a secondary leg
            out["config2_mmr_2pow20"] = {"error": repr(e)}
    if world == 1 and not args.no_prove:
        try:
            out["simple_merkle_tree_2pow24"] = run_merkle_tree(torch, pkg, lib, 24)
        except Exception as e:
            out["simple_merkle_tree_2pow24"] = {"error": repr(e)}
    if world == 1 and not args.no_prove:
        # the commit kernels of the prover under the same discipline as the hash kernel: commit phase ms at the
        # config-3 / config-4 shapes with B4 beside them, and the transform kernels' rooflines at the points where HBM is the bound
        import copy
        ca = copy.copy(args)
        ca.steps, ca.warmup = 20, 3
        try:
            cr = run_commit(ca, torch, pkg, lib)
            out["commit_phase"] = {"value": cr["value"], "unit": "ms", "what": cr["metric"], "roofline": cr["roofline"],
                                   "details": cr["details"], "large_points": cr.get("large_points"),
                                   "how": "python bench.py --workload commit"}
        except Exception as e:
            out["commit_phase"] = {"error": repr(e)}
    if world == 1 and not args.no_prove:
        # BASELINE.json's second metric (ms/proof), measured after and outside the timed region above
        import copy
        pa = copy.copy(args)
        pa.steps, pa.warmup = 30, 5
        try:
            pr = run_prove(pa, torch, pkg, lib, cpu_seconds=3.0)
            out["ms_per_proof_mmr_plonky2_verifier"] = {
                "value": pr["value"], "unit": "ms", "config": pr["config"]["workload"],
                "throughput": pr.get("throughput"), "throughput_threads": pr.get("throughput_threads"),
                "verify_ms": pr.get("verify_ms"), "verify_batch_proofs_per_s": pr.get("verify_batch_proofs_per_s"),
                "us_per_wave_permutation": (pr.get("roofline") or {}).get("us_per_permutation"),
                "cpu_baseline": pr.get("cpu_baseline"),
                "how": "python bench.py --workload prove"}
        except Exception as e:  # the headline line must survive a failure of the secondary leg
            out["ms_per_proof_mmr_plonky2_verifier"] = {"error": repr(e)}
        try:
            pa.steps, pa.warmup = 10, 2
            rr = run_recursion(pa, torch, pkg, lib, cpu_baseline=not args.no_cpu_baseline)
            out["ms_per_proof_mmr_plonky2_verifier_1_recursion"] = {
                "value": rr["value"], "unit": "ms", "config": rr["config"], "verify_outer_ms": rr["verify_outer_ms"],
                "throughput": rr.get("throughput"),
                "cpu_baseline": rr.get("cpu_baseline"), "how": "python bench.py --workload recursion"}
        except Exception as e:
            out["ms_per_proof_mmr_plonky2_verifier_1_recursion"] = {"error": repr(e)}
    return out


def run_probe(cmd, env=None, timeout=300):
    """A throughput probe in its own process -> its last JSON line, or {"error": ...}: a slow, hung or crashed probe must not
    cost the caller the JSON line it has already measured."""
    import subprocess
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)
    except (subprocess.TimeoutExpired, OSError) as e:
        return {"error": "probe did not finish: %r" % (e,)}
    try:
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return {"error": (r.stdout + r.stderr)[-400:]}


def _prof_region(lib, pkg, fn, reps):
    """Run fn() reps times with the library's HIP-event profiler on (events on the stream the kernels are launched on) ->
    (wall ms per call, summed profiled-kernel ms per call, profiled launches per call)."""
    import torch
    torch.cuda.synchronize()
    lib.p2mt_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / reps
    kern_ms, kern_n = C.c_float(0), C.c_int(0)
    pkg._native.check(lib.p2mt_profile_read(C.byref(kern_ms), C.byref(kern_n)))
    lib.p2mt_profile_enable(0)
    return wall, kern_ms.value / reps, kern_n.value / float(reps)


def _rand_field_dev(torch, count, seed):
    """count uniform field elements, generated on the device (the host needs seconds for 2^27 values)"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    hi = torch.randint(0, 0xFFFFFFFF, (count,), dtype=torch.int64, device="cuda", generator=g)  # < 2^32 - 1 => value < p
    lo = torch.randint(0, 1 << 32, (count,), dtype=torch.int64, device="cuda", generator=g)
    return (hi << 32) | lo


def run_commit(args, torch, pkg, lib, large=True):
    """Secondary metric (ms/proof, commit phase only): the three PolynomialBatch commits of one prove at the
    outer-circuit shape of config 4 (135 / 20 / 16 polynomials, 2^12 -> 2^15, cap height 4) and at config 3's
    (2^6 -> 2^9), with BASELINE.md's B4 beside them (the same three commits by the C restatement on all host cores, caps
    compared).  NOT a full plonky2 prove (no witness generation, quotient evaluation or FRI).
    `large`: the two points where HBM, not launch latency, is what a transform kernel is up against: the x8 coset
    LDE of the batched prover's 32 x 135 polynomials (1.13 GB out) and 128 transforms of 2^20 points (1.07 GB), each with the
    roofline of its kernel: algorithmic bytes per launch (72 B per coefficient for the LDE, 16 B per point for a
    transform pass) / the launch's HIP-event duration."""
    N = pkg._native
    res = {}
    for name, log_n in (("config4_outer_d12", 12), ("config3_d6", 6)):
        n = 1 << log_n
        shapes = [(135, True), (20, True), (16, False)]
        rng = np.random.default_rng(5)
        bufs = []
        for w, is_values in shapes:
            host = rng.integers(0, pkg.GOLDILOCKS_FIELD_ORDER, size=(w, n), dtype=np.uint64)
            bufs.append((torch.from_numpy(host.view(np.int64)).cuda(), host, w, is_values))
        caps = [torch.zeros(16 * 4, dtype=torch.int64, device="cuda") for _ in bufs]
        dig = torch.zeros(((n << 3) * 2) * 4, dtype=torch.int64, device="cuda")

        def one_prove():
            for (d_polys, _, w, is_values), cap in zip(bufs, caps):
                N.check(lib.p2mt_polynomial_batch_commit_dev(N.ptr(d_polys), int(is_values), w, log_n, 3, 4, None, N.ptr(dig),
                                                             N.ptr(cap)))

        for _ in range(args.warmup):
            one_prove()
        ms, lde_ms, lde_n = _prof_region(lib, pkg, one_prove, args.steps)
        lde_bytes = sum(w * n * 8 * (1 + 8) for _, _, w, _ in bufs)  # read coeffs once, write the x8 LDE once
        entry = {"ms_per_proof_commit_phase": ms, "lde_kernels_ms": lde_ms, "lde_launches": lde_n,
                 "lde_algorithmic_bytes": lde_bytes,
                 "lde_algorithmic_GBps": lde_bytes / (lde_ms * 1e-3) / 1e9 if lde_ms > 0 else None}
        if not args.no_cpu_baseline:
            o = _oracle().Oracle()
            t0 = time.perf_counter()
            cpu = [o.polynomial_batch_commit_parallel(host, is_values, 3, 4) for _, host, _, is_values in bufs]
            cpu_ms = (time.perf_counter() - t0) * 1e3
            for (cap_cpu, _), cap in zip(cpu, caps):
                assert np.array_equal(cap.cpu().numpy().view(np.uint64).reshape(16, 4), cap_cpu), "GPU cap != oracle cap"
            entry["cpu_baseline"] = {"value": cpu_ms, "unit": "ms", "cores": cpu[0][1], "kind": "port",
                                     "sample": "B4: oracle_polynomial_batch_commit_parallel (oracle/fft.c: textbook transforms + the "
                                               "tuned scalar Poseidon port, OpenMP over polynomials / leaves / tree levels), the same "
                                               "three commits once; all three caps equal the GPU's",
                                     "cpu_over_gpu": cpu_ms / ms}
        res[name] = entry
    main = res["config4_outer_d12"]
    out = {"metric": "ms/proof, commit phase only (3 x PolynomialBatch: 135/20/16 polys, 2^12 -> 2^15, cap 4)",
           "value": main["ms_per_proof_commit_phase"], "unit": "ms", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": main["ms_per_proof_commit_phase"], "higher_is_better": False,
           "scaling": "replicas only", "vs_baseline": None, "dtype": "u64 (Goldilocks)", "data": "synthetic",
           "config": {"workload": "commit phase of mmr_plonky2_verifier_1_recursion outer prove (synthetic wire "
                                  "matrix); NOT a full plonky2 prove"},
           "roofline": {"bound": "hbm", "achieved": main["lde_algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s",
                        "frac": (main["lde_algorithmic_GBps"] or 0) / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "k_coset_lde12_v2 (x8 coset LDE, 72 B per coefficient), the three launches of ONE prove: 171 "
                                  "polynomials = 50 MB written, 1 368 workgroups -- a launch this small is mostly ramp-up and "
                                  "tail; `large_points` is the same kernel with the chip full"},
           "details": res}
    if large:
        lp = {}
        # (1) the batched prover's LDE: 32 proofs x 135 wire polynomials, 2^12 -> 2^15
        w, log_n = 32 * 135, 12
        d_in = _rand_field_dev(torch, w << log_n, 1)
        d_out = torch.empty(w << (log_n + 3), dtype=torch.int64, device="cuda")

        def lde():
            N.check(lib.p2mt_coset_lde_leaf_order_dev(N.ptr(d_in), log_n, 3, 7, w, N.ptr(d_out)))

        for _ in range(3):
            lde()
        wall, k_ms, k_n = _prof_region(lib, pkg, lde, 10)
        algo = (w << log_n) * 72
        pm = pmc_summary("k_coset_lde12_v2@4320x2^12") or {}
        lp["coset_lde_x8_4320_polys_2pow12"] = {
            "what": "p2mt_coset_lde_leaf_order_dev: 32 x 135 polynomials of 2^12 coefficients -> x8 (141.6 MB in, 1 132 MB out), "
                    "one launch, one pass over HBM",
            "wall_ms": wall,
            "roofline": {"bound": "hbm", "achieved": algo / (k_ms / max(k_n, 1) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / (k_ms / max(k_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pm.get("hbm_bytes_per_launch"),
                         "kernel": "k_coset_lde12_v2", "launch_ms": k_ms / max(k_n, 1), "launches_timed": int(k_n * 10),
                         "algorithmic_bytes_per_launch": algo,
                         "valu_instr_per_output_point": pm.get("valu_instr_per_point"),
                         "simd_cycles_per_valu_instr": pm.get("simd_cycles_per_valu_instr"),
                         "note": "issue-bound, not HBM-bound: ~137 VALU instructions per output point (12 butterfly stages + 3 field "
                                 "multiplications on a 32-bit integer VALU) at one instruction per ~5 SIMD cycles; a plain copy of "
                                 "the same bytes takes 0.25 ms (tools/ubench_granule.hip)"}}
        del d_in, d_out
        # (2) 128 transforms of 2^20 points, natural order in and out: two passes over HBM
        w, log_n = 128, 20
        d = _rand_field_dev(torch, w << log_n, 2)

        def ntt():
            N.check(lib.p2mt_ntt_batch_dev(N.ptr(d), log_n, w, 0))

        for _ in range(2):
            ntt()
        wall, k_ms, k_n = _prof_region(lib, pkg, ntt, 10)
        algo = (w << log_n) * 16
        pm = pmc_summary("k_ntt20_pass@128x2^20") or {}
        per_pass_ms = k_ms / max(k_n, 1)
        lp["ntt_128_polys_2pow20"] = {
            "what": "p2mt_ntt_batch_dev (fft_with_options, natural order in and out): 128 polynomials of 2^20 points (1 074 MB), "
                    "two launches = two passes over HBM (LDS holds 2^14 points: two passes is the minimum at this size)",
            "wall_ms": wall, "whole_transform_algorithmic_GBps": algo / (wall * 1e-3) / 1e9,
            "whole_transform_frac_of_hbm_peak": algo / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "roofline": {"bound": "hbm", "achieved": algo / (per_pass_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": algo / (per_pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pm.get("hbm_bytes_per_launch"),
                         "kernel": "k_ntt20_pass (one 1024-point pass: reads and writes every point once)",
                         "launch_ms": per_pass_ms, "launches_timed": int(k_n * 10), "algorithmic_bytes_per_launch": algo,
                         "valu_instr_per_point": pm.get("valu_instr_per_point"),
                         "note": "per PASS kernel (16 B per point per launch); the whole transform moves 32 B per point, so its "
                                 "fraction of peak is `whole_transform_frac_of_hbm_peak`.  A plain copy with the passes' access "
                                 "patterns runs at 5.2 TB/s (tools/ubench_granule.hip): the passes are bound by VALU issue "
                                 "(~105-120 instructions per point), not by the 128-byte granules"}}
        del d
        out["large_points"] = lp
    return out


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
            o = _oracle().Oracle()
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
    assign = pkg.synthetic.assign_mmr_proof
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
    # Bound: VALU issue of a single wavefront -- v_mad_u64_u32 at 8 cycles, other VALU at 4 cycles, instruction mix from the ISA
    # (DESIGN.md 4.3): 16 400 cycles at the 2.4 GHz peak engine clock (20 040 before the partial rounds were batched in threes).
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
    # 9 rounds of (54 mads, 59 other VALU) + 7 groups of three partial rounds of (121 mads, 129 other) -- the batched form of round 3
    peak_perms = 2.4e9 / (9 * (54 * 8 + 59 * 4) + 7 * (121 * 8 + 129 * 4))
    out["roofline"] = {"bound": "valu-issue (one wavefront; the chain is latency-bound, no HBM or MFMA roof applies)",
                       "achieved": 1e6 / perm_us, "peak": peak_perms, "unit": "wave-permutations/s per wavefront",
                       "frac": (1e6 / perm_us) / peak_perms, "traffic": None, "kernel": "k_challenger (permute_wave)",
                       "us_per_permutation": perm_us, "permutations_timed": 1024,
                       "note": "this is the DEVICE transcript's permutation (what the batched passes run); a single prove keeps its "
                               "transcript (~110 permutations) and its witness's PoseidonGate chain (41 rows) on a host core "
                               "(csrc/host_poseidon.hip, 0.86 us per permutation with AVX-512) and chains ~60 wave-permutations on the "
                               "device (leaf sponges 22, Merkle levels ~20, FRI ~10) plus ~40 small launches; per-kernel split in "
                               "profiles/r05_prove_timeline_d6_hostchain.txt"}
    acc, reason = C.c_int(0), C.c_int(0)
    Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
    assert acc.value == 1, "the product's verifier rejects the product's proof (reason %d)" % reason.value
    per_call = []  # a verification is ~1 ms of host wall time: one pre-empted call would move a mean of a few dozen
    for _ in range(max(args.steps, 20)):
        t0 = time.perf_counter()
        Nn.check(lib.p2mt_circuit_verify(cd._h, Nn.ptr(proof), proof.size, C.byref(acc), C.byref(reason)))
        per_call.append((time.perf_counter() - t0) * 1e3)
    out["verify_ms"] = float(np.median(per_call))
    out["verify_ms_mean"] = float(np.mean(per_call))
    # batched verify: 256 proofs per pass (transcripts and Merkle paths with the proof index in grid z, field arithmetic on host threads)
    many = np.ascontiguousarray(np.tile(proof, (256, 1)))
    accs, reasons = (C.c_int * 256)(), (C.c_int * 256)()
    Nn.check(lib.p2mt_circuit_verify_batch(cd._h, Nn.ptr(many), 256, many.shape[1], accs, reasons))
    assert all(accs), "batched verify rejects the product's proof"
    t0 = time.perf_counter()
    for _ in range(5):
        Nn.check(lib.p2mt_circuit_verify_batch(cd._h, Nn.ptr(many), 256, many.shape[1], accs, reasons))
    out["verify_batch_proofs_per_s"] = 5 * 256 / (time.perf_counter() - t0)
    if args.threads > 1:
        # throughput: one prover per host thread (own stream, circuit handle, witness), in a separate process so that it can
        # run with blocking synchronisation (a device flag that must precede the HIP context; it frees the host cores the
        # spinning waits burn and lets 32 provers share the box's 16 cores) while the latency leg above keeps spinning
        env = dict(os.environ, GPU_MAX_HW_QUEUES=str(min(args.threads, 32)))
        out["throughput_threads"] = run_probe([sys.executable, os.path.join(ROOT, "tools", "prove_threads_probe.py"), "4",
                                               str(args.threads), "3"], env=env)
        if "error" not in out["throughput_threads"]:
            out["throughput_threads"]["note"] = ("independent provers on one GPU, one per host thread and stream (bound by the "
                                                 "device's dispatch-packet rate)")
        # the batched prover (p2mt_batch_prover_*): 256 statements per pass, the proof index in a grid dimension of every launch;
        # three host threads so that one pass's latency-bound transcript overlaps the others' grind
        out["throughput"] = run_probe([sys.executable, os.path.join(ROOT, "tools", "prove_batch_probe.py"), "4", "3", "256", "3"],
                                      env=dict(os.environ))
        if "error" not in out["throughput"]:
            out["throughput"]["note"] = ("p2mt_batch_prover: proofs bit-identical to the single-proof path's; `value` above "
                                         "stays the single-proof latency")
    if not args.no_cpu_baseline:
        from oracle import circuit as OC
        o = _oracle().Oracle()
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
                                         "(Python host logic around C field work, 1 thread; ~75 %% of it is the 2^16-hash "
                                         "proof-of-work grind): a parity leg with a clock on it, NOT a CPU prover worth a ratio "
                                         "(commit_phase.details.*.cpu_baseline is the all-cores C baseline, B4)" % reps}
    return out


def run_prove_replicas(args, torch, pkg, lib, rank, world, dist):
    """--workload prove on N > 1 GPUs: the prover does not shard (one proof is a dependent chain on one device: "replicas only",
    DESIGN.md 6), so every rank runs its own batched prover (p2mt_batch_prover, 128 different statements per pass) on its own
    GPU; a step = one pass per rank; value = all ranks' proofs / the slowest rank's time (barrier on both sides)."""
    Nn = pkg._native
    B = 128
    P = pkg.GOLDILOCKS_FIELD_ORDER
    cd, leaf_t, proof_ts, peak_ts = pkg.verify_mmr_proof_circuit(20, 1)
    pws = []
    for k in range(B):
        rng = np.random.default_rng(7000 + 1000 * rank + k)
        leaf = int(rng.integers(0, P, dtype=np.uint64))
        siblings = rng.integers(0, P, size=(20, 4), dtype=np.uint64)
        lefts = rng.integers(0, 2, size=20).astype(np.uint8)
        cur = np.array([leaf, 0, 0, 0], np.uint64)
        for sb, l in zip(siblings, lefts):
            cur = pkg.two_to_one(sb, cur) if l else pkg.two_to_one(cur, sb)
        pw = pkg.PartialWitness()
        pkg.synthetic.assign_mmr_proof(leaf_t, proof_ts, peak_ts, cd.prover_only.public_inputs,
                                       (leaf, siblings, lefts, cur.reshape(1, 4), cur.copy()), pw.set_target)
        pws.append(pw)
    bp = pkg.BatchProver(cd, B)
    warr = (C.c_void_p * B)(*[w._h for w in pws])
    out = np.zeros((B, cd.info.proof_len), np.uint64)

    def step():
        Nn.check(lib.p2mt_batch_prover_prove(bp._h, warr, B, Nn.ptr(out), cd.info.proof_len, None))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        step()
    assert np.array_equal(out[0], cd.prove(pws[0])) and cd.verify(out[B - 1])  # batch == one at a time; accepted
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * B * args.steps / dt
    return {"metric": "proofs/s mmr_plonky2_verifier (batched prover, one replica per GPU)", "value": value, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup), "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 (Goldilocks + quadratic extension)",
            "data": "synthetic",
            "config": {"workload": "mmr_plonky2_verifier circuit_data.prove through p2mt_batch_prover, %d different statements per "
                                   "pass and rank (20 path elements + 1 peak, degree 2^%d), replicas only: no data-path collective"
                                   % (B, cd.degree_bits), "proofs_per_step": world * B}}


def run_recursion(args, torch, pkg, lib, cpu_baseline=True):
    """BASELINE.json config 4 / the second half of its metric: ms/proof of mmr_plonky2_verifier_1_recursion -- inner prove + outer
    prove (/root/reference/src/mmr/mmr_plonky2_verifier_1_recursion.rs:191-192 and :217-218) for one leaf of a 2^20-leaf MMR.
    One step = inner_circuit_data.prove(pw1) -> pw2.set_proof_with_pis_target(inner proof) -> main_circuit_data.prove(pw2), host
    PartialWitness in, host proof words out.  Both circuits are built once (as in the reference's driver the build is not part of
    `prove`); the outer circuit is plonky2's in-circuit verifier: 2^12 rows, 13 gate types."""
    Nn = pkg._native
    leaves = splitmix_leaves(1 << 20, 0x5EED0000 + 3)
    mmr = pkg.MMR.from_leaves(leaves)
    root = mmr.bagging_the_peaks()
    idx = 777777
    pr = mmr.get_proof_normal_index(idx)
    t0 = time.perf_counter()
    inner, leaf_t, proof_ts = pkg.verify_inner_merkle_proof_circuit(len(pr.siblings), len(pr.peaks))
    inner_build_ms = (time.perf_counter() - t0) * 1e3
    pw1 = pkg.PartialWitness()
    pw1.set_target(leaf_t, int(leaves[idx]))
    for (ht, bt), sib, left in zip(proof_ts, pr.siblings, pr.lefts):
        pw1.set_hash_target(ht, [int(x) for x in sib])
        pw1.set_bool_target(bt, bool(left))
    for i, pk in enumerate(pr.peaks):
        pw1.set_hash_target(inner.prover_only.public_inputs[4 * i:4 * i + 4], [int(x) for x in pk])
    t0 = time.perf_counter()
    outer, pt, vd, peak_ts = pkg.complete_verification_circuit_with_inner_proof(inner.common, len(pr.peaks))
    outer_build_ms = (time.perf_counter() - t0) * 1e3
    inner_proof = np.zeros(inner.info.proof_len, np.uint64)
    final_proof = np.zeros(outer.info.proof_len, np.uint64)
    pw2 = pkg.PartialWitness()
    times = {"inner": 0.0, "outer": 0.0}

    def one(timed=False):
        t0 = time.perf_counter()
        Nn.check(lib.p2mt_circuit_prove(inner._h, pw1._h, Nn.ptr(inner_proof), inner_proof.size))
        t1 = time.perf_counter()
        Nn.check(lib.p2mt_pw_clear(pw2._h))
        pw2.set_proof_with_pis_target(pt, inner_proof)
        pw2.set_verifier_data_target(vd, inner.verifier_only)
        for t, pk in zip(peak_ts, pr.peaks):
            pw2.set_hash_target(t, [int(x) for x in pk])
        for k, t in enumerate(outer.prover_only.public_inputs):
            pw2.set_target(t, int(root[k]))
        Nn.check(lib.p2mt_circuit_prove(outer._h, pw2._h, Nn.ptr(final_proof), final_proof.size))
        if timed:
            times["inner"] += t1 - t0
            times["outer"] += time.perf_counter() - t1

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one(True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    assert outer.verify(final_proof), "the product's verifier rejects the outer proof"
    assert np.array_equal(final_proof[-4:], root), "the outer proof's public input is not the MMR root"
    per_call = []
    for _ in range(max(args.steps // 2, 10)):
        t0 = time.perf_counter()
        outer.verify(final_proof)
        per_call.append((time.perf_counter() - t0) * 1e3)
    verify_ms = float(np.median(per_call))  # median of per-call wall times (see run_prove)
    counts = list(outer.info.gate_counts)
    out = {"metric": "ms/proof mmr_plonky2_verifier_1_recursion (inner + outer prove, one leaf of a 2^20-leaf MMR)", "value": ms,
           "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False,
           "scaling": "replicas only", "vs_baseline": None, "dtype": "u64 (Goldilocks + quadratic extension)", "data": "synthetic",
           "config": {"workload": "mmr_plonky2_verifier_1_recursion: inner_circuit_data.prove(pw1) (2^%d rows) + "
                                  "main_circuit_data.prove(pw2) (outer circuit = in-circuit verifier of the inner proof, 2^%d rows), "
                                  "standard_recursion_config, 20 path elements + 1 peak, host PartialWitness -> host proof words"
                                  % (inner.degree_bits, outer.degree_bits),
                      "inner_ms": times["inner"] * 1e3 / args.steps, "outer_ms": times["outer"] * 1e3 / args.steps,
                      "inner_proof_words": int(inner.info.proof_len), "outer_proof_words": int(outer.info.proof_len),
                      "circuit_build_ms": {"inner": inner_build_ms, "outer": outer_build_ms},
                      "outer_gate_rows": {k: int(counts[i]) for i, k in enumerate(
                          ("noop", "constant", "public_input", "arithmetic", "poseidon", "base_sum", "arithmetic_extension",
                           "mul_extension", "reducing", "reducing_extension", "random_access", "coset_interpolation",
                           "poseidon_mds"))}},
           "verify_outer_ms": verify_ms, "public_inputs": [int(x) for x in final_proof[-4:]],
           "roofline": {"bound": "latency", "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                        "note": "dependency chains (witness levels, leaf sponges, Merkle levels, host transcript phases), not bandwidth: "
                                "per-kernel split in profiles/r05_prove_timeline_recursion_hostchain.txt"}}
    if getattr(args, "workload", "") in ("recursion", "mmr"):
        # throughput through the batched prover (its own process: ~3.7 GB of per-proof blocks per thread at B = 32; four threads so
        # that a pass's one-workgroup witness interpreter and its latency-bound launches overlap the other passes' hashing: 1.46 k /
        # 1.66 k / 1.70 k proofs/s with 2 / 3 / 4 threads, round 4).  A shorter probe in the default line.
        secs = "4" if args.workload == "recursion" else "2"
        out["throughput"] = run_probe([sys.executable, os.path.join(ROOT, "tools", "recursion_batch_probe.py"), "32", secs, "4"])
        if "error" not in out["throughput"]:
            out["throughput"]["note"] = ("p2mt_batch_prover on the inner and on the outer circuit, four host threads; proofs "
                                         "bit-identical to the one-at-a-time path's; `value` above stays the single-proof latency")
    if cpu_baseline:
        from oracle import circuit as OC, recursion as R
        o = _oracle().Oracle()
        t0 = time.perf_counter()
        oi, oleaf, oproof_ts = OC.verify_inner_merkle_proof_circuit(o, len(pr.siblings), len(pr.peaks))
        opw = {oleaf: int(leaves[idx])}
        for (ht, bt), sib, left in zip(oproof_ts, pr.siblings, pr.lefts):
            for k in range(4):
                opw[ht[k]] = int(sib[k])
            opw[bt] = int(left)
        for i, pk in enumerate(pr.peaks):
            for k in range(4):
                opw[oi.public_inputs[4 * i + k]] = int(pk[k])
        oo, opt, ovd, opeak_ts = R.complete_verification_circuit_with_inner_proof(o, R.CommonData(oi), len(pr.peaks))
        build_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        want_inner = oi.prove(opw)
        opw2 = {}
        R.set_proof_with_pis_target(opw2.__setitem__, opt, want_inner)
        R.set_verifier_data_target(opw2.__setitem__, ovd, oi)
        for t, pk in zip(opeak_ts, pr.peaks):
            for k in range(4):
                opw2[t[k]] = int(pk[k])
        for k in range(4):
            opw2[oo.public_inputs[k]] = int(root[k])
        want = oo.prove(opw2)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        assert np.array_equal(inner_proof, want_inner) and np.array_equal(final_proof, want), "GPU proofs != oracle proofs"
        assert oo.verify(final_proof) == (True, 0), "oracle verifier rejects the GPU outer proof"
        out["cpu_baseline"] = {"value": cpu_ms, "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "oracle/circuit.py + oracle/recursion.py + oracle/*.c: inner + outer prove of the same "
                                         "circuits and witnesses, 1 proof (Python host logic + C field work, 1 thread; circuit "
                                         "builds excluded: %.1f s): a parity leg with a clock on it, no ratio is quoted" % build_s,
                               "proofs_equal": True}
    return out


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(n):
    """`python bench.py --gpus N ...` -> `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`
    as a child process (one rank per GPU over RCCL); returns its exit code.  Rank 0's JSON line reaches stdout unchanged."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser(
        description=__doc__.split("\n\n")[0],
        epilog="multi-GPU examples (no launcher needed; torch.distributed.run works too):\n"
               "  python bench.py --gpus 8                                   weak scaling: 2^24 leaves per GPU, one 2^27-leaf MMR\n"
               "  python bench.py --gpus 8 --scaling strong                  BASELINE's headline: one 2^24-leaf MMR over 8 GPUs\n"
               "  python bench.py --gpus 8 --log-leaves 23                   BASELINE config 5: 2^26 leaves, 2^23 per GPU\n"
               "  python bench.py --gpus 2 --backend gloo --single-device    rehearsal of the N > 1 path on a 1-GPU box\n"
               "  python bench.py --gpus 1 --force-collective                rehearsal of the RCCL exchange with one rank",
        formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-leaves", type=int, default=24,
                    help="2^this leaves per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", default=os.environ.get("P2MT_BENCH_SCALING", "weak"), choices=["weak", "strong"],
                    help="weak (default): per-GPU work fixed, every rank builds a 2^log_leaves shard of one 2^(log_leaves+log2 N)-leaf "
                         "MMR; strong: one 2^log_leaves-leaf MMR split over the ranks (BASELINE's 2^24 at 1/2/4/8 GPUs).  "
                         "Config 5 (2^26 over 8 GPUs): --gpus 8 --log-leaves 23, or --scaling strong --log-leaves 26.")
    ap.add_argument("--variant", default=None, help="mds,partial (e.g. 2,0) Poseidon kernel variant")
    ap.add_argument("--workload", default="mmr", choices=["mmr", "commit", "fri", "prove", "recursion"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prove", action="store_true", help="mmr workload: skip the secondary ms/proof leg")
    ap.add_argument("--threads", type=int, default=32, help="--workload prove: concurrent provers for the throughput leg (1 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --single-device exist only to exercise the N>1 code "
                         "path on a 1-GPU box")
    ap.add_argument("--single-device", action="store_true", help="test only: every rank uses GPU 0")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: still create the process group (world_size 1) and go through the device-resident exchange "
                         "(root -> all_gather_into_tensor -> combine launch): rehearses the RCCL path on a 1-GPU box")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's launcher would.  This runs before
        # torch is imported or any HIP call is made (a process that has touched the GPU must never be replaced or forked), the
        # ranks are fresh child processes, and their output and exit code are passed through.
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or without a "
                         "launcher at all (bench.py starts its own ranks)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    if not args.single_device and local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU (%d visible): one process per GPU; --single-device exists for rehearsals only"
                         % (local_rank, torch.cuda.device_count()))
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1 or args.force_collective:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:  # --force-collective without a launcher: a one-rank group on the loopback
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
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
    elif args.workload == "prove" and world > 1:
        out = run_prove_replicas(args, torch, pkg, lib, rank, world, dist)
    elif args.workload == "prove":
        out = run_prove(args, torch, pkg, lib) if rank == 0 else None
    elif args.workload == "recursion":
        out = run_recursion(args, torch, pkg, lib, cpu_baseline=not args.no_cpu_baseline) if rank == 0 else None
    else:
        out = run_mmr(args, torch, pkg, lib, rank, world, local_rank, dist)
    if rank == 0 and out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
