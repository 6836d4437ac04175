#!/usr/bin/env python3
"""bench.py -- Poseidon hashes/s of a full MMR build (BASELINE.json metric), one process per GPU.

A "step" is one complete build of a 2^LOG-leaf MMR (default 2^24, the size BASELINE.json's metric is quoted
on) from leaves already resident in HBM: reset + extend == 2^24 x MMR::add_leaf
(/root/reference/src/mmr/merkle_mountain_ranges.rs:89-120) + the all-gather/top-levels combine when N > 1.
Unit of work: one `two_to_one` Poseidon permutation; an N-leaf build performs N - popcount(N) of them.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (weak scaling: every rank builds its own 2^LOG shard)

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
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


def splitmix_leaves(n, seed):
    from conftest import splitmix_leaves as f
    return f(n, seed)


def cpu_baseline(target_seconds=12.0, max_log=22):
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
    return {"value": hashes / dt, "unit": "Poseidon hashes/s", "cores": 1, "kind": "port",
            "sample": "oracle/mmr.c add_leaf loop, first 2^%d leaves of the bench input, %.1f s" % (log_n, dt),
            "_root": m.bagging_the_peaks(), "_log_n": log_n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log-leaves", type=int, default=24, help="leaves per GPU = 2^this")
    ap.add_argument("--variant", default=None, help="mds,partial (e.g. 1,0) Poseidon kernel variant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    pkg = ge.load_package()
    pkg.init(local_rank)
    if args.variant:
        pkg.set_variant(*[int(x) for x in args.variant.split(",")])
    lib = pkg.lib()
    import ctypes as C
    mds, partial = C.c_int(), C.c_int()
    lib.p2mt_get_variant(C.byref(mds), C.byref(partial))

    n = 1 << args.log_leaves
    host_leaves = splitmix_leaves(n, 0x5EED0000 + 24 + 1000 * rank)
    d_leaves = torch.from_numpy(host_leaves.view(np.int64)).cuda()
    from plonky2_merkle_trees_amd import distributed as pdist
    shard = pdist.ShardedMMR(pkg, n, rank, world, dist)

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
    pkg._native.check(lib.p2mt_timer_start())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        root = step()
    kernel_ms = C.c_float(0)
    pkg._native.check(lib.p2mt_timer_stop(C.byref(kernel_ms)))
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hashes_per_rank = n - bin(n).count("1")
    total_hashes = hashes_per_rank * world + (world - 1)
    ms_per_step = elapsed * 1e3 / args.steps
    value = total_hashes / (ms_per_step * 1e-3)

    if rank == 0:
        # HIP-event time of this rank's launches over the timed region (library stream)
        dev_ms_per_step = kernel_ms.value / args.steps
        achieved_gbs = hashes_per_rank * ALGO_BYTES_PER_HASH / (dev_ms_per_step * 1e-3) / 1e9
        out = {
            "metric": "Poseidon hashes/s (MMR build, 2^%d leaves per GPU)" % args.log_leaves,
            "value": value, "unit": "Poseidon hashes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64 (Goldilocks, integer VALU)", "data": "synthetic",
            "config": {"workload": "mmr::merkle_mountain_ranges build, 2^%d leaves per GPU, device-resident leaves"
                                   % args.log_leaves,
                       "leaves_per_gpu": n, "hashes_per_step": total_hashes,
                       "poseidon_variant": {"mds": mds.value, "partial": partial.value},
                       "sharding": "leaf ranges per rank + all-gather of 32-byte shard roots" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_mmr_level (all levels of one build)", "device_ms_per_step": dev_ms_per_step,
                         "algorithmic_bytes_per_hash": ALGO_BYTES_PER_HASH,
                         "note": "Poseidon is integer-issue bound (~1e3 64-bit modmuls per 72 B); see DESIGN.md"},
            "root": [int(x) for x in root],
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            # parity check of the sample: GPU build of the same prefix must give the oracle's root
            sub = pkg.MMR.from_leaves(host_leaves[:1 << cb["_log_n"]])
            assert np.array_equal(sub.bagging_the_peaks(), cb["_root"]), "GPU root != oracle root on the CPU sample"
            cb = {k: v for k, v in cb.items() if not k.startswith("_")}
            cb["gpu_over_cpu"] = value / cb["value"]
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
