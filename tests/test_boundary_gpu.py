"""GPU tests of the boundary behaviours added in round 2: failure atomicity of the add_leaf queue and of checkpoints, the
device-side shard-root combine (N > 1 path on one device), the full-size bit-exact check of the headline configuration,
plonky2's own `MerkleTree.digests` order, and bench.py's multi-rank modes launched as real child processes."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


def test_failed_flush_keeps_queued_leaves(pkg, oracle):
    """a flush that fails (here: the device allocation while growing) must not drop the queued leaves --
    len()/num_leaves() already count them -- and the operation must be retryable."""
    N = pkg._native
    rng = np.random.default_rng(5)
    first = rng.integers(0, P, size=700, dtype=np.uint64)
    queued = rng.integers(0, P, size=900, dtype=np.uint64)
    m = pkg.MMR.from_leaves(first)          # exact first allocation: 2 * 700 nodes
    for v in queued:
        m.add_leaf(int(v))
    assert m.num_leaves == 1600
    N.check(N.lib().p2mt_debug_fail_allocs(1))
    try:
        with pytest.raises(N.P2mtError) as e:
            m.bagging_the_peaks()           # observes the MMR -> flush -> grow -> injected ENOMEM
        assert e.value.code == N.P2MT_ENOMEM
    finally:
        N.check(N.lib().p2mt_debug_fail_allocs(0))
    assert m.num_leaves == 1600 and len(m) == 2 * 1600 - bin(1600).count("1")
    om = oracle.mmr(np.concatenate([first, queued]))
    assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())   # the retry succeeds on the full leaf set
    assert np.array_equal(m.elements, om.elements)
    # the same for a bulk extend: a failed call leaves the handle exactly as it was
    more = rng.integers(0, P, size=5000, dtype=np.uint64)
    N.check(N.lib().p2mt_debug_fail_allocs(1))
    try:
        with pytest.raises(N.P2mtError):
            m.extend(more)
    finally:
        N.check(N.lib().p2mt_debug_fail_allocs(0))
    assert m.num_leaves == 1600 and np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())
    m.extend(more)
    om2 = oracle.mmr(np.concatenate([first, queued, more]))
    assert np.array_equal(m.elements, om2.elements)


def test_checkpoint_load_is_validated_before_anything_changes(pkg, oracle, tmp_path):
    """a truncated / oversized / lying checkpoint is refused with EINVAL before any large allocation, and a
    failed load leaves the handle untouched."""
    N = pkg._native
    leaves = np.arange(1, 1001, dtype=np.uint64)
    m = pkg.MMR.from_leaves(leaves)
    good = str(tmp_path / "good.mmr")
    m.save(good)
    blob = open(good, "rb").read()
    keep = pkg.MMR.from_leaves(leaves[:10])
    before = keep.elements.copy()

    def refused(data):
        path = str(tmp_path / "bad.mmr")
        open(path, "wb").write(data)
        rc = N.lib().p2mt_mmr_load(keep._h, os.fsencode(path))
        assert rc == N.P2MT_EINVAL, rc
        assert keep.num_leaves == 10 and np.array_equal(keep.elements, before)

    refused(blob[:-32])                                     # truncated payload
    refused(blob + b"\0" * 32)                              # trailing bytes
    # header claiming 2^39 leaves (tens of TiB of nodes) over a tiny payload: rejected by the size check, no allocation
    n_big = 1 << 39
    hdr = bytearray(blob[:32])
    hdr[8:16] = n_big.to_bytes(8, "little")
    hdr[16:24] = (2 * n_big - 1).to_bytes(8, "little")
    refused(bytes(hdr) + blob[32:])
    corrupt = bytearray(blob)
    corrupt[40] ^= 1                                        # checksum mismatch
    refused(bytes(corrupt))
    ok = pkg.MMR.load(good)
    assert np.array_equal(ok.elements, oracle.mmr(leaves).elements)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_combine_shard_roots_on_one_device(pkg, oracle, world):
    """The N > 1 device path on ONE GPU: W shards built as separate device-resident MMRs with extend_dev,
    p2mt_mmr_combine_shard_roots on their roots; shard spans, top nodes and root against the monolithic GPU MMR and the oracle."""
    import torch
    k = 14
    n, n_local = 1 << k, (1 << k) // world
    leaves = pkg.synthetic.splitmix_leaves(n, 77 + world)
    d_all = torch.from_numpy(leaves.view(np.int64)).cuda()
    mono = pkg.MMR()
    mono.extend_dev(d_all, n)
    mono_el = mono.elements
    full = oracle.mmr(leaves)
    assert np.array_equal(mono_el, full.elements)
    shards = [pkg.ShardedMMR(pkg, n_local, r, world, None) for r in range(world)]
    roots = []
    for r, sh in enumerate(shards):
        sh.local.reset()
        sh.local.extend_dev(d_all[r * n_local:(r + 1) * n_local], n_local)
        roots.append(sh.local.bagging_the_peaks())
        fp = sh.first_pos()
        assert np.array_equal(mono_el[fp:fp + 2 * n_local - 1], sh.local.elements), "shard %d span" % r
    roots = np.array(roots, np.uint64)
    sh = shards[0]
    sh.gather_roots = lambda local_root: roots          # the all-gather's result (exchange itself: tests/test_sharded_gloo.py)
    root = sh.finish(roots[0])
    assert np.array_equal(root, full.bagging_the_peaks()) and np.array_equal(root, mono.bagging_the_peaks())
    off = 0
    for h in range(1, sh.g + 1):
        for j in range(world >> h):
            assert np.array_equal(mono_el[sh.top_node_pos(h, j)], sh.top_nodes[off + j]), (h, j)
        off += world >> h
    # a cross-shard proof assembled from the last shard's bottom path and the combined top nodes verifies on the GPU
    last = shards[-1]
    last.shard_roots, last.top_nodes, last.root = roots, sh.top_nodes, root
    g = n - 3
    pr_local = last.local.get_proof_normal_index(g - (world - 1) * n_local)
    ts, tl = last.top_siblings(world - 1)
    sib, lefts = np.concatenate([pr_local.siblings, ts]), np.concatenate([pr_local.lefts, tl])
    ref = full.get_proof_normal_index(g)
    assert np.array_equal(sib, ref["siblings"]) and np.array_equal(lefts, ref["lefts"])
    assert pkg.MMR_proof(2 * n - 1, sib, lefts, root[None]).verify(int(leaves[g]), root)


def test_full_size_bit_exact_2pow24(pkg, oracle):
    """The headline configuration (2^24 leaves, the bench's own input) compared with the oracle at the FULL size: root and the
    SHA-256 of all 33 554 431 nodes, not a prefix."""
    import torch
    leaves = pkg.synthetic.bench_leaves(24, 0)
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    m = pkg.MMR()
    m.reserve(1 << 24)
    m.extend_dev(d, 1 << 24)
    gpu_root = m.bagging_the_peaks()
    gpu_sha = hashlib.sha256(m.elements.tobytes()).hexdigest()
    del m
    el, threads = oracle.mmr_build_pow2_parallel(leaves, min(os.cpu_count() or 1, 32))
    assert np.array_equal(el[-1], gpu_root)
    assert hashlib.sha256(el.tobytes()).hexdigest() == gpu_sha


@pytest.mark.parametrize("n,w,cap", [(64, 3, 2), (512, 135, 4), (16, 5, 4), (32, 20, 0), (4096, 16, 4)])
def test_plonky2_digest_layout(pkg, oracle, n, w, cap):
    """p2mt_merkle_digests_to_plonky2_layout == plonky2's own fill_subtree order (oracle/merkle_cap.py), and MerkleTree::prove's
    index formula finds the same siblings in it as the level-major walk."""
    from oracle import merkle_cap as MC
    rng = np.random.default_rng(n + w)
    leaves = rng.integers(0, P, size=(n, w), dtype=np.uint64)
    t = pkg.MerkleCapTree.new(leaves, cap)
    got = t.plonky2_digests()
    if n <= 512:
        want, want_cap = MC.merkle_tree_new(oracle, leaves, cap)
        assert np.array_equal(t.cap, want_cap)
        assert np.array_equal(got, want)
    for leaf in (0, 1, n // 2 - 1, n - 1, int(rng.integers(0, n))):
        assert np.array_equal(MC.prove(got, n, cap, leaf), t.prove(leaf))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("scaling,log_leaves", [("weak", 15), ("strong", 16)])
def test_bench_two_ranks_on_one_device(pkg, oracle, scaling, log_leaves):
    """bench.py's N = 2 path as the driver launches it (torch.distributed.run, fresh child processes), both ranks on GPU 0 with
    the gloo backend for the 32-byte exchange: build_dev -> all-gather -> p2mt_mmr_combine_shard_roots.  The printed root must be
    the oracle's root of the concatenated shards."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--log-leaves", str(log_leaves), "--scaling", scaling, "--no-prove",
           "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    local_log = log_leaves - 1 if scaling == "strong" else log_leaves
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["config"]["leaves_per_gpu"] == 1 << local_log
    leaves = np.concatenate([pkg.synthetic.bench_leaves(local_log, r) for r in range(2)])
    assert [int(x) for x in oracle.mmr(leaves).bagging_the_peaks()] == out["root"]
    assert out["config"]["hashes_per_step"] == 2 * ((1 << local_log) - 1) + 1


def test_bench_prove_replicas_on_one_device(pkg):
    """bench.py --workload prove --gpus 2: the prover does not shard, so every rank runs its own batched prover (replicas only);
    launched as the driver would (two child processes, both on GPU 0 here, gloo for the barrier)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--single-device", "--workload", "prove"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["unit"] == "proofs/s"
    assert out["config"]["proofs_per_step"] == 256 and out["value"] > 1000


# ---------------------------------------------------------------- round 3: the exchange stays in HBM; bench.py starts its own ranks
@pytest.mark.parametrize("world", [1, 2, 8, 64])
def test_device_resident_exchange(pkg, oracle, world):
    """p2mt_sharded_mmr_build_dev for ranks 0 and world - 1 of `world` shards on one device (the other ranks' roots arrive through the
    host-transport callback): root -> exchange -> p2mt_mmr_combine_shard_roots_dev -> one read-back, against the oracle's monolithic
    MMR (roots, every top node at its post-order position, the root)."""
    import ctypes as C
    import torch
    N, lib = pkg._native, pkg._native.lib()
    k = 12
    n, n_local = 1 << k, (1 << k) // world
    leaves = pkg.synthetic.splitmix_leaves(n, 1234 + world)
    d_all = torch.from_numpy(leaves.view(np.int64)).cuda()
    full = oracle.mmr(leaves)
    full_el = full.elements
    geo = pkg.ShardedMMR(pkg, n_local, 0, world, None)
    shard_roots = np.stack([full_el[geo.first_pos(q) + 2 * n_local - 2] for q in range(world)])
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))

    def exchange(user, mine, all_out):
        for q in range(4 * world):
            all_out[q] = int(shard_roots.reshape(-1)[q])
        return 0
    cb = CB(exchange)
    for r in sorted({0, world - 1}):
        h = C.c_void_p()
        if world == 1:
            N.check(lib.p2mt_sharded_mmr_create(C.byref(h), n_local, 0, 1, None))
        else:
            N.check(lib.p2mt_sharded_mmr_create_exchange(C.byref(h), n_local, r, world, cb, None))
        try:
            N.check(lib.p2mt_sharded_mmr_build_dev(h, N.ptr(d_all[r * n_local:(r + 1) * n_local])))
            root, roots, top = np.zeros(4, np.uint64), np.zeros((world, 4), np.uint64), np.zeros((max(world - 1, 1), 4), np.uint64)
            N.check(lib.p2mt_sharded_mmr_root(h, N.ptr(root), N.ptr(roots), N.ptr(top)))
            assert np.array_equal(root, full.bagging_the_peaks())
            assert np.array_equal(roots, shard_roots)
            off = 0
            for hh in range(1, geo.g + 1):
                for j in range(world >> hh):
                    assert np.array_equal(full_el[geo.top_node_pos(hh, j)], top[off + j]), (hh, j)
                off += world >> hh
        finally:
            N.check(lib.p2mt_sharded_mmr_destroy(h))
    d_roots = torch.zeros(4 * world, dtype=torch.int64, device="cuda")
    # refusals of the device entry point
    lib, ptr = pkg.lib(), pkg._native.ptr
    assert lib.p2mt_mmr_combine_shard_roots_dev(ptr(d_roots), 3, None, ptr(d_roots)) == pkg._native.P2MT_EINVAL
    assert lib.p2mt_mmr_combine_shard_roots_dev(ptr(d_roots), 2048, None, ptr(d_roots)) == pkg._native.P2MT_EINVAL
    assert lib.p2mt_mmr_combine_shard_roots_dev(None, 2, None, ptr(d_roots)) == pkg._native.P2MT_EINVAL


def _bench_json(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.mark.parametrize("scaling,log_leaves", [("weak", 14), ("strong", 15)])
def test_bench_starts_its_own_ranks(pkg, oracle, scaling, log_leaves):
    """`python bench.py --gpus 2 ...` with NO launcher around it (the command the driver uses for N = 1, with --gpus 2): bench.py
    spawns torch.distributed.run itself before touching the GPU, forwards rank 0's JSON line and the exit code."""
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
           "--single-device", "--log-leaves", str(log_leaves), "--scaling", scaling, "--no-prove", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    local_log = log_leaves - 1 if scaling == "strong" else log_leaves
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    leaves = np.concatenate([pkg.synthetic.bench_leaves(local_log, r) for r in range(2)])
    assert [int(x) for x in oracle.mmr(leaves).bagging_the_peaks()] == out["root"]
    # a failing child is not swallowed: more ranks than the (single) device without --single-device must exit non-zero
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--backend", "gloo", "--log-leaves", "10", "--no-prove", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=dict(env_clean, HSA_ENABLE_IPC_MODE_LEGACY="0"), timeout=900, cwd=ROOT)
    import torch
    if torch.cuda.device_count() < 2:
        assert bad.returncode != 0


def test_bench_rccl_exchange_one_rank(pkg, oracle):
    """The RCCL branch itself on the one GPU this box has: process group "nccl" with one rank, the device-resident exchange
    (p2mt_mmr_root_dev -> all_gather_into_tensor over RCCL -> combine launch -> read-back) inside bench.py's timed steps."""
    out = _bench_json([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "3",
                       "--warmup", "1", "--log-leaves", "16", "--no-prove", "--no-cpu-baseline"])
    leaves = pkg.synthetic.bench_leaves(16, 0)
    assert [int(x) for x in oracle.mmr(leaves).bagging_the_peaks()] == out["root"]
    assert out["config"]["exchange"].startswith("device")


def test_elements_to_pinned_host_memory(pkg, oracle):
    """p2mt_mmr_copy_elements_async and p2mt_mmr_extend_dev_to_host (the extend that streams the elements it appends, chunk by chunk,
    while later chunks hash): on an empty and on a non-empty MMR, ragged sizes, chunk sizes that do and do not divide the extend --
    every element == the oracle's `for leaf { add_leaf }` (merkle_mountain_ranges.rs:89-120)"""
    import torch
    N, lib = pkg._native, pkg._native.lib()
    first, more = 12345, (1 << 17) + 4099
    leaves = pkg.synthetic.splitmix_leaves(first + more, 0x5EED0707)
    om = oracle.mmr(leaves)
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    for chunk_log in (10, 14, 20):
        m = pkg.MMR.from_leaves(leaves[:first])
        len0 = len(m)
        pin = pkg.mmr.PinnedBuffer(4 * (len(om.elements) - len0))
        m.extend_dev_to_host(d[first:], more, pin, chunk_log=chunk_log)
        N.check(lib.p2mt_sync())
        assert len(m) == len(om.elements)
        assert np.array_equal(pin.array.reshape(-1, 4), om.elements[len0:])
        assert np.array_equal(m.bagging_the_peaks(), om.bagging_the_peaks())
        whole = pkg.mmr.PinnedBuffer(4 * len(m))
        m.copy_elements_async(0, len(m), whole)
        N.check(lib.p2mt_sync())
        assert np.array_equal(whole.array.reshape(-1, 4), om.elements)
        with pytest.raises(pkg.P2mtPanic):
            m.copy_elements_async(len(m), 1, whole)
        with pytest.raises(pkg.P2mtPanic):
            m.extend_dev_to_host(d, 16, whole, chunk_log=9)
        pin.free()
        whole.free()


def test_sharded_mmr_behind_the_c_abi(pkg, oracle):
    """p2mt_sharded_mmr_* (csrc/p2mt_sharded.hip): (1) a ONE-rank RCCL communicator made in C -- p2mt_nccl_unique_id +
    ncclCommInitRank inside p2mt_sharded_mmr_create_with_id, no torch.distributed anywhere -- carries the all-gather of the build;
    (2) world = 4 on one device through the host-exchange callback: every rank's handle gets the four shard roots from the callback,
    hashes the top levels on the device and assembles proofs for the leaves it owns; root, top nodes and proofs == the oracle's
    monolithic MMR (merkle_mountain_ranges.rs:89-120, :209-223)."""
    import ctypes as C
    import torch
    N, lib = pkg._native, pkg._native.lib()
    n_local = 1 << 12
    # (1) one rank, real RCCL
    leaves = pkg.synthetic.splitmix_leaves(n_local, 0x5EED0801)
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    uid = np.zeros(128, np.uint8)
    N.check(lib.p2mt_nccl_unique_id(N.ptr(uid)))
    assert uid.any()
    h = C.c_void_p()
    N.check(lib.p2mt_sharded_mmr_create_with_id(C.byref(h), n_local, 0, 1, N.ptr(uid)))
    try:
        om = oracle.mmr(leaves)
        for _ in range(3):
            N.check(lib.p2mt_sharded_mmr_build_dev(h, N.ptr(d)))
        root, roots = np.zeros(4, np.uint64), np.zeros((1, 4), np.uint64)
        N.check(lib.p2mt_sharded_mmr_root(h, N.ptr(root), N.ptr(roots), None))
        assert np.array_equal(root, om.bagging_the_peaks()) and np.array_equal(roots[0], root)
        sib, lefts, ns = np.zeros((64, 4), np.uint64), np.zeros(64, np.uint8), C.c_int(0)
        N.check(lib.p2mt_sharded_mmr_proof(h, 777, N.ptr(sib), N.ptr(lefts), C.byref(ns), N.ptr(root)))
        ref = om.get_proof_normal_index(777)
        assert ns.value == 12 and np.array_equal(sib[:12], ref["siblings"]) and np.array_equal(lefts[:12], ref["lefts"])
        assert lib.p2mt_sharded_mmr_proof(h, n_local, N.ptr(sib), N.ptr(lefts), C.byref(ns), N.ptr(root)) == N.P2MT_EINVAL
    finally:
        N.check(lib.p2mt_sharded_mmr_destroy(h))
    # arguments
    assert lib.p2mt_sharded_mmr_create(C.byref(h), n_local, 0, 2, None) == N.P2MT_EINVAL      # world > 1 without a communicator
    assert lib.p2mt_sharded_mmr_create(C.byref(h), n_local + 1, 0, 1, None) == N.P2MT_EINVAL  # not a power of two
    assert lib.p2mt_sharded_mmr_create(C.byref(h), n_local, 3, 2, None) == N.P2MT_EINVAL
    # (2) four ranks on one device, roots through the callback
    world = 4
    all_leaves = pkg.synthetic.splitmix_leaves(n_local * world, 0x5EED0802)
    full = oracle.mmr(all_leaves)
    shard_roots = np.stack([oracle.mmr(all_leaves[r * n_local:(r + 1) * n_local]).bagging_the_peaks() for r in range(world)])
    calls = []
    CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64))

    def exchange(user, mine, all_out):
        r = len(calls) % world
        calls.append([mine[k] for k in range(4)])
        assert calls[-1] == [int(x) for x in shard_roots[r]]
        for k in range(4 * world):
            all_out[k] = int(shard_roots.reshape(-1)[k])
        return 0
    cb = CB(exchange)
    handles = []
    try:
        for r in range(world):
            hr = C.c_void_p()
            # (world > 1 needs a transport: the unique-id constructor would wait for four real ranks, so these handles get the callback)
            N.check(lib.p2mt_sharded_mmr_create_exchange(C.byref(hr), n_local, r, world, cb, None))
            handles.append(hr)
            mine = all_leaves[r * n_local:(r + 1) * n_local]
            N.check(lib.p2mt_sharded_mmr_build(hr, N.ptr(mine)))
            root, roots, top = np.zeros(4, np.uint64), np.zeros((world, 4), np.uint64), np.zeros((world - 1, 4), np.uint64)
            N.check(lib.p2mt_sharded_mmr_root(hr, N.ptr(root), N.ptr(roots), N.ptr(top)))
            assert np.array_equal(root, full.bagging_the_peaks()) and np.array_equal(roots, shard_roots)
            for g in (r * n_local, r * n_local + 1234, (r + 1) * n_local - 1):
                sib, lefts, ns = np.zeros((64, 4), np.uint64), np.zeros(64, np.uint8), C.c_int(0)
                N.check(lib.p2mt_sharded_mmr_proof(hr, g, N.ptr(sib), N.ptr(lefts), C.byref(ns), N.ptr(root)))
                ref = full.get_proof_normal_index(g)
                assert ns.value == 14 and np.array_equal(sib[:14], ref["siblings"]) and np.array_equal(lefts[:14], ref["lefts"])
                assert oracle.mmr_proof_verify(sib[:14], lefts[:14], root[None], all_leaves[g], root)
            other = ((r + 1) % world) * n_local
            assert lib.p2mt_sharded_mmr_proof(hr, other, N.ptr(sib), N.ptr(lefts), C.byref(ns), N.ptr(root)) == N.P2MT_EINVAL
        assert len(calls) == world
    finally:
        for hr in handles:
            N.check(lib.p2mt_sharded_mmr_destroy(hr))


def test_device_exchange_through_a_process_group(pkg, oracle):
    """ShardedMMR with an RCCL process group is a CALLER of the C ABI (p2mt_sharded_mmr_*): it ships the ncclUniqueId over the group
    once and the build -- extend -> root -> ncclAllGather -> combine -- is one library call.  One rank on the loopback here (RCCL
    takes one rank per device); root and proofs == oracle."""
    import torch
    import torch.distributed as dist
    n = 1 << 13
    leaves = pkg.synthetic.splitmix_leaves(n, 4321)
    d = torch.from_numpy(leaves.view(np.int64)).cuda()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    try:
        sh = pkg.ShardedMMR(pkg, n, 0, 1, dist)
        assert sh._use_c()
        om = oracle.mmr(leaves)
        for _ in range(3):
            root = sh.build_dev(d)
        assert np.array_equal(root, om.bagging_the_peaks())
        assert np.array_equal(sh.shard_roots[0], root) and sh.top_nodes.shape == (0, 4)
        pr = sh.get_proof_normal_index(777)
        ref = om.get_proof_normal_index(777)
        assert np.array_equal(pr.siblings, ref["siblings"]) and np.array_equal(pr.lefts, ref["lefts"])
        assert np.array_equal(sh.local.elements, om.elements)  # the borrowed local shard is a full MMR handle
        assert np.array_equal(sh.build(leaves), root)
        sh.close()
    finally:
        dist.destroy_process_group()
