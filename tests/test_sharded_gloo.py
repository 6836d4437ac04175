"""N > 1 path on CPU: world_size 2 and 4 over gloo.  No GPU here, so the per-shard hashing is supplied by the
oracle (test infrastructure) and what is exercised is the product's exchange and geometry: the single
all-gather of 32-byte shard roots, shard/top-node post-order positions, and cross-shard proof assembly
(owner broadcast + top siblings) -- checked against the oracle's monolithic MMR of all leaves."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001
N_LOCAL = 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _OracleShard:
    """Test double for the device-resident shard: same get_proof_normal_index surface, oracle-backed."""

    def __init__(self, om):
        self.om = om

    def get_proof_normal_index(self, i):
        import types
        pr = self.om.get_proof_normal_index(i)
        return types.SimpleNamespace(siblings=pr["siblings"], lefts=pr["lefts"])


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle_lib import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = ge.load_package()
        o = Oracle()
        all_leaves = np.array([(i * 0x9E3779B97F4A7C15 + 99) % P for i in range(N_LOCAL * world)], dtype=np.uint64)
        full = o.mmr(all_leaves)
        full_el = full.elements
        mine = all_leaves[rank * N_LOCAL:(rank + 1) * N_LOCAL]
        local = o.mmr(mine)
        sh = pkg.ShardedMMR(pkg, N_LOCAL, rank, world, dist)

        # 1) the exchange: one all-gather of the 32-byte roots, rank order
        roots = sh.gather_roots(local.bagging_the_peaks())
        assert roots.shape == (world, 4)
        for r in range(world):
            exp = o.mmr(all_leaves[r * N_LOCAL:(r + 1) * N_LOCAL]).bagging_the_peaks()
            assert np.array_equal(roots[r], exp)

        # 2) geometry: this shard's nodes are a contiguous span of the global post-order array
        fp = sh.first_pos()
        assert np.array_equal(full_el[fp:fp + 2 * N_LOCAL - 1], local.elements)
        assert sh.global_len() == len(full)

        # 3) top nodes (hashed by the oracle here; on the GPU box p2mt_mmr_combine_shard_roots does it)
        top, level = [], roots
        while level.shape[0] > 1:
            level = np.stack([o.two_to_one(level[2 * j], level[2 * j + 1]) for j in range(level.shape[0] // 2)])
            top.append(level)
        sh.shard_roots, sh.top_nodes, sh.root = roots, np.concatenate(top), level[0]
        assert np.array_equal(sh.root, full.bagging_the_peaks())
        off = 0
        for h in range(1, sh.g + 1):
            for j in range(world >> h):
                assert np.array_equal(full_el[sh.top_node_pos(h, j)], sh.top_nodes[off + j])
            off += world >> h

        # 4) cross-shard proofs: owner's bottom siblings are broadcast, top siblings come from the gathered roots
        sh._local = _OracleShard(local)
        for g in (0, 1, N_LOCAL - 1, N_LOCAL, N_LOCAL * world - 1, (N_LOCAL * world) // 2 + 3):
            pr = sh.get_proof_normal_index(g)
            ref = full.get_proof_normal_index(g)
            assert np.array_equal(pr.siblings, ref["siblings"]), g
            assert np.array_equal(pr.lefts, ref["lefts"]), g
            assert np.array_equal(pr.peaks, ref["peaks"]) and pr.mmr_size == ref["mmr_size"]
            assert o.mmr_proof_verify(pr.siblings, pr.lefts, pr.peaks, all_leaves[g], sh.root)
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure to the parent
        import traceback
        q.put((rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_mmr_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_world1_geometry():
    import __graft_entry__ as ge
    pkg = ge.load_package()
    sh = pkg.ShardedMMR(pkg, 1 << 10, 0, 1, None)
    assert sh.first_pos() == 0 and sh.global_len() == 2047
    roots = sh.gather_roots([1, 2, 3, 4])
    assert roots.tolist() == [[1, 2, 3, 4]]
