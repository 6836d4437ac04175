"""Leaf-range sharding of one 2^k-leaf MMR across the GPUs of a node.

One process per GPU.  Rank r of `world` (a power of two) owns leaves [r*n_local, (r+1)*n_local), builds that
perfect subtree locally (no data-path collective), then ONE all-gather of the 32-byte shard roots
(torch.distributed: backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) and
log2(world) top levels hashed redundantly on every rank.  With RCCL the whole build is ONE C-ABI call (`p2mt_sharded_mmr_build_dev`,
csrc/p2mt_sharded.hip: extend -> root -> ncclAllGather -> combine launch, all on the library stream, the roots never leave HBM) and this
class is a caller of it: it only gets the 128-byte ncclUniqueId from rank 0 to the other ranks over torch.distributed, once; with gloo
(the CPU tests) the roots travel through the host (`gather_roots` / `finish`).  The exchange is latency-bound (world x 32 B);
link bandwidth is irrelevant, so no ring/bucket tuning applies.

Global post-order geometry (A.4): rank r's nodes occupy [first_pos(r), first_pos(r) + 2*n_local - 1) with
first_pos(r) = 2*L0 - popcount(L0), L0 = r*n_local; the top node of height h above the shard roots whose
last shard is s sits at node_pos((s+1)*n_local - 1, log2(n_local) + h).
"""
import numpy as np

from . import _native as N
from .mmr import MMR, MMR_proof


def _log2(x):
    assert x > 0 and x & (x - 1) == 0, "must be a power of two"
    return x.bit_length() - 1


class ShardedMMR:
    def __init__(self, pkg, n_local, rank=0, world=1, dist=None):
        self.n_local, self.rank, self.world, self.dist = n_local, rank, world, dist
        self.k_local, self.g = _log2(n_local), _log2(world)
        self._local = None        # device-resident shard, created on first use (needs a GPU)
        self.shard_roots = None   # (world, 4) after a build
        self.top_nodes = None     # (world-1, 4) level-major bottom-up
        self.root = None

    # ---- the C-ABI handle (device path): p2mt_sharded_mmr owns the local shard, the communicator and the exchange
    def _c(self):
        if getattr(self, "_ch", None) is not None:
            return self._ch
        import ctypes as C
        lib = N.lib()
        h = C.c_void_p()
        if self.world == 1 and self.dist is None:  # no process group: nothing to exchange with
            N.check(lib.p2mt_sharded_mmr_create(C.byref(h), self.n_local, 0, 1, None))
        else:
            # a communicator of the library's own: rank 0 draws the id, everybody gets it over the process group that exists anyway
            uid = np.zeros(128, np.uint8)
            if self.rank == 0:
                N.check(lib.p2mt_nccl_unique_id(N.ptr(uid)))
            if self.world > 1:
                import torch
                dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
                t = torch.from_numpy(uid).to(dev)
                self.dist.broadcast(t, src=0)
                uid = t.cpu().numpy().copy()
            N.check(lib.p2mt_sharded_mmr_create_with_id(C.byref(h), self.n_local, self.rank, self.world, N.ptr(uid)))
        self._ch = h
        return h

    @property
    def local(self):
        if self._local is None:
            if self._use_c():
                self._local = MMR.borrowed(N.lib().p2mt_sharded_mmr_local(self._c()), keepalive=self)
            else:
                self._local = MMR()
                self._local.reserve(self.n_local)
        return self._local

    def close(self):
        if getattr(self, "_ch", None) is not None:
            self._local = None
            N.check(N.lib().p2mt_sharded_mmr_destroy(self._ch))
            self._ch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- geometry (pure index maths)
    def first_pos(self, rank=None):
        r = self.rank if rank is None else rank
        return N.lib().p2mt_mmr_shard_first_pos(self.n_local, r)

    def global_len(self):
        n = self.n_local * self.world
        return 2 * n - 1

    def top_node_pos(self, h, j):
        """height h >= 1 above the shard roots, index j among the nodes of that height"""
        last_leaf = ((j + 1) << h) * self.n_local - 1
        return N.lib().p2mt_mmr_node_pos(last_leaf, self.k_local + h)

    # ---- communication: the single exchange of the path
    def gather_roots(self, local_root):
        """all-gather of the 32-byte shard roots; returns (world, 4) u64 on the host."""
        local_root = N.as_u64(local_root).reshape(4)
        if self.world == 1:
            return local_root[None].copy()
        import torch
        backend = self.dist.get_backend()
        dev = "cuda" if backend == "nccl" else "cpu"
        mine = torch.from_numpy(local_root.view(np.int64).copy()).to(dev)
        out = torch.empty(self.world * 4, dtype=torch.int64, device=dev)
        self.dist.all_gather_into_tensor(out, mine)
        return out.cpu().numpy().view(np.uint64).reshape(self.world, 4).copy()

    # ---- build
    def _device_exchange(self):
        """True when the exchange can stay in HBM: RCCL ("nccl") moves device tensors, gloo needs host memory."""
        return self.dist is not None and self.dist.get_backend() == "nccl"

    def _use_c(self):
        """the whole build behind the C ABI: RCCL process groups, and the single process without a group"""
        return self._device_exchange() or (self.dist is None and self.world == 1)

    def build_dev(self, d_leaves):
        """d_leaves: this rank's n_local leaves, resident in HBM (torch tensor or raw device pointer)."""
        if self._use_c():
            N.check(N.lib().p2mt_sharded_mmr_build_dev(self._c(), N.ptr(d_leaves)))
            return self.finish_dev()
        self.local.reset()
        self.local.extend_dev(d_leaves, self.n_local)
        local_root = self.local.bagging_the_peaks()  # perfect subtree: one peak == its root
        return self.finish(local_root)

    def build(self, leaves):
        if self._use_c():
            leaves = N.as_u64(leaves).reshape(-1)
            assert leaves.size == self.n_local
            N.check(N.lib().p2mt_sharded_mmr_build(self._c(), N.ptr(leaves)))
            return self.finish_dev()
        self.local.reset()
        self.local.extend(leaves)
        return self.finish(self.local.bagging_the_peaks())

    def finish_dev(self):
        """One read-back behind the C-ABI build: [shard roots | top nodes | root]."""
        w = self.world
        roots = np.zeros((w, 4), np.uint64)
        top = np.zeros((max(w - 1, 1), 4), np.uint64)
        root = np.zeros(4, np.uint64)
        N.check(N.lib().p2mt_sharded_mmr_root(self._c(), N.ptr(root), N.ptr(roots), N.ptr(top)))
        self.shard_roots, self.top_nodes, self.root = roots, top[:w - 1].copy(), root
        return self.root

    def finish(self, local_root):
        self.shard_roots = self.gather_roots(local_root)
        if self.world == 1:
            self.top_nodes = np.zeros((0, 4), np.uint64)
            self.root = self.shard_roots[0].copy()
            return self.root
        top = np.zeros((self.world - 1, 4), np.uint64)
        root = np.zeros(4, np.uint64)
        N.check(N.lib().p2mt_mmr_combine_shard_roots(N.ptr(self.shard_roots), self.world, N.ptr(top), N.ptr(root)))
        self.top_nodes, self.root = top, root
        return root

    # ---- proofs spanning shards: bottom k_local siblings from the owner, top g from the gathered roots
    def top_siblings(self, owner):
        sib, lefts = [], []
        level = self.shard_roots
        off, idx, cnt = 0, owner, self.world
        while cnt > 1:
            sib.append(level[idx ^ 1])
            lefts.append(idx & 1)
            level = self.top_nodes[off:off + cnt // 2]
            off += cnt // 2
            idx >>= 1
            cnt //= 2
        return np.array(sib, dtype=np.uint64).reshape(-1, 4), np.array(lefts, dtype=np.uint8)

    def get_proof_normal_index(self, global_leaf):
        owner, local_idx = divmod(global_leaf, self.n_local)
        k = self.k_local
        if owner == self.rank:
            pr = self.local.get_proof_normal_index(local_idx)
            sib, lefts = pr.siblings, pr.lefts
        else:
            sib, lefts = np.zeros((k, 4), np.uint64), np.zeros(k, np.uint8)
        if self.world > 1:
            import torch
            dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"
            buf = np.concatenate([sib.reshape(-1).view(np.int64), lefts.astype(np.int64)])
            t = torch.from_numpy(buf.copy()).to(dev)
            self.dist.broadcast(t, src=owner)
            buf = t.cpu().numpy()
            sib = buf[:4 * k].view(np.uint64).reshape(k, 4).copy()
            lefts = buf[4 * k:].astype(np.uint8)
        ts, tl = self.top_siblings(owner)
        return MMR_proof(self.global_len(), np.concatenate([sib, ts]), np.concatenate([lefts, tl]),
                         self.root[None].copy())
