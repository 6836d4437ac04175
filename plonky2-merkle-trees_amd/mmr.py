"""Host-side mirror of /root/reference/src/mmr/merkle_mountain_ranges.rs over the C ABI.

  MMR::{new, add_leaf, bagging_the_peaks, get_peaks, get_proof, get_proof_normal_index,
        get_subtree_proof_elm} (:84-223), MMR_proof::verify (:232-252),
  get_mmr_index (:257-270), get_heights_bitmap_for_mmr_size (:39-81)
plus the bulk constructors the reference lacks (from_leaves / extend).  `elements` lives
in HBM behind an opaque handle; all hashing happens in the HIP library.
"""
import ctypes as C
import os

import numpy as np

from . import _native as N


def get_heights_bitmap_for_mmr_size(mmr_size):
    rem = C.c_size_t(0)
    bm = N.lib().p2mt_get_heights_bitmap_for_mmr_size(mmr_size, C.byref(rem))
    return bm, rem.value


def get_mmr_index(leaf_normal_index):
    r = N.lib().p2mt_get_mmr_index(leaf_normal_index)
    if r < 0:
        raise N.P2mtPanic(int(r), "get_mmr_index: i32 overflow in the reference (n >= 2^30)")
    return int(r)


class MMR_proof:
    """struct MMR_proof { mmr_size, merkle_proof: Vec<(HashOut, bool)>, peaks } (:15-23)."""

    def __init__(self, mmr_size, siblings, lefts, peaks):
        self.mmr_size = mmr_size
        self.siblings = siblings  # (k, 4) u64
        self.lefts = lefts        # (k,) u8, 1 = sibling on the left
        self.peaks = peaks        # (n_peaks, 4) u64

    @property
    def merkle_proof(self):
        return [(self.siblings[i], bool(self.lefts[i])) for i in range(len(self.lefts))]

    def verify(self, leaf, root):
        """MMR_proof::verify (:232-252). Raises P2mtPanic(ENOTPEAK) where the reference's assert! fires (:245)."""
        res = C.c_int(0)
        root = N.as_u64(root).reshape(4)
        sib = N.as_u64(self.siblings).reshape(-1, 4)
        lefts = np.ascontiguousarray(np.asarray(self.lefts, dtype=np.uint8))
        peaks = N.as_u64(self.peaks).reshape(-1, 4)
        N.check(N.lib().p2mt_mmr_proof_verify(N.ptr(sib), N.ptr(lefts), sib.shape[0], N.ptr(peaks), peaks.shape[0],
                                              int(leaf), N.ptr(root), C.byref(res)))
        return bool(res.value)


class MMR:
    """struct MMR { elements: Vec<HashOut> } (:8-12), device-resident."""

    def __init__(self):
        h = C.c_void_p()
        N.check(N.lib().p2mt_mmr_create(C.byref(h)))
        self._h = h

    @staticmethod
    def borrowed(handle, keepalive=None):
        """a view of a p2mt_mmr handle somebody else owns (p2mt_sharded_mmr_local): never destroyed from here"""
        m = MMR.__new__(MMR)
        m._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        m._borrowed = True
        m._keepalive = keepalive
        return m

    # ---- reference API
    @staticmethod
    def new():
        return MMR()

    def add_leaf(self, leaf):
        """MMR::add_leaf (:89-120); queued and flushed in bulk before the MMR is next observed."""
        N.check(N.lib().p2mt_mmr_add_leaf(self._h, int(leaf)))

    def bagging_the_peaks(self):
        out = np.zeros(4, np.uint64)
        N.check(N.lib().p2mt_mmr_root(self._h, N.ptr(out)))
        return out

    def get_peaks(self):
        out = np.zeros((N.MAX_PROOF_LEN, 4), np.uint64)
        n = C.c_int(0)
        N.check(N.lib().p2mt_mmr_peaks(self._h, N.ptr(out), C.byref(n)))
        return out[:n.value].copy()

    def get_proof(self, mmr_index):
        sib = np.zeros((N.MAX_PROOF_LEN, 4), np.uint64)
        lefts = np.zeros(N.MAX_PROOF_LEN, np.uint8)
        peaks = np.zeros((N.MAX_PROOF_LEN, 4), np.uint64)
        ns, npk, sz = C.c_int(0), C.c_int(0), C.c_size_t(0)
        N.check(N.lib().p2mt_mmr_proof(self._h, mmr_index, N.ptr(sib), N.ptr(lefts), C.byref(ns), N.ptr(peaks),
                                       C.byref(npk), C.byref(sz)))
        return MMR_proof(sz.value, sib[:ns.value].copy(), lefts[:ns.value].copy(), peaks[:npk.value].copy())

    def get_proof_normal_index(self, normal_index):
        return self.get_proof(get_mmr_index(normal_index))

    def get_subtree_proof_elm(self, mmr_index):
        return self.get_proof(mmr_index).merkle_proof

    # ---- bulk API (what the GPU is for)
    @staticmethod
    def from_leaves(leaves):
        m = MMR()
        m.extend(leaves)
        return m

    def reserve(self, n_leaves):
        N.check(N.lib().p2mt_mmr_reserve(self._h, n_leaves))

    def reset(self):
        N.check(N.lib().p2mt_mmr_reset(self._h))

    def extend(self, leaves):
        """k x add_leaf in level-synchronous launches; `leaves` is host data (numpy / list)."""
        leaves = N.as_u64(leaves).reshape(-1)
        N.check(N.lib().p2mt_mmr_extend(self._h, N.ptr(leaves), leaves.size))

    def extend_dev(self, d_leaves, k):
        """`d_leaves`: device pointer (int) or a CUDA/HIP torch tensor of k u64 (viewed as int64)."""
        N.check(N.lib().p2mt_mmr_extend_dev(self._h, N.ptr(d_leaves), k))

    @property
    def num_leaves(self):
        return N.lib().p2mt_mmr_num_leaves(self._h)

    def __len__(self):
        return N.lib().p2mt_mmr_len(self._h)

    @property
    def elements_dev(self):
        return N.lib().p2mt_mmr_elements_dev(self._h)

    @property
    def elements(self):
        return self.copy_elements(0, len(self))

    def copy_elements(self, first, count):
        out = np.zeros((count, 4), np.uint64)
        N.check(N.lib().p2mt_mmr_copy_elements(self._h, first, count, N.ptr(out)))
        return out

    def copy_elements_async(self, first, count, pinned):
        """enqueue-only copy of elements [first, first + count) into a PinnedBuffer; complete after sync()"""
        N.check(N.lib().p2mt_mmr_copy_elements_async(self._h, first, count, pinned.ptr))

    def extend_dev_to_host(self, d_leaves, k, pinned, chunk_log=22):
        """extend from device-resident leaves while streaming the appended elements into a PinnedBuffer (chunks of 2^chunk_log
        leaves: the copy of one chunk overlaps the hashing of the next); complete after sync()"""
        N.check(N.lib().p2mt_mmr_extend_dev_to_host(self._h, N.ptr(d_leaves), k, chunk_log, pinned.ptr))

    def save(self, path):
        """checkpoint: header + `elements` as LE u64x4 records (post-order)"""
        N.check(N.lib().p2mt_mmr_save(self._h, os.fsencode(path)))

    @staticmethod
    def load(path):
        m = MMR()
        N.check(N.lib().p2mt_mmr_load(m._h, os.fsencode(path)))
        return m

    def get_proof_batch(self, mmr_indices, max_siblings=N.MAX_PROOF_LEN):
        idx = N.as_u64(mmr_indices).reshape(-1)
        m = idx.size
        sib = np.zeros((m, max_siblings, 4), np.uint64)
        lefts = np.zeros((m, max_siblings), np.uint8)
        ns = np.zeros(m, np.int32)
        N.check(N.lib().p2mt_mmr_proof_batch(self._h, N.ptr(idx), m, max_siblings, N.ptr(sib), N.ptr(lefts), N.ptr(ns)))
        return sib, lefts, ns

    def __del__(self):
        try:
            if self._h and not getattr(self, "_borrowed", False):
                N.lib().p2mt_mmr_destroy(self._h)
            self._h = None
        except Exception:
            pass


class PinnedBuffer:
    """page-locked host memory from the library (p2mt_host_alloc_pinned) viewed as a numpy array of u64"""

    def __init__(self, n_words):
        p = C.c_void_p()
        N.check(N.lib().p2mt_host_alloc_pinned(n_words * 8, C.byref(p)))
        self.ptr = p
        self.n_words = n_words
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n_words,))

    def free(self):
        if self.ptr:
            self.array = None
            N.check(N.lib().p2mt_host_free_pinned(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def verify_proof_batch(siblings, lefts, n_siblings, peaks, leaves, root):
    """Batched MMR_proof::verify against one peak set: returns int8 status per proof
    (1 = true, 0 = false, -5 = the reference's assert!(peaks.contains) would panic)."""
    sib = N.as_u64(siblings)
    m, max_sib = sib.shape[0], sib.shape[1]
    lefts = np.ascontiguousarray(np.asarray(lefts, dtype=np.uint8)).reshape(m, max_sib)
    ns = np.ascontiguousarray(np.asarray(n_siblings, dtype=np.int32)).reshape(m)
    peaks = N.as_u64(peaks).reshape(-1, 4)
    leaves = N.as_u64(leaves).reshape(m)
    root = N.as_u64(root).reshape(4)
    status = np.zeros(m, np.int8)
    N.check(N.lib().p2mt_mmr_proof_verify_batch(N.ptr(sib), N.ptr(lefts), N.ptr(ns), max_sib, N.ptr(peaks),
                                                peaks.shape[0], N.ptr(leaves), N.ptr(root), m, N.ptr(status)))
    return status
