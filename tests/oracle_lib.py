"""ctypes binding of oracle/liboracle.so (the CPU restatement; TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001

_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)


def build_oracle(force=False):
    if os.environ.get("P2MT_ORACLE_SO"):  # the sanitizer leg (oracle/Makefile asan) points the tests at its own build
        return os.environ["P2MT_ORACLE_SO"]
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def build_fast_port_native():
    """The tuned scalar port compiled for THIS machine (-O3 -march=native) into a temp dir: bench.py's cpu_baseline.port_fast.
    (liboracle.so itself is built without -march=native because the built file travels to the GPU box.)"""
    import tempfile
    d = tempfile.mkdtemp(prefix="p2mt_fast_port_")
    so = os.path.join(d, "libfastport.so")
    # clang (ROCm's LLVM is in the image) schedules the 128-bit multiply / branch-free reduce chains ~30 % better than gcc 11
    clang = "/opt/rocm/lib/llvm/bin/clang"
    cc = [clang] if os.path.exists(clang) else ["gcc"]
    try:
        subprocess.check_call(cc + ["-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", so,
                                    os.path.join(ORACLE_DIR, "poseidon_fast.c"), "-I", ORACLE_DIR])
    except (subprocess.CalledProcessError, OSError):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", so,
                               os.path.join(ORACLE_DIR, "poseidon_fast.c"), "-I", ORACLE_DIR])
    lib = C.CDLL(so)
    _bind_fast(lib)
    return lib


def build_avx512_port_native():
    """oracle/poseidon_avx512.c compiled for THIS machine (-O3 -march=native) into a temp dir, or None when the host has no
    AVX-512 (the file then compiles to stubs): bench.py's cpu_baseline.port_fast on hosts that have it."""
    import tempfile
    d = tempfile.mkdtemp(prefix="p2mt_avx512_port_")
    so = os.path.join(d, "libavx512port.so")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    src = os.path.join(ORACLE_DIR, "poseidon_avx512.c")
    for cc in ([clang] if os.path.exists(clang) else []) + ["gcc"]:
        try:
            subprocess.check_call([cc, "-O3", "-march=native", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", so, src, "-I", ORACLE_DIR],
                                  stderr=subprocess.DEVNULL)
            break
        except (subprocess.CalledProcessError, OSError):
            continue
    else:
        return None
    lib = C.CDLL(so)
    lib.oracle_avx512_available.restype = C.c_int
    if not lib.oracle_avx512_available():
        return None
    lib.oracle_avx512_permute_batch.argtypes = [_u64p, _u64p, C.c_size_t]
    lib.oracle_avx512_mmr_build_pow2.argtypes = [_u64p, C.c_size_t, _u64p, C.c_int]
    lib.oracle_avx512_mmr_build_pow2.restype = C.c_int
    return lib


def avx512_permute_batch(lib, states):
    a = _arr(states, (-1, 12))
    out = np.empty_like(a)
    lib.oracle_avx512_permute_batch(_ptr(a), _ptr(out), a.shape[0])
    return out


def avx512_mmr_build_pow2(lib, leaves, threads=1):
    a = _arr(leaves)
    el = np.empty((2 * a.size - 1, 4), np.uint64)
    used = lib.oracle_avx512_mmr_build_pow2(_ptr(a), a.size, _ptr(el), threads)
    return el, used


def avx512_mmr_build_pow2_into(lib, leaves, el, threads=1):
    """the same into a caller-owned (already resident) node array"""
    a = _arr(leaves)
    assert el.shape == (2 * a.size - 1, 4) and el.dtype == np.uint64 and el.flags.c_contiguous
    used = lib.oracle_avx512_mmr_build_pow2(_ptr(a), a.size, _ptr(el), threads)
    return el, used


def _bind_fast(lib):
    lib.oracle_fast_poseidon_permute.argtypes = [_u64p]
    lib.oracle_fast_two_to_one_batch.argtypes = [_u64p, _u64p, C.c_size_t]
    lib.oracle_fast_mmr_add_leaf_loop.argtypes = [_u64p, C.c_size_t, _u64p]
    lib.oracle_fast_mmr_build_pow2.argtypes = [_u64p, C.c_size_t, _u64p, C.c_int]


def _ptr(a):
    return a.ctypes.data_as(_u64p)


def _arr(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    if shape is not None:
        a = a.reshape(shape)
    return a


# ---- plonky2 FRI structures (oracle/oracle.h)
class FriParams(C.Structure):
    _fields_ = [("degree_bits", C.c_uint32), ("rate_bits", C.c_uint32), ("cap_height", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32), ("num_query_rounds", C.c_uint32), ("num_reductions", C.c_uint32),
                ("reduction_arity_bits", C.c_uint32 * 8)]


class FriOracle(C.Structure):
    _fields_ = [("coeffs", _u64p), ("leaves", _u64p), ("digests", _u64p), ("n_polys", C.c_uint64)]


class FriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("polys", C.POINTER(C.c_uint32)), ("n_polys", C.c_uint64)]


class PlonkDesc(C.Structure):
    _fields_ = [(k, C.c_uint32) for k in ("degree_bits", "num_wires", "num_routed", "num_constants", "num_selectors",
                                          "num_challenges", "quotient_degree_factor", "num_gates")] + \
               [(k, C.c_uint32 * 16) for k in ("gate_kind", "gate_selector", "group_start", "group_end")]


class ChallengerState(C.Structure):
    _fields_ = [("state", C.c_uint64 * 12), ("inp", C.c_uint64 * 8), ("out", C.c_uint64 * 8), ("n_in", C.c_uint32),
                ("n_out", C.c_uint32)]


def make_fri_batches(batch_cls, batches):
    """batches: [(point(2,), [(oracle_index, poly_index), ...]), ...] -> (ctypes array, keep-alive list)"""
    arr = (batch_cls * len(batches))()
    keep = []
    for i, (point, polys) in enumerate(batches):
        pl = np.ascontiguousarray(np.asarray(polys, dtype=np.uint32).reshape(-1, 2))
        keep.append(pl)
        arr[i].point[0], arr[i].point[1] = int(point[0]), int(point[1])
        arr[i].polys = pl.ctypes.data_as(C.POINTER(C.c_uint32))
        arr[i].n_polys = pl.shape[0]
    return arr, keep


class Oracle:
    def __init__(self):
        self.lib = lib = C.CDLL(build_oracle())
        lib.oracle_gl_add.restype = lib.oracle_gl_sub.restype = lib.oracle_gl_mul.restype = C.c_uint64
        lib.oracle_gl_pow.restype = lib.oracle_gl_inv.restype = C.c_uint64
        lib.oracle_gl_primitive_root_of_unity.restype = C.c_uint64
        for f in (lib.oracle_gl_add, lib.oracle_gl_sub, lib.oracle_gl_mul, lib.oracle_gl_pow):
            f.argtypes = [C.c_uint64, C.c_uint64]
        lib.oracle_gl_inv.argtypes = [C.c_uint64]
        lib.oracle_gl_primitive_root_of_unity.argtypes = [C.c_uint]
        lib.oracle_poseidon_permute.argtypes = [_u64p]
        lib.oracle_two_to_one.argtypes = [_u64p, _u64p, _u64p]
        lib.oracle_poseidon_gate_witness.argtypes = [_u64p, C.c_int, _u64p]
        lib.oracle_hash_no_pad.argtypes = [_u64p, C.c_size_t, _u64p]
        lib.oracle_hash_or_noop.argtypes = [_u64p, C.c_size_t, _u64p]
        lib.oracle_merkle_build.argtypes = [_u64p, C.c_size_t, _u64p, _u64p]
        lib.oracle_merkle_get_proof.argtypes = [_u64p, C.c_size_t, C.c_size_t, _u64p]
        lib.oracle_merkle_get_in_between_hashes.argtypes = [_u64p, _u64p, C.c_size_t, C.c_size_t, _u64p]
        lib.oracle_verify_merkle_proof.argtypes = [C.c_uint64, C.c_size_t, _u64p, _u64p, C.c_size_t]
        lib.oracle_get_heights_bitmap_for_mmr_size.argtypes = [C.c_size_t, C.POINTER(C.c_size_t)]
        lib.oracle_get_heights_bitmap_for_mmr_size.restype = C.c_uint64
        lib.oracle_get_mmr_index.argtypes = [C.c_size_t]
        lib.oracle_get_mmr_index.restype = C.c_size_t
        lib.oracle_mmr_new.restype = C.c_void_p
        lib.oracle_mmr_free.argtypes = [C.c_void_p]
        lib.oracle_mmr_add_leaf.argtypes = [C.c_void_p, C.c_uint64]
        lib.oracle_mmr_add_leaves.argtypes = [C.c_void_p, _u64p, C.c_size_t]
        lib.oracle_mmr_build_pow2_parallel.argtypes = [_u64p, C.c_size_t, _u64p, C.c_int]
        _bind_fast(lib)
        lib.oracle_poseidon_round_constants.argtypes = [_u64p]
        lib.oracle_mmr_len.argtypes = [C.c_void_p]
        lib.oracle_mmr_len.restype = C.c_size_t
        lib.oracle_mmr_elements.argtypes = [C.c_void_p]
        lib.oracle_mmr_elements.restype = _u64p
        lib.oracle_mmr_get_peaks.argtypes = [C.c_void_p, _u64p]
        lib.oracle_mmr_bagging_the_peaks.argtypes = [C.c_void_p, _u64p]
        lib.oracle_mmr_get_subtree_proof_elm.argtypes = [C.c_void_p, C.c_size_t, _u64p, _u8p]
        lib.oracle_mmr_get_proof.argtypes = [C.c_void_p, C.c_size_t, _u64p, _u8p, C.POINTER(C.c_int), _u64p,
                                             C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        lib.oracle_mmr_proof_verify.argtypes = [_u64p, _u8p, C.c_int, _u64p, C.c_int, C.c_uint64, _u64p]
        lib.oracle_fft.argtypes = [_u64p, C.c_uint]
        lib.oracle_ifft.argtypes = [_u64p, C.c_uint]
        lib.oracle_coset_lde.argtypes = [_u64p, C.c_uint, C.c_uint, C.c_uint64, _u64p]
        lib.oracle_merkle_cap_commit.argtypes = [_u64p, C.c_size_t, C.c_size_t, C.c_uint, _u64p, _u64p]
        lib.oracle_polynomial_batch_commit.argtypes = [_u64p, C.c_int, C.c_size_t, C.c_uint, C.c_uint, C.c_uint,
                                                       _u64p, _u64p, _u64p]
        lib.oracle_permutation_partial_products.argtypes = [_u64p, _u64p, _u64p, _u64p, _u64p, C.c_size_t, C.c_size_t,
                                                            C.c_uint, C.c_uint, _u64p]
        lib.oracle_plonk_quotient_polys.argtypes = [C.POINTER(PlonkDesc)] + [_u64p] * 9
        lib.oracle_plonk_check_openings.argtypes = [C.POINTER(PlonkDesc)] + [_u64p] * 13
        lib.oracle_gate_constraints_row.argtypes = [C.c_uint, _u64p, _u64p, _u64p, _u64p]
        lib.oracle_challenger_init.argtypes = [C.POINTER(ChallengerState)]
        lib.oracle_challenger_observe.argtypes = [C.POINTER(ChallengerState), _u64p, C.c_size_t]
        lib.oracle_challenger_get.argtypes = [C.POINTER(ChallengerState)]
        lib.oracle_challenger_get.restype = C.c_uint64
        lib.oracle_ext_mul.argtypes = [_u64p, _u64p, _u64p]
        lib.oracle_ext_inv.argtypes = [_u64p, _u64p]
        lib.oracle_fri_params_standard.argtypes = [C.c_uint, C.POINTER(FriParams)]
        lib.oracle_fri_proof_len.argtypes = [C.POINTER(FriParams), C.c_size_t, _u64p]
        lib.oracle_fri_proof_len.restype = C.c_size_t
        lib.oracle_eval_polys_ext.argtypes = [_u64p, C.c_size_t, C.c_uint, _u64p, _u64p]
        lib.oracle_fri_prove.argtypes = [C.POINTER(FriOracle), C.c_size_t, C.POINTER(FriBatch), C.c_size_t,
                                         C.POINTER(FriParams), C.POINTER(ChallengerState), _u64p]
        lib.oracle_fri_verify.argtypes = [_u64p, C.c_size_t, _u64p, C.POINTER(FriBatch), C.c_size_t, _u64p,
                                          C.POINTER(FriParams), C.POINTER(ChallengerState), _u64p, C.POINTER(C.c_int)]

    # ---- field
    def mul(self, a, b): return self.lib.oracle_gl_mul(a, b)
    def add(self, a, b): return self.lib.oracle_gl_add(a, b)
    def sub(self, a, b): return self.lib.oracle_gl_sub(a, b)
    def pow(self, a, e): return self.lib.oracle_gl_pow(a, e)
    def inv(self, a): return self.lib.oracle_gl_inv(a)
    def root_of_unity(self, log_n): return self.lib.oracle_gl_primitive_root_of_unity(log_n)

    # ---- poseidon
    def permute(self, state):
        s = _arr(state).copy()
        assert s.size == 12
        self.lib.oracle_poseidon_permute(_ptr(s))
        return s

    def permute_batch(self, states):
        s = _arr(states).reshape(-1, 12).copy()
        for row in s:
            self.lib.oracle_poseidon_permute(_ptr(row))
        return s

    def poseidon_gate_witness(self, state, swap):
        s, out = _arr(state).copy(), np.zeros(135, np.uint64)
        self.lib.oracle_poseidon_gate_witness(_ptr(s), int(swap), _ptr(out))
        return out

    def two_to_one(self, l, r):
        l, r, out = _arr(l), _arr(r), np.zeros(4, np.uint64)
        self.lib.oracle_two_to_one(_ptr(l), _ptr(r), _ptr(out))
        return out

    def hash_no_pad(self, x):
        x, out = _arr(x), np.zeros(4, np.uint64)
        self.lib.oracle_hash_no_pad(_ptr(x), x.size, _ptr(out))
        return out

    def hash_or_noop(self, x):
        x, out = _arr(x), np.zeros(4, np.uint64)
        self.lib.oracle_hash_or_noop(_ptr(x), x.size, _ptr(out))
        return out

    # ---- simple merkle tree
    def merkle_build(self, leaves):
        leaves = _arr(leaves)
        n = leaves.size
        levels = np.zeros((max(2 * n - 2, 1), 4), np.uint64)
        root = np.zeros(4, np.uint64)
        k = self.lib.oracle_merkle_build(_ptr(leaves), n, _ptr(levels), _ptr(root))
        if k < 0:
            raise ValueError("reference panics: leaf count must be a power of two >= 2")
        return k, levels, root

    def merkle_get_proof(self, levels, n, idx):
        k = n.bit_length() - 1
        out = np.zeros((k, 4), np.uint64)
        rc = self.lib.oracle_merkle_get_proof(_ptr(levels), n, idx, _ptr(out))
        if rc < 0:
            raise IndexError("reference asserts leaf_index < n")
        return out

    def merkle_get_in_between_hashes(self, levels, root, n, idx):
        k = n.bit_length() - 1
        out = np.zeros((k, 4), np.uint64)
        root = _arr(root)
        rc = self.lib.oracle_merkle_get_in_between_hashes(_ptr(levels), _ptr(root), n, idx, _ptr(out))
        if rc < 0:
            raise IndexError("reference asserts leaf_index < n")
        return out[:rc]

    def verify_merkle_proof(self, leaf, idx, root, hashes):
        root, hashes = _arr(root), _arr(hashes).reshape(-1, 4)
        return bool(self.lib.oracle_verify_merkle_proof(int(leaf), idx, _ptr(root), _ptr(hashes), hashes.shape[0]))

    # ---- mmr
    def heights_bitmap(self, size):
        rem = C.c_size_t(0)
        bm = self.lib.oracle_get_heights_bitmap_for_mmr_size(size, C.byref(rem))
        return bm, rem.value

    def get_mmr_index(self, n): return self.lib.oracle_get_mmr_index(n)

    def mmr(self, leaves=None):
        return OracleMMR(self, leaves)

    def mmr_build_pow2_parallel(self, leaves, threads=0, out=None):
        """level-parallel build on `threads` host threads (0 = all); returns (elements, threads)."""
        leaves = _arr(leaves)
        n = leaves.size
        assert n and n & (n - 1) == 0
        el = out if out is not None else np.empty((2 * n - 1, 4), np.uint64)
        threads = self.lib.oracle_mmr_build_pow2_parallel(_ptr(leaves), n, _ptr(el), threads)
        return el, threads

    def fast_mmr_add_leaf_loop(self, leaves, lib=None):
        """for leaf { add_leaf } on the tuned scalar port, one thread -> elements (2n - popcount n, 4)"""
        leaves = _arr(leaves)
        n = leaves.size
        el = np.empty((2 * n - bin(n).count("1"), 4), np.uint64)
        (lib or self.lib).oracle_fast_mmr_add_leaf_loop(_ptr(leaves), n, _ptr(el))
        return el

    def fast_mmr_build_pow2(self, leaves, threads=0, out=None, lib=None):
        leaves = _arr(leaves)
        n = leaves.size
        assert n and n & (n - 1) == 0
        el = out if out is not None else np.empty((2 * n - 1, 4), np.uint64)
        threads = (lib or self.lib).oracle_fast_mmr_build_pow2(_ptr(leaves), n, _ptr(el), threads)
        return el, threads

    def poseidon_round_constants(self):
        out = np.zeros(360, np.uint64)
        self.lib.oracle_poseidon_round_constants(_ptr(out))
        return out

    def fast_permute(self, state):
        s = _arr(state).copy()
        self.lib.oracle_fast_poseidon_permute(_ptr(s))
        return s

    def mmr_proof_verify(self, siblings, lefts, peaks, leaf, root):
        sib = _arr(siblings).reshape(-1, 4)
        lf = np.ascontiguousarray(np.asarray(lefts, dtype=np.uint8))
        pk = _arr(peaks).reshape(-1, 4)
        root = _arr(root)
        rc = self.lib.oracle_mmr_proof_verify(_ptr(sib), lf.ctypes.data_as(_u8p), sib.shape[0], _ptr(pk),
                                              pk.shape[0], int(leaf), _ptr(root))
        if rc < 0:
            raise AssertionError("reference panics: assert!(self.peaks.contains(&next_hash))")
        return bool(rc)

    # ---- fft / commit
    def fft(self, a):
        a = _arr(a).copy()
        self.lib.oracle_fft(_ptr(a), a.size.bit_length() - 1)
        return a

    def ifft(self, a):
        a = _arr(a).copy()
        self.lib.oracle_ifft(_ptr(a), a.size.bit_length() - 1)
        return a

    def polynomial_batch_commit_parallel(self, polys, is_values, rate_bits=3, cap_height=4, threads=0):
        """B4: the commit on `threads` host cores (0 = all) -> (cap, threads used).  Test infrastructure / cpu_baseline only."""
        polys = _arr(polys)
        n_polys, n = polys.shape
        big = n << rate_bits
        leaves = np.zeros((big, n_polys), np.uint64)
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        fn = self.lib.oracle_polynomial_batch_commit_parallel
        fn.restype = C.c_int
        fn.argtypes = [_u64p, C.c_int, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, _u64p, _u64p, C.c_int]
        used = fn(_ptr(polys), int(is_values), n_polys, n.bit_length() - 1, rate_bits, cap_height, _ptr(leaves), _ptr(cap), threads)
        if used < 0:
            raise ValueError("oracle_polynomial_batch_commit_parallel: status %d" % used)
        return cap, used

    def coset_lde(self, coeffs, rate_bits, shift=7):
        c = _arr(coeffs)
        out = np.zeros(c.size << rate_bits, np.uint64)
        self.lib.oracle_coset_lde(_ptr(c), c.size.bit_length() - 1, rate_bits, shift, _ptr(out))
        return out

    def merkle_cap_commit(self, leaves, cap_height):
        leaves = _arr(leaves)
        n, w = leaves.shape
        k = n.bit_length() - 1
        n_dig = sum(n >> j for j in range(k - cap_height))
        digests = np.zeros((max(n_dig, 1), 4), np.uint64)
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        rc = self.lib.oracle_merkle_cap_commit(_ptr(leaves), n, w, cap_height, _ptr(digests), _ptr(cap))
        if rc < 0:
            raise ValueError("bad merkle_cap_commit shape")
        return digests[:n_dig], cap

    def polynomial_batch_commit(self, polys, is_values, rate_bits=3, cap_height=4):
        polys = _arr(polys)
        n_polys, n = polys.shape
        log_n = n.bit_length() - 1
        big = n << rate_bits
        k = log_n + rate_bits
        leaves = np.zeros((big, n_polys), np.uint64)
        n_dig = sum(big >> j for j in range(k - cap_height))
        digests = np.zeros((max(n_dig, 1), 4), np.uint64)
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        rc = self.lib.oracle_polynomial_batch_commit(_ptr(polys), int(bool(is_values)), n_polys, log_n, rate_bits,
                                                     cap_height, _ptr(leaves), _ptr(digests), _ptr(cap))
        if rc < 0:
            raise ValueError("bad polynomial_batch_commit shape")
        return leaves, digests[:n_dig], cap

    # ---- permutation argument (oracle/plonk.c)
    def permutation_partial_products(self, wires, sigmas, k_is, betas, gammas, chunk=8):
        """-> (zs (num_challenges, n), partial_products (num_challenges, num_prods, n))"""
        wires, sigmas, k_is = _arr(wires), _arr(sigmas), _arr(k_is)
        betas, gammas = _arr(betas).reshape(-1), _arr(gammas).reshape(-1)
        num_routed, n = wires.shape
        nc = betas.size
        num_prods = (num_routed + chunk - 1) // chunk - 1
        out = np.zeros((nc * (1 + num_prods), n), np.uint64)
        rc = self.lib.oracle_permutation_partial_products(_ptr(wires), _ptr(sigmas), _ptr(k_is), _ptr(betas), _ptr(gammas),
                                                          nc, num_routed, n.bit_length() - 1, chunk, _ptr(out))
        if rc != 0:
            raise ValueError("oracle_permutation_partial_products: status %d" % rc)
        return out[:nc], out[nc:].reshape(nc, num_prods, n)

    # ---- gate constraints / quotient / opening check (oracle/plonk.c)
    def ifft_rows(self, a):
        return np.stack([self.ifft(r) for r in _arr(a)])

    def plonk_desc(self, degree_bits, num_wires, num_routed, num_constants, num_selectors, num_challenges,
                   quotient_degree_factor, gate_kinds, gate_selectors, gate_groups):
        d = PlonkDesc(degree_bits, num_wires, num_routed, num_constants, num_selectors, num_challenges,
                      quotient_degree_factor, len(gate_kinds))
        for i, (k, s, (gs, ge)) in enumerate(zip(gate_kinds, gate_selectors, gate_groups)):
            d.gate_kind[i], d.gate_selector[i], d.group_start[i], d.group_end[i] = k, s, gs, ge
        return d

    def gate_constraints_row(self, kind, wires_row, consts, pi_hash):
        """unfiltered constraints of gate `kind` on one row of wires -> array of constraint values (all zero on a valid row)"""
        w, c, ph, out = _arr(wires_row), _arr(consts), _arr(pi_hash), np.zeros(123, np.uint64)
        n = self.lib.oracle_gate_constraints_row(kind, _ptr(w), _ptr(c), _ptr(ph), _ptr(out))
        assert n >= 0
        return out[:n]

    def plonk_quotient_polys(self, desc, k_is, cs_leaves, wires_leaves, zs_leaves, pi_hash, betas, gammas, alphas):
        """compute_quotient_polys -> (num_challenges, quotient_degree_factor << degree_bits) coefficients"""
        args = [_arr(x) for x in (k_is, cs_leaves, wires_leaves, zs_leaves, pi_hash, betas, gammas, alphas)]
        out = np.zeros((desc.num_challenges, desc.quotient_degree_factor << desc.degree_bits), np.uint64)
        rc = self.lib.oracle_plonk_quotient_polys(C.byref(desc), *[_ptr(a) for a in args], _ptr(out))
        if rc != 0:
            raise ValueError("oracle_plonk_quotient_polys: status %d" % rc)
        return out

    def plonk_check_openings(self, desc, k_is, zeta, constants, sigmas, wires, zs, next_zs, pps, quotient, pi_hash, betas,
                             gammas, alphas):
        args = [_arr(x) for x in (k_is, zeta, constants, sigmas, wires, zs, next_zs, pps, quotient, pi_hash, betas, gammas,
                                  alphas)]
        return bool(self.lib.oracle_plonk_check_openings(C.byref(desc), *[_ptr(a) for a in args]))

    # ---- challenger / FRI (oracle/fri.c)
    def challenger(self):
        return OracleChallenger(self)

    def ext_mul(self, x, y):
        x, y, out = _arr(x), _arr(y), np.zeros(2, np.uint64)
        self.lib.oracle_ext_mul(_ptr(x), _ptr(y), _ptr(out))
        return out

    def ext_inv(self, x):
        x, out = _arr(x), np.zeros(2, np.uint64)
        self.lib.oracle_ext_inv(_ptr(x), _ptr(out))
        return out

    def fri_params_standard(self, degree_bits, **override):
        p = FriParams()
        self.lib.oracle_fri_params_standard(degree_bits, C.byref(p))
        for k, v in override.items():
            if k == "reduction_arity_bits":
                p.num_reductions = len(v)
                for i, a in enumerate(v):
                    p.reduction_arity_bits[i] = a
            else:
                setattr(p, k, v)
        return p

    def fri_proof_len(self, params, n_polys):
        n_polys = _arr(n_polys)
        return self.lib.oracle_fri_proof_len(C.byref(params), n_polys.size, _ptr(n_polys))

    def eval_polys_ext(self, coeffs, point):
        coeffs, point = _arr(coeffs), _arr(point)
        n_polys, n = coeffs.shape
        out = np.zeros((n_polys, 2), np.uint64)
        self.lib.oracle_eval_polys_ext(_ptr(coeffs), n_polys, n.bit_length() - 1, _ptr(point), _ptr(out))
        return out

    def fri_prove(self, oracles, batches, params, challenger):
        """oracles: [(coeffs (n_polys, n), leaves (N, n_polys), digests (nd, 4)), ...]"""
        arr = (FriOracle * len(oracles))()
        keep = []
        for i, (coeffs, leaves, digests) in enumerate(oracles):
            c, l, d = _arr(coeffs), _arr(leaves), _arr(digests)
            keep += [c, l, d]
            arr[i].coeffs, arr[i].leaves, arr[i].digests, arr[i].n_polys = _ptr(c), _ptr(l), _ptr(d), c.shape[0]
        barr, keep2 = make_fri_batches(FriBatch, batches)
        proof = np.zeros(self.fri_proof_len(params, [o[0].shape[0] for o in oracles]), np.uint64)
        rc = self.lib.oracle_fri_prove(arr, len(oracles), barr, len(batches), C.byref(params), C.byref(challenger.st),
                                       _ptr(proof))
        if rc != 0:
            raise ValueError("oracle_fri_prove: status %d" % rc)
        return proof

    def fri_verify(self, n_polys, caps, batches, openings, params, challenger, proof):
        """-> (accepted: bool, reason: int)"""
        n_polys, caps, proof = _arr(n_polys), _arr(caps), _arr(proof)
        openings = _arr(np.concatenate([np.asarray(o, np.uint64).reshape(-1) for o in openings]))
        barr, keep = make_fri_batches(FriBatch, batches)
        reason = C.c_int(0)
        rc = self.lib.oracle_fri_verify(_ptr(n_polys), n_polys.size, _ptr(caps), barr, len(batches), _ptr(openings),
                                        C.byref(params), C.byref(challenger.st), _ptr(proof), C.byref(reason))
        if rc < 0:
            raise ValueError("oracle_fri_verify: malformed arguments")
        return bool(rc), reason.value


class OracleChallenger:
    def __init__(self, oracle):
        self.o = oracle
        self.st = ChallengerState()
        oracle.lib.oracle_challenger_init(C.byref(self.st))

    def clone(self):
        c = OracleChallenger(self.o)
        C.memmove(C.byref(c.st), C.byref(self.st), C.sizeof(ChallengerState))
        return c

    def observe(self, elements):
        e = _arr(elements).reshape(-1)
        self.o.lib.oracle_challenger_observe(C.byref(self.st), _ptr(e), e.size)

    def get_challenge(self):
        return self.o.lib.oracle_challenger_get(C.byref(self.st))

    def get_n_challenges(self, n):
        return np.array([self.get_challenge() for _ in range(n)], np.uint64)


class OracleMMR:
    def __init__(self, oracle, leaves=None):
        self.o = oracle
        self.h = oracle.lib.oracle_mmr_new()
        if leaves is not None:
            self.add_leaves(leaves)

    def __del__(self):
        try:
            self.o.lib.oracle_mmr_free(self.h)
        except Exception:
            pass

    def add_leaf(self, leaf): self.o.lib.oracle_mmr_add_leaf(self.h, int(leaf))

    def add_leaves(self, leaves):
        leaves = _arr(leaves)
        self.o.lib.oracle_mmr_add_leaves(self.h, _ptr(leaves), leaves.size)

    def __len__(self): return self.o.lib.oracle_mmr_len(self.h)

    @property
    def elements(self):
        n = len(self)
        if n == 0:
            return np.zeros((0, 4), np.uint64)
        p = self.o.lib.oracle_mmr_elements(self.h)
        return np.ctypeslib.as_array(p, shape=(n * 4,)).reshape(n, 4).copy()

    def get_peaks(self):
        out = np.zeros((64, 4), np.uint64)
        n = self.o.lib.oracle_mmr_get_peaks(self.h, _ptr(out))
        if n < 0:
            raise OverflowError("reference panics in get_peaks (empty MMR or len >= 2^32)")
        return out[:n].copy()

    def bagging_the_peaks(self):
        out = np.zeros(4, np.uint64)
        if self.o.lib.oracle_mmr_bagging_the_peaks(self.h, _ptr(out)) < 0:
            raise OverflowError("reference panics in get_peaks (empty MMR or len >= 2^32)")
        return out

    def get_proof(self, mmr_index):
        sib = np.zeros((64, 4), np.uint64)
        lefts = np.zeros(64, np.uint8)
        peaks = np.zeros((64, 4), np.uint64)
        ns, npk, sz = C.c_int(0), C.c_int(0), C.c_size_t(0)
        rc = self.o.lib.oracle_mmr_get_proof(self.h, mmr_index, _ptr(sib), lefts.ctypes.data_as(_u8p), C.byref(ns),
                                             _ptr(peaks), C.byref(npk), C.byref(sz))
        if rc < 0:
            raise IndexError("reference panics: index out of bounds")
        return {"mmr_size": sz.value, "siblings": sib[:ns.value].copy(), "lefts": lefts[:ns.value].copy(),
                "peaks": peaks[:npk.value].copy()}

    def get_proof_normal_index(self, normal_index):
        return self.get_proof(self.o.get_mmr_index(normal_index))
