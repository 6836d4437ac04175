"""Host-side mirror of the opening proof that ends CircuitData::prove (mmr_plonky2_verifier.rs:148,
mmr_plonky2_verifier_1_recursion.rs:192,218): plonky2's Challenger (iop/challenger.rs), OpeningSet evaluation
(plonk/proof.rs) and PolynomialBatch::prove_openings -> fri_proof (fri/oracle.rs, fri/prover.rs), over the C ABI
(include/p2mt.h).  Extension elements are (a, b) pairs of u64: a + bX in F[X]/(X^2 - 7)."""
import ctypes as C

import numpy as np

from . import _native as N


class FriParams(C.Structure):
    """fri/mod.rs FriParams (hiding = false)."""
    _fields_ = [("degree_bits", C.c_uint32), ("rate_bits", C.c_uint32), ("cap_height", C.c_uint32),
                ("proof_of_work_bits", C.c_uint32), ("num_query_rounds", C.c_uint32), ("num_reductions", C.c_uint32),
                ("reduction_arity_bits", C.c_uint32 * 8)]

    @staticmethod
    def standard(degree_bits, **override):
        """CircuitConfig::standard_recursion_config().fri_config.fri_params(degree_bits, false)."""
        p = FriParams()
        N.check(N.lib().p2mt_fri_params_standard(degree_bits, C.addressof(p)))
        for k, v in override.items():
            if k == "reduction_arity_bits":
                p.num_reductions = len(v)
                for i in range(8):
                    p.reduction_arity_bits[i] = v[i] if i < len(v) else 0
            else:
                setattr(p, k, v)
        return p

    def arity_bits(self):
        return [self.reduction_arity_bits[i] for i in range(self.num_reductions)]


class _FriOracle(C.Structure):
    _fields_ = [("coeffs", N.u64p), ("leaves", N.u64p), ("digests", N.u64p), ("n_polys", C.c_uint64)]


class _FriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("polys", C.POINTER(C.c_uint32)), ("n_polys", C.c_uint64)]


class Challenger:
    """plonky2 Challenger<F, PoseidonHash>; the sponge lives in device memory."""

    def __init__(self, _handle=None):
        if _handle is None:
            _handle = C.c_void_p()
            N.check(N.lib().p2mt_challenger_create(C.byref(_handle)))
        self._h = _handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().p2mt_challenger_destroy(h)

    def clone(self):
        h = C.c_void_p()
        N.check(N.lib().p2mt_challenger_clone(self._h, C.byref(h)))
        return Challenger(h)

    def observe_elements(self, elements):
        e = N.as_u64(elements).reshape(-1)
        N.check(N.lib().p2mt_challenger_observe(self._h, N.ptr(e), e.size))

    observe_element = observe_hash = observe_cap = observe_extension_elements = observe_elements

    def get_n_challenges(self, n):
        out = np.zeros(n, np.uint64)
        N.check(N.lib().p2mt_challenger_get_challenges(self._h, n, N.ptr(out)))
        return out

    def get_challenge(self):
        return int(self.get_n_challenges(1)[0])

    def get_extension_challenge(self):
        return self.get_n_challenges(2)

    def state(self):
        out = np.zeros(30, np.uint64)
        N.check(N.lib().p2mt_challenger_get_state(self._h, N.ptr(out)))
        return out

    def set_state(self, words):
        w = N.as_u64(words).reshape(30)
        N.check(N.lib().p2mt_challenger_set_state(self._h, N.ptr(w)))


def eval_polys_ext(coeffs, point):
    """PolynomialCoeffs::eval of every row at one extension point -> (n_polys, 2)."""
    c = N.as_u64(coeffs)
    c = c.reshape(-1, c.shape[-1])
    n = c.shape[1]
    if n <= 0 or n & (n - 1):
        raise N.P2mtPanic(N.P2MT_EINVAL, "polynomial length must be a power of two")
    pt = N.as_u64(point).reshape(2)
    out = np.zeros((c.shape[0], 2), np.uint64)
    N.check(N.lib().p2mt_eval_polys_ext(N.ptr(c), c.shape[0], n.bit_length() - 1, N.ptr(pt), N.ptr(out)))
    return out


def fri_proof_len(params, n_polys):
    a = N.as_u64(n_polys).reshape(-1)
    return N.lib().p2mt_fri_proof_len(C.addressof(params), a.size, N.ptr(a))


def _batch_array(batches, keep):
    barr = (_FriBatch * len(batches))()
    for i, (point, polys) in enumerate(batches):
        pl = np.ascontiguousarray(np.asarray(polys, dtype=np.uint32).reshape(-1, 2))
        keep.append(pl)
        barr[i].point[0], barr[i].point[1] = int(point[0]), int(point[1])
        barr[i].polys = pl.ctypes.data_as(C.POINTER(C.c_uint32))
        barr[i].n_polys = pl.shape[0]
    return barr


def openings(batches, oracles):
    """OpeningSet::to_fri_openings: per batch an (n_polys, 2) array of the polynomials' values at the batch's point."""
    arr = (_FriOracle * len(oracles))()
    keep = []
    degree_bits = None
    for i, o in enumerate(oracles):
        c = N.as_u64(o.polynomials)
        keep.append(c)
        arr[i].coeffs, arr[i].n_polys = c.ctypes.data_as(N.u64p), o.n_polys
        degree_bits = o.degree_log
    barr = _batch_array(batches, keep)
    total = sum(len(pl) for _, pl in batches)
    out = np.zeros((total, 2), np.uint64)
    N.check(N.lib().p2mt_fri_openings(C.addressof(arr), len(oracles), C.addressof(barr), len(batches), degree_bits,
                                      N.ptr(out)))
    res, off = [], 0
    for _, pl in batches:
        res.append(out[off:off + len(pl)])
        off += len(pl)
    return res


def prove_openings(batches, oracles, challenger, params):
    """PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params).

    batches: FriInstanceInfo.batches as [(point (2,), [(oracle index, polynomial index), ...]), ...];
    oracles: commit.PolynomialBatch objects (with .polynomials and leaves).  Returns the FriProof words
    (layout: include/p2mt.h)."""
    arr = (_FriOracle * len(oracles))()
    keep = []
    for i, o in enumerate(oracles):
        t = o.merkle_tree
        if o.polynomials is None or t.leaves is None:
            raise N.P2mtPanic(N.P2MT_EINVAL, "prove_openings needs the batch's coefficients and leaves")
        c, l = N.as_u64(o.polynomials), N.as_u64(t.leaves)
        d = N.as_u64(t.digests) if len(t.digests) else np.zeros((1, 4), np.uint64)
        keep += [c, l, d]
        arr[i].coeffs, arr[i].leaves, arr[i].digests = (x.ctypes.data_as(N.u64p) for x in (c, l, d))
        arr[i].n_polys = o.n_polys
    barr = _batch_array(batches, keep)
    total = fri_proof_len(params, [o.n_polys for o in oracles])
    if total == 0:
        raise N.P2mtPanic(N.P2MT_EINVAL, "unsupported FriParams")
    proof = np.zeros(total, np.uint64)
    N.check(N.lib().p2mt_fri_prove_openings(C.addressof(arr), len(oracles), C.addressof(barr), len(batches),
                                            C.addressof(params), challenger._h, N.ptr(proof)))
    return proof
