"""Host-side mirror of the plonky2 surface the reference's verifier circuits use (mmr_plonky2_verifier.rs:13-151):
CircuitBuilder, CircuitData {prove}, PartialWitness -- over the C ABI (include/p2mt.h).  A Target is an opaque integer
handle; a HashOutTarget is a list of 4, a BoolTarget a Target."""
import ctypes as C

import numpy as np

from . import _native as N


class CircuitInfo(C.Structure):
    _fields_ = [("degree_bits", C.c_uint32), ("num_gate_types", C.c_uint32), ("num_selectors", C.c_uint32),
                ("num_constants_sigmas", C.c_uint32), ("num_public_inputs", C.c_uint32),
                ("num_partial_products", C.c_uint32), ("gate_counts", C.c_uint32 * 16), ("gate_kinds", C.c_uint32 * 16),
                ("gate_selector", C.c_uint32 * 16), ("group_start", C.c_uint32 * 16), ("group_end", C.c_uint32 * 16),
                ("proof_len", C.c_uint64), ("fri_proof_len", C.c_uint64)]


def _targets(ts):
    return np.ascontiguousarray(np.asarray([int(t) for t in ts], dtype=np.uint64))


class CircuitBuilder:
    """CircuitBuilder::<GoldilocksField, 2>::new(CircuitConfig::standard_recursion_config())."""

    def __init__(self):
        self._h = C.c_void_p()
        N.check(N.lib().p2mt_cb_create(C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().p2mt_cb_destroy(h)

    def _out(self, fn, *args):
        t = C.c_uint64()
        N.check(fn(self._h, *args, C.byref(t)))
        return t.value

    def add_virtual_target(self):
        return self._out(N.lib().p2mt_cb_add_virtual_target)

    def add_virtual_hash(self):
        return [self.add_virtual_target() for _ in range(4)]

    def add_virtual_bool_target_safe(self):
        return self._out(N.lib().p2mt_cb_add_virtual_bool_target_safe)

    def constant(self, c):
        return self._out(N.lib().p2mt_cb_constant, int(c))

    def zero(self):
        return self.constant(0)

    def one(self):
        return self.constant(1)

    def connect(self, x, y):
        N.check(N.lib().p2mt_cb_connect(self._h, x, y))

    def arithmetic(self, const_0, const_1, multiplicand_0, multiplicand_1, addend):
        return self._out(N.lib().p2mt_cb_arithmetic, int(const_0), int(const_1), multiplicand_0, multiplicand_1, addend)

    def add(self, x, y):
        return self._out(N.lib().p2mt_cb_add, x, y)

    def sub(self, x, y):
        return self._out(N.lib().p2mt_cb_sub, x, y)

    def mul(self, x, y):
        return self._out(N.lib().p2mt_cb_mul, x, y)

    def mul_add(self, x, y, z):
        return self._out(N.lib().p2mt_cb_mul_add, x, y, z)

    def mul_sub(self, x, y, z):
        return self._out(N.lib().p2mt_cb_mul_sub, x, y, z)

    def not_(self, b):
        return self._out(N.lib().p2mt_cb_not, b)

    def or_(self, b1, b2):
        return self._out(N.lib().p2mt_cb_or, b1, b2)

    def assert_bool(self, b):
        N.check(N.lib().p2mt_cb_assert_bool(self._h, b))

    def is_equal(self, x, y):
        return self._out(N.lib().p2mt_cb_is_equal, x, y)

    def _hash(self, fn, inputs):
        ins, out = _targets(inputs), np.zeros(4, np.uint64)
        N.check(fn(self._h, N.ptr(ins), ins.size, N.ptr(out)))
        return [int(t) for t in out]

    def hash_n_to_hash_no_pad(self, inputs):
        return self._hash(N.lib().p2mt_cb_hash_n_to_hash_no_pad, inputs)

    def hash_or_noop(self, inputs):
        return self._hash(N.lib().p2mt_cb_hash_or_noop, inputs)

    def register_public_inputs(self, targets):
        ts = _targets(targets)
        N.check(N.lib().p2mt_cb_register_public_inputs(self._h, N.ptr(ts), ts.size))

    def register_public_input(self, target):
        self.register_public_inputs([target])

    def num_gates(self):
        return N.lib().p2mt_cb_num_gates(self._h)

    # ---- recursion (mmr_plonky2_verifier_1_recursion.rs:95-104): plonky2's in-circuit verifier
    def add_virtual_proof_with_pis(self, inner_circuit_data):
        """builder.add_virtual_proof_with_pis(&inner.common) -> ProofWithPublicInputsTarget"""
        n = inner_circuit_data.info.proof_len
        out = np.zeros(n, np.uint64)
        N.check(N.lib().p2mt_cb_add_virtual_proof_with_pis(self._h, inner_circuit_data._h, N.ptr(out), n))
        return ProofWithPublicInputsTarget([int(t) for t in out], inner_circuit_data.info.num_public_inputs)

    def add_virtual_verifier_data(self, cap_height):
        """builder.add_virtual_verifier_data(cap_height) -> VerifierCircuitTarget"""
        out = np.zeros(68, np.uint64)
        N.check(N.lib().p2mt_cb_add_virtual_verifier_data(self._h, cap_height, N.ptr(out)))
        return VerifierCircuitTarget([int(t) for t in out])

    def verify_proof(self, proof_with_pis, inner_verifier_data, inner_circuit_data):
        """builder.verify_proof::<PoseidonGoldilocksConfig>(&proof_with_pis, &inner_verifier_data, &inner.common)"""
        pt, vd = proof_with_pis._array, inner_verifier_data._array
        N.check(N.lib().p2mt_cb_verify_proof(self._h, N.ptr(pt), pt.size, N.ptr(vd), inner_circuit_data._h))

    def build(self):
        """builder.build::<PoseidonGoldilocksConfig>()"""
        h = C.c_void_p()
        N.check(N.lib().p2mt_cb_build(self._h, C.byref(h)))
        return CircuitData(h)


class ProofWithPublicInputsTarget:
    """One target per word of a proof of the inner circuit (the order CircuitData.prove writes); public_inputs = the last ones."""

    def __init__(self, targets, num_public_inputs):
        self.targets = targets
        self.public_inputs = targets[len(targets) - num_public_inputs:]
        self._array = _targets(targets)   # the same handles as one contiguous u64 array (what the C ABI takes)


class VerifierCircuitTarget:
    """constants_sigmas_cap (16 HashOutTargets) and circuit_digest (1 HashOutTarget), flat"""

    def __init__(self, targets):
        self.targets = targets
        self.constants_sigmas_cap = [targets[4 * i:4 * i + 4] for i in range(16)]
        self.circuit_digest = targets[64:68]
        self._array = _targets(targets)


class PartialWitness:
    def __init__(self):
        self._h = C.c_void_p()
        N.check(N.lib().p2mt_pw_create(C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().p2mt_pw_destroy(h)

    def set_target(self, target, value):
        N.check(N.lib().p2mt_pw_set_target(self._h, target, int(value)))

    def clear(self):
        """forget every assignment (the handle is reused for the next statement)"""
        N.check(N.lib().p2mt_pw_clear(self._h))

    def set_bool_target(self, target, value):
        self.set_target(target, 1 if value else 0)

    def set_hash_target(self, targets, value):
        for t, v in zip(targets, value):
            self.set_target(t, v)

    def set_proof_with_pis_target(self, proof_target, proof):
        """pw.set_proof_with_pis_target(&target, &proof) (mmr_plonky2_verifier_1_recursion.rs:201)"""
        pt, words = proof_target._array, N.as_u64(proof).reshape(-1)
        assert pt.size == words.size, "proof does not match its target"
        N.check(N.lib().p2mt_pw_set_proof_with_pis_target(self._h, N.ptr(pt), N.ptr(words), words.size))

    def set_verifier_data_target(self, verifier_data_target, inner_circuit_data):
        """pw.set_verifier_data_target(&target, &inner.verifier_only) (:202)"""
        vd = verifier_data_target._array
        N.check(N.lib().p2mt_pw_set_verifier_data_target(self._h, N.ptr(vd), inner_circuit_data._h))


class _ProverOnly:
    def __init__(self, public_inputs):
        self.public_inputs = public_inputs


class CircuitData:
    """CircuitData<GoldilocksField, PoseidonGoldilocksConfig, 2>; the prover data lives in device memory."""

    def __init__(self, handle):
        self._h = handle
        self.info = CircuitInfo()
        N.check(N.lib().p2mt_circuit_get_info(self._h, C.addressof(self.info)))
        pis = np.zeros(max(self.info.num_public_inputs, 1), np.uint64)
        N.check(N.lib().p2mt_circuit_public_inputs(self._h, N.ptr(pis)))
        self.prover_only = _ProverOnly([int(t) for t in pis[:self.info.num_public_inputs]])
        self.degree_bits = self.info.degree_bits
        self.common = self          # `inner_circuit_data.common` / `.verifier_only`: the handle itself carries both
        self.verifier_only = self

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().p2mt_circuit_destroy(h)

    def constants_sigmas(self):
        """-> (values (num_constants_sigmas, n), cap (16, 4), circuit_digest (4,))"""
        vals = np.zeros((self.info.num_constants_sigmas, 1 << self.degree_bits), np.uint64)
        cap, digest = np.zeros((16, 4), np.uint64), np.zeros(4, np.uint64)
        N.check(N.lib().p2mt_circuit_constants_sigmas(self._h, N.ptr(vals), N.ptr(cap), N.ptr(digest)))
        return vals, cap, digest

    def generate_witness(self, pw):
        wires = np.zeros((135, 1 << self.degree_bits), np.uint64)
        N.check(N.lib().p2mt_circuit_generate_witness(self._h, pw._h, N.ptr(wires)))
        return wires

    def prove(self, pw):
        """circuit_data.prove(pw) -> ProofWithPublicInputs as words (layout: include/p2mt.h)."""
        proof = np.zeros(self.info.proof_len, np.uint64)
        N.check(N.lib().p2mt_circuit_prove(self._h, pw._h, N.ptr(proof), proof.size))
        return proof

    def proof_to_bytes(self, proof):
        """ProofWithPublicInputs::to_bytes() in plonky2's Buffer order (one length byte in front of every Merkle path)"""
        p = N.as_u64(proof).reshape(-1)
        out = np.zeros(N.lib().p2mt_proof_bytes_len(self._h), np.uint8)
        N.check(N.lib().p2mt_proof_to_bytes(self._h, N.ptr(p), p.size, out.ctypes.data_as(C.c_void_p), out.size))
        return out.tobytes()

    def proof_from_bytes(self, data):
        """ProofWithPublicInputs::from_bytes(bytes, &common_data) -> proof words"""
        b = np.frombuffer(bytes(data), np.uint8)
        out = np.zeros(self.info.proof_len, np.uint64)
        N.check(N.lib().p2mt_proof_from_bytes(self._h, b.ctypes.data_as(C.c_void_p), b.size, N.ptr(out), out.size))
        return out

    def verify(self, proof, with_reason=False):
        """circuit_data.verify(proof): raises P2mtPanic (plonky2 returns Err) unless the proof is accepted; with_reason=True
        returns (accepted, reason) instead (reasons: include/p2mt.h)."""
        p = N.as_u64(proof).reshape(-1)
        acc, reason = C.c_int(0), C.c_int(0)
        N.check(N.lib().p2mt_circuit_verify(self._h, N.ptr(p), p.size, C.byref(acc), C.byref(reason)))
        if with_reason:
            return bool(acc.value), reason.value
        if not acc.value:
            raise N.P2mtPanic(N.P2MT_EINVAL, "proof rejected (reason %d)" % reason.value)
        return True

    def verify_batch(self, proofs):
        """circuit_data.verify for many proofs in passes of up to 256 (p2mt_circuit_verify_batch) -> (accepted, reasons) lists"""
        p = np.ascontiguousarray(N.as_u64(proofs).reshape(-1, self.info.proof_len))
        n = p.shape[0]
        acc, reason = (C.c_int * n)(), (C.c_int * n)()
        N.check(N.lib().p2mt_circuit_verify_batch(self._h, N.ptr(p), n, p.shape[1], acc, reason))
        return [bool(a) for a in acc], list(reason)

    def prove_trace(self):
        n = 1 << self.degree_bits
        out = {}
        for key, what, shape in (("wires", 0, (135, n)), ("zs_pp", 1, (20, n)), ("quotient_chunks", 2, (16, n)),
                                 ("challenges", 3, (8,)), ("pi_hash", 4, (4,))):
            a = np.zeros(shape, np.uint64)
            N.check(N.lib().p2mt_circuit_prove_trace(self._h, what, N.ptr(a)))
            out[key] = a
        return out


class BatchProver:
    """Up to `batch` proofs of one circuit per pass of the prover pipeline (p2mt_batch_prover): the proof index rides in a
    grid dimension of every launch.  Borrows `circuit` (keep it alive; one thread at a time).  prove(witnesses) ->
    (len(witnesses), proof_len) proof words, bit-identical to circuit.prove(w) for each witness."""

    def __init__(self, circuit, batch):
        self.circuit = circuit
        h = C.c_void_p()
        N.check(N.lib().p2mt_batch_prover_create(circuit._h, int(batch), C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            N.lib().p2mt_batch_prover_destroy(h)

    @property
    def batch(self):
        return int(N.lib().p2mt_batch_prover_batch(self._h))

    def prove(self, witnesses, status=False):
        n = len(witnesses)
        plen = self.circuit.info.proof_len
        out = np.zeros((n, plen), np.uint64)
        warr = (C.c_void_p * n)(*[w._h for w in witnesses])
        st = (C.c_int * n)()
        rc = N.lib().p2mt_batch_prover_prove(self._h, warr, n, N.ptr(out), plen, st)
        if status:
            return out, rc, list(st)
        N.check(rc)
        return out


def prove_many(circuits, witnesses):
    """Independent proves spread over len(circuits) worker threads inside the library (one per handle, each on its own
    stream).  circuits: distinct builds of the same circuit; witnesses: PartialWitness objects.  -> (len(witnesses), proof_len)."""
    n, k = len(witnesses), len(circuits)
    plen = circuits[0].info.proof_len
    out = np.zeros((n, plen), np.uint64)
    carr = (C.c_void_p * k)(*[c._h for c in circuits])
    warr = (C.c_void_p * n)(*[w._h for w in witnesses])
    status = (C.c_int * n)()
    N.check(N.lib().p2mt_circuit_prove_many(carr, k, warr, n, N.ptr(out), plen, status))
    return out
