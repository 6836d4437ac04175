"""ctypes binding of libp2mt_hip.so -- the C ABI declared in include/p2mt.h.

There is NO fallback: if the shared library is missing this module raises at import of `lib()`,
and every compute entry point fails with P2MT_EHIP when no HIP device is present.
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# (P2MT_LIB_PATH: the sanitizer leg points the binding at its own host-only AddressSanitizer build of the same sources,
#  csrc/Makefile `asan`; there is still no fallback -- the named library is the one that must load)
LIB_PATH = os.environ.get("P2MT_LIB_PATH") or os.path.join(PKG_DIR, "libp2mt_hip.so")

P2MT_OK, P2MT_EINVAL, P2MT_ENOMEM, P2MT_EHIP, P2MT_ERANGE, P2MT_ENOTPEAK = 0, -1, -2, -3, -4, -5
MAX_PROOF_LEN = 64

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)
i8p = C.POINTER(C.c_int8)
i32p = C.POINTER(C.c_int32)
intp = C.POINTER(C.c_int)
sizep = C.POINTER(C.c_size_t)
voidp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/p2mt.h (checked by tests/test_abi_cpu.py)
SIGNATURES = {
    "p2mt_init": (C.c_int, [C.c_int]),
    "p2mt_device_count": (C.c_int, []),
    "p2mt_set_stream": (C.c_int, [voidp]),
    "p2mt_get_stream": (C.c_int, [C.POINTER(voidp)]),
    "p2mt_thread_stream_create": (C.c_int, []),
    "p2mt_thread_stream_destroy": (C.c_int, []),
    "p2mt_set_throughput_mode": (C.c_int, [C.c_int]),
    "p2mt_sync": (C.c_int, []),
    "p2mt_last_error": (C.c_char_p, []),
    "p2mt_set_variant": (C.c_int, [C.c_int, C.c_int]),
    "p2mt_get_variant": (C.c_int, [intp, intp]),
    "p2mt_get_build_config": (C.c_int, [intp, intp, intp]),
    "p2mt_mmr_stage1_levels": (C.c_int, [C.c_size_t]),
    "p2mt_debug_force_fallback": (C.c_int, [C.c_int]),
    "p2mt_debug_fail_allocs": (C.c_int, [C.c_int]),
    "p2mt_debug_field_op": (C.c_int, [C.c_int, voidp, voidp, C.c_size_t, voidp, voidp]),
    "p2mt_debug_partial_group": (C.c_int, [C.c_int, voidp, C.c_size_t, voidp, voidp]),
    "p2mt_host_poseidon_permute": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_debug_host_transcript": (C.c_int, [C.c_int]),
    "p2mt_debug_host_chain": (C.c_int, [C.c_int]),
    "p2mt_circuit_schedule_info": (C.c_int, [voidp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "p2mt_debug_host_challenger": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp]),
    "p2mt_debug_plan_knobs": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "p2mt_debug_plan_profile": (C.c_int, [C.c_int]),
    "p2mt_debug_plan_profile_read": (C.c_int64, [voidp, C.c_size_t]),
    "p2mt_profile_enable": (C.c_int, [C.c_int]),
    "p2mt_profile_read": (C.c_int, [C.POINTER(C.c_float), intp]),
    "p2mt_timer_start": (C.c_int, []),
    "p2mt_timer_stop": (C.c_int, [C.POINTER(C.c_float)]),
    "p2mt_poseidon_permute_batch": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_poseidon_permute_batch_dev": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_two_to_one_batch": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_two_to_one_batch_dev": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_hash_or_noop_batch": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_hash_or_noop_batch_dev": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_hash_no_pad_batch": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_hash_no_pad_batch_dev": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_poseidon_gate_witness_batch": (C.c_int, [voidp, voidp, C.c_size_t, voidp]),
    "p2mt_poseidon_gate_witness_batch_dev": (C.c_int, [voidp, voidp, C.c_size_t, voidp]),
    "p2mt_merkle_build_pow2": (C.c_int, [voidp, C.c_size_t, voidp, voidp]),
    "p2mt_merkle_build_pow2_dev": (C.c_int, [voidp, C.c_size_t, voidp, voidp]),
    "p2mt_merkle_get_proof": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_merkle_get_in_between_hashes": (C.c_int, [voidp, voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_verify_merkle_proof_batch": (C.c_int, [voidp, voidp, voidp, voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_get_heights_bitmap_for_mmr_size": (C.c_uint64, [C.c_size_t, sizep]),
    "p2mt_get_mmr_index": (C.c_int64, [C.c_size_t]),
    "p2mt_mmr_create": (C.c_int, [C.POINTER(voidp)]),
    "p2mt_mmr_destroy": (C.c_int, [voidp]),
    "p2mt_mmr_reserve": (C.c_int, [voidp, C.c_size_t]),
    "p2mt_mmr_reset": (C.c_int, [voidp]),
    "p2mt_mmr_add_leaf": (C.c_int, [voidp, C.c_uint64]),
    "p2mt_mmr_flush": (C.c_int, [voidp]),
    "p2mt_mmr_extend": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_mmr_extend_dev": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_mmr_num_leaves": (C.c_size_t, [voidp]),
    "p2mt_mmr_len": (C.c_size_t, [voidp]),
    "p2mt_mmr_elements_dev": (voidp, [voidp]),
    "p2mt_mmr_copy_elements": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_host_alloc_pinned": (C.c_int, [C.c_size_t, C.POINTER(voidp)]),
    "p2mt_host_free_pinned": (C.c_int, [voidp]),
    "p2mt_mmr_copy_elements_async": (C.c_int, [voidp, C.c_size_t, C.c_size_t, voidp]),
    "p2mt_mmr_extend_dev_to_host": (C.c_int, [voidp, voidp, C.c_size_t, C.c_uint, voidp]),
    "p2mt_mmr_save": (C.c_int, [voidp, C.c_char_p]),
    "p2mt_mmr_load": (C.c_int, [voidp, C.c_char_p]),
    "p2mt_mmr_peaks": (C.c_int, [voidp, voidp, intp]),
    "p2mt_mmr_root": (C.c_int, [voidp, voidp]),
    "p2mt_mmr_root_dev": (C.c_int, [voidp, voidp]),
    "p2mt_mmr_proof": (C.c_int, [voidp, C.c_size_t, voidp, voidp, intp, voidp, intp, sizep]),
    "p2mt_mmr_proof_batch": (C.c_int, [voidp, voidp, C.c_size_t, C.c_size_t, voidp, voidp, voidp]),
    "p2mt_mmr_proof_batch_dev": (C.c_int, [voidp, voidp, C.c_size_t, C.c_size_t, voidp, voidp, voidp]),
    "p2mt_mmr_proof_verify_batch_dev": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp, C.c_int, voidp, voidp,
                                                  C.c_size_t, voidp]),
    "p2mt_mmr_proof_verify": (C.c_int, [voidp, voidp, C.c_int, voidp, C.c_int, C.c_uint64, voidp, intp]),
    "p2mt_mmr_proof_verify_batch": (C.c_int, [voidp, voidp, voidp, C.c_size_t, voidp, C.c_int, voidp, voidp,
                                              C.c_size_t, voidp]),
    "p2mt_mmr_combine_shard_roots": (C.c_int, [voidp, C.c_size_t, voidp, voidp]),
    "p2mt_mmr_combine_shard_roots_dev": (C.c_int, [voidp, C.c_size_t, voidp, voidp]),
    "p2mt_sharded_mmr_create": (C.c_int, [C.POINTER(voidp), C.c_size_t, C.c_int, C.c_int, voidp]),
    "p2mt_nccl_unique_id": (C.c_int, [voidp]),
    "p2mt_sharded_mmr_create_with_id": (C.c_int, [C.POINTER(voidp), C.c_size_t, C.c_int, C.c_int, voidp]),
    "p2mt_sharded_mmr_create_exchange": (C.c_int, [C.POINTER(voidp), C.c_size_t, C.c_int, C.c_int, voidp, voidp]),
    "p2mt_sharded_mmr_set_exchange": (C.c_int, [voidp, voidp, voidp]),
    "p2mt_sharded_mmr_destroy": (C.c_int, [voidp]),
    "p2mt_sharded_mmr_local": (voidp, [voidp]),
    "p2mt_sharded_mmr_build_dev": (C.c_int, [voidp, voidp]),
    "p2mt_sharded_mmr_build": (C.c_int, [voidp, voidp]),
    "p2mt_sharded_mmr_finish": (C.c_int, [voidp]),
    "p2mt_sharded_mmr_root": (C.c_int, [voidp, voidp, voidp, voidp]),
    "p2mt_sharded_mmr_proof": (C.c_int, [voidp, C.c_size_t, voidp, voidp, intp, voidp]),
    "p2mt_mmr_shard_first_pos": (C.c_size_t, [C.c_size_t, C.c_size_t]),
    "p2mt_mmr_node_pos": (C.c_size_t, [C.c_size_t, C.c_uint]),
    "p2mt_ntt_batch": (C.c_int, [voidp, C.c_uint, C.c_size_t, C.c_int]),
    "p2mt_ntt_batch_dev": (C.c_int, [voidp, C.c_uint, C.c_size_t, C.c_int]),
    "p2mt_coset_lde_batch": (C.c_int, [voidp, C.c_uint, C.c_uint, C.c_uint64, C.c_size_t, voidp]),
    "p2mt_coset_lde_batch_dev": (C.c_int, [voidp, C.c_uint, C.c_uint, C.c_uint64, C.c_size_t, voidp]),
    "p2mt_coset_lde_leaf_order_dev": (C.c_int, [voidp, C.c_uint, C.c_uint, C.c_uint64, C.c_size_t, voidp]),
    "p2mt_merkle_cap_commit": (C.c_int, [voidp, C.c_size_t, C.c_size_t, C.c_uint, voidp, voidp]),
    "p2mt_merkle_cap_commit_dev": (C.c_int, [voidp, C.c_size_t, C.c_size_t, C.c_uint, voidp, voidp]),
    "p2mt_merkle_digests_to_plonky2_layout": (C.c_int, [voidp, C.c_size_t, C.c_uint, voidp]),
    "p2mt_merkle_digests_to_plonky2_layout_dev": (C.c_int, [voidp, C.c_size_t, C.c_uint, voidp]),
    "p2mt_polynomial_batch_commit": (C.c_int, [voidp, C.c_int, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, voidp,
                                               voidp, voidp]),
    "p2mt_polynomial_batch_commit_dev": (C.c_int, [voidp, C.c_int, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, voidp,
                                                   voidp, voidp]),
    "p2mt_permutation_partial_products": (C.c_int, [voidp, voidp, voidp, voidp, voidp, C.c_size_t, C.c_size_t, C.c_uint,
                                                    C.c_uint, voidp]),
    "p2mt_permutation_partial_products_dev": (C.c_int, [voidp, voidp, voidp, voidp, voidp, C.c_size_t, C.c_size_t,
                                                        C.c_uint, C.c_uint, voidp]),
    "p2mt_challenger_create": (C.c_int, [C.POINTER(voidp)]),
    "p2mt_challenger_destroy": (C.c_int, [voidp]),
    "p2mt_challenger_clone": (C.c_int, [voidp, C.POINTER(voidp)]),
    "p2mt_challenger_observe": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_challenger_observe_dev": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_challenger_get_challenges": (C.c_int, [voidp, C.c_size_t, voidp]),
    "p2mt_challenger_get_challenges_dev": (C.c_int, [voidp, C.c_size_t, voidp]),
    "p2mt_challenger_get_state": (C.c_int, [voidp, voidp]),
    "p2mt_challenger_set_state": (C.c_int, [voidp, voidp]),
    "p2mt_challenger_reset": (C.c_int, [voidp]),
    "p2mt_challenger_duplex_dev": (C.c_int, [voidp, voidp, C.c_size_t, voidp, C.c_size_t]),
    "p2mt_challenger_restart_duplex_dev": (C.c_int, [voidp, voidp, C.c_size_t, voidp, C.c_size_t]),
    "p2mt_eval_polys_ext": (C.c_int, [voidp, C.c_size_t, C.c_uint, voidp, voidp]),
    "p2mt_eval_polys_ext_dev": (C.c_int, [voidp, C.c_size_t, C.c_uint, voidp, voidp]),
    "p2mt_fri_openings": (C.c_int, [voidp, C.c_size_t, voidp, C.c_size_t, C.c_uint, voidp]),
    "p2mt_fri_openings_dev": (C.c_int, [voidp, C.c_size_t, voidp, C.c_size_t, C.c_uint, voidp]),
    "p2mt_fri_params_standard": (C.c_int, [C.c_uint, voidp]),
    "p2mt_fri_proof_len": (C.c_size_t, [voidp, C.c_size_t, voidp]),
    "p2mt_fri_prove_openings": (C.c_int, [voidp, C.c_size_t, voidp, C.c_size_t, voidp, voidp, voidp]),
    "p2mt_fri_prove_openings_dev": (C.c_int, [voidp, C.c_size_t, voidp, C.c_size_t, voidp, voidp, voidp]),
    "p2mt_cb_create": (C.c_int, [C.POINTER(voidp)]),
    "p2mt_cb_destroy": (C.c_int, [voidp]),
    "p2mt_cb_add_virtual_target": (C.c_int, [voidp, u64p]),
    "p2mt_cb_add_virtual_bool_target_safe": (C.c_int, [voidp, u64p]),
    "p2mt_cb_constant": (C.c_int, [voidp, C.c_uint64, u64p]),
    "p2mt_cb_connect": (C.c_int, [voidp, C.c_uint64, C.c_uint64]),
    "p2mt_cb_arithmetic": (C.c_int, [voidp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_add": (C.c_int, [voidp, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_sub": (C.c_int, [voidp, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_mul": (C.c_int, [voidp, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_mul_add": (C.c_int, [voidp, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_mul_sub": (C.c_int, [voidp, C.c_uint64, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_not": (C.c_int, [voidp, C.c_uint64, u64p]),
    "p2mt_cb_or": (C.c_int, [voidp, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_assert_bool": (C.c_int, [voidp, C.c_uint64]),
    "p2mt_cb_is_equal": (C.c_int, [voidp, C.c_uint64, C.c_uint64, u64p]),
    "p2mt_cb_hash_n_to_hash_no_pad": (C.c_int, [voidp, voidp, C.c_size_t, voidp]),
    "p2mt_cb_hash_or_noop": (C.c_int, [voidp, voidp, C.c_size_t, voidp]),
    "p2mt_cb_register_public_inputs": (C.c_int, [voidp, voidp, C.c_size_t]),
    "p2mt_cb_num_gates": (C.c_size_t, [voidp]),
    "p2mt_cb_build": (C.c_int, [voidp, C.POINTER(voidp)]),
    "p2mt_circuit_destroy": (C.c_int, [voidp]),
    "p2mt_circuit_get_info": (C.c_int, [voidp, voidp]),
    "p2mt_circuit_public_inputs": (C.c_int, [voidp, voidp]),
    "p2mt_circuit_constants_sigmas": (C.c_int, [voidp, voidp, voidp, voidp]),
    "p2mt_pw_create": (C.c_int, [C.POINTER(voidp)]),
    "p2mt_pw_destroy": (C.c_int, [voidp]),
    "p2mt_pw_clear": (C.c_int, [voidp]),
    "p2mt_pw_set_target": (C.c_int, [voidp, C.c_uint64, C.c_uint64]),
    "p2mt_cb_add_virtual_proof_with_pis": (C.c_int, [voidp, voidp, voidp, C.c_size_t]),
    "p2mt_cb_add_virtual_verifier_data": (C.c_int, [voidp, C.c_uint, voidp]),
    "p2mt_cb_verify_proof": (C.c_int, [voidp, voidp, C.c_size_t, voidp, voidp]),
    "p2mt_pw_set_proof_with_pis_target": (C.c_int, [voidp, voidp, voidp, C.c_size_t]),
    "p2mt_pw_set_verifier_data_target": (C.c_int, [voidp, voidp, voidp]),
    "p2mt_circuit_generate_witness": (C.c_int, [voidp, voidp, voidp]),
    "p2mt_circuit_prove": (C.c_int, [voidp, voidp, voidp, C.c_size_t]),
    "p2mt_circuit_prove_trace": (C.c_int, [voidp, C.c_int, voidp]),
    "p2mt_circuit_verify": (C.c_int, [voidp, voidp, C.c_size_t, intp, intp]),
    "p2mt_circuit_prove_many": (C.c_int, [voidp, C.c_size_t, voidp, C.c_size_t, voidp, C.c_size_t, voidp]),
    "p2mt_debug_witness_trace": (C.c_int, [voidp, C.c_int, voidp, C.c_size_t, C.POINTER(C.c_size_t)]),
    "p2mt_circuit_verify_batch": (C.c_int, [voidp, voidp, C.c_size_t, C.c_size_t, voidp, voidp]),
    "p2mt_proof_bytes_len": (C.c_size_t, [voidp]),
    "p2mt_proof_to_bytes": (C.c_int, [voidp, voidp, C.c_size_t, voidp, C.c_size_t]),
    "p2mt_proof_from_bytes": (C.c_int, [voidp, voidp, C.c_size_t, voidp, C.c_size_t]),
    "p2mt_batch_prover_create": (C.c_int, [voidp, C.c_size_t, C.POINTER(voidp)]),
    "p2mt_batch_prover_destroy": (C.c_int, [voidp]),
    "p2mt_batch_prover_prove": (C.c_int, [voidp, voidp, C.c_size_t, voidp, C.c_size_t, voidp]),
    "p2mt_batch_prover_batch": (C.c_size_t, [voidp]),
}


class P2mtError(RuntimeError):
    """Non-zero status from the C ABI.  Where the reference panics, `code` says which panic."""

    def __init__(self, code, msg):
        super().__init__("p2mt status %d: %s" % (code, msg))
        self.code = code


class P2mtPanic(P2mtError):
    """The reference would panic here (assert!/unwrap/log2_strict)."""


_lib = None


def lib():
    """Load libp2mt_hip.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64, and a process that
        # loads /opt/rocm's copy first (through this library) and torch's afterwards ends up with two runtimes, the second of
        # which sees no GPU ("No HIP GPUs are available").  Loading torch first makes the dynamic loader resolve this library's
        # libamdhip64.so.7 to the copy already in the process.  (torch is plumbing here: device tensors, streams, distributed.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(l, name)
            f.restype = res
            f.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc == P2MT_OK:
        return
    msg = lib().p2mt_last_error().decode("utf-8", "replace")
    if rc in (P2MT_EINVAL, P2MT_ERANGE, P2MT_ENOTPEAK):
        raise P2mtPanic(rc, msg)
    raise P2mtError(rc, msg)


def as_u64(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.uint64))
    if shape is not None:
        a = a.reshape(shape)
    return a


def ptr(a):
    """void* of a numpy array, a raw integer device address, or a torch tensor (host or device)."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        return C.c_void_p(a.data_ptr())
    raise TypeError("cannot take the address of %r" % type(a))
