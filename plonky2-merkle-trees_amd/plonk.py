"""Host-side mirror of the permutation-argument stage of CircuitData::prove (mmr_plonky2_verifier.rs:148,
mmr_plonky2_verifier_1_recursion.rs:192,218): plonky2's all_wires_permutation_partial_products, over the C ABI."""
import numpy as np

from . import _native as N

QUOTIENT_DEGREE_FACTOR = 8  # standard_recursion_config: max_quotient_degree_factor
NUM_ROUTED_WIRES = 80


def coset_shifts(num_shifts):
    """get_unique_coset_shifts: k_j = MULTIPLICATIVE_GROUP_GENERATOR^j."""
    p, out, v = 0xFFFFFFFF00000001, [], 1
    for _ in range(num_shifts):
        out.append(v)
        v = v * 7 % p
    return np.array(out, dtype=np.uint64)


def all_wires_permutation_partial_products(wires, sigmas, betas, gammas, k_is=None, chunk=QUOTIENT_DEGREE_FACTOR):
    """wires, sigmas: (num_routed, n) values on the subgroup.  Returns (zs (num_challenges, n),
    partial_products (num_challenges, num_prods, n)); np.concatenate([zs, pps.reshape(-1, n)]) is the batch plonky2
    commits next."""
    wires, sigmas = N.as_u64(wires), N.as_u64(sigmas)
    num_routed, n = wires.shape
    if sigmas.shape != wires.shape or n <= 0 or n & (n - 1):
        raise N.P2mtPanic(N.P2MT_EINVAL, "wires/sigmas must be (num_routed, 2^k)")
    k_is = coset_shifts(num_routed) if k_is is None else N.as_u64(k_is).reshape(num_routed)
    betas, gammas = N.as_u64(betas).reshape(-1), N.as_u64(gammas).reshape(-1)
    if betas.size != gammas.size:
        raise N.P2mtPanic(N.P2MT_EINVAL, "one gamma per beta")
    nc = betas.size
    num_prods = (num_routed + chunk - 1) // chunk - 1 if chunk > 0 else 0
    out = np.zeros((nc * (1 + max(num_prods, 0)), n), np.uint64)
    N.check(N.lib().p2mt_permutation_partial_products(N.ptr(wires), N.ptr(sigmas), N.ptr(k_is), N.ptr(betas),
                                                      N.ptr(gammas), nc, num_routed, n.bit_length() - 1, chunk,
                                                      N.ptr(out)))
    return out[:nc], out[nc:].reshape(nc, num_prods, n)
