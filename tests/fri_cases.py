"""Shared builders for the FRI tests: a synthetic opening instance shaped like plonky2's (all polynomials of every
oracle opened at zeta, the first polynomials of the last oracle also at g*zeta -- FRI_ORACLES / FriInstanceInfo in
plonk/plonk_common.rs)."""
import numpy as np

P = 0xFFFFFFFF00000001


def rand(shape, seed):
    return np.random.default_rng(seed).integers(0, P, size=shape, dtype=np.uint64)


def make_instance(oracle, degree_bits, n_polys_per_oracle, seed, n_next=2):
    """-> (coeff arrays, batches, zeta, g*zeta)"""
    n = 1 << degree_bits
    coeffs = [rand((k, n), seed + 17 * i) for i, k in enumerate(n_polys_per_oracle)]
    zeta = rand(2, seed + 999)
    g = oracle.lib.oracle_gl_primitive_root_of_unity(degree_bits)
    gz = np.array([oracle.lib.oracle_gl_mul(int(zeta[0]), g), oracle.lib.oracle_gl_mul(int(zeta[1]), g)], np.uint64)
    all_polys = [(oi, pi) for oi, k in enumerate(n_polys_per_oracle) for pi in range(k)]
    last = len(n_polys_per_oracle) - 1
    nxt = [(last, pi) for pi in range(min(n_next, n_polys_per_oracle[last]))]
    return coeffs, [(zeta, all_polys), (gz, nxt)]


def oracle_commit(oracle, coeffs, params):
    """[(coeffs, leaves, digests)], caps"""
    out, caps = [], []
    for c in coeffs:
        leaves, dig, cap = oracle.polynomial_batch_commit(c, False, params.rate_bits, params.cap_height)
        out.append((c, leaves, dig))
        caps.append(cap)
    return out, np.concatenate(caps)


def openings_of(oracle, coeffs, batches):
    return [np.concatenate([oracle.eval_polys_ext(coeffs[oi][pi:pi + 1], pt) for oi, pi in pl]) for pt, pl in batches]
