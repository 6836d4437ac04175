"""GPU parity for the permutation-argument stage (Z + partial products) against oracle/plonk.c, through the C ABI.
PARITY UNPINNED with respect to plonky2 itself; bit-exact against the oracle, plus the argument's own
closing property."""
import numpy as np
import pytest

import __graft_entry__ as ge
from plonk_cases import P, make_permutation_instance, row_chunk_quotients

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.init(0)
    return p


@pytest.mark.parametrize("degree_bits,num_routed,chunk", [(0, 8, 8), (1, 3, 2), (3, 8, 8), (4, 80, 8), (5, 10, 4), (4, 7, 3),
                                                          (6, 80, 8), (10, 80, 8), (12, 80, 8), (13, 16, 8)])
def test_partial_products_vs_oracle(pkg, oracle, degree_bits, num_routed, chunk):
    if degree_bits >= 10:  # random columns (no copy constraints): the oracle comparison is what matters at size
        rng = np.random.default_rng(degree_bits)
        wires = rng.integers(0, P, size=(num_routed, 1 << degree_bits), dtype=np.uint64)
        sigmas = rng.integers(0, P, size=(num_routed, 1 << degree_bits), dtype=np.uint64)
        k_is = pkg.plonk.coset_shifts(num_routed)
    else:
        wires, sigmas, k_is, _ = make_permutation_instance(degree_bits, num_routed, 70 + degree_bits)
    betas = np.array([12345678901234567, P + 3], np.uint64)  # second pair is non-canonical on purpose
    gammas = np.array([987654321987654321, 11], np.uint64)
    zs, pps = pkg.all_wires_permutation_partial_products(wires, sigmas, betas, gammas, k_is, chunk)
    zs_o, pps_o = oracle.permutation_partial_products(wires, sigmas, k_is, betas, gammas, chunk)
    assert np.array_equal(zs, zs_o) and np.array_equal(pps, pps_o)


def test_grand_product_closes_on_device(pkg):
    degree_bits, num_routed, chunk = 7, 80, 8
    wires, sigmas, k_is, xs = make_permutation_instance(degree_bits, num_routed, 123)
    zs, pps = pkg.all_wires_permutation_partial_products(wires, sigmas, [5], [9], k_is, chunk)
    n = 1 << degree_bits
    assert int(zs[0, 0]) == 1
    q = row_chunk_quotients(wires, sigmas, k_is, xs, 5, 9, n - 1, chunk)
    assert int(pps[0, -1, n - 1]) * q[-1] % P == 1
    bad_w, bad_s, _, _ = make_permutation_instance(degree_bits, num_routed, 123, satisfied=False)
    zs2, pps2 = pkg.all_wires_permutation_partial_products(bad_w, bad_s, [5], [9], k_is, chunk)
    q2 = row_chunk_quotients(bad_w, bad_s, k_is, xs, 5, 9, n - 1, chunk)
    assert int(pps2[0, -1, n - 1]) * q2[-1] % P != 1


def test_partial_products_panics(pkg):
    wires, sigmas, k_is, _ = make_permutation_instance(3, 8, 1)
    with pytest.raises(pkg.P2mtPanic):
        pkg.all_wires_permutation_partial_products(wires, sigmas, [1], [2], k_is, 1)       # max_degree > 1
    with pytest.raises(pkg.P2mtPanic):
        pkg.all_wires_permutation_partial_products(wires, sigmas[:, :4], [1], [2], k_is)  # ragged
    # a zero denominator: w + beta*sigma + gamma = 0 at one position (plonky2 panics on the division)
    w2, s2 = wires.copy(), sigmas.copy()
    beta, gamma = 3, 5
    w2[0, 0] = np.uint64((P - (beta * int(s2[0, 0]) + gamma) % P) % P)
    with pytest.raises(pkg.P2mtPanic):
        pkg.all_wires_permutation_partial_products(w2, s2, [beta], [gamma], k_is)
