"""CPU checks of the permutation-argument restatement (oracle/plonk.c; parity unpinned): Z starts at 1, every partial
product is the running product plonky2 defines, and the grand product closes exactly when the copy constraints hold."""
import numpy as np
import pytest

from plonk_cases import P, make_permutation_instance, row_chunk_quotients


@pytest.mark.parametrize("degree_bits,num_routed,chunk", [(3, 8, 8), (4, 80, 8), (5, 10, 4), (4, 7, 3), (6, 80, 8)])
def test_partial_products_and_closing(oracle, degree_bits, num_routed, chunk):
    wires, sigmas, k_is, xs = make_permutation_instance(degree_bits, num_routed, 50 + degree_bits)
    betas = np.array([12345678901234567, 7], np.uint64)
    gammas = np.array([987654321987654321, 11], np.uint64)
    zs, pps = oracle.permutation_partial_products(wires, sigmas, k_is, betas, gammas, chunk)
    n = 1 << degree_bits
    num_chunks = (num_routed + chunk - 1) // chunk
    assert zs.shape == (2, n) and pps.shape == (2, num_chunks - 1, n)
    for c in range(2):
        beta, gamma = int(betas[c]), int(gammas[c])
        assert int(zs[c, 0]) == 1
        for row in (0, 1, n // 2, n - 1):
            q = row_chunk_quotients(wires, sigmas, k_is, xs, beta, gamma, row, chunk)
            acc = int(zs[c, row])
            for k in range(num_chunks - 1):
                acc = acc * q[k] % P
                assert int(pps[c, k, row]) == acc
            z_next = acc * q[-1] % P
            assert z_next == (int(zs[c, row + 1]) if row + 1 < n else 1)  # closes: Z(g x_last) = Z(x_0) = 1


def test_violated_copy_constraint_does_not_close(oracle):
    wires, sigmas, k_is, xs = make_permutation_instance(4, 16, 91, satisfied=False)
    zs, pps = oracle.permutation_partial_products(wires, sigmas, k_is, [3], [5], 8)
    q = row_chunk_quotients(wires, sigmas, k_is, xs, 3, 5, 15, 8)
    z_next = int(pps[0, 0, 15]) * q[-1] % P
    assert z_next != 1


def test_bad_shapes(oracle):
    wires, sigmas, k_is, _ = make_permutation_instance(3, 8, 1)
    with pytest.raises(ValueError):
        oracle.permutation_partial_products(wires, sigmas, k_is, [1], [2], 1)  # max_degree must be > 1
