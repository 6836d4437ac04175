"""Builders for the permutation-argument tests: wires that satisfy a random set of copy constraints and the sigma
columns that encode them (plonky2: sigma_j(x_i) = k_{j'} * x_{i'} for the position (j', i') the permutation sends
(j, i) to; k_j = 7^j, x_i = w^i)."""
import numpy as np

P = 0xFFFFFFFF00000001


def gl_root(log_n):
    g = pow(7, (P - 1) >> 32, P)
    for _ in range(log_n, 32):
        g = g * g % P
    return g


def make_permutation_instance(degree_bits, num_routed, seed, satisfied=True):
    """-> wires (R, n), sigmas (R, n), k_is (R,) as uint64"""
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    w = gl_root(degree_bits)
    xs = [pow(w, i, P) for i in range(n)]
    k_is = [pow(7, j, P) for j in range(num_routed)]
    total = num_routed * n
    # random partition of the positions into cycles: shuffle, cut into runs, rotate each run
    order = rng.permutation(total)
    sigma_pos = np.arange(total)
    values = np.zeros(total, dtype=object)
    at = 0
    while at < total:
        ln = int(rng.integers(1, 6))
        run = order[at:at + ln]
        v = int(rng.integers(0, P, dtype=np.uint64))
        for a, b in zip(run, np.roll(run, -1)):
            sigma_pos[a] = b
            values[a] = v
        at += ln
    if not satisfied:
        values[order[0]] = (int(values[order[0]]) + 1) % P if len(order) > 1 and sigma_pos[order[0]] != order[0] else values[order[0]]
    wires = np.array([int(v) for v in values], dtype=np.uint64).reshape(num_routed, n)
    sig = np.zeros(total, dtype=np.uint64)
    for pos in range(total):
        jj, ii = divmod(int(sigma_pos[pos]), n)
        sig[pos] = k_is[jj] * xs[ii] % P
    return wires, sig.reshape(num_routed, n), np.array(k_is, dtype=np.uint64), xs


def row_chunk_quotients(wires, sigmas, k_is, xs, beta, gamma, row, chunk):
    """Python-integer restatement of one row's chunk products (for the closing check)."""
    R = wires.shape[0]
    out = []
    for k in range(0, R, chunk):
        num = den = 1
        for j in range(k, min(R, k + chunk)):
            wv = int(wires[j, row])
            num = num * ((wv + beta * (int(k_is[j]) * xs[row] % P) + gamma) % P) % P
            den = den * ((wv + beta * int(sigmas[j, row]) + gamma) % P) % P
        out.append(num * pow(den, P - 2, P) % P)
    return out
