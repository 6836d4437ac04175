"""The product's HOST-side permutation and Challenger (csrc/host_poseidon.hip: what single proves and verifications run their transcript
on, as plonky2 does -- iop/challenger.rs, reached from /root/reference/src/mmr/mmr_plonky2_verifier.rs:148-150) against the oracle.
CPU only: no device call is made."""
import ctypes as C

import numpy as np

import __graft_entry__ as ge

P = 0xFFFFFFFF00000001


def _lib():
    pkg = ge.load_package()
    return pkg, pkg._native.lib()


def test_host_permutation_equals_oracle(oracle):
    pkg, lib = _lib()
    N = pkg._native
    rng = np.random.default_rng(11)
    e = [0, 1, P - 1, P, P + 1, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63]
    states = np.array([[v] * 12 for v in e] + [[e[(i + j) % len(e)] for j in range(12)] for i in range(len(e))], dtype=np.uint64)
    states = np.concatenate([states, rng.integers(0, 1 << 64, size=(3000, 12), dtype=np.uint64), np.arange(12, dtype=np.uint64)[None]])
    got = np.zeros_like(states)
    N.check(lib.p2mt_host_poseidon_permute(N.ptr(states), N.ptr(got), states.shape[0]))
    assert np.array_equal(got, oracle.permute_batch(states))
    assert [int(x) for x in got[-1][:2]] == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47]  # SURVEY A.2 known answer
    assert (got < np.uint64(P)).all()
    # in place
    st = states.copy()
    N.check(lib.p2mt_host_poseidon_permute(N.ptr(st), N.ptr(st), st.shape[0]))
    assert np.array_equal(st, got)


def test_host_challenger_equals_oracle(oracle):
    """observe / squeeze phases of every shape a prove or a verification uses (whole chunks, ragged tails, squeezes that straddle a
    refill, observe after a partial squeeze), non-canonical inputs included"""
    pkg, lib = _lib()
    N = pkg._native
    rng = np.random.default_rng(5)
    for trial in range(20):
        n_phases = int(rng.integers(1, 12))
        n_obs = rng.integers(0, 90, size=n_phases).astype(np.uint32)
        n_sq = rng.integers(0, 30, size=n_phases).astype(np.uint32)
        if trial == 0:
            n_obs, n_sq = np.array([72, 64, 64, 500, 64, 64, 9], np.uint32), np.array([4, 2, 2, 2, 2, 2, 29], np.uint32)
        elems = rng.integers(0, 1 << 64, size=int(n_obs.sum()) + 1, dtype=np.uint64)
        out = np.zeros(int(n_sq.sum()) + 1, np.uint64)
        N.check(lib.p2mt_debug_host_challenger(N.ptr(elems), N.ptr(n_obs), N.ptr(n_sq), len(n_obs), N.ptr(out)))
        ch = oracle.challenger()
        exp, at = [], 0
        for k in range(len(n_obs)):
            if n_obs[k]:
                ch.observe(elems[at:at + int(n_obs[k])] % np.uint64(P))
            at += int(n_obs[k])
            exp += [ch.get_challenge() for _ in range(int(n_sq[k]))]
        assert out[:-1].tolist() == [int(x) for x in exp]


def test_every_host_permutation_variant_equals_oracle():
    """The dispatcher times its candidates on the machine it runs on (scalar compare-and-branch / branch-free, and the AVX-512 full
    rounds where the CPU has them); this pins each in turn (P2MT_HOST_POSEIDON, read once per process) and runs the two tests above."""
    import os
    import subprocess
    import sys
    here = os.path.abspath(__file__)
    for variant in ("br", "bf", "v512br", "v512bf"):  # (the v512 names fall back to the timed choice on a CPU without AVX-512)
        env = dict(os.environ, P2MT_HOST_POSEIDON=variant)
        r = subprocess.run([sys.executable, "-m", "pytest", here, "-x", "-q", "-k", "equals_oracle and not every"], env=env,
                           capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(here)))
        assert r.returncode == 0, (variant, r.stdout[-1500:], r.stderr[-500:])
        assert "2 passed" in r.stdout, (variant, r.stdout[-500:])
