import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (*.so are git-ignored): build them once (hipcc cross-compiles
    gfx950 without a GPU; a few minutes).  The GPU box receives the prebuilt files with the snapshot."""
    import subprocess
    lib = os.path.join(ROOT, "plonky2-merkle-trees_amd", "libp2mt_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "plonky2-merkle-trees_amd", "csrc"), "-j4"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden():
    import json
    out = {}
    gdir = os.path.join(ROOT, "tests", "golden")
    for f in os.listdir(gdir):
        if f.endswith(".json"):
            out[f[:-5]] = json.load(open(os.path.join(gdir, f)))
    return out


def splitmix_leaves(n, seed):
    """n uniform Goldilocks elements: SplitMix64(seed) with rejection of values >= p
    (mirrors rng.gen_range(0..GOLDILOCKS_FIELD_ORDER), merkle_mountain_ranges.rs:336)."""
    import numpy as np
    P = 0xFFFFFFFF00000001
    out = np.empty(n, dtype=np.uint64)
    filled = 0
    state = np.uint64(seed)
    with np.errstate(over="ignore"):
        while filled < n:
            m = max(1024, int((n - filled) * 1.01))
            idx = np.arange(1, m + 1, dtype=np.uint64)
            z = state + idx * np.uint64(0x9E3779B97F4A7C15)
            state = z[-1]
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            z = z[z < np.uint64(P)]
            take = min(z.size, n - filled)
            out[filled:filled + take] = z[:take]
            filled += take
    return out
