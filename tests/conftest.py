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
    """The bench's leaf generator (lives in the package: plonky2-merkle-trees_amd/synthetic.py)."""
    import __graft_entry__ as ge
    return ge.load_package().synthetic.splitmix_leaves(n, seed)
