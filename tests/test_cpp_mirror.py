"""The C++ host-side mirror (include/p2mt.hpp): compiles on CPU; on a GPU box the reference's own unit tests,
re-expressed in C++ (tests/cpp/test_mirror.cpp), run through it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_mirror")


def _build():
    pkg_dir = os.path.join(ROOT, "plonky2-merkle-trees_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", EXE, os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp"),
                           "-L", pkg_dir, "-lp2mt_hip", "-Wl,-rpath,$ORIGIN/../../plonky2-merkle-trees_amd",
                           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_mirror_compiles_and_refuses_without_gpu():
    _build()
    import __graft_entry__ as ge
    if ge.load_package().device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 77 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_reference_unit_tests_through_cpp_mirror():
    if not os.path.exists(EXE):
        _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "8 reference tests + prover pieces + the recursion passed" in r.stdout


def _pick_hash_circuit(builder, pick_hash):
    x, y = builder.add_virtual_target(), builder.add_virtual_target()
    m1 = builder.mul(x, y)
    m2 = builder.mul(m1, y)
    m3 = builder.mul(m2, x)
    h1, h2 = builder.add_virtual_hash(), builder.add_virtual_hash()
    pick_left = builder.add_virtual_bool_target_safe()
    out = pick_hash(builder, h1, h2, pick_left)
    builder.register_public_inputs(out)
    builder.register_public_inputs([m3])
    return builder.build()


@pytest.mark.gpu
def test_pick_hash_call_order_is_the_references_in_every_mirror():
    """common.rs:48-55 issues four `mul`s and then four `mul_add`s.  With arithmetic in front of it (three `mul`s leave an
    ArithmeticGate partly filled) any other interleaving lands in different gate slots: the C++ mirror, the Python mirror and the
    oracle's builder must give ONE circuit digest (round 2's C++ mirror interleaved mul / mul_add per element)."""
    import sys
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import __graft_entry__ as ge
    from oracle import circuit as OC
    from oracle_lib import Oracle
    pkg = ge.load_package()
    pkg.init(0)
    from plonky2_merkle_trees_amd import mmr_plonky2_verifier as G
    cd = _pick_hash_circuit(pkg.CircuitBuilder(), G.pick_hash)
    ocd = _pick_hash_circuit(OC.CircuitBuilder(Oracle()), OC.pick_hash)
    want = [int(v) for v in np.asarray(ocd.circuit_digest).reshape(4)]
    assert [int(v) for v in cd.constants_sigmas()[2]] == want
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("pick_hash_call_order digest:")][0]
    assert [int(v) for v in line.split(":")[1].split()] == want
