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
