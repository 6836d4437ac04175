"""Sanitizer leg: the oracle's C restatement rebuilt with AddressSanitizer + UndefinedBehaviorSanitizer
(`make -C oracle asan`) and its own pinned tests re-run against that build in a child process.  GPU ASan / XNACK runs are not
available on this pool, so this is where out-of-bounds reads, overflowing shifts and misaligned accesses in the checker -- the thing
every parity claim rests on -- are hunted.  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if asan is None:
        pytest.skip("no libasan in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    env = dict(os.environ, P2MT_ORACLE_SO=so, LD_PRELOAD=":".join(x for x in (asan, ubsan) if x),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               OMP_NUM_THREADS="4")
    # the pinned goldens (Poseidon, trees, MMR index tables), the FFT/FRI identities and one full plonk prove + verify
    tests = ["tests/test_oracle_golden.py", "tests/test_oracle_fri.py", "tests/test_oracle_plonk.py",
             "tests/test_oracle_circuit.py::test_config3_shape_prove_verify"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + tests,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
