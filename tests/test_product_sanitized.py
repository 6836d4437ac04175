"""Sanitizer leg for the PRODUCT's host code: libp2mt_hip.so's sources rebuilt host-only
(`hipcc --offload-host-only`: no kernels) with AddressSanitizer + UBSan (`make -C plonky2-merkle-trees_amd/csrc asan`), and the CPU
tests of the C ABI -- every exported symbol, index tables, host-side tree indexing, the no-fallback refusals, the circuit builder's
host logic for five circuit shapes -- re-run against that build in a child process (clang's ASan runtime preloaded into python,
P2MT_LIB_PATH selecting the library).  GPU ASan / XNACK runs are not available on this pool; the kernels are covered by the parity
suite, the host side by this.  CPU only."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(1200)
def test_product_host_code_under_asan_ubsan():
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("clang's shared ASan runtime is not in this image")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "plonky2-merkle-trees_amd", "csrc"), "-s", "-j4", "asan"])
    so = os.path.join(ROOT, "plonky2-merkle-trees_amd", "libp2mt_hip_asan.so")
    env = dict(os.environ, P2MT_LIB_PATH=so, LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_abi_cpu.py", "tests/test_sharded_gloo.py::test_world1_geometry"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1100)
    tail = r.stdout[-3000:] + r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
