"""CPU-only: the wait states hipcc does not place are placed.

The MFMA results of the matrix-pipe MDS layers (csrc/poseidon_fast.hip.h) are read by inline-asm v_mad_u64_u32 behind a hand-placed
`s_nop`; the compiler's hazard recogniser does not look into an asm string.  tools/isa_hazards.py unbundles the gfx950 code object of every
built .o, disassembles it and measures, for EVERY v_mfma site of EVERY kernel, the wait states up to the first instruction that reads or
overwrites the result: a compiler upgrade or a scheduling change that breaks the guard fails here instead of corrupting one node in 10^10."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_hazards  # noqa: E402

CSRC = os.path.join(ROOT, "plonky2-merkle-trees_amd", "csrc")


@pytest.fixture(scope="module")
def reports():
    import __graft_entry__ as ge
    ge.load_package()  # builds the library (and with it the objects) when it is missing
    objs = sorted(glob.glob(os.path.join(CSRC, "*.o")))
    assert objs, "no built objects under csrc/ (run __graft_entry__.build())"
    if not os.path.exists(isa_hazards.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    return {os.path.basename(o): isa_hazards.analyse(o) for o in objs}


def test_objects_hold_gfx950_code_objects(reports):
    """(host-only translation units, e.g. the in-circuit verifier's builder, carry an empty device entry)"""
    with_code = [n for n, r in reports.items() if r["code_objects"] >= 1]
    for must in ("p2mt_mmr.o", "p2mt_hash.o", "p2mt_commit.o", "p2mt_fri.o", "p2mt_circuit.o", "p2mt_verify_dev.o"):
        assert must in with_code, must


def test_mfma_results_are_not_read_early(reports):
    """8-pass XDL (32x32x32 i8): 11 wait states; 4-pass: 7; 2-pass: 5 (LLVM GCNHazardRecognizer, gfx940 family)."""
    bad = [(n, v) for n, r in reports.items() for v in r["mfma_violations"]]
    assert not bad, bad[:5]
    # the check must actually have seen the kernels it is there for: the stage-1 MMR kernel and the hash / commit kernels
    assert reports["p2mt_mmr.o"]["mfma_sites"] >= 100
    assert reports["p2mt_hash.o"]["mfma_sites"] >= 8
    for n, r in reports.items():
        if r["mfma_sites"]:
            assert r["mfma_min_wait_states"] >= 5, (n, r["mfma_min_wait_states"])


def test_checker_sees_a_violation_when_there_is_one():
    """the checker itself: an MFMA read 3 instructions later is flagged, the same behind s_nop 15 is not; a chained accumulate is fine"""
    mf = ("v_mfma_i32_32x32x32_i8", ["v[0:15]", "v[16:19]", "v[20:23]", "0"])
    rd = ("v_mad_u64_u32", ["v[30:31]", "s[0:1]", "v3", "v40", "v[32:33]"])
    fill = ("v_add_u32_e32", ["v50", "v51", "v52"])
    _, _, bad, _, _ = isa_hazards.check({"k": [mf, fill, fill, fill, rd]})
    assert len(bad) == 1 and bad[0]["wait_states"] == 3 and bad[0]["required"] == 11
    _, mn, bad, _, _ = isa_hazards.check({"k": [mf, ("s_nop", ["15"]), rd]})
    assert not bad and mn >= 11
    chain = ("v_mfma_i32_32x32x32_i8", ["v[0:15]", "v[16:19]", "v[20:23]", "v[0:15]"])
    _, _, bad, _, _ = isa_hazards.check({"k": [mf, chain, ("s_nop", ["15"]), rd]})
    assert not bad
    # VALU carry written and read back to back is reported by the (informational) SGPR rule
    _, _, _, _, sb = isa_hazards.check({"k": [("v_sub_co_u32_e64", ["v0", "s[0:1]", "v1", "v2"]),
                                              ("v_subbrev_co_u32_e64", ["v3", "s[0:1]", "0", "v4", "s[0:1]"])]})
    assert len(sb) == 1
