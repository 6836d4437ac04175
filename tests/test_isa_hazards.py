"""CPU-only: the wait states hipcc does not place are placed.

The MFMA results of the matrix-pipe MDS layers (csrc/poseidon_fast.hip.h) are read by inline-asm v_mad_u64_u32 behind a hand-placed
`s_nop`; the compiler's hazard recogniser does not look into an asm string.  tools/isa_hazards.py unbundles the gfx950 code object of every
built .o, disassembles it and measures, for EVERY v_mfma site of EVERY kernel, the wait states up to the first instruction that reads or
overwrites the result: a compiler upgrade or a scheduling change that breaks the guard fails here instead of corrupting one node in 10^10."""
import glob
import os
import re
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


SCC_READERS = ("s_addc_u32", "s_subb_u32", "s_cselect_b32", "s_cselect_b64", "s_cmov_b32", "s_cmov_b64", "s_cbranch_scc0", "s_cbranch_scc1")
SCC_NEUTRAL = re.compile(r"^s_(mov|movk|cmov|cselect|mul_i32|mul_hi|mulk|brev|ff0|ff1|flbit|sext|bitset|bitreplicate|getpc|setpc|swappc|call|pack|load|store|"
                         r"buffer|scratch|nop|waitcnt|barrier|sleep|setprio|sendmsg|branch|cbranch|endpgm|sethalt|setkill|getreg|setreg|movrel|dcache|"
                         r"icache|trap|rfe|code_end|memtime|memrealtime|atc_probe|ttrace|inst_prefetch|clause|version|incperflevel|decperflevel|"
                         r"set_gpr_idx|wakeup|atomic)")


def _asm_blocks(text):
    """(start offset, text) of every `asm(...)` / `asm volatile(...)` statement of a source file"""
    out, i = [], 0
    while True:
        m = re.compile(r"\basm\s*(volatile\s*)?\(").search(text, i)
        if not m:
            return out
        k, depth, in_str = m.end() - 1, 0, False
        while True:
            c = text[k]
            if c == '"' and text[k - 1] != "\\":
                in_str = not in_str
            elif not in_str:
                depth += c == "("
                depth -= c == ")"
                if depth == 0:
                    break
            k += 1
        out.append((m.start(), text[m.start():k + 1]))
        i = k + 1


def test_inline_asm_declares_scc_and_vcc_clobbers():
    """An inline-asm block whose text runs a scalar ALU instruction (they write SCC) must say so (`: "scc"`), and one that names vcc or
    uses an e32 carry form must clobber "vcc": the compiler keeps `s_cmp -> s_cselect / s_cbranch_scc` and `s_add_u32 -> s_addc_u32`
    pairs live across asm blocks it believes leave SCC alone.  That is what round 3 took for a compiler defect ("a wave-uniform if / else
    chain around the flag-form arithmetic keeps the fall-through side") and what round 5 found as a 4 GB-off table pointer in
    k_ntt20_pass whenever the sticky flag was set (`s_add_u32 / [asm: s_or_b64 sticky] / s_addc_u32`): the ntt_arith blocks ran
    `s_or_b64 sticky, sticky, carry` without the clobber."""
    bad, seen = [], 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip"))):
        text = open(path).read()
        for off, blk in _asm_blocks(text):
            code = " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', blk))  # (the constraint strings hold no mnemonic)
            ops = re.findall(r"\b(s_[a-z0-9_]+)", code)
            writes_scc = [o for o in ops if not SCC_NEUTRAL.match(o)]
            line = text.count("\n", 0, off) + 1
            if writes_scc:
                seen += 1
                if '"scc"' not in blk:
                    bad.append((os.path.basename(path), line, "scc", writes_scc[0]))
            if (re.search(r"\bvcc\b", code) or re.search(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32_e32|v_cmpx?_\w+_e32|v_div_scale", code)) and '"vcc"' not in blk:
                bad.append((os.path.basename(path), line, "vcc", ""))
    assert seen >= 8, seen  # the flag-form field arithmetic is what this is about
    assert not bad, bad


def test_no_scc_consumer_fed_by_inline_asm():
    """The same thing seen from the gfx950 code of every kernel: no SCC reader (s_addc / s_subb / s_cselect / s_cmov / s_cbranch_scc) may
    take its SCC from a sticky-flag accumulate of the inline asm (`s_or_b64 sX, sX, sY` right behind a VALU instruction with a scalar
    carry-out) -- the compiler cannot mean to, so where it happens an asm block sits between a producer and its consumer unannounced.
    Runs on the objects `build()` made, in the build container: a missing clobber fails here instead of corrupting values on the GPU
    box only when a rare flag is set."""
    import subprocess
    import tempfile
    if not os.path.exists(isa_hazards.OBJDUMP):
        pytest.skip("llvm-objdump not in this image")
    import __graft_entry__ as ge
    ge.load_package()
    sites, flagged = 0, []
    for obj in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
        for blob in isa_hazards.extract_gfx950(obj):
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(blob)
                f.flush()
                dis = subprocess.run([isa_hazards.OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
            cur, funcs = None, {}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
                if m:
                    cur = funcs.setdefault(m.group(1), [])
                    continue
                if cur is None or not line.startswith("\t"):
                    continue
                m = re.match(r"^\t(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
                if m:
                    cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
            for name, ins in funcs.items():
                if not ins:
                    continue
                base = ins[0][0]
                targets = set()
                for a, op, ops in ins:
                    if op.startswith("s_cbranch") or op == "s_branch":
                        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", ops)
                        if m:
                            targets.add(base + int(m.group(1), 16))
                        else:
                            m2 = re.match(r"^(\d+)", ops)
                            if m2:
                                off = int(m2.group(1))
                                targets.add(a + 4 + 4 * (off - 65536 if off >= 32768 else off))
                writer = None  # index of the instruction whose SCC is current on the straight-line path (None: unknown / block entry)
                for i, (a, op, ops) in enumerate(ins):
                    if a in targets:
                        writer = None
                    if op in SCC_READERS and writer is not None:
                        flagged.append((os.path.basename(obj), name[:80], hex(a), op))
                    if op.startswith("s_") and not SCC_NEUTRAL.match(op):
                        writer = None
                        if op == "s_or_b64":
                            o = [x.strip() for x in ops.split(",")]
                            if len(o) == 3 and o[0] == o[1] and o[0].startswith("s[") and o[2].startswith("s["):
                                prev = " ".join(p[1] for p in ins[max(0, i - 3):i])
                                if re.search(r"v_(subb?|addc?|subbrev)_co_u32|v_mad_u64_u32|v_cmp", prev):
                                    writer = i  # asm-form accumulate: its SCC is nobody's business
                                    sites += 1
    assert sites > 1000, sites  # the check saw the flag-form arithmetic it is there for
    assert not flagged, flagged[:20]
