#!/usr/bin/env python3
"""Static hazard check of the gfx950 code objects inside the built .o files.

hipcc pads the hazards of the instructions IT emits, but nothing inside an inline-asm string: the wait states between an MFMA and the
first reader of its result (poseidon_fast.hip.h: `s_nop 15` / `s_nop 7` ahead of the inline-asm mads) and between a VALU instruction that
writes an SGPR / VCC carry and the VALU instruction that reads it (ntt_arith.hip.h) are placed by hand.  A compiler upgrade or a
scheduling change can move things silently; this reads the ISA that ships.

Rules (LLVM GCNHazardRecognizer, gfx940 family; /opt/skills/guides/cdna_hip_programming.md 5.7):
  * XDL (MFMA) writes a VGPR -> any instruction that reads or overwrites it: passes + 3 wait states (2-pass 5, 4-pass 7, 8-pass 11,
    16-pass 19), except the next MFMA taking the whole result as its accumulator input;
  * VALU writes an SGPR (or VCC) -> VALU reads it: 2 wait states.
A wait state is one issued instruction of the same wave (s_nop N counts N + 1).

  python tools/isa_hazards.py plonky2-merkle-trees_amd/csrc/p2mt_mmr.o [...]   -> one JSON line per object
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def extract_gfx950(path):
    """-> list of code-object byte strings for gfx950 found in the clang offload bundle(s) of a host object"""
    data = open(path, "rb").read()
    out, at = [], 0
    while True:
        at = data.find(MAGIC, at)
        if at < 0:
            break
        (n,) = struct.unpack_from("<Q", data, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(data[at + off:at + off + size])
        at += len(MAGIC)
    return out


REG = re.compile(r"\b([vas])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(operand_text):
    """set of (file, index) named in an operand string; vcc counts as ('s', 'vcc')"""
    s = set()
    for m in REG.finditer(operand_text):
        lo = int(m.group(2) if m.group(2) is not None else m.group(4))
        hi = int(m.group(3) if m.group(3) is not None else m.group(4))
        for i in range(lo, hi + 1):
            s.add((m.group(1), i))
    if re.search(r"\bvcc\b", operand_text):
        s.add(("s", "vcc"))
    return s


def mfma_wait_states(op):
    m = re.match(r"v_(?:s?mfma[a-z]*)_[a-z0-9]+_(\d+)x(\d+)x(\d+)", op)
    if not m:
        return 19
    mm, nn = int(m.group(1)), int(m.group(2))
    if op.startswith("v_mfma_f32_32x32x2") or op.startswith("v_mfma_f64"):
        return 19
    return {32: 11, 16: 7, 4: 5}.get(mm, 19)


def parse(disasm):
    """-> {function: [(op, [operand strings])]}"""
    funcs, cur = {}, None
    for line in disasm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        text = line.split("//")[0].strip()
        if not text:
            continue
        parts = text.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((parts[0], ops))
    return funcs


def is_valu(op):
    return op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_smfma")


def states(op, ops):
    if op == "s_nop":
        return int(ops[0], 0) + 1
    return 1


def check(funcs):
    """-> (mfma_sites, min mfma distance found, [mfma violations], sgpr_sites, [sgpr violations])"""
    mfma_sites, mfma_min, mfma_bad, sgpr_sites, sgpr_bad = 0, None, [], 0, []
    for name, ins in funcs.items():
        for i, (op, ops) in enumerate(ins):
            if op.startswith("v_mfma") or op.startswith("v_smfma"):
                mfma_sites += 1
                need, dst = mfma_wait_states(op), regs(ops[0])
                dist, j = 0, i + 1
                while j < len(ins) and dist < need:
                    op2, ops2 = ins[j]
                    if op2 in ("s_endpgm", "s_branch", "s_setpc_b64"):
                        dist = need
                        break
                    touched = set().union(*[regs(o) for o in ops2]) if ops2 else set()
                    if op2.startswith("v_mfma") or op2.startswith("v_smfma"):
                        # MFMA after MFMA (both compiler-visible, listed for completeness): the whole result as the accumulator input
                        # of the next one needs nothing; an overlapping SrcC needs `passes` wait states, SrcA / SrcB passes + 2; a later
                        # MFMA that only overwrites the registers is ordered by the in-order matrix pipe and ends the window
                        passes = need - 3
                        ab = (regs(ops2[1]) | regs(ops2[2])) & dst if len(ops2) >= 3 else set()
                        cc = regs(ops2[3]) & dst if len(ops2) >= 4 else set()
                        if ab and dist < passes + 2:
                            need = passes + 2
                            break
                        if cc and regs(ops2[3]) != dst and dist < passes:
                            need = passes
                            break
                        if regs(ops2[0]) & dst and not ab:
                            dist = need
                            break
                    elif touched & dst:
                        break
                    dist += states(op2, ops2)
                    j += 1
                else:
                    if j >= len(ins):
                        dist = need
                if mfma_min is None or dist < mfma_min:
                    mfma_min = dist
                if dist < need:
                    mfma_bad.append({"function": name, "index": i, "op": op, "wait_states": dist, "required": need})
            if is_valu(op) and ops:
                # SGPR / VCC results: VOP3 carry-out / compare destinations are operands 0 or 1; e32 carries write vcc implicitly
                written = set()
                for o in ops[:2]:
                    written |= {r for r in regs(o) if r[0] == "s"}
                if re.match(r"v_(add|sub|subrev)_co_u32_e32|v_(addc|subb|subbrev)_co_u32_e32|v_cmp", op) and not any(r[0] == "s" for r in written):
                    written.add(("s", "vcc"))
                if not written:
                    continue
                dist, j = 0, i + 1
                while j < len(ins) and dist < 2:
                    op2, ops2 = ins[j]
                    if op2 in ("s_endpgm", "s_branch", "s_setpc_b64"):
                        break
                    if is_valu(op2):
                        # sources: every operand but the destinations (operand 0, and operand 1 when it is an SGPR carry-out)
                        srcs = ops2[1:]
                        if len(ops2) > 1 and re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_u32|v_mad_[ui]64_[ui]32|v_div_scale", op2) and \
                                regs(ops2[1]) and all(r[0] == "s" for r in regs(ops2[1])):
                            srcs = ops2[2:]
                        read = set().union(*[regs(o) for o in srcs]) if srcs else set()
                        if re.search(r"_e32$", op2) and re.match(r"v_(addc|subb|subbrev)_co_u32|v_cndmask_b32", op2):
                            read.add(("s", "vcc"))
                        if read & written:
                            sgpr_sites += 1
                            sgpr_bad.append({"function": name, "index": i, "op": op, "reader": op2, "wait_states": dist})
                            break
                    # an instruction that overwrites the register ends the hazard window for it
                    dist += states(op2, ops2)
                    j += 1
    return mfma_sites, mfma_min, mfma_bad, sgpr_sites, sgpr_bad


def analyse(path):
    res = {"object": os.path.basename(path), "code_objects": 0, "mfma_sites": 0, "mfma_min_wait_states": None, "mfma_violations": [],
           "sgpr_violations": []}
    for blob in extract_gfx950(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        ms, mn, mb, _, sb = check(parse(dis))
        res["code_objects"] += 1
        res["mfma_sites"] += ms
        if mn is not None and (res["mfma_min_wait_states"] is None or mn < res["mfma_min_wait_states"]):
            res["mfma_min_wait_states"] = mn
        res["mfma_violations"] += mb
        res["sgpr_violations"] += sb
    return res


if __name__ == "__main__":
    for p in sys.argv[1:]:
        r = analyse(p)
        by_fn = {}
        for v in r["sgpr_violations"]:
            by_fn[v["function"]] = by_fn.get(v["function"], 0) + 1
        r["sgpr_violations_by_function"] = by_fn
        r["sgpr_violations"] = len(r["sgpr_violations"])
        print(json.dumps(r))
