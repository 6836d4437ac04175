#!/usr/bin/env python3
"""Static issue-cost estimate per kernel from a hipcc -save-temps .s file, using the measured gfx950 cost model
(profiles/r01_valu_issue_rates_gfx950.txt): full-rate VALU = 1 unit, other VALU = 2 units.  Loop bodies are
weighted by the trip counts given on the command line as label=count (e.g. .LBB0_1=4)."""
import collections
import re
import sys

FULL = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_not_b32"}


def units(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if not base.startswith("v_"):
        return 0
    return 1 if base in FULL else 2


def main(path, weights):
    lines = open(path).read().split("\n")
    cur = None
    funcs = collections.OrderedDict()
    for l in lines:
        m = re.match(r"^(\w+):\s+; @", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
        if cur is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            funcs[cur].append(("label", m.group(1)))
        elif l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";")):
            funcs[cur].append(("ins", l.strip().split()[0], l.strip()))
    for name, items in funcs.items():
        # blocks: label -> instrs ; a loop is a block that branches back to its own label
        blocks, label = collections.OrderedDict(), "entry"
        blocks[label] = []
        for it in items:
            if it[0] == "label":
                label = it[1]
                blocks[label] = []
            else:
                blocks[label].append(it)
        total_u = total_i = 0
        desc = []
        for lab, ins in blocks.items():
            w = weights.get(lab, 1)
            selfloop = any(i[2].startswith("s_cbranch") and lab in i[2] for i in ins)
            u = sum(units(i[1]) for i in ins)
            n = sum(1 for i in ins if i[1].startswith("v_"))
            total_u += u * w
            total_i += n * w
            if n > 20:
                desc.append("%s%s: valu=%d units=%d x%d" % (lab, "(loop)" if selfloop else "", n, u, w))
        print("%s: weighted VALU instrs=%d units=%d" % (name, total_i, total_u))
        for d in desc:
            print("    " + d)


if __name__ == "__main__":
    w = {}
    for a in sys.argv[2:]:
        k, v = a.split("=")
        w[k] = int(v)
    main(sys.argv[1], w)
