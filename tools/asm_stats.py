#!/usr/bin/env python3
"""Per-kernel instruction mix of a hipcc -save-temps .s file (gfx950): static counts, not dynamic."""
import collections
import re
import sys

QUARTER = ("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_i64_i32")


def main(path, top=18):
    lines = open(path).read().split("\n")
    cur, funcs = None, collections.OrderedDict()
    for l in lines:
        m = re.match(r"^(\w+):\s+; @", l)
        if m:
            cur = m.group(1)
            funcs[cur] = []
            continue
        if l.startswith(".Lfunc_end"):
            cur = None
        if cur and l.startswith("\t") and not l.strip().startswith((".", ";")) and l.strip():
            funcs[cur].append(l.strip().split()[0])
    for name, ins in funcs.items():
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        salu = sum(v for k, v in c.items() if k.startswith("s_"))
        q = sum(c[k] for k in QUARTER)
        print("%s: total=%d valu=%d salu=%d quarter-rate=%d" % (name, len(ins), valu, salu, q))
        print("   ", ", ".join("%s=%d" % kv for kv in c.most_common(top)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 18)
