#!/usr/bin/env python3
"""Per-kernel timeline of ONE step from a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 bench.py ...`).

usage: rocpd_timeline.py <results.db> <first-kernel-of-a-step substring> > profiles/rNN_name.txt
Prints the launches between the last two occurrences of the marker kernel and the per-kernel totals."""
import collections
import sqlite3
import sys


def main(path, marker):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(idx) < 2:
        raise SystemExit("marker kernel %r seen %d times" % (marker, len(idx)))
    a, b = idx[-2], idx[-1]
    t0 = rows[a][1]
    agg = collections.OrderedDict()
    print("# columns: start offset (us), duration (us), kernel, workgroups")
    for name, start, end, gx, wx in rows[a:b]:
        nm = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:44]
        print("%8.1f %7.1f  %-44s %d" % ((start - t0) / 1e3, (end - start) / 1e3, nm, gx // max(wx, 1)))
        c = agg.setdefault(nm, [0, 0.0])
        c[0] += 1
        c[1] += (end - start) / 1e3
    span = (rows[b][1] - t0) / 1e3
    busy = sum(v[1] for v in agg.values())
    print("# step span %.1f us, %d launches, device busy %.1f us (%.0f %%)" % (span, b - a, busy, 100 * busy / span))
    print("# per kernel: launches, total us")
    for k, v in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("#   %-44s %4d %9.1f" % (k, v[0], v[1]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
