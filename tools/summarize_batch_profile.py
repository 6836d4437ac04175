#!/usr/bin/env python3
"""Per-kernel time of ONE pass of a batched prover out of a rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv).

  python tools/summarize_batch_profile.py <kernel_stats.csv> <kernel that runs once per pass> <proofs per pass> [header line ...]

Prints the table kept under profiles/ (r04_prove_batch_b256.txt, r04_recursion_batch_b32.txt): calls per pass, microseconds per pass,
average duration and share, for every kernel above 0.05 % of the pass."""
import csv
import sys


def main():
    path, marker, per_pass = sys.argv[1], sys.argv[2], int(sys.argv[3])
    for h in sys.argv[4:]:
        print("# " + h)
    rows = list(csv.DictReader(open(path)))
    passes = [int(r["Calls"]) for r in rows if marker in r["Name"]][0]
    total = sum(int(r["TotalDurationNs"]) for r in rows)
    print("# passes profiled: %d; kernel time per pass %.1f us = %.1f us per proof" % (passes, total / passes / 1e3, total / passes / 1e3 / per_pass))
    print("%-64s %10s %12s %10s %6s" % ("kernel", "calls/pass", "us/pass", "avg us", "%"))
    for r in rows:
        share = 100.0 * int(r["TotalDurationNs"]) / total
        if share < 0.05:
            continue
        name = r["Name"].replace("(anonymous namespace)::", "")
        print("%-64s %10.1f %12.1f %10.1f %6.1f" % (name[:64], int(r["Calls"]) / passes, int(r["TotalDurationNs"]) / passes / 1e3,
                                                     float(r["AverageNs"]) / 1e3, share))


if __name__ == "__main__":
    main()
