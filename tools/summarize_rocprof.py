#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + PMC passes) into one small text file for profiles/.

usage: summarize_rocprof.py <prof_dir> <builds_in_pmc_runs> > profiles/rNN_name.txt
PMC conventions follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KB;
on gfx950 FETCH_SIZE reports 1/2 of a coalesced streaming read, WRITE_SIZE is exact; each counter set is
collected in its own pass.
"""
import collections
import csv
import glob
import os
import sys


def main(d, builds):
    out = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        out.append("== kernel-trace --stats (%s)" % os.path.relpath(f, d))
        for r in csv.DictReader(open(f)):
            out.append("%-90s calls=%-5s total_ms=%9.3f avg_us=%10.2f  %5s%%  min_us=%9.2f max_us=%9.2f" % (
                r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                r["Percentage"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*_counter_collection.csv"), recursive=True)):
        out.append("== pmc pass %s (sums over all dispatches; %d builds in this run)" % (os.path.relpath(f, d), builds))
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        calls = collections.Counter()
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[(k, r["Counter_Name"])] += 1
            meta[k] = (r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"])
        for k, v in agg.items():
            out.append("  %s  [vgpr=%s sgpr=%s lds=%s wg=%s]" % ((k,) + meta[k]))
            for c, x in v.items():
                line = "      %-22s sum=%.6g  dispatches=%d  per_build=%.6g" % (c, x, calls[(k, c)], x / builds)
                if c == "FETCH_SIZE":
                    line += "  => %.1f MB/build raw, %.1f MB/build with the gfx950 x2 streaming-read correction" % (
                        x / builds / 1e3, 2 * x / builds / 1e3)
                if c == "WRITE_SIZE":
                    line += "  => %.1f MB/build" % (x / builds / 1e3)
                out.append(line)
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
