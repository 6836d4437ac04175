#!/usr/bin/env python3
"""profiles/r04_commit_phase.txt + the transform kernels' entries of profiles/pmc_summary.json from two tools/profile_cmd.sh runs
(tools/ntt_lde_probe.py --what lde / --what ntt):

  python tools/summarize_commit_profile.py gpurun_out/<lde dir> gpurun_out/<ntt dir> [profiles/rNN_commit_phase.txt] > profiles/rNN_commit_phase.txt

Counters follow /opt/skills/guides/MI355X_MICROARCH.md: one counter set per pass; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE reported
at half the bytes of a wide streaming read on gfx950 (x2 below); GRBM_GUI_ACTIVE sums the 8 XCDs."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+(\S.*?)\s+\[vgpr=(\d+) sgpr=(\d+) lds=(\d+) wg=(\d+)\]", line)
        if m:
            cur = m.group(1)
            out.setdefault(cur, {})["_meta"] = m.groups()[1:]
            continue
        if line.startswith("=="):
            cur = None
        m = re.match(r"\s+(\w+)\s+sum=([\d.e+]+)\s+dispatches=(\d+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2)) / int(m.group(3))
        m = re.match(r"(\S.*?)\s+calls=(\d+)\s+total_ms=\s*([\d.]+)\s+avg_us=\s*([\d.]+)\s+([\d.]+)%\s+min_us=\s*([\d.]+)\s+max_us=\s*([\d.]+)", line)
        if m:
            out.setdefault(m.group(1)[:70], {})["_trace"] = (int(m.group(2)), float(m.group(4)), float(m.group(6)), float(m.group(7)))
    return out


def pick(d, sub):
    r = {}
    for k, v in d.items():
        if sub in k:
            for kk, vv in v.items():
                r.setdefault(kk, vv)
    return r


def section(title, k, points, algo_bytes, per_point_name):
    calls, avg_us, mn, mx = k["_trace"]
    cycles = k["GRBM_GUI_ACTIVE"] / 8
    valu = k["SQ_INSTS_VALU"]
    wave_points = points / 64.0
    fetch, write = k.get("FETCH_SIZE", 0) * 1e3, k.get("WRITE_SIZE", 0) * 1e3
    lines = [title,
             "  kernel trace: %d launches, avg %.1f us (min %.1f, max %.1f)   [vgpr %s sgpr %s lds %s B, workgroup %s]" % ((calls, avg_us, mn, mx) + tuple(k["_meta"])),
             "  algorithmic bytes per launch %.1f MB -> %.0f GB/s = %.3f of the 8 TB/s HBM peak" % (algo_bytes / 1e6, algo_bytes / (avg_us * 1e-6) / 1e9, algo_bytes / (avg_us * 1e-6) / 8e12),
             "  HBM traffic per launch (PMC, own passes): FETCH_SIZE %.1f MB raw (x2 = %.1f MB for streaming reads), WRITE_SIZE %.1f MB -> %.1f MB" % (fetch / 1e6, 2 * fetch / 1e6, write / 1e6, (2 * fetch + write) / 1e6),
             "  SQ_INSTS_VALU %.4g per launch = %.1f VALU instructions per %s; SALU %.4g; LDS instructions %.4g" % (valu, valu / wave_points, per_point_name, k["SQ_INSTS_SALU"], k["SQ_INSTS_LDS"]),
             "  GRBM_GUI_ACTIVE / 8 = %.4g cycles (%.2f GHz over the traced duration); x 1024 SIMDs / VALU instructions = %.2f SIMD cycles per VALU instruction" % (cycles, cycles / (avg_us * 1e3), cycles * 1024 / valu),
             "     (the micro-benchmarked issue cost of this instruction mix is 4.3-5.2 cycles: profiles/r01_valu_issue_rates_gfx950.txt)",
             "  wave time: SQ_WAIT_ANY %.0f %% (s_waitcnt / barrier), SQ_WAIT_INST_ANY %.0f %% (ready, waiting to issue) of SQ_WAVE_CYCLES" % (100 * k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 100 * k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"]),
             "  LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % (k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_LDS_IDX_ACTIVE"], 1)), ""]
    summary = {"valu_instr_per_point": round(valu / wave_points, 1), "hbm_bytes_per_launch": 2 * fetch + write,
               "simd_cycles_per_valu_instr": round(cycles * 1024 / valu, 2), "effective_clock_ghz": round(cycles / (avg_us * 1e3), 2),
               "launch_us_traced": avg_us}
    return lines, summary


OUT_NAME = "profiles/r04_commit_phase.txt"


def main(lde_dir, ntt_dir):
    lde = parse(os.path.join(lde_dir, "summary.txt"))
    ntt = parse(os.path.join(ntt_dir, "summary.txt"))
    out = ["# The transform kernels of the commit step at the sizes where HBM, not launch latency, is what they are up against (VERDICT r3",
           "# item 1b).  rocprofv3 --kernel-trace --stats, then one --pmc pass per counter set (tools/profile_cmd.sh), of",
           "#   python tools/ntt_lde_probe.py --what lde   (x8 coset LDE of 32 x 135 polynomials of 2^12 coefficients: 141.6 MB in, 1 132 MB out)",
           "#   python tools/ntt_lde_probe.py --what ntt   (128 transforms of 2^20 points, natural order in and out: two passes over HBM)",
           "# bench.py's `commit_phase.large_points` measures the same launches live with HIP events.", ""]
    pj = {}
    k = pick(lde, "k_coset_lde12_v2")
    l, s = section("== k_coset_lde12_v2: one launch = 34 560 workgroups (polynomial, coset); 72 B per coefficient (8 read + 64 written)", k,
                   4320 * 4096 * 8, 4320 * 4096 * 72, "wave of 64 output points")
    out += l
    s["source"] = OUT_NAME + " (tools/profile_cmd.sh: one --pmc pass per counter set; traffic = 2 x FETCH_SIZE + WRITE_SIZE)"
    pj["k_coset_lde12_v2@4320x2^12"] = s
    for sub, title, key in (("k_ntt20_pass<0, false, true", "== k_ntt20_pass<forward, columns in, twiddle>: pass 1 (reads 16 B-granule columns... see DESIGN.md); 16 B per point", "pass1"),
                            ("k_ntt20_pass<0, true, false", "== k_ntt20_pass<forward, rows in>: pass 2 (reads whole rows, writes the result transposed = natural order); 16 B per point", "pass2")):
        k = pick(ntt, sub)
        if "_trace" not in k:
            continue
        l, s = section(title, k, 128 * (1 << 20), 128 * (1 << 20) * 16, "wave of 64 points")
        out += l
        s["source"] = OUT_NAME
        pj["k_ntt20_pass@128x2^20:" + key] = s
    if "k_ntt20_pass@128x2^20:pass1" in pj and "k_ntt20_pass@128x2^20:pass2" in pj:
        a, b = pj["k_ntt20_pass@128x2^20:pass1"], pj["k_ntt20_pass@128x2^20:pass2"]
        pj["k_ntt20_pass@128x2^20"] = {"valu_instr_per_point": round((a["valu_instr_per_point"] + b["valu_instr_per_point"]) / 2, 1),
                                       "hbm_bytes_per_launch": (a["hbm_bytes_per_launch"] + b["hbm_bytes_per_launch"]) / 2,
                                       "source": OUT_NAME + " (mean of the two passes)"}
    out += ["== reference points (tools/ubench_granule.hip, same bytes, no arithmetic)",
            "  plain copy, 16 B per lane: 4.8 TB/s; 1024-row x 16-column tiles read and written as 128-byte granules one row apart (pass 1's pattern): 5.2 TB/s;",
            "  rows in / granules out (pass 2's): 5.3 TB/s; with 64-byte granules 4.7 / 4.0 TB/s.  The access patterns are not what bounds the passes."]
    print("\n".join(out))
    path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    cur = json.load(open(path))
    cur.update(pj)
    json.dump(cur, open(path, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 3:
        OUT_NAME = sys.argv[3]  # the file the caller redirects stdout to (recorded as `source` in pmc_summary.json)
    main(sys.argv[1], sys.argv[2])
