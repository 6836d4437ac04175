#!/usr/bin/env python3
"""plan_probe.py -- A/B of the one-launch tree build (csrc/p2mt_plan.hip) against the separate launches, on one GPU.

For each size: the node array of every configuration must equal the separate-launch build's (SHA-256 of all elements + root);
then ms per build (reset + extend_dev + root read-back, like bench.py's step) and, with --timeline, the per-item device-clock
timeline of one launch summarised per (kind, level).

  python tools/plan_probe.py --logs 20,22,24 --steps 20 --timeline
"""
import argparse
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logs", default="20,22,24")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--configs", default="off;0,16,12;1,16,12", help="';'-separated: off | order,tq,tw")
    ap.add_argument("--timeline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--dump", default=None, help="write the raw per-item rows of the profiled launch of each config to this prefix")
    a = ap.parse_args()
    pkg = ge.load_package()
    import torch
    N = pkg._native
    lib = N.lib()
    N.check(lib.p2mt_init(0))
    for lg in [int(x) for x in a.logs.split(",")]:
        n = 1 << lg
        host = pkg.synthetic.splitmix_leaves(n, 0x5EED0000 + lg)
        d = torch.from_numpy(host.view(np.int64)).cuda()
        ref_sha = ref_root = None
        for cfg in a.configs.split(";"):
            if cfg == "off":
                N.check(lib.p2mt_debug_plan_knobs(0, -1, -1, -1, -1))
            else:
                o, tq, tw = [int(x) for x in cfg.split(",")]
                N.check(lib.p2mt_debug_plan_knobs(1, 16, o, tq, tw))
            m = pkg.mmr.MMR()
            m.reserve(n)
            times = []
            root = None
            for it in range(a.warmup + a.steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m.reset()
                m.extend_dev(d, n)
                root = m.bagging_the_peaks()
                times.append((time.perf_counter() - t0) * 1e3)
            times = np.array(times[a.warmup:])
            line = "log %d cfg %-10s ms mean %.4f median %.4f min %.4f" % (lg, cfg, times.mean(), np.median(times), times.min())
            if not a.no_check:
                el = m.elements
                sha = hashlib.sha256(el.tobytes()).hexdigest()
                if ref_sha is None:
                    ref_sha, ref_root = sha, root.copy()
                ok = sha == ref_sha and (root == ref_root).all()
                line += "  nodes %s" % ("== separate launches" if ok else "DIFFER")
                if not ok:
                    print(line, flush=True)
                    sys.exit(2)
            print(line, flush=True)
            if a.timeline and cfg != "off":
                N.check(lib.p2mt_debug_plan_profile(1))
                m.reset()
                m.extend_dev(d, n)
                m.bagging_the_peaks()
                N.check(lib.p2mt_debug_plan_profile(0))
                cap = 1 << 16
                rows = np.zeros((cap, 8), np.uint64)
                cnt = lib.p2mt_debug_plan_profile_read(N.ptr(rows), cap)
                if cnt < 0:
                    N.check(int(cnt))
                rows = rows[:cnt]
                if a.dump:
                    np.save("%s_log%d_%s.npy" % (a.dump, lg, cfg.replace(",", "_")), rows)
                summarize(rows)
            del m


def summarize(rows):
    kinds = "SUQW"
    r = rows.astype(np.int64)
    t_end = r[:, 5].max() / 100.0
    s_rows = r[r[:, 0] == 0]
    last_s = s_rows[:, 5].max() / 100.0 if len(s_rows) else 0.0
    print("   launch %.1f us, last stage-1 item ends at %.1f us (tail %.1f us), %d items" % (t_end, last_s, t_end - last_s, len(r)))
    if len(s_rows):
        dur = (s_rows[:, 5] - s_rows[:, 3]) / 100.0
        print("   S items: duration mean %.1f min %.1f max %.1f us; starts %.1f .. %.1f" % (
            dur.mean(), dur.min(), dur.max(), s_rows[:, 3].min() / 100.0, s_rows[:, 3].max() / 100.0))
    for k in (1, 2, 3):
        for h in sorted(set(r[r[:, 0] == k][:, 1])):
            x = r[(r[:, 0] == k) & (r[:, 1] == h)]
            wait = (x[:, 4] - x[:, 3]) / 100.0
            run = (x[:, 5] - x[:, 4]) / 100.0
            print("   %s level %2d: %5d items  start %8.1f .. %8.1f  end %8.1f .. %8.1f  wait mean %6.1f max %6.1f  run mean %5.1f max %5.1f us" % (
                kinds[k], h, len(x), x[:, 3].min() / 100.0, x[:, 3].max() / 100.0, x[:, 5].min() / 100.0, x[:, 5].max() / 100.0,
                wait.mean(), wait.max(), run.mean(), run.max()))


if __name__ == "__main__":
    main()
