#!/usr/bin/env python3
"""A/B of Poseidon kernel variants on the stage-1 launch of a 2^LOG-leaf MMR build, back to back in ONE process on ONE box
(box-to-box spread is ~7 %, so only same-run comparisons mean anything).

  python tools/ab_variants.py [--log-leaves 24] [--reps 5] 2,0 2,2 ...

Per variant "mds,partial" (p2mt_set_variant): stage-1 launch duration (HIP events on the library stream, p2mt_profile_*), whole
build wall time, SHA-256 of the node array -- which must be the same for every variant (bit-exactness) -- and, with --oracle,
the oracle's root of the same leaves.  Variants are visited round-robin --rounds times so drift shows up as spread."""
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
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--log-leaves", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    pkg.init(0)
    lib = pkg.lib()
    n = 1 << args.log_leaves
    leaves = pkg.synthetic.bench_leaves(args.log_leaves, 0)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).cuda()
    m = pkg.MMR()
    m.reserve(n)
    variants = [tuple(int(x) for x in v.split(",")) for v in args.variants]
    sha, res = {}, {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            pkg.set_variant(*v)
            for _ in range(2):
                m.reset()
                m.extend_dev(d_leaves, n)
            root = m.bagging_the_peaks()
            if v not in sha:
                sha[v] = hashlib.sha256(m.elements.tobytes()).hexdigest()
            torch.cuda.synchronize()
            lib.p2mt_profile_enable(1)
            t0 = time.perf_counter()
            for _ in range(args.reps):
                m.reset()
                m.extend_dev(d_leaves, n)
                root = m.bagging_the_peaks()
            wall = (time.perf_counter() - t0) * 1e3 / args.reps
            ms, cnt = C.c_float(0), C.c_int(0)
            pkg._native.check(lib.p2mt_profile_read(C.byref(ms), C.byref(cnt)))
            lib.p2mt_profile_enable(0)
            res[v].append((ms.value / max(cnt.value, 1), wall))
            print("round %d variant %s: stage-1 launch %.3f ms, build %.3f ms, root %s" % (rnd, v, ms.value / max(cnt.value, 1), wall,
                                                                                       [hex(int(x)) for x in root]), flush=True)
    base = variants[0]
    ok = all(sha[v] == sha[base] for v in variants)
    print("node-array SHA-256 per variant:", {str(v): sha[v][:16] for v in variants}, "ALL EQUAL" if ok else "MISMATCH")
    for v in variants:
        k = np.array([r[0] for r in res[v]])
        w = np.array([r[1] for r in res[v]])
        kb = np.array([r[0] for r in res[base]])
        print("variant %s: stage-1 %.3f ms (min %.3f, max %.3f), build %.3f ms, stage-1 vs %s: %+.1f %%"
              % (v, k.mean(), k.min(), k.max(), w.mean(), base, (k.mean() / kb.mean() - 1) * 100))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
