#!/usr/bin/env python3
"""Soak: the 2^LOG-leaf MMR build N times in a row on the shipped kernels, the root compared every time and the SHA-256 of the whole node
array at the start and at the end (the dense MDS layers run as MFMAs whose results are read by inline-assembly mads behind a
software-managed hazard guard: a marginal guard would show up as rare, timing-dependent wrong nodes).
  python tools/soak_build.py [--log-leaves 24] [--builds 500]"""
import argparse
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-leaves", type=int, default=24)
    ap.add_argument("--builds", type=int, default=500)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    pkg.init(0)
    n = 1 << args.log_leaves
    leaves = pkg.synthetic.bench_leaves(args.log_leaves, 0)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).cuda()
    m = pkg.MMR()
    m.reserve(n)
    m.extend_dev(d_leaves, n)
    root0 = m.bagging_the_peaks().copy()
    sha0 = hashlib.sha256(m.elements.tobytes()).hexdigest()
    bad = 0
    for i in range(args.builds):
        m.reset()
        m.extend_dev(d_leaves, n)
        if not np.array_equal(m.bagging_the_peaks(), root0):
            bad += 1
            print("build %d: root differs" % i, flush=True)
    sha1 = hashlib.sha256(m.elements.tobytes()).hexdigest()
    print("soak: %d builds of 2^%d leaves, %d wrong roots, node-array SHA-256 %s -> %s (%s)" % (
        args.builds, args.log_leaves, bad, sha0[:16], sha1[:16], "equal" if sha0 == sha1 else "DIFFERENT"))
    sys.exit(1 if bad or sha0 != sha1 else 0)


if __name__ == "__main__":
    main()
