#!/usr/bin/env python3
"""MMR.elements to the host: pageable copy, pinned copy, and the extend that streams what it appends (chunked, copies overlapped
with hashing).  ms from `reset` to the last byte on the host, node arrays compared by SHA-256."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
pkg.init(0)
import torch  # noqa: E402
lib, N = pkg.lib(), pkg._native
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << lg
host = pkg.synthetic.splitmix_leaves(n, 0x5EED0000 + lg)
d = torch.from_numpy(host.view(np.int64)).cuda()
m = pkg.mmr.MMR()
m.reserve(n)
out = {"log_leaves": lg}


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def build():
    m.reset()
    m.extend_dev(d, n)
    m.bagging_the_peaks()


out["build_ms"] = timed(build)
count = len(m)
pageable = np.zeros((count, 4), np.uint64)


def build_pageable():
    build()
    N.check(lib.p2mt_mmr_copy_elements(m._h, 0, count, N.ptr(pageable)))


out["build_plus_elements_pageable_ms"] = timed(build_pageable, 3)
ref = hashlib.sha256(pageable.tobytes()).hexdigest()
pin = pkg.mmr.PinnedBuffer(4 * count)


def build_pinned():
    build()
    m.copy_elements_async(0, count, pin)
    N.check(lib.p2mt_sync())


out["build_plus_elements_pinned_ms"] = timed(build_pinned, 3)
out["pinned_equal"] = hashlib.sha256(pin.array.tobytes()).hexdigest() == ref
for cl in (20, 21, 22, 23):
    if cl > lg:
        continue
    pin.array[:] = 0

    def build_overlapped():
        m.reset()
        m.extend_dev_to_host(d, n, pin, chunk_log=cl)
        N.check(lib.p2mt_sync())
    out["build_plus_elements_overlapped_ms_chunk%d" % cl] = timed(build_overlapped, 3)
    out["overlapped_equal_chunk%d" % cl] = hashlib.sha256(pin.array.tobytes()).hexdigest() == ref
    assert np.array_equal(m.bagging_the_peaks(), pageable[-1])
print(json.dumps(out))
