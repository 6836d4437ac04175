#!/usr/bin/env python3
"""Time the transform kernels of the commit step at sizes where HBM is the bound:

  lde   p2mt_coset_lde_leaf_order_dev   n_polys x 2^log_n coefficients -> x 2^rate_bits   (72 B per coefficient at rate 3)
  ntt   p2mt_ntt_batch_dev              n_polys x 2^log_n points, natural -> natural       (16 B per point)

  python tools/ntt_lde_probe.py [--what lde,ntt] [--lde-log 12 --lde-polys 4320] [--ntt-log 20 --ntt-polys 128] [--reps 10]

Prints one JSON line per point: wall ms per call (torch events on the library's stream = the null stream), algorithmic GB/s and
the fraction of the 8 TB/s HBM peak, plus a size-independent correctness check (LDE: Horner evaluation of a few polynomials at a
few points; NTT: inverse(forward(x)) == x and forward == Horner at a few points).  Meant to run under rocprofv3 as well."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

P = 0xFFFFFFFF00000001
HBM_PEAK_GBS = 8000.0


def root_of_unity(log_n):
    g = pow(7, (P - 1) >> 32, P)
    for _ in range(log_n, 32):
        g = g * g % P
    return g


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def horner(coeffs, x):
    acc = 0
    for c in coeffs[::-1]:
        acc = (acc * x + int(c)) % P
    return acc


def time_calls(torch, fn, reps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return float(np.median(ms)), float(np.min(ms))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="lde,ntt")
    ap.add_argument("--lde-log", type=int, default=12)
    ap.add_argument("--lde-polys", type=int, default=32 * 135)
    ap.add_argument("--rate-bits", type=int, default=3)
    ap.add_argument("--ntt-log", type=int, default=20)
    ap.add_argument("--ntt-polys", type=int, default=128)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    pkg = ge.load_package()
    N = pkg._native
    lib = pkg.lib()
    N.check(lib.p2mt_init(0))
    rng = np.random.default_rng(11)

    def rand_dev(count):
        # uniform u64 below p, generated on the device (the host would need seconds for 2^27 values)
        g = torch.Generator(device="cuda")
        g.manual_seed(1234)
        hi = torch.randint(0, 0xFFFFFFFF, (count,), dtype=torch.int64, device="cuda", generator=g)  # < 2^32 - 1 => value < p
        lo = torch.randint(0, 1 << 32, (count,), dtype=torch.int64, device="cuda", generator=g)
        return (hi << 32) | lo

    for what in args.what.split(","):
        if what == "lde":
            log_n, w, r = args.lde_log, args.lde_polys, args.rate_bits
            n, big = 1 << log_n, 1 << (log_n + r)
            d_in = rand_dev(w * n)
            d_out = torch.empty(w * big, dtype=torch.int64, device="cuda")

            def call():
                N.check(lib.p2mt_coset_lde_leaf_order_dev(N.ptr(d_in), log_n, r, 7, w, N.ptr(d_out)))

            med, mn = time_calls(torch, call, args.reps)
            # check: a few polynomials at a few points against Horner
            ok = True
            wN = root_of_unity(log_n + r)
            for p in (0, w // 2, w - 1):
                coeffs = d_in[p * n:(p + 1) * n].cpu().numpy().view(np.uint64)
                vals = d_out[p * big:(p + 1) * big].cpu().numpy().view(np.uint64)
                for i in (0, 1, big // 3, big - 1):
                    x = 7 * pow(wN, i, P) % P
                    ok &= int(vals[brev(i, log_n + r)]) % P == horner(coeffs, x)
            algo = w * n * 8 * (1 + (1 << r))
            print(json.dumps({"what": "coset_lde_leaf_order", "log_n": log_n, "n_polys": w, "rate_bits": r,
                              "bytes_in": w * n * 8, "bytes_out": w * big * 8, "ms_median": med, "ms_min": mn,
                              "algorithmic_bytes": algo, "algorithmic_GBps": algo / (med * 1e-3) / 1e9,
                              "frac_of_hbm_peak": algo / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, "horner_check": bool(ok)}), flush=True)
            del d_in, d_out
        elif what == "ntt":
            log_n, w = args.ntt_log, args.ntt_polys
            n = 1 << log_n
            d = rand_dev(w * n)
            keep = {p: d[p * n:(p + 1) * n].cpu().numpy().view(np.uint64).copy() for p in (0, w - 1)}

            def call():
                N.check(lib.p2mt_ntt_batch_dev(N.ptr(d), log_n, w, 0))

            N.check(lib.p2mt_ntt_batch_dev(N.ptr(d), log_n, w, 0))
            torch.cuda.synchronize()
            ok = True
            wn = root_of_unity(log_n)
            for p, coeffs in keep.items():
                vals = d[p * n:(p + 1) * n].cpu().numpy().view(np.uint64)
                for i in (0, 1, n // 3, n - 1):
                    ok &= int(vals[i]) % P == horner(coeffs, pow(wn, i, P))
            N.check(lib.p2mt_ntt_batch_dev(N.ptr(d), log_n, w, 1))
            torch.cuda.synchronize()
            for p, coeffs in keep.items():
                back = d[p * n:(p + 1) * n].cpu().numpy().view(np.uint64)
                ok &= bool(np.array_equal(back % np.uint64(P), coeffs % np.uint64(P)))
            med, mn = time_calls(torch, call, args.reps)
            algo = w * n * 16
            print(json.dumps({"what": "ntt_batch (natural -> natural)", "log_n": log_n, "n_polys": w, "bytes": w * n * 8,
                              "ms_median": med, "ms_min": mn, "algorithmic_bytes": algo,
                              "algorithmic_GBps": algo / (med * 1e-3) / 1e9,
                              "frac_of_hbm_peak": algo / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "roundtrip_and_horner_check": bool(ok)}), flush=True)
            del d


if __name__ == "__main__":
    main()
