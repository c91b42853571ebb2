#!/usr/bin/env python3
"""Micro-benchmark of troyn_ntt: python tools/ntt_bench.py [--logn 14] [--limbs 5] [--bits 50] [--batch 256] [--rows 6]
Reports algorithmic GB/s (16*N bytes per limb-polynomial) for forward (out-of-place and in-place) and inverse."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--logn", type=int, default=14)
    ap.add_argument("--limbs", type=int, default=5)
    ap.add_argument("--bits", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rows", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    pkg = entry.load_package()
    n = 1 << a.logn
    q = pkg.capi.coeff_modulus_create(n, [a.bits] * (a.limbs + 1))
    plan = pkg.Plan("cuda:0", a.logn, q)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.empty((a.batch, a.rows, a.limbs, n), dtype=torch.int64, device="cuda")
    for l in range(a.limbs):
        x[:, :, l, :] = torch.randint(0, q[l], (a.batch, a.rows, n), dtype=torch.int64, device="cuda", generator=g)
    y = torch.empty_like(x)
    lp = a.batch * a.rows * a.limbs
    nbytes = 16.0 * n * lp

    def timeit(name, fn):
        if a.only and a.only != name:
            return
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        print("%-22s logN=%d limb-polys=%d  %8.3f ms  %8.1f GB/s algorithmic  (%.2f us per limb per CU)" % (
            name, a.logn, lp, ms, nbytes / ms / 1e6, ms * 1e3 * 256 / lp))

    timeit("fwd_out_of_place", lambda: plan.ntt(x, a.rows, a.limbs, out=y))
    timeit("fwd_in_place", lambda: plan.ntt(y, a.rows, a.limbs))
    timeit("inv_out_of_place", lambda: plan.ntt(x, a.rows, a.limbs, inverse=True, out=y))
    timeit("inv_in_place", lambda: plan.ntt(y, a.rows, a.limbs, inverse=True))


if __name__ == "__main__":
    main()
