#!/usr/bin/env python3
"""What a batch of B single-object calls costs when it runs as ONE launch sequence (the call-combining layer of the C++ mirror hands the
library exactly this): CKKS N = 16384, 6 x 50-bit, B = 1 .. 64, the three calls and the fused entry, digit-parallel inner product forced
off / on / default (plan option TROYN_KS_SPLIT).
usage: python tools/small_batch_sweep.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import time

from tools.bench_configs import residues


def timed(fn, reps):
    # 50 ms of the loop's own work first: the clocks ramp for 20-25 ms after an idle gap (tools/ramp_probe.py)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    n, L = 16384, 5
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    plan = pkg.Plan(dev, 14, q)
    keys = [residues(pkg, (2,), q, n, dev, gen) for _ in range(L)]
    res = {}
    for B in (1, 2, 4, 8, 16, 32, 64):
        x, y = residues(pkg, (B, 2), q[:L], n, dev, gen), residues(pkg, (B, 2), q[:L], n, dev, gen)
        out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
        prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)
        relin = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
        row = {}
        for split in ("default", "0", "1"):
            plan.set_option("TROYN_KS_SPLIT", None if split == "default" else split)
            t_f = timed(lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), 100)
            t_r = timed(lambda: plan.relinearize(L, prod, keys, out=relin, is_ckks=True, is_ntt_form=True), 100)
            row["split_" + split] = {"fused_us": round(t_f * 1e6, 1), "relinearize_us": round(t_r * 1e6, 1)}
        plan.set_option("TROYN_KS_SPLIT", None)
        t_m = timed(lambda: plan.dyadic_convolute(x, 2, y, 2, L, out=prod), 100)
        t_s = timed(lambda: plan.divide_and_round_q_last_ntt(L, relin, 2, out=out), 100)
        row["multiply_us"] = round(t_m * 1e6, 1)
        row["rescale_us"] = round(t_s * 1e6, 1)
        res["batch_%d" % B] = row
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
