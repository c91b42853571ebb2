#!/usr/bin/env python3
"""Latency of the headline operation (CKKS N=16384, 6x50-bit: multiply + relinearize + rescale, fused entry) at small batches.
usage: python tools/latency_small_batch.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
from tools.bench_configs import residues, timed


def main():
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(3)
    n, L = 16384, 5
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    plan = pkg.Plan(dev, 14, q)
    keys = [residues(pkg, (2,), q, n, dev, gen) for _ in range(L)]
    res = {}
    for B in (1, 8, 64, 256, 1024):
        x, y = residues(pkg, (B, 2), q[:L], n, dev, gen), residues(pkg, (B, 2), q[:L], n, dev, gen)
        out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
        t = timed(lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), 50 if B < 256 else 10)
        res["batch_%d" % B] = {"latency_us": round(t * 1e6, 1), "ops_per_s": round(B / t, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
