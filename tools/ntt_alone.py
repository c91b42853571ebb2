#!/usr/bin/env python3
"""The transform alone (BASELINE's second figure, NTT GB/s against HBM): forward and inverse, out of place, per ring size.
python tools/ntt_alone.py [log_n ...]   (run under rocprofv3 --kernel-trace --stats to see which kernels serve it)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
dev = torch.device("cuda", 0)


def run(log_n, bits, L, B):
    n = 1 << log_n
    gen = torch.Generator(device=dev).manual_seed(11)
    q = pkg.capi.coeff_modulus_create(n, bits)
    plan = pkg.Plan(dev, log_n, q)
    for a in sys.argv[1:]:
        if "=" in a:
            plan.set_option(*a.split("=", 1))
    x = bench.uniform_residues(torch, (B, 2), q[:L], n, dev, gen)
    y, z = torch.empty_like(x), torch.empty_like(x)
    tf = bench.timed(torch, lambda: plan.ntt(x, 2, L, out=y), 20)
    ti = bench.timed(torch, lambda: plan.ntt(y, 2, L, out=z, inverse=True), 20)
    assert torch.equal(z, x)
    tp = bench.timed(torch, lambda: plan.ntt(x, 2, L), 20)          # in place (x is scratch from here on)
    byts = 16.0 * n * L * 2 * B
    print(json.dumps({"n": n, "bits": bits, "L": L, "batch": B, "limb_polys": 2 * L * B, "ntt_GBps": round(byts / tf / 1e9, 1), "intt_GBps": round(byts / ti / 1e9, 1),
                      "ntt_in_place_GBps": round(byts / tp / 1e9, 1), "ntt_us": round(tf * 1e6, 1), "intt_us": round(ti * 1e6, 1)}), flush=True)


if __name__ == "__main__":
    sel = [int(a) for a in sys.argv[1:] if a.isdigit()] or [13, 14, 15]
    if 13 in sel:
        run(13, [40, 40, 40], 2, 2048)
    if 14 in sel:
        run(14, [50] * 6, 5, 512)
    if 15 in sel:
        run(15, [50] * 11, 10, 64)
