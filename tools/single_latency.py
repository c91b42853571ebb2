#!/usr/bin/env python3
"""Latency of ONE ciphertext through the C-ABI (ctypes layer, a stream wait after every call): fused multiply + relinearize + rescale and the three
calls, for chains of one and of both arithmetic classes.  python tools/single_latency.py [OPTION=value ...]   (plan options, e.g. TROYN_KS_MAC=split)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
dev = torch.device("cuda", 0)
OPTS = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]


def run(n, bits, L, B=1):
    gen = torch.Generator(device=dev).manual_seed(7)
    q = pkg.capi.coeff_modulus_create(n, bits)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    for k, v in OPTS:
        plan.set_option(k, v)
    x, y = bench.uniform_residues(torch, (B, 2), q[:L], n, dev, gen), bench.uniform_residues(torch, (B, 2), q[:L], n, dev, gen)
    keys = [bench.uniform_residues(torch, (2,), q, n, dev, gen) for _ in range(L)]
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)
    rel = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)

    def lat(f, reps=300):
        for _ in range(30):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    r = {"n": n, "bits": bits, "L": L, "batch": B, "options": dict(OPTS)}
    r["fused_us"] = round(lat(lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)), 1)
    r["multiply_us"] = round(lat(lambda: plan.dyadic_convolute(x, 2, y, 2, L, out=prod)), 1)
    r["relinearize_us"] = round(lat(lambda: plan.relinearize(L, prod, keys, out=rel, is_ckks=True, is_ntt_form=True)), 1)
    r["rescale_us"] = round(lat(lambda: plan.divide_and_round_q_last_ntt(L, rel, 2, out=out)), 1)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    run(8192, [60, 40, 40, 60], 3)
    run(8192, [40, 40, 40, 40], 3)
    run(16384, [50] * 6, 5)
    run(16384, [60, 50, 50, 50, 50, 60], 5)
    run(32768, [50] * 6, 5)
