#!/usr/bin/env python3
"""Chains with moduli of 2^50 and more: relinearize and multiply + relinearize + rescale throughput, with the inner-product launch timed by
the library's kernel timer.  --ab: the fused entry against the three-call composition inside it (plan option TROYN_MRR_MIXED=0).
  python tools/bench_mixed.py [--ab]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench


def run(pkg, dev, n, bits, L, B, reps=10, opts=()):
    log_n = n.bit_length() - 1
    gen = torch.Generator(device=dev).manual_seed(7)
    q = pkg.capi.coeff_modulus_create(n, bits)
    K = len(q)
    plan = pkg.Plan(dev, log_n, q)
    for name, val in opts:
        plan.set_option(name, val)
    prod = bench.uniform_residues(torch, (B, 3), q[:L], n, dev, gen)
    x, y = bench.uniform_residues(torch, (B, 2), q[:L], n, dev, gen), bench.uniform_residues(torch, (B, 2), q[:L], n, dev, gen)
    keys = [bench.uniform_residues(torch, (2,), q, n, dev, gen) for _ in range(L)]
    out2 = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
    r = {"chain": bits, "n": n, "L": L, "batch": B}
    with bench.KernelTimer(pkg, plan.lib, bench.TIMER_KS) as kt:
        tr = bench.timed(torch, lambda: plan.relinearize(L, prod, keys, out=out2, is_ckks=True, is_ntt_form=True), reps)
        ks_ms, ks_n = kt.read()
    r["relinearize_ops_per_s"] = round(B / tr, 1)
    if ks_n:
        r["inner_product_ms"] = round(ks_ms / ks_n, 4)
        r["inner_product_hbm_frac"] = round(bench.ksmac_alg_bytes(B, n, L, True) / (ks_ms / ks_n * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, 4)
    if L >= 2:
        t = bench.timed(torch, lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), reps)
        r["ckks_mul_relin_rescale_ops_per_s"] = round(B / t, 1)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ab", action="store_true")
    a = ap.parse_args()
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    shapes = [(8192, [60, 40, 40, 60], 3, 1024), (16384, [60, 50, 50, 50, 50, 60], 5, 512), (8192, [60, 60, 60, 60], 3, 1024), (32768, [60, 50, 50, 60], 3, 256)]
    for n, bits, L, B in shapes:
        modes = [("default", ())] + ([("three_calls_inside_the_entry", (("TROYN_MRR_MIXED", "0"),))] if a.ab else [])
        for name, opts in modes:
            r = run(pkg, dev, n, bits, L, B, opts=opts)
            r["mode"] = name
            print(json.dumps(r), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
