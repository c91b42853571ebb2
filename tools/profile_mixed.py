#!/usr/bin/env python3
"""Kernel-trace workload: relinearize on the reference bench tool's default chain N = 8192 {60,40,40,60} (L = 3), batch 1024, 30 calls.
  cd /tmp && rocprofv3 --kernel-trace --stats -d <dir> -o mixed -- python3 <repo>/tools/profile_mixed.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, log_n, L, B = 8192, 13, 3, 1024
q = pkg.capi.coeff_modulus_create(n, [60, 40, 40, 60])
gen = torch.Generator(device=dev).manual_seed(1)


def residues(shape_prefix, mods):
    out = torch.empty(tuple(shape_prefix) + (len(mods), n), dtype=torch.int64, device=dev)
    for l, m in enumerate(mods):
        out[..., l, :] = torch.randint(0, m, tuple(shape_prefix) + (n,), dtype=torch.int64, device=dev, generator=gen)
    return out


ct3 = residues((B, 3), q[:L])
keys = [residues((2,), q) for _ in range(L)]
out = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
plan = pkg.Plan(dev, log_n, q)
for _ in range(40):
    plan.relinearize(L, ct3, keys, out=out, is_ckks=True, is_ntt_form=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    plan.relinearize(L, ct3, keys, out=out, is_ckks=True, is_ntt_form=True)
torch.cuda.synchronize()
print("relinearize ops/s", B * 30 / (time.perf_counter() - t0))
