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
# MIXED_SHAPE="n:bits,...:L:batch" selects another chain (default: the tool's default chain)
_shape = os.environ.get("MIXED_SHAPE", "8192:60,40,40,60:3:1024").split(":")
n, L, B = int(_shape[0]), int(_shape[2]), int(_shape[3])
log_n = n.bit_length() - 1
q = pkg.capi.coeff_modulus_create(n, [int(b) for b in _shape[1].split(",")])
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
if os.environ.get("MIXED_CHAIN", "") == "1":       # the fused entry (multiply + relinearize + rescale) on the same chain
    x, y = residues((B, 2), q[:L]), residues((B, 2), q[:L])
    o2 = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
    for _ in range(20):
        plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=o2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=o2)
    torch.cuda.synchronize()
    print("mul+relin+rescale ops/s", B * 20 / (time.perf_counter() - t0))
