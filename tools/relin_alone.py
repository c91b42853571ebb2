#!/usr/bin/env python3
"""Evaluator::relinearize alone (CKKS N = 16384, 6 x 50-bit, NTT form) at a large batch: which launches serve it (run under rocprofv3 --kernel-trace --stats).
python tools/relin_alone.py [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n, L = 16384, 5
gen = torch.Generator(device=dev).manual_seed(5)
q = pkg.capi.coeff_modulus_create(n, [50] * 6)
plan = pkg.Plan(dev, 14, q)
prod = bench.uniform_residues(torch, (B, 3), q[:L], n, dev, gen)
keys = [bench.uniform_residues(torch, (2,), q, n, dev, gen) for _ in range(L)]
out = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
t = bench.timed(torch, lambda: plan.relinearize(L, prod, keys, out=out, is_ckks=True, is_ntt_form=True), 20)
print(json.dumps({"batch": B, "relinearize_ops_per_s": round(B / t, 1), "ms": round(t * 1e3, 4)}))
