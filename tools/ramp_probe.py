#!/usr/bin/env python3
"""Probe: how long the clocks take to come up after an idle gap.  The fused CKKS chain over 256 pairs (about 1 ms per call) is called 300
times back to back after the device idled for `gap` seconds; every call is bracketed by events.  Prints the duration of call i."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, log_n, L, B = 16384, 14, 5, 256
q = pkg.capi.coeff_modulus_create(n, [50] * 6)
gen = torch.Generator(device=dev).manual_seed(1)


def residues(shape_prefix, mods):
    out = torch.empty(tuple(shape_prefix) + (len(mods), n), dtype=torch.int64, device=dev)
    for l, m in enumerate(mods):
        out[..., l, :] = torch.randint(0, m, tuple(shape_prefix) + (n,), dtype=torch.int64, device=dev, generator=gen)
    return out


a, b = residues((B, 2), q[:L]), residues((B, 2), q[:L])
keys = [residues((2,), q) for _ in range(L)]
out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
plan = pkg.Plan(dev, log_n, q)
os.environ["TROYN_MRR_CHUNK"] = "0"
for _ in range(3):
    plan.ckks_multiply_relinearize_rescale(L, a, b, keys, out=out)
torch.cuda.synchronize()
res = {}
for gap in (0.0, 0.02, 0.1, 1.0, 5.0):
    time.sleep(gap)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(301)]
    ev[0].record()
    for i in range(300):
        plan.ckks_multiply_relinearize_rescale(L, a, b, keys, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(300)]
    pick = [0, 1, 2, 5, 10, 20, 40, 60, 80, 100, 150, 200, 299]
    res[gap] = {i: round(ms[i], 3) for i in pick}
    cum = 0.0
    settled = None
    tail = sum(ms[250:]) / 50
    for i, v in enumerate(ms):
        cum += v
        if settled is None and v < 1.02 * tail:
            settled = (i, round(cum, 1))
    print(f"gap {gap:4.2f} s: tail {tail:.3f} ms/call; first call within 2 % of it: index {settled[0]} after {settled[1]} ms; mean of first 20: {sum(ms[:20]) / 20:.3f}", flush=True)
    print("   ", res[gap], flush=True)
