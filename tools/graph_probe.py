#!/usr/bin/env python3
"""What a hipGraph would buy the single-object fused call: one multiply + relinearize + rescale of ONE CKKS N = 16384 ciphertext pair
(ten launches) issued eagerly with a stream wait after every op, against the same launch sequence captured once (torch.cuda.CUDAGraph = hipGraph) and replayed.
  python tools/graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, L = 16384, 5
q = pkg.capi.coeff_modulus_create(n, [50] * 6)
plan = pkg.Plan(dev, 14, q)
gen = torch.Generator(device=dev).manual_seed(3)
x, y = bench.uniform_residues(torch, (1, 2), q[:L], n, dev, gen), bench.uniform_residues(torch, (1, 2), q[:L], n, dev, gen)
keys = [bench.uniform_residues(torch, (2,), q, n, dev, gen) for _ in range(L)]
out = torch.empty((1, 2, L - 1, n), dtype=torch.int64, device=dev)


def eager(reps):
    for _ in range(300):
        plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


ref = plan.ckks_multiply_relinearize_rescale(L, x, y, keys).clone()
t_eager = eager(2000)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)
out.zero_()
for _ in range(300):
    g.replay()
    torch.cuda.synchronize()
assert torch.equal(out, ref), "graph replay differs"
t0 = time.perf_counter()
for _ in range(2000):
    g.replay()
    torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / 2000
print("single fused op, eager launches + stream wait: %.1f us; captured hipGraph replay + wait: %.1f us" % (t_eager * 1e6, t_graph * 1e6))
