#!/usr/bin/env python3
"""Probe: the fused CKKS chain over a batch of 1024 pairs, evaluated as chunks of C items round-robin on S streams (each stream with
its own plan / workspace), against one launch sequence over the whole batch.  Small chunks keep the chain's intermediates in the
256 MB Infinity Cache; several streams overlap the FP64-bound inner product of one chunk with the memory-bound transforms of another."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, log_n, L, B = 16384, 14, 5, 1024
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
ref = torch.empty_like(out)
plans = [pkg.Plan(dev, log_n, q) for _ in range(8)]
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
plans[0].ckks_multiply_relinearize_rescale(L, a, b, keys, out=ref)
torch.cuda.synchronize()


def run(C, S):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event()
    ev.record(main)
    for s in range(S):
        streams[s].wait_event(ev)
    for i, lo in enumerate(range(0, B, C)):
        s = i % S
        with torch.cuda.stream(streams[s]):
            plans[s].ckks_multiply_relinearize_rescale(L, a[lo:lo + C], b[lo:lo + C], keys, out=out[lo:lo + C])
    for s in range(S):
        e = torch.cuda.Event()
        e.record(streams[s])
        main.wait_event(e)


res = {}
for C, S in ((1024, 1), (512, 2), (256, 1), (256, 2), (256, 4), (128, 1), (128, 2), (128, 4), (128, 8), (64, 1), (64, 2), (64, 4), (64, 8), (32, 4), (32, 8), (16, 8)):
    out.zero_()
    run(C, S)
    torch.cuda.synchronize()
    ok = bool(torch.equal(out, ref))
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        run(C, S)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    res["chunk%d_streams%d" % (C, S)] = {"ops_per_s": round(B / t, 1), "ms": round(t * 1e3, 3), "identical": ok}
    print("chunk %4d streams %d: %9.1f ops/s  %.3f ms  identical=%s" % (C, S, B / t, t * 1e3, ok), flush=True)
json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "chunk_probe.json"), "w"), indent=1)
