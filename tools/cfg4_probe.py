#!/usr/bin/env python3
"""Probe: cfg4 (BFV N=32768 L=10 multiply + relinearize) with chunks alternating on S streams (own plan / workspace per stream)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, log_n, L, t_plain = 32768, 15, 10, 1032193
q = pkg.capi.coeff_modulus_create(n, [50] * 11)
gen = torch.Generator(device=dev).manual_seed(1)
def residues(prefix, mods):
    out = torch.empty(tuple(prefix) + (len(mods), n), dtype=torch.int64, device=dev)
    for l, m in enumerate(mods):
        out[..., l, :] = torch.randint(0, m, tuple(prefix) + (n,), dtype=torch.int64, device=dev, generator=gen)
    return out
TOTAL = 256
x, y = residues((TOTAL, 2), q[:L]), residues((TOTAL, 2), q[:L])
keys = [residues((2,), q) for _ in range(L)]
out = torch.empty((TOTAL, 2, L, n), dtype=torch.int64, device=dev)
ref = torch.empty_like(out)
NS = 4
plans = [pkg.Plan(dev, log_n, q) for _ in range(NS)]
behz = [pkg.Behz(p, L, t_plain) for p in plans]
prods = [torch.empty((128, 3, L, n), dtype=torch.int64, device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
def run(C, S, dst):
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    for s in range(S): streams[s].wait_event(ev)
    for i, lo in enumerate(range(0, TOTAL, C)):
        s = i % S
        with torch.cuda.stream(streams[s]):
            behz[s].multiply(x[lo:lo + C], 2, y[lo:lo + C], 2, out=prods[s][:C])
            plans[s].relinearize(L, prods[s][:C], keys, out=dst[lo:lo + C], is_ckks=False, is_ntt_form=False)
    for s in range(S):
        e = torch.cuda.Event(); e.record(streams[s]); main.wait_event(e)
run(64, 1, ref); torch.cuda.synchronize()
for C, S in ((64, 1), (64, 2), (32, 2), (128, 2), (64, 3), (64, 4), (32, 4), (64, 1)):
    out.zero_(); run(C, S, out); torch.cuda.synchronize()
    ok = bool(torch.equal(out, ref))
    t0 = time.perf_counter()
    for _ in range(3): run(C, S, out)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 3
    print("chunk %3d streams %d: %8.1f ops/s identical=%s" % (C, S, TOTAL / t, ok), flush=True)
