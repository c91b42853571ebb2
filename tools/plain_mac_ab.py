#!/usr/bin/env python3
"""A/B of the ct x pt multiply-accumulate of BASELINE config 5 (32 x 512 weight plaintexts, N = 8192 {60,40,40,60}, L = 3) under two weight layouts:
the reference's (every weight a stand-alone plaintext [L][N]) and a packed one ([destination][limb][chunk of 512][term][512]: what a workgroup
reads is contiguous).  Both results are compared word for word.  VERDICT r03 item 7; output -> profiles/r04_plain_mac_ab.txt."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as entry

pkg = entry.load_package()
dev = torch.device("cuda", 0)
n, I, J, L = 8192, 32, 512, 3
q = pkg.capi.coeff_modulus_create(n, [60, 40, 40, 60])
plan = pkg.Plan(dev, 13, q)
gen = torch.Generator(device=dev).manual_seed(11)
av = torch.empty((I, 2, L, n), dtype=torch.int64, device=dev)
w = torch.empty((I, J, L, n), dtype=torch.int64, device=dev)
for l, m in enumerate(q[:L]):
    av[:, :, l, :].random_(0, m, generator=gen)
    w[:, :, l, :].random_(0, m, generator=gen)
# packed: [J][L][n/512][I][512]
wp = w.view(I, J, L, n // 512, 512).permute(1, 2, 3, 0, 4).contiguous()
out = torch.empty((J, 2, L, n), dtype=torch.int64, device=dev)
out2 = torch.empty_like(out)
terms = I * J
arr = lambda v: (C.c_void_p * terms)(*v)
cts = [av[i].data_ptr() for j in range(J) for i in range(I)]
pts = [w[i, j].data_ptr() for j in range(J) for i in range(I)]
ppk = [wp[j].data_ptr() for j in range(J) for i in range(I)]          # only the first term's pointer of a destination is used
d1 = [out[j].data_ptr() for j in range(J) for i in range(I)]
d2 = [out2[j].data_ptr() for j in range(J) for i in range(I)]
ws = torch.empty(int(plan.lib.troyn_multiply_plain_accumulate_workspace_bytes(terms)), dtype=torch.uint8, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
TIMER = 3      # TROYN_TIMER_PLAIN_MAC


def run(pt, dst, env):
    if env:
        os.environ["TROYN_PLAIN_MAC"] = env
    else:
        os.environ.pop("TROYN_PLAIN_MAC", None)
    pkg.capi.check(plan.lib.troyn_multiply_plain_accumulate(plan.h, 0, L, 2, arr(cts), arr(pt), arr(dst), terms, 1, C.c_void_p(ws.data_ptr()), ws.numel(), stream))


import bench
out3 = torch.empty_like(out)
d3 = [out3[j].data_ptr() for j in range(J) for i in range(I)]
out4 = torch.empty_like(out)
d4 = [out4[j].data_ptr() for j in range(J) for i in range(I)]
for name, pt, dst, env in (("one destination / workgroup", pts, d3, "single"), ("two destinations / workgroup", pts, d4, "dual"), ("four destinations / workgroup", pts, d1, None),
                           ("packed layout", ppk, d2, "packed"),
                           ("one destination / workgroup", pts, d3, "single"), ("two destinations / workgroup", pts, d4, "dual"), ("four destinations / workgroup", pts, d1, None),
                           ("packed layout", ppk, d2, "packed")):
    for _ in range(30):
        run(pt, dst, env)
    torch.cuda.synchronize()
    with bench.KernelTimer(pkg, plan.lib, bench.TIMER_PLAIN_MAC) as kt:
        for _ in range(20):
            run(pt, dst, env)
        torch.cuda.synchronize()
        ms, cnt = kt.read()
    alg = terms * L * n * 8.0 + I * 2 * L * n * 8.0 + J * 2 * L * n * 8.0
    print("%-30s launch %.4f ms  %.1f GB/s  (%.3f of 8 TB/s)" % (name, ms / cnt, alg / (ms / cnt * 1e-3) / 1e9, alg / (ms / cnt * 1e-3) / 8e12))
print("identical results:", bool(torch.equal(out, out2)) and bool(torch.equal(out, out3)) and bool(torch.equal(out, out4)))
