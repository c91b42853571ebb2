#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations (bench.py covers configs[2], the headline):
  cfg2  BFV  N=8192  L=2      NTT(a), NTT(b) + dyadic 2x2->3 + INTT (the "NTT+dyadic-mul+INTT" pipeline), ciphertexts/s
  cfg3  CKKS N=16384 L=5 K=6  relinearize alone, ops/s
  cfg4  BFV  N=32768 L=10     BEHZ multiply (+ relinearize) of independent ciphertext pairs, ops/s
  cfg5  BFV  N=8192 {60,40,40,60}  ct x pt multiply-accumulate of the matmul application, weight plaintexts/s
Synthetic uniform residues.  usage: python tools/bench_configs.py [--only cfg4]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry


def residues(pkg, shape_prefix, q, n, dev, gen):
    out = torch.empty(tuple(shape_prefix) + (len(q), n), dtype=torch.int64, device=dev)
    for l, m in enumerate(q):
        out[..., l, :] = torch.randint(0, m, tuple(shape_prefix) + (n,), dtype=torch.int64, device=dev, generator=gen)
    return out


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    pkg = entry.load_package()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(7)
    res = {}

    if a.only in ("", "cfg2"):
        n, B = 8192, 2048
        q = pkg.capi.coeff_modulus_create(n, [40, 40, 40])
        L = 2
        plan = pkg.Plan(dev, 13, q)
        x, y = residues(pkg, (B, 2), q[:L], n, dev, gen), residues(pkg, (B, 2), q[:L], n, dev, gen)
        xn, yn = torch.empty_like(x), torch.empty_like(y)
        prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)

        def step():
            plan.ntt(x, 2, L, out=xn); plan.ntt(y, 2, L, out=yn)
            plan.dyadic_convolute(xn, 2, yn, 2, L, out=prod)
            plan.ntt(prod, 3, L, inverse=True)
        t = timed(step, a.reps)
        alg = (4 * 16 + 56 + 3 * 16) * n * L * B          # 168*N*L bytes per product, unfused
        res["cfg2"] = {"what": "BFV N=8192 L=2: 4 NTT + dyadic 2x2->3 + 3 INTT per ciphertext pair", "batch": B,
                       "ciphertext_products_per_s": round(B / t, 1), "algorithmic_GBps": round(alg / t / 1e9, 1)}

    if a.only in ("", "cfg3"):
        n, B = 16384, 256
        q = pkg.capi.coeff_modulus_create(n, [50] * 6)
        K, L = 6, 5
        plan = pkg.Plan(dev, 14, q)
        ct3 = residues(pkg, (B, 3), q[:L], n, dev, gen)
        keys = [residues(pkg, (2,), q, n, dev, gen) for _ in range(L)]
        out = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
        t = timed(lambda: plan.relinearize(L, ct3, keys, out=out, is_ckks=True, is_ntt_form=True), a.reps)
        res["cfg3_relinearize"] = {"what": "CKKS N=16384 L=5 K=6 relinearize (3 -> 2 polynomials)", "batch": B, "ops_per_s": round(B / t, 1)}

    if a.only in ("", "cfg4"):
        n, B = 32768, 64
        q = pkg.capi.coeff_modulus_create(n, [50] * 11)
        K, L, t_plain = 11, 10, 1032193
        plan = pkg.Plan(dev, 15, q)
        behz = pkg.Behz(plan, L, t_plain)
        x, y = residues(pkg, (B, 2), q[:L], n, dev, gen), residues(pkg, (B, 2), q[:L], n, dev, gen)
        prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)
        keys = [residues(pkg, (2,), q, n, dev, gen) for _ in range(L)]
        out = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
        tm = timed(lambda: behz.multiply(x, 2, y, 2, out=prod), max(3, a.reps // 2))
        tr = timed(lambda: plan.relinearize(L, prod, keys, out=out, is_ckks=False, is_ntt_form=False), max(3, a.reps // 2))
        ts = timed(lambda: behz.multiply(x, 2, x, 2, out=prod), max(3, a.reps // 2))
        res["cfg4"] = {"what": "BFV N=32768 L=10: BEHZ multiply, then relinearize", "batch": B,
                       "multiply_ops_per_s": round(B / tm, 1), "square_ops_per_s": round(B / ts, 1), "relinearize_ops_per_s": round(B / tr, 1),
                       "multiply_relinearize_ops_per_s": round(B / (tm + tr), 1)}

    if a.only in ("", "bfv_mul"):
        # BEHZ multiply of independent ciphertext pairs at the whole-limb sizes (cfg2's parameters, the matmul app's chain, a 50-bit chain)
        for name, n, bits, L, B in (("N8192_3x40", 8192, [40] * 3, 2, 1024), ("N8192_60_40_40_60", 8192, [60, 40, 40, 60], 3, 512),
                                    ("N4096_3x36", 4096, [36] * 3, 2, 2048), ("N8192_5x50", 8192, [50] * 5, 4, 512),
                                    ("N16384_6x50", 16384, [50] * 6, 5, 256)):
            q = pkg.capi.coeff_modulus_create(n, bits)
            plan = pkg.Plan(dev, n.bit_length() - 1, q)
            behz = pkg.Behz(plan, L, 1032193)
            x, y = residues(pkg, (B, 2), q[:L], n, dev, gen), residues(pkg, (B, 2), q[:L], n, dev, gen)
            prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)
            tm = timed(lambda: behz.multiply(x, 2, y, 2, out=prod), a.reps)
            res["bfv_mul_" + name] = {"batch": B, "multiply_ops_per_s": round(B / tm, 1)}

    if a.only in ("", "cfg5"):
        # 512x512x512 matmul packed into N=8192 slots: the kernel is ret[b][j] = sum_i a[b][i] (.) w[i][j]
        import ctypes as C
        n, I, J, Bt = 8192, 256, 64, 1                      # 16 384 weight plaintexts
        q = pkg.capi.coeff_modulus_create(n, [60, 40, 40, 60])
        L = 3
        plan = pkg.Plan(dev, 13, q)
        av = residues(pkg, (Bt, I, 2), q[:L], n, dev, gen)
        w = torch.empty((I, J, L, n), dtype=torch.int64, device=dev)
        for l, m in enumerate(q[:L]):
            w[:, :, l, :].random_(0, m, generator=gen)
        out = torch.empty((Bt, J, 2, L, n), dtype=torch.int64, device=dev)
        cts, pts, dsts = [], [], []
        for i in range(I):
            for j in range(J):
                for b in range(Bt):
                    cts.append(av[b, i].data_ptr()); pts.append(w[i, j].data_ptr()); dsts.append(out[b, j].data_ptr())
        terms = len(cts)
        arr = lambda v: (C.c_void_p * terms)(*v)
        ca, pa, da = arr(cts), arr(pts), arr(dsts)          # pointer tables built once, as a matmul call would
        nbytes = plan.lib.troyn_multiply_plain_accumulate_workspace_bytes(terms)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def mac():
            pkg.capi.check(plan.lib.troyn_multiply_plain_accumulate(plan.h, 0, L, 2, ca, pa, da, terms, 1, C.c_void_p(ws.data_ptr()), ws.numel(), stream))
        t = timed(mac, a.reps)
        alg = terms * (2 + 1) * L * n * 8 + J * Bt * 2 * L * n * 8        # every ct and pt word of a term read once, outputs written once
        res["cfg5_mac"] = {"what": "ct x pt multiply-accumulate, N=8192 L=3: %d x %d weight plaintexts, %d input block(s)" % (I, J, Bt),
                           "terms_per_s": round(terms / t, 1), "algorithmic_GBps": round(alg / t / 1e9, 1), "ms": round(t * 1e3, 3),
                           "weight_GBps": round(terms * L * n * 8 / t / 1e9, 1)}

    if a.only in ("", "cfg5", "cfg5_e2e"):
        import subprocess
        drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "cpp", "matmul_driver")
        res["cfg5_e2e"] = {"what": "BFV 512x512x512 y = x*w + s through MatmulHelper (N=8192, {60,40,40,60}, t=2^21), the flow of examples/10_bfv_matmul.cu: "
                                   "encrypted inputs x plaintext weights, inputs/outputs through their wire formats"}
        for tag, flags in (("plain_outputs", ["0", "0"]), ("mod_switched_outputs", ["0", "1"]), ("packed_outputs", ["1", "1"])):
            r = subprocess.run([drv, "512", "512", "512", "5"] + flags, capture_output=True, text=True, timeout=900)
            lines = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
            ms = dict(zip(lines["ms"][0::2], [float(v) for v in lines["ms"][1::2]]))
            compute = ms["encrypt_inputs"] + ms["matmul_repeat"] + ms["mod_switch"] + ms["pack"] + ms["add_bias"] + ms["decrypt"]
            res["cfg5_e2e"][tag] = {"block": lines["block"][:3], "objects": " ".join(lines["objects"]), "wire_bytes": " ".join(lines["bytes"]), "ms": ms,
                                    "latency_ms_encrypt_to_decrypt_without_wire": round(compute, 2),
                                    "ms_repeat": dict(zip(lines["ms_repeat"][0::2], [float(v) for v in lines["ms_repeat"][1::2]])),
                                    "correct": "OK" in r.stdout}
        # CPU baseline of the matmul core (the oracle's multiply_plain_ntt + add, one thread), on a bounded sample of terms
        import numpy as np
        O = entry.load_oracle()
        q = [int(v) for v in O.coeff_modulus_create(8192, [60, 40, 40, 60])]
        ctx = O.Context("bfv", 8192, q, 1 << 21)
        ct, pt = ctx.random_ct(1, 2, 3), ctx.random_ct(2, 1, 3)[0]
        acc = np.zeros_like(ct)
        sample, t0 = 300, time.perf_counter()
        for _ in range(sample):
            term = ctx.multiply_plain_ntt(3, ct, pt)
            for l in range(3):
                acc[:, l] = (acc[:, l] + term[:, l]) % np.uint64(q[l])
        per_term = (time.perf_counter() - t0) / sample
        res["cfg5_e2e"]["cpu_matmul_core_ms_estimate"] = round(per_term * 16384 * 1e3, 1)
        res["cfg5_e2e"]["cpu_sample"] = "%d of 16384 multiply_plain_ntt+add terms on one host thread (oracle), extrapolated" % sample
        # the other phases of the packed flow on one host thread, each timed on a few objects and multiplied out:
        # encryption of the 32 input blocks, the NTTs around the product, the packing of 32 groups of 16, decryption of 32
        rng = O.Rng(3)
        sk = ctx.secret_key(rng)
        pk = ctx.public_key(rng, sk)
        msg = O.fill_uniform(9, 1 << 21, 8192)

        def cpu_timed(fn, reps):
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            return (time.perf_counter() - t0) / reps, out
        t_enc, c = cpu_timed(lambda: ctx.encrypt_asymmetric_bfv(rng, pk, msg), 3)
        t_ntt, _ = cpu_timed(lambda: ctx.from_ntt(ctx.to_ntt(c, 2, 3), 2, 3), 3)            # one forward + one inverse transform of a ciphertext
        t_dec, _ = cpu_timed(lambda: ctx.decrypt_bfv(sk, c), 3)
        keys = {(8192 // 16) * (1 << (k + 1)) + 1: ctx.random_keys(50 + k, 2) for k in range(4)}
        low = ctx.mod_switch_scale_to_next(3, c)
        t_pack, _ = cpu_timed(lambda: ctx.pack_rlwe_ciphertexts(2, [low] * 16, keys, 2 * 8192 - 15, 16, 1), 1)
        phases = {"encrypt_32_inputs": t_enc * 32, "ntt_32_inputs_intt_512_outputs": t_ntt / 2 * (32 + 512), "matmul_core": per_term * 16384,
                  "pack_32_groups_of_16": t_pack * 32, "decrypt_32_outputs": t_dec * 32}
        res["cfg5_e2e"]["cpu_packed_flow_ms_estimate"] = {k: round(v * 1e3, 1) for k, v in phases.items()}
        res["cfg5_e2e"]["cpu_packed_flow_ms_estimate"]["total"] = round(sum(phases.values()) * 1e3, 1)
        res["cfg5_e2e"]["cpu_packed_flow_note"] = "oracle (plain C restatement of the reference's host branches), one host thread, per-object timings multiplied out"

    if a.only in ("", "u64"):
        # the integer-butterfly policy (ArithU64): any modulus >= 2^50, e.g. the reference's troybench default {60,40,40,60}
        # (test/bench/he_operations.cu:19-33) at N = 8192 -- every launch whose table slice contains a 60-bit prime takes it
        n, B = 8192, 1024
        q = pkg.capi.coeff_modulus_create(n, [60, 40, 40, 60])
        K, L = 4, 3
        plan = pkg.Plan(dev, 13, q)
        x = residues(pkg, (B, 2), q[:L], n, dev, gen)
        xn = torch.empty_like(x)
        t_f = timed(lambda: plan.ntt(x, 2, L, out=xn), a.reps)
        t_i = timed(lambda: plan.ntt(xn, 2, L, inverse=True, out=x), a.reps)
        nb = 16.0 * n * B * 2 * L
        keys = [residues(pkg, (2,), q, n, dev, gen) for _ in range(L)]
        ct3 = residues(pkg, (B, 3), q[:L], n, dev, gen)
        out = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
        t_rn = timed(lambda: plan.relinearize(L, ct3, keys, out=out, is_ckks=True, is_ntt_form=True), a.reps)
        t_rc = timed(lambda: plan.relinearize(L, ct3, keys, out=out, is_ckks=False, is_ntt_form=False), a.reps)
        # same shape with 40-bit primes only (FP64 policy) for comparison
        q2 = pkg.capi.coeff_modulus_create(n, [40, 40, 40, 40])
        plan2 = pkg.Plan(dev, 13, q2)
        y = residues(pkg, (B, 2), q2[:L], n, dev, gen)
        yn = torch.empty_like(y)
        t_f2 = timed(lambda: plan2.ntt(y, 2, L, out=yn), a.reps)
        keys2 = [residues(pkg, (2,), q2, n, dev, gen) for _ in range(L)]
        ct32 = residues(pkg, (B, 3), q2[:L], n, dev, gen)
        t_rn2 = timed(lambda: plan2.relinearize(L, ct32, keys2, out=out, is_ckks=True, is_ntt_form=True), a.reps)
        res["u64_path"] = {"what": "N=8192 {60,40,40,60} (L=3, K=4): integer butterflies (ArithU64) vs the same shape on {40,40,40,40} (exact-FP64 policy)", "batch": B,
                           "ntt_forward_GBps": round(nb / t_f / 1e9, 1), "ntt_forward_hbm_frac": round(nb / t_f / 1e9 / 8000.0, 4),
                           "ntt_inverse_GBps": round(nb / t_i / 1e9, 1), "ntt_inverse_hbm_frac": round(nb / t_i / 1e9 / 8000.0, 4),
                           "relinearize_ntt_form_ops_per_s": round(B / t_rn, 1), "relinearize_coeff_form_ops_per_s": round(B / t_rc, 1),
                           "fp64_policy_same_shape": {"ntt_forward_GBps": round(nb / t_f2 / 1e9, 1), "relinearize_ntt_form_ops_per_s": round(B / t_rn2, 1)}}

    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
