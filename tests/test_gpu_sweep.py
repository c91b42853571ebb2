"""Seeded shape x switch sweep against the oracle: the dispatch matrix of the library (arithmetic class per limb, ring
size, level, batch-dependent workgroup orders, ~20 per-plan switches) has outgrown hand-picked parametrisations -- bugs in this kind of code
live at shape x switch boundaries (batch not a multiple of 8, L = K - 1 vs lower levels, the <= 128-workgroup threshold).  One fixed seed,
~1000 cases: N in 2^10 .. 2^15 (3 % of the cases: 2^16 / 2^17, K <= 3, batch <= 3), K in 2 .. 12, bit sizes 27 .. 60 (mixed classes included), L in 1 .. K - 1, batch in {1, 2, 3, 5, 8, 12, 24, 64},
scheme x form x assign method, and one random set of switches per case.  Operations: switch_key, relinearize, rescale / mod-switch, BEHZ
multiply, the fused multiply -> relinearize -> rescale entry, multiply_plain_accumulate.  A failure prints a one-line reproducer
(`SWEEP_CASE=<index> python -m pytest tests/test_gpu_sweep.py -m gpu -k one_case`).
The reference's own matrix is parameter sets x operations (test/evaluator.cu:276-392)."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = int(os.environ.get("SWEEP_SEED", "20261003"))      # (another seed / more cases for a longer soak: SWEEP_SEED=7 SWEEP_CASES=5000)
CASES = int(os.environ.get("SWEEP_CASES", "1000"))
BATCHES = (1, 2, 3, 5, 8, 12, 24, 64)
OPS = ("switch_key", "relinearize", "rescale", "mod_switch", "bfv_multiply", "fused_chain", "plain_mac")
SWITCHES = {
    "TROYN_KS_ORDER": ("plain", "item", "row", "band"),
    "TROYN_KS_SPLIT": ("0", "1"),
    "TROYN_KS_DIAG": ("loop",),
    "TROYN_KS_MAC": ("split", "fused"),
    "TROYN_KS_TAIL": ("split",),
    "TROYN_KS_MAC_SHOUP": ("0",),
    "TROYN_KS_ROWS": ("1", "2"),
    "TROYN_NTT_ARITH": ("u64",),
    "TROYN_NTT_SPLIT": ("0", "1"),
    "TROYN_NTT_HALF": ("0", "0x3f3f", "0x0040", "0x8000"),
    "TROYN_NTT_SMALL_TWO_PASS": ("0",),
    "TROYN_NTT_OVERLAP": ("0",),
    "TROYN_MRR": ("calls",),
    "TROYN_MRR_MIXED": ("0",),
    "TROYN_MRR_SMALL": ("0",),
    "TROYN_MRR_CHUNK": ("8", "16"),
    "TROYN_MRR_STREAMS": ("1", "3"),
    "TROYN_BFV_TENSOR": ("split", "fused"),
    "TROYN_TENSOR_WGS": ("2", "3"),
    "TROYN_BEHZ": ("v1",),
    "TROYN_BEHZ_LIFT": ("split",),
    "TROYN_PLAIN_MAC": ("v1", "single", "dual", "quad"),
}


def make_case(index):
    """case `index` of the sweep, a pure function of (SEED, index)"""
    rng = random.Random(SEED * 1000003 + index)
    op = OPS[index % len(OPS)]
    log_n = rng.choice((10, 11, 12, 13, 13, 14, 14, 15))
    if op == "bfv_multiply":
        log_n = rng.choice((10, 11, 12, 13, 13, 14, 15))
    elif op != "plain_mac" and rng.random() < 0.03:
        log_n = rng.choice((16, 17))        # the sizes above 32768 (utils/constants.h:13 allows 131072): the unfused key-switch chain, multi-pass transforms
    kmax = {10: 12, 11: 12, 12: 10, 13: 8, 14: 6, 15: 4, 16: 3, 17: 3}[log_n]
    K = rng.randint(2, kmax)
    style = rng.choice(("narrow", "wide", "mixed", "mixed", "ckks_like", "any"))
    if style == "narrow":
        bits = [rng.randint(27, 49) for _ in range(K)]
    elif style == "wide":
        bits = [rng.randint(51, 60) for _ in range(K)]
    elif style == "ckks_like":
        bits = [60] + [rng.choice((40, 45, 50)) for _ in range(K - 2)] + [60] if K >= 2 else [60]
        bits = bits[:K]
    else:
        bits = [rng.choice((rng.randint(27, 50), rng.randint(50, 60))) for _ in range(K)]
    L = rng.randint(1, K - 1)
    if op in ("rescale", "mod_switch", "fused_chain"):
        L = rng.randint(2, K) if op != "fused_chain" else (rng.randint(2, K - 1) if K >= 3 else 0)
    batch = rng.choice(BATCHES)
    if log_n == 15 and batch == 64 and K > 3:
        batch = 24
    if log_n >= 16:
        batch = min(batch, 3)
    scheme = rng.choice(("ckks", "bfv"))
    if op in ("rescale", "fused_chain"):
        scheme = "ckks"
    if op in ("mod_switch", "bfv_multiply", "plain_mac"):
        scheme = "bfv"
    assign = rng.randint(0, 2)
    nsw = rng.choice((0, 1, 1, 2, 3))
    opts = {}
    for name in rng.sample(sorted(SWITCHES), nsw):
        opts[name] = rng.choice(SWITCHES[name])
    return {"index": index, "op": op, "n": 1 << log_n, "bits": bits, "L": L, "batch": batch, "scheme": scheme, "assign": assign, "opts": opts,
            "seed": rng.randint(1, 1 << 30)}


def run_case(O, pkg, dev, c):
    import torch
    n, bits, L, batch = c["n"], c["bits"], c["L"], c["batch"]
    K = len(bits)
    if L < 1 or (c["op"] == "fused_chain" and (K < 3 or L < 2)):
        return "skipped"
    t = 65537 if c["scheme"] == "bfv" else 0
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(c["scheme"], n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    for name, val in c["opts"].items():
        plan.set_option(name, val)
    sd = c["seed"] % 100000
    check = sorted({0, batch // 2, batch - 1})
    is_ckks = c["scheme"] == "ckks"
    ntt_form = is_ckks
    if c["op"] in ("switch_key", "relinearize", "fused_chain"):
        keys = ctx.random_keys(sd, L)
        dkeys = [pkg.to_device(k, dev) for k in keys]
    if c["op"] == "switch_key":
        base = np.stack([ctx.random_ct(sd + 5 + i, 1, L)[0] for i in range(min(batch, 3))])
        tg = base[np.arange(batch) % base.shape[0]]
        d0b = np.stack([ctx.random_ct(sd + 40 + i, 2, L) for i in range(min(batch, 3))])
        d0 = d0b[np.arange(batch) % d0b.shape[0]]
        dd = pkg.to_device(d0, dev)
        plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=c["assign"], is_ckks=is_ckks, is_ntt_form=ntt_form)
        got = pkg.to_host(dd)
        for i in check:
            assert np.array_equal(got[i], ctx.switch_key(L, ntt_form, tg[i], keys, assign=c["assign"], dest=d0[i])), "item %d" % i
    elif c["op"] == "relinearize":
        base = np.stack([ctx.random_ct(sd + 70 + i, 3, L) for i in range(min(batch, 3))])
        ct3 = base[np.arange(batch) % base.shape[0]]
        got = pkg.to_host(plan.relinearize(L, pkg.to_device(ct3, dev), dkeys, is_ckks=is_ckks, is_ntt_form=ntt_form))
        for i in check:
            assert np.array_equal(got[i], ctx.relinearize(L, ntt_form, ct3[i], keys)), "item %d" % i
    elif c["op"] in ("rescale", "mod_switch"):
        p = 2 + (c["seed"] & 1)
        base = np.stack([ctx.random_ct(sd + 13 + i, p, L) for i in range(min(batch, 3))])
        x = base[np.arange(batch) % base.shape[0]]
        if c["op"] == "rescale":
            got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(x, dev), p))
        else:
            got = pkg.to_host(plan.divide_and_round_q_last(L, pkg.to_device(x, dev), p))
        for i in check:
            assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i])), "item %d" % i
    elif c["op"] == "bfv_multiply":
        behz = pkg.Behz(plan, L, t)
        base_a = np.stack([ctx.random_ct(sd + 231 + i, 2, L) for i in range(min(batch, 3))])
        base_b = np.stack([ctx.random_ct(sd + 247 + i, 2, L) for i in range(min(batch, 3))])
        a, b = base_a[np.arange(batch) % base_a.shape[0]], base_b[(np.arange(batch) + 1) % base_b.shape[0]]
        got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
        for i in check[:2]:
            assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], b[i])), "item %d" % i
    elif c["op"] == "fused_chain":
        base_a = np.stack([ctx.random_ct(sd + 100 + i, 2, L) for i in range(min(batch, 3))])
        base_b = np.stack([ctx.random_ct(sd + 200 + i, 2, L) for i in range(min(batch, 3))])
        a, b = base_a[np.arange(batch) % base_a.shape[0]], base_b[(np.arange(batch) + 1) % base_b.shape[0]]
        got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(a, dev), pkg.to_device(b, dev), dkeys))
        for i in check:
            e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
            assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), "item %d" % i
    elif c["op"] == "plain_mac":
        rng = random.Random(c["seed"])
        B, I, J = rng.randint(1, 2), rng.randint(1, 9), rng.choice((1, 2, 3, 4, 8))
        a = np.stack([np.stack([ctx.random_ct(sd + 100 + bb * 31 + i, 2, L) for i in range(I)]) for bb in range(B)])
        w = np.stack([np.stack([ctx.random_ct(sd + 900 + i * 17 + j, 1, L)[0] for j in range(J)]) for i in range(I)])
        da, dw = pkg.to_device(a, dev), pkg.to_device(w, dev)
        out = torch.empty((B, J, 2, L, n), dtype=torch.int64, device=dev)
        cts, pts, dsts = [], [], []
        for i in range(I):
            for j in range(J):
                for bb in range(B):
                    cts.append(da[bb, i]); pts.append(dw[i, j]); dsts.append(out[bb, j])
        plan.multiply_plain_accumulate(cts, pts, dsts, 2, L, set_zero=True)
        got = pkg.to_host(out)
        qv = [int(x) for x in q[:L]]
        for bb, j in {(0, 0), (B - 1, J - 1)}:
            acc = np.zeros((2, L, n), dtype=np.uint64)
            for i in range(I):
                term = ctx.multiply_plain_ntt(L, a[bb, i], w[i, j])
                for l in range(L):
                    acc[:, l] = (acc[:, l] + term[:, l]) % np.uint64(qv[l])
            assert np.array_equal(got[bb, j], acc), "destination (%d, %d)" % (bb, j)
    return "ok"


def describe(c):
    return "SWEEP_CASE=%d  op=%s n=%d bits=%s L=%d batch=%d scheme=%s assign=%d opts=%s" % (
        c["index"], c["op"], c["n"], c["bits"], c["L"], c["batch"], c["scheme"], c["assign"], c["opts"])


def test_sweep(O, pkg, dev):
    failures, done = [], 0
    for index in range(CASES):
        c = make_case(index)
        try:
            done += run_case(O, pkg, dev, c) == "ok"
        except Exception as e:      # an assertion or a library error: the reproducer line, then the next case
            failures.append("%s  -> %s: %s" % (describe(c), type(e).__name__, str(e)[:160]))
    assert not failures, "%d of %d cases failed:\n" % (len(failures), CASES) + "\n".join(failures)
    assert done >= CASES * 0.9, done


def test_one_case(O, pkg, dev):
    """SWEEP_CASE=<index>: one case of the sweep with its traceback (the reproducer the sweep prints)"""
    index = os.environ.get("SWEEP_CASE")
    if index is None:
        pytest.skip("set SWEEP_CASE=<index>")
    c = make_case(int(index))
    print(describe(c))
    assert run_case(O, pkg, dev, c) in ("ok", "skipped")
