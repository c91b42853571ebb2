"""GPU parity: chains with moduli of 2^50 and more (the reference bench tool's default {60,40,40,60}, {60,50,...,60} CKKS chains) through the
integer half-tile inner product (csrc/ksmaci_kernels.hpp) and the per-class fused chain, against the oracle.
Reference: evaluator_keyswitching_core.cu:904-919, :987-1051 (every modulus alike); test/bench/he_operations.cu:22-24 (default log_q)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(O, pkg, dev, scheme, n, bits, t=0):
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(scheme, n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    return ctx, plan, q


WIDE_CHAINS = [
    (8192, [60, 40, 40, 60], 3),           # the reference tool's default chain: rows 0 and 3 integer, rows 1, 2 FP64 with wide digits
    (8192, [60, 60, 60], 2),               # every row integer
    (8192, [59, 60, 60], 2),               # a digit of a modulus ABOVE the row's: reduced while loading (Modulus::reduce)
    (8192, [60, 55, 50, 45, 60], 2),       # lower level: L < K - 1
    (8192, [40, 60], 1),                   # a single digit: only the diagonal shortcut in NTT form
    (16384, [60, 50, 50, 50, 50, 60], 5),  # half tiles
    (16384, [55, 55, 56], 2),
    (16384, [60, 50, 50, 60], 2),          # lower level of the half-tile kernel
    (32768, [60, 50, 50, 60], 3),          # quarter tiles
    (32768, [58, 59, 60], 2),
]


@pytest.fixture(autouse=True, params=["fused", "default"])
def inner_product_form(request, monkeypatch):
    """Every test of this module twice: a few ciphertexts of a chain with moduli >= 2^50 take the two-launch inner product by default (digit
    transforms, then the multiply-accumulate: the one-launch kernels have no digit-parallel form and a launch that cannot fill the chip is one
    long chain -- ks_small_mixed in troyn.hip); TROYN_KS_MAC=fused (read when the plan is created) keeps ksmac2<WIDE> + ksmaci_kernel at every
    size, which is what these tests were written for.  Results are the same words either way."""
    if request.param == "fused":
        monkeypatch.setenv("TROYN_KS_MAC", "fused")
    else:
        monkeypatch.delenv("TROYN_KS_MAC", raising=False)
    return request.param


@pytest.mark.parametrize("scheme,ntt_form", [("ckks", True), ("bfv", False)])
@pytest.mark.parametrize("n,bits,L", WIDE_CHAINS)
def test_switch_key_wide_rows(O, pkg, dev, scheme, ntt_form, n, bits, L):
    t = 1032193 if scheme == "bfv" else 0
    ctx, plan, q = _setup(O, pkg, dev, scheme, n, bits, t)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    for batch in (2, 8):       # 8: the XCD-aware workgroup order
        tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])
        dt = pkg.to_device(tg, dev)
        for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
            d0 = np.stack([ctx.random_ct(40 + i, 2, L) for i in range(batch)])
            dd = pkg.to_device(d0, dev)
            plan.switch_key(L, dt, dkeys, dest=dd, assign=assign, is_ckks=(scheme == "ckks"), is_ntt_form=ntt_form)
            got = pkg.to_host(dd)
            for i in sorted({0, 1, batch - 1}):
                exp = ctx.switch_key(L, ntt_form, tg[i], keys, assign=assign, dest=d0[i])
                assert np.array_equal(got[i], exp), "switch_key mismatch (batch %d, assign=%d, item %d)" % (batch, assign, i)
            if batch == 8 and assign != pkg.ASSIGN_OVERWRITE:
                break       # one pass over the assign methods is enough for the second workgroup order
        assert np.array_equal(pkg.to_host(dt), tg)


@pytest.mark.parametrize("n,bits,L", [(8192, [50, 50, 50, 50], 3), (16384, [45, 45, 45, 45], 2), (32768, [50] * 3, 2)])
def test_forced_integer_rows(O, pkg, dev, monkeypatch, n, bits, L):
    """TROYN_NTT_ARITH=u64: a chain of narrow moduli on the integer kernels (every row through ksmaci_kernel): same words as the FP64 path"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    batch = 3
    ct3 = np.stack([ctx.random_ct(70 + i, 3, L) for i in range(batch)])
    d3 = pkg.to_device(ct3, dev)
    ref = pkg.to_host(plan.relinearize(L, d3, dkeys, is_ckks=True, is_ntt_form=True))
    monkeypatch.setenv("TROYN_NTT_ARITH", "u64")
    plan_i = pkg.Plan(dev, n.bit_length() - 1, q)
    got = pkg.to_host(plan_i.relinearize(L, d3, dkeys, is_ckks=True, is_ntt_form=True))
    for i in range(batch):
        exp = ctx.relinearize(L, True, ct3[i], keys)
        assert np.array_equal(ref[i], exp), i
        assert np.array_equal(got[i], exp), i


@pytest.mark.parametrize("n,bits,L", [(8192, [60, 40, 40, 60], 3), (16384, [60, 50, 50, 50, 50, 60], 5), (32768, [60, 50, 50, 60], 3)])
def test_wide_rows_corner_operands(O, pkg, dev, n, bits, L):
    """all-(q-1) targets and keys: every lazy range of the integer butterflies and of the Shoup accumulation at its upper end"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    K = len(q)
    keys = [np.stack([np.stack([np.full(n, q[r] - 1, dtype=np.uint64) for r in range(K)]) for _ in range(2)]) for _ in range(L)]
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([np.stack([np.full(n, q[l] - 1, dtype=np.uint64) for l in range(L)]),
                   np.stack([(np.arange(n, dtype=np.uint64) % 2) * np.uint64(q[l] - 1) for l in range(L)])])
    got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    for i in range(2):
        assert np.array_equal(got[i], ctx.switch_key(L, True, tg[i], keys, assign=pkg.ASSIGN_OVERWRITE)), i


MIXED_FUSED = [
    (16384, [60, 50, 50, 50, 50, 60], 5, 8),   # the usual CKKS shape: wide first and special primes around 50-bit scaling primes
    (16384, [60, 50, 50, 50, 50, 60], 3, 3),   # below the top level
    (8192, [60, 40, 40, 60], 3, 8),            # the reference bench tool's default chain
    (8192, [60, 40, 40, 60], 2, 2),
    (8192, [55, 55, 56], 2, 5),                # every limb wide
    (8192, [49, 50, 51, 40], 3, 3),            # limbs straddling 2^50 under a narrow special prime: T_l of a wide dropped limb, T_s as doubles
    (8192, [40, 60, 40, 50], 3, 2),            # wide limb in the middle, narrow special
    (8192, [40, 40, 60], 2, 2),                # narrow data limbs under a wide special prime: T_s as u64 words into FP64 kernels
    (16384, [50, 50, 55, 50], 3, 8),           # the dropped limb is the wide one: T_l as u64 words into FP64 kernels
    (32768, [60, 50, 50, 60], 3, 2),           # two-pass transforms
    (32768, [50, 60, 50, 50], 3, 8),
]


@pytest.mark.parametrize("n,bits,L,batch", MIXED_FUSED)
def test_fused_chain_mixed_moduli(O, pkg, dev, n, bits, L, batch):
    """multiply -> relinearize -> rescale as one call on chains with moduli of 2^50 and more (per-class launches: ksmaci_kernel and the integer
    forms of the fused transforms next to the FP64 kernels) == the three public calls == the composition of the calls inside the entry
    (TROYN_MRR_MIXED=0) == the oracle"""
    import torch
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(21, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(200 + i, 2, L) for i in range(batch)])
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    got = plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    prod = plan.dyadic_convolute(da, 2, db, 2, L)
    relin = plan.relinearize(L, prod, dkeys, is_ckks=True, is_ntt_form=True)
    three = plan.divide_and_round_q_last_ntt(L, relin, 2)
    assert torch.equal(got, three), "fused entry differs from the three-call composition"
    plan.set_option("TROYN_MRR_MIXED", "0")
    composed = plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    plan.set_option("TROYN_MRR_MIXED", None)
    assert torch.equal(got, composed)
    got = pkg.to_host(got)
    for i in sorted({0, batch // 2, batch - 1}):
        e = ctx.ckks_multiply(L, a[i], b[i])
        e = ctx.relinearize(L, True, e, keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i
    assert np.array_equal(pkg.to_host(da), a) and np.array_equal(pkg.to_host(db), b)


@pytest.mark.parametrize("n,bits,L,batch", [(8192, [40, 40, 40, 40], 3, 8), (16384, [50] * 6, 5, 8), (32768, [50] * 4, 3, 2)])
def test_fused_chain_forced_integer(O, pkg, dev, monkeypatch, n, bits, L, batch):
    """TROYN_NTT_ARITH=u64: the whole fused chain on the integer kernels (a chain of narrow moduli): same words as the FP64 chain and the oracle"""
    import torch
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(21, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(200 + i, 2, L) for i in range(batch)])
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    ref = plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    monkeypatch.setenv("TROYN_NTT_ARITH", "u64")
    plan_i = pkg.Plan(dev, n.bit_length() - 1, q)
    got = plan_i.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    assert torch.equal(got, ref)
    got = pkg.to_host(got)
    for i in sorted({0, batch - 1}):
        e = ctx.ckks_multiply(L, a[i], b[i])
        e = ctx.relinearize(L, True, e, keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i


def test_fused_chain_mixed_corner_operands(O, pkg, dev):
    """all-(q-1) operands and keys through the per-class fused chain"""
    n, bits, L = 16384, [60, 50, 50, 50, 50, 60], 5
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    K = len(q)
    keys = [np.stack([np.stack([np.full(n, q[r] - 1, dtype=np.uint64) for r in range(K)]) for _ in range(2)]) for _ in range(L)]
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([np.stack([np.stack([np.full(n, q[l] - 1, dtype=np.uint64) for l in range(L)]) for _ in range(2)]),
                  np.stack([np.stack([(np.arange(n, dtype=np.uint64) % 3 == 0) * np.uint64(q[l] - 1) for l in range(L)]) for _ in range(2)])])
    da = pkg.to_device(a, dev)
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, da, da, dkeys))
    for i in range(2):
        e = ctx.ckks_multiply(L, a[i], a[i])
        e = ctx.relinearize(L, True, e, keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i


@pytest.mark.parametrize("n,bits,L,batch", [(8192, [60, 40, 40, 60], 3, 192), (8192, [40, 60, 40, 55, 50], 4, 128), (16384, [60, 50, 50, 60], 3, 96), (32768, [60, 50, 60], 2, 160)])
def test_per_class_launches_in_flight_together(O, pkg, dev, n, bits, L, batch):
    """Launches over limbs of both classes run one launch per run of one class; for launches of >= 512 limb-polynomials every other run goes to a side
    stream (forked from / joined to the caller's stream by events) so that integer (issue-bound) and FP64 (memory-side) runs overlap.  Same words with the
    overlap off (TROYN_NTT_OVERLAP=0) and in the oracle: key switch in both forms (three assign methods: AddInplace reads the old destination), rescale,
    plain transforms and the fused chain; the two-pass size keeps one scratch part per run."""
    import torch
    q = O.coeff_modulus_create(n, bits)
    plans = {"on": pkg.Plan(dev, n.bit_length() - 1, q), "off": pkg.Plan(dev, n.bit_length() - 1, q)}
    plans["off"].set_option("TROYN_NTT_OVERLAP", "0")
    for scheme, ntt_form in (("ckks", True), ("bfv", False)):
        ctx = O.Context(scheme, n, q, 0 if scheme == "ckks" else 65537)
        keys = ctx.random_keys(3, L)
        dkeys = [pkg.to_device(k, dev) for k in keys]
        base = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(4)])
        tg = base[np.arange(batch) % 4]
        d0b = np.stack([ctx.random_ct(40 + i, 2, L) for i in range(4)])
        d0 = d0b[(np.arange(batch) + 1) % 4]
        for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
            res = {}
            for name, plan in plans.items():
                dd = pkg.to_device(d0, dev)
                plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=assign, is_ckks=ntt_form, is_ntt_form=ntt_form)
                res[name] = dd
            assert torch.equal(res["on"], res["off"]), (scheme, assign)
            got = pkg.to_host(res["on"])
            for i in (0, 5, batch - 1):
                assert np.array_equal(got[i], ctx.switch_key(L, ntt_form, tg[i], keys, assign=assign, dest=d0[i])), (scheme, assign, i)
    ctx = O.Context("ckks", n, q)
    keys = ctx.random_keys(21, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    ba = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(4)])
    bb = np.stack([ctx.random_ct(200 + i, 2, L) for i in range(4)])
    a, b = ba[np.arange(batch) % 4], bb[(np.arange(batch) // 4) % 4]
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    fused = {name: plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys) for name, plan in plans.items()}
    assert torch.equal(fused["on"], fused["off"])
    resc = {name: plan.divide_and_round_q_last_ntt(L, da, 2) for name, plan in plans.items()}
    assert torch.equal(resc["on"], resc["off"])
    fwd = {name: plan.ntt(plan.ntt(da, 2, L, inverse=True), 2, L) for name, plan in plans.items()}
    assert torch.equal(fwd["on"], da) and torch.equal(fwd["off"], da)
    got, gr = pkg.to_host(fused["on"]), pkg.to_host(resc["on"])
    for i in (0, 7, batch - 1):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i
        assert np.array_equal(gr[i], ctx.mod_switch_scale_to_next(L, a[i])), i
