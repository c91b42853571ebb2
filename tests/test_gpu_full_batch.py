"""Every item of a batch against the oracle, one case per workgroup order: the XCD-dealt orders -- ksmaci_kernel's
`grouped` (batch a multiple of 8) and plain order, the limb-parallel grid of mrr_quartet_kernel (a few ciphertexts), the per-class launches with and
without the side stream -- were checked on 2-3 items per batch; a permutation confined to other slots of a group of 8 would have passed.  All operands
are distinct, so a swapped pair of items cannot compare equal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WIDE = [(8192, [60, 40, 40, 60], 3), (16384, [60, 50, 50, 50, 50, 60], 5), (32768, [60, 50, 60], 2)]


def _setup(O, pkg, dev, scheme, n, bits, t=0):
    q = O.coeff_modulus_create(n, bits)
    return O.Context(scheme, n, q, t), pkg.Plan(dev, n.bit_length() - 1, q), q


@pytest.mark.parametrize("form", ["one-launch", "two-launch"])
@pytest.mark.parametrize("batch", [24, 21])
@pytest.mark.parametrize("n,bits,L", WIDE)
def test_switch_key_wide_rows_every_item(O, pkg, dev, n, bits, L, batch, form):
    """24: ksmaci's grouped order (b = (blk / per) * 8 + r % 8); 21: plain; TROYN_KS_MAC=fused keeps the one-launch inner product for a launch this small"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    plan.set_option("TROYN_KS_MAC", "fused" if form == "one-launch" else "split")
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([ctx.random_ct(500 + i, 1, L)[0] for i in range(batch)])
    d0 = np.stack([ctx.random_ct(900 + i, 2, L) for i in range(batch)])
    dd = pkg.to_device(d0, dev)
    plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=pkg.ASSIGN_ADD_INPLACE, is_ckks=True, is_ntt_form=True)
    got = pkg.to_host(dd)
    for i in range(batch):
        assert np.array_equal(got[i], ctx.switch_key(L, True, tg[i], keys, assign=pkg.ASSIGN_ADD_INPLACE, dest=d0[i])), i


@pytest.mark.parametrize("batch", [3, 24, 21])
@pytest.mark.parametrize("n,bits,L", WIDE + [(16384, [50] * 6, 5), (8192, [40] * 4, 3), (32768, [50] * 4, 3)])
def test_fused_chain_every_item(O, pkg, dev, n, bits, L, batch):
    """3: the latency-bound form (two-pass transforms, mrr_quartet_kernel's one-thread-per-output-limb grid); 24 / 21: grouped and plain orders of
    the throughput kernels"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(21, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(300 + i, 2, L) for i in range(batch)])
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(a, dev), pkg.to_device(b, dev), dkeys))
    for i in range(batch):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i


@pytest.mark.parametrize("overlap", ["on", "off"])
def test_per_class_side_stream_every_item(O, pkg, dev, overlap):
    """launches of >= 512 limb-polynomials of a mixed chain: every other per-class run on the side stream (RunOverlap) or one after the other"""
    n, bits, L, batch = 8192, [60, 40, 40, 60], 3, 96
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    if overlap == "off":
        plan.set_option("TROYN_NTT_OVERLAP", "0")
    keys = ctx.random_keys(21, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(300 + i, 2, L) for i in range(batch)])
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    fused = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys))
    resc = pkg.to_host(plan.divide_and_round_q_last_ntt(L, da, 2))
    relin = pkg.to_host(plan.switch_key(L, da[:, 1].contiguous(), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    for i in range(batch):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(fused[i], ctx.mod_switch_scale_to_next(L, e)), i
        assert np.array_equal(resc[i], ctx.mod_switch_scale_to_next(L, a[i])), i
        assert np.array_equal(relin[i], ctx.switch_key(L, True, a[i, 1], keys, assign=pkg.ASSIGN_OVERWRITE)), i
