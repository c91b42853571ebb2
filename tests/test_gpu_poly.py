"""GPU parity: element-wise RNS ops and dyadic products vs the oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(O, n, bits, scheme="ckks", t=0):
    q = O.coeff_modulus_create(n, bits)
    return O.Context(scheme, n, q, t), q


@pytest.mark.parametrize("n,bits", [(32, [30, 30, 30, 30]), (4096, [40, 60]), (8192, [40, 40, 40])])
def test_elementwise(O, pkg, dev, n, bits):
    ctx, q = _ctx(O, n, bits)
    L = len(q)
    mods = ctx.moduli()
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    a, b = ctx.random_ct(1, 3, L), ctx.random_ct(2, 3, L)
    a[0, 0, :4] = 0                    # negate(0) = 0 edge (uint_small_mod.h:30-36)
    a[0, 1, :2] = q[1] - 1
    b[0, 1, :2] = q[1] - 1             # add wraps exactly once
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    af, bf = a.reshape(-1), b.reshape(-1)
    out = np.empty_like(af)
    O.lib().orc_add_ps(O.ptr(af), O.ptr(bf), 3, n, mods, L, O.ptr(out))
    assert np.array_equal(pkg.to_host(plan.add(da, db, L)).reshape(-1), out)
    O.lib().orc_sub_ps(O.ptr(af), O.ptr(bf), 3, n, mods, L, O.ptr(out))
    assert np.array_equal(pkg.to_host(plan.sub(da, db, L)).reshape(-1), out)
    O.lib().orc_negate_ps(O.ptr(af), 3, n, mods, L, O.ptr(out))
    assert np.array_equal(pkg.to_host(plan.negate(da, L)).reshape(-1), out)
    O.lib().orc_dyadic_product_ps(O.ptr(af), O.ptr(bf), 3, n, mods, L, O.ptr(out))
    assert np.array_equal(pkg.to_host(plan.dyadic_product(da, db, L)).reshape(-1), out)
    for scalar in (1, 1032193, (1 << 32), q[0] + 5, (1 << 64) - 1):
        O.lib().orc_multiply_scalar_ps(O.ptr(af), scalar, 3, n, mods, L, O.ptr(out))
        assert np.array_equal(pkg.to_host(plan.multiply_scalar(da, scalar, L)).reshape(-1), out)
    # in-place (out aliases a)
    O.lib().orc_add_ps(O.ptr(af), O.ptr(bf), 3, n, mods, L, O.ptr(out))
    plan.add(da, db, L, out=da)
    assert np.array_equal(pkg.to_host(da).reshape(-1), out)


@pytest.mark.parametrize("n,bits,pa,pb", [(32, [40, 40, 40], 2, 2), (1024, [50, 50], 3, 2), (8192, [40, 40, 40], 2, 3),
                                          (2048, [60], 3, 3), (16384, [50] * 5, 2, 2), (64, [30, 30], 1, 2), (64, [30, 30], 4, 1)])
def test_dyadic_convolute(O, pkg, dev, n, bits, pa, pb):
    ctx, q = _ctx(O, n, bits)
    L = len(q)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    batch = 3
    a = np.stack([ctx.random_ct(10 + i, pa, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(20 + i, pb, L) for i in range(batch)])
    got = pkg.to_host(plan.dyadic_convolute(pkg.to_device(a, dev), pa, pkg.to_device(b, dev), pb, L))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.ckks_multiply(L, a[i], b[i]))


def test_dyadic_square(O, pkg, dev):
    n = 4096
    ctx, q = _ctx(O, n, [45, 45, 45])
    L = 3
    plan = pkg.Plan(dev, 12, q)
    a = np.stack([ctx.random_ct(3 + i, 2, L) for i in range(2)])
    got = pkg.to_host(plan.dyadic_square(pkg.to_device(a, dev), L))
    for i in range(2):
        exp = np.zeros(3 * L * n, dtype=np.uint64)
        ai = a[i].reshape(-1)
        O.lib().orc_dyadic_square(O.ptr(ai), ctx.moduli(), L, n, O.ptr(exp))
        assert np.array_equal(got[i].reshape(-1), exp)
        # dyadic_square == dyadic_convolute(a, a)
        assert np.array_equal(got[i], ctx.ckks_multiply(L, a[i], a[i]))


def test_modulus_slice(O, pkg, dev):
    # limb l uses modulus mod_start + l
    n = 256
    ctx, q = _ctx(O, n, [30, 40, 50, 60])
    plan = pkg.Plan(dev, 8, q)
    sub = q[2:4]
    ctx2 = O.Context("ckks", n, sub)
    a, b = ctx2.random_ct(1, 2, 2), ctx2.random_ct(2, 2, 2)
    got = pkg.to_host(plan.dyadic_convolute(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2, 2, mod_start=2))
    assert np.array_equal(got[0], ctx2.ckks_multiply(2, a, b))
