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


@pytest.mark.parametrize("n,bits", [(32, [30, 30, 30, 30]), (4096, [40, 60]), (8192, [60, 40, 40, 60]), (16384, [50] * 3)])
def test_modulo_and_multiply_uint64operand(O, pkg, dev, n, bits):
    """utils::modulo_ps / multiply_uint64operand_ps (utils/poly_small_mod.cu:119-180, :752-814): inputs are ARBITRARY 64-bit words"""
    import ctypes as C
    ctx, q = _ctx(O, n, bits)
    L = len(q)
    mods = ctx.moduli()
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 1 << 64, size=(3, L, n), dtype=np.uint64)
    a[0, :, 0] = 0
    a[0, :, 1] = (1 << 64) - 1
    for l in range(L):
        a[1, l, :6] = [q[l] - 1, q[l], q[l] + 1, 2 * q[l] - 1, 2 * q[l], (1 << 64) - q[l]]
    da = pkg.to_device(a, dev)
    af = a.reshape(-1)
    out = np.empty_like(af)
    O.lib().orc_modulo_ps(O.ptr(af), 3, n, mods, L, O.ptr(out))
    assert np.array_equal(out.reshape(a.shape), a % np.array(q, dtype=np.uint64)[None, :, None])     # the checker itself
    assert np.array_equal(pkg.to_host(plan.modulo(da, L)).reshape(-1), out)
    # a limb slice: moduli 1..L-1
    if L > 2:
        sub = np.ascontiguousarray(a[:, 1:])
        exp = np.empty(sub.size, dtype=np.uint64)
        O.lib().orc_modulo_ps(O.ptr(sub.reshape(-1)), 3, n, O.moduli_array(q[1:]), L - 1, O.ptr(exp))
        assert np.array_equal(pkg.to_host(plan.modulo(pkg.to_device(sub, dev), L - 1, mod_start=1)).reshape(-1), exp)
    for operands in ([1] * L, [qq - 1 for qq in q], [int(x) % qq for x, qq in zip(rng.integers(0, 1 << 62, size=L), q)]):
        ops = (O.MulOp * L)()
        for l in range(L):
            O.lib().orc_mulop_init(C.byref(ops[l]), operands[l], C.byref(mods[l]))
        O.lib().orc_multiply_uint64operand_ps(O.ptr(af), ops, 3, n, mods, L, O.ptr(out))
        # first principles: multiply_uint64operand_mod is x * w mod q for every 64-bit x
        exp = np.array([[(int(x) * operands[l]) % q[l] for x in a[1, l, :8]] for l in range(L)], dtype=np.uint64)
        assert np.array_equal(out.reshape(a.shape)[1, :, :8], exp)
        dops = pkg.to_device(np.array([[o.operand, o.quotient] for o in ops], dtype=np.uint64), dev)
        assert np.array_equal(pkg.to_host(plan.multiply_uint64operand(da, dops, L)).reshape(-1), out)
    # in place
    plan.modulo(da, L, out=da)
    O.lib().orc_modulo_ps(O.ptr(af), 3, n, mods, L, O.ptr(out))
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
