"""GPU parity of the BGV-only steps with the oracle's restatement of the reference's host branches:
mod_t_and_divide_q_last_ntt, decrypt_mod_t (exact_convey_array), the ski_util5 key-switch tail -- on uniform residues
and uniform key material -- and a genuine-key pipeline whose every stage is compared word for word and decrypted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(O, pkg, dev, n, bits, t):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    return O.Context("bgv", n, q, t), pkg.Plan(dev, n.bit_length() - 1, q), q


@pytest.mark.parametrize("n,bits,t", [(8192, [40, 40, 40], 1032193), (4096, [36, 36, 37], 65537), (1024, [50, 50, 50, 50], 1 << 20), (64, [30, 30, 30], 257)])
def test_mod_switch_and_decrypt_mod_t(O, pkg, dev, n, bits, t):
    ctx, plan, q = _setup(O, pkg, dev, n, bits, t)
    K = len(q)
    for L in range(K, 1, -1):
        if np.gcd(q[L - 1] % t, t) != 1:
            continue                                            # q_last not invertible mod t: the reference refuses too
        bgv = pkg.Bgv(plan, L, t)
        assert bgv.inv_q_last_mod_t == ctx.bgv_inv_q_last_mod_t(L)
        x = np.stack([ctx.random_ct(10 * L + b, 2, L) for b in range(3)])
        got = pkg.to_host(bgv.mod_t_and_divide_q_last_ntt(pkg.to_device(x, dev), 2))
        for b in range(3):
            assert np.array_equal(got[b], ctx.mod_t_and_divide_q_last_ntt(L, x[b])), (L, b)
    for L in range(1, K + 1):
        bgv = pkg.Bgv(plan, L, t)
        ph = np.stack([ctx.random_ct(77 + b, 1, L)[0] for b in range(2)])
        ph[0, :, :4] = 0
        for cf in (1, 3):
            if np.gcd(cf, t) != 1:
                continue
            got = pkg.to_host(bgv.decrypt_mod_t(pkg.to_device(ph, dev), cf))
            inv = pow(cf, -1, t)
            for b in range(2):
                want = (ctx.decrypt_mod_t(L, ph[b]).astype(object) * inv) % t
                assert [int(v) for v in got[b]] == [int(v) for v in want], (L, cf, b)
    pl = O.fill_uniform(5, t, n)
    bgv = pkg.Bgv(plan, K - 1, t)
    assert [int(v) for v in pkg.to_host(bgv.multiply_scalar_mod_t(pkg.to_device(pl, dev), 12345))] == [int(v) * 12345 % t for v in pl]


@pytest.mark.parametrize("n,bits,t", [(8192, [40, 40, 40], 1032193), (4096, [36, 36, 37, 37], 65537), (16384, [50, 50, 50, 50], 65537), (64, [30, 30, 30], 257)])
def test_switch_key_bgv_tail(O, pkg, dev, n, bits, t):
    ctx, plan, q = _setup(O, pkg, dev, n, bits, t)
    K = len(q)
    key_level = pkg.Bgv(plan, K, t)
    for L in range(K - 1, 0, -1):
        keys_h = ctx.random_keys(40 + L, L)
        keys_d = [pkg.to_device(k, dev) for k in keys_h]
        target = np.stack([ctx.random_ct(3 * L + b, 1, L)[0] for b in range(2)])
        old = np.stack([ctx.random_ct(9 * L + b, 2, L) for b in range(2)])
        for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
            dest = pkg.to_device(old.copy(), dev)
            key_level.switch_key(L, pkg.to_device(target, dev), keys_d, dest=dest, assign=assign)
            got = pkg.to_host(dest)
            for b in range(2):
                want = ctx.switch_key(L, True, target[b], keys_h, assign=assign, dest=old[b].copy())
                assert np.array_equal(got[b], want), (L, assign, b)
        ct3 = np.stack([ctx.random_ct(21 * L + b, 3, L) for b in range(2)])
        got = pkg.to_host(key_level.relinearize(L, pkg.to_device(ct3, dev), keys_d))
        for b in range(2):
            assert np.array_equal(got[b], ctx.relinearize(L, True, ct3[b], keys_h)), (L, b)


def test_bgv_pipeline_on_gpu(O, pkg, dev):
    """genuine keys: multiply -> relinearize -> two modulus switches on the GPU equal the oracle word for word and decrypt
    (on the GPU: dot product with s, INTT, decrypt_mod_t with the correction factor) to the product of the plaintexts"""
    from test_oracle_bgv import negacyclic
    n, t = 4096, 65537
    ctx, plan, q = _setup(O, pkg, dev, n, [40, 40, 40, 40], t)
    K, L = len(q), len(q) - 1
    rng = O.Rng(8)
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    rk = ctx.relin_keys(rng, sk)
    rs = np.random.RandomState(2)
    m1, m2 = (rs.randint(0, t, n).astype(np.uint64) for _ in range(2))
    c1, c2 = ctx.encrypt_asymmetric_bgv(rng, pk, m1), ctx.encrypt_asymmetric_bgv(rng, pk, m2)
    want = negacyclic(m1, m2, t)
    d1, d2 = pkg.to_device(c1, dev), pkg.to_device(c2, dev)
    prod = plan.dyadic_convolute(d1, 2, d2, 2, L)
    assert np.array_equal(pkg.to_host(prod).reshape(3, L, n), ctx.ckks_multiply(L, c1, c2))
    key_level = pkg.Bgv(plan, K, t)
    rel = key_level.relinearize(L, prod, [pkg.to_device(k, dev) for k in rk])
    rel_h = ctx.relinearize(L, True, ctx.ckks_multiply(L, c1, c2), rk)
    assert np.array_equal(pkg.to_host(rel)[0], rel_h)
    cur, cur_h, cf = rel, rel_h, 1
    dsk = pkg.to_device(sk, dev)
    for level in range(L, 1, -1):
        bgv = pkg.Bgv(plan, level, t)
        cur = bgv.mod_t_and_divide_q_last_ntt(cur, 2)
        cur_h = ctx.mod_switch_scale_to_next(level, cur_h)
        cf = cf * bgv.inv_q_last_mod_t % t
        assert np.array_equal(pkg.to_host(cur)[0, :, :, :], cur_h)
        lv = level - 1
        # decrypt on the GPU: c0 + c1 (.) s in NTT form, INTT, decrypt_mod_t
        c = cur.view(2, lv, n)
        phase = plan.add(c[0].contiguous(), plan.dyadic_product(c[1].contiguous(), dsk[:lv].contiguous(), lv), lv)
        phase = plan.ntt(phase.view(1, lv, n), 1, lv, inverse=True)
        dec = pkg.to_host(pkg.Bgv(plan, lv, t).decrypt_mod_t(phase, cf))[0]
        assert np.array_equal(dec.astype(np.uint64), want), level
        assert np.array_equal(ctx.decrypt_bgv(sk, cur_h, cf), want)
