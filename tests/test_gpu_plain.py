"""GPU parity of the ciphertext x plaintext path (BASELINE config 5): centralize,
transform_plain_to_ntt, multiply_plain (both forms) and the batched multiply-accumulate of the matmul application."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,bits,t", [(8192, [60, 40, 40, 60], 1 << 21),     # examples/10_bfv_matmul.cu parameters
                                      (8192, [40, 40, 40], 1032193), (64, [40, 40, 40], 769)])
def test_centralize_and_multiply_plain(O, pkg, dev, n, bits, t):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    L = len(q) - 1
    m = O.fill_uniform(77, t, 3 * n).reshape(3, n)
    m[0, :4] = [0, 1, t - 1, (t + 1) // 2]
    cen = plan.plain_centralize(L, t, pkg.to_device(m, dev))
    cen_h = pkg.to_host(cen)
    for i in range(3):
        assert np.array_equal(cen_h[i], ctx.plain_centralize(L, m[i]))
    # multiply_plain_normal: centralize -> NTT(plain), NTT(ct), broadcast product, INTT
    ct = np.stack([ctx.random_ct(5 + i, 2, L) for i in range(3)])
    pt_ntt = plan.ntt(cen, 1, L)
    dct = plan.ntt(pkg.to_device(ct, dev), 2, L)
    prod = plan.ntt(plan.dyadic_broadcast_product(dct, 2, pt_ntt, L), 2, L, inverse=True)
    got = pkg.to_host(prod)
    for i in range(3):
        assert np.array_equal(got[i], ctx.multiply_plain_normal(L, ct[i], m[i])), i
    # shared plaintext (pt_bstride = 0) and NTT-form product against the oracle's multiply_plain_ntt
    shared = plan.dyadic_broadcast_product(dct, 2, pt_ntt[0].contiguous(), L, shared_plain=True)
    dct_h, ptn_h = pkg.to_host(dct), pkg.to_host(pt_ntt)
    for i in range(3):
        assert np.array_equal(pkg.to_host(shared)[i], ctx.multiply_plain_ntt(L, dct_h[i], ptn_h[0]))


@pytest.mark.parametrize("mode", [None, "single", "dual"], ids=["default", "one_destination", "two_destinations"])
@pytest.mark.parametrize("n,bits,B,I,J", [(4096, [36, 36, 37], 2, 5, 3), (8192, [60, 40, 40, 60], 1, 40, 2), (32, [40, 40, 40], 3, 2, 2),
                                        (2048, [60, 60, 61 - 1], 1, 71, 2),    # 71 terms of 60-bit residues: two lazy-sum folds + a 3-term remainder
                                        (4096, [36, 36, 37], 2, 5, 4), (2048, [60, 60, 61 - 1], 1, 67, 8)])   # J % 4 == 0: four destinations per workgroup by default
def test_multiply_plain_accumulate_matmul_pattern(O, pkg, dev, monkeypatch, mode, n, bits, B, I, J):
    """ret[b][j] = sum_i a[b][i] (.) w[i][j]  (MatmulHelper::matmul, app/matmul.cu:352-370) in ONE launch; destinations that share their
    ciphertext operands are computed 4 / 2 / 1 per workgroup (TROYN_PLAIN_MAC, read per call)"""
    monkeypatch.delenv("TROYN_PLAIN_MAC", raising=False)
    if mode:
        if J % 2 and mode == "dual":
            pytest.skip("an odd number of columns never pairs")
        monkeypatch.setenv("TROYN_PLAIN_MAC", mode)
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    ctx = O.Context("bfv", n, q, 1 << 16)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    L = len(q) - 1
    a = np.stack([np.stack([ctx.random_ct(100 + b * 31 + i, 2, L) for i in range(I)]) for b in range(B)])        # [B][I][2][L][N]
    w = np.stack([np.stack([ctx.random_ct(900 + i * 17 + j, 1, L)[0] for j in range(J)]) for i in range(I)])     # [I][J][L][N]
    da, dw = pkg.to_device(a, dev), pkg.to_device(w, dev)
    out = torch.empty((B, J, 2, L, n), dtype=torch.int64, device=dev)
    cts, pts, dsts = [], [], []
    for i in range(I):
        for j in range(J):
            for b in range(B):
                cts.append(da[b, i]); pts.append(dw[i, j]); dsts.append(out[b, j])
    plan.multiply_plain_accumulate(cts, pts, dsts, 2, L, set_zero=True)
    got = pkg.to_host(out)
    qv = [int(x) for x in q[:L]]
    for b in range(B):
        for j in range(J):
            acc = np.zeros((2, L, n), dtype=np.uint64)
            for i in range(I):
                term = ctx.multiply_plain_ntt(L, a[b, i], w[i, j])
                for l in range(L):
                    acc[:, l] = (acc[:, l] + term[:, l]) % np.uint64(qv[l])
            assert np.array_equal(got[b, j], acc), (b, j)
    # accumulate on top of existing destinations (set_zero = False)
    plan.multiply_plain_accumulate(cts[:B * J], pts[:B * J], dsts[:B * J], 2, L, set_zero=False)
    got2 = pkg.to_host(out)
    for b in range(B):
        for j in range(J):
            term = ctx.multiply_plain_ntt(L, a[b, 0], w[0, j])
            exp = np.stack([(got[b, j][:, l] + term[:, l]) % np.uint64(qv[l]) for l in range(L)], axis=1)
            assert np.array_equal(got2[b, j], exp), (b, j)


@pytest.mark.parametrize("scheme,ntt,n,bits", [("bfv", False, 8192, [40, 40, 40]), ("ckks", True, 16384, [50] * 6),
                                               ("ckks", True, 32, [40, 40, 40]), ("bfv", False, 64, [40, 40, 40])])
def test_apply_galois(O, pkg, dev, scheme, ntt, n, bits):
    """GaloisTool permutations (both forms) and Evaluator::apply_galois = permutation + key switch"""
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    ctx = O.Context(scheme, n, q, 1032193 if scheme == "bfv" else 0)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    L = len(q) - 1
    ct = np.stack([ctx.random_ct(3 + i, 2, L) for i in range(2)])
    dct = pkg.to_device(ct, dev)
    keys = ctx.random_keys(13, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    for step in (1, -3, 0):
        g = ctx.galois_element_from_step(step)
        for form in (False, True):
            got = pkg.to_host(plan.apply_galois_poly(dct, L, g, form))
            for i in range(2):
                assert np.array_equal(got[i], ctx.apply_galois(L, form, g, ct[i])), (step, form)
        rot = pkg.to_host(plan.apply_galois(L, dct, g, dkeys, is_ckks=(scheme == "ckks"), is_ntt_form=ntt))
        for i in range(2):
            assert np.array_equal(rot[i], ctx.apply_galois_ct(L, ntt, g, ct[i], keys)), step
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.apply_galois_poly(dct, L, 4, False)         # even element


@pytest.mark.parametrize("n,t", [(8192, 1032193), (64, 1 << 21), (4096, (1 << 61) - 1)])
def test_apply_galois_plain(O, pkg, dev, n, t):
    """GaloisTool::apply on plaintexts modulo the plain modulus (Evaluator::apply_galois_plain, coefficient form): X -> X^g with the
    negacyclic sign, checked against the index arithmetic of utils/galois.cu:43-66 and, for t among the q_i, against the oracle"""
    q = [int(v) for v in O.coeff_modulus_create(max(n, 32), [40, 40, 40])]
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    rng = np.random.default_rng(n)
    x = rng.integers(0, t, size=(3, n), dtype=np.uint64)
    dx = pkg.to_device(x, dev)
    for g in (3, 2 * n - 1, n + 1, 5 ** 7 % (2 * n)):
        got = pkg.to_host(plan.apply_galois_plain(dx, t, g))
        idx = np.arange(n, dtype=np.uint64) * np.uint64(g)
        dst = (idx & np.uint64(n - 1)).astype(np.int64)
        neg = ((idx >> np.uint64(n.bit_length() - 1)) & np.uint64(1)).astype(bool)
        exp = np.empty_like(x)
        exp[:, dst] = np.where(neg[None, :], (np.uint64(t) - x) % np.uint64(t), x)
        assert np.array_equal(got, exp), g
    # the same kernel with t = q_0 must agree with the plan-modulus form the oracle covers
    ctx = O.Context("bfv", n, q, 1032193) if n >= 32 else None
    y = rng.integers(0, q[0], size=(1, 1, n), dtype=np.uint64)
    assert np.array_equal(pkg.to_host(plan.apply_galois_plain(pkg.to_device(y.reshape(1, n), dev), q[0], 3)).reshape(1, n), ctx.apply_galois(1, False, 3, y).reshape(1, n))
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.apply_galois_plain(dx, t, 2 * n + 1)


def test_rotation_decrypts(O, pkg, dev):
    """rotate_rows / rotate_columns with genuine Galois keys: device result == oracle, and it decrypts to the rotated slots"""
    n, t = 8192, 1032193
    q = [int(v) for v in O.coeff_modulus_create(n, [40, 40, 40])]
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, 13, q)
    rng = O.Rng(5)
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    ct = ctx.encrypt_asymmetric_bfv(rng, pk, ctx.batch_encode(list(range(1, n + 1))))
    dct = pkg.to_device(ct[None], dev)
    row = n // 2
    for step in (1, -2, 0):
        g = ctx.galois_element_from_step(step)
        gk = ctx.galois_key(rng, sk, g)
        got = pkg.to_host(plan.apply_galois(2, dct, g, [pkg.to_device(k, dev) for k in gk], is_ckks=False, is_ntt_form=False))[0]
        assert np.array_equal(got, ctx.apply_galois_ct(2, False, g, ct, gk))
        dec = [int(v) for v in ctx.batch_decode(ctx.decrypt_bfv(sk, got))]
        if step == 0:
            assert dec[:3] == [row + 1, row + 2, row + 3] and dec[row:row + 3] == [1, 2, 3]
        else:
            want = [((i + step) % row) + 1 for i in range(4)]
            assert dec[:4] == want and dec[row:row + 4] == [w + row for w in want]


@pytest.mark.parametrize("n,bits,t,count,stride,batch", [
    (8192, [60, 40, 40, 60], 1 << 21, 16, 16, 300),        # BASELINE config 5's weight blocks: 16 coefficients per plaintext, mixed chain (split by class, >= 512 limb-polys)
    (8192, [60, 40, 40, 60], 1 << 21, 8192, 8192, 3),      # full plaintexts, a few: one launch of the integer kernels over all limbs
    (8192, [40, 40, 40], 1032193, 100, 128, 5),            # row stride above the coefficient count, FP64 class
    (16384, [50] * 4, 65537, 16384, 16384, 2),             # small launch: two-pass form of the transform
    (16384, [50] * 4, 65537, 1000, 1000, 200),             # whole-limb tiles, half-word LDS variant of the plain forward kernel
    (16384, [60, 50, 50, 60], (1 << 30) + 3, 77, 80, 150),
    (32768, [50] * 3, 786433, 5000, 5000, 4),              # two passes: the loader acts in the strided first pass
    (65536, [55, 56], 12289, 65536, 65536, 2),
    (4096, [36, 36, 37], 40961, 4096, 4096, 7),
    (1024, [30, 30], 12289, 10, 10, 9),
    (64, [40, 40], 97, 64, 64, 3),                         # N < 1024: falls back to the two launches
    (8192, [40, 40, 40], (1 << 45) + 59, 500, 500, 3),     # t not below every modulus: the general lift, two launches
])
def test_plain_centralize_ntt_one_launch(O, pkg, dev, n, bits, t, count, stride, batch):
    """troyn_plain_centralize_ntt = Evaluator::transform_plain_to_ntt (evaluator_transform_ntt.cu:35-70): scaling_variant::centralize in the loader of the forward
    transform.  Same words as the two calls (which the tests above pin to the oracle) and as the oracle's centralize + NTT directly; values at and around
    the threshold (t + 1) / 2, 0 and t - 1 included; the coefficients beyond `count` are never read (poisoned here)."""
    import torch
    q = O.coeff_modulus_create(n, bits)
    L = len(bits) - 1
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    rng = np.random.default_rng(n + count + batch)
    plain = rng.integers(0, t, size=(batch, stride), dtype=np.uint64)
    edge = np.array([0, 1, t - 1, (t + 1) // 2, (t + 1) // 2 - 1, (t + 1) // 2 + 1, t // 2], dtype=np.uint64)
    plain[0, :min(len(edge), count)] = edge[:min(len(edge), count)]
    plain[:, count:] = np.uint64(0xDEADBEEFDEADBEEF)       # padding between rows: must not enter the transform
    dp = pkg.to_device(plain, dev)
    got = pkg.to_host(plan.plain_centralize_ntt(L, t, dp, coeff_count=count))
    # the two launches on zero-padded full rows
    full = np.zeros((batch, n), dtype=np.uint64)
    full[:, :count] = plain[:, :count]
    two = pkg.to_host(plan.ntt(plan.plain_centralize(L, t, pkg.to_device(full, dev)), 1, L))
    assert np.array_equal(got, two.reshape(got.shape))
    fast = all(t < int(v) for v in q[:L])
    for i in sorted({0, batch // 2, batch - 1}):
        if fast:
            want = ctx.to_ntt(ctx.plain_centralize(L, full[i])[None], 1, L)[0]
            assert np.array_equal(got[i], want), i
