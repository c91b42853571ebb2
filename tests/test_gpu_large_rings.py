"""GPU parity above N = 32768.  The reference allows poly_modulus_degree up to 131072 (src/utils/constants.h:13; its
transforms are multi-pass there, src/fgk/ntt_grouped.cu:281-292).  Here N = 65536 / 131072 take the unfused key-switch chain (decomposition
transforms, inner product, inverse transforms, ski_util7-style tail as separate launches: DESIGN 4.7) and the multi-pass transforms; no
other `-m gpu` test ran switch_key / relinearize / rescale / mod-switch / the fused entry at these sizes.  Every item of every batch is
compared with the oracle; chains of both arithmetic classes (all moduli < 2^50: FP64 butterflies; >= 2^50: integer) and a mixed one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHAINS = [
    (65536, [50, 50, 50], "fp64"),
    (65536, [58, 57, 59], "int"),
    (65536, [60, 45, 60], "mixed"),
    (131072, [50, 50, 50], "fp64"),
    (131072, [55, 56, 57], "int"),
    (131072, [40, 60], "mixed2"),
]
IDS = ["N%d-%s" % (n, tag) for n, _, tag in CHAINS]


def _setup(O, pkg, dev, scheme, n, bits, t=0):
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(scheme, n, q, t)
    return ctx, pkg.Plan(dev, n.bit_length() - 1, q), q


@pytest.mark.parametrize("scheme", ["ckks", "bfv"])
@pytest.mark.parametrize("n,bits,tag", CHAINS, ids=IDS)
def test_switch_key_and_relinearize(O, pkg, dev, n, bits, tag, scheme):
    """switch_key_internal (evaluator_keyswitching_core.cu:757-1052) under the three assign methods + relinearize, at the top level and one below"""
    t = 786433 if scheme == "bfv" else 0
    ctx, plan, q = _setup(O, pkg, dev, scheme, n, bits, t)
    is_ckks = ntt_form = scheme == "ckks"
    for L in sorted({len(bits) - 1, max(1, len(bits) - 2)}):
        keys = ctx.random_keys(3 + L, L)
        dkeys = [pkg.to_device(k, dev) for k in keys]
        batch = 3 if n == 65536 else 2
        tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])
        dt = pkg.to_device(tg, dev)
        for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
            d0 = np.stack([ctx.random_ct(40 + i, 2, L) for i in range(batch)])
            dd = pkg.to_device(d0, dev)
            plan.switch_key(L, dt, dkeys, dest=dd, assign=assign, is_ckks=is_ckks, is_ntt_form=ntt_form)
            got = pkg.to_host(dd)
            for i in range(batch):
                assert np.array_equal(got[i], ctx.switch_key(L, ntt_form, tg[i], keys, assign=assign, dest=d0[i])), (L, assign, i)
        assert np.array_equal(pkg.to_host(dt), tg)
        ct3 = np.stack([ctx.random_ct(70 + i, 3, L) for i in range(batch)])
        got = pkg.to_host(plan.relinearize(L, pkg.to_device(ct3, dev), dkeys, is_ckks=is_ckks, is_ntt_form=ntt_form))
        for i in range(batch):
            assert np.array_equal(got[i], ctx.relinearize(L, ntt_form, ct3[i], keys)), (L, i)


@pytest.mark.parametrize("n,bits,tag", CHAINS, ids=IDS)
def test_rescale_and_mod_switch(O, pkg, dev, n, bits, tag):
    """divide_and_round_q_last_ntt (CKKS rescale) and divide_and_round_q_last (BFV mod-switch), utils/rns_tool.cu:374-694; 2 and 3 polynomials"""
    for scheme in ("ckks", "bfv"):
        ctx, plan, q = _setup(O, pkg, dev, scheme, n, bits, 786433 if scheme == "bfv" else 0)
        for L in range(2, len(bits) + 1):
            for p in (2, 3):
                batch = 2
                x = np.stack([ctx.random_ct(13 + 7 * p + i, p, L) for i in range(batch)])
                fn = plan.divide_and_round_q_last_ntt if scheme == "ckks" else plan.divide_and_round_q_last
                got = pkg.to_host(fn(L, pkg.to_device(x, dev), p))
                for i in range(batch):
                    assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i])), (scheme, L, p, i)


@pytest.mark.parametrize("n,bits,tag", [c for c in CHAINS if len(c[1]) >= 3], ids=[i for i, c in zip(IDS, CHAINS) if len(c[1]) >= 3])
def test_fused_chain(O, pkg, dev, n, bits, tag):
    """troyn_ckks_multiply_relinearize_rescale at sizes where it composes the three calls: == oracle's multiply -> relinearize -> rescale"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    L = len(bits) - 1
    keys = ctx.random_keys(17, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    batch = 3 if n == 65536 else 2
    a = np.stack([ctx.random_ct(100 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(200 + i, 2, L) for i in range(batch)])
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(a, dev), pkg.to_device(b, dev), dkeys))
    for i in range(batch):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i


@pytest.mark.parametrize("n,bits,tag", CHAINS, ids=IDS)
def test_transforms_round_trip_and_match(O, pkg, dev, n, bits, tag):
    """plain forward / inverse transforms of every limb at these sizes vs the oracle (fgk/ntt_grouped.cu multi-pass path), and INTT(NTT(x)) == x"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    L = len(bits)
    x = np.stack([ctx.random_ct(9 + i, 2, L) for i in range(2)])
    dx = pkg.to_device(x, dev)
    fwd = plan.ntt(dx, 2, L)
    got = pkg.to_host(fwd)
    for i in range(2):
        assert np.array_equal(got[i], ctx.to_ntt(x[i], 2, L)), i
    back = pkg.to_host(plan.ntt(fwd, 2, L, inverse=True))
    assert np.array_equal(back, x)
