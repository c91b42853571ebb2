"""GPU parity: key switching / relinearize / modulus switching / BEHZ multiply vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(O, pkg, dev, scheme, n, bits, t=0):
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(scheme, n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    return ctx, plan, q


@pytest.mark.parametrize("n,bits,L,batch", [(16384, [50] * 6, 5, 16), (8192, [40, 40, 40, 40], 3, 24), (4096, [36] * 5, 2, 8), (8192, [40, 40, 40, 40], 3, 12)])
def test_switch_key_batches_of_eight(O, pkg, dev, n, bits, L, batch):
    """batches that are multiples of 8 take the XCD-aware workgroup order of the fused inner product (all L+1 rows of an item
    on one XCD); every item still equals the oracle (batch 12: the plain row-major order)"""
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])
    got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    for i in sorted({0, 1, 7, batch // 2, batch - 1}):
        assert np.array_equal(got[i], ctx.switch_key(L, True, tg[i], keys, assign=pkg.ASSIGN_OVERWRITE)), i


@pytest.mark.parametrize("scheme,ntt_form,n,bits,L", [
    ("ckks", True, 32, [40, 40, 40], 2),
    ("ckks", True, 32, [60, 40, 40, 60], 3),      # the reference's test moduli (test/evaluator.cu)
    ("ckks", True, 1024, [30, 30, 30, 30], 2),     # lower level: L < K-1
    ("ckks", True, 8192, [40, 40, 40, 40], 3),
    ("ckks", True, 16384, [50] * 6, 5),            # BASELINE config 3
    ("ckks", True, 4096, [36] * 10, 9),            # fused inner product with > 8 digits (accumulator re-centring)
    ("ckks", True, 2048, [55, 55, 56], 2),         # moduli >= 2^50: integer butterflies in the fused kernels
    ("bfv", False, 2048, [54, 54, 55], 2),
    ("bfv", False, 32, [40, 40, 40], 2),
    ("bfv", False, 4096, [36, 36, 37], 2),
    ("bfv", False, 8192, [40, 40, 40], 2),         # BASELINE config 2 shape
    ("bfv", False, 32768, [50] * 4, 3),
    ("ckks", True, 32768, [50] * 4, 3),            # two-pass transforms with the fused tail: AddInplace / OverwriteExceptFirst read the old destination
    ("ckks", True, 8192, [50] * 4, 3),             # N = 8192 instance of the half-tile inner product (whole-limb tiles)
    ("ckks", True, 8192, [40, 40], 1),             # a single digit: only the diagonal shortcut and one transformed row
    ("ckks", True, 16384, [45, 45, 45], 1),        # lower level (L < K-1) of the half-tile kernel
    ("bfv", False, 2048, [27, 27], 1),
])
def test_switch_key(O, pkg, dev, scheme, ntt_form, n, bits, L):
    t = 1032193 if scheme == "bfv" else 0
    ctx, plan, q = _setup(O, pkg, dev, scheme, n, bits, t)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    batch = 2
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])
    dt = pkg.to_device(tg, dev)
    is_ckks = scheme == "ckks"
    for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
        d0 = np.stack([ctx.random_ct(40 + i, 2, L) for i in range(batch)])
        dd = pkg.to_device(d0, dev)
        plan.switch_key(L, dt, dkeys, dest=dd, assign=assign, is_ckks=is_ckks, is_ntt_form=ntt_form)
        got = pkg.to_host(dd)
        for i in range(batch):
            exp = ctx.switch_key(L, ntt_form, tg[i], keys, assign=assign, dest=d0[i])
            assert np.array_equal(got[i], exp), "switch_key mismatch (assign=%d, item %d)" % (assign, i)
    # the target must be left untouched
    assert np.array_equal(pkg.to_host(dt), tg)


@pytest.mark.parametrize("scheme,ntt_form,n,bits,L", [
    ("ckks", True, 64, [40, 40, 40], 2),
    ("ckks", True, 16384, [50] * 6, 5),
    ("bfv", False, 8192, [40, 40, 40], 2),
])
def test_relinearize(O, pkg, dev, scheme, ntt_form, n, bits, L):
    t = 1032193 if scheme == "bfv" else 0
    ctx, plan, q = _setup(O, pkg, dev, scheme, n, bits, t)
    keys = ctx.random_keys(9, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    batch = 3
    ct3 = np.stack([ctx.random_ct(70 + i, 3, L) for i in range(batch)])
    got = pkg.to_host(plan.relinearize(L, pkg.to_device(ct3, dev), dkeys, is_ckks=(scheme == "ckks"), is_ntt_form=ntt_form))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.relinearize(L, ntt_form, ct3[i], keys))


@pytest.mark.parametrize("tail", ["fused", "split"])
@pytest.mark.parametrize("n,bits,L", [(4096, [36, 36, 37], 2), (2048, [54, 54, 55], 2), (32768, [50] * 4, 3), (8192, [60, 40, 40, 60], 3)])
def test_coefficient_form_tail(O, pkg, dev, monkeypatch, n, bits, L, tail):
    """BFV (coefficient form): the inverse transforms of the data rows finish the key switch in their epilogue (rounding fix from the
    special-prime row, divide by the special prime, assign / accumulate, relinearize's trailing add); TROYN_KS_TAIL=split keeps the
    separate ski_util7 launch.  FP64 class, integer class, two-pass size, mixed chain; every assign method and relinearize."""
    if tail == "split":
        monkeypatch.setenv("TROYN_KS_TAIL", "split")
    else:
        monkeypatch.delenv("TROYN_KS_TAIL", raising=False)
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, 1032193)
    keys = ctx.random_keys(13, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    batch = 3
    tg = np.stack([ctx.random_ct(15 + i, 1, L)[0] for i in range(batch)])
    for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
        d0 = np.stack([ctx.random_ct(140 + i, 2, L) for i in range(batch)])
        dd = pkg.to_device(d0, dev)
        plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=assign, is_ckks=False, is_ntt_form=False)
        got = pkg.to_host(dd)
        for i in range(batch):
            assert np.array_equal(got[i], ctx.switch_key(L, False, tg[i], keys, assign=assign, dest=d0[i])), (assign, i)
    ct3 = np.stack([ctx.random_ct(170 + i, 3, L) for i in range(batch)])
    got = pkg.to_host(plan.relinearize(L, pkg.to_device(ct3, dev), dkeys, is_ckks=False, is_ntt_form=False))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.relinearize(L, False, ct3[i], keys)), i


@pytest.mark.parametrize("n,bits,L", [(32, [40, 40, 40], 3), (32, [30, 40, 50, 60], 4), (32, [60, 50, 40, 30], 4),
                                      (8192, [40, 40, 40], 2), (16384, [50] * 6, 5), (32768, [50] * 3, 3)])
def test_ckks_rescale(O, pkg, dev, n, bits, L):
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    batch, p = 2, 2
    x = np.stack([ctx.random_ct(13 + i, p, L) for i in range(batch)])
    got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(x, dev), p))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i]))
    # 3-polynomial ciphertext (rescale before relinearize)
    x3 = ctx.random_ct(77, 3, L)
    got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(x3, dev), 3))
    assert np.array_equal(got[0], ctx.mod_switch_scale_to_next(L, x3))


@pytest.mark.parametrize("n,bits,L", [(32, [40, 40, 40], 3), (32, [30, 40, 50, 60], 4), (32, [60, 50, 40, 30], 2),
                                      (8192, [40, 40, 40], 2), (32768, [50] * 4, 4)])
def test_bfv_mod_switch(O, pkg, dev, n, bits, L):
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, 1032193)
    batch, p = 2, 2
    x = np.stack([ctx.random_ct(21 + i, p, L) for i in range(batch)])
    got = pkg.to_host(plan.divide_and_round_q_last(L, pkg.to_device(x, dev), p))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i]))


def test_mod_switch_drop(O, pkg, dev):
    n, L = 1024, 4
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, [40] * 5)
    x = np.stack([ctx.random_ct(3 + i, 2, L) for i in range(2)])
    got = pkg.to_host(plan.mod_switch_drop(L, L - 1, pkg.to_device(x, dev), 2))
    for i in range(2):
        assert np.array_equal(got[i], ctx.mod_switch_drop_to_next(L, x[i]))
    got = pkg.to_host(plan.mod_switch_drop(L, 1, pkg.to_device(x, dev), 2))
    assert np.array_equal(got, x[:, :, :1, :])


@pytest.mark.parametrize("n,bits,L,t,pa,pb", [
    (32, [40, 40, 40], 2, 1032193 % (1 << 20) | 1, 2, 2),
    (32, [60, 60, 60], 3, 65537, 2, 2),
    (32, [30], 1, 97, 2, 2),
    (1024, [27, 27, 27], 2, 12289, 2, 3),
    (8192, [40, 40, 40], 2, 1032193, 2, 2),       # BASELINE config 2
    (8192, [60, 40, 40, 60], 3, 1 << 21, 2, 2),   # matmul app parameters (non-prime t)
    (16384, [50] * 6, 5, 1032193, 2, 2),
    (32768, [50] * 11, 10, 1032193, 2, 2),        # BASELINE config 4
    (2048, [59] * 7, 6, 40961, 2, 2),             # wide moduli, more than one group of four carry-free terms
    (1024, [36] * 17, 16, 12289, 2, 2),           # largest base the per-size kernels are instantiated for
    (1024, [30] * 18, 17, 12289, 2, 2),           # beyond it: first-generation kernels
    (4096, [49, 50, 51, 40], 3, 65537, 3, 2),     # one modulus above 2^50: integer class
    (2048, [40, 40, 40], 2, 12289, 2, 2),         # whole-limb tensor kernel at N = 2048 (16 coefficients per thread), FP64 class
    (65536, [50, 50, 50], 2, 786433, 2, 2),       # N = 65536: two-pass transforms with 16 strided points
])
def test_bfv_multiply(O, pkg, dev, n, bits, L, t, pa, pb, behz_gen):
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, t)
    behz = pkg.Behz(plan, L, t)
    rt = O.lib().orc_context_rns_tool(ctx.h, L)
    bsk = np.zeros(O.lib().orc_rns_tool_base_Bsk_size(rt), dtype=np.uint64)
    O.lib().orc_rns_tool_base_Bsk(rt, O.ptr(bsk))
    assert behz.base_Bsk == [int(v) for v in bsk]
    batch = 2
    a = np.stack([ctx.random_ct(31 + i, pa, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(47 + i, pb, L) for i in range(batch)])
    got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), pa, pkg.to_device(b, dev), pb))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], b[i]))
    da = pkg.to_device(a, dev)                       # one buffer passed twice: the squaring shortcut
    got = pkg.to_host(behz.multiply(da, pa, da, pa))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], a[i])), i


@pytest.mark.parametrize("n,bits,L,t", [(8192, [40, 40, 40], 2, 1032193), (16384, [50] * 6, 5, 1032193), (32768, [50] * 11, 10, 1032193), (4096, [36] * 5, 4, 65537)])
def test_bfv_multiply_auxiliary_base(O, pkg, dev, monkeypatch, n, bits, L, t):
    """Every q_i below 2^50: by default the multiply works in an auxiliary base of primes BELOW 2^50 (all transforms on the FP64 butterflies) sized by the
    reference's own criterion, bits(prod(B') m_sk') > 32 + bits(t) + bits(q) (utils/rns_tool.cu:50-62); TROYN_BEHZ_BASE=ref keeps the reference's 61-bit base.
    Results must not depend on the base.  Checked: the working base meets that criterion and is disjoint from the key chain, the known-answer
    hook still reports the reference's base, the products under both bases equal the oracle's (which keeps the reference's base), and a handle created with
    the small-prime base also serves the first-generation kernels."""
    monkeypatch.delenv("TROYN_BEHZ", raising=False)
    monkeypatch.delenv("TROYN_BEHZ_BASE", raising=False)
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, t)
    plan.set_option("TROYN_BEHZ_BASE", "ref")     # (read by troyn_behz_create from the plan)
    ref = pkg.Behz(plan, L, t)
    plan.set_option("TROYN_BEHZ_BASE", None)
    fast = pkg.Behz(plan, L, t)
    assert ref.working_base_size == len(ref.base_Bsk) and fast.base_Bsk == ref.base_Bsk
    # capacity: the reference's criterion (utils/rns_tool.cu:50-62) on primes of ~50 bits: bits(prod(B') m_sk') > 32 + bits(t) + bits(q)
    q_bits = sum(int(v).bit_length() for v in q[:L])                   # >= bits of the product
    assert 49.9 * fast.working_base_size > 32 + int(t).bit_length() + (q_bits - L) and fast.working_base_size >= 2      # (may be fewer rows than the reference's)
    batch = 3
    a = np.stack([ctx.random_ct(231 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(247 + i, 2, L) for i in range(batch)])
    # corner operands next to the random ones: all q - 1 and alternating 0 / q - 1
    qa = np.array(q[:L], dtype=np.uint64)[None, :, None]
    a[1] = np.broadcast_to(qa - 1, a[1].shape)
    b[2][..., ::2] = 0
    b[2][..., 1::2] = np.broadcast_to(qa - 1, b[2][..., 1::2].shape)
    want = [ctx.bfv_multiply(L, a[i], b[i]) for i in range(batch)]
    for h in (fast, ref):
        got = pkg.to_host(h.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
        for i in range(batch):
            assert np.array_equal(got[i], want[i]), (h is fast, i)
    plan.set_option("TROYN_BEHZ", "v1")          # first-generation kernels on the handle that holds the small-prime base
    got = pkg.to_host(fast.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
    for i in range(batch):
        assert np.array_equal(got[i], want[i]), ("v1 kernels", i)


@pytest.mark.parametrize("tensor", ["fused", "split"])
@pytest.mark.parametrize("bits,L", [([50] * 4, 3), ([55] * 3, 2), ([55, 50, 50, 50], 3), ([40] * 2, 1)])
def test_bfv_multiply_two_pass_sizes(O, pkg, dev, monkeypatch, bits, L, tensor):
    """N = 32768: the tensor product is formed between the passes of the transforms (tensor_core_kernel: last forward pass, dyadic
    product, first inverse pass in one launch) when both bases are of one arithmetic class; TROYN_BFV_TENSOR=split keeps the separate
    launches.  Chains: FP64 class, integer class, mixed (never fused), single modulus."""
    if tensor == "split":
        monkeypatch.setenv("TROYN_BFV_TENSOR", "split")
    else:
        monkeypatch.delenv("TROYN_BFV_TENSOR", raising=False)
    n, t = 32768, 786433
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, t)
    behz = pkg.Behz(plan, L, t)
    batch = 3
    a = np.stack([ctx.random_ct(131 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(147 + i, 2, L) for i in range(batch)])
    got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], b[i])), i
    # squaring: the same operand twice (two buffers), and one buffer passed twice (Evaluator::square: one lift, two forward transforms)
    got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), 2, pkg.to_device(a, dev), 2))
    assert np.array_equal(got[0], ctx.bfv_multiply(L, a[0], a[0]))
    da = pkg.to_device(a, dev)
    got = pkg.to_host(behz.multiply(da, 2, da, 2))
    for i in range(batch):
        assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], a[i])), i
    assert np.array_equal(pkg.to_host(da), a)


@pytest.mark.parametrize("bits,L", [([45] * 2, 1), ([40, 45, 49], 2), ([50] * 8, 7), ([36] * 13, 12), ([30] * 16, 15)])
def test_bfv_multiply_lift_with_first_pass(O, pkg, dev, bits, L):
    """N = 32768, every modulus below 2^50: the operand's first forward pass over base q, the lift q -> Bsk (fgk/rns_tool.cu:7-100) and the
    first pass over the lifted rows run as ONE launch (behz2_lift_pass1_kernel; rows of both bases meet in LDS); TROYN_BEHZ_LIFT=split keeps the
    three launches.  Both must give the oracle's product; 2 x 2 components, an odd batch, corner operands, the squaring shortcut.  L = 15: more
    rows than the kernel's LDS budget covers -- the entry falls back to the three launches by itself."""
    n, t = 32768, 65537
    ctx, plan, q = _setup(O, pkg, dev, "bfv", n, bits, t)
    batch = 3
    a = np.stack([ctx.random_ct(331 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(347 + i, 2, L) for i in range(batch)])
    qa = np.array(q[:L], dtype=np.uint64)[None, :, None]
    a[1] = np.broadcast_to(qa - 1, a[1].shape)
    b[1][..., ::2] = 0
    b[1][..., 1::2] = np.broadcast_to(qa - 1, b[1][..., 1::2].shape)
    want = [ctx.bfv_multiply(L, a[i], b[i]) for i in (0, 1)]
    outs = {}
    for mode in (None, "split"):
        plan.set_option("TROYN_BEHZ_LIFT", mode)
        behz = pkg.Behz(plan, L, t)
        got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
        for i in (0, 1):
            assert np.array_equal(got[i], want[i]), (mode, i)
        da = pkg.to_device(a, dev)
        sq = pkg.to_host(behz.multiply(da, 2, da, 2))
        outs[mode] = (got, sq)
    assert np.array_equal(outs[None][0], outs["split"][0]) and np.array_equal(outs[None][1], outs["split"][1])
    assert np.array_equal(outs[None][1][2], ctx.bfv_multiply(L, a[2], a[2]))


def test_error_behaviour(O, pkg, dev):
    # misuse -> invalid_argument-style errors, mirroring the reference's checks
    n = 1024
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, [40, 40, 40])
    x = pkg.to_device(ctx.random_ct(1, 2, 2), dev)
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.divide_and_round_q_last_ntt(1, x[:, :, :1].contiguous(), 2)     # no next level
    keys = [pkg.to_device(k, dev) for k in ctx.random_keys(1, 2)]
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.switch_key(3, x, keys)                                          # L > K-1
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.switch_key(2, x[0, 0].contiguous(), keys[:1])                   # key index out of range
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        pkg.Plan(dev, 10, [1 << 61])                                         # modulus too large
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        pkg.Plan(dev, 10, [97])                                              # no 2N-th root of unity


@pytest.mark.parametrize("n,bits,L,batch", [(16384, [50] * 6, 5, 8), (16384, [50] * 6, 3, 3), (8192, [40, 40, 40, 40], 3, 8), (8192, [50] * 3, 2, 5),
                                            (4096, [36] * 4, 3, 2), (32768, [50] * 4, 3, 2), (32768, [50] * 6, 5, 8), (8192, [55, 55, 56], 2, 2),
                                            (32768, [50] * 4, 3, 40), (32768, [45] * 6, 5, 24)])      # (N = 32768 beyond the small launches: merged strided passes, one thread per octet)
def test_ckks_multiply_relinearize_rescale_fused(O, pkg, dev, n, bits, L, batch):
    """the one-call chain equals the three public calls AND the oracle's multiply -> relinearize -> mod_switch_scale_to_next
    (fast path: N = 8192 / 16384 / 32768 with moduli < 2^50, also below the top level; the other shapes compose the three calls)"""
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
    got = pkg.to_host(got)
    for i in sorted({0, batch // 2, batch - 1}):
        e = ctx.ckks_multiply(L, a[i], b[i])
        e = ctx.relinearize(L, True, e, keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i
    # the operands are left untouched
    assert np.array_equal(pkg.to_host(da), a) and np.array_equal(pkg.to_host(db), b)


@pytest.mark.parametrize("n,bits,L,batch", [(16384, [50] * 6, 5, 1), (16384, [50] * 6, 5, 2), (16384, [50] * 6, 2, 1), (16384, [45, 49, 40, 50], 3, 3), (16384, [50] * 6, 4, 1),
                                            (8192, [40] * 4, 3, 1), (8192, [50, 49, 40], 2, 2), (8192, [36] * 5, 4, 3),
                                            (32768, [50] * 6, 5, 1), (32768, [50, 45, 40], 2, 2)])
def test_fused_chain_single_objects(O, pkg, dev, n, bits, L, batch):
    """N = 8192 / 16384 / 32768, a few ciphertexts: every launch of the chain's tail takes the two-pass form and its three strided passes (special rows, dropped limb
    with the key switch's rounding fix, output limbs with both fixes) run as ONE launch on the shared quartets (mrr_quartet_kernel; T_s and T_l stay in
    registers).  TROYN_MRR_SMALL=0 keeps the six launches.  Both equal the oracle's multiply -> relinearize -> rescale (evaluator_keyswitching_core.cu:570-658,
    utils/rns_tool.cu:523-627), below the top level as well; corner operands (all q - 1)."""
    import torch
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, bits)
    keys = ctx.random_keys(29, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = np.stack([ctx.random_ct(500 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(600 + i, 2, L) for i in range(batch)])
    qa = np.array(q[:L], dtype=np.uint64)[None, :, None]
    b[batch - 1] = np.broadcast_to(qa - 1, b[batch - 1].shape)
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    got = plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    plan.set_option("TROYN_MRR_SMALL", "0")
    six = plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys)
    plan.set_option("TROYN_MRR_SMALL", None)
    assert torch.equal(got, six), "merged tail differs from the six-launch tail"
    # the separate calls merge the strided passes of their tails the same way (mrr_quartet_load_kernel): relinearize / switch_key in NTT form
    # (every assign method: AddInplace reads the old destination) and rescale, against the unmerged launches
    prod = plan.dyadic_convolute(da, 2, db, 2, L)
    outs = {}
    for mode in (None, "0"):
        plan.set_option("TROYN_MRR_SMALL", mode)
        relin = plan.relinearize(L, prod, dkeys, is_ckks=True, is_ntt_form=True)
        three = plan.divide_and_round_q_last_ntt(L, relin, 2)
        sk = []
        for assign in (0, 1, 2):
            dd = da.clone()
            plan.switch_key(L, db[:, 1].contiguous(), dkeys, dest=dd, assign=assign, is_ckks=True, is_ntt_form=True)
            sk.append(dd)
        outs[mode] = [relin, three] + sk
    plan.set_option("TROYN_MRR_SMALL", None)
    for x, y in zip(outs[None], outs["0"]):
        assert torch.equal(x, y)
    assert torch.equal(got, outs[None][1]), "fused entry differs from the three-call composition"
    got = pkg.to_host(got)
    for i in range(batch):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), i
    assert np.array_equal(pkg.to_host(da), a) and np.array_equal(pkg.to_host(db), b)


def test_fused_entry_errors_and_timer(O, pkg, dev):
    """error behaviour of the one-call chain (the reference's messages for the step that would have failed) and the kernel-timer hook"""
    import ctypes as C
    import torch
    n, L = 8192, 3
    ctx, plan, q = _setup(O, pkg, dev, "ckks", n, [40, 40, 40, 40])
    keys = [pkg.to_device(k, dev) for k in ctx.random_keys(2, L)]
    a = pkg.to_device(np.stack([ctx.random_ct(1, 2, L)]), dev)
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.ckks_multiply_relinearize_rescale(1, a[:, :, :1].contiguous(), a[:, :, :1].contiguous(), keys)     # nothing to rescale to
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.ckks_multiply_relinearize_rescale(L + 1, a, a, keys)                                                  # no special prime left
    lib = plan.lib
    ws = torch.empty(64, dtype=torch.uint8, device=dev)
    out = torch.empty((1, 2, L - 1, n), dtype=torch.int64, device=dev)
    kp = (C.c_void_p * L)(*[k.data_ptr() for k in keys])
    rc = lib.troyn_ckks_multiply_relinearize_rescale(plan.h, L, C.c_void_p(a.data_ptr()), C.c_void_p(a.data_ptr()), kp, C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(ws.data_ptr()), ws.numel(), 1, None)
    assert rc == -3 and b"workspace too small" in lib.troyn_last_error()
    # kernel timer: one span per key-switch inner product while enabled, none when disabled
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 1))
    for _ in range(3):
        plan.ckks_multiply_relinearize_rescale(L, a, a, keys)
    ms, cnt = C.c_double(0.0), C.c_uint64(0)
    pkg.capi.check(lib.troyn_kernel_timer_read(0, C.byref(ms), C.byref(cnt)))
    assert cnt.value == 3 and 0.0 < ms.value < 1000.0
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 0))
    plan.ckks_multiply_relinearize_rescale(L, a, a, keys)
    pkg.capi.check(lib.troyn_kernel_timer_read(0, C.byref(ms), C.byref(cnt)))
    assert cnt.value == 0
    assert lib.troyn_kernel_timer_enable(7, 1) == -1
