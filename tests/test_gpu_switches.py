"""Every A/B switch of the library has both sides executed by the suite.  The switches are environment variables read ONCE, when a plan is
created (csrc/troyn.hip TroynOptions), and settable per plan (Plan.set_option = troyn_plan_set_option); the tests below set the environment
before they create their plan, so one process runs both sides; results must stay bit-identical to the oracle.

  TROYN_KS_MAC=split        separate NTT + accumulate launches (default: ksmac2_kernel / ksmaci_kernel; "v1" is accepted and ignored since round 5:
                            the first-generation kernel's N >= 8192 instantiations left the library)
  TROYN_KS_ORDER=row        row-major workgroup order of ksmac2_kernel (default: band / item)
  TROYN_KS_DIAG=loop        the diagonal digit of an NTT-form key switch as an iteration of ksmac2's digit loop (default: in its epilogue)
  TROYN_NTT_SMALL_TWO_PASS=0  N = 8192 / 16384 launches of a few limbs keep the whole-limb tile (default: the two-pass form of the larger rings, 4 workgroups per limb)
  TROYN_KS_MAC_SHOUP=0      integer inner product with Barrett-128 terms (default: the keys' Shoup quotients are prepared once per call, lazy Shoup terms)
  TROYN_KS_SPLIT=0 | 1      digit-parallel form of the inner product (one workgroup per digit + a reducer) off / forced on (default: on when the
                            launch would occupy at most half of the chip -- the batches of 8 used below take it, so "0" is the other side here)
  TROYN_MRR=calls           the fused multiply -> relinearize -> rescale entry composes the three public calls
  TROYN_NTT_HALF=<mask>     half-word LDS tiles per kernel variant of the whole-limb N = 16384 FP64 transforms (default 0x0127)
  TROYN_NTT_ARITH=u64       integer butterflies for every modulus
  TROYN_KS_MAC=fused        a few ciphertexts of a chain with moduli >= 2^50 keep the one-launch inner product (default: two launches below 256 workgroups;
                            tests/test_gpu_wide_rows.py runs every test in both forms)
  TROYN_MRR_SMALL=0 | serial  single objects at N = 8192 / 16384: unmerged tails / one thread per quartet (tests/test_gpu_keyswitch.py::test_fused_chain_single_objects)
  TROYN_BEHZ_LIFT=split     N = 32768 BFV multiply: lift and floor apart from the strided transform passes (test_bfv_multiply_lift_with_first_pass)
  TROYN_BFV_TENSOR=fused    tensor_core_kernel also for a few ciphertexts at the whole-limb sizes (the behz_gen fixture forces it for three of its four parameters)
(TROYN_BEHZ, TROYN_KS_TAIL, TROYN_BFV_TENSOR, TROYN_TENSOR_WGS, TROYN_PLAIN_MAC are covered by parametrised tests next to their kernels; the sweep draws
random sets of all of them.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(O, pkg, dev, n, bits, L, batch=8):
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    return q, ctx, plan, keys, dkeys


@pytest.mark.parametrize("env", [{}, {"TROYN_KS_MAC": "split"}, {"TROYN_KS_ORDER": "row"}, {"TROYN_NTT_ARITH": "u64"},
                                 {"TROYN_NTT_HALF": "0x3f3f"}, {"TROYN_NTT_HALF": "0"}, {"TROYN_KS_DIAG": "loop"}, {"TROYN_KS_SPLIT": "0"}, {"TROYN_KS_SPLIT": "1"},
                                 {"TROYN_KS_SPLIT": "0", "TROYN_KS_DIAG": "loop"}, {"TROYN_NTT_ARITH": "u64", "TROYN_KS_MAC_SHOUP": "0", "TROYN_KS_MAC": "fused"},
                                 {"TROYN_NTT_ARITH": "u64", "TROYN_KS_MAC": "fused"}],
                         ids=["default", "ks_mac_split", "ks_order_row", "ntt_arith_u64", "ntt_half_all", "ntt_half_none", "ks_diag_in_loop",
                              "ks_digits_serial", "ks_digits_parallel", "ks_digits_serial_diag_in_loop", "integer_inner_product_barrett_terms", "integer_inner_product_one_launch"])
@pytest.mark.parametrize("n,bits,L", [(16384, [50] * 6, 5), (8192, [40, 40, 40, 40], 3)])
def test_switch_key_under_every_switch(O, pkg, dev, monkeypatch, env, n, bits, L):
    for k in ("TROYN_KS_MAC", "TROYN_KS_ORDER", "TROYN_NTT_ARITH", "TROYN_NTT_HALF", "TROYN_KS_DIAG", "TROYN_KS_SPLIT", "TROYN_KS_MAC_SHOUP"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, bits, L)
    batch = 8
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])
    for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE):
        d0 = np.stack([ctx.random_ct(40 + i, 2, L) for i in range(batch)])
        dd = pkg.to_device(d0, dev)
        plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=assign, is_ckks=True, is_ntt_form=True)
        got = pkg.to_host(dd)
        for i in (0, 3, 7):
            assert np.array_equal(got[i], ctx.switch_key(L, True, tg[i], keys, assign=assign, dest=d0[i])), (env, assign, i)


@pytest.mark.parametrize("env", [{}, {"TROYN_MRR": "calls"}, {"TROYN_KS_ORDER": "row"}, {"TROYN_NTT_HALF": "0x3f3f"}, {"TROYN_NTT_HALF": "0"},
                                 {"TROYN_KS_SPLIT": "0"}, {"TROYN_KS_SPLIT": "1"}, {"TROYN_KS_SPLIT": "0", "TROYN_KS_ORDER": "row"}],
                         ids=["default", "mrr_calls", "ks_order_row", "ntt_half_all", "ntt_half_none", "ks_digits_serial", "ks_digits_parallel", "ks_digits_serial_order_row"])
def test_fused_chain_under_every_switch(O, pkg, dev, monkeypatch, env):
    for k in ("TROYN_MRR", "TROYN_KS_ORDER", "TROYN_NTT_HALF", "TROYN_KS_SPLIT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, L = 16384, 5
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [50] * 6, L)
    batch = 8
    a = np.stack([ctx.random_ct(11 + i, 2, L) for i in range(batch)])
    b = np.stack([ctx.random_ct(29 + i, 2, L) for i in range(batch)])
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(a, dev), pkg.to_device(b, dev), dkeys))
    for i in (0, 5, 7):
        e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a[i], b[i]), keys)
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, e)), (env, i)


@pytest.mark.parametrize("env,batch", [({"TROYN_MRR_CHUNK": "16"}, 40), ({"TROYN_MRR_CHUNK": "8", "TROYN_MRR_STREAMS": "3"}, 24), ({"TROYN_MRR_CHUNK": "136"}, 264), ({}, 264)],
                         ids=["chunks_16_16_8", "three_streams", "two_halves_ragged", "default_one_chunk"])
def test_fused_chain_chunked_on_internal_streams(O, pkg, dev, monkeypatch, env, batch):
    """the fused entry can cut a batch into chunks that alternate on internal streams (TROYN_MRR_CHUNK / TROYN_MRR_STREAMS, csrc/troyn.hip;
    the default is one chunk on the caller's stream); every item must be the oracle's result, whatever the chunking, including a ragged
    last chunk"""
    for k in ("TROYN_MRR_CHUNK", "TROYN_MRR_STREAMS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, L = 8192, 3
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [50] * 4, L)
    base_a = np.stack([ctx.random_ct(11 + i, 2, L) for i in range(8)])
    base_b = np.stack([ctx.random_ct(29 + i, 2, L) for i in range(8)])
    a = np.concatenate([base_a] * (batch // 8))
    b = np.concatenate([np.roll(base_b, i, axis=0) for i in range(batch // 8)])      # item i of block r: a[i] x b[(i - r) mod 8]
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(a, dev), pkg.to_device(b, dev), dkeys))
    cache = {}
    for item in sorted({0, 7, 8, 15, 16, 23, batch // 2, batch // 2 - 1, batch - 9, batch - 1}):
        i, r = item % 8, item // 8
        key = (i, (i - r) % 8)
        if key not in cache:
            e = ctx.relinearize(L, True, ctx.ckks_multiply(L, base_a[key[0]], base_b[key[1]]), keys)
            cache[key] = ctx.mod_switch_scale_to_next(L, e)
        assert np.array_equal(got[item], cache[key]), (env, item)


@pytest.mark.parametrize("half", ["0x3f3f", "0", "0x0021", "0x0027", "0x0127"])
def test_ntt_half_word_variants(O, pkg, dev, monkeypatch, half):
    """whole-limb N = 16384 transforms, forward and inverse, plain and rescale-fused, with every half-word LDS variant on / off"""
    monkeypatch.setenv("TROYN_NTT_HALF", half)
    n, L = 16384, 4
    q = O.coeff_modulus_create(n, [50] * 4)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, 14, q)
    x = np.stack([ctx.random_ct(3 + i, 2, L) for i in range(3)])
    fw = pkg.to_host(plan.ntt(pkg.to_device(x, dev), 2, L))
    for i in range(3):
        assert np.array_equal(fw[i], ctx.to_ntt(x[i], 2, L))
    back = pkg.to_host(plan.ntt(pkg.to_device(fw, dev), 2, L, inverse=True))
    assert np.array_equal(back, x)
    got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(fw, dev), 2))
    for i in range(3):
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, fw[i]))


@pytest.mark.parametrize("n,bits,L,batch,ckks", [(16384, [50] * 6, 5, 128, True), (16384, [50] * 5, 4, 256, True), (8192, [40] * 4, 3, 256, True),
                                                  (8192, [60, 40, 40, 60], 3, 256, True), (8192, [50, 50, 60, 40, 50], 4, 256, True),
                                                  (32768, [50] * 4, 3, 64, False), (32768, [50] * 5, 4, 128, True)],
                         ids=["n16384_6rows", "n16384_5rows_odd", "n8192_4rows", "n8192_mixed_chain", "n8192_mixed_odd", "n32768_bfv", "n32768_5rows_odd"])
def test_key_switch_workgroup_orders(O, pkg, dev, monkeypatch, n, bits, L, batch, ckks):
    """ksmac2_kernel's workgroup orders (csrc/ksmac_kernels.hpp): the default for batches that fill whole bands is `band` (two rows x the
    items that fill an XCD); item / row / plain must give the same words, and EVERY item is checked against the oracle.  Odd row
    counts (a band with one row) and mixed chains (FP64 rows picked by a mask) included."""
    monkeypatch.delenv("TROYN_KS_ORDER", raising=False)
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q) if ckks else O.Context("bfv", n, q, 65537)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    keys = ctx.random_keys(2024, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(batch)])      # every item its own operand: a permutation inside a group of 8 cannot hide
    dtg = pkg.to_device(tg, dev)
    res = {}
    for order in ("band", "item", "row", "plain"):
        plan.set_option("TROYN_KS_ORDER", order)
        dd = pkg.to_device(np.zeros((batch, 2, L, n), dtype=np.uint64), dev)
        plan.switch_key(L, dtg, dkeys, dest=dd, assign=pkg.ASSIGN_OVERWRITE, is_ckks=ckks, is_ntt_form=ckks)
        res[order] = pkg.to_host(dd)
    for order in ("item", "row", "plain"):
        assert np.array_equal(res["band"], res[order]), order
    d0 = np.zeros((2, L, n), dtype=np.uint64)
    for item in range(batch):           # ALL items against the oracle; the other orders equal this one word for word
        want = ctx.switch_key(L, ckks, tg[item], keys, assign=pkg.ASSIGN_OVERWRITE, dest=d0)
        assert np.array_equal(res["band"][item], want), item


@pytest.mark.parametrize("n,batch", [(16384, 1), (16384, 13), (16384, 129), (16384, 136), (16384, 256), (8192, 5), (8192, 264), (32768, 3), (32768, 64)])
def test_fused_chain_batch_shapes(O, pkg, dev, monkeypatch, n, batch):
    """the fused entry on batches that select every workgroup order of its inner product (a multiple of the band size: band; a multiple
    of 8: item; anything else: plain) and leave ragged last groups in the transform kernels' XCD placement; every item is the oracle's"""
    for k in ("TROYN_MRR_CHUNK", "TROYN_MRR_STREAMS", "TROYN_KS_ORDER"):
        monkeypatch.delenv(k, raising=False)
    L = 3
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [50] * 4, L)
    base_a = np.stack([ctx.random_ct(11 + i, 2, L) for i in range(4)])
    base_b = np.stack([ctx.random_ct(29 + i, 2, L) for i in range(4)])
    ia = np.arange(batch) % 4
    ib = (np.arange(batch) // 4 + np.arange(batch)) % 4
    got = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, pkg.to_device(base_a[ia], dev), pkg.to_device(base_b[ib], dev), dkeys))
    cache = {}
    for item in range(batch):
        key = (int(ia[item]), int(ib[item]))
        if key not in cache:
            cache[key] = ctx.mod_switch_scale_to_next(L, ctx.relinearize(L, True, ctx.ckks_multiply(L, base_a[key[0]], base_b[key[1]]), keys))
        assert np.array_equal(got[item], cache[key]), item


@pytest.mark.parametrize("n,bits,L", [(8192, [50, 50], 1), (16384, [50, 50, 50], 1), (32768, [50, 50, 50], 1), (8192, [40, 40, 40], 2)])
@pytest.mark.parametrize("scheme", ["ckks", "bfv"])
def test_relinearize_shortest_chains(O, pkg, dev, n, bits, L, scheme):
    """L = 1: the digit loop of ksmac2's DG instantiation (NTT form: the only digit is the diagonal one, applied in the epilogue) runs zero
    times for a data row; coefficient form takes the NODIAG instantiation.  Both against the oracle."""
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q) if scheme == "ckks" else O.Context("bfv", n, q, 65537)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    ntt = scheme == "ckks"
    ct3 = np.stack([ctx.random_ct(70 + i, 3, L) for i in range(8)])
    got = pkg.to_host(plan.relinearize(L, pkg.to_device(ct3, dev), dkeys, is_ckks=ntt, is_ntt_form=ntt))
    for i in range(8):
        assert np.array_equal(got[i], ctx.relinearize(L, ntt, ct3[i], keys)), i


@pytest.mark.parametrize("split", ["0", "1"])
@pytest.mark.parametrize("scheme,ntt_form,n,bits,L,batch", [("bfv", False, 32768, [50] * 4, 3, 2), ("ckks", True, 32768, [50] * 4, 3, 1), ("bfv", False, 8192, [40, 40, 40], 2, 3),
                                                            ("ckks", True, 16384, [50] * 6, 5, 1), ("ckks", True, 16384, [45, 45, 45], 2, 5), ("ckks", True, 8192, [50] * 4, 2, 64)])
def test_digit_parallel_inner_product(O, pkg, dev, monkeypatch, split, scheme, ntt_form, n, bits, L, batch):
    """the digit-parallel form of the key-switch inner product (small batches: ksmac2 SPLITJ + ksmac_split_reduce_kernel) against the oracle,
    next to the serial form on the same operands: the three ring sizes, coefficient-form and NTT-form targets (no / diagonal-digit reducer
    epilogue), a lower level, a single item and the largest batch the slots are provisioned for"""
    monkeypatch.setenv("TROYN_KS_SPLIT", split)
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(scheme, n, q, 1032193 if scheme == "bfv" else 0)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([ctx.random_ct(5 + (i % 7), 1, L)[0] for i in range(batch)])
    d0 = np.stack([ctx.random_ct(40 + (i % 5), 2, L) for i in range(batch)])
    dd = pkg.to_device(d0, dev)
    plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=pkg.ASSIGN_ADD_INPLACE, is_ckks=scheme == "ckks", is_ntt_form=ntt_form)
    got = pkg.to_host(dd)
    for i in sorted({0, batch // 2, batch - 1}):
        assert np.array_equal(got[i], ctx.switch_key(L, ntt_form, tg[i], keys, assign=pkg.ASSIGN_ADD_INPLACE, dest=d0[i])), (split, i)


@pytest.mark.parametrize("two_pass", ["1", "0"])
@pytest.mark.parametrize("bits", [[50] * 6, [60, 50, 50, 50, 50, 60]])
def test_small_launches_at_n16384(O, pkg, dev, monkeypatch, two_pass, bits):
    """a single N = 16384 ciphertext: its transforms are launches of 2 - 10 limb-polynomials, which take the two-pass kernels (4 workgroups per limb and pass)
    instead of one whole-limb workgroup per limb; TROYN_NTT_SMALL_TWO_PASS=0 is the other side.  Plain transforms (both directions, FP64 and mixed
    arithmetic classes), the separate key switch and the fused chain against the oracle."""
    monkeypatch.setenv("TROYN_NTT_SMALL_TWO_PASS", two_pass)
    n, L = 16384, 5
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, 14, q)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    a = ctx.random_ct(11, 2, L)
    b = ctx.random_ct(29, 2, L)
    da, db = pkg.to_device(a[None], dev), pkg.to_device(b[None], dev)
    # forward / inverse round trip and the oracle's transform
    f = plan.ntt(da, 2, L)
    assert np.array_equal(pkg.to_host(f)[0], ctx.to_ntt(a, 2, L))
    assert np.array_equal(pkg.to_host(plan.ntt(f, 2, L, inverse=True))[0], a)
    # separate calls and the fused chain
    e = ctx.relinearize(L, True, ctx.ckks_multiply(L, a, b), keys)
    prod = plan.dyadic_convolute(da, 2, db, 2, L)
    relin = plan.relinearize(L, prod, dkeys, is_ckks=True, is_ntt_form=True)
    assert np.array_equal(pkg.to_host(relin)[0], e)
    want = ctx.mod_switch_scale_to_next(L, e)
    assert np.array_equal(pkg.to_host(plan.divide_and_round_q_last_ntt(L, relin, 2))[0], want)
    assert np.array_equal(pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys))[0], want)


def test_environment_is_read_only_at_plan_creation(O, pkg, dev, monkeypatch):
    """No getenv on a call path: the environment is read inside troyn_plan_create only, so flipping it afterwards changes nothing, while
    troyn_plan_set_option does.  Observable through the library's kernel timer: the fused inner product is ONE timed launch per key switch, the
    unfused chain (TROYN_KS_MAC=split) has none.  An unknown option name is an error.  Results equal the oracle throughout."""
    n, L = 8192, 3
    monkeypatch.setenv("TROYN_KS_ORDER", "row")
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [40, 40, 40, 40], L)
    monkeypatch.setenv("TROYN_KS_ORDER", "definitely-not-an-order")     # ignored: the plan exists already
    monkeypatch.setenv("TROYN_NTT_ARITH", "u64")
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        plan.set_option("TROYN_NO_SUCH_SWITCH", "1")
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(8)])
    want = [ctx.switch_key(L, True, tg[i], keys, assign=pkg.ASSIGN_OVERWRITE) for i in (0, 7)]
    t0 = _ks_launches(pkg, plan, lambda: plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    plan.set_option("TROYN_KS_MAC", "split")       # the unfused path has no inner-product launch inside the timer region
    t1 = _ks_launches(pkg, plan, lambda: plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    plan.set_option("TROYN_KS_MAC", None)
    t2 = _ks_launches(pkg, plan, lambda: plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    assert (t0, t1, t2) == (1, 0, 1), (t0, t1, t2)
    got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[7], want[1])


def _ks_launches(pkg, plan, fn):
    """inner-product launches the library's kernel timer saw while fn ran (TROYN_TIMER_KS_INNER_PRODUCT)"""
    import ctypes as C
    import torch
    lib = plan.lib
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 1))
    fn()
    torch.cuda.synchronize()
    ms, cnt = C.c_double(), C.c_uint64()
    pkg.capi.check(lib.troyn_kernel_timer_read(0, C.byref(ms), C.byref(cnt)))
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 0))
    return int(cnt.value)


@pytest.mark.parametrize("split", ["", "1"])
def test_many_digits_never_take_the_digit_parallel_form(O, pkg, dev, monkeypatch, split):
    """the reducer of the digit-parallel inner product adds its L slots in plain doubles, exact only for L <= 15 (|slot| <= p/2 + 1, p < 2^50);
    L = 17 digits of 50-bit primes on a single ciphertext -- a launch small enough to want the digit-parallel form, also when it is forced on -- must
    take the serial form (accumulators re-centred every 8 digits) and equal the oracle"""
    if split:
        monkeypatch.setenv("TROYN_KS_SPLIT", split)
    else:
        monkeypatch.delenv("TROYN_KS_SPLIT", raising=False)
    n, L = 8192, 17
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [50] * 18, L)
    tg = np.stack([ctx.random_ct(5, 1, L)[0], np.stack([np.full(n, q[l] - 1, dtype=np.uint64) for l in range(L)])])
    for ntt_form, c in ((True, ctx), (False, O.Context("bfv", n, q, 65537))):
        got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=ntt_form, is_ntt_form=ntt_form))
        for i in range(2):
            assert np.array_equal(got[i], c.switch_key(L, ntt_form, tg[i], keys, assign=pkg.ASSIGN_OVERWRITE)), (ntt_form, i)


def test_option_values_are_validated(O, pkg, dev, monkeypatch):
    """a value an option does not have is an error (TROYN_TENSOR_WGS=foo used to parse as 0 and select the two-workgroup kernel; a misspelt word
    silently meant "default").  None / "" stay the default; a bad value in the environment fails troyn_plan_create; a refused value changes nothing."""
    for k in ("TROYN_KS_ORDER", "TROYN_TENSOR_WGS", "TROYN_KS_MAC"):
        monkeypatch.delenv(k, raising=False)
    n, L = 8192, 3
    q, ctx, plan, keys, dkeys = _case(O, pkg, dev, n, [40, 40, 40, 40], L)
    for name, bad in (("TROYN_TENSOR_WGS", "foo"), ("TROYN_TENSOR_WGS", "5"), ("TROYN_KS_ORDER", "definitely-not-an-order"), ("TROYN_KS_MAC", "v1"),
                      ("TROYN_NTT_HALF", "0x1ffff"), ("TROYN_MRR_STREAMS", "9"), ("TROYN_NTT_ARITH", "u32"), ("TROYN_PLAIN_MAC", "triple"), ("TROYN_NTT_OVERLAP", "off")):
        with pytest.raises(pkg.capi.TroynInvalidArgument):
            plan.set_option(name, bad)
    for name, good in (("TROYN_TENSOR_WGS", "3"), ("TROYN_TENSOR_WGS", None), ("TROYN_KS_ORDER", "row"), ("TROYN_KS_ORDER", ""), ("TROYN_NTT_HALF", "0x3f3f"),
                       ("TROYN_NTT_HALF", None), ("TROYN_KS_MAC", "fused"), ("TROYN_KS_MAC", None)):
        plan.set_option(name, good)
    tg = np.stack([ctx.random_ct(5 + i, 1, L)[0] for i in range(2)])
    got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    assert np.array_equal(got[1], ctx.switch_key(L, True, tg[1], keys, assign=pkg.ASSIGN_OVERWRITE))
    monkeypatch.setenv("TROYN_KS_ORDER", "definitely-not-an-order")
    with pytest.raises(pkg.capi.TroynInvalidArgument):
        pkg.Plan(dev, 13, q)


def test_behz_follows_later_plan_options(O, pkg, dev, monkeypatch):
    """troyn_behz_create snapshots the plan's options into the auxiliary plan; a later set_option on the plan must reach it (both bases of one
    multiply on the same transform policies).  Observable: results stay the oracle's for switches set AFTER the handle was created, and the
    integer policy forced afterwards gives the same words as a plan created with it."""
    for k in ("TROYN_NTT_ARITH", "TROYN_BFV_TENSOR", "TROYN_NTT_HALF"):
        monkeypatch.delenv(k, raising=False)
    n, bits, L, t = 8192, [40, 40, 40], 2, 1032193
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, 13, q)
    behz = pkg.Behz(plan, L, t)
    a = np.stack([ctx.random_ct(31 + i, 2, L) for i in range(2)])
    b = np.stack([ctx.random_ct(47 + i, 2, L) for i in range(2)])
    want = [ctx.bfv_multiply(L, a[i], b[i]) for i in range(2)]
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    for name, value in ((None, None), ("TROYN_NTT_ARITH", "u64"), ("TROYN_BFV_TENSOR", "fused"), ("TROYN_NTT_HALF", "0"), ("TROYN_NTT_ARITH", None)):
        if name:
            plan.set_option(name, value)
        got = pkg.to_host(behz.multiply(da, 2, db, 2))
        for i in range(2):
            assert np.array_equal(got[i], want[i]), (name, value, i)
