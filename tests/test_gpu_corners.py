"""GPU parity on the corners of the exact-FP64 arithmetic policy (csrc/dev_math_f64.hpp, ksmac_kernels.hpp).

The optimised kernels carry residues below 2^50 as integer-valued doubles and rely on magnitude bounds (a block of 5 butterfly layers
that starts re-centred stays below 2^53).  Uniform random residues sit in the middle of those bounds; the vectors here sit on their
edges: every coefficient q-1, alternating 0 / q-1, the largest lazy inputs the reference allows (4q-1 forward, 2q-1 inverse), impulses
in every position class, (q-1)/2 and (q+1)/2 (the re-centring boundary) -- for the six largest 50-bit primes (BASELINE config 3's
chain, the primes closest to the 2^50 limit of the policy) and for a chain that straddles 2^50 (the policy switches to the integer
butterflies).  Expected values come from the oracle's integer arithmetic; everything is compared bit for bit through the C-ABI.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _is_prime(n):
    if n < 2:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % p == 0:
            return n == p
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def _first_prime_above(bound, factor):
    v = (bound // factor + 1) * factor + 1
    while not _is_prime(v):
        v += factor
    return v


def _patterns(q, n, lazy_mult):
    """corner polynomials of one limb: name -> uint64[n] (values below lazy_mult * q)"""
    top = lazy_mult * q - 1
    idx = np.arange(n)
    pats = {
        "all_q-1": np.full(n, q - 1, dtype=np.uint64),
        "alt_0_q-1": np.where(idx % 2 == 0, 0, q - 1).astype(np.uint64),
        "alt_q-1_0": np.where(idx % 2 == 0, q - 1, 0).astype(np.uint64),
        "blocks32": np.where((idx // 32) % 2 == 0, q - 1, 1).astype(np.uint64),
        "lazy_top": np.full(n, top, dtype=np.uint64),
        "half_lo": np.full(n, (q - 1) // 2, dtype=np.uint64),
        "half_hi": np.full(n, (q + 1) // 2, dtype=np.uint64),
        "ramp_top": ((q - 1 - idx) % q).astype(np.uint64),
    }
    for pos in (0, 1, 31, 32, n // 2 - 1, n // 2, n - 1):
        v = np.zeros(n, dtype=np.uint64)
        v[pos] = q - 1
        pats["impulse_%d" % pos] = v
    return pats


def _chains(O, n):
    """(name, moduli): the largest 50-bit primes = 1 mod 2n (FP64 policy) and a chain straddling 2^50 (integer policy)"""
    below = O.coeff_modulus_create(n, [50] * 6)
    above = _first_prime_above(1 << 50, 2 * n)
    assert max(below) < (1 << 50) < above
    # the largest 61-bit primes (the BEHZ auxiliary base): the integer butterflies keep values below 8q, i.e. just under 2^64 here
    top = O.get_primes(2 * n, 61, 3)
    assert all((1 << 60) < v < (1 << 61) for v in top)
    return [("six_50bit", below), ("straddle_2^50", [max(below), above, min(below)]), ("three_61bit", top)]


@pytest.mark.parametrize("log_n", [13, 14, 15])
def test_ntt_corner_vectors(O, pkg, dev, log_n):
    n = 1 << log_n
    for cname, q in _chains(O, n):
        L = len(q)
        tables = [O.NTTTables(log_n, qi) for qi in q]
        plan = pkg.Plan(dev, log_n, q)
        for inverse, lazy in ((False, 4), (True, 2)):
            names = sorted(_patterns(q[0], n, lazy))
            x = np.stack([np.stack([_patterns(q[j], n, lazy)[nm] for j in range(L)]) for nm in names])   # [pattern][limb][n]
            d = pkg.to_device(x.reshape(len(names), 1, L, n), dev)
            plan.ntt(d, 1, L, inverse=inverse)
            got = pkg.to_host(d).reshape(len(names), L, n)
            exp = x.copy().reshape(-1)
            (O.ntt_inverse if inverse else O.ntt_forward)(exp, len(names), L, log_n, tables)
            exp = exp.reshape(len(names), L, n)
            for i, nm in enumerate(names):
                assert np.array_equal(got[i], exp[i]), "%s N=%d %s pattern %s" % (cname, n, "INTT" if inverse else "NTT", nm)


def _corner_ct(q, L, n, pcount, name):
    return np.stack([np.stack([_patterns(q[j], n, 1)[name] for j in range(L)]) for _ in range(pcount)])


@pytest.mark.parametrize("n,bits,L", [(16384, [50] * 6, 5), (8192, [50] * 6, 5), (8192, [40] * 11, 10), (4096, [36] * 10, 9)])
def test_switch_key_corner_vectors(O, pkg, dev, n, bits, L):
    """the fused inner product: extreme targets against random keys AND against keys that are all q-1 (largest <digit, key> terms;
    L = 9 / 10 also crosses the accumulators' re-centring after 8 digits)"""
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    K = len(q)
    rkeys = ctx.random_keys(3, L)
    ckeys = [np.stack([np.stack([np.full(n, q[k] - 1, dtype=np.uint64) for k in range(K)]) for _ in range(2)]) for _ in range(L)]
    for keys in (rkeys, ckeys):
        dkeys = [pkg.to_device(k, dev) for k in keys]
        names = ["all_q-1", "alt_0_q-1", "half_hi", "impulse_0", "impulse_%d" % (n - 1), "ramp_top"]
        tg = np.stack([_corner_ct(q, L, n, 1, nm)[0] for nm in names] + [ctx.random_ct(5, 1, L)[0], ctx.random_ct(6, 1, L)[0]])   # batch of 8
        for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE):
            d0 = np.stack([_corner_ct(q, L, n, 2, "all_q-1") for _ in range(tg.shape[0])])
            dd = pkg.to_device(d0, dev)
            plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=assign, is_ckks=True, is_ntt_form=True)
            got = pkg.to_host(dd)
            for i in range(tg.shape[0]):
                exp = ctx.switch_key(L, True, tg[i], keys, assign=assign, dest=d0[i])
                assert np.array_equal(got[i], exp), "item %d assign %d" % (i, assign)


def test_switch_key_straddling_moduli(O, pkg, dev):
    """a key chain with one modulus above 2^50: every fused kernel of the key switch takes the integer butterflies"""
    n, L = 8192, 2
    q = _chains(O, n)[1][1]
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, 13, q)
    keys = ctx.random_keys(3, L)
    dkeys = [pkg.to_device(k, dev) for k in keys]
    tg = np.stack([_corner_ct(q, L, n, 1, nm)[0] for nm in ("all_q-1", "alt_q-1_0", "impulse_1", "half_lo")])
    got = pkg.to_host(plan.switch_key(L, pkg.to_device(tg, dev), dkeys, assign=pkg.ASSIGN_OVERWRITE, is_ckks=True, is_ntt_form=True))
    for i in range(tg.shape[0]):
        assert np.array_equal(got[i], ctx.switch_key(L, True, tg[i], keys, assign=pkg.ASSIGN_OVERWRITE)), i


@pytest.mark.parametrize("n,bits,L", [(16384, [50] * 6, 5), (8192, [50] * 4, 4), (32768, [50] * 4, 3)])
def test_rescale_corner_vectors(O, pkg, dev, n, bits, L):
    """fused CKKS rescale (INTT of the last limb, then the forward NTT with the rounding prologue and the divide epilogue)"""
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    names = ["all_q-1", "alt_0_q-1", "alt_q-1_0", "half_lo", "half_hi", "impulse_0", "impulse_%d" % (n // 2), "ramp_top"]
    x = np.stack([_corner_ct(q, L, n, 2, nm) for nm in names])
    got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(x, dev), 2))
    for i, nm in enumerate(names):
        assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i])), nm


@pytest.mark.parametrize("n,bits,L", [(16384, [50] * 6, 5), (8192, [50] * 5, 4), (32768, [50] * 4, 3), (8192, [50] * 12, 11), (16384, [50] * 11, 9), (8192, [50] * 3, 2)])
@pytest.mark.parametrize("key_kind", ["random", "all_q-1"])
def test_pipeline_corner_vectors(O, pkg, dev, n, bits, L, key_kind):
    """multiply -> relinearize -> rescale on extreme operands, through the three calls AND through the fused entry (the path bench.py
    times): the fused tail sums scale_by + two tensor products - y to ~3.1 p before its re-centring, the MULPAIR / LAST_LIMB loaders and
    the double-format digits have their own range assumptions -- all exercised at the edges here, with random keys and keys that are all q-1.
    L = 9 / 11 cross the accumulators' re-centring after 8 digits with the diagonal digit and the tensor terms added in ksmac2's epilogue
    (round 3), L = 2 is the shortest chain the fused entry takes."""
    q = O.coeff_modulus_create(n, bits)
    K = len(q)
    ctx = O.Context("ckks", n, q)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    if key_kind == "random":
        keys = ctx.random_keys(11, L)
    else:
        keys = [np.stack([np.stack([np.full(n, q[k] - 1, dtype=np.uint64) for k in range(K)]) for c in range(2)]) for j in range(L)]
    dkeys = [pkg.to_device(k, dev) for k in keys]
    names = ["all_q-1", "alt_0_q-1", "half_hi", "ramp_top", "alt_q-1_0", "half_lo", "impulse_0", "blocks32"]     # 8 items: the XCD-grouped order
    a = np.stack([_corner_ct(q, L, n, 2, nm) for nm in names])
    b = np.stack([_corner_ct(q, L, n, 2, nm) for nm in reversed(names)])
    b[0] = a[0]                                                    # all q-1 times all q-1
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    prod = plan.dyadic_convolute(da, 2, db, 2, L)
    relin = plan.relinearize(L, prod, dkeys, is_ckks=True, is_ntt_form=True)
    got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, relin, 2))
    fused = pkg.to_host(plan.ckks_multiply_relinearize_rescale(L, da, db, dkeys))
    for i in range(len(names)):
        e = ctx.ckks_multiply(L, a[i], b[i])
        e = ctx.relinearize(L, True, e, keys)
        e = ctx.mod_switch_scale_to_next(L, e)
        assert np.array_equal(got[i], e), ("three calls", names[i])
        assert np.array_equal(fused[i], e), ("fused entry", names[i])


@pytest.mark.parametrize("n,bits,L,t", [(4096, [50] * 6, 5, 65537), (2048, [59] * 6, 5, 40961), (1024, [36] * 11, 10, 12289)])
def test_bfv_multiply_corner_vectors(O, pkg, dev, n, bits, L, t, behz_gen):
    """BEHZ conversions on extreme residues: the carry-free partial sums of behz2_kernels.hpp are largest when every residue is q-1"""
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    behz = pkg.Behz(plan, L, t)
    names = ["all_q-1", "alt_0_q-1", "half_hi", "ramp_top", "impulse_0"]
    a = np.stack([_corner_ct(q, L, n, 2, nm) for nm in names])
    b = np.stack([_corner_ct(q, L, n, 2, nm) for nm in reversed(names)])
    got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), 2, pkg.to_device(b, dev), 2))
    for i in range(len(names)):
        assert np.array_equal(got[i], ctx.bfv_multiply(L, a[i], b[i])), names[i]


@pytest.mark.parametrize("n,bits", [(8192, [40, 40, 60]), (16384, [40, 40, 40, 60]), (8192, [60, 40, 40, 60]), (8192, [50, 40, 60, 40, 60]), (32768, [40, 40, 60]), (32768, [60, 40, 60])],
                         ids=["n8192_40_40_60", "n16384_40_40_40_60", "n8192_60_40_40_60", "n8192_50_40_60_40_60", "n32768_40_40_60", "n32768_60_40_60"])
def test_wide_dropped_prime_over_narrow_limbs(O, pkg, dev, monkeypatch, n, bits):
    """the prime a fused tail divides by (special prime of the key switch, last limb of the rescale) has 60 bits while data limbs are
    narrow: the narrow limbs take the FP64 butterflies and their loader must reduce the 60-bit word with integer arithmetic first
    (ArithF64::load_io); chains of both classes split their tail / rescale launches by class.  Key switch in NTT and coefficient form,
    three assign methods, and rescale at every level; each with the launches of mixed chains split by class (TROYN_NTT_SPLIT=1), with the
    default (only plain transforms split) and with no split at all (=0: every limb follows the widest into the integer butterflies)."""
    q = O.coeff_modulus_create(n, bits)
    K = len(q)
    log_n = n.bit_length() - 1
    for split in ("1", "", "0"):
        monkeypatch.setenv("TROYN_NTT_SPLIT", split)
        for scheme, ntt_form in (("ckks", True), ("bfv", False)):
            ctx = O.Context(scheme, n, q) if scheme == "ckks" else O.Context(scheme, n, q, 65537)
            plan = pkg.Plan(dev, log_n, q)
            L = K - 1
            keys = ctx.random_keys(7, L)
            dkeys = [pkg.to_device(k, dev) for k in keys]
            names = ["all_q-1", "alt_q-1_0", "half_hi", "impulse_1"]
            tg = np.stack([_corner_ct(q, L, n, 1, nm)[0] for nm in names] + [ctx.random_ct(5 + i, 1, L)[0] for i in range(4)])
            for assign in (pkg.ASSIGN_OVERWRITE, pkg.ASSIGN_ADD_INPLACE, pkg.ASSIGN_OVERWRITE_EXCEPT_FIRST):
                d0 = np.stack([ctx.random_ct(50 + i, 2, L) for i in range(tg.shape[0])])
                dd = pkg.to_device(d0, dev)
                plan.switch_key(L, pkg.to_device(tg, dev), dkeys, dest=dd, assign=assign, is_ckks=ntt_form, is_ntt_form=ntt_form)
                got = pkg.to_host(dd)
                for i in range(tg.shape[0]):
                    assert np.array_equal(got[i], ctx.switch_key(L, ntt_form, tg[i], keys, assign=assign, dest=d0[i])), (split, scheme, assign, i)
        ctx = O.Context("ckks", n, q)
        plan = pkg.Plan(dev, log_n, q)
        for L in range(2, K + 1):
            x = np.stack([_corner_ct(q, L, n, 2, nm) for nm in ("all_q-1", "half_lo", "half_hi", "ramp_top")] + [ctx.random_ct(9 + i, 2, L) for i in range(4)])
            got = pkg.to_host(plan.divide_and_round_q_last_ntt(L, pkg.to_device(x, dev), 2))
            for i in range(x.shape[0]):
                assert np.array_equal(got[i], ctx.mod_switch_scale_to_next(L, x[i])), (split, "rescale", L, i)


@pytest.mark.parametrize("n,bits,L,t,pa,pb", [
    (8192, [40, 40, 40], 2, 1032193, 2, 2),                       # BASELINE config 2
    (8192, [40, 40, 40], 2, (1 << 40) - 87, 3, 3),                # the widest t the chain carries (t < q_i), 3-component operands
    (8192, [50, 50, 50, 50], 3, (1 << 49) - 69, 3, 2),
    (32768, [50] * 11, 10, 1032193, 2, 2),                        # BASELINE config 4
    (32768, [50] * 11, 10, (1 << 49) - 69, 2, 2),
    (16384, [50] * 6, 5, (1 << 30) + 3, 3, 3),
])
def test_bfv_multiply_worst_case_magnitudes(O, pkg, dev, n, bits, L, t, pa, pb):
    """the default auxiliary base of primes below 2^50 is sized by the reference's criterion (utils/rns_tool.cu:50-62) with two bits of slack;
    random operands do not reach the magnitudes the criterion is about.  Operands at the extremes -- every coefficient q_i - 1 (the centred value -1 in every
    limb is NOT the worst case, so also the residues of +-floor(q/2) and alternating signs), the widest plain modulus, 3-component ciphertexts (the tensor
    sums up to three products per output) -- against the oracle, which keeps the reference's 61-bit base; both bases of the library must agree with it."""
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    qL = [int(v) for v in q[:L]]
    Q = 1
    for v in qL:
        Q *= v
    half = Q // 2

    def const_poly(value):           # every coefficient = value (an integer in [0, Q)), as residues
        return np.stack([np.full(n, value % m, dtype=np.uint64) for m in qL])

    def alt_poly(v0, v1):
        rows = []
        for m in qL:
            r = np.empty(n, dtype=np.uint64)
            r[0::2] = v0 % m
            r[1::2] = v1 % m
            rows.append(r)
        return np.stack(rows)

    shapes = [const_poly(Q - 1), const_poly(half), const_poly(half + 1), alt_poly(half, half + 1), alt_poly(Q - 1, 1), alt_poly(half, Q - 1)]
    a = np.stack([np.stack([shapes[(i + p) % len(shapes)] for p in range(pa)]) for i in range(len(shapes))])
    b = np.stack([np.stack([shapes[(2 * i + p + 1) % len(shapes)] for p in range(pb)]) for i in range(len(shapes))])
    want = [ctx.bfv_multiply(L, a[i], b[i]) for i in range(len(shapes))]
    for base in (None, "ref"):
        plan.set_option("TROYN_BEHZ_BASE", base)
        behz = pkg.Behz(plan, L, t)
        got = pkg.to_host(behz.multiply(pkg.to_device(a, dev), pa, pkg.to_device(b, dev), pb))
        for i in range(len(shapes)):
            assert np.array_equal(got[i], want[i]), (base, i)
        sq = pkg.to_host(behz.multiply(pkg.to_device(a, dev), pa, pkg.to_device(a, dev), pa))
        for i in (0, 1, 3):
            assert np.array_equal(sq[i], ctx.bfv_multiply(L, a[i], a[i])), (base, "square", i)
