"""GPU parity of the RLWE / LWE packing path (evaluator_lwes.cu): negacyclic shift, division by the
degree, LWE extraction and Evaluator::pack_rlwe_ciphertexts (the matmul application's pack_outputs) against the oracle's
restatement of the reference's host branch, on uniform residues and uniform key material (every step is defined on
arbitrary residues), plus one run with genuine keys that is decrypted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(O, pkg, dev, n, bits, t=257):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    return O.Context("bfv", n, q, t), pkg.Plan(dev, n.bit_length() - 1, q), q, len(q) - 1


@pytest.mark.parametrize("n,bits", [(64, [30, 30, 30]), (8192, [60, 40, 40, 60]), (4096, [36, 36, 37])])
def test_shift_inv_degree_extract(O, pkg, dev, n, bits):
    ctx, plan, q, L = _setup(O, pkg, dev, n, bits)
    a = ctx.random_ct(3, 2, L)
    a[0, 0, :3] = 0                                            # zero coefficients keep their sign-less representation
    da = pkg.to_device(a, dev)
    for shift in (0, 1, n - 1, n, n + 5, 2 * n - 1):
        assert np.array_equal(pkg.to_host(plan.negacyclic_shift(da, L, shift)), ctx.negacyclic_shift(L, a, shift)), shift
    # (shift = 2N is the identity: the reference reduces the shift implicitly, test_negacyclic_shift_takes_any_shift below)
    assert np.array_equal(pkg.to_host(plan.negacyclic_shift(da, L, 2 * n)), a)
    for scalar in (1, 2, n // 4):
        assert np.array_equal(pkg.to_host(plan.multiply_inv_degree(da, L, scalar)), ctx.multiply_inv_degree(L, a, scalar))
    b = ctx.random_ct(4, 2, L)
    db = pkg.to_device(b, dev)
    terms = [0, 1, n - 1, n // 2 + 3]
    c0, c1 = plan.extract_lwe(L, [da, db, da, db], terms)
    c0, c1 = pkg.to_host(c0), pkg.to_host(c1)
    for i, (src, term) in enumerate(zip([a, b, a, b], terms)):
        w0, w1 = ctx.extract_lwe(L, src, term)
        assert np.array_equal(c0[i], w0) and np.array_equal(c1[i], w1), i


@pytest.mark.parametrize("n,bits,interval,out_interval,sizes", [
    (64, [30, 30, 30], 4, 1, [4, 3, 1]),                      # full, ragged and single-member groups
    (64, [30, 30, 30], 8, 2, [3, 4]),                         # output_interval > 1: field trace at the end
    (1024, [40, 40, 40, 40], 4, 1, [4, 2]),
    (8192, [60, 40, 40, 60], 4, 1, [4]),                      # BASELINE config 5 parameters, input_block = 4
])
def test_pack_rlwe_matches_oracle(O, pkg, dev, n, bits, interval, out_interval, sizes):
    ctx, plan, q, L = _setup(O, pkg, dev, n, bits)
    maxc = interval // out_interval
    elements = [(n // interval) * (1 << (layer + 1)) + 1 for layer in range(maxc.bit_length() - 1)]
    d = n
    while out_interval != 1 and d > n // out_interval:
        elements.append(d + 1)
        d >>= 1
    keys_h = {g: ctx.random_keys(1000 + g, L) for g in set(elements)}
    keys_d = {g: [pkg.to_device(k, dev) for k in ks] for g, ks in keys_h.items()}
    shift = 2 * n - (interval - 1)
    groups_h = [[ctx.random_ct(50 * gi + i + 1, 2, L) for i in range(sz)] for gi, sz in enumerate(sizes)]
    groups_d = [[pkg.to_device(c, dev) for c in grp] for grp in groups_h]
    got = pkg.to_host(plan.pack_rlwe_ciphertexts(L, groups_d, keys_d, shift, interval, out_interval))
    for gi, grp in enumerate(groups_h):
        want = ctx.pack_rlwe_ciphertexts(L, grp, keys_h, shift, interval, out_interval)
        assert np.array_equal(got[gi], want), gi


def test_pack_rlwe_decrypts(O, pkg, dev):
    """genuine keys: the packed ciphertext holds coefficient k*I + I-1 of input j at position k*I + j (what
    MatmulHelper::pack_outputs relies on, app/matmul.cu:572-619)"""
    n, I, t = 256, 4, 257
    ctx, plan, q, L = _setup(O, pkg, dev, n, [36, 36, 37], t)
    rng = O.Rng(11)
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    rs = np.random.RandomState(3)
    msgs = [rs.randint(0, t, n).astype(np.uint64) for _ in range(3)]
    cts = [ctx.encrypt_asymmetric_bfv(rng, pk, m) for m in msgs]
    keys_h = {(n // I) * (1 << (layer + 1)) + 1: None for layer in range(2)}
    for g in keys_h:
        keys_h[g] = ctx.galois_key(rng, sk, g)
    keys_d = {g: [pkg.to_device(k, dev) for k in ks] for g, ks in keys_h.items()}
    out = pkg.to_host(plan.pack_rlwe_ciphertexts(L, [[pkg.to_device(c, dev) for c in cts]], keys_d, 2 * n - (I - 1), I, 1))[0]
    assert np.array_equal(out, ctx.pack_rlwe_ciphertexts(L, cts, keys_h, 2 * n - (I - 1), I, 1))
    dec = ctx.decrypt_bfv(sk, out)
    want = np.zeros(n, dtype=np.uint64)
    for j, m in enumerate(msgs):
        want[j::I] = m[I - 1::I]
    assert np.array_equal(dec, want)


def test_negacyclic_shift_takes_any_shift(O, pkg, dev):
    """the reference reduces the shift implicitly (index (shift + k) & (N - 1), sign from bit log2 N: utils/poly_small_mod.cu:927-944); its own test packs with
    shift = 2N + (a non-positive offset) (test/lwe.cu:223): shift = 2N is the identity, 2N + s == s, 3N == N (negation)"""
    n, L = 1024, 2
    q = O.coeff_modulus_create(n, [40, 40, 41])
    ctx = O.Context("bfv", n, q, 65537)
    plan = pkg.Plan(dev, 10, q)
    x = ctx.random_ct(3, 2, L)
    dx = pkg.to_device(x[None], dev)
    base = {s: pkg.to_host(plan.negacyclic_shift(dx.view(-1, L, n), L, s)) for s in (0, 5, n)}
    for s, ref in ((2 * n, 0), (2 * n + 5, 5), (3 * n, n), (4 * n + 5, 5)):
        assert np.array_equal(pkg.to_host(plan.negacyclic_shift(dx.view(-1, L, n), L, s)), base[ref]), s
    assert np.array_equal(base[0].reshape(x.shape), x)
