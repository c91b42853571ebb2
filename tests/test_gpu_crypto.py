"""GPU parity of the callers either side of the evaluator path: context PRNG samplers, key generation, BFV
encryption (Delta scaling) and decryption (decrypt_scale_and_round) -- against the oracle and against the
reference's own recorded config-1 digests.  Everything here runs through the C-ABI on the device."""
import numpy as np
import pytest
import torch

from test_oracle_config1 import G

pytestmark = pytest.mark.gpu


def _ctx(O, pkg, dev):
    n = G["poly_modulus_degree"]
    q = [int(v) for v in O.coeff_modulus_create(n, G["coeff_modulus_bits"])]
    t = G["plain_modulus"]
    return O.Context("bfv", n, q, t), pkg.Plan(dev, n.bit_length() - 1, q), q, t, n


def test_prng_block_and_samplers(O, pkg, dev):
    ctx, plan, q, t, n = _ctx(O, pkg, dev)
    ref, gpu = O.Rng(0xABCDEF, 7), pkg.Prng(plan, 0xABCDEF, 7)
    for nmod in (3, 2, 1):
        assert ref.sample_uint64() == gpu.sample_uint64()
        assert np.array_equal(ref.ternary(n, q[:nmod]), pkg.to_host(gpu.ternary(nmod)))
        assert np.array_equal(ref.centered_binomial(n, q[:nmod]), pkg.to_host(gpu.centered_binomial(nmod)))
        assert np.array_equal(ref.uniform(n, q[:nmod]), pkg.to_host(gpu.uniform(nmod)))
    assert ref.sample_uint64() == gpu.sample_uint64()          # counters advanced identically


def test_batched_samplers_reproduce_sequential_positions(O, pkg, dev):
    """troyn_sample_centered_binomial_strided / troyn_sample_uniform_multi (batched encryption): item i equals what the
    oracle's generator yields at the position a sequential encryption would use (seed block, then N/2 noise blocks)."""
    ctx, plan, q, t, n = _ctx(O, pkg, dev)
    count, nmod, per_ct = 19, 2, 1 + n // 2                    # 19 crosses the 16-generator launch chunk
    ref, gpu = O.Rng(0x1234, 0x99), pkg.Prng(plan, 0x1234, 0x99)
    for _ in range(3):
        assert ref.sample_uint64() == gpu.sample_uint64()
    base = gpu.counter
    want_seed, want_noise, want_c1 = [], [], []
    for i in range(count):
        sd = ref.sample_uint64()
        want_seed.append(sd)
        want_c1.append(O.Rng(sd, 0).uniform(n, q[:nmod]))
        want_noise.append(ref.centered_binomial(n, q[:nmod]))
    seeds = []
    for i in range(count):
        gpu.counter = base + i * per_ct
        seeds.append((gpu.sample_uint64(), 0))
    assert [s[0] for s in seeds] == want_seed
    gpu.counter = base + 1
    noise = pkg.to_host(gpu.centered_binomial_strided(nmod, count, per_ct))
    c1 = pkg.to_host(pkg.sample_uniform_multi(plan, nmod, seeds))
    assert np.array_equal(noise, np.stack(want_noise)) and np.array_equal(c1, np.stack(want_c1))


@pytest.mark.parametrize("n,bits,t", [(8192, [40, 40, 40], 1032193), (4096, [36, 36, 37], 1 << 21), (1024, [50, 50], 65537)])
def test_scale_up_and_decrypt_round(O, pkg, dev, n, bits, t):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    ctx = O.Context("bfv", n, q, t)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    for L in range(len(q), 0, -1):
        behz = pkg.Behz(plan, L, t)
        rt = ctx.rns_tool(L)
        assert behz.gamma == O.lib().orc_rns_tool_gamma(rt)
        # decrypt_scale_and_round on arbitrary residues (the function is defined for any input)
        phase = np.stack([ctx.random_ct(11 + i, 1, L)[0] for i in range(3)])
        got = pkg.to_host(behz.decrypt_scale_and_round(pkg.to_device(phase, dev)))
        for i in range(3):
            assert np.array_equal(got[i], ctx.decrypt_scale_and_round(L, phase[i])), (L, i)
        # scale_up followed by decrypt_scale_and_round returns the plaintext (no noise): semantic round trip
        m = O.fill_uniform(5 + L, t, 2 * n).reshape(2, n)
        m[0, :4] = [0, 1, t - 1, t // 2]
        up = behz.scale_up(pkg.to_device(m, dev))
        assert np.array_equal(pkg.to_host(behz.decrypt_scale_and_round(up)), m)
        # add / subtract forms
        base = np.stack([ctx.random_ct(21 + i, 1, L)[0] for i in range(2)])
        db = pkg.to_device(base, dev)
        plus = pkg.to_host(behz.scale_up(pkg.to_device(m, dev), src=db))
        minus = pkg.to_host(behz.scale_up(pkg.to_device(m, dev), src=db, subtract=True))
        up_h = pkg.to_host(up)
        for l in range(L):
            assert np.array_equal(plus[:, l], (base[:, l] + up_h[:, l]) % np.uint64(q[l]))
            assert np.array_equal(minus[:, l], (base[:, l] + np.uint64(q[l]) - up_h[:, l]) % np.uint64(q[l]))


def test_config1_keygen_encrypt_decrypt_on_gpu(O, pkg, dev):
    """The whole examples/99_quickstart.cu flow on the device with the reference's seed: the ciphertext and its
    square must hash to the digests the REFERENCE produced; decryption must return the slots."""
    ctx, plan, q, t, n = _ctx(O, pkg, dev)
    K, L = 3, 2
    # -- oracle side (expected values) --
    orng = O.Rng(G["seed"])
    sk_o = ctx.secret_key(orng)
    pk_o = ctx.public_key(orng, sk_o)
    plain = ctx.batch_encode(G["message"])
    # -- device side --
    rng = pkg.Prng(plan, G["seed"])
    sk = plan.ntt(rng.ternary(K)[None], 1, K)[0]                                   # KeyGenerator ctor
    assert np.array_equal(pkg.to_host(sk), sk_o)
    seed = 0
    while seed == 0:
        seed = rng.sample_uint64()
    c1 = pkg.Prng(plan, seed).uniform(K)                                            # rlwe::symmetric, NTT form
    e = plan.ntt(rng.centered_binomial(K)[None], 1, K)[0]
    c0 = plan.negate(plan.add(plan.dyadic_product(sk, c1, K), e, K), K)
    pk = torch.stack([c0, c1])
    assert np.array_equal(pkg.to_host(pk), pk_o)
    # encrypt_asymmetric at the key level, then divide_and_round_q_last, then + round(q/t * m)
    u = plan.ntt(rng.ternary(K)[None], 1, K)[0]
    ct_k = torch.stack([plan.dyadic_product(u, pk[j], K) for j in range(2)])
    plan.ntt(ct_k[None], 2, K, inverse=True)
    for j in range(2):
        ct_k[j] = plan.add(ct_k[j], rng.centered_binomial(K), K)
    ct = plan.divide_and_round_q_last(K, ct_k[None], 2)[0]
    behz = pkg.Behz(plan, L, t)
    dplain = pkg.to_device(plain[None], dev)
    ct[0] = behz.scale_up(dplain, src=ct[0][None].contiguous())[0]
    ct_h = pkg.to_host(ct)
    assert "%016x" % O.fnv_words(ct_h) == G["ciphertext_digest"]
    sq = behz.multiply(ct[None].contiguous(), 2, ct[None].contiguous(), 2)
    assert "%016x" % O.fnv_words(pkg.to_host(sq)[0]) == G["multiply_digest"]
    # relinearize with genuine keys generated on the device (KeyGenerator::create_relin_keys)
    sk2 = plan.dyadic_product(sk, sk, K)
    keys = []
    for i in range(L):
        seed = 0
        while seed == 0:
            seed = rng.sample_uint64()
        k1 = pkg.Prng(plan, seed).uniform(K)
        ek = plan.ntt(rng.centered_binomial(K)[None], 1, K)[0]
        k0 = plan.negate(plan.add(plan.dyadic_product(sk, k1, K), ek, K), K)
        factor = q[K - 1] % q[i]
        k0[i] = plan.add(k0[i][None].contiguous(), plan.multiply_scalar(sk2[i][None].contiguous(), factor, 1, mod_start=i), 1, mod_start=i)[0]
        keys.append(torch.stack([k0, k1]).contiguous())
    orng2 = O.Rng(G["seed"])
    sk_o2 = ctx.secret_key(orng2); ctx.public_key(orng2, sk_o2)
    orng2.ternary(n, q); orng2.centered_binomial(n, q); orng2.centered_binomial(n, q)      # the draws of encryption
    keys_o = ctx.relin_keys(orng2, sk_o2)
    for i in range(L):
        assert np.array_equal(pkg.to_host(keys[i]), keys_o[i]), i
    rel = plan.relinearize(L, sq, keys, is_ckks=False, is_ntt_form=False)
    # decrypt: c0 + c1 * s, then decrypt_scale_and_round, then decode on the oracle side
    def decrypt(c, pc):
        acc = None
        sp = sk[:L].contiguous()
        for i in range(1, pc):
            term = plan.ntt(c[i][None].clone(), 1, L)[0]
            term = plan.dyadic_product(term, sp, L)
            acc = term if acc is None else plan.add(acc, term, L)
            sp = plan.dyadic_product(sp, sk[:L].contiguous(), L)
        acc = plan.ntt(acc[None].contiguous(), 1, L, inverse=True)[0]
        acc = plan.add(acc, c[0].contiguous(), L)
        return pkg.to_host(behz.decrypt_scale_and_round(acc[None].contiguous()))[0]
    assert np.array_equal(decrypt(ct, 2), plain)
    want = [1, 4, 9, 16, 0, 0]
    assert [int(v) for v in ctx.batch_decode(decrypt(sq[0], 3))[:6]] == want
    assert [int(v) for v in ctx.batch_decode(decrypt(rel[0], 2))[:6]] == want
    assert np.array_equal(decrypt(rel[0], 2), ctx.decrypt_bfv(sk_o, pkg.to_host(rel)[0]))
