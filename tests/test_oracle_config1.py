"""Pin the oracle against real reference OUTPUT: BASELINE config 1 (examples/99_quickstart.cu parameters) run through
the reference's CPU branch with seed 0x123 produced the two digests in tests/golden/config1_digests.json (SURVEY.md
Appendix C).  Reproducing them needs every hot-path piece bit-exact: NTT/INTT over q and t, dyadic products, the
modulus switch used inside encryption (divide_and_round_q_last) and the whole BEHZ multiply.  Runs without a GPU."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "config1_digests.json")))


def config1_ciphertext(O):
    n = G["poly_modulus_degree"]
    q = O.coeff_modulus_create(n, G["coeff_modulus_bits"])
    assert [int(v) for v in q] == G["coeff_modulus"]
    t = int(O.get_primes(2 * n, 20, 1)[0])           # PlainModulus::batching(8192, 20)
    assert t == G["plain_modulus"]
    ctx = O.Context("bfv", n, q, t)
    rng = O.Rng(G["seed"])
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    plain = ctx.batch_encode(G["message"])
    ct = ctx.encrypt_asymmetric_bfv(rng, pk, plain)
    return ctx, sk, ct


def test_aes128_known_answer(O):
    k = G["aes128_fips197"]
    got = O.aes128_encrypt_block(bytes.fromhex(k["plaintext"]), bytes.fromhex(k["key"]))
    assert got.hex() == k["ciphertext"]


def test_samplers_shape(O):
    q = G["coeff_modulus"]
    r = O.Rng(7)
    tern = r.ternary(64, q)
    for i, qi in enumerate(q):
        assert set(int(v) for v in tern[i]) <= {0, 1, qi - 1}
    assert np.array_equal(tern[0] == 1, tern[1] == 1)            # same small value in every limb
    cbd = r.centered_binomial(64, q)
    small = np.where(cbd[0] > q[0] // 2, cbd[0].astype(np.int64) - q[0], cbd[0].astype(np.int64))
    assert np.abs(small).max() <= 21
    uni = r.uniform(64, q)
    assert all((uni[i] < q[i]).all() for i in range(len(q)))
    # determinism: same seed -> same stream
    assert O.Rng(7).sample_uint64() == O.Rng(7).sample_uint64()


def test_config1_ciphertext_digest(O):
    _, _, ct = config1_ciphertext(O)
    assert "%016x" % O.fnv_words(ct) == G["ciphertext_digest"]


def test_config1_multiply_digest(O):
    ctx, _, ct = config1_ciphertext(O)
    prod = ctx.bfv_multiply(2, ct, ct)
    assert prod.shape == (3, 2, G["poly_modulus_degree"])
    assert "%016x" % O.fnv_words(prod) == G["multiply_digest"]


def test_config1_end_to_end_semantics(O):
    """examples/99_quickstart.cu flow with genuine keys: the oracle's BEHZ multiply, key switching (relinearize with
    real relinearization keys) and modulus switch must DECRYPT to the right slots -- a first-principles check of those
    paths that does not depend on any restated constant."""
    ctx, sk, ct = config1_ciphertext(O)
    plain = ctx.batch_encode(G["message"])
    dec = ctx.decrypt_bfv(sk, ct)
    assert np.array_equal(dec, plain)
    assert [int(v) for v in ctx.batch_decode(dec)[:6]] == [1, 2, 3, 4, 0, 0]
    rng = O.Rng(99)
    prod = ctx.bfv_multiply(2, ct, ct)
    want = [1, 4, 9, 16, 0, 0]
    d3 = ctx.decrypt_bfv(sk, prod)
    assert [int(v) for v in ctx.batch_decode(d3)[:6]] == want
    rel = ctx.relinearize(2, False, prod, ctx.relin_keys(rng, sk))
    assert np.array_equal(ctx.decrypt_bfv(sk, rel), d3)
    low = ctx.mod_switch_scale_to_next(2, rel)
    assert np.array_equal(ctx.decrypt_bfv(sk, low), d3)
    # add: (m + m) decrypts to 2m
    two = np.stack([(ct[p].astype(object) * 2 % np.array(ctx.q[:2], dtype=object)[:, None]).astype(np.uint64) for p in range(2)])
    assert [int(v) for v in ctx.batch_decode(ctx.decrypt_bfv(sk, two))[:4]] == [2, 4, 6, 8]


def test_rotations_and_plain_multiply_semantics(O):
    """Galois automorphisms (with genuine Galois keys) and ciphertext x plaintext products decrypt to the expected
    slots: first-principles validation of the oracle's apply_galois / key switch / centralize paths."""
    n = G["poly_modulus_degree"]
    ctx, sk, ct = config1_ciphertext(O)
    rng = O.Rng(4242)
    w = ctx.batch_encode([3, 5, 7, 11])
    assert [int(v) for v in ctx.batch_decode(ctx.decrypt_bfv(sk, ctx.multiply_plain_normal(2, ct, w)))[:5]] == [3, 10, 21, 44, 0]
    pk = ctx.public_key(rng, sk)
    ramp = ctx.encrypt_asymmetric_bfv(rng, pk, ctx.batch_encode(list(range(1, n + 1))))
    row = n // 2
    for step in (1, -2, 0):
        g = ctx.galois_element_from_step(step)
        rot = ctx.apply_galois_ct(2, False, g, ramp, ctx.galois_key(rng, sk, g))
        dec = [int(v) for v in ctx.batch_decode(ctx.decrypt_bfv(sk, rot))]
        if step == 0:
            assert dec[:3] == [row + 1, row + 2, row + 3] and dec[row:row + 3] == [1, 2, 3]
        else:
            assert dec[:4] == [((i + step) % row) + 1 for i in range(4)]
    # NTT-form permutation == coefficient-form permutation conjugated by the transform
    a = ctx.random_ct(9, 1, 2)
    g = ctx.galois_element_from_step(5)
    via_ntt = ctx.from_ntt(ctx.apply_galois(2, True, g, ctx.to_ntt(a, 1, 2)), 1, 2)
    assert np.array_equal(via_ntt, ctx.apply_galois(2, False, g, a))
