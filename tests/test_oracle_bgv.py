"""CPU checks of the oracle's BGV restatement.  The reference's recorded outputs (tests/golden) are BFV
only, so BGV is anchored on what the operations mean: every step of a pipeline -- asymmetric encryption (noise times t,
mod_t_and_divide_q_last_ntt from the key level), addition, dyadic multiply, relinearization through the ski_util5 key-switch
tail, two modulus switches with their correction factors, a Galois automorphism -- decrypts (exact_convey_array +
correction-factor inverse) to the plaintext it must hold.  HIP parity with this restatement: tests/test_gpu_bgv.py."""
import numpy as np
import pytest


def negacyclic(a, b, t):
    n = len(a)
    full = np.convolve(np.array([int(x) for x in a], dtype=object), np.array([int(x) for x in b], dtype=object))
    r = [0] * n
    for i, v in enumerate(full):
        if i < n:
            r[i] += v
        else:
            r[i - n] -= v
    return np.array([int(x) % t for x in r], dtype=np.uint64)


@pytest.mark.parametrize("n,bits,t", [(1024, [40, 40, 40, 40], 65537), (256, [36, 36, 37], 257)])
def test_bgv_pipeline_semantics(O, n, bits, t):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    ctx = O.Context("bgv", n, q, t)
    rng = O.Rng(5)
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    rk = ctx.relin_keys(rng, sk)
    L = len(q) - 1
    rs = np.random.RandomState(0)
    m1, m2 = (rs.randint(0, t, n).astype(np.uint64) for _ in range(2))
    c1, c2 = ctx.encrypt_asymmetric_bgv(rng, pk, m1), ctx.encrypt_asymmetric_bgv(rng, pk, m2)
    assert np.array_equal(ctx.decrypt_bgv(sk, c1), m1)
    qv = np.array(q[:L], dtype=np.uint64).reshape(1, L, 1)
    assert np.array_equal(ctx.decrypt_bgv(sk, (c1 + c2) % qv), (m1 + m2) % np.uint64(t))
    want = negacyclic(m1, m2, t)
    prod = ctx.ckks_multiply(L, c1, c2)                       # bgv_multiply is the same dyadic product (evaluator.cu:150-173)
    assert np.array_equal(ctx.decrypt_bgv(sk, prod), want)
    rel = ctx.relinearize(L, True, prod, rk)
    assert np.array_equal(ctx.decrypt_bgv(sk, rel), want)
    cf, cur = 1, rel
    for level in range(L, 1, -1):
        cur = ctx.mod_switch_scale_to_next(level, cur)
        cf = cf * ctx.bgv_inv_q_last_mod_t(level) % t
        assert cur.shape == (2, level - 1, n)
        assert np.array_equal(ctx.decrypt_bgv(sk, cur, cf), want), level
    g = 3
    rot = ctx.apply_galois_ct(L, True, g, c1, ctx.galois_key(rng, sk, g))
    exp = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        r = (i * g) % (2 * n)
        exp[r % n] = m1[i] if r < n else (t - int(m1[i])) % t
    assert np.array_equal(ctx.decrypt_bgv(sk, rot), exp)


def test_decrypt_mod_t_is_the_centred_remainder(O):
    """exact_convey_array: for a phase x = m + t*e with |x| < q/2 the result is x mod t (centred lift)"""
    n, t = 64, 257
    q = [int(v) for v in O.coeff_modulus_create(n, [30, 30, 30])]
    ctx = O.Context("bgv", n, q, t)
    L = 2
    Q = q[0] * q[1]
    rs = np.random.RandomState(4)
    xs = [int(v) for v in rs.randint(-2 ** 40, 2 ** 40, n)]
    xs[0], xs[1], xs[2], xs[3] = 0, -1, Q // 4, -(Q // 4)            # (at +-Q/2 the double-precision vote v is a coin toss)
    phase = np.array([[x % q[l] for x in xs] for l in range(L)], dtype=np.uint64)
    assert [int(v) for v in ctx.decrypt_mod_t(L, phase)] == [x % t for x in xs]
