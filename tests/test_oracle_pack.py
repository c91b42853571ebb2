"""CPU checks of the oracle's RLWE / LWE packing restatement (evaluator_lwes.cu): the reference holds no golden vector for
these, so the restatement is anchored on what the operations mean -- the packed ciphertext decrypts to the coefficients
the reference's callers read (MatmulHelper::pack_outputs / decrypt_outputs, app/matmul.cu:536-566,:572-619) -- and on the
algebra of the primitives.  Parity of the HIP path with this restatement: tests/test_gpu_pack.py."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def world(O):
    n, t = 64, 257
    q = [int(v) for v in O.coeff_modulus_create(n, [30, 30, 30])]
    ctx = O.Context("bfv", n, q, t)
    rng = O.Rng(7)
    sk = ctx.secret_key(rng)
    return dict(n=n, t=t, q=q, L=2, ctx=ctx, rng=rng, sk=sk, pk=ctx.public_key(rng, sk))


def test_shift_and_inv_degree_algebra(O, world):
    ctx, n, q, L = world["ctx"], world["n"], world["q"], world["L"]
    a = ctx.random_ct(1, 2, L)
    qv = np.array(q[:L], dtype=np.uint64).reshape(1, L, 1)
    assert np.array_equal(ctx.negacyclic_shift(L, ctx.negacyclic_shift(L, a, 5), 2 * n - 5), a)
    assert not np.any((a + ctx.negacyclic_shift(L, a, n)) % qv)          # X^N = -1
    assert np.array_equal(ctx.multiply_inv_degree(L, a, n), a)          # N^-1 * N = 1
    one = np.zeros((1, L, n), dtype=np.uint64)
    one[0, :, 0] = 1
    x3 = ctx.negacyclic_shift(L, one, n + 3)                             # X^(N+3) = -X^3
    assert all(int(x3[0, l, 3]) == q[l] - 1 for l in range(L)) and np.count_nonzero(x3) == L


def test_pack_rlwe_semantics(O, world):
    ctx, n, t, L, rng, sk, pk = (world[k] for k in ("ctx", "n", "t", "L", "rng", "sk", "pk"))
    I = 4
    rs = np.random.RandomState(1)
    msgs = [rs.randint(0, t, n).astype(np.uint64) for _ in range(3)]
    cts = [ctx.encrypt_asymmetric_bfv(rng, pk, m) for m in msgs]
    keys = {(n // I) * (1 << (layer + 1)) + 1: None for layer in range(2)}
    for g in keys:
        keys[g] = ctx.galois_key(rng, sk, g)
    out = ctx.pack_rlwe_ciphertexts(L, cts, keys, 2 * n - (I - 1), I, 1)
    want = np.zeros(n, dtype=np.uint64)
    for j, m in enumerate(msgs):
        want[j::I] = m[I - 1::I]
    assert np.array_equal(ctx.decrypt_bfv(sk, out), want)


def test_extract_assemble_pack_lwe_semantics(O, world):
    """Evaluator::pack_lwe_ciphertexts_new (evaluator_lwes.cu:200-230): LWE i lands on coefficient i * N / 2^l, every other
    coefficient is cleared by the field trace"""
    ctx, n, t, L, rng, sk, pk = (world[k] for k in ("ctx", "n", "t", "L", "rng", "sk", "pk"))
    rs = np.random.RandomState(2)
    m = rs.randint(0, t, n).astype(np.uint64)
    ct = ctx.encrypt_asymmetric_bfv(rng, pk, m)
    terms = [0, 5, n - 1]
    lwes = [ctx.extract_lwe(L, ct, term) for term in terms]
    # an assembled LWE decrypts to its term in the constant coefficient
    for (c0, c1), term in zip(lwes, terms):
        assert int(ctx.decrypt_bfv(sk, ctx.assemble_lwe(L, c0, c1))[0]) == int(m[term])
    count, l = len(lwes), 2
    keys = {}
    d = n
    while d > 1:
        keys[d + 1] = ctx.galois_key(rng, sk, d + 1)
        d >>= 1
    rl = [ctx.assemble_lwe(L, c0, c1) for c0, c1 in lwes]
    out = ctx.pack_rlwe_ciphertexts(L, rl, keys, 0, n, n >> l)
    dec = ctx.decrypt_bfv(sk, out)
    want = np.zeros(n, dtype=np.uint64)
    for i, term in enumerate(terms):
        want[i * (n >> l)] = m[term]
    assert np.array_equal(dec, want)
