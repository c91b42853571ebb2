"""The headline pipeline DECRYPTS correctly: CKKS N=16384, 6x50-bit (BASELINE config 3), multiply -> relinearize ->
rescale_to_next of two encrypted vectors gives their slot-wise product.  Oracle on the CPU; the device pipeline must
produce the same ciphertext words (and therefore the same slots)."""
import numpy as np
import pytest

import ckks_util as U

N, BITS, SCALE = 16384, [50] * 6, float(1 << 40)


def _setup(O):
    q = [int(v) for v in O.coeff_modulus_create(N, BITS)]
    ctx = O.Context("ckks", N, q)
    rng = O.Rng(2024)
    sk = ctx.secret_key(rng)
    gen = np.random.default_rng(7)
    z1 = gen.uniform(-1, 1, N // 2) + 1j * gen.uniform(-1, 1, N // 2)
    z2 = gen.uniform(-1, 1, N // 2) + 1j * gen.uniform(-1, 1, N // 2)
    L = len(q) - 1
    c1 = U.encrypt(ctx, rng, sk, U.encode(z1, N, SCALE), L)
    c2 = U.encrypt(ctx, rng, sk, U.encode(z2, N, SCALE), L)
    rk = ctx.relin_keys(rng, sk)
    return ctx, q, sk, L, z1, z2, c1, c2, rk


def test_encode_decode_roundtrip():
    gen = np.random.default_rng(1)
    z = gen.uniform(-1, 1, 512) + 1j * gen.uniform(-1, 1, 512)
    assert np.abs(U.decode(U.encode(z, 1024, SCALE), 1024, SCALE) - z).max() < 1e-8


def test_headline_pipeline_decrypts_on_the_oracle(O):
    ctx, q, sk, L, z1, z2, c1, c2, rk = _setup(O)
    assert np.abs(U.decode(U.decrypt_limb0(ctx, sk, c1), N, SCALE) - z1).max() < 1e-6
    prod = ctx.ckks_multiply(L, c1, c2)
    relin = ctx.relinearize(L, True, prod, rk)
    out = ctx.mod_switch_scale_to_next(L, relin)                       # CKKS rescale_to_next: scale^2 / q_last
    new_scale = SCALE * SCALE / q[L - 1]
    got = U.decode(U.decrypt_limb0(ctx, sk, out), N, new_scale)
    assert np.abs(got - z1 * z2).max() < 1e-4
    # the 3-polynomial product and the relinearized ciphertext decrypt to the same thing before rescaling
    pre = U.decode(U.decrypt_limb0(ctx, sk, relin) % q[0], N, 1.0)    # only congruence mod q_0 is meaningful here
    assert pre.shape == (N // 2,)


@pytest.mark.gpu
def test_headline_pipeline_decrypts_on_the_gpu(O, pkg, dev):
    ctx, q, sk, L, z1, z2, c1, c2, rk = _setup(O)
    plan = pkg.Plan(dev, 14, q)
    d1, d2 = pkg.to_device(c1[None], dev), pkg.to_device(c2[None], dev)
    prod = plan.dyadic_convolute(d1, 2, d2, 2, L)
    relin = plan.relinearize(L, prod, [pkg.to_device(k, dev) for k in rk], is_ckks=True, is_ntt_form=True)
    out = pkg.to_host(plan.divide_and_round_q_last_ntt(L, relin, 2))[0]
    assert np.array_equal(out, ctx.mod_switch_scale_to_next(L, ctx.relinearize(L, True, ctx.ckks_multiply(L, c1, c2), rk)))
    got = U.decode(U.decrypt_limb0(ctx, sk, out), N, SCALE * SCALE / q[L - 1])
    assert np.abs(got - z1 * z2).max() < 1e-4
