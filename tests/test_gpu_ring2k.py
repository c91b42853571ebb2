"""GPU parity of the ring-2^k polynomial encoder (src/app/bfv_ring2k.cu) with the oracle's restatement on Python integers, for the three
element widths and plaintext moduli up to 2^128, plus an encrypt -> multiply by a plaintext -> decrypt -> scale_down run whose result
is the negacyclic product in Z_{2^k} (examples/13_ring2k.cu).  Oracle pinning for this row is semantic (see tests/test_oracle_ring2k.py)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,bits,L,elem_bits,k", [(64, [60, 60, 60, 60], 2, 64, 64), (64, [60, 60, 60, 60], 2, 64, 40), (1024, [50, 50, 50], 2, 32, 32),
                                                  (1024, [50, 50, 50], 1, 32, 17), (8192, [60, 60, 60, 60], 3, 128, 128), (4096, [60, 60, 60, 60], 3, 128, 65),
                                                  (16384, [60, 60, 60, 60], 3, 64, 64)])
def test_ring2k_encoder_matches_oracle(O, pkg, dev, n, bits, L, elem_bits, k):
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    ref = O.Ring2k(n, q[:L], k, elem_bits)
    gpu = pkg.Ring2k(plan, L, k, elem_bits)
    assert gpu.gamma == ref.gamma
    rnd = random.Random(k * 1000 + n)
    count = min(n, 200) - 3                                                 # a partial polynomial: the tail must be zero
    m = [rnd.getrandbits(k) for _ in range(count)]
    m[:6] = [0, 1, (1 << k) - 1, 1 << (k - 1), (1 << (k - 1)) + 1, (1 << (k - 1)) - 1]
    up, ce = pkg.to_host(gpu.scale_up(m)), pkg.to_host(gpu.centralize(m))
    assert np.array_equal(up, ref.scale_up(m)) and np.array_equal(ce, ref.centralize(m))
    assert not up[:, count:].any() and not ce[:, count:].any()
    # scale_down on scaled-up values plus noise, and on arbitrary residues
    noisy = ref.scale_up(m).astype(object)
    for c in range(count):
        e = rnd.randint(-5000, 5000)
        for l in range(L):
            noisy[l, c] = (int(noisy[l, c]) + e) % q[l]
    noisy = noisy.astype(np.uint64)
    got = gpu.scale_down(pkg.to_device(noisy, dev))
    assert got == ref.scale_down(noisy) and got[:count] == m
    arbitrary = np.stack([O.fill_uniform(77 + l, q[l], n) for l in range(L)])
    if n <= 1024:
        assert gpu.scale_down(pkg.to_device(arbitrary, dev)) == ref.scale_down(arbitrary)
    with pytest.raises(Exception):
        pkg.Ring2k(plan, L, elem_bits // 2, elem_bits)                      # k must exceed half the element width


def test_ring2k_product_decrypts(O, pkg, dev):
    """scale_up -> encrypt (oracle keys) -> ct x centralized plaintext on the GPU -> phase -> scale_down = negacyclic product mod 2^64"""
    n, k = 4096, 64
    q = [int(v) for v in O.coeff_modulus_create(n, [60, 60, 60, 60])]
    ctx = O.Context("bfv", n, q, 1 << 20)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    L = 3
    enc = pkg.Ring2k(plan, L, k, 64)
    rnd = random.Random(9)
    a = [rnd.getrandbits(k) for _ in range(40)]
    b = [rnd.getrandbits(k) for _ in range(25)]
    rng = O.Rng(4)
    sk = ctx.secret_key(rng)
    pk = ctx.public_key(rng, sk)
    zero = ctx.encrypt_asymmetric_bfv(rng, pk, np.zeros(1, dtype=np.uint64))           # an encryption of 0: (c0, c1) with c0 + c1 s = noise
    ct = pkg.to_device(zero, dev).clone()
    ct[0] = plan.add(ct[0].contiguous(), enc.scale_up(a), L)
    pt = plan.ntt(enc.centralize(b).view(1, L, n), 1, L)
    prod = plan.ntt(plan.dyadic_broadcast_product(plan.ntt(ct.view(1, 2, L, n), 2, L), 2, pt, L), 2, L, inverse=True).view(2, L, n)
    dsk = pkg.to_device(sk, dev)
    c1s = plan.ntt(plan.dyadic_product(plan.ntt(prod[1].contiguous().view(1, L, n), 1, L).view(L, n), dsk[:L].contiguous(), L).view(1, L, n), 1, L, inverse=True)
    phase = plan.add(prod[0].contiguous(), c1s.view(L, n), L)
    got = enc.scale_down(phase)
    want = [0] * n
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            want[i + j] = (want[i + j] + x * y) & ((1 << k) - 1)
    assert got == want


@pytest.mark.parametrize("n,bits,L,elem_bits,k", [(64, [60, 60, 60, 60], 2, 64, 64), (64, [60, 60, 60, 60], 3, 64, 40), (1024, [50, 50, 50], 2, 32, 32),
                                                  (1024, [50, 50, 50], 1, 32, 17), (8192, [60, 60, 60, 60], 3, 128, 128), (4096, [60, 60, 60, 60], 3, 128, 65),
                                                  (16384, [60, 60, 60, 60], 4, 64, 64), (32768, [60, 40, 40, 60], 3, 64, 33), (65536, [60, 60, 60], 2, 128, 90)])
def test_ring2k_decentralize_matches_oracle(O, pkg, dev, n, bits, L, elem_bits, k):
    """decentralize (bfv_ring2k.cu:752-911): the double-precision quotient estimate is summed in the reference's order, so the HIP kernel equals the oracle
    bit for bit on arbitrary residues (where the estimate sits anywhere in [0, L)) as well as on centred lifts; the reference's round trip holds"""
    q = [int(v) for v in O.coeff_modulus_create(n, bits)]
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    ref = O.Ring2k(n, q[:L], k, elem_bits)
    gpu = pkg.Ring2k(plan, L, k, elem_bits)
    rnd = random.Random(k * 77 + n)
    m = [rnd.getrandbits(k) for _ in range(n)]
    m[:6] = [0, 1, (1 << k) - 1, 1 << (k - 1), (1 << (k - 1)) + 1, (1 << (k - 1)) - 1]
    ce = gpu.centralize(m)
    assert gpu.decentralize(ce) == m
    cf = rnd.getrandbits(k) | 1
    assert gpu.decentralize(ce, cf) == [x * pow(cf, -1, 1 << k) & ((1 << k) - 1) for x in m]
    rows = min(n, 2048)                                  # the oracle is a Python loop: compare a window of arbitrary residues
    arbitrary = np.stack([O.fill_uniform(91 + l, q[l], n) for l in range(L)])
    small = O.Ring2k(rows, q[:L], k, elem_bits)
    small.gamma = ref.gamma
    assert gpu.decentralize(pkg.to_device(arbitrary, dev))[:rows] == small.decentralize(arbitrary[:, :rows])
    assert gpu.decentralize(pkg.to_device(arbitrary, dev), cf)[:rows] == small.decentralize(arbitrary[:, :rows], cf)
    with pytest.raises(Exception):
        gpu.decentralize(ce, 2)                          # inverse_ring2k: the correction factor must be odd
