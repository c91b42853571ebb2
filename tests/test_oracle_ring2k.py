"""CPU checks of the oracle's ring-2^k encoder restatement (src/app/bfv_ring2k.cu).  The reference holds no recorded vectors for it, so
the restatement is anchored on what the encoder must do: scale_down(scale_up(m) + noise) = m for every width and k, and centralize is
the centred lift.  HIP parity with this restatement: tests/test_gpu_ring2k.py."""
import random

import numpy as np
import pytest


@pytest.mark.parametrize("elem_bits,k,L", [(64, 64, 2), (64, 50, 2), (32, 32, 1), (32, 20, 1), (128, 128, 3), (128, 100, 3), (128, 65, 2)])
def test_ring2k_round_trip(O, elem_bits, k, L):
    n = 64
    q = [int(v) for v in O.coeff_modulus_create(n, [60, 60, 60, 60])]
    r = O.Ring2k(n, q[:L], k, elem_bits)
    rnd = random.Random(k)
    m = [rnd.getrandbits(k) for _ in range(n)]
    m[:4] = [0, 1, (1 << k) - 1, 1 << (k - 1)]
    up = r.scale_up(m)
    noise = [rnd.randint(-1000, 1000) for _ in range(n)]
    noisy = np.array([[(int(up[l, c]) + noise[c]) % ql for c in range(n)] for l, ql in enumerate(r.q)], dtype=np.uint64)
    assert r.scale_down(noisy) == m
    ce = r.centralize(m)
    for c in range(n):
        lift = m[c] if m[c] <= r.t_half else m[c] - (1 << k)
        assert all(int(ce[l, c]) == lift % ql for l, ql in enumerate(r.q))


@pytest.mark.parametrize("elem_bits,k,L", [(64, 64, 2), (64, 50, 2), (32, 32, 1), (32, 20, 1), (128, 128, 3), (128, 100, 3), (128, 65, 2), (64, 33, 4)])
def test_ring2k_decentralize_round_trip(O, elem_bits, k, L):
    """the reference's own test of decentralize (test/app/bfv_ring2k.cu:235-266): decentralize(centralize(m)) == m; here also for the product of two centred lifts
    (what a ciphertext x plaintext product holds before decryption: the true integer, far below Q/2 in magnitude), for a correction factor, and at the
    half-way points of the quotient estimate"""
    n = 64
    q = [int(v) for v in O.coeff_modulus_create(n, [60, 60, 60, 60])]
    r = O.Ring2k(n, q[:L], k, elem_bits)
    rnd = random.Random(k + 1000 * L)
    m = [rnd.getrandbits(k) for _ in range(n)]
    m[:6] = [0, 1, (1 << k) - 1, 1 << (k - 1), (1 << (k - 1)) + 1, (1 << (k - 1)) - 1]
    assert r.decentralize(r.centralize(m)) == m
    cf = rnd.getrandbits(k) | 1
    assert r.decentralize(r.centralize(m), cf) == [x * pow(cf, -1, 1 << k) & r.mask for x in m]
    with pytest.raises(ValueError):
        r.decentralize(r.centralize(m), 2)
    if r.Q > 1 << (2 * k + 8):                      # integers below Q / 2 in magnitude come back reduced mod 2^k, whatever their sign
        ints = [rnd.randint(-(1 << (2 * k)), 1 << (2 * k)) for _ in range(n)]
        res = np.array([[v % ql for v in ints] for ql in r.q], dtype=np.uint64)
        assert r.decentralize(res) == [v & r.mask for v in ints]
