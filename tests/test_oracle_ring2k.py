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
