"""GPU key switching against the exact big-integer specification (tests/ks_spec.py) -- never against the oracle's orc_switch_key.

Small rings (N = 32 / 64): everything by definition.  N = 8192 / 16384 (the ksmac2_kernel sizes, BASELINE configs 2 / 3): the
specification's negacyclic products and exact rounded division on Python integers; only the NTT <-> coefficient form conversions
of the operands use the oracle's transform, which tests/test_keyswitch_spec.py pins to the by-definition transform."""
import numpy as np
import pytest

from ks_spec import switch_key_spec
from test_keyswitch_spec import _ntt_polys, make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,bits,L,order", [(32, [50, 50, 50, 50], 3, None), (64, [60, 40, 40, 60], 3, None), (32, [50, 50, 50], 2, "reversed")])
@pytest.mark.parametrize("scheme,is_ntt", [("ckks", True), ("bfv", False)])
def test_small_rings_by_definition(O, pkg, dev, n, bits, L, order, scheme, is_ntt):
    q = O.coeff_modulus_create(n, bits)
    if order == "reversed":
        q = sorted(q, reverse=True)
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    digits, keys, keys_ntt, _ = make_case(O, n, q, L, 23, True)
    target_c = np.array(digits, dtype=np.uint64)
    target = _ntt_polys(target_c, q[:L]) if is_ntt else target_c
    rng = np.random.default_rng(5)
    dest_c = [[[int(v) for v in rng.integers(0, q[l], size=n, dtype=np.uint64)] for l in range(L)] for c in range(2)]
    dest = np.array(dest_c, dtype=np.uint64)
    dest_in = np.stack([_ntt_polys(dest[c], q[:L]) for c in range(2)]) if is_ntt else dest
    dkeys = [pkg.to_device(k, dev) for k in keys_ntt]
    for assign in (0, 1, 2):
        exp = np.array(switch_key_spec(q, L, digits, keys, dest_c, assign), dtype=np.uint64)
        if is_ntt:
            exp = np.stack([_ntt_polys(exp[c], q[:L]) for c in range(2)])
        dd = pkg.to_device(dest_in[None].copy(), dev)
        plan.switch_key(L, pkg.to_device(target[None].copy(), dev), dkeys, dest=dd, assign=assign, is_ckks=(scheme == "ckks"), is_ntt_form=is_ntt)
        assert np.array_equal(pkg.to_host(dd)[0], exp), assign


@pytest.mark.parametrize("n,bits,L,is_ntt", [(8192, [40, 40, 40], 2, False),          # BASELINE config 2 (BFV, coefficient form)
                                             (8192, [50, 50, 50, 50], 3, True),
                                             (16384, [50] * 6, 5, True),                # BASELINE config 3: ksmac2_kernel<14> + fused tail
                                             (8192, [60, 40, 40, 60], 3, True)])        # the reference's default chain (integer policy)
def test_ksmac_sizes_against_spec(O, pkg, dev, n, bits, L, is_ntt):
    q = O.coeff_modulus_create(n, bits)
    K = len(q)
    qs, h = q[-1], q[-1] // 2
    ctx = O.Context("ckks", n, q)                    # used for NTT <-> coefficient conversions only
    plan = pkg.Plan(dev, n.bit_length() - 1, q)
    rng = np.random.default_rng(77)
    digits = np.stack([rng.integers(0, q[j], size=n, dtype=np.uint64) for j in range(L)])
    keys_c = [np.stack([np.stack([rng.integers(0, q[k], size=n, dtype=np.uint64) for k in range(K)]) for c in range(2)]) for j in range(L)]
    # digit 0 = the constant polynomial 1 and the special-prime rows of key 0 solved for: the special-prime component of the inner
    # product sits on the rounding boundary at the first coefficients (tests/test_keyswitch_spec.py::make_case)
    from ks_spec import negacyclic
    digits[0] = 0
    digits[0, 0] = 1
    edge = [h - 1, h, h + 1, 0, 1, qs - 1, qs - h, qs - h - 1, qs - h + 1, qs - 2]
    for c in range(2):
        w = [int(v) for v in rng.integers(0, qs, size=n, dtype=np.uint64)]
        w[:len(edge)] = edge if c == 0 else edge[::-1]
        rest = [0] * n
        for j in range(1, L):
            rest = [(x + y) % qs for x, y in zip(rest, negacyclic(digits[j], keys_c[j][c][K - 1], qs))]
        keys_c[0][c][K - 1] = np.array([(x - y) % qs for x, y in zip(w, rest)], dtype=np.uint64)
    key_tables = list(range(K))

    def to_ntt_rows(x, rows):        # x [len(rows)][N] under moduli q[rows]
        out = np.empty_like(x)
        for i, r in enumerate(rows):
            c1 = O.Context("ckks", n, [q[r], q[(r + 1) % K]])
            out[i] = c1.to_ntt(x[i][None, None], 1, 1)[0, 0]
        return out

    keys_ntt = [np.stack([to_ntt_rows(keys_c[j][c], key_tables) for c in range(2)]) for j in range(L)]
    target = to_ntt_rows(digits, list(range(L))) if is_ntt else digits
    dest_c = np.stack([np.stack([rng.integers(0, q[l], size=n, dtype=np.uint64) for l in range(L)]) for c in range(2)])
    dest_in = np.stack([to_ntt_rows(dest_c[c], list(range(L))) for c in range(2)]) if is_ntt else dest_c
    dkeys = [pkg.to_device(k, dev) for k in keys_ntt]
    dig_l = [[int(v) for v in row] for row in digits]
    keys_l = [[[[int(v) for v in keys_c[j][c][k]] for k in range(K)] for c in range(2)] for j in range(L)]
    dest_l = [[[int(v) for v in dest_c[c][l]] for l in range(L)] for c in range(2)]
    res = np.array(switch_key_spec(q, L, dig_l, keys_l, dest_l, 1), dtype=np.uint64)        # Overwrite = the bare result
    qa = np.array(q[:L], dtype=np.uint64)[None, :, None]
    for assign in (1, 0):
        exp = res if assign == 1 else (dest_c + res) % qa          # both < 2^61: no wrap
        if is_ntt:
            exp = np.stack([to_ntt_rows(exp[c], list(range(L))) for c in range(2)])
        batch = 8            # the XCD-grouped workgroup order of ksmac2 needs a multiple of 8; every item is the same case
        dd = pkg.to_device(np.repeat(dest_in[None], batch, axis=0), dev)
        plan.switch_key(L, pkg.to_device(np.repeat(target[None], batch, axis=0), dev), dkeys, dest=dd, assign=assign, is_ckks=is_ntt, is_ntt_form=is_ntt)
        got = pkg.to_host(dd)
        for i in (0, 3, 7):
            assert np.array_equal(got[i], exp), (assign, i)
    del ctx
