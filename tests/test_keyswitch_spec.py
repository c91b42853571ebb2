"""Key switching pinned INDEPENDENTLY of the restated code: the oracle's orc_switch_key (oracle/troy_oracle.c, the restatement
of /root/reference/src/evaluator_keyswitching_core.cu:757-1052) against the exact big-integer specification of tests/ks_spec.py,
with inputs that put the special-prime residue on the rounding boundary (floor(q_K/2) +- 1, 0, q_K - 1; :583-597) and chains in
which q_K is larger / smaller than the data primes (the `qk > qi` branch of ski_util6, :589-593).  CPU only."""
import numpy as np
import pytest

from ks_spec import minimal_primitive_root, negacyclic, ntt_by_definition, switch_key_spec


def _ntt_polys(x, q):
    return np.array([ntt_by_definition([int(v) for v in row], qq) for row, qq in zip(x, q)], dtype=np.uint64)


def test_spec_building_blocks(O):
    """the spec's own pieces: psi equals the reference's known answers (test/utils/ntt.cu:38,47-49 via tests/golden/ref_kats.json are
    checked in test_oracle_kats.py; here: the oracle's table root), the by-definition transform equals the oracle's butterflies, and
    the Kronecker product equals the schoolbook sum"""
    rng = np.random.default_rng(1)
    for n, bits in ((32, [40, 40, 40]), (64, [50, 50]), (32, [60, 30])):
        q = O.coeff_modulus_create(n, bits)
        ctx = O.Context("ckks", n, q)
        for l, ql in enumerate(q):
            assert minimal_primitive_root(2 * n, ql) == O.NTTTables(n.bit_length() - 1, ql).root
        x = np.stack([rng.integers(0, ql, size=n, dtype=np.uint64) for ql in q])[None]
        assert np.array_equal(ctx.to_ntt(x, 1, len(q))[0], _ntt_polys(x[0], q))
    a = [int(v) for v in rng.integers(0, 1 << 50, size=32)]
    b = [int(v) for v in rng.integers(0, 1 << 50, size=32)]
    m = (1 << 50) - 27
    ref = [0] * 32
    for i in range(32):
        for j in range(32):
            k = i + j
            if k >= 32:
                ref[k - 32] -= a[i] * b[j]
            else:
                ref[k] += a[i] * b[j]
    assert negacyclic(a, b, m) == [v % m for v in ref]


def make_case(O, n, q, L, seed, boundary):
    """digits / keys in coefficient form (python ints) + their NTT forms; boundary: the
    special-prime component of the inner product is forced onto the rounding boundary"""
    K = len(q)
    qs = q[K - 1]
    h = qs // 2
    rng = np.random.default_rng(seed)
    digits = [[int(v) for v in rng.integers(0, q[j], size=n, dtype=np.uint64)] for j in range(L)]
    keys = [[[[int(v) for v in rng.integers(0, q[k], size=n, dtype=np.uint64)] for k in range(K)] for c in range(2)] for j in range(L)]
    wanted = None
    if boundary:
        # digit j1 is the constant polynomial 1, and the special-prime row of ITS key is solved for, so that the special-prime
        # component of the inner product is exactly `wanted` (boundary values first, uniform elsewhere) with every other operand random
        j1 = seed % L
        digits[j1] = [1] + [0] * (n - 1)
        wanted = []
        for c in range(2):
            w = [int(v) for v in rng.integers(0, qs, size=n, dtype=np.uint64)]
            edge = [h - 1, h, h + 1, 0, 1, qs - 1, qs - h, qs - h - 1, qs - h + 1, qs - 2]
            w[:len(edge)] = edge if c == 0 else edge[::-1]
            rest = [0] * n
            for j in range(L):
                if j != j1:
                    rest = [(x + y) % qs for x, y in zip(rest, negacyclic(digits[j], keys[j][c][K - 1], qs))]
            keys[j1][c][K - 1] = [(x - y) % qs for x, y in zip(w, rest)]
            wanted.append(w)
    keys_ntt = [np.stack([_ntt_polys(keys[j][c], q) for c in range(2)]) for j in range(L)]
    return digits, keys, keys_ntt, wanted


CHAINS = [
    ("ascending 50-bit (cfg3's order: q_K above every q_j)", 32, None, [50, 50, 50, 50], 3),
    ("special prime BELOW the data primes", 32, "reversed", [50, 50, 50], 2),
    ("reference default chain {60,40,40,60}", 64, None, [60, 40, 40, 60], 3),
    ("special prime 30 bits below 60-bit data primes", 32, "reversed", [30, 60, 60], 2),
    ("lower level of a longer chain", 64, None, [40, 40, 40, 40, 40], 2),
]


@pytest.mark.parametrize("desc,n,order,bits,L", CHAINS, ids=[c[0] for c in CHAINS])
@pytest.mark.parametrize("scheme", ["ckks", "bfv"])
@pytest.mark.parametrize("is_ntt", [True, False])
@pytest.mark.parametrize("boundary", [False, True])
def test_switch_key_matches_exact_spec(O, desc, n, order, bits, L, scheme, is_ntt, boundary):
    q = O.coeff_modulus_create(n, bits)
    if order == "reversed":
        q = sorted(q, reverse=True)              # smallest last = special
    ctx = O.Context(scheme, n, q, 0 if scheme == "ckks" else 65537)
    digits, keys, keys_ntt, wanted = make_case(O, n, q, L, 17 + L, boundary)
    target_c = np.array(digits, dtype=np.uint64)
    target = _ntt_polys(target_c, q[:L]) if is_ntt else target_c
    rng = np.random.default_rng(99)
    dest_c = [[[int(v) for v in rng.integers(0, q[l], size=n, dtype=np.uint64)] for l in range(L)] for c in range(2)]
    dest = np.array(dest_c, dtype=np.uint64)
    dest_in = np.stack([_ntt_polys(dest[c], q[:L]) for c in range(2)]) if is_ntt else dest
    if boundary:
        qs, K = q[-1], len(q)
        for c in range(2):           # the case really sits on the boundary
            sc = [0] * n
            for jj in range(L):
                sc = [(x + y) % qs for x, y in zip(sc, negacyclic(digits[jj], keys[jj][c][K - 1], qs))]
            assert sc == wanted[c] and {qs // 2 - 1, qs // 2, qs // 2 + 1, 0, qs - 1} <= set(sc)
    for assign in (0, 1, 2):
        exp_c = switch_key_spec(q, L, digits, keys, dest_c, assign)
        exp = np.array(exp_c, dtype=np.uint64)
        if is_ntt:
            exp = np.stack([_ntt_polys(exp[c], q[:L]) for c in range(2)])
        got = ctx.switch_key(L, is_ntt, target, keys_ntt, assign=assign, dest=dest_in)
        assert np.array_equal(got, exp), "orc_switch_key differs from the exact specification (assign %d)" % assign
