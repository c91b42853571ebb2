"""GPU parity: negacyclic NTT / INTT through the C-ABI vs the oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAT_Q = 0xffffffffffc0001   # reference test/utils/ntt.cu:35


def _rand_polys(O, seed, shape, moduli_for_last2):
    """uniform residues; shape (..., ncomp, n); component j reduced below moduli_for_last2[j]"""
    out = np.empty(shape, dtype=np.uint64)
    flat = out.reshape(-1, shape[-2], shape[-1])
    for i in range(flat.shape[0]):
        for j in range(shape[-2]):
            flat[i, j] = O.fill_uniform(seed * 131 + i * 17 + j, moduli_for_last2[j], shape[-1])
    return out


def test_tables_match_oracle_and_reference_kats(O, pkg, dev):
    # reference test/utils/ntt.cu:33-51 (PrimitiveRoots) through the product's own table builder
    plan = pkg.Plan(dev, 2, [KAT_Q])
    rp = plan.root_powers(0)
    assert [int(x) for x in rp[:, 0]] == [1, 288794978602139552, 178930308976060547, 748001537669050592]
    # full table equality with the oracle for a production-size modulus chain
    q = O.coeff_modulus_create(8192, [40, 40, 40])
    plan = pkg.Plan(dev, 13, q)
    for i, qi in enumerate(q):
        t = O.NTTTables(13, qi)
        assert plan.root(i) == t.root
        fwd, inv = plan.root_powers(i), plan.root_powers(i, inverse=True)
        for idx in list(range(0, 8192, 257)) + [1, 2, 3, 8191]:
            assert int(fwd[idx, 0]) == t.root_power(idx) and int(fwd[idx, 1]) == t.root_power(idx, quotient=True)
            assert int(inv[idx, 0]) == t.root_power(idx, inverse=True) and int(inv[idx, 1]) == t.root_power(idx, True, True)


def test_reference_kat_tiny_transforms(O, pkg, dev):
    # reference test/utils/ntt.cu:53-75: N=2, poly (1,1) -> (288794978602139553, 864126526004445282)
    plan = pkg.Plan(dev, 1, [KAT_Q])
    for inp, exp in (([0, 0], [0, 0]), ([1, 0], [1, 1]), ([1, 1], [288794978602139553, 864126526004445282])):
        x = pkg.to_device(np.array(inp, dtype=np.uint64), dev)
        plan.ntt(x, 1, 1)
        assert [int(v) for v in pkg.to_host(x)] == exp
    # :112-138 INTT(NTT(x)) == x for N = 32, x = 0..31
    plan = pkg.Plan(dev, 5, [KAT_Q])
    x0 = np.arange(32, dtype=np.uint64)
    x = pkg.to_device(x0, dev)
    plan.ntt(x, 1, 1)
    plan.ntt(x, 1, 1, inverse=True)
    assert np.array_equal(pkg.to_host(x), x0)


@pytest.mark.parametrize("log_n,bits", [(5, [30, 30, 30]), (9, [40, 40]), (10, [40, 50, 60]), (11, [50, 30]),
                                        (12, [60, 60]), (13, [40, 40, 40]), (14, [50] * 6), (15, [50] * 3), (16, [55, 55])])
def test_ntt_intt_bit_exact(O, pkg, dev, log_n, bits):
    n = 1 << log_n
    q = O.coeff_modulus_create(n, bits)
    L = len(q)
    tables = [O.NTTTables(log_n, qi) for qi in q]
    plan = pkg.Plan(dev, log_n, q)
    batch, pcount = 3, 2
    x = _rand_polys(O, 1234 + log_n, (batch, pcount, L, n), q)
    # forward, in place
    d = pkg.to_device(x, dev)
    plan.ntt(d, pcount, L)
    exp = x.copy().reshape(-1)
    O.ntt_forward(exp, batch * pcount, L, log_n, tables)
    got = pkg.to_host(d).reshape(-1)
    assert np.array_equal(got, exp), "forward NTT differs from oracle"
    assert all((got.reshape(-1, L, n)[:, j] < q[j]).all() for j in range(L)), "forward output must be canonical"
    # inverse, out of place
    d2 = pkg.to_device(np.zeros_like(x), dev)
    plan.ntt(d, pcount, L, inverse=True, out=d2)
    exp2 = exp.copy()
    O.ntt_inverse(exp2, batch * pcount, L, log_n, tables)
    assert np.array_equal(pkg.to_host(d2).reshape(-1), exp2), "inverse NTT differs from oracle"
    assert np.array_equal(exp2, x.reshape(-1)), "round trip"
    # source of the out-of-place transform untouched
    assert np.array_equal(pkg.to_host(d).reshape(-1), exp)


def test_ntt_lazy_inputs(O, pkg, dev):
    # forward accepts inputs in [0, 4q), inverse in [0, 2q)
    log_n, n = 13, 8192
    q = O.coeff_modulus_create(n, [50, 50])
    tables = [O.NTTTables(log_n, qi) for qi in q]
    plan = pkg.Plan(dev, log_n, q)
    x = _rand_polys(O, 99, (1, 1, 2, n), [4 * qi for qi in q])
    d = pkg.to_device(x, dev)
    plan.ntt(d, 1, 2)
    exp = x.copy().reshape(-1)
    O.ntt_forward(exp, 1, 2, log_n, tables)
    assert np.array_equal(pkg.to_host(d).reshape(-1), exp)
    y = _rand_polys(O, 98, (1, 1, 2, n), [2 * qi for qi in q])
    d = pkg.to_device(y, dev)
    plan.ntt(d, 1, 2, inverse=True)
    exp = y.copy().reshape(-1)
    O.ntt_inverse(exp, 1, 2, log_n, tables)
    assert np.array_equal(pkg.to_host(d).reshape(-1), exp)


@pytest.mark.parametrize("log_n", [6, 13])
def test_indexer_modes(O, pkg, dev, log_n):
    # NTTTableIndexer (utils/ntt.h:105-124): KeySwitchingSetProducts / KeySwitchingSkipFinals
    n = 1 << log_n
    q = O.coeff_modulus_create(n, [40, 40, 40, 41])
    K, L = 4, 2
    tables = [O.NTTTables(log_n, qi) for qi in q]
    plan = pkg.Plan(dev, log_n, q)
    # set products: pcount = L+1 rows, ncomp = L; row k uses table k (or last for k == L)
    row_mod = [q[0], q[1], q[3]]
    x = np.empty((1, L + 1, L, n), dtype=np.uint64)
    for k in range(L + 1):
        for j in range(L):
            x[0, k, j] = O.fill_uniform(5 + k * 3 + j, row_mod[k], n)
    d = pkg.to_device(x, dev)
    plan.ntt(d, L + 1, L, mode=pkg.IDX_KS_SET_PRODUCTS, decomp=L, table_count=K)
    exp = x.copy().reshape(-1)
    O.ntt_forward(exp, L + 1, L, log_n, tables, mode=1, decomp=L)
    assert np.array_equal(pkg.to_host(d).reshape(-1), exp)
    # skip finals: pcount = 2, ncomp = L+1; component j uses table j (or last for j == L)
    comp_mod = [q[0], q[1], q[3]]
    y = np.empty((1, 2, L + 1, n), dtype=np.uint64)
    for k in range(2):
        for j in range(L + 1):
            y[0, k, j] = O.fill_uniform(50 + k * 3 + j, comp_mod[j], n)
    d = pkg.to_device(y, dev)
    plan.ntt(d, 2, L + 1, inverse=True, mode=pkg.IDX_KS_SKIP_FINALS, decomp=L, table_count=K)
    exp = y.copy().reshape(-1)
    O.ntt_inverse(exp, 2, L + 1, log_n, tables, mode=2, decomp=L)
    assert np.array_equal(pkg.to_host(d).reshape(-1), exp)


def test_ntt_linearity_full_size(O, pkg, dev):
    # size-independent property at the BASELINE size (N=32768, 10 limbs): NTT(a+b) = NTT(a)+NTT(b) mod q,
    # and INTT(NTT(a)) == a, on a batch that fills the chip
    log_n, n = 15, 32768
    q = O.coeff_modulus_create(n, [50] * 11)[:10]
    plan = pkg.Plan(dev, log_n, q)
    import torch
    g = torch.Generator(device="cpu").manual_seed(7)
    batch = 16
    qa = np.array(q, dtype=np.uint64).reshape(1, 1, 10, 1)
    a = (torch.randint(0, 2 ** 62, (batch, 2, 10, n), generator=g, dtype=torch.int64).numpy().view(np.uint64) % qa)
    b = (torch.randint(0, 2 ** 62, (batch, 2, 10, n), generator=g, dtype=torch.int64).numpy().view(np.uint64) % qa)
    da, db = pkg.to_device(a, dev), pkg.to_device(b, dev)
    dsum = plan.add(da, db, 10)
    fa, fb, fs = plan.ntt(da.clone(), 2, 10), plan.ntt(db.clone(), 2, 10), plan.ntt(dsum.clone(), 2, 10)
    assert torch.equal(plan.add(fa, fb, 10), fs)
    back = plan.ntt(fa.clone(), 2, 10, inverse=True)
    assert torch.equal(back, da)


def test_reference_kat_galois(pkg, dev):
    """GaloisTool known answers of the reference (test/utils/galois.cu:50-76, committed as data in tests/golden/ref_kats.json): X -> X^3 on 0..7 modulo 17,
    coefficient form and NTT form, through troyn_apply_galois"""
    import json
    import os
    a = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_kats.json")))["galois"]["apply"]
    plan = pkg.Plan(dev, 3, [a["modulus"]])
    x = pkg.to_device(np.array(a["input"], dtype=np.uint64).reshape(1, 1, 8), dev)
    assert pkg.to_host(plan.apply_galois_poly(x, 1, a["element"], False)).reshape(-1).tolist() == a["coefficient_form"]
    assert pkg.to_host(plan.apply_galois_poly(x, 1, a["element"], True)).reshape(-1).tolist() == a["ntt_form"]
    assert pkg.to_host(plan.apply_galois_plain(x.view(1, 8), a["modulus"], a["element"])).reshape(-1).tolist() == a["coefficient_form"]
