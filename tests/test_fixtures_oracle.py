"""The committed fixtures (tests/golden/fixtures_small.npz, fixtures_digests.json) must be reproduced by the oracle:
guards the checker itself against accidental changes.  Runs without a GPU."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = np.load(os.path.join(HERE, "golden", "fixtures_small.npz"))
META = json.load(open(os.path.join(HERE, "golden", "fixtures_digests.json")))


def _ops(ctx, scheme, L, seed):
    a = ctx.random_ct(seed, 2, L)
    b = ctx.random_ct(seed + 1, 2, L)
    keys = ctx.random_keys(seed + 2, L)
    ct3 = ctx.random_ct(seed + 3, 3, L)
    out = {"a": a, "b": b, "ct3": ct3, "ntt_a": ctx.to_ntt(a, 2, L), "intt_a": ctx.from_ntt(a, 2, L),
           "dyadic_ab": ctx.ckks_multiply(L, a, b), "relin": ctx.relinearize(L, scheme == "ckks", ct3, keys)}
    if L >= 2:
        out["mod_switch_scale"] = ctx.mod_switch_scale_to_next(L, a)
        out["mod_switch_drop"] = ctx.mod_switch_drop_to_next(L, a)
    if scheme == "bfv":
        out["bfv_multiply"] = ctx.bfv_multiply(L, a, b)
    return out


@pytest.mark.parametrize("name", sorted(META["small"]))
def test_small_fixtures(O, name):
    m = META["small"][name]
    assert [int(v) for v in O.coeff_modulus_create(m["n"], m["bits"])] == m["q"]
    ctx = O.Context(m["scheme"], m["n"], m["q"], m["t"])
    got = _ops(ctx, m["scheme"], m["L"], m["seed"])
    for k, v in got.items():
        assert np.array_equal(v, SMALL["%s/%s" % (name, k)]), k


@pytest.mark.parametrize("name", sorted(META["large"]))
def test_large_digests(O, name):
    m = META["large"][name]
    ctx = O.Context(m["scheme"], m["n"], m["q"], m["t"])
    got = _ops(ctx, m["scheme"], m["L"], m["seed"])
    for k, v in got.items():
        e = m["ops"][k]
        flat = np.ascontiguousarray(v).reshape(-1)
        assert list(v.shape) == e["shape"], k
        assert "%016x" % O.fnv_words(flat) == e["digest"], k
        assert [int(x) for x in flat[:8]] == e["first8"] and [int(x) for x in flat[-8:]] == e["last8"], k
