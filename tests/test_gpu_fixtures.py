"""HIP path vs the COMMITTED fixtures: expected values come from tests/golden/, not from an oracle call.  Small rings
compare whole arrays; BASELINE configs 2-4 compare 64-bit digests (+ first/last words) of outputs computed from the
seeded inputs."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = np.load(os.path.join(HERE, "golden", "fixtures_small.npz"))
META = json.load(open(os.path.join(HERE, "golden", "fixtures_digests.json")))


def _gpu_ops(pkg, dev, m, a, b, ct3, keys):
    """[1]-batched device evaluation of every fixture op; returns host arrays"""
    L, n, scheme = m["L"], m["n"], m["scheme"]
    plan = pkg.Plan(dev, n.bit_length() - 1, m["q"])
    da, db, d3 = (pkg.to_device(x[None], dev) for x in (a, b, ct3))
    dkeys = [pkg.to_device(k, dev) for k in keys]
    out = {}
    out["ntt_a"] = pkg.to_host(plan.ntt(da.clone(), 2, L))[0]
    out["intt_a"] = pkg.to_host(plan.ntt(da.clone(), 2, L, inverse=True))[0]
    out["dyadic_ab"] = pkg.to_host(plan.dyadic_convolute(da, 2, db, 2, L))[0]
    is_ckks = scheme == "ckks"
    out["relin"] = pkg.to_host(plan.relinearize(L, d3, dkeys, is_ckks=is_ckks, is_ntt_form=is_ckks))[0]
    if L >= 2:
        if is_ckks:
            out["mod_switch_scale"] = pkg.to_host(plan.divide_and_round_q_last_ntt(L, da, 2))[0]
        else:
            out["mod_switch_scale"] = pkg.to_host(plan.divide_and_round_q_last(L, da, 2))[0]
        out["mod_switch_drop"] = pkg.to_host(plan.mod_switch_drop(L, L - 1, da, 2))[0]
    if scheme == "bfv":
        behz = pkg.Behz(plan, L, m["t"])
        out["bfv_multiply"] = pkg.to_host(behz.multiply(da, 2, db, 2))[0]
    return out


@pytest.mark.parametrize("name", sorted(META["small"]))
def test_small_fixtures_on_gpu(pkg, dev, name):
    m = META["small"][name]
    f = lambda k: SMALL["%s/%s" % (name, k)]
    got = _gpu_ops(pkg, dev, m, f("a"), f("b"), f("ct3"), list(f("keys")))
    for k, v in got.items():
        assert np.array_equal(v, f(k)), k


@pytest.mark.parametrize("name", sorted(META["large"]))
def test_large_digests_on_gpu(O, pkg, dev, name):
    m = META["large"][name]
    ctx = O.Context(m["scheme"], m["n"], m["q"], m["t"])          # input generator only (SplitMix streams)
    s, L = m["seed"], m["L"]
    a, b, ct3, keys = ctx.random_ct(s, 2, L), ctx.random_ct(s + 1, 2, L), ctx.random_ct(s + 3, 3, L), ctx.random_keys(s + 2, L)
    assert "%016x" % O.fnv_words(a) == m["ops"]["a"]["digest"]
    got = _gpu_ops(pkg, dev, m, a, b, ct3, keys)
    for k, v in got.items():
        e = m["ops"][k]
        flat = np.ascontiguousarray(v).reshape(-1)
        assert "%016x" % O.fnv_words(flat) == e["digest"], k
        assert [int(x) for x in flat[:8]] == e["first8"] and [int(x) for x in flat[-8:]] == e["last8"], k
