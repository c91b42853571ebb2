"""BASELINE config 1 on the GPU against the reference's own recorded output: the HIP BEHZ multiply of the seed-0x123
ciphertext must hash to the digest the reference produced (tests/golden/config1_digests.json), not merely agree with
the oracle."""
import numpy as np
import pytest

from test_oracle_config1 import G, config1_ciphertext

pytestmark = pytest.mark.gpu


def test_config1_multiply_matches_reference_digest(O, pkg, dev):
    ctx, _, ct = config1_ciphertext(O)
    assert "%016x" % O.fnv_words(ct) == G["ciphertext_digest"]
    plan = pkg.Plan(dev, ctx.log_n, ctx.q)
    behz = pkg.Behz(plan, 2, ctx.t)
    x = pkg.to_device(ct[None], dev)
    got = pkg.to_host(behz.multiply(x, 2, x, 2))[0]
    assert "%016x" % O.fnv_words(got) == G["multiply_digest"]


def test_config1_relinearize_then_drop(O, pkg, dev):
    # the rest of the example's chain after multiply: relinearize + mod_switch_to_next, against the oracle
    ctx, _, ct = config1_ciphertext(O)
    plan = pkg.Plan(dev, ctx.log_n, ctx.q)
    behz = pkg.Behz(plan, 2, ctx.t)
    x = pkg.to_device(ct[None], dev)
    prod = behz.multiply(x, 2, x, 2)
    keys_h = ctx.random_keys(5, 2)
    keys = [pkg.to_device(k, dev) for k in keys_h]
    rel = plan.relinearize(2, prod, keys, is_ntt_form=False)
    want = ctx.relinearize(2, False, pkg.to_host(prod)[0], keys_h)
    assert np.array_equal(pkg.to_host(rel)[0], want)
    nxt = plan.divide_and_round_q_last(2, rel, 2)
    assert np.array_equal(pkg.to_host(nxt)[0], ctx.mod_switch_scale_to_next(2, want))
