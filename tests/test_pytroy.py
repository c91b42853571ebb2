"""The reference's Python surface (pytroy / pytroy_raw) over the C++ mirror: parameter objects work without a GPU;
the evaluation flow (written like pybind/tests/test_basics.py and test_he_operations.py) runs on the GPU."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "troy-nova_amd")


@pytest.fixture(scope="module")
def pytroy():
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import torch  # noqa: F401  (first: one HIP runtime per process -- torch bundles its own libamdhip64)
    try:
        import pytroy as m
    except ImportError as e:
        pytest.fail("pytroy_raw is not built (python -c 'import __graft_entry__ as g; g.build()'): %s" % e)
    return m


def _params(pytroy, scheme, n, bits, t_bits=20):
    p = pytroy.EncryptionParameters(scheme)
    p.set_poly_modulus_degree(n)
    p.set_coeff_modulus(pytroy.CoeffModulus.create(n, bits))
    if scheme != pytroy.SchemeType.CKKS:
        p.set_plain_modulus(pytroy.PlainModulus.batching(n, t_bits))
    return p


def test_parameter_objects(pytroy):
    assert pytroy.it_works() == 42
    assert pytroy.Modulus(1234).value() == 1234
    bits = [60, 40, 40, 60]
    moduli = pytroy.CoeffModulus.create(8192, bits)
    assert [m.bit_count() for m in moduli] == bits
    p = _params(pytroy, pytroy.SchemeType.BFV, 8192, bits)
    assert p.scheme() == pytroy.SchemeType.BFV and p.poly_modulus_degree() == 8192
    assert [m.value() for m in p.coeff_modulus()] == [m.value() for m in moduli]
    assert p.plain_modulus().value() == 1032193
    ctx = pytroy.HeContext(p)
    assert not ctx.on_device() and ctx.using_keyswitching()
    chain, cd = [], ctx.key_context_data()
    while cd is not None:
        chain.append(len(cd.parms().coeff_modulus()))
        cd = cd.next_context_data()
    assert chain == [4, 3, 2, 1]
    assert ctx.first_parms_id() == ctx.key_context_data().next_context_data().parms_id()
    assert pytroy.parms_id_zero.is_zero()


def test_surface_is_complete(pytroy):
    """every method name the reference's pybind module registers (tests/golden/pytroy_surface.json, listed from pybind/src/*.cu by
    tests/golden/make_pytroy_surface.py) exists on the corresponding class of the built module"""
    import pytroy.pytroy_raw as raw
    surface = json.load(open(os.path.join(ROOT, "tests", "golden", "pytroy_surface.json")))
    missing = []
    for group, entry in surface.items():
        have = set()
        for cls in entry["classes"]:
            have |= set(dir(raw)) if cls == "<module>" else set(dir(getattr(raw, cls)))
        missing += ["%s.%s" % (group, n) for n in entry["names"] if n not in have]
    assert not missing, missing
    assert sum(len(e["names"]) for e in surface.values()) > 400
    # keyword names and their order: every argument list the reference declares is a prefix of one of the overloads here
    import re
    wrong, checked = [], 0
    for group, entry in surface.items():
        for name, overloads in entry.get("signatures", {}).items():
            mine = []
            for cls in entry["classes"]:
                fn = getattr(raw, name, None) if cls == "<module>" else getattr(getattr(raw, cls), name, None)
                for sig in re.findall(r"%s\((.*?)\) ->" % name, (getattr(fn, "__doc__", "") or "")):
                    args = [a.split(":")[0].strip() for a in re.split(r",\s*(?![^\[]*\])", sig) if a.strip()]
                    mine.append([a for a in args if a != "self"])
            for want in overloads:
                checked += 1
                if not any(have[:len(want)] == want for have in mine):
                    wrong.append((group, name, want, mine[:3]))
    assert not wrong, wrong
    assert checked > 200


@pytest.mark.gpu
def test_quickstart_flow_in_python(pytroy, dev):
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "config1_digests.json")))
    p = _params(pytroy, pytroy.SchemeType.BFV, 8192, [40, 40, 40])
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Classical128, G["seed"])
    ctx.to_device_inplace()
    assert ctx.pool() == pytroy.MemoryPool.global_pool()
    encoder = pytroy.BatchEncoder(ctx)
    encoder.to_device_inplace()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_public_key(keygen.create_public_key(False))
    decryptor = pytroy.Decryptor(ctx, keygen.secret_key())
    evaluator = pytroy.Evaluator(ctx)
    c = encryptor.encrypt_asymmetric_new(encoder.encode_simd_new([1, 2, 3, 4]))

    def digest(words):
        h = 1469598103934665603
        for w in words:
            h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h
    assert digest(c.data()) == G["ciphertext_digest"]                 # the reference's own output for this seed
    dec = lambda ct: encoder.decode_simd_new(decryptor.decrypt_new(ct)).tolist()
    assert dec(c)[:5] == [1, 2, 3, 4, 0]
    assert dec(evaluator.add_new(c, c))[:4] == [2, 4, 6, 8]
    prod = evaluator.multiply_new(c, c)
    assert digest(prod.data()) == G["multiply_digest"] and prod.polynomial_count() == 3
    relin = evaluator.relinearize_new(prod, keygen.create_relin_keys(False))
    assert relin.polynomial_count() == 2 and dec(relin)[:4] == [1, 4, 9, 16]
    low = evaluator.mod_switch_to_next_new(relin)
    assert low.coeff_modulus_size() == 1 and dec(low)[:4] == [1, 4, 9, 16]
    w = encoder.encode_simd_new([3, 5, 7, 11])
    assert dec(evaluator.multiply_plain_new(c, w))[:4] == [3, 10, 21, 44]
    gk = keygen.create_galois_keys(False)
    r = encryptor.encrypt_asymmetric_new(encoder.encode_simd_new(list(range(1, 8193))))
    assert dec(evaluator.rotate_rows_new(r, 3, gk))[:3] == [4, 5, 6]
    assert dec(evaluator.rotate_columns_new(r, gk))[:2] == [4097, 4098]
    # save / load through byte strings (reference layout)
    blob = c.save(ctx)
    assert len(blob) == 1 + 32 + 24 + 1 + 2 * 2 * 8192 * 8
    assert pytroy.Ciphertext.load_new(blob, ctx).data() == c.data()
    rk = keygen.create_relin_keys(True)                                  # seeded keys: c1 regenerated on load
    rk2 = pytroy.RelinKeys()
    rk2.load(rk.save(ctx), ctx)
    assert dec(evaluator.relinearize_new(prod, rk2))[:4] == [1, 4, 9, 16]
    host = c.clone()
    host.to_host_inplace()
    with pytest.raises(ValueError):
        evaluator.add_new(host, c)                                      # std::invalid_argument -> ValueError
    pytroy.MemoryPool.destroy_global_pool()


@pytest.mark.gpu
def test_ckks_flow_in_python(pytroy, dev):
    import cmath
    import random
    p = _params(pytroy, pytroy.SchemeType.CKKS, 8192, [40, 40, 40, 40])
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Classical128, 99)
    ctx.to_device_inplace()
    enc = pytroy.CKKSEncoder(ctx)
    kg = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_public_key(kg.create_public_key(False))
    dec = pytroy.Decryptor(ctx, kg.secret_key())
    ev = pytroy.Evaluator(ctx)
    rnd = random.Random(3)
    z1 = [complex(rnd.uniform(-1, 1), rnd.uniform(-1, 1)) for _ in range(enc.slot_count())]
    z2 = [complex(rnd.uniform(-1, 1), rnd.uniform(-1, 1)) for _ in range(enc.slot_count())]
    scale = float(1 << 30)
    c1 = encryptor.encrypt_asymmetric_new(enc.encode_complex64_simd_new(z1, None, scale))
    c2 = encryptor.encrypt_asymmetric_new(enc.encode_complex64_simd_new(z2, None, scale))
    rk = kg.create_relin_keys(False)
    m = ev.relinearize_new(ev.multiply_new(c1, c2), rk)
    ev.rescale_to_next_inplace(m)
    got = enc.decode_complex64_simd_new(dec.decrypt_new(m)).tolist()
    assert max(abs(g - a * b) for g, a, b in zip(got, z1, z2)) < 1e-2
    # the fused method (an addition to the reference's surface) is word-identical to the three calls, single and batched
    f = ev.multiply_relinearize_rescale_new(c1, c2, rk)
    assert f.data() == m.data() and f.scale() == m.scale() and f.parms_id() == m.parms_id()
    outs = [pytroy.Ciphertext() for _ in range(4)]
    ev.multiply_relinearize_rescale_batched([c1, c2, c1, c2], [c2, c1, c2, c1], rk, outs)
    assert all(o.data() == m.data() for o in outs)
    gk = kg.create_galois_keys_from_steps([2], False)
    rot = enc.decode_complex64_simd_new(dec.decrypt_new(ev.rotate_vector_new(c1, 2, gk))).tolist()
    assert max(abs(rot[i] - z1[(i + 2) % len(z1)]) for i in range(len(z1))) < 2e-2   # scale 2^30: key-switch noise ~2^-10
    pytroy.MemoryPool.destroy_global_pool()


@pytest.mark.gpu
def test_ckks_scale_out_of_bounds_throws(pytroy, dev):
    """is_scale_within_bounds (evaluator_utils.h:307-323): the reference refuses a CKKS product whose scale reaches the total
    coefficient-modulus bit count (evaluator.cu:140-143) instead of returning a ciphertext that decrypts to garbage"""
    p = _params(pytroy, pytroy.SchemeType.CKKS, 4096, [40, 40, 40])
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Nil, 7)
    ctx.to_device_inplace()
    enc = pytroy.CKKSEncoder(ctx)
    kg = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_public_key(kg.create_public_key(False))
    ev = pytroy.Evaluator(ctx)
    z = [complex(0.5, 0.0)] * enc.slot_count()
    ok = encryptor.encrypt_asymmetric_new(enc.encode_complex64_simd_new(z, None, float(1 << 30)))
    ev.multiply_new(ok, ok)                                            # 2^60 < 2^80 data bits: accepted
    big = encryptor.encrypt_asymmetric_new(enc.encode_complex64_simd_new(z, None, float(1 << 40)))
    with pytest.raises(ValueError, match="Scale out of bounds"):
        ev.multiply_new(big, big)                                      # 2^80 >= 80 bits of data modulus
    with pytest.raises(ValueError, match="Scale out of bounds"):
        ev.square_new(big)
    pytroy.MemoryPool.destroy_global_pool()


@pytest.mark.gpu
def test_lwe_packing_flow_in_python(pytroy, dev):
    """pybind/tests-style use of the LWE / packing defs of pybind/src/evaluator.cu: extract_lwe_new, assemble_lwe_new,
    pack_lwe_ciphertexts_new(_batched), pack_rlwe_ciphertexts_new, negacyclic_shift_new, add_plain_new, field traces"""
    n = 4096
    p = _params(pytroy, pytroy.SchemeType.BFV, n, [40, 40, 40, 40])
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Nil, 0x99)
    ctx.to_device_inplace()
    t = p.plain_modulus().value()
    encoder = pytroy.BatchEncoder(ctx)
    encoder.to_device_inplace()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_secret_key(keygen.secret_key())
    decryptor = pytroy.Decryptor(ctx, keygen.secret_key())
    evaluator = pytroy.Evaluator(ctx)
    auto = keygen.create_automorphism_keys(False)
    msg = [(7 * i + 3) % t for i in range(n)]
    dec = lambda ct: encoder.decode_polynomial_new(decryptor.decrypt_new(ct)).tolist()
    c = encryptor.encrypt_symmetric_new(encoder.encode_polynomial_new(msg), False)
    assert dec(c) == msg
    # X^5 * m: coefficients move up by 5, the wrapped ones change sign
    sh = dec(evaluator.negacyclic_shift_new(c, 5))
    assert sh[5:] == msg[:n - 5] and sh[:5] == [(t - v) % t for v in msg[n - 5:]]
    bias = [(11 * i + 1) % t for i in range(n)]
    assert dec(evaluator.add_plain_new(c, encoder.encode_polynomial_new(bias))) == [(a + b) % t for a, b in zip(msg, bias)]
    assert dec(evaluator.sub_plain_new(c, encoder.encode_polynomial_new(bias))) == [(a - b) % t for a, b in zip(msg, bias)]
    terms = [0, 9, n - 1, 1234, 77]
    lwes = [evaluator.extract_lwe_new(c, term) for term in terms]
    assert lwes[0].poly_modulus_degree() == n and lwes[0].coeff_modulus_size() == 3
    for lwe, term in zip(lwes, terms):
        assert dec(evaluator.assemble_lwe_new(lwe))[0] == msg[term]
    packed = dec(evaluator.pack_lwe_ciphertexts_new(lwes, auto))
    stride = n // 8                                                       # 5 LWEs -> 8 slots
    want = [0] * n
    for i, term in enumerate(terms):
        want[i * stride] = msg[term]
    assert packed == want
    both = evaluator.pack_lwe_ciphertexts_new_batched([lwes, lwes[:2]], auto)
    assert dec(both[0]) == want
    w2 = [0] * n
    w2[0], w2[n // 8] = msg[terms[0]], msg[terms[1]]                      # the tree depth follows the largest group
    assert dec(both[1]) == w2
    # MatmulHelper::pack_outputs's call: members carry results on coefficients = 3 (mod 4); member k lands on offset k
    others = [encryptor.encrypt_symmetric_new(encoder.encode_polynomial_new([(msg[i] * (k + 2)) % t for i in range(n)]), False) for k in range(3)]
    merged = dec(evaluator.pack_rlwe_ciphertexts_new([c] + others, auto, 2 * n - 3, 4, 1))
    for k in range(4):
        src = msg if k == 0 else [(v * (k + 1)) % t for v in msg]
        assert merged[k::4] == src[3::4], k
    pytroy.MemoryPool.destroy_global_pool()


@pytest.mark.gpu
def test_matmul_and_conv2d_helpers_in_python(pytroy, dev):
    """pybind/src/matmul_helper.cu and conv2d_helper.cu names over numpy arrays: the flows of examples/10_bfv_matmul.cu (packed
    outputs) and 14_bfv_conv2d.cu from Python"""
    import numpy as np
    n, t = 8192, 1 << 21
    p = pytroy.EncryptionParameters(pytroy.SchemeType.BFV)
    p.set_poly_modulus_degree(n)
    p.set_coeff_modulus(pytroy.CoeffModulus.create(n, [60, 40, 40, 60]))
    p.set_plain_modulus(t)
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Classical128, 0x42)
    ctx.to_device_inplace()
    encoder = pytroy.BatchEncoder(ctx)
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_secret_key(keygen.secret_key())
    decryptor = pytroy.Decryptor(ctx, keygen.secret_key())
    evaluator = pytroy.Evaluator(ctx)
    rs = np.random.RandomState(5)
    m, r, o = 20, 33, 17
    x, w, s = (rs.randint(0, t, shape).astype(np.uint64) for shape in ((m, r), (r, o), (m, o)))
    helper = pytroy.MatmulHelper(m, r, o, n, pytroy.MatmulObjective.EncryptLeft, True)
    assert helper.input_block() == 16 and helper.pack_lwe()
    w_enc, s_enc = helper.encode_weights(encoder, w), helper.encode_outputs(encoder, s)
    x_enc = pytroy.Cipher2d.load_new(helper.encrypt_inputs(encryptor, encoder, x).save(ctx), ctx)
    y = helper.matmul(evaluator, x_enc, w_enc)
    y.mod_switch_to_next_inplace(evaluator)
    y = helper.pack_outputs(evaluator, keygen.create_automorphism_keys(False), y)
    y.add_plain_inplace(evaluator, s_enc)
    y = helper.deserialize_outputs(evaluator, helper.serialize_outputs(evaluator, y))
    got = helper.decrypt_outputs(encoder, decryptor, y).reshape(m, o)
    want = ((x.astype(object) @ w.astype(object) + s.astype(object)) % t).astype(np.uint64)
    assert np.array_equal(got, want)
    # conv2d
    bs, ic, oc, H, W, kh, kw = 2, 3, 4, 12, 11, 3, 2
    xi = rs.randint(0, t, (bs, ic, H, W)).astype(np.uint64)
    wk = rs.randint(0, t, (oc, ic, kh, kw)).astype(np.uint64)
    conv = pytroy.Conv2dHelper(bs, ic, oc, H, W, kh, kw, n)
    yc = conv.conv2d(evaluator, pytroy.Cipher2d.load_new(conv.encrypt_inputs(encryptor, encoder, xi).save(ctx), ctx), conv.encode_weights(encoder, wk))
    gotc = conv.decrypt_outputs(encoder, decryptor, yc).reshape(bs, oc, H - kh + 1, W - kw + 1)
    wantc = np.zeros(gotc.shape, dtype=object)
    for a in range(kh):
        for b in range(kw):
            wantc += np.einsum("bchw,oc->bohw", xi[:, :, a:a + H - kh + 1, b:b + W - kw + 1].astype(object), wk[:, :, a, b].astype(object))
    assert np.array_equal(gotc, (wantc % t).astype(np.uint64))
    with pytest.raises(ValueError):
        helper.encode_weights(encoder, w[:-1])
    pytroy.MemoryPool.destroy_global_pool()


@pytest.mark.gpu
def test_wider_surface_in_python(pytroy, dev):
    """the rest of the reference's Python names: plaintext scaling (BatchEncoder.scale_up / centralize / scale_down), the ring-2^k
    encoder, apply_galois_plain, accessors and wire formats of keys and parameters, save_terms / load_terms, the CKKS integer
    encodings and the CKKS forms of Conv2dHelper"""
    import numpy as np
    n = 8192
    p = _params(pytroy, pytroy.SchemeType.BFV, n, [40, 40, 40])
    t = p.plain_modulus().value()
    assert pytroy.EncryptionParameters.load_new(p.save()).parms_id() == p.parms_id() and p.serialized_size_upperbound() == len(p.save())
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Classical128, 0x77)
    ctx.to_device_inplace()
    encoder = pytroy.BatchEncoder(ctx)
    assert encoder.row_count() == 2 and encoder.column_count() == n // 2 and encoder.simd_encoding_supported()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_secret_key(keygen.secret_key())
    pk = keygen.create_public_key(True)
    assert pk.contains_seed()
    pk2 = pytroy.PublicKey.load_new(pk.save(ctx), ctx)                       # load expands the seed
    assert not pk2.contains_seed() and pk2.parms_id() == ctx.key_parms_id() and len(pk.save(ctx)) <= pk.serialized_size_upperbound(ctx)
    encryptor.set_public_key(pk2)
    assert encryptor.public_key().parms_id() == pk2.parms_id() and encryptor.secret_key().on_device()
    decryptor = pytroy.Decryptor(ctx, keygen.secret_key())
    evaluator = pytroy.Evaluator(ctx)
    rs = np.random.RandomState(3)
    msg = [int(v) for v in rs.randint(0, t, 100)]
    plain = encoder.encode_polynomial_new(msg)
    # numpy arrays are accepted wherever the reference takes py::array_t, and decode returns numpy arrays
    from_np = encoder.decode_polynomial_new(encoder.encode_polynomial_new(np.array(msg, dtype=np.uint64)))
    assert isinstance(from_np, np.ndarray) and from_np.dtype == np.uint64 and from_np.tolist() == msg
    simd = np.arange(n, dtype=np.uint64) % np.uint64(t)
    assert np.array_equal(encoder.decode_simd_new(encoder.encode_simd_new(simd)), simd)
    # scale_up / scale_down / centralize, partial RNS plaintexts as operands
    up = encoder.scale_up_new(plain)
    assert up.coeff_count() == 100 and len(up.obtain_data()) == 200 and not up.parms_id().is_zero() and up.parms_id() == ctx.first_parms_id()
    assert encoder.decode_polynomial_new(encoder.scale_down_new(up)).tolist() == msg
    c = encryptor.encrypt_symmetric_new(up, False)
    assert encoder.decode_polynomial_new(decryptor.decrypt_new(c)).tolist()[:100] == msg
    ca = encryptor.encrypt_asymmetric_new(plain)
    assert not ca.is_transparent() and pytroy.Ciphertext().is_transparent() and ca.correction_factor() == 1 and ca.seed() == 0
    assert ca.data_address() != 0 and ca.device_index() == 0 and ca.to_host().on_device() is False
    cen = encoder.centralize_new(plain)
    assert encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(ca, cen))).tolist() == \
        encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(ca, plain))).tolist()
    # save_terms / load_terms: only the listed coefficients of c0 travel
    terms = [0, 5, 17, 99]
    blob = ca.save_terms(ctx, terms)
    assert len(blob) <= ca.serialized_terms_size_upperbound(ctx, terms) and len(blob) < len(ca.save(ctx))
    back = encoder.decode_polynomial_new(decryptor.decrypt_new(pytroy.Ciphertext.load_terms_new(blob, ctx, terms))).tolist()
    assert [back[i] for i in terms] == [msg[i] for i in terms]
    # the plaintext automorphism against the one under encryption
    slots = [int(v) for v in rs.randint(0, t, n)]
    ps = encoder.encode_simd_new(slots)
    gk = keygen.create_galois_keys(False)
    gk2 = pytroy.KSwitchKeys.load_new(gk.save(ctx), ctx)
    assert gk2.parms_id() == gk.parms_id() and len(gk.save(ctx)) <= gk.serialized_size_upperbound(ctx)
    gk3 = pytroy.GaloisKeys.load_new(gk.save(ctx), ctx)                      # a GaloisKeys again (the reference registers the class on its own)
    assert isinstance(gk3, pytroy.GaloisKeys) and isinstance(gk.clone(), pytroy.GaloisKeys)
    rotated = encoder.decode_simd_new(decryptor.decrypt_new(evaluator.rotate_rows_new(encryptor.encrypt_symmetric_new(ps, False), 1, gk3))).tolist()
    g = 3                                                                     # rotate_rows by one step = the generator itself
    assert encoder.decode_simd_new(evaluator.apply_galois_plain_new(ps, g)).tolist() == rotated and rotated != slots
    sk_plain = keygen.secret_key().get_plaintext()
    assert sk_plain.is_ntt_form() and pytroy.SecretKey(sk_plain).parms_id() == ctx.key_parms_id()
    gen = pytroy.RandomGenerator(7)
    a, b = gen.sample_uint64(), gen.sample_uint64()
    gen.reset_seed(7)
    assert a != b and gen.sample_uint64() == a
    assert pytroy.Modulus(97).reduce_mul(96, 96) == 1 and pytroy.ParmsID.zero().is_zero()
    assert [m.bit_count() for m in pytroy.PlainModulus.batching_multiple(n, [20, 21])] == [20, 21]

    # ring-2^k: products in Z_{2^64} (wider coefficient modulus)
    p2 = pytroy.EncryptionParameters(pytroy.SchemeType.BFV)
    p2.set_poly_modulus_degree(n)
    p2.set_coeff_modulus(pytroy.CoeffModulus.create(n, [60, 60, 60, 60]))
    p2.set_plain_modulus(1 << 20)
    ctx2 = pytroy.HeContext(p2, True, pytroy.SecurityLevel.Nil, 0x78)
    ctx2.to_device_inplace()
    kg2 = pytroy.KeyGenerator(ctx2)
    enc2 = pytroy.Encryptor(ctx2)
    enc2.set_secret_key(kg2.secret_key())
    dec2 = pytroy.Decryptor(ctx2, kg2.secret_key())
    ev2 = pytroy.Evaluator(ctx2)
    ring = pytroy.PolynomialEncoderRing2k64(ctx2, 64)
    assert ring.t_bit_length() == 64 and ring.on_device()
    av = rs.randint(0, 2 ** 63, 40, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1)
    bv = rs.randint(0, 2 ** 63, 30, dtype=np.int64).astype(np.uint64)
    prod = ev2.multiply_plain_new(enc2.encrypt_symmetric_new(ring.scale_up_new(av, None), False), ring.centralize_new(bv, None))
    got = ring.scale_down_new(dec2.bfv_decrypt_without_scaling_down_new(prod))
    want = np.zeros(n, dtype=object)
    for i, x in enumerate(av):
        for j, y in enumerate(bv):
            want[i + j] = (want[i + j] + int(x) * int(y)) % (1 << 64)
    assert [int(v) for v in got] == [int(v) for v in want]

    # CKKS: integer encodings and Conv2dHelper's real-valued forms
    p3 = _params(pytroy, pytroy.SchemeType.CKKS, n, [60, 40, 40, 60])
    ctx3 = pytroy.HeContext(p3, True, pytroy.SecurityLevel.Classical128, 0x79)
    ctx3.to_device_inplace()
    ck = pytroy.CKKSEncoder(ctx3)
    assert ck.poly_modulus_degree() == n
    zs = (rs.uniform(-1, 1, n // 2) + 1j * rs.uniform(-1, 1, n // 2)).astype(np.complex128)
    back = ck.decode_complex64_simd_new(ck.encode_complex64_simd_new(zs, None, 2.0 ** 40))
    assert isinstance(back, np.ndarray) and back.dtype == np.complex128 and np.abs(back - zs).max() < 1e-6
    assert ck.decode_float64_polynomial_new(ck.encode_integer64_polynomial_new([3, -4, 5], None)).tolist()[:4] == [3.0, -4.0, 5.0, 0.0]
    assert all(abs(v - (2 - 1j)) < 1e-6 for v in ck.decode_complex64_simd_new(ck.encode_complex64_single_new(2 - 1j, None, 2.0 ** 30)).tolist())
    assert all(abs(v + 9) < 1e-9 for v in ck.decode_complex64_simd_new(ck.encode_integer64_single_new(-9, None)).tolist())
    kg3 = pytroy.KeyGenerator(ctx3)
    enc3 = pytroy.Encryptor(ctx3)
    enc3.set_secret_key(kg3.secret_key())
    dec3 = pytroy.Decryptor(ctx3, kg3.secret_key())
    ev3 = pytroy.Evaluator(ctx3)
    bs, ic, oc, H, W, kh, kw = 2, 3, 4, 12, 11, 3, 2
    xi, wk = rs.uniform(-1, 1, (bs, ic, H, W)), rs.uniform(-1, 1, (oc, ic, kh, kw))
    conv = pytroy.Conv2dHelper(bs, ic, oc, H, W, kh, kw, n)
    scale = 2.0 ** 20
    yc = conv.conv2d(ev3, pytroy.Cipher2d.load_new(conv.encrypt_inputs_doubles(enc3, ck, xi, None, scale).save(ctx3), ctx3), conv.encode_weights_doubles(ck, wk, None, scale))
    gotc = conv.decrypt_outputs_doubles(ck, dec3, yc).reshape(bs, oc, H - kh + 1, W - kw + 1)
    wantc = np.zeros(gotc.shape)
    for a in range(kh):
        for b in range(kw):
            wantc += np.einsum("bchw,oc->bohw", xi[:, :, a:a + H - kh + 1, b:b + W - kw + 1], wk[:, :, a, b])
    assert np.abs(gotc - wantc).max() < 1e-3
    pytroy.destroy_memory_pool()


@pytest.mark.gpu
def test_ring2k_helpers_in_python(pytroy, dev):
    """MatmulHelper / Conv2dHelper over Z_{2^k} from Python (pybind/tests/test_matmul.py, test_conv2d.py use these names): y = x * w + s
    modulo 2^64 and a convolution modulo 2^32"""
    import numpy as np
    n = 8192
    p = pytroy.EncryptionParameters(pytroy.SchemeType.BFV)
    p.set_poly_modulus_degree(n)
    p.set_coeff_modulus(pytroy.CoeffModulus.create(n, [60, 60, 60, 60, 60]))
    p.set_plain_modulus(1 << 20)
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Nil, 0x99)
    ctx.to_device_inplace()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_secret_key(keygen.secret_key())
    decryptor = pytroy.Decryptor(ctx, keygen.secret_key())
    evaluator = pytroy.Evaluator(ctx)
    rs = np.random.RandomState(17)
    ring64 = pytroy.PolynomialEncoderRing2k64(ctx, 64)
    m, r, o = 7, 20, 9
    x, w, s = (rs.randint(0, 2 ** 63, size, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1) for size in (m * r, r * o, m * o))
    helper = pytroy.MatmulHelper(m, r, o, n, pytroy.MatmulObjective.EncryptLeft, False)
    w_enc = helper.encode_weights_ring2k64(ring64, w, None)
    x_enc = pytroy.Cipher2d.load_new(helper.encrypt_inputs_ring2k64(encryptor, ring64, x, None).save(ctx), ctx)
    y = helper.matmul(evaluator, x_enc, w_enc)
    y.add_plain_inplace(evaluator, helper.encode_outputs_ring2k64(ring64, s, None))
    got = helper.decrypt_outputs_ring2k64(ring64, decryptor, y)
    want = (x.reshape(m, r).astype(object) @ w.reshape(r, o).astype(object) + s.reshape(m, o).astype(object)) % (1 << 64)
    assert [int(v) for v in got] == [int(v) for v in want.reshape(-1)]
    ring32 = pytroy.PolynomialEncoderRing2k32(ctx, 32)
    bs, ic, oc, H, W, kh, kw = 1, 2, 3, 9, 8, 3, 2
    xi = rs.randint(0, 2 ** 32, (bs, ic, H, W), dtype=np.int64).astype(np.uint32)
    wk = rs.randint(0, 2 ** 32, (oc, ic, kh, kw), dtype=np.int64).astype(np.uint32)
    conv = pytroy.Conv2dHelper(bs, ic, oc, H, W, kh, kw, n)
    xin = conv.encrypt_inputs_ring2k32(encryptor, ring32, xi.reshape(-1), None)
    with pytest.raises(ValueError):                                          # seed-compressed as sent: c1 has to be expanded (load does it) before use
        conv.conv2d(evaluator, xin, conv.encode_weights_ring2k32(ring32, wk.reshape(-1), None))
    xin.expand_seed(ctx)
    gotc = conv.decrypt_outputs_ring2k32(ring32, decryptor, conv.conv2d(evaluator, xin, conv.encode_weights_ring2k32(ring32, wk.reshape(-1), None)))
    wantc = np.zeros((bs, oc, H - kh + 1, W - kw + 1), dtype=object)
    for a in range(kh):
        for b in range(kw):
            wantc += np.einsum("bchw,oc->bohw", xi[:, :, a:a + H - kh + 1, b:b + W - kw + 1].astype(object), wk[:, :, a, b].astype(object))
    assert [int(v) for v in gotc] == [int(v) for v in (wantc % (1 << 32)).reshape(-1)]
    pytroy.destroy_memory_pool()


@pytest.mark.gpu
def test_memory_pools_in_python(pytroy, dev):
    """pybind/tests/test_basics.py's pool checks (device side): objects report the pool they were allocated from -- the global pool by
    default, a caller's pool when one is passed"""
    n = 8192
    p = _params(pytroy, pytroy.SchemeType.BFV, n, [60, 40, 40, 60])
    global_pool = pytroy.MemoryPool.global_pool()
    assert global_pool is not None
    ctx = pytroy.HeContext(p)
    encoder = pytroy.BatchEncoder(ctx)
    ctx.to_device_inplace()
    encoder.to_device_inplace()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    pk = keygen.create_public_key(False)
    encryptor.set_public_key(pk)
    assert ctx.pool() == global_pool and pk.pool() == global_pool
    encoded = encoder.encode_simd_new([1, 2, 3, 4])
    assert encoded.pool() == global_pool
    assert encryptor.encrypt_asymmetric_new(encoded).pool() == global_pool
    # custom pools
    context_pool, text_pool = pytroy.MemoryPool(), pytroy.MemoryPool()
    assert context_pool != global_pool and text_pool != context_pool and text_pool != global_pool
    ctx2 = pytroy.HeContext(p)
    enc2 = pytroy.BatchEncoder(ctx2)
    ctx2.to_device_inplace(context_pool)
    enc2.to_device_inplace(context_pool)
    kg2 = pytroy.KeyGenerator(ctx2)
    e2 = pytroy.Encryptor(ctx2)
    pk2 = kg2.create_public_key(False, context_pool)
    e2.set_public_key(pk2)
    assert ctx2.pool() == context_pool and pk2.pool() == context_pool
    encoded2 = enc2.encode_simd_new([1, 2, 3, 4], text_pool)
    assert encoded2.pool() == text_pool
    encrypted2 = e2.encrypt_asymmetric_new(encoded2, text_pool)
    assert encrypted2.pool() == text_pool
    dec = pytroy.Decryptor(ctx2, kg2.secret_key())
    assert enc2.decode_simd_new(dec.decrypt_new(encrypted2)).tolist()[:5] == [1, 2, 3, 4, 0]
    pytroy.destroy_memory_pool()


@pytest.mark.gpu
def test_zstd_wire_format_interoperates(pytroy, dev):
    """CompressionMode.Zstd (round 6: through the zstd runtime library): objects round-trip, structured data shrinks, and -- the interoperability that matters -- a
    frame written by ANOTHER zstd producer with the streaming API (no content size in the frame header: what the reference's compression_zstd.cpp emits) inside the
    reference's framing [mode = 1][u64 compressed size][frame] loads as the same object.  The other producer here is pyarrow's CompressedOutputStream."""
    import struct
    pa = pytest.importorskip("pyarrow")
    p = _params(pytroy, pytroy.SchemeType.BFV, 8192, [40, 40, 40])
    ctx = pytroy.HeContext(p, True, pytroy.SecurityLevel.Classical128, 0x123)
    ctx.to_device_inplace()
    encoder = pytroy.BatchEncoder(ctx)
    encoder.to_device_inplace()
    keygen = pytroy.KeyGenerator(ctx)
    encryptor = pytroy.Encryptor(ctx)
    encryptor.set_secret_key(keygen.secret_key())
    Z, N = pytroy.CompressionMode.Zstd, pytroy.CompressionMode.Nil
    plain = encoder.encode_polynomial_new([5, 4, 3, 2, 1] + [0] * 100)
    nil, z = plain.save(N), plain.save(Z)
    assert z[0] == 1 and len(z) < len(nil) and len(z) <= plain.serialized_size_upperbound(Z)
    assert pytroy.Plaintext.load_new(z).data() == plain.data()
    ct = encryptor.encrypt_symmetric_new(plain, True)
    cz = ct.save(ctx, Z)
    assert len(cz) <= ct.serialized_size_upperbound(ctx, Z)            # uniform residues do not compress: written raw (mode byte 0) when the frame would be longer
    back = pytroy.Ciphertext.load_new(cz, ctx)
    assert not back.contains_seed() and encoder.decode_polynomial_new(pytroy.Decryptor(ctx, keygen.secret_key()).decrypt_new(back)).tolist()[:5] == [5, 4, 3, 2, 1]
    # a frame from a streaming writer, in the reference's framing
    raw = nil[1:]
    sink = pa.BufferOutputStream()
    w = pa.CompressedOutputStream(sink, "zstd")
    w.write(raw)
    w.close()
    frame = sink.getvalue().to_pybytes()
    foreign = bytes([1]) + struct.pack("<Q", len(frame)) + frame
    assert pytroy.Plaintext.load_new(foreign).data() == plain.data() and frame != z[9:]      # another producer's bytes, the same object
    # and a frame of ours opens with another zstd consumer
    assert pa.decompress(z[9:], decompressed_size=len(raw), codec="zstd").to_pybytes() == raw
