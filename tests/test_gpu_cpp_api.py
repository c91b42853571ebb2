"""GPU parity through the host-side C++ mirror of the reference API (troy::Evaluator in
troy-nova_amd/troy/troy.h): a C++ program written like the reference's own tests is run and every
result is compared bit-for-bit with the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "cpp", "evaluator_driver")


def _read_dump(path):
    raw = np.fromfile(path, dtype=np.uint64)
    out, pos = [], 0
    while pos < raw.size:
        p, l, n, ntt = (int(x) for x in raw[pos:pos + 4])
        pos += 4
        out.append((raw[pos:pos + p * l * n].reshape(p, l, n), bool(ntt)))
        pos += p * l * n
    return out


@pytest.mark.parametrize("scheme,n,t,bits", [("ckks", 32, 0, [40, 40, 40, 40]), ("ckks", 8192, 0, [40, 40, 40, 40]),
                                             ("ckks", 16384, 0, [50] * 6), ("bfv", 32, 65537, [40, 40, 40]),
                                             ("bfv", 8192, 1032193, [40, 40, 40]),
                                             ("bgv", 32, 65537, [40, 40, 40]), ("bgv", 8192, 1032193, [40, 40, 40]), ("bgv", 16384, 65537, [50] * 5)])
def test_evaluator_cpp_api(O, dev, tmp_path, scheme, n, t, bits):
    if not os.path.exists(DRIVER):
        pytest.fail("tests/cpp/evaluator_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = str(tmp_path / "dump.bin")
    r = subprocess.run([DRIVER, scheme, str(n), str(t), out] + [str(b) for b in bits], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    q = O.coeff_modulus_create(n, bits)
    ctx = O.Context(scheme, n, q, t)
    L = len(q) - 1
    ntt = scheme != "bfv"                                    # CKKS and BGV ciphertexts live in NTT form
    a, b = ctx.random_ct(11, 2, L), ctx.random_ct(29, 2, L)
    keys = ctx.random_keys(7, L)
    d = _read_dump(out)
    prod = ctx.ckks_multiply(L, a, b) if ntt else ctx.bfv_multiply(L, a, b)
    assert np.array_equal(d[0][0], prod) and d[0][1] == ntt
    relin = ctx.relinearize(L, ntt, prod, keys)
    assert np.array_equal(d[1][0], relin)
    assert np.array_equal(d[2][0], ctx.mod_switch_scale_to_next(L, relin))
    mods = ctx.moduli()
    af, bf = a.reshape(-1), b.reshape(-1)
    tmp = np.empty_like(af)
    O.lib().orc_add_ps(O.ptr(af), O.ptr(bf), 2, n, mods, L, O.ptr(tmp))
    assert np.array_equal(d[3][0].reshape(-1), tmp)
    O.lib().orc_sub_ps(O.ptr(af), O.ptr(bf), 2, n, mods, L, O.ptr(tmp))
    assert np.array_equal(d[4][0].reshape(-1), tmp)
    O.lib().orc_negate_ps(O.ptr(af), 2, n, mods, L, O.ptr(tmp))
    assert np.array_equal(d[5][0].reshape(-1), tmp)
    exp = ctx.from_ntt(a, 2, L) if ntt else ctx.to_ntt(a, 2, L)
    assert np.array_equal(d[6][0], exp) and d[6][1] == (not ntt)
    assert np.array_equal(d[7][0], relin)            # multiply_inplace + relinearize_inplace
    # 3-poly + 2-poly add / sub: the longer operand's tail is copied (add) or negated (sub)
    pf = prod.reshape(-1)
    O.lib().orc_add_ps(O.ptr(np.ascontiguousarray(pf[:2 * L * n])), O.ptr(af), 2, n, mods, L, O.ptr(tmp))
    assert np.array_equal(d[8][0][:2].reshape(-1), tmp) and np.array_equal(d[8][0][2], prod[2])
    O.lib().orc_sub_ps(O.ptr(af), O.ptr(np.ascontiguousarray(pf[:2 * L * n])), 2, n, mods, L, O.ptr(tmp))
    neg = np.empty(L * n, dtype=np.uint64)
    O.lib().orc_negate_ps(O.ptr(np.ascontiguousarray(prod[2].reshape(-1))), 1, n, mods, L, O.ptr(neg))
    assert np.array_equal(d[9][0][:2].reshape(-1), tmp) and np.array_equal(d[9][0][2].reshape(-1), neg)


def test_quickstart_cpp_api(dev, tmp_path):
    """BASELINE config 1 through the C++ mirror (keygen, encoder, encryptor, evaluator, decryptor on the GPU): the
    digests must equal the ones the reference itself produced for seed 0x123 (tests/golden/config1_digests.json)."""
    import json
    drv = os.path.join(ROOT, "tests", "cpp", "quickstart_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/quickstart_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "config1_digests.json")))
    ct_file = str(tmp_path / "ct.bin")
    r = subprocess.run([drv, hex(G["seed"]), ct_file], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    # serialization: the reference's raw layout (utils/serialize.h, ciphertext.cu:93-140), parsed here field by field
    import hashlib
    import struct
    n, L = 8192, 2
    body = 2 * L * n * 8
    assert kv["ser_ct"] == [str(1 + 32 + 24 + 1 + body), "1"] and kv["ser_params"] == ["1"]
    assert kv["ser_relin"] == ["1", "4", "9", "16", "0", "0"] and kv["ser_plain"] == ["3", "5", "7", "11", "0", "0"]
    assert kv["ser_seeded"] == [str(1 + 32 + 24 + 1 + 8 + body // 2), "1"] and kv["ser_seeded_dec"] == ["9", "8", "7", "0", "0", "0"]
    raw = open(ct_file, "rb").read()
    words = [1, n] + G["coeff_modulus"][:L] + [G["plain_modulus"]]              # scheme BFV = 1, first data level
    assert raw[0] == 0                                                           # CompressionMode::Nil
    assert raw[1:33] == hashlib.blake2b(struct.pack("<%dQ" % len(words), *words), digest_size=32).digest()     # ParmsID
    assert struct.unpack("<3Q", raw[33:57]) == (2, L, n) and raw[57] == 0b100   # not NTT, no seed, on device
    h = 1469598103934665603
    for w in struct.unpack("<%dQ" % (2 * L * n), raw[58:]):
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert "%016x" % h == G["ciphertext_digest"]                                 # payload = the reference's ciphertext words
    assert kv["ct_digest"] == [G["ciphertext_digest"]]
    assert kv["mul_digest"] == [G["multiply_digest"]]
    assert kv["decrypt"] == ["1", "2", "3", "4", "0", "0"]
    assert kv["add"] == ["2", "4", "6", "8", "0", "0"]
    for k in ("mul", "relin", "modswitch"):
        assert kv[k] == ["1", "4", "9", "16", "0", "0"], k
    assert kv["relin_polys"] == ["2"] and kv["low_limbs"] == ["1"]
    assert kv["mul2"] == ["5", "12", "21", "32", "0", "0"]
    assert kv["sub"] == ["4", "4", "4", "4", "0", "0"]
    assert kv["mulplain"] == ["3", "10", "21", "44", "0", "0"]
    assert kv["macc"] == ["13", "22", "35", "60", "0", "0"]       # {1,2,3,4}*{3,5,7,11} + {5,6,7,8}*2
    assert kv["rot1"] == ["2", "3", "4", "5", "6", "7"]
    assert kv["rot3"] == ["4", "5", "6", "7", "8", "9"]
    assert kv["rotm2"] == ["4095", "4096", "1", "2", "3", "4"]
    assert kv["rotcol"] == ["4097", "4098", "4099", "4100", "4101", "4102"]
    assert kv["host_plain_rejected"] == ["1"]
    # a different seed gives a different ciphertext but the same plaintext results
    r2 = subprocess.run([drv, "0x456"], capture_output=True, text=True, timeout=600)
    kv2 = {ln.split()[0]: ln.split()[1:] for ln in r2.stdout.splitlines() if ln.strip()}
    assert kv2["ct_digest"] != kv["ct_digest"] and kv2["mul"] == kv["mul"]


def test_ckks_cpp_api(dev):
    """BASELINE config 3's parameters through the C++ mirror with the CKKS encoder: every result must decode to the
    expected slots (tolerances: fresh 1e-6, after one multiplication + rescale at scale 2^30 1e-3)."""
    drv = os.path.join(ROOT, "tests", "cpp", "ckks_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/ckks_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    assert float(kv["encode_roundtrip"][0]) < 1e-8
    assert float(kv["decrypt"][0]) < 1e-6 and float(kv["add"][0]) < 1e-6
    assert kv["levels"][0] == "4" and abs(float(kv["levels"][2]) - 30.0) < 0.01
    for k in ("mul_relin_rescale", "multiply_plain"):
        assert float(kv[k][0]) < 1e-3, k
    for k in ("rotate1", "rotate-3", "conjugate"):
        assert float(kv[k][0]) < 1e-4, k


def test_fused_multiply_relinearize_rescale_cpp_api(dev):
    """Evaluator::multiply_relinearize_rescale{,_new,_inplace,_batched} (the fused chain behind the mirror of the reference's API) is
    bit-identical -- payload, parms_id, scale -- to multiply + relinearize + rescale_to_next at BASELINE config 3's parameters, for single
    objects, uniform batches and a batch with mixed levels; it decrypts to the product; the reference's argument checks still throw."""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/he_bench_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    for k in ("fused_single_identical", "fused_inplace_identical", "fused_batched_identical", "fused_mixed_levels_identical"):
        assert kv[k][0] == "1", k
    assert float(kv["fused_single_error"][0]) < 1e-3
    assert kv["fused_errors"][0] == "2"
    # call combining: 8 threads x 4 rounds of the three calls and of the fused call on distinct operands at two levels, word-identical to the
    # uncombined calls; an argument error stays with its thread; a thread that is alone is not combined
    assert kv["combined_identical"][0] == "1" and kv["combined_errors"][0] == "1"
    assert int(kv["combined_calls"][0]) > 0 and int(kv["combined_largest_batch"][0]) >= 2
    assert kv["combined_alone_calls"][0] == "0" and kv["combined_alone_identical"][0] == "1"


@pytest.mark.parametrize("dims,pack_lwe,mod_switch,objective", [
    ((25, 30, 35), 0, 1, "left"), ((25, 30, 35), 1, 1, "left"),                       # the example's two runs
    ((4, 600, 7), 0, 0, "left"), ((128, 64, 96), 1, 0, "left"), ((3, 5, 70), 1, 1, "left"), ((128, 64, 96), 0, 1, "left"),
    ((512, 512, 512), 1, 1, "left"),                                                   # BASELINE config 5 at its quoted size, packed outputs
    ((25, 30, 35), 0, 1, "right"), ((128, 64, 96), 1, 1, "right"),                     # plaintext inputs x encrypted weights
    ((25, 30, 35), 0, 1, "crossed"), ((6, 40, 9), 0, 0, "crossed")])                   # both encrypted (BGV)
def test_matmul_cpp_api(dev, dims, pack_lwe, mod_switch, objective):
    """BASELINE config 5 path, the whole flow of examples/10_bfv_matmul.cu: y = x * w + s with encrypted x through
    troy::linear::MatmulHelper (inputs and outputs through their wire formats, optional mod-switch and output packing)
    equals the plain result mod t."""
    drv = os.path.join(ROOT, "tests", "cpp", "matmul_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/matmul_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv] + [str(d) for d in dims] + ["1", str(pack_lwe), str(mod_switch), objective], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout and "mismatches 0 of %d" % (dims[0] * dims[2]) in r.stdout, r.stdout + r.stderr
    assert "fly_mismatches 0" in r.stdout                                   # matmul_fly / add_bias_inplace_fly == the held-weights forms, word for word
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    sizes = dict(zip(kv["bytes"][0::2], [int(v) for v in kv["bytes"][1::2]]))
    objs = dict(zip(kv["objects"][0::2], [int(v) for v in kv["objects"][1::2]]))
    L = 2 if mod_switch else 3
    full = 1 + 32 + 3 * 8 + 1 + 2 * L * 8192 * 8                          # one whole ciphertext on the wire (ciphertext.cu:98-139)
    if objective == "crossed":
        return                                                              # BGV: correction factor on the wire, unseeded operands
    sent = objs["weights"] if objective == "right" else objs["inputs"]
    framing = sizes["inputs"] - sent * (1 + 32 + 3 * 8 + 1 + 8 + 3 * 8192 * 8)  # seed-compressed: c0 + the 8-byte seed of c1
    assert 16 <= framing <= 8 + 8 * sent and framing % 8 == 0                  # Cipher2d framing: row count + one size per row
    if pack_lwe:
        assert sizes["outputs"] == objs["outputs"] * full
    else:
        assert sizes["outputs"] < objs["outputs"] * full                 # save_terms: c0 cut down to the result coefficients


@pytest.mark.parametrize("n,count", [(8192, 19), (4096, 3)])
def test_batched_encrypt_decrypt_cpp_api(dev, n, count):
    """encryptor.h encrypt_symmetric_batched / decryptor.h decrypt_batched: bit-identical to the per-object calls on an
    identically seeded context (the generator positions are reproduced), generators left at the same position."""
    drv = os.path.join(ROOT, "tests", "cpp", "batched_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/batched_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, str(n), str(count)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert "encrypt_mismatches 0 of" in r.stdout and "decrypt_mismatches 0 of" in r.stdout and "ntt_form_rejected 1" in r.stdout


@pytest.mark.parametrize("scheme,count", [("bfv", 6), ("ckks", 5), ("bfv", 2)])
def test_batched_ops_cpp_api(dev, scheme, count):
    """every Evaluator x_batched form equals the per-object call bit for bit (scattered operands, adjacent windows, in place,
    mixed batches, below the batching threshold)"""
    drv = os.path.join(ROOT, "tests", "cpp", "batched_ops_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/batched_ops_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme, str(count)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln and ln not in ("OK",)]
    assert len(lines) >= 26 and all(ln.endswith(" 0") or ln in ("size_mismatch_rejected 1", "u_prng_seed 1 zero_batched 1") for ln in lines), r.stdout


def test_bgv_cpp_api(dev):
    """the BGV scheme through the mirror: every evaluator result decrypts to the plain computation on the slots"""
    drv = os.path.join(ROOT, "tests", "cpp", "bgv_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/bgv_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert "form ntt=1 cf=1 L=3" in r.stdout and "serialized_cf_equal 1" in r.stdout
    checks = [ln for ln in r.stdout.splitlines() if ln.split()[-1].isdigit() and not ln.startswith(("form", "mod_switch L", "correction", "serialized"))]
    assert len(checks) >= 13 and all(ln.endswith(" 0") for ln in checks), r.stdout


@pytest.mark.parametrize("scheme,n", [("bfv", 32), ("bgv", 32), ("bfv", 4096), ("bgv", 4096)])
def test_lwe_cpp_api(dev, scheme, n):
    """the reference's test/lwe.cu through the mirror: extract / assemble, pack_lwe_ciphertexts(_batched), pack_rlwe_ciphertexts(_batched) with its
    parameter sets (N = 32, {60,40,40,60}, 20-bit t; shifts given as 2N + offset) and a larger ring, BFV and -- new in round 6 -- BGV (the packing tree keeps BGV's
    key switch on NTT-form operands as the reference does, evaluator_lwes.cu:655-657); every result decrypts to the expected polynomial"""
    drv = os.path.join(ROOT, "tests", "cpp", "lwe_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/lwe_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    checks = [ln for ln in r.stdout.splitlines() if ln.split()[-1].isdigit() and not ln.startswith("scheme")]
    assert len(checks) >= 4 + 6 + 3 + 6 and all(ln.endswith(" 0") for ln in checks), r.stdout


@pytest.mark.parametrize("scheme", ["bfv", "bgv", "ckks"])
@pytest.mark.parametrize("n", [32, 8192])
def test_special_prime_for_encryption_cpp_api(dev, scheme, n):
    """the reference's test/special_prime_for_encryption.cu:16-70: EncryptionParameters::set_use_special_prime_for_encryption(true) makes the first level the key
    level (he_context.cu:77) -- fresh ciphertexts carry all four primes of {60,40,40,60}; asymmetric and symmetric encryption decrypt to the message, and an
    addition at that level works"""
    drv = os.path.join(ROOT, "tests", "cpp", "special_prime_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/special_prime_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "first_is_key_level 1 first_limbs 4" in r.stdout and "mismatches 0" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("scheme", ["bfv", "bgv", "ckks"])
def test_every_call_uses_the_pool_it_is_given(dev, scheme):
    """the idea of the reference's test/multithread.cu:1250-1340 (SharedContextMultiPools), sharpened: the context's pool AND the global pool are set to deny
    (MemoryPool::deny, utils/memory_pool.h:100 -- added to the mirror in round 6 with the rest of the reference's pool surface), then four host threads with a pool
    each run the encoder / Encryptor / Decryptor / Evaluator / LWE / KeyGenerator calls with that pool: every result reports that pool and is correct, and no
    internal temporary falls back to another pool (it would throw) -- what makes the multi-device mode (pool i on device i) safe"""
    drv = os.path.join(ROOT, "tests", "cpp", "pools_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/pools_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "failures 0" in r.stdout, r.stdout + r.stderr


def test_multithread_cpp_api(dev):
    """test/test_multithread.cu's scenario: host threads sharing one context, keys and the global pool, each on its own
    per-thread stream; the Evaluator methods are const and re-entrant, the context generator is the only shared mutable state"""
    drv = os.path.join(ROOT, "tests", "cpp", "multithread_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/multithread_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "4", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "completed 32 wrong 0" in r.stdout and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("streams,threads", [(None, 40), ("3", 12), ("per-thread", 12), ("16", 24)])
def test_multithread_stream_sets_cpp_api(dev, streams, threads):
    """Round 6: host threads share a bounded set of streams per device (troy.cpp current_stream(): 16 while <= 17 threads use the library, 8 above;
    TROY_STREAMS=<n> | per-thread).  40 threads cross the threshold (threads move to another stream only when theirs has drained), 12 threads on 3 streams share
    each stream four ways (blocks released by one thread are reused by its stream mates at once), per-thread is the mapping of rounds 1-5: every result of the
    reference's multi-thread scenario (encrypt -> multiply -> relinearize -> rotate -> mod-switch -> decrypt) must still decrypt correctly."""
    drv = os.path.join(ROOT, "tests", "cpp", "multithread_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/multithread_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = {k: v for k, v in os.environ.items() if k not in ("TROY_STREAMS", "TROY_COMBINE")}
    if streams:
        env["TROY_STREAMS"] = streams
    r = subprocess.run([drv, str(threads), "6"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "completed %d wrong 0" % (threads * 6) in r.stdout and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_call_combining_stress_cpp_api(dev):
    """Call combining under ragged arrival: 24 host threads x 300 ops of random kind (three calls / fused / multiply alone) on two levels with
    random pauses, stream waits, late starters and early leavers; every result word-identical to the uncombined call, nothing hangs."""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/he_bench_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "stress", "300"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    assert kv["stress_wrong"][0] == "0" and int(kv["stress_ops"][0]) >= 24 * 100 and int(kv["stress_combined_calls"][0]) > 0, r.stdout


def test_fused_method_on_wide_chain_cpp_api(dev):
    """Evaluator::multiply_relinearize_rescale{_new,_inplace,_batched} on the usual CKKS chain {60,50,50,50,50,60} (N = 16384): word-identical to the three calls
    (single objects, a batch of 16, mixed levels), decrypts to the slot-wise product, the reference's error behaviour, and the same under call combining"""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/he_bench_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "check60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    for key in ("fused_single_identical", "fused_inplace_identical", "fused_batched_identical", "fused_mixed_levels_identical", "combined_identical"):
        assert kv[key][0] == "1", r.stdout
    assert float(kv["fused_single_error"][0]) < 1e-4 and kv["fused_errors"][0] == "2", r.stdout


def test_pool_high_water_mark_cpp_api(dev):
    """MemoryPool: a block released by another LIVE host thread is not reused while fresh memory is available (no device-wide wait on the N-thread path);
    above the high-water mark (set_high_water_bytes / TROY_POOL_HIGH_WATER_MB) the pool synchronises once and reuses instead of growing"""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/he_bench_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    # the policy under test is the DEFAULT mapping of host threads to streams: with call combining or a one-stream set every thread shares a stream and reuse is immediate
    env = {k: v for k, v in os.environ.items() if k not in ("TROY_COMBINE", "TROY_STREAMS", "TROY_POOL_HIGH_WATER_MB")}
    r = subprocess.run([drv, "pool"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    assert kv["pool_uncapped_held_MB"][0] == "96" and kv["pool_capped_held_MB"][0] == "48" and kv["pool_concurrent_allocate_clashes"][0] == "0", r.stdout


def test_multi_device_mode_cpp_api(dev):
    """The reference tool's `-c N -mp -md` mode (test/bench/he_operations.cu:33-34, :139-147; test/test_multithread.cu:18-37; readme.md:179-202): thread i
    works in MemoryPool::create(i % device_count()), one context per device moved there with to_device_inplace(pool), every KeyGenerator built from the
    same secret key.  On a one-GPU box: 6 pools on device 0.  All contexts share the secret key; on every pool the fused call equals the three calls,
    decrypts to the product through ANOTHER context's decryptor, and the *_batched results equal the single-object ones; per-device rates are reported."""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/he_bench_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "devices", "6", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    assert kv["pools"][0] == "6" and kv["devices_same_secret_key"][0] == "1" and kv["devices_identical"][0] == "1", r.stdout
    assert int(kv["contexts"][0]) == min(int(kv["devices"][0]), 6)
    for name in ("single_three_calls", "single_fused", "batch64_three_calls", "batch64_fused"):
        total = float(kv["devices_%s_ops_per_s" % name][0])
        per = sum(float(kv["devices_%s_device%d_ops_per_s" % (name, d)][0]) for d in range(int(kv["devices"][0])))
        assert total > 0 and abs(total - per) <= 1e-6 * total + 1.0, (name, total, per)


def test_multithread_call_combining_cpp_api(dev):
    """The same program (BFV: encrypt -> multiply -> relinearize -> add -> mod-switch -> decrypt on 8 host threads) with TROY_COMBINE=1:
    one shared stream, the multiply / relinearize calls of concurrent threads run as batches (troy.h "Call combining"), the other calls
    are queued between them; every result still decrypts correctly and calls were in fact combined."""
    drv = os.path.join(ROOT, "tests", "cpp", "multithread_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/multithread_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    env = dict(os.environ, TROY_COMBINE="1", TROY_COMBINE_WINDOW_US="2000")
    r = subprocess.run([drv, "8", "8"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "completed 64 wrong 0" in r.stdout and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    assert kv["combining"][0] == "1" and int(kv["combining"][2]) > 0, r.stdout


def test_basics_cpp_api(dev):
    """examples/1_bfv_basics.cu and 3_levels.cu in spirit: SEAL's default chain, qualifiers, the invariant noise budget shrinking
    along a computation and with the modulus, until it reaches zero and decryption stops being correct"""
    drv = os.path.join(ROOT, "tests", "cpp", "basics_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/basics_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert "chain 3 primes, key level 109 bits, first level 72 bits" in r.stdout          # bfv_default(4096): 36 + 36 + 37 bits
    assert "qualifiers batching 0 fast_plain_lift 1 descending 1 security 1" in r.stdout


@pytest.mark.parametrize("dims,pack_lwe,mod_switch", [((25, 30, 35), 0, 1), ((25, 30, 35), 1, 1), ((64, 48, 40), 1, 0), ((7, 100, 9), 0, 0)])
def test_ckks_matmul_cpp_api(dev, dims, pack_lwe, mod_switch):
    """examples/11_ckks_matmul.cu: y = x * w + s on real matrices through MatmulHelper and the CKKS encoder (its two configurations
    and two more), every output within the scale-2^20 tolerance of the plain result"""
    drv = os.path.join(ROOT, "tests", "cpp", "ckks_matmul_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/ckks_matmul_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv] + [str(d) for d in dims] + [str(pack_lwe), str(mod_switch)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


@pytest.mark.parametrize("shape,mod_switch,objective", [((2, 3, 5, 15, 15, 3, 3), 0, "left"),          # the example's run
                                                        ((2, 3, 5, 15, 15, 3, 3), 1, "left"), ((1, 4, 6, 40, 37, 5, 3), 1, "left"),
                                                        ((3, 2, 2, 9, 9, 1, 1), 0, "left"), ((2, 3, 5, 15, 15, 3, 3), 1, "right"),
                                                        ((4, 16, 8, 32, 32, 3, 3), 1, "left"), ((1, 1, 2, 100, 120, 3, 3), 0, "left"), ((2, 2, 1, 70, 130, 4, 2), 1, "left")])  # several overlapping tiles
def test_conv2d_cpp_api(dev, shape, mod_switch, objective):
    """examples/14_bfv_conv2d.cu: y = conv2d(x, w) + s with encrypted images (or encrypted kernels) through Conv2dHelper, tiles that
    overlap by the kernel size, the outputs' partial wire format; equals the plain cross-correlation mod t"""
    drv = os.path.join(ROOT, "tests", "cpp", "conv2d_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/conv2d_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv] + [str(v) for v in shape] + [str(mod_switch), objective], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "mismatches 0 of" in r.stdout, r.stdout + r.stderr


def test_ring2k_cpp_api(dev):
    """examples/13_ring2k.cu through PolynomialEncoderRing2k<uint32_t / uint64_t / unsigned __int128>: products in Z_{2^k}, also after a
    modulus switch"""
    drv = os.path.join(ROOT, "tests", "cpp", "ring2k_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/ring2k_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert r.stdout.count("example 1 random_mismatches 0 after_mod_switch 0") == 5 and "narrow_k_rejected 1" in r.stdout
    # y = x * w + s over Z_{2^k} through MatmulHelper's ring-2^k forms, packed and not
    assert r.stdout.count("mismatches 0 of 153") == 4, r.stdout


def test_conv2d_ckks_ring2k_cpp_api(dev):
    """Conv2dHelper beyond BFV mod t: the CKKS forms (examples/15_ckks_conv2d.cu, both objectives) and the ring-2^k forms through
    PolynomialEncoderRing2k<uint64_t / uint32_t / unsigned __int128>, with the wire formats in between"""
    drv = os.path.join(ROOT, "tests", "cpp", "conv2d_ext_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/conv2d_ext_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert r.stdout.count("mismatches 0 of 800") == 3 and "ckks max_error" in r.stdout


def test_plain_ops_cpp_api(dev):
    """plaintext-side API: BatchEncoder::scale_up / scale_down / centralize / decentralize with partial RNS plaintexts as operands of
    encrypt / add_plain / multiply_plain, Evaluator::apply_galois_plain against the rotation under encryption (BFV, BGV, CKKS), the
    integer and single-value CKKS encodings, is_transparent"""
    drv = os.path.join(ROOT, "tests", "cpp", "plain_ops_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/plain_ops_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "FAIL" not in r.stdout, r.stdout + r.stderr
    assert r.stdout.count(" ok\n") >= 54, r.stdout


@pytest.mark.parametrize("scheme,log_t,bits", [
    ("bfv", 20, [40, 40, 40]), ("bfv", 35, [30, 30, 30, 30]), ("bfv", 20, [60, 40, 40, 60]), ("bfv", 35, [60, 30, 30, 60]),
    ("bgv", 20, [40, 40, 40]), ("bgv", 35, [30, 30, 30, 30]), ("bgv", 20, [60, 40, 40, 60]),
    ("ckks", 0, [60, 60, 60]), ("ckks", 0, [40, 40, 40]), ("ckks", 0, [60, 40, 40, 60])])
def test_decrypt_and_compare_matrix(dev, scheme, log_t, bits):
    """The reference's own device-test matrix (test/evaluator.cu:276-690, :1115-1233: N = 32, these coefficient moduli, seed 0x123,
    CKKS at scale 2^20 with tolerance 1e-2) through the C++ mirror: encode -> encrypt -> multiply / square / relinearize /
    key-switch to another key / mod-switch / rescale / rotate / conjugate -> decrypt -> decode equals the plain computation.
    An oracle-independent pin: a wrong key switch, tensor product or rescale does not decrypt to the right message."""
    drv = os.path.join(ROOT, "tests", "cpp", "semantics_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/semantics_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme, str(log_t)] + [str(b) for b in bits], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and "FAIL" not in r.stdout, r.stdout + r.stderr
    names = [ln.split()[0] for ln in r.stdout.splitlines() if " pass " in ln]
    assert "multiply" in names and "relinearize" in names and "keyswitching" in names


@pytest.mark.parametrize("n", [32, 2048])
def test_operand_forms_cpp_api(dev, n):
    """the reference's evaluator scenarios whose operands are in the non-default representation (test/evaluator.cu: test_add_subtract_ntt / _intt,
    test_add_plain_scaled(_ntt), test_multiply_plain_ntt, test_multiply_plain_centralized, test_transform_plain_ntt, test_mod_switch_plain_to_next)
    through the mirror, BFV / BGV / CKKS, full and partial (coeff_count = N / 3) plaintexts; every result is decrypted"""
    drv = os.path.join(ROOT, "tests", "cpp", "forms_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/forms_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.endswith(" ok")]
    assert len(lines) >= 50 and "FAIL" not in r.stdout, r.stdout


@pytest.mark.parametrize("scheme,n", [("bfv", 32), ("bgv", 32), ("ckks", 32), ("bfv", 8192), ("bgv", 4096), ("ckks", 8192)])
def test_serialize_cpp_api(dev, scheme, n):
    """the reference's test/serialize.cu through the mirror: EncryptionParameters, Plaintext, Ciphertext (plain, seeded, 3 polynomials, terms), SecretKey,
    PublicKey, KSwitchKeys, RelinKeys, GaloisKeys -- save, bytes written == serialized_size_upperbound, T::load_new, and then the loaded object is used
    (decrypt / apply_keyswitching / relinearize / rotate); its parameter sets (N = 32, {60,40,40,60}) and production-size rings"""
    drv = os.path.join(ROOT, "tests", "cpp", "serialize_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/serialize_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, scheme, str(n)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert sum(ln.endswith(" ok") for ln in r.stdout.splitlines()) >= 28 and "FAIL" not in r.stdout, r.stdout


@pytest.mark.parametrize("args", [("bfv", "32"), ("bgv", "32"), ("ckks", "32"), ("bfv", "8192"), ("bgv", "4096"), ("ckks", "8192"), ("budget",)])
def test_encryptor_cpp_api(dev, args):
    """the reference's test/encryptor.cu and test/encryptor_batched.cu through the mirror: encrypt_zero at two levels, full / partial SIMD messages, BFV scale_up /
    scale_down around the ciphertext, equal u_prng => equal c1, for one ciphertext and for batches of 16; test_invariant_noise_budget (BFV and BGV: 30..40 bits fresh,
    <= 10 after one square, 0 after two)"""
    drv = os.path.join(ROOT, "tests", "cpp", "encryptor_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/encryptor_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert sum(ln.endswith(" ok") for ln in r.stdout.splitlines()) >= (2 if args[0] == "budget" else 19) and "FAIL" not in r.stdout, r.stdout


def test_context_cpp_api(dev):
    """the reference's test/he_context.cu through the mirror (the host part also runs in the CPU suite, tests/test_mirror_host.py) plus HeContextToDevice: the N = 4
    chains of the three schemes move to the device"""
    drv = os.path.join(ROOT, "tests", "cpp", "context_driver")
    if not os.path.exists(drv):
        pytest.fail("tests/cpp/context_driver is not built (python -c 'import __graft_entry__ as g; g.build()')")
    r = subprocess.run([drv, "device"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert sum(ln.endswith(" ok") for ln in r.stdout.splitlines()) >= 39 and "FAIL" not in r.stdout, r.stdout
