// C++ driver for tests/test_gpu_cpp_api.py::test_serialize_cpp_api: the reference's test/serialize.cu replayed through the mirror -- every
// serializable object (EncryptionParameters, Plaintext, Ciphertext incl. seeded and 3-polynomial, SecretKey, PublicKey, KSwitchKeys, RelinKeys,
// GaloisKeys, Ciphertext terms) goes through  save -> size == serialized_size_upperbound -> T::load_new  and is then USED (decrypt, key switch,
// relinearize, rotate), with the reference's parameter sets (N = 32, {60,40,40,60}) and a production-size ring.
//   serialize_driver <bfv|bgv|ckks> <N>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%-70s %s\n", what, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

static bool sizes_ok = true;
static CompressionMode wire_mode = CompressionMode::Nil;      // the second pass of main() repeats everything with Zstd (test/serialize_zstd.cu)
static bool size_ok(size_t written, size_t stream_bytes, size_t bound) {
    // Nil: the bound IS the size (serialize.cu:17); Zstd: an upper bound (serialize_zstd.cu:17)
    return written == stream_bytes && (wire_mode == CompressionMode::Nil ? bound == stream_bytes : bound >= stream_bytes);
}
template <typename T> static void reserialize(T& t) {                                     // serialize.cu:14-20
    std::stringstream ss;
    const size_t written = t.save(ss, wire_mode);
    sizes_ok = sizes_ok && size_ok(written, ss.str().size(), t.serialized_size_upperbound(wire_mode));
    t = T::load_new(ss);
}
template <typename T> static void reserialize(T& t, HeContextPointer context) {           // serialize.cu:22-28
    std::stringstream ss;
    const size_t written = t.save(ss, context, wire_mode);
    sizes_ok = sizes_ok && size_ok(written, ss.str().size(), t.serialized_size_upperbound(context, wire_mode));
    t = T::load_new(ss, context);
}
static void reserialize_terms(Ciphertext& t, HeContextPointer context, const std::vector<size_t>& terms) {   // serialize.cu:373-378
    std::stringstream ss;
    t.save_terms(ss, context, terms, MemoryPool::GlobalPool(), wire_mode);
    sizes_ok = sizes_ok && size_ok(ss.str().size(), ss.str().size(), t.serialized_terms_size_upperbound(context, terms, wire_mode));
    t = Ciphertext::load_terms_new(ss, context, terms);
}

struct Fixture {
    SchemeType scheme;
    size_t n;
    uint64_t t = 0;
    double scale = 0, tol = 0;
    HeContextPointer context;
    std::unique_ptr<BatchEncoder> batch;
    std::unique_ptr<CKKSEncoder> ckks;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<Encryptor> encryptor;
    std::unique_ptr<Decryptor> decryptor;
    std::unique_ptr<Evaluator> evaluator;
    std::mt19937_64 gen{17};
    using Vec = std::vector<double>;             // integer messages are held as doubles too (t < 2^53)

    Fixture(SchemeType s, size_t n_) : scheme(s), n(n_) {
        EncryptionParameters parms(scheme);
        parms.set_poly_modulus_degree(n);
        parms.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        if (scheme != SchemeType::CKKS) { parms.set_plain_modulus(PlainModulus::batching(n, 20)); t = parms.plain_modulus().value(); }
        else { scale = double(1ull << (n > 64 ? 36 : 20)); tol = 1e-2; }      // the reference's N = 32 set uses 2^20; larger rings need the noise further below the tolerance
        {   // serialize.cu:31-58
            const ParmsID id = parms.parms_id();
            EncryptionParameters copy = parms;
            {   // raw fields, no compression header (encryption_parameters.cu:53-112)
                std::stringstream ss;
                const size_t written = copy.save(ss);
                sizes_ok = sizes_ok && copy.serialized_size_upperbound() == ss.str().size() && written == ss.str().size();
                copy = EncryptionParameters::load_new(ss);
            }
            check(sizes_ok && copy.parms_id() == id && copy.poly_modulus_degree() == n && copy.coeff_modulus().size() == 4, "EncryptionParameters: save / load_new keeps parms_id");
        }
        context = HeContext::create(parms, true, SecurityLevel::Nil, 0x123);
        context->to_device_inplace();
        if (scheme == SchemeType::CKKS) ckks = std::make_unique<CKKSEncoder>(context); else batch = std::make_unique<BatchEncoder>(context);
        keygen = std::make_unique<KeyGenerator>(context);
        encryptor = std::make_unique<Encryptor>(context);
        encryptor->set_public_key(keygen->create_public_key(false));
        encryptor->set_secret_key(keygen->secret_key());
        decryptor = std::make_unique<Decryptor>(context, keygen->secret_key());
        evaluator = std::make_unique<Evaluator>(context);
    }
    size_t slots() const { return scheme == SchemeType::CKKS ? n / 2 : n; }
    Vec random_simd() {
        Vec v(slots());
        if (scheme == SchemeType::CKKS) { std::uniform_real_distribution<double> U(-10, 10); for (auto& x : v) x = U(gen); }
        else for (auto& x : v) x = double(gen() % t);
        return v;
    }
    Vec random_poly() { Vec v = random_simd(); v.resize(n); if (scheme == SchemeType::CKKS) { std::uniform_real_distribution<double> U(-10, 10); for (auto& x : v) x = U(gen); } return v; }
    Plaintext encode_simd(const Vec& v) {
        if (scheme == SchemeType::CKKS) { std::vector<std::complex<double>> c(v.size()); for (size_t i = 0; i < v.size(); i++) c[i] = {v[i], 0.0}; return ckks->encode_complex64_simd_new(c, std::nullopt, scale); }
        std::vector<uint64_t> u(v.begin(), v.end());
        return batch->encode_new(u);
    }
    Vec decode_simd(const Plaintext& p) {
        if (scheme == SchemeType::CKKS) { auto c = ckks->decode_complex64_simd_new(p); Vec v(c.size()); for (size_t i = 0; i < c.size(); i++) v[i] = c[i].real(); return v; }
        auto u = batch->decode_new(p);
        return Vec(u.begin(), u.end());
    }
    Plaintext encode_poly(const Vec& v) {
        if (scheme == SchemeType::CKKS) return ckks->encode_float64_polynomial_new(v, std::nullopt, scale);
        std::vector<uint64_t> u(v.begin(), v.end());
        return batch->encode_polynomial_new(u);
    }
    Vec decode_poly(const Plaintext& p) {
        if (scheme == SchemeType::CKKS) return ckks->decode_float64_polynomial_new(p);
        auto u = batch->decode_polynomial_new(p);
        return Vec(u.begin(), u.end());
    }
    bool near(const Vec& a, const Vec& b, double tolerance) const {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); i++) if (!(std::abs(a[i] - b[i]) <= tolerance)) return false;
        return true;
    }
    bool same(const Vec& a, const Vec& b) const { return near(a, b, scheme == SchemeType::CKKS ? tol : 0.0); }
    Vec mul(const Vec& a, const Vec& b) const {
        Vec r(a.size());
        for (size_t i = 0; i < a.size(); i++) {
            if (scheme == SchemeType::CKKS) r[i] = a[i] * b[i];
            else r[i] = double((unsigned __int128)uint64_t(a[i]) * uint64_t(b[i]) % t);
        }
        return r;
    }
    Vec rotate1(const Vec& a) const {                                                         // GeneralVector::rotate(1): rows for BFV / BGV, the whole vector for CKKS
        Vec r(a.size());
        if (scheme == SchemeType::CKKS) { for (size_t i = 0; i < a.size(); i++) r[i] = a[(i + 1) % a.size()]; return r; }
        const size_t h = a.size() / 2;
        for (size_t i = 0; i < h; i++) { r[i] = a[(i + 1) % h]; r[h + i] = a[h + (i + 1) % h]; }
        return r;
    }
};

// CKKS messages are real slot values at scale 2^20 (the reference's set: tolerance 1e-2; products of two get a looser bound)
static void run(SchemeType scheme, size_t n) {
    Fixture f(scheme, n);
    const bool ckks = scheme == SchemeType::CKKS;
    HeContextPointer he = f.context;
    {   // test_plaintext (serialize.cu:60-68)
        const auto m = f.random_simd();
        Plaintext e = f.encode_simd(m);
        const std::vector<uint64_t> before = e.data().to_vector();
        sizes_ok = true;
        reserialize(e);
        check(sizes_ok, "Plaintext: bytes written == serialized_size_upperbound");
        check(e.data().to_vector() == before && f.same(f.decode_simd(e), m), "Plaintext: load_new(save(p)) decodes to the message");
    }
    {   // test_ciphertext (serialize.cu:101-141)
        const auto m = f.random_simd();
        Plaintext e = f.encode_simd(m);
        Ciphertext c = f.encryptor->encrypt_symmetric_new(e, true);
        Ciphertext cl = c.clone();
        check(cl.contains_seed(), "encrypt_symmetric(save_seed) keeps the seed");
        cl.expand_seed(he);
        check(!cl.contains_seed() && f.same(f.decode_simd(f.decryptor->decrypt_new(cl)), m), "expand_seed, then decrypt");
        sizes_ok = true;
        c = f.encryptor->encrypt_asymmetric_new(e);
        reserialize(c, he);
        check(f.same(f.decode_simd(f.decryptor->decrypt_new(c)), m), "Ciphertext (asymmetric): reserialize, decrypt");
        c = f.encryptor->encrypt_symmetric_new(e, false);
        reserialize(c, he);
        check(f.same(f.decode_simd(f.decryptor->decrypt_new(c)), m), "Ciphertext (symmetric): reserialize, decrypt");
        c = f.encryptor->encrypt_symmetric_new(e, true);
        {
            std::stringstream a, b;
            c.save(a, he);
            Ciphertext full = c.clone();
            full.expand_seed(he);
            full.save(b, he);
            const size_t poly = c.poly_modulus_degree() * c.coeff_modulus_size() * 8;
            check(b.str().size() - a.str().size() == poly - 8, "a seeded ciphertext is one polynomial (minus the seed) shorter on the wire");
        }
        reserialize(c, he);
        check(!c.contains_seed() && f.same(f.decode_simd(f.decryptor->decrypt_new(c)), m), "Ciphertext (seeded): reserialize expands the seed, decrypt");
        Ciphertext sq = f.evaluator->square_new(c);
        reserialize(sq, he);
        {
            const auto got = f.decode_simd(f.decryptor->decrypt_new(sq)), want = f.mul(m, m);
            double worst = 0;
            for (size_t i = 0; i < want.size() && i < got.size(); i++) worst = std::max(worst, std::abs(got[i] - want[i]));
            if (ckks) std::printf("   squared ciphertext: scale %.3g, largest error %.3g\n", sq.scale(), worst);
            check(sq.polynomial_count() == 3 && f.near(got, want, ckks ? f.tol : 0.0), "Ciphertext with 3 polynomials: reserialize, decrypt (the reference's tolerance)");
        }
        check(sizes_ok, "Ciphertext: bytes written == serialized_size_upperbound (4 cases)");
    }
    {   // test_secret_public_key (serialize.cu:174-208)
        const auto m = f.random_simd();
        Plaintext e = f.encode_simd(m);
        sizes_ok = true;
        SecretKey sk = f.keygen->secret_key();
        reserialize(sk);
        Encryptor enc(he);
        enc.set_secret_key(sk);
        check(f.same(f.decode_simd(f.decryptor->decrypt_new(enc.encrypt_symmetric_new(e, false))), m), "SecretKey: reserialize, encrypt with it, decrypt with the original");
        PublicKey pk = f.keygen->create_public_key(false);
        reserialize(pk, he);
        enc.set_public_key(pk);
        check(f.same(f.decode_simd(f.decryptor->decrypt_new(enc.encrypt_asymmetric_new(e))), m), "PublicKey: reserialize, encrypt, decrypt");
        pk = f.keygen->create_public_key(true);
        check(pk.contains_seed(), "create_public_key(save_seed) keeps the seed");
        reserialize(pk, he);
        enc.set_public_key(pk);
        check(!pk.contains_seed() && f.same(f.decode_simd(f.decryptor->decrypt_new(enc.encrypt_asymmetric_new(e))), m), "PublicKey (seeded): reserialize, encrypt, decrypt");
        check(sizes_ok, "SecretKey / PublicKey: bytes written == serialized_size_upperbound");
    }
    {   // test_kswitch_keys (serialize.cu:241-340)
        sizes_ok = true;
        KeyGenerator other(he);
        Encryptor enc_other(he);
        enc_other.set_secret_key(other.secret_key());
        for (bool seed : {false, true}) {
            KSwitchKeys ksk = f.keygen->create_keyswitching_key(other.secret_key(), seed);
            reserialize(ksk, he);
            const auto m = f.random_simd();
            Ciphertext c = enc_other.encrypt_symmetric_new(f.encode_simd(m), false);
            Ciphertext sw = f.evaluator->apply_keyswitching_new(c, ksk);
            check(f.same(f.decode_simd(f.decryptor->decrypt_new(sw)), m), seed ? "KSwitchKeys (seeded): reserialize, apply_keyswitching" : "KSwitchKeys: reserialize, apply_keyswitching");
        }
        for (bool seed : {false, true}) {
            RelinKeys rk = f.keygen->create_relin_keys(seed);
            reserialize(rk, he);
            const auto m1 = f.random_simd(), m2 = f.random_simd();
            Ciphertext prod = f.evaluator->multiply_new(f.encryptor->encrypt_asymmetric_new(f.encode_simd(m1)), f.encryptor->encrypt_asymmetric_new(f.encode_simd(m2)));
            Ciphertext rel = f.evaluator->relinearize_new(prod, rk);
                check(rel.polynomial_count() == 2 && f.near(f.decode_simd(f.decryptor->decrypt_new(rel)), f.mul(m1, m2), ckks ? 0.5 : 0.0), seed ? "RelinKeys (seeded): reserialize, multiply + relinearize" : "RelinKeys: reserialize, multiply + relinearize");
            }
        for (bool seed : {false, true}) {
            GaloisKeys gk = f.keygen->create_galois_keys(seed);
            reserialize(gk, he);
            const auto m = f.random_simd();
            Ciphertext c = f.encryptor->encrypt_asymmetric_new(f.encode_simd(m));
            Ciphertext rot = ckks ? f.evaluator->rotate_vector_new(c, 1, gk) : f.evaluator->rotate_rows_new(c, 1, gk);
            check(f.same(f.decode_simd(f.decryptor->decrypt_new(rot)), f.rotate1(m)), seed ? "GaloisKeys (seeded): reserialize, rotate by 1" : "GaloisKeys: reserialize, rotate by 1");
        }
        check(sizes_ok, "KSwitchKeys / RelinKeys / GaloisKeys: bytes written == serialized_size_upperbound");
    }
    {   // test_ciphertext_terms (serialize.cu:380-424)
        sizes_ok = true;
        const std::vector<size_t> terms = {1, 3, 5, 7};
        auto terms_equal = [&](const Fixture::Vec& a, const Fixture::Vec& b, double tolerance) {
            for (size_t term : terms) if (!(std::abs(a[term] - b[term]) <= tolerance)) return false;
            return true;
        };
        for (bool seed : {false, true}) {
            const auto m = f.random_poly();
            Ciphertext c = f.encryptor->encrypt_symmetric_new(f.encode_poly(m), seed);
            reserialize_terms(c, he, terms);
            check(terms_equal(f.decode_poly(f.decryptor->decrypt_new(c)), m, ckks ? f.tol : 0.0), seed ? "Ciphertext terms (seeded): the saved terms decrypt" : "Ciphertext terms: the saved terms decrypt");
        }
        const auto m = f.random_poly();
        Ciphertext c = f.encryptor->encrypt_asymmetric_new(f.encode_poly(m));
        Ciphertext prod = f.evaluator->multiply_new(c, c);
        const auto truth = f.decode_poly(f.decryptor->decrypt_new(prod));
        reserialize_terms(prod, he, terms);
        check(terms_equal(f.decode_poly(f.decryptor->decrypt_new(prod)), truth, ckks ? 0.5 : 0.0), "Ciphertext terms of a 3-polynomial product");
        check(sizes_ok, "Ciphertext terms: bytes written == serialized_terms_size_upperbound");
        bool threw = false;
        {
            std::stringstream ss;
            c.save(ss, he);
            try { Ciphertext::load_terms_new(ss, he, terms); } catch (const std::logic_error&) { threw = true; }
        }
        check(threw, "load_terms of a full ciphertext is rejected");
        threw = false;
        {
            std::stringstream ss;
            c.save_terms(ss, he, terms);
            try { Ciphertext::load_new(ss, he); } catch (const std::logic_error&) { threw = true; }
        }
        check(threw, "load of a terms-only ciphertext is rejected");
    }
}

int main(int argc, char** argv) {
    try {
        const char* s = argc > 1 ? argv[1] : "bfv";
        const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 32;
        const SchemeType scheme = !std::strcmp(s, "ckks") ? SchemeType::CKKS : !std::strcmp(s, "bgv") ? SchemeType::BGV : SchemeType::BFV;
        std::printf("scheme %s N %zu\n", s, n);
        run(scheme, n);
        if (utils::compression::available(CompressionMode::Zstd)) {      // test/serialize_zstd.cu: the same scenarios with every object compressed
            std::printf("-- Zstd\n");
            wire_mode = CompressionMode::Zstd;
            run(scheme, n);
            // ciphertexts of uniform residues do not compress (they are written raw, mode byte Nil); structured data does
            Fixture f(scheme, n);
            Plaintext zeros = f.encode_simd(Fixture::Vec(f.slots(), 0.0));
            std::stringstream a, b;
            zeros.save(a, CompressionMode::Nil); zeros.save(b, CompressionMode::Zstd);
            check(b.str().size() < a.str().size() / 4 && Plaintext::load_new(b).data().to_vector() == zeros.data().to_vector(), "an all-zero plaintext shrinks under Zstd and comes back");
        } else std::printf("(no zstd runtime library: the Zstd pass is skipped)\n");
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
