// C++ driver for tests/test_gpu_cpp_api.py::test_encryptor_cpp_api: the reference's test/encryptor.cu and test/encryptor_batched.cu replayed
// through the mirror -- encrypt_zero (symmetric / asymmetric, first and second level), full and 30 %-filled SIMD messages, BFV scale_up before /
// scale_down after, the same u_prng giving the same c1, each for one ciphertext and for a batch of 16; plus test_invariant_noise_budget.
//   encryptor_driver <bfv|bgv|ckks> <N>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%-72s %s\n", what, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

struct Suite {
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
    std::mt19937_64 gen{29};
    using Vec = std::vector<double>;

    Suite(SchemeType s, size_t n_) : scheme(s), n(n_) {
        EncryptionParameters parms(scheme);
        parms.set_poly_modulus_degree(n);
        parms.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        if (scheme != SchemeType::CKKS) { parms.set_plain_modulus(PlainModulus::batching(n, 20)); t = parms.plain_modulus().value(); }
        else { scale = double(1ull << (n > 64 ? 36 : 20)); tol = 1e-2; }
        context = HeContext::create(parms, true, SecurityLevel::Nil, 0x123);
        context->to_device_inplace();
        if (scheme == SchemeType::CKKS) ckks = std::make_unique<CKKSEncoder>(context); else batch = std::make_unique<BatchEncoder>(context);
        keygen = std::make_unique<KeyGenerator>(context);
        encryptor = std::make_unique<Encryptor>(context);
        encryptor->set_public_key(keygen->create_public_key(false));
        encryptor->set_secret_key(keygen->secret_key());
        decryptor = std::make_unique<Decryptor>(context, keygen->secret_key());
    }
    double zero_scale() const { return n > 64 ? scale : 1e6; }   // the reference reads its N = 32 zero encryptions at 1e6; the noise of a large ring needs a larger scale to stay below the tolerance
    size_t slots() const { return scheme == SchemeType::CKKS ? n / 2 : n; }
    Vec random_simd(size_t count) {
        Vec v(count);
        if (scheme == SchemeType::CKKS) { std::uniform_real_distribution<double> U(-10, 10); for (auto& x : v) x = U(gen); }
        else for (auto& x : v) x = double(gen() % t);
        return v;
    }
    Plaintext encode(const Vec& v) {
        if (scheme == SchemeType::CKKS) { std::vector<std::complex<double>> c(v.size()); for (size_t i = 0; i < v.size(); i++) c[i] = {v[i], 0.0}; return ckks->encode_complex64_simd_new(c, std::nullopt, scale); }
        return batch->encode_new(std::vector<uint64_t>(v.begin(), v.end()));
    }
    Vec decode(const Plaintext& p) {
        if (scheme == SchemeType::CKKS) { auto c = ckks->decode_complex64_simd_new(p); Vec v(c.size()); for (size_t i = 0; i < c.size(); i++) v[i] = std::abs(c[i].imag()) <= tol ? c[i].real() : 1e300; return v; }
        auto u = batch->decode_new(p);
        return Vec(u.begin(), u.end());
    }
    bool near(const Vec& a, Vec b) const {
        b.resize(a.size(), 0.0);
        for (size_t i = 0; i < a.size(); i++) if (!(std::abs(a[i] - b[i]) <= tol)) return false;
        return true;
    }
};

static bool same_words(const Ciphertext& a, const Ciphertext& b, size_t poly) { return a.poly(poly).to_vector() == b.poly(poly).to_vector(); }

static void run_single(Suite& f) {
    const bool ckks = f.scheme == SchemeType::CKKS;
    const ParmsID first = f.context->first_parms_id();
    const Suite::Vec zeros(f.slots(), 0.0);
    auto dec = [&](Ciphertext c) { if (ckks && c.scale() == 1.0) c.scale() = f.zero_scale(); return f.decode(f.decryptor->decrypt_new(c)); };
    Ciphertext c = f.encryptor->encrypt_zero_symmetric_new(false);
    check(c.parms_id() == first && f.near(dec(c), zeros), "encrypt_zero_symmetric: first level, decrypts to zeros");
    c = f.encryptor->encrypt_zero_asymmetric_new();
    check(c.parms_id() == first && f.near(dec(c), zeros), "encrypt_zero_asymmetric: first level, decrypts to zeros");
    const ParmsID second = f.context->first_context_data().value()->next_context_data().value()->parms_id();
    c = f.encryptor->encrypt_zero_asymmetric_new(second);
    check(c.parms_id() == second && f.near(dec(c), zeros), "encrypt_zero_asymmetric(second level)");
    c = f.encryptor->encrypt_zero_symmetric_new(false, second);
    check(c.parms_id() == second && f.near(dec(c), zeros), "encrypt_zero_symmetric(second level)");
    auto m = f.random_simd(f.slots());
    check(f.near(f.decode(f.decryptor->decrypt_new(f.encryptor->encrypt_symmetric_new(f.encode(m), false))), m), "all slots, symmetric");
    if (f.scheme == SchemeType::BFV) {
        Plaintext p = f.batch->scale_up_new(f.encode(m), std::nullopt);
        check(f.near(f.decode(f.decryptor->decrypt_new(f.encryptor->encrypt_symmetric_new(p, false))), m), "BFV: scale_up before encrypt");
        Plaintext d = f.decryptor->bfv_decrypt_without_scaling_down_new(f.encryptor->encrypt_symmetric_new(f.encode(m), false));
        f.batch->scale_down_inplace(d);
        check(f.near(f.decode(d), m), "BFV: bfv_decrypt_without_scaling_down, then scale_down");
    }
    check(f.near(f.decode(f.decryptor->decrypt_new(f.encryptor->encrypt_asymmetric_new(f.encode(m)))), m), "all slots, asymmetric");
    const size_t used = size_t(int(f.slots() * 0.3));
    m = f.random_simd(used);
    check(f.near(f.decode(f.decryptor->decrypt_new(f.encryptor->encrypt_asymmetric_new(f.encode(m)))), m), "30 % of the slots, asymmetric");
    m = f.random_simd(used);
    Plaintext plain = f.encode(m);
    check(f.near(f.decode(f.decryptor->decrypt_new(f.encryptor->encrypt_symmetric_new(plain, false))), m), "30 % of the slots, symmetric");
    utils::RandomGenerator rng1(0x1234), rng2(0x1234), rng3(0x1235);
    Ciphertext c1 = f.encryptor->encrypt_symmetric_new(plain, false, &rng1), c2 = f.encryptor->encrypt_symmetric_new(plain, false, &rng2), c3 = f.encryptor->encrypt_symmetric_new(plain, false, &rng3);
    check(same_words(c1, c2, 1) && !same_words(c1, c3, 1) && !same_words(c1, c2, 0), "the same u_prng gives the same c1 (and another seed another c1)");
}

static void run_batched(Suite& f) {
    const size_t B = 16;
    const bool ckks = f.scheme == SchemeType::CKKS;
    const ParmsID first = f.context->first_parms_id();
    const Suite::Vec zeros(f.slots(), 0.0);
    auto all_level = [](const std::vector<Ciphertext>& v, const ParmsID& id) { for (const auto& c : v) if (!(c.parms_id() == id)) return false; return true; };
    auto dec_all = [&](std::vector<Ciphertext>& v, const std::vector<Suite::Vec>& want) {
        if (ckks) for (auto& c : v) if (c.scale() == 1.0) c.scale() = f.zero_scale();
        std::vector<Plaintext> d = f.decryptor->decrypt_batched_new(batch_utils::collect_const_pointer(v));
        if (d.size() != want.size()) return false;
        for (size_t i = 0; i < d.size(); i++) if (!f.near(f.decode(d[i]), want[i])) return false;
        return true;
    };
    const std::vector<Suite::Vec> zs(B, zeros);
    std::vector<Ciphertext> c = f.encryptor->encrypt_zero_symmetric_new_batched(B, false);
    check(c.size() == B && all_level(c, first) && dec_all(c, zs), "batched encrypt_zero_symmetric");
    c = f.encryptor->encrypt_zero_asymmetric_new_batched(B);
    check(all_level(c, first) && dec_all(c, zs), "batched encrypt_zero_asymmetric");
    const ParmsID second = f.context->first_context_data().value()->next_context_data().value()->parms_id();
    c = f.encryptor->encrypt_zero_asymmetric_new_batched(B, second);
    check(all_level(c, second) && dec_all(c, zs), "batched encrypt_zero_asymmetric(second level)");
    c = f.encryptor->encrypt_zero_symmetric_new_batched(B, false, second);
    check(all_level(c, second) && dec_all(c, zs), "batched encrypt_zero_symmetric(second level)");
    auto messages = [&](size_t count) { std::vector<Suite::Vec> m(B); for (auto& v : m) v = f.random_simd(count); return m; };
    auto encode_all = [&](const std::vector<Suite::Vec>& m) { std::vector<Plaintext> p; for (const auto& v : m) p.push_back(f.encode(v)); return p; };
    std::vector<Suite::Vec> m = messages(f.slots());
    std::vector<Plaintext> plain = encode_all(m);
    c = f.encryptor->encrypt_symmetric_new_batched(batch_utils::collect_const_pointer(plain), false);
    check(dec_all(c, m), "batched: all slots, symmetric");
    if (f.scheme == SchemeType::BFV) {
        std::vector<Plaintext> up = encode_all(m);
        for (auto& p : up) p = f.batch->scale_up_new(p, std::nullopt);
        c = f.encryptor->encrypt_symmetric_new_batched(batch_utils::collect_const_pointer(up), false);
        check(dec_all(c, m), "batched BFV: scale_up before encrypt");
        c = f.encryptor->encrypt_symmetric_new_batched(batch_utils::collect_const_pointer(plain), false);
        std::vector<Plaintext> d = f.decryptor->bfv_decrypt_without_scaling_down_batched_new(batch_utils::collect_const_pointer(c));
        bool ok = d.size() == B;
        for (size_t i = 0; i < B && ok; i++) { f.batch->scale_down_inplace(d[i]); ok = f.near(f.decode(d[i]), m[i]); }
        check(ok, "batched BFV: bfv_decrypt_without_scaling_down_batched, then scale_down");
    }
    c = f.encryptor->encrypt_asymmetric_new_batched(batch_utils::collect_const_pointer(plain));
    check(dec_all(c, m), "batched: all slots, asymmetric");
    const size_t used = size_t(int(f.slots() * 0.3));
    m = messages(used); plain = encode_all(m);
    c = f.encryptor->encrypt_asymmetric_new_batched(batch_utils::collect_const_pointer(plain));
    check(dec_all(c, m), "batched: 30 % of the slots, asymmetric");
    m = messages(used); plain = encode_all(m);
    const auto ptrs = batch_utils::collect_const_pointer(plain);
    c = f.encryptor->encrypt_symmetric_new_batched(ptrs, false);
    check(dec_all(c, m), "batched: 30 % of the slots, symmetric");
    utils::RandomGenerator rng1(0x1234), rng2(0x1234);
    std::vector<Ciphertext> c1 = f.encryptor->encrypt_symmetric_new_batched(ptrs, false, &rng1), c2 = f.encryptor->encrypt_symmetric_new_batched(ptrs, false, &rng2);
    bool same = true, distinct = true;
    for (size_t i = 0; i < B; i++) { same = same && same_words(c1[i], c2[i], 1); if (i) distinct = distinct && !same_words(c1[i], c1[0], 1); }
    check(same && distinct && dec_all(c1, m), "batched: the same u_prng gives the same c1 per item (items differ from each other)");
    // seeded batch == seeded singles from the same generator state
    utils::RandomGenerator rng3(0x77), rng4(0x77);
    std::vector<Ciphertext> sb = f.encryptor->encrypt_symmetric_new_batched(ptrs, true, &rng3);
    bool seeded = true;
    for (size_t i = 0; i < B; i++) seeded = seeded && sb[i].contains_seed();
    for (auto& x : sb) x.expand_seed(f.context);
    check(seeded && dec_all(sb, m), "batched: save_seed keeps a seed per item; expand_seed, decrypt");
}

static void run_noise_budget(bool bgv) {                        // encryptor.cu:182-224
    const size_t n = 4096;
    EncryptionParameters parms(bgv ? SchemeType::BGV : SchemeType::BFV);
    parms.set_poly_modulus_degree(n);
    parms.set_plain_modulus(PlainModulus::batching(n, 20));
    parms.set_coeff_modulus(CoeffModulus::create(n, {35, 30, 35}));
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil);
    BatchEncoder encoder(context);
    context->to_device_inplace();
    encoder.to_device_inplace();
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_public_key(keygen.create_public_key(false));
    encryptor.set_secret_key(keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    Ciphertext c = encryptor.encrypt_zero_asymmetric_new();
    size_t budget = decryptor.invariant_noise_budget(c);
    const std::vector<uint64_t> zeros(n, 0);
    std::printf("   %s noise budget: fresh %zu", bgv ? "BGV" : "BFV", budget);
    bool ok = budget >= 30 && budget <= 40 && encoder.decode_new(decryptor.decrypt_new(c)) == zeros;
    evaluator.square_inplace(c);
    budget = decryptor.invariant_noise_budget(c);
    std::printf(", squared %zu", budget);
    ok = ok && budget <= 10 && encoder.decode_new(decryptor.decrypt_new(c)) == zeros;
    evaluator.square_inplace(c);
    budget = decryptor.invariant_noise_budget(c);
    std::printf(", squared twice %zu\n", budget);
    size_t non_zero = 0;
    for (uint64_t v : encoder.decode_new(decryptor.decrypt_new(c))) non_zero += v != 0;
    check(ok && budget == 0 && non_zero > 4000, bgv ? "BGV invariant_noise_budget: 30..40, <= 10, 0 (then garbage)" : "BFV invariant_noise_budget: 30..40, <= 10, 0 (then garbage)");
}

int main(int argc, char** argv) {
    try {
        const char* s = argc > 1 ? argv[1] : "bfv";
        if (!std::strcmp(s, "budget")) {
            run_noise_budget(false);
            run_noise_budget(true);
        } else {
            const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 32;
            const SchemeType scheme = !std::strcmp(s, "ckks") ? SchemeType::CKKS : !std::strcmp(s, "bgv") ? SchemeType::BGV : SchemeType::BFV;
            std::printf("scheme %s N %zu\n", s, n);
            Suite f(scheme, n);
            run_single(f);
            run_batched(f);
        }
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
