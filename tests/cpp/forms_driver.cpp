// C++ driver for tests/test_gpu_cpp_api.py::test_operand_forms_cpp_api: the reference's evaluator scenarios whose operands are in the
// NON-default representation (test/evaluator.cu: test_add_subtract_ntt / _intt, test_add_plain_scaled(_ntt), test_multiply_plain_ntt,
// test_multiply_plain_centralized, test_transform_plain_ntt, test_mod_switch_plain_to_next), replayed through the mirror for BFV, BGV and
// CKKS with full and partial (coeff_count = N / 3) plaintexts: every result is decrypted and compared with the plain computation.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%-66s %s\n", what, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

static std::vector<uint64_t> padded(std::vector<uint64_t> v, size_t n) { v.resize(n, 0); return v; }

// negacyclic product of two coefficient vectors mod t (the reference's GeneralHeContext::mul_poly)
static std::vector<uint64_t> mul_poly(const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, uint64_t t, size_t n) {
    std::vector<uint64_t> r(n, 0);
    for (size_t i = 0; i < a.size(); i++) {
        if (!a[i]) continue;
        for (size_t j = 0; j < b.size(); j++) {
            const uint64_t p = (unsigned __int128)a[i] * b[j] % t;
            const size_t k = i + j;
            if (k < n) r[k] = (r[k] + p) % t; else r[k - n] = (r[k - n] + t - p) % t;
        }
    }
    return r;
}

static void run_bfv_like(SchemeType scheme, size_t n) {
    const size_t cc = n / 3;
    const bool bgv = scheme == SchemeType::BGV;
    EncryptionParameters parms(scheme);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::create(n, {40, 40, 40, 40}));   // three data primes: the next level still has the noise room for a plain product
    parms.set_plain_modulus(PlainModulus::batching(n, 20));
    const uint64_t t = parms.plain_modulus().value();
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil, 0x77);
    context->to_device_inplace();
    BatchEncoder encoder(context);
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_public_key(keygen.create_public_key(false));
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    std::mt19937_64 gen(bgv ? 12 : 11);
    std::printf("-- %s\n", bgv ? "BGV" : "BFV");
    auto rnd = [&](size_t count) { std::vector<uint64_t> v(count); for (auto& x : v) x = gen() % t; return v; };
    auto addv = [&](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b, bool sub) {
        std::vector<uint64_t> r(a.size());
        for (size_t i = 0; i < a.size(); i++) r[i] = (a[i] + (sub ? t - b[i] : b[i])) % t;
        return r;
    };
    auto mulv = [&](const std::vector<uint64_t>& a, const std::vector<uint64_t>& b) {
        std::vector<uint64_t> r(a.size());
        for (size_t i = 0; i < a.size(); i++) r[i] = (unsigned __int128)a[i] * b[i] % t;
        return r;
    };
    const ParmsID first = context->first_parms_id();
    {   // add / sub with both operands moved to the other representation (BFV: to NTT; BGV: from NTT), evaluator.cu test_add_subtract_ntt / _intt
        const std::vector<uint64_t> m1 = rnd(n), m2 = rnd(n);
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_new(m1)), c2 = encryptor.encrypt_asymmetric_new(encoder.encode_new(m2));
        const bool was_ntt = c1.is_ntt_form();
        check(was_ntt == bgv, "default ciphertext form (BFV coefficient, BGV NTT)");
        auto flip = [&](Ciphertext& c) { if (c.is_ntt_form()) evaluator.transform_from_ntt_inplace(c); else evaluator.transform_to_ntt_inplace(c); };
        flip(c1); flip(c2);
        Ciphertext added = evaluator.add_new(c1, c2), subtracted = evaluator.sub_new(c1, c2), negated = evaluator.negate_new(c1);
        check(added.is_ntt_form() == !was_ntt, "add keeps the operands' (non-default) form");
        flip(added); flip(subtracted); flip(negated);
        check(encoder.decode_new(decryptor.decrypt_new(added)) == addv(m1, m2, false), "add in the non-default form");
        check(encoder.decode_new(decryptor.decrypt_new(subtracted)) == addv(m1, m2, true), "sub in the non-default form");
        check(encoder.decode_new(decryptor.decrypt_new(negated)) == addv(std::vector<uint64_t>(n, 0), m1, true), "negate in the non-default form");
        bool threw = false;
        Ciphertext c3 = encryptor.encrypt_asymmetric_new(encoder.encode_new(m2));
        try { evaluator.add_new(c1, c3); } catch (const std::invalid_argument&) { threw = true; }
        check(threw, "add of operands in different forms is rejected");
    }
    if (!bgv) {   // test_add_plain_scaled / _ntt: a scaled-up (RNS) plaintext as the addend, in coefficient and in NTT form; full and partial
        for (size_t count : {n, cc}) {
            for (bool ntt : {false, true}) {
                const std::vector<uint64_t> m1 = rnd(count), m2 = rnd(count);
                Plaintext e2 = encoder.encode_polynomial_new(m2);
                Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_polynomial_new(m1));
                encoder.scale_up_inplace(e2, c1.parms_id());
                bool shape = e2.coeff_count() == count && e2.poly_modulus_degree() == n && e2.data().size() == e2.coeff_modulus_size() * count;
                if (ntt) {
                    evaluator.transform_to_ntt_inplace(c1);
                    evaluator.transform_plain_to_ntt_inplace(e2, c1.parms_id());
                    shape = shape && e2.coeff_count() == n && e2.data().size() == e2.coeff_modulus_size() * n && e2.is_ntt_form();
                }
                Ciphertext added = evaluator.add_plain_new(c1, e2), subtracted = evaluator.sub_plain_new(c1, e2);
                if (ntt) { evaluator.transform_from_ntt_inplace(added); evaluator.transform_from_ntt_inplace(subtracted); }
                char label[96];
                std::snprintf(label, sizeof label, "add_plain(scaled%s), coeff_count %zu", ntt ? ", NTT" : "", count);
                check(shape && encoder.decode_polynomial_new(decryptor.decrypt_new(added)) == padded(addv(m1, m2, false), n), label);
                std::snprintf(label, sizeof label, "sub_plain(scaled%s), coeff_count %zu", ntt ? ", NTT" : "", count);
                check(encoder.decode_polynomial_new(decryptor.decrypt_new(subtracted)) == padded(addv(m1, m2, true), n), label);
            }
        }
    }
    {   // test_multiply_plain_ntt: the plaintext transformed ahead of time (centralize + NTT), SIMD and partial polynomial
        const std::vector<uint64_t> m1 = rnd(n), m2 = rnd(n);
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_new(m1));
        Plaintext e2 = encoder.encode_new(m2);
        evaluator.transform_plain_to_ntt_inplace(e2, c1.parms_id());
        check(e2.is_ntt_form() && e2.parms_id() == c1.parms_id(), "transform_plain_to_ntt sets the form and the level");
        Ciphertext prod = evaluator.multiply_plain_new(c1, e2);
        check(prod.is_ntt_form() == c1.is_ntt_form(), "multiply_plain(NTT plaintext) returns the ciphertext's form");
        check(encoder.decode_new(decryptor.decrypt_new(prod)) == mulv(m1, m2), "multiply_plain(ct, NTT plaintext), SIMD");
        if (!c1.is_ntt_form()) {      // the ciphertext moved to NTT form as well: one dyadic product
            Ciphertext c1n = evaluator.transform_to_ntt_new(c1);
            Ciphertext pn = evaluator.multiply_plain_new(c1n, e2);
            check(pn.is_ntt_form(), "multiply_plain(NTT ct, NTT plaintext) stays in NTT form");
            evaluator.transform_from_ntt_inplace(pn);
            check(encoder.decode_new(decryptor.decrypt_new(pn)) == mulv(m1, m2), "multiply_plain(NTT ct, NTT plaintext)");
        }
        const std::vector<uint64_t> p1 = rnd(cc), p2 = rnd(cc);
        Ciphertext d1 = encryptor.encrypt_asymmetric_new(encoder.encode_polynomial_new(p1));
        Plaintext f2 = encoder.encode_polynomial_new(p2);
        evaluator.transform_plain_to_ntt_inplace(f2, d1.parms_id());
        check(f2.coeff_count() == n && f2.poly_modulus_degree() == n && f2.data().size() == f2.coeff_modulus_size() * n, "partial plaintext -> NTT: full shape");
        check(encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(d1, f2))) == mul_poly(p1, p2, t, n), "multiply_plain(ct, NTT plaintext), partial polynomial");
    }
    {   // test_multiply_plain_centralized: full SIMD, full polynomial, partial polynomial
        const std::vector<uint64_t> m1 = rnd(n), m2 = rnd(n);
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_new(m1));
        Plaintext e2 = encoder.encode_new(m2);
        encoder.centralize_inplace(e2, std::nullopt);
        check(encoder.decode_new(decryptor.decrypt_new(evaluator.multiply_plain_new(c1, e2))) == mulv(m1, m2), "multiply_plain(ct, centralized), SIMD");
        for (size_t count : {n, cc}) {
            const std::vector<uint64_t> p1 = rnd(count), p2 = rnd(count);
            Ciphertext d1 = encryptor.encrypt_asymmetric_new(encoder.encode_polynomial_new(p1));
            Plaintext f2 = encoder.encode_polynomial_new(p2);
            encoder.centralize_inplace(f2, std::nullopt);
            const bool shape = f2.coeff_count() == count && f2.poly_modulus_degree() == n && f2.data().size() == f2.coeff_modulus_size() * count;
            char label[96];
            std::snprintf(label, sizeof label, "multiply_plain(ct, centralized), coeff_count %zu", count);
            check(shape && encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(d1, f2))) == mul_poly(p1, p2, t, n), label);
            // at the next level (mod switch first)
            Ciphertext d2 = evaluator.mod_switch_to_next_new(d1);
            Plaintext g2 = encoder.centralize_new(encoder.encode_polynomial_new(p2), d2.parms_id());
            std::snprintf(label, sizeof label, "... at the next level, coeff_count %zu", count);
            check(g2.data().size() == g2.coeff_modulus_size() * count && g2.coeff_modulus_size() + 1 == f2.coeff_modulus_size() &&
                  encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(d2, g2))) == mul_poly(p1, p2, t, n), label);
        }
    }
    {   // test_transform_plain_ntt: NTT(INTT(x)) == x on the words; then encrypt a transformed plaintext
        for (size_t count : {n, cc}) {
            Plaintext e = encoder.encode_polynomial_new(rnd(count));
            evaluator.transform_plain_to_ntt_inplace(e, first);
            Plaintext back = evaluator.transform_plain_from_ntt_new(e);
            check(!back.is_ntt_form() && back.coeff_count() == n, "transform_plain_from_ntt: coefficient form, full length");
            Plaintext again = evaluator.transform_plain_to_ntt_new(back, e.parms_id());
            char label[96];
            std::snprintf(label, sizeof label, "to_ntt(from_ntt(p)) == p word for word, coeff_count %zu", count);
            check(e.data().to_vector() == again.data().to_vector(), label);
        }
        const std::vector<uint64_t> p = rnd(cc);
        Plaintext e = encoder.encode_polynomial_new(p);
        if (!bgv) encoder.scale_up_inplace(e, std::nullopt);
        evaluator.transform_plain_to_ntt_inplace(e, first);
        bool ok = true, threw = false;
        try {
            Ciphertext c = encryptor.encrypt_asymmetric_new(e);
            if (!bgv) { check(c.is_ntt_form(), "encrypt(NTT-form scaled plaintext) is in NTT form"); evaluator.transform_from_ntt_inplace(c); }
            ok = encoder.decode_polynomial_new(decryptor.decrypt_new(c)) == padded(p, n);
        } catch (const std::exception& ex) { threw = true; std::printf("   (%s)\n", ex.what()); }
        if (bgv) check(threw || ok, "BGV: encrypt(NTT-form plaintext) either works or is rejected");
        else check(!threw && ok, "decrypt(encrypt(NTT-form scaled plaintext)) == m");
    }
}

static void run_ckks(size_t n) {
    const size_t cc = n / 3;
    std::printf("-- CKKS\n");
    EncryptionParameters parms(SchemeType::CKKS);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil, 0x3c);
    context->to_device_inplace();
    CKKSEncoder encoder(context);
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_public_key(keygen.create_public_key(false));
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    const double scale = std::pow(2.0, 30), tol = 1e-4;
    std::mt19937_64 gen(21);
    std::uniform_real_distribution<double> U(-8, 8);
    auto rnd_c = [&]() { std::vector<std::complex<double>> v(n / 2); for (auto& x : v) x = {U(gen), U(gen)}; return v; };
    auto rnd_d = [&](size_t count) { std::vector<double> v(count); for (auto& x : v) x = U(gen); return v; };
    auto near_c = [&](const std::vector<std::complex<double>>& a, const std::vector<std::complex<double>>& b, double tolerance) {
        if (a.size() != b.size()) return false;
        for (size_t i = 0; i < a.size(); i++) if (std::abs(a[i] - b[i]) > tolerance) return false;
        return true;
    };
    auto near_d = [&](std::vector<double> a, std::vector<double> b, double tolerance) {
        a.resize(n, 0); b.resize(n, 0);
        for (size_t i = 0; i < n; i++) if (std::abs(a[i] - b[i]) > tolerance) return false;
        return true;
    };
    {   // test_add_subtract_intt: CKKS ciphertexts moved to coefficient form, added there, moved back
        const auto m1 = rnd_c(), m2 = rnd_c();
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_complex64_simd_new(m1, std::nullopt, scale));
        Ciphertext c2 = encryptor.encrypt_asymmetric_new(encoder.encode_complex64_simd_new(m2, std::nullopt, scale));
        check(c1.is_ntt_form(), "default CKKS ciphertext form is NTT");
        evaluator.transform_from_ntt_inplace(c1); evaluator.transform_from_ntt_inplace(c2);
        Ciphertext added = evaluator.add_new(c1, c2), subtracted = evaluator.sub_new(c1, c2);
        check(!added.is_ntt_form() && added.scale() == scale, "add in coefficient form keeps form and scale");
        evaluator.transform_to_ntt_inplace(added); evaluator.transform_to_ntt_inplace(subtracted);
        std::vector<std::complex<double>> s(n / 2), d(n / 2);
        for (size_t i = 0; i < n / 2; i++) { s[i] = m1[i] + m2[i]; d[i] = m1[i] - m2[i]; }
        check(near_c(encoder.decode_complex64_simd_new(decryptor.decrypt_new(added)), s, tol), "CKKS add in coefficient form");
        check(near_c(encoder.decode_complex64_simd_new(decryptor.decrypt_new(subtracted)), d, tol), "CKKS sub in coefficient form");
    }
    {   // test_mod_switch_plain_to_next
        const auto m = rnd_c();
        Plaintext e = encoder.encode_complex64_simd_new(m, std::nullopt, scale);
        Plaintext next = evaluator.mod_switch_plain_to_next_new(e);
        check(next.coeff_modulus_size() + 1 == e.coeff_modulus_size() && next.scale() == scale && near_c(encoder.decode_complex64_simd_new(next), m, tol), "mod_switch_plain_to_next keeps the message");
        Plaintext same = evaluator.mod_switch_plain_to_new(e, context->first_parms_id());
        check(same.parms_id() == e.parms_id() && near_c(encoder.decode_complex64_simd_new(same), m, tol), "mod_switch_plain_to(first) is the identity");
        const ParmsID last = context->last_parms_id();
        Plaintext bottom = evaluator.mod_switch_plain_to_new(e, last);
        check(bottom.parms_id() == last && bottom.coeff_modulus_size() == 1 && near_c(encoder.decode_complex64_simd_new(bottom), m, tol), "mod_switch_plain_to(last)");
        bool threw = false;
        try { evaluator.mod_switch_plain_to_new(bottom, context->first_parms_id()); } catch (const std::invalid_argument&) { threw = true; }
        check(threw, "mod_switch_plain_to a higher level is rejected");
    }
    {   // test_transform_plain_ntt on CKKS plaintexts (default NTT): from_ntt then to_ntt restores the words; coefficient-form plaintext encrypts
        for (size_t count : {n, cc}) {
            const auto v = rnd_d(count);
            Plaintext e = encoder.encode_float64_polynomial_new(v, std::nullopt, scale);
            check(e.is_ntt_form(), "CKKS polynomial plaintexts are in NTT form");
            Plaintext coeff = evaluator.transform_plain_from_ntt_new(e);
            Plaintext again = evaluator.transform_plain_to_ntt_new(coeff, e.parms_id());
            char label[96];
            std::snprintf(label, sizeof label, "CKKS to_ntt(from_ntt(p)) == p word for word, coeff_count %zu", count);
            check(!coeff.is_ntt_form() && e.data().to_vector() == again.data().to_vector(), label);
            Ciphertext c = encryptor.encrypt_asymmetric_new(coeff);
            check(!c.is_ntt_form(), "encrypt(coefficient-form CKKS plaintext) is in coefficient form");
            evaluator.transform_to_ntt_inplace(c);
            std::snprintf(label, sizeof label, "decrypt(encrypt(coefficient-form plaintext)), coeff_count %zu", count);
            check(near_d(encoder.decode_float64_polynomial_new(decryptor.decrypt_new(c)), v, tol), label);
        }
    }
    {   // add_plain / multiply_plain with polynomial plaintexts (partial), evaluator.cu test_add_plain / test_multiply_plain CKKS branches
        const auto v1 = rnd_d(cc), v2 = rnd_d(cc);
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(encoder.encode_float64_polynomial_new(v1, std::nullopt, scale));
        Plaintext e2 = encoder.encode_float64_polynomial_new(v2, std::nullopt, scale);
        std::vector<double> sum(n, 0), prod(n, 0);
        for (size_t i = 0; i < cc; i++) sum[i] = v1[i] + v2[i];
        for (size_t i = 0; i < cc; i++) for (size_t j = 0; j < cc; j++) { const size_t k = i + j; if (k < n) prod[k] += v1[i] * v2[j]; else prod[k - n] -= v1[i] * v2[j]; }
        check(near_d(encoder.decode_float64_polynomial_new(decryptor.decrypt_new(evaluator.add_plain_new(c1, e2))), sum, tol), "CKKS add_plain, partial polynomial");
        Ciphertext p = evaluator.multiply_plain_new(c1, e2);
        check(p.scale() == scale * scale && near_d(encoder.decode_float64_polynomial_new(decryptor.decrypt_new(p)), prod, 1e-2), "CKKS multiply_plain, partial polynomial");
    }
}

int main(int argc, char** argv) {
    try {
        const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2048;      // 32: the reference's own test size (the small-ring kernels)
        std::printf("N %zu\n", n);
        run_bfv_like(SchemeType::BFV, n);
        run_bfv_like(SchemeType::BGV, n);
        run_ckks(n);
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
