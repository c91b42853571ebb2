// C++ driver for tests/test_gpu_cpp_api.py::test_plain_ops_cpp_api: the plaintext-side API of the mirror --
// BatchEncoder::scale_up / scale_down / centralize / decentralize with partial RNS plaintexts (batch_encoder.cu:558-662) and their use
// as operands of encrypt / add_plain / multiply_plain; Evaluator::apply_galois_plain against the rotation of the encrypted vector;
// CKKSEncoder's integer and single-value encodings; Ciphertext::is_transparent; Modulus::reduce_mul_uint64.
#include <cmath>
#include <cstdio>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static int failures = 0;
static void check(bool ok, const char* what) {
    std::printf("%-58s %s\n", what, ok ? "ok" : "FAIL");
    if (!ok) failures++;
}

static std::vector<uint64_t> padded(std::vector<uint64_t> v, size_t n) { v.resize(n, 0); return v; }

static void run_bfv_like(SchemeType scheme) {
    const size_t n = 8192;
    const bool bgv = scheme == SchemeType::BGV;
    EncryptionParameters parms(scheme);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::create(n, {40, 40, 40}));
    parms.set_plain_modulus(PlainModulus::batching(n, 20));
    const uint64_t t = parms.plain_modulus().value();
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Classical128, 0x91a);
    context->to_device_inplace();
    BatchEncoder encoder(context);
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_secret_key(keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    GaloisKeys gk = keygen.create_galois_keys(false);
    std::mt19937_64 gen(5);
    std::printf("-- %s\n", bgv ? "BGV" : "BFV");

    for (size_t cc : {size_t(100), n}) {
        std::vector<uint64_t> m(cc), m2(n);
        for (auto& v : m) v = gen() % t;
        for (auto& v : m2) v = gen() % t;
        Plaintext pm = encoder.encode_polynomial_new(m), pm2 = encoder.encode_polynomial_new(m2);
        char label[96];
        // centralize / decentralize
        Plaintext cen = encoder.centralize_new(pm, std::nullopt);
        std::snprintf(label, sizeof label, "centralize shape (coeff_count %zu)", cc);
        check(cen.coeff_count() == cc && cen.data().size() == 2 * cc && !(cen.parms_id() == parms_id_zero), label);
        std::snprintf(label, sizeof label, "decentralize(centralize(m)) == m (coeff_count %zu)", cc);
        check(encoder.decode_polynomial_new(encoder.decentralize_new(cen)) == m, label);
        Ciphertext c2 = encryptor.encrypt_symmetric_new(pm2, false);
        const std::vector<uint64_t> direct = encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(c2, pm)));
        const std::vector<uint64_t> via = encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.multiply_plain_new(c2, cen)));
        std::snprintf(label, sizeof label, "multiply_plain(centralized) == multiply_plain(m) (%zu)", cc);
        check(direct == via, label);
        if (bgv) continue;                                   // scale_up / scale_down are BFV-only, as in the reference
        Plaintext up = encoder.scale_up_new(pm, std::nullopt);
        std::snprintf(label, sizeof label, "scale_up shape (coeff_count %zu)", cc);
        check(up.coeff_count() == cc && up.data().size() == 2 * cc && !up.is_ntt_form(), label);
        std::snprintf(label, sizeof label, "scale_down(scale_up(m)) == m (coeff_count %zu)", cc);
        check(encoder.decode_polynomial_new(encoder.scale_down_new(up)) == m, label);
        std::snprintf(label, sizeof label, "decrypt(encrypt(scale_up(m))) == m (coeff_count %zu)", cc);
        check(padded(encoder.decode_polynomial_new(decryptor.decrypt_new(encryptor.encrypt_symmetric_new(up, false))), n) == padded(m, n), label);
        std::vector<uint64_t> sum(n);
        for (size_t i = 0; i < n; i++) sum[i] = (m2[i] + (i < cc ? m[i] : 0)) % t;
        std::snprintf(label, sizeof label, "add_plain(ct, scale_up(m)) (coeff_count %zu)", cc);
        check(padded(encoder.decode_polynomial_new(decryptor.decrypt_new(evaluator.add_plain_new(c2, up))), n) == sum, label);
        // lower level
        const ParmsID second = context->first_context_data().value()->next_context_data().value()->parms_id();
        Plaintext up2 = encoder.scale_up_new(pm, second);
        std::snprintf(label, sizeof label, "scale_up at the next level round-trips (%zu)", cc);
        check(up2.data().size() == cc && encoder.decode_polynomial_new(encoder.scale_down_new(up2)) == m, label);
    }
    {
        bool threw = false;
        try { encoder.scale_down_new(encoder.encode_polynomial_new({1, 2, 3})); } catch (const std::invalid_argument&) { threw = true; } catch (const std::logic_error&) { threw = bgv; }
        check(threw, "scale_down of a mod-t plaintext is rejected");
    }
    // apply_galois_plain: the plaintext automorphism equals the automorphism under encryption
    std::vector<uint64_t> slots(n);
    for (auto& v : slots) v = gen() % t;
    Plaintext ps = encoder.encode_new(slots);
    Ciphertext cs = encryptor.encrypt_symmetric_new(ps, false);
    for (int step : {1, -4, 0}) {
        const size_t g = utils::galois_element_from_step(n, step);
        const std::vector<uint64_t> from_plain = encoder.decode_new(evaluator.apply_galois_plain_new(ps, g));
        const std::vector<uint64_t> from_cipher = encoder.decode_new(decryptor.decrypt_new(evaluator.apply_galois_new(cs, g, gk)));
        char label[64];
        std::snprintf(label, sizeof label, "apply_galois_plain == apply_galois (step %d)", step);
        bool moved = from_plain != slots;
        check(from_plain == from_cipher && moved, label);
    }
    {
        Plaintext shortp = encoder.encode_polynomial_new({1, 2, 3});
        const std::vector<uint64_t> got = encoder.decode_polynomial_new(evaluator.apply_galois_plain_new(shortp, 3));    // 1 + 2 X^3 + 3 X^6
        bool ok = got.size() == n && got[0] == 1 && got[3] == 2 && got[6] == 3;
        for (size_t i = 0; i < n && ok; i++) ok = (i == 0 || i == 3 || i == 6) ? true : got[i] == 0;
        check(ok, "apply_galois_plain zero-pads short plaintexts");
        bool threw = false;
        try { evaluator.apply_galois_plain_new(ps, 4); } catch (const std::invalid_argument&) { threw = true; }
        check(threw, "even Galois elements are rejected");
    }
    check(Ciphertext().is_transparent() && !cs.is_transparent(), "is_transparent");
    {
        // the small host-side types user programs touch (utils/box.h, plaintext.h to_string, timer.h, compression.h)
        check(encoder.encode_polynomial_new({4, 3, 0, 0x7ff}).to_string() == "7FFx^3 + 3x^1 + 4" && encoder.encode_polynomial_new({0, 0}).to_string() == "0", "Plaintext::to_string");
        const ParmsID first = context->first_parms_id();
        check(first[0] == first.v[0] && first[3] == first.v[3], "ParmsID::operator[]");
        utils::ConstSlice<Modulus> q = parms.coeff_modulus();
        utils::Array<Modulus> host(q.size(), false);
        host.copy_from_slice(q);
        check(q.size() == 3 && host[2].value() == q[2].value() && q.to_vector().size() == 3 && CoeffModulus::create(n, {40, 40}).to_vector().size() == 2, "ConstSlice / Array");
        {   // Zstd is available exactly when the zstd runtime library can be loaded; either it round-trips or it is refused loudly
            bool consistent = utils::compression::available(CompressionMode::Nil);
            std::stringstream ss;
            Plaintext probe = encoder.encode_polynomial_new({1, 2, 3});
            if (utils::compression::available(CompressionMode::Zstd)) {
                const size_t written = probe.save(ss, CompressionMode::Zstd);
                consistent = consistent && written == ss.str().size() && written <= probe.serialized_size_upperbound(CompressionMode::Zstd) &&
                             Plaintext::load_new(ss).data().to_vector() == probe.data().to_vector();
            } else {
                bool threw = false;
                try { probe.save(ss, CompressionMode::Zstd); } catch (const std::invalid_argument&) { threw = true; }
                consistent = consistent && threw;
            }
            check(consistent, "compression::available agrees with what save does");
        }
        bench::TimerSingle timer;
        timer.tick(); timer.tock();
        check(timer.count() == 1 && context->first_context_data_pointer() != nullptr && context->get_context_data_pointer(parms_id_zero) == nullptr, "TimerSingle, *_pointer getters");
    }
    const Modulus q0 = parms.coeff_modulus()[0];
    check(q0.reduce_mul_uint64(q0.value() - 1, q0.value() - 1) == 1, "Modulus::reduce_mul_uint64");
}

static void run_ckks() {
    const size_t n = 8192;
    std::printf("-- CKKS\n");
    EncryptionParameters parms(SchemeType::CKKS);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Classical128, 0x3c);
    context->to_device_inplace();
    CKKSEncoder encoder(context);
    KeyGenerator keygen(context);
    Encryptor encryptor(context);
    encryptor.set_secret_key(keygen.secret_key());
    Decryptor decryptor(context, keygen.secret_key());
    Evaluator evaluator(context);
    GaloisKeys gk = keygen.create_galois_keys(false);
    const double scale = std::pow(2.0, 40);
    // integers at scale 1
    Plaintext pi = encoder.encode_integer64_polynomial_new({1, -2, 3, -(1ll << 40)}, std::nullopt);
    std::vector<double> back = encoder.decode_float64_polynomial_new(pi);
    check(pi.scale() == 1.0 && pi.is_ntt_form() && back[0] == 1 && back[1] == -2 && back[2] == 3 && back[3] == -std::pow(2.0, 40) && back[4] == 0, "encode_integer64_polynomial");
    Plaintext p7 = encoder.encode_integer64_single_new(-7, std::nullopt);
    bool all7 = true;
    for (const auto& v : encoder.decode_complex64_simd_new(p7)) all7 = all7 && std::abs(v - std::complex<double>(-7, 0)) < 1e-9;
    check(all7, "encode_integer64_single fills every slot");
    // an integer plaintext times a ciphertext keeps the scale
    std::vector<std::complex<double>> vals(n / 2);
    std::mt19937_64 gen(8);
    std::uniform_real_distribution<double> U(-1, 1);
    for (auto& v : vals) v = {U(gen), U(gen)};
    Plaintext pv = encoder.encode_complex64_simd_new(vals, std::nullopt, scale);
    Ciphertext cv = encryptor.encrypt_symmetric_new(pv, false);
    Ciphertext c3 = evaluator.multiply_plain_new(cv, encoder.encode_integer64_single_new(3, std::nullopt));
    double err = 0;
    const auto d3 = encoder.decode_complex64_simd_new(decryptor.decrypt_new(c3));
    for (size_t i = 0; i < vals.size(); i++) err = std::max(err, std::abs(d3[i] - 3.0 * vals[i]));
    check(c3.scale() == scale && err < 1e-6, "ciphertext x encode_integer64_single(3): same scale, 3x values");
    // a single complex value in every slot
    const std::complex<double> z(1.5, -2.25);
    bool allz = true;
    for (const auto& v : encoder.decode_complex64_simd_new(encoder.encode_complex64_single_new(z, std::nullopt, scale))) allz = allz && std::abs(v - z) < 1e-7;
    check(allz, "encode_complex64_single");
    // apply_galois_plain on an NTT-form RNS plaintext = rotate_vector under encryption
    for (int step : {1, -3}) {
        const size_t g = utils::galois_element_from_step(n, step);
        const auto from_plain = encoder.decode_complex64_simd_new(evaluator.apply_galois_plain_new(pv, g));
        const auto from_cipher = encoder.decode_complex64_simd_new(decryptor.decrypt_new(evaluator.rotate_vector_new(cv, step, gk)));
        double e = 0, moved = 0;
        for (size_t i = 0; i < vals.size(); i++) { e = std::max(e, std::abs(from_plain[i] - from_cipher[i])); moved = std::max(moved, std::abs(from_plain[i] - vals[i])); }
        char label[64];
        std::snprintf(label, sizeof label, "apply_galois_plain == rotate_vector (step %d)", step);
        check(e < 1e-6 && moved > 1e-3, label);
    }
}

int main() {
    try {
        run_bfv_like(SchemeType::BFV);
        run_bfv_like(SchemeType::BGV);
        run_ckks();
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
