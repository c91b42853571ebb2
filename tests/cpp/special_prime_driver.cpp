// C++ driver for tests/test_gpu_cpp_api.py::test_special_prime_for_encryption_cpp_api: the reference's test/special_prime_for_encryption.cu:16-70 --
// EncryptionParameters::set_use_special_prime_for_encryption(true): the first level IS the key level (he_context.cu:77), ciphertexts carry every prime of the
// chain incl. the special one; encrypt (asymmetric and symmetric) -> decrypt must give the message back for BFV, BGV and CKKS.   usage: special_prime_driver <bfv|bgv|ckks> <N>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

int main(int argc, char** argv) {
    try {
        const std::string sch = argc > 1 ? argv[1] : "bfv";
        const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 32;
        const SchemeType scheme = sch == "bgv" ? SchemeType::BGV : sch == "ckks" ? SchemeType::CKKS : SchemeType::BFV;
        EncryptionParameters params(scheme);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        if (scheme != SchemeType::CKKS) params.set_plain_modulus(PlainModulus::batching(n, 20));
        params.set_use_special_prime_for_encryption(true);
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0x123);
        context->to_device_inplace();
        std::printf("first_is_key_level %d first_limbs %zu\n", context->first_parms_id() == context->key_parms_id() ? 1 : 0,
                    context->first_context_data().value()->parms().coeff_modulus().size());
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false));
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator ev(context);
        std::mt19937_64 gen(5);
        size_t bad = 0;
        if (scheme == SchemeType::CKKS) {
            CKKSEncoder encoder(context);
            const double scale = std::pow(2.0, n > 64 ? 36 : 20);      // (the reference uses 2^20 at N = 32; fresh-encryption noise grows with N)
            std::vector<std::complex<double>> m(encoder.slot_count());
            std::uniform_real_distribution<double> U(-10.0, 10.0);
            for (auto& v : m) v = {U(gen), U(gen)};
            Plaintext p = encoder.encode_complex64_simd_new(m, std::nullopt, scale);
            {   // the *_slice forms (ckks_encoder.h:93-283) give the words of the vector forms
                const Plaintext ps = encoder.encode_complex64_simd_slice_new(utils::ConstSlice<std::complex<double>>(m.data(), m.size(), false), std::nullopt, scale);
                bad += ps.data().to_vector() != p.data().to_vector();
                const std::vector<std::complex<double>> a = encoder.decode_complex64_simd_new(p), b = encoder.decode_complex64_simd_slice_new(p).to_vector();
                bad += a != b;
                std::vector<double> coeffs(2 * m.size());
                for (size_t i = 0; i < coeffs.size(); i++) coeffs[i] = 0.25 * (double)(i % 17) - 1.0;
                const Plaintext pf = encoder.encode_float64_polynomial_new(coeffs, std::nullopt, scale);
                bad += encoder.encode_float64_polynomial_slice_new(utils::ConstSlice<double>(coeffs.data(), coeffs.size(), false), std::nullopt, scale).data().to_vector() != pf.data().to_vector();
                std::vector<double> out(coeffs.size());
                encoder.decode_float64_polynomial_slice(pf, utils::Slice<double>(out.data(), out.size(), false));
                bad += out != encoder.decode_float64_polynomial_new(pf);
            }
            for (int sym = 0; sym < 2; sym++) {
                Ciphertext c = sym ? encryptor.encrypt_symmetric_new(p, false) : encryptor.encrypt_asymmetric_new(p);
                if (c.coeff_modulus_size() != 4) bad++;
                const auto got = encoder.decode_complex64_simd_new(decryptor.decrypt_new(c));
                for (size_t i = 0; i < m.size(); i++) bad += !(std::abs(got[i] - m[i]) < 1e-2);
                // an operation at that level: add to itself
                const auto twice = encoder.decode_complex64_simd_new(decryptor.decrypt_new(ev.add_new(c, c)));
                for (size_t i = 0; i < m.size(); i++) bad += !(std::abs(twice[i] - 2.0 * m[i]) < 2e-2);
            }
        } else {
            BatchEncoder encoder(context);
            const uint64_t t = params.plain_modulus().value();
            std::vector<uint64_t> m(encoder.slot_count());
            for (auto& v : m) v = gen() % t;
            Plaintext p = encoder.encode_new(m);
            for (int sym = 0; sym < 2; sym++) {
                Ciphertext c = sym ? encryptor.encrypt_symmetric_new(p, false) : encryptor.encrypt_asymmetric_new(p);
                if (c.coeff_modulus_size() != 4) bad++;
                const auto got = encoder.decode_new(decryptor.decrypt_new(c));
                for (size_t i = 0; i < m.size(); i++) bad += got[i] != m[i];
                const auto twice = encoder.decode_new(decryptor.decrypt_new(ev.add_new(c, c)));
                for (size_t i = 0; i < m.size(); i++) bad += twice[i] != (2 * m[i]) % t;
            }
        }
        std::printf("mismatches %zu\n", bad);
        std::printf(bad ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return bad ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
