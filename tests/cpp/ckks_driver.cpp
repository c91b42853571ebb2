// C++ driver for tests/test_gpu_cpp_api.py::test_ckks_cpp_api: BASELINE config 3's parameters through the mirror API --
// CKKSEncoder, KeyGenerator, Encryptor, Evaluator (multiply, relinearize, rescale_to_next, rotate_vector,
// complex_conjugate, multiply_plain), Decryptor -- printing the maximum slot error of each result.
#include <cmath>
#include <complex>
#include <cstdio>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;
using cd = std::complex<double>;

static double max_err(const std::vector<cd>& got, const std::vector<cd>& want) {
    double e = 0;
    for (size_t i = 0; i < want.size(); i++) e = std::max(e, std::abs(got[i] - want[i]));
    return e;
}

int main() {
    try {
        const size_t n = 16384;
        EncryptionParameters params(SchemeType::CKKS);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {50, 50, 50, 50, 50, 50}));
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, 0x5eed);
        context->to_device_inplace();
        CKKSEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false));
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator evaluator(context);
        RelinKeys rk = keygen.create_relin_keys(false);
        GaloisKeys gk = keygen.create_galois_keys_from_steps({1, -3, 0}, false);

        const size_t slots = encoder.slot_count();
        std::mt19937_64 gen(7);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        std::vector<cd> z1(slots), z2(slots), prod(slots), rot1(slots), rotm3(slots), conj(slots), zw(slots);
        for (size_t i = 0; i < slots; i++) { z1[i] = cd(U(gen), U(gen)); z2[i] = cd(U(gen), U(gen)); }
        for (size_t i = 0; i < slots; i++) {
            prod[i] = z1[i] * z2[i];
            rot1[i] = z1[(i + 1) % slots];
            rotm3[i] = z1[(i + slots - 3) % slots];
            conj[i] = std::conj(z1[i]);
            zw[i] = z1[i] * 0.5;
        }
        const double scale = std::pow(2.0, 40);
        Plaintext p1 = encoder.encode_complex64_simd_new(z1, std::nullopt, scale), p2 = encoder.encode_complex64_simd_new(z2, std::nullopt, scale);
        std::printf("encode_roundtrip %.3e\n", max_err(encoder.decode_complex64_simd_new(p1), z1));
        Ciphertext c1 = encryptor.encrypt_asymmetric_new(p1), c2 = encryptor.encrypt_asymmetric_new(p2);
        std::printf("decrypt %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(c1)), z1));

        Ciphertext m = evaluator.multiply_new(c1, c2);
        evaluator.relinearize_inplace(m, rk);
        evaluator.rescale_to_next_inplace(m);
        std::printf("levels %zu scale_log2 %.3f\n", m.coeff_modulus_size(), std::log2(m.scale()));
        std::printf("mul_relin_rescale %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(m)), prod));

        std::printf("rotate1 %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(evaluator.rotate_vector_new(c1, 1, gk))), rot1));
        std::printf("rotate-3 %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(evaluator.rotate_vector_new(c1, -3, gk))), rotm3));
        std::printf("conjugate %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(evaluator.complex_conjugate_new(c1, gk))), conj));

        Plaintext half = encoder.encode_float64_single_new(0.5, std::nullopt, scale);
        Ciphertext mp = evaluator.multiply_plain_new(c1, half);
        evaluator.rescale_to_next_inplace(mp);
        std::printf("multiply_plain %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(mp)), zw));

        Ciphertext s = evaluator.add_new(c1, c2);
        std::vector<cd> sum(slots);
        for (size_t i = 0; i < slots; i++) sum[i] = z1[i] + z2[i];
        std::printf("add %.3e\n", max_err(encoder.decode_complex64_simd_new(decryptor.decrypt_new(s)), sum));
        std::printf("OK\n");
        MemoryPool::Destroy();
        return 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
