// C++ driver for tests/test_gpu_cpp_api.py::test_ring2k_cpp_api: examples/13_ring2k.cu -- elements of Z_{2^k} through
// PolynomialEncoderRing2k<T>: scale_up -> encrypt -> multiply by a centralized plaintext -> bfv_decrypt_without_scaling_down ->
// scale_down gives the negacyclic product mod 2^k; for T = uint32_t, uint64_t and unsigned __int128 (N=16384, six 60-bit primes).
#include <cstdio>
#include <random>

#include "../../troy-nova_amd/troy/ring2k.h"

using namespace troy;
typedef unsigned __int128 u128;

template <typename T>
static bool run(const HeContextPointer& he, const Encryptor& encryptor, const Decryptor& decryptor, const Evaluator& evaluator, size_t plain_bits, const char* name) {
    linear::PolynomialEncoderRing2k<T> encoder(he, plain_bits);
    const T mask = encoder.t_mask();
    // the example's vectors first
    std::vector<T> lhs = {1, 2, 3}, rhs = {4, 5, 6};
    Plaintext plain_lhs = encoder.scale_up_new(lhs, std::nullopt), plain_rhs = encoder.centralize_new(rhs, std::nullopt);
    Ciphertext cipher_lhs = encryptor.encrypt_symmetric_new(plain_lhs, false);
    Ciphertext cipher_result = evaluator.multiply_plain_new(cipher_lhs, plain_rhs);
    std::vector<T> decoded = encoder.scale_down_new(decryptor.bfv_decrypt_without_scaling_down_new(cipher_result));
    std::vector<T> truth = {4, 13, 28, 27, 18};
    truth.resize(decoded.size(), 0);
    bool ok = decoded == truth;
    // random full-width elements, wrap-around included
    std::mt19937_64 gen(sizeof(T) * 100 + plain_bits);
    auto rnd = [&]() { T v = static_cast<T>(gen()); if (sizeof(T) == 16) v = (v << 32 << 32) | static_cast<T>(gen()); return static_cast<T>(v & mask); };
    const size_t na = 70, nb = 33, n = encoder.slot_count();
    std::vector<T> a(na), b(nb), want(n, 0);
    for (auto& v : a) v = rnd();
    for (auto& v : b) v = rnd();
    for (size_t i = 0; i < na; i++) for (size_t j = 0; j < nb; j++) want[i + j] = static_cast<T>((want[i + j] + a[i] * b[j]) & mask);
    Ciphertext ca = encryptor.encrypt_symmetric_new(encoder.scale_up_new(a, std::nullopt), false);
    Ciphertext cr = evaluator.multiply_plain_new(ca, encoder.centralize_new(b, std::nullopt));
    // also survives a modulus switch (the helper of the lower level is used)
    Ciphertext low = evaluator.mod_switch_to_next_new(cr);
    const std::vector<T> got = encoder.scale_down_new(decryptor.bfv_decrypt_without_scaling_down_new(cr));
    const std::vector<T> got_low = encoder.scale_down_new(decryptor.bfv_decrypt_without_scaling_down_new(low));
    size_t bad = 0, bad_low = 0;
    for (size_t i = 0; i < n; i++) { bad += got[i] != want[i]; bad_low += got_low[i] != want[i]; }
    std::printf("%s k=%zu example %d random_mismatches %zu after_mod_switch %zu\n", name, plain_bits, ok ? 1 : 0, bad, bad_low);
    return ok && bad == 0 && bad_low == 0;
}

int main() {
    try {
        const size_t n = 16384;
        EncryptionParameters parms(SchemeType::BFV);
        parms.set_poly_modulus_degree(n);
        parms.set_plain_modulus(1 << 20);                               // unused by the ring-2^k encoder
        parms.set_coeff_modulus(CoeffModulus::create(n, {60, 60, 60, 60, 60, 60}));   // 300 data bits: room for 128-bit elements times 127-bit multipliers
        HeContextPointer he = HeContext::create(parms, true, SecurityLevel::Classical128, 0x2f);
        he->to_device_inplace();
        KeyGenerator keygen(he);
        Encryptor encryptor(he);
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(he, keygen.secret_key());
        Evaluator evaluator(he);
        bool ok = true;
        ok = run<uint64_t>(he, encryptor, decryptor, evaluator, 64, "uint64") && ok;
        ok = run<uint64_t>(he, encryptor, decryptor, evaluator, 44, "uint64") && ok;
        ok = run<uint32_t>(he, encryptor, decryptor, evaluator, 32, "uint32") && ok;
        ok = run<u128>(he, encryptor, decryptor, evaluator, 128, "uint128") && ok;
        ok = run<u128>(he, encryptor, decryptor, evaluator, 80, "uint128") && ok;
        bool threw = false;
        try { linear::PolynomialEncoderRing2k<uint64_t> bad(he, 32); } catch (const std::invalid_argument&) { threw = true; }
        std::printf("narrow_k_rejected %d\n", threw ? 1 : 0);
        ok = ok && threw;
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
