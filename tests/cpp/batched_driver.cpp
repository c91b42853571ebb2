// C++ driver for tests/test_gpu_cpp_api.py::test_batched_encrypt_decrypt_cpp_api: Encryptor::encrypt_symmetric_batched produces,
// bit for bit, the ciphertexts of encrypt_symmetric called once per plaintext on an identically seeded context, and
// Decryptor::decrypt_batched the plaintexts of decrypt called once per ciphertext.
// usage: batched_driver <poly_degree> <count>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

struct Party {
    HeContextPointer context;
    std::unique_ptr<BatchEncoder> encoder;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<Encryptor> encryptor;
    std::unique_ptr<Decryptor> decryptor;
    Party(size_t n, uint64_t t, uint64_t seed) {
        EncryptionParameters params(SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        params.set_plain_modulus(t);
        context = HeContext::create(params, true, SecurityLevel::Nil, seed);
        context->to_device_inplace();
        encoder = std::make_unique<BatchEncoder>(context);
        keygen = std::make_unique<KeyGenerator>(context);
        encryptor = std::make_unique<Encryptor>(context);
        encryptor->set_secret_key(keygen->secret_key());
        decryptor = std::make_unique<Decryptor>(context, keygen->secret_key());
    }
};

int main(int argc, char** argv) {
    try {
        const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 8192, count = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 19;
        const uint64_t t = 1ull << 21;
        Party a(n, t, 0x2024), b(n, t, 0x2024);
        std::mt19937_64 gen(5);
        std::vector<Plaintext> pa, pb;
        for (size_t i = 0; i < count; i++) {
            std::vector<uint64_t> v(i % 3 == 0 ? n : n / 2 + i);      // ragged coefficient counts
            for (auto& x : v) x = gen() % t;
            pa.push_back(a.encoder->encode_polynomial_new(v));
            pb.push_back(b.encoder->encode_polynomial_new(v));
        }
        std::vector<Ciphertext> ca, cb(count);
        for (size_t i = 0; i < count; i++) ca.push_back(a.encryptor->encrypt_symmetric_new(pa[i], false));
        std::vector<const Plaintext*> pp;
        std::vector<Ciphertext*> cp;
        for (size_t i = 0; i < count; i++) { pp.push_back(&pb[i]); cp.push_back(&cb[i]); }
        b.encryptor->encrypt_symmetric_batched(pp, false, cp);
        size_t enc_bad = 0;
        for (size_t i = 0; i < count; i++) {
            enc_bad += ca[i].data().to_vector() != cb[i].data().to_vector();
            enc_bad += ca[i].parms_id() != cb[i].parms_id() || cb[i].is_ntt_form() || cb[i].polynomial_count() != 2;
        }
        // the generators are left at the same position: one more sequential encryption on each side agrees
        enc_bad += a.encryptor->encrypt_symmetric_new(pa[0], false).data().to_vector() != b.encryptor->encrypt_symmetric_new(pb[0], false).data().to_vector();
        std::printf("encrypt_mismatches %zu of %zu\n", enc_bad, count);

        std::vector<Plaintext> da, db(count);
        for (size_t i = 0; i < count; i++) da.push_back(a.decryptor->decrypt_new(ca[i]));
        std::vector<const Ciphertext*> ccp;
        std::vector<Plaintext*> dp;
        for (size_t i = 0; i < count; i++) { ccp.push_back(&cb[i]); dp.push_back(&db[i]); }
        b.decryptor->decrypt_batched(ccp, dp);
        size_t dec_bad = 0;
        for (size_t i = 0; i < count; i++) {
            dec_bad += da[i].data().to_vector() != db[i].data().to_vector();
            std::vector<uint64_t> got = b.encoder->decode_polynomial_new(db[i]), want = a.encoder->decode_polynomial_new(pa[i]);
            got.resize(n, 0); want.resize(n, 0);
            dec_bad += got != want;
        }
        std::printf("decrypt_mismatches %zu of %zu\n", dec_bad, count);
        // mixed input (one NTT-form ciphertext) takes the per-ciphertext path and still throws what decrypt throws
        bool threw = false;
        {
            Evaluator ev(b.context);
            Ciphertext c = cb[0].clone();
            ev.transform_to_ntt_inplace(c);
            std::vector<const Ciphertext*> m = {&cb[1], &c};
            Plaintext p0, p1;
            std::vector<Plaintext*> d = {&p0, &p1};
            try { b.decryptor->decrypt_batched(m, d); } catch (const std::invalid_argument&) { threw = true; }
        }
        std::printf("ntt_form_rejected %d\n", threw ? 1 : 0);
        const bool ok = enc_bad == 0 && dec_bad == 0 && threw;
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
