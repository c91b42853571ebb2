// C++ driver for tests/test_gpu_cpp_api.py::test_quickstart_cpp_api: BASELINE config 1 (the parameters and flow of
// the reference's examples/99_quickstart.cu) written against the host-side mirror troy/troy.h -- context, encoder,
// key generation, encryption, add / multiply / relinearize / mod-switch, decryption -- all on the GPU.
// Prints `key value` lines; the test compares the digests with the reference's own recorded output.
// usage: quickstart_driver <seed>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static uint64_t fnv_words(const std::vector<uint64_t>& v) {   // the digest of SURVEY.md Appendix C
    uint64_t h = 1469598103934665603ull;
    for (uint64_t w : v) { h ^= w; h *= 1099511628211ull; }
    return h;
}

static void print_slots(const char* name, const std::vector<uint64_t>& v) {
    std::printf("%s", name);
    for (size_t i = 0; i < 6; i++) std::printf(" %llu", (unsigned long long)v[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    try {
        const uint64_t seed = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 0x123;
        EncryptionParameters params(SchemeType::BFV);
        params.set_poly_modulus_degree(8192);
        params.set_coeff_modulus(CoeffModulus::create(8192, {40, 40, 40}));
        params.set_plain_modulus(PlainModulus::batching(8192, 20));
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, seed);
        context->to_device_inplace();

        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        PublicKey public_key = keygen.create_public_key(false);
        Encryptor encryptor(context);
        encryptor.set_public_key(public_key);
        Ciphertext c = encryptor.encrypt_asymmetric_new(encoder.encode_new({1, 2, 3, 4}));
        std::printf("ct_digest %016llx\n", (unsigned long long)fnv_words(c.data().to_vector()));

        Evaluator evaluator(context);
        Decryptor decryptor(context, keygen.secret_key());
        print_slots("decrypt", encoder.decode_new(decryptor.decrypt_new(c)));

        Ciphertext sum = evaluator.add_new(c, c);
        print_slots("add", encoder.decode_new(decryptor.decrypt_new(sum)));

        Ciphertext prod = evaluator.multiply_new(c, c);
        std::printf("mul_digest %016llx\n", (unsigned long long)fnv_words(prod.data().to_vector()));
        print_slots("mul", encoder.decode_new(decryptor.decrypt_new(prod)));

        RelinKeys relin_keys = keygen.create_relin_keys(false);
        Ciphertext relin = evaluator.relinearize_new(prod, relin_keys);
        std::printf("relin_polys %zu\n", relin.polynomial_count());
        print_slots("relin", encoder.decode_new(decryptor.decrypt_new(relin)));

        Ciphertext low = evaluator.mod_switch_to_next_new(relin);
        std::printf("low_limbs %zu\n", low.coeff_modulus_size());
        print_slots("modswitch", encoder.decode_new(decryptor.decrypt_new(low)));

        // symmetric encryption and a second multiplication depth-1 product of two different messages
        encryptor.set_secret_key(keygen.secret_key());
        Ciphertext d = encryptor.encrypt_symmetric_new(encoder.encode_new({5, 6, 7, 8}), false);
        Ciphertext cd = evaluator.relinearize_new(evaluator.multiply_new(c, d), relin_keys);
        print_slots("mul2", encoder.decode_new(decryptor.decrypt_new(cd)));
        Ciphertext diff = evaluator.sub_new(d, c);
        print_slots("sub", encoder.decode_new(decryptor.decrypt_new(diff)));

        // ciphertext x plaintext: multiply_plain (coefficient-form ciphertext, plaintext at parms_id_zero) and the
        // matmul-style accumulate out = c (.) w0 + d (.) w1 in NTT form
        Plaintext w0 = encoder.encode_new({3, 5, 7, 11}), w1 = encoder.encode_new({2, 2, 2, 2});
        print_slots("mulplain", encoder.decode_new(decryptor.decrypt_new(evaluator.multiply_plain_new(c, w0))));
        Ciphertext cn = evaluator.transform_to_ntt_new(c), dn = evaluator.transform_to_ntt_new(d);
        Plaintext w0n = evaluator.transform_plain_to_ntt_new(w0, c.parms_id()), w1n = evaluator.transform_plain_to_ntt_new(w1, c.parms_id());
        Ciphertext acc;
        evaluator.multiply_plain_accumulate({&cn, &dn}, {&w0n, &w1n}, {&acc, &acc}, true);
        evaluator.transform_from_ntt_inplace(acc);
        print_slots("macc", encoder.decode_new(decryptor.decrypt_new(acc)));

        // rotations with genuine Galois keys (only the power-of-two steps are generated: 3 = 4 - 1 goes through NAF)
        GaloisKeys galois_keys = keygen.create_galois_keys(false);
        std::vector<uint64_t> ramp(8192);
        for (size_t i = 0; i < ramp.size(); i++) ramp[i] = i + 1;
        Ciphertext r = encryptor.encrypt_asymmetric_new(encoder.encode_new(ramp));
        print_slots("rot1", encoder.decode_new(decryptor.decrypt_new(evaluator.rotate_rows_new(r, 1, galois_keys))));
        print_slots("rot3", encoder.decode_new(decryptor.decrypt_new(evaluator.rotate_rows_new(r, 3, galois_keys))));
        print_slots("rotm2", encoder.decode_new(decryptor.decrypt_new(evaluator.rotate_rows_new(r, -2, galois_keys))));
        print_slots("rotcol", encoder.decode_new(decryptor.decrypt_new(evaluator.rotate_columns_new(r, galois_keys))));

        // serialization in the reference's raw format: round trips, a seed-compressed symmetric ciphertext, key sets
        {
            std::stringstream ss;
            const size_t ct_bytes = c.save(ss, context);
            Ciphertext c_back = Ciphertext::load_new(ss, context);
            std::printf("ser_ct %zu %d\n", ct_bytes, (int)(c_back.data().to_vector() == c.data().to_vector() && c_back.parms_id() == c.parms_id()));
            std::stringstream sk_s, rk_s, pt_s, pr_s, seeded_s;
            keygen.secret_key().save(sk_s);
            relin_keys.save(rk_s, context);
            w0.save(pt_s);
            params.save(pr_s);
            SecretKey sk_back = SecretKey::load_new(sk_s);
            RelinKeys rk_back; rk_back.load(rk_s, context);
            Plaintext w0_back = Plaintext::load_new(pt_s);
            EncryptionParameters pr_back; pr_back.load(pr_s);
            std::printf("ser_params %d\n", (int)(pr_back.parms_id() == params.parms_id()));
            Decryptor dec2(context, sk_back);
            print_slots("ser_relin", encoder.decode_new(dec2.decrypt_new(evaluator.relinearize_new(prod, rk_back))));
            print_slots("ser_plain", encoder.decode_new(w0_back));
            Ciphertext seeded = encryptor.encrypt_symmetric_new(encoder.encode_new({9, 8, 7}), true);
            const size_t seeded_bytes = seeded.save(seeded_s, context);
            std::printf("ser_seeded %zu %d\n", seeded_bytes, (int)seeded.contains_seed());
            Ciphertext seeded_back = Ciphertext::load_new(seeded_s, context);
            print_slots("ser_seeded_dec", encoder.decode_new(decryptor.decrypt_new(seeded_back)));
            if (argc > 2) { std::ofstream f(argv[2], std::ios::binary); c.save(f, context); }
        }

        // misuse: host-resident plaintext / ciphertext
        bool threw = false;
        try { Plaintext p = encoder.encode_new({1}); p.to_host_inplace(); encryptor.encrypt_asymmetric_new(p); } catch (const std::invalid_argument&) { threw = true; }
        std::printf("host_plain_rejected %d\n", threw ? 1 : 0);
        std::printf("OK\n");
        MemoryPool::Destroy();
        return 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
