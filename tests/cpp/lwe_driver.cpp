// C++ driver for tests/test_gpu_cpp_api.py::test_lwe_cpp_api: the reference's test/lwe.cu through the mirror -- extract_lwe / assemble_lwe,
// pack_lwe_ciphertexts(_batched), pack_rlwe_ciphertexts(_batched) for BFV and BGV (lwe.cu:13-37 test_extract_lwe, :64-99 test_pack_lwes, :130-175
// test_pack_lwes_batched, :212-245 test_pack_rlwes, :280-325 test_pack_rlwes_batched), with the reference's parameter sets (N = 32, {60,40,40,60}, 20-bit t)
// and a larger ring.  Every result is decrypted and compared with the expected polynomial.   usage: lwe_driver <bfv|bgv> <N>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static size_t failures = 0;
static void check(const char* name, const std::vector<uint64_t>& got, const std::vector<uint64_t>& want) {
    size_t bad = got.size() != want.size();
    for (size_t i = 0; i < got.size() && i < want.size(); i++) bad += got[i] != want[i];
    std::printf("%s %zu\n", name, bad);
    failures += bad != 0;
}

int main(int argc, char** argv) {
    try {
        const bool bgv = argc > 1 && std::strcmp(argv[1], "bgv") == 0;
        const size_t n = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 32;
        EncryptionParameters params(bgv ? SchemeType::BGV : SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        params.set_plain_modulus(PlainModulus::batching(n, 20));
        const uint64_t t = params.plain_modulus().value();
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0x123);
        context->to_device_inplace();
        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false));
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator ev(context);
        GaloisKeys auto_key = keygen.create_automorphism_keys(false);
        std::mt19937_64 gen(7);
        auto random_poly = [&] { std::vector<uint64_t> v(n); for (auto& x : v) x = gen() % t; return v; };
        auto enc = [&](const std::vector<uint64_t>& m) { return encryptor.encrypt_asymmetric_new(encoder.encode_polynomial_new(m)); };
        auto dec = [&](const Ciphertext& c) { return encoder.decode_polynomial_new(decryptor.decrypt_new(c)); };
        std::printf("scheme %s n %zu t %llu\n", bgv ? "bgv" : "bfv", n, (unsigned long long)t);

        {   // the encoders' *_slice forms (batch_encoder.h:55-131): host and device views give the words of the vector forms
            const std::vector<uint64_t> m = random_poly();
            const Plaintext want = encoder.encode_new(m), want_poly = encoder.encode_polynomial_new(m);
            utils::DynamicArray dm = utils::DynamicArray::from_vector(m);
            dm.to_device_inplace(MemoryPool::GlobalPool());
            const utils::ConstSlice<uint64_t> host_view(m.data(), m.size(), false), dev_view(dm.raw_pointer(), dm.size(), true);
            check("encode_slice_host", encoder.encode_slice_new(host_view).data().to_vector(), want.data().to_vector());
            check("encode_slice_device", encoder.encode_slice_new(dev_view).data().to_vector(), want.data().to_vector());
            check("encode_polynomial_slice_device", encoder.encode_polynomial_slice_new(dev_view).data().to_vector(), want_poly.data().to_vector());
            check("decode_slice_new", encoder.decode_slice_new(want).to_vector(), m);
            utils::DynamicArray back(n, true, MemoryPool::GlobalPool());
            encoder.decode_slice(want, utils::Slice<uint64_t>(back.raw_pointer(), back.size(), true));
            check("decode_slice_device", back.to_vector(), m);
            std::vector<uint64_t> hb(n);
            encoder.decode_polynomial_slice(want_poly, utils::Slice<uint64_t>(hb.data(), hb.size(), false));
            check("decode_polynomial_slice_host", hb, m);
            std::vector<Plaintext> many = encoder.encode_slice_new_batched({host_view, dev_view});
            check("encode_slice_new_batched", many[1].data().to_vector(), want.data().to_vector());
        }
        {   // test_extract_lwe
            const std::vector<uint64_t> m = random_poly();
            Ciphertext c = enc(m);
            for (size_t term : {size_t(0), size_t(1), size_t(3), size_t(7)}) {
                LWECiphertext lwe = ev.extract_lwe_new(c, term);
                Ciphertext back = ev.assemble_lwe_new(lwe);
                if (bgv) ev.transform_to_ntt_inplace(back);
                const std::vector<uint64_t> got = dec(back);
                check("extract_lwe", {got[0]}, {m[term]});
            }
        }
        auto pack_lwes = [&](size_t count, bool batched) {
            std::vector<std::vector<uint64_t>> msgs(count);
            std::vector<LWECiphertext> lwes;
            for (size_t i = 0; i < count; i++) { msgs[i] = random_poly(); lwes.push_back(ev.extract_lwe_new(enc(msgs[i]), 0)); }
            size_t r = 1; while (r < count) r *= 2;
            const size_t interval = n / r;
            std::vector<uint64_t> truth(n, 0);
            for (size_t i = 0; i < count; i++) truth[i * interval] = msgs[i][0];
            std::vector<const LWECiphertext*> ptrs;
            for (const LWECiphertext& l : lwes) ptrs.push_back(&l);
            if (!batched) check("pack_lwes", dec(ev.pack_lwe_ciphertexts_new(ptrs, auto_key)), truth);
            else {
                // two groups: the full set and its first three members (lwe.cu test_pack_lwes_batched packs groups of different sizes)
                const size_t small = std::min<size_t>(3, count);
                std::vector<const LWECiphertext*> few(ptrs.begin(), ptrs.begin() + static_cast<std::ptrdiff_t>(small));
                std::vector<Ciphertext> out = ev.pack_lwe_ciphertexts_new_batched({ptrs, few}, auto_key);
                check("pack_lwes_batched_full", dec(out[0]), truth);
                std::vector<uint64_t> truth2(n, 0);
                for (size_t i = 0; i < small; i++) truth2[i * interval] = msgs[i][0];       // the tree depth is set by the largest group
                check("pack_lwes_batched_few", dec(out[1]), truth2);
            }
        };
        pack_lwes(std::min<size_t>(n, 32), false);
        pack_lwes(7, false);
        pack_lwes(std::min<size_t>(n, 32), true);
        pack_lwes(7, true);

        auto pack_rlwes = [&](size_t count, size_t input_interval, size_t output_interval, int shift_, bool batched) {
            const size_t shift = 2 * n + shift_;        // shift_ <= 0
            std::vector<std::vector<uint64_t>> msgs(count);
            std::vector<Ciphertext> cts;
            for (size_t i = 0; i < count; i++) { msgs[i] = random_poly(); cts.push_back(enc(msgs[i])); }
            std::vector<uint64_t> truth(n, 0);
            for (size_t i = 0; i < count; i++)
                for (size_t j = 0; j < n; j += input_interval) truth[i * output_interval + j] = msgs[i][j - shift_];
            std::vector<const Ciphertext*> ptrs;
            for (const Ciphertext& c : cts) ptrs.push_back(&c);
            if (!batched) check("pack_rlwes", dec(ev.pack_rlwe_ciphertexts_new(ptrs, auto_key, shift, input_interval, output_interval)), truth);
            else {
                std::vector<Ciphertext> out = ev.pack_rlwe_ciphertexts_new_batched({ptrs, ptrs}, auto_key, shift, input_interval, output_interval);
                check("pack_rlwes_batched_0", dec(out[0]), truth);
                check("pack_rlwes_batched_1", dec(out[1]), truth);
            }
        };
        for (int batched = 0; batched < 2; batched++) {
            pack_rlwes(32, 32, 1, 0, batched != 0);
            pack_rlwes(16, 16, 1, 0, batched != 0);
            pack_rlwes(3, 8, 2, -3, batched != 0);
        }
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
