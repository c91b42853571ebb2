// C++ driver for tests/test_gpu_cpp_api.py::test_bgv_cpp_api: the BGV scheme through the mirror API -- BatchEncoder, KeyGenerator,
// Encryptor (asymmetric and symmetric), Evaluator (add, multiply, relinearize, mod_switch_to_next, rotate_rows, add_plain,
// multiply_plain, additions of operands with different correction factors), Decryptor -- every result decrypted and compared
// with the plain computation on the slots.
#include <cstdio>
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

int main() {
    try {
        const size_t n = 8192;
        EncryptionParameters params(SchemeType::BGV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {50, 50, 50, 50}));
        params.set_plain_modulus(PlainModulus::batching(n, 20));
        const uint64_t t = params.plain_modulus().value();
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0xb6f);
        context->to_device_inplace();
        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false));
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator ev(context);
        RelinKeys rk = keygen.create_relin_keys(false);
        GaloisKeys gk = keygen.create_galois_keys_from_steps({1}, false);

        std::mt19937_64 gen(9);
        std::vector<uint64_t> a(n), b(n), sum(n), prod(n), rot(n), apl(n), mpl(n), mixed(n);
        for (size_t i = 0; i < n; i++) { a[i] = gen() % t; b[i] = gen() % t; }
        auto mulmod = [t](uint64_t x, uint64_t y) { return (uint64_t)(((unsigned __int128)x * y) % t); };
        for (size_t i = 0; i < n; i++) {
            sum[i] = (a[i] + b[i]) % t;
            prod[i] = mulmod(a[i], b[i]);
            const size_t row = i / (n / 2), col = i % (n / 2);
            rot[i] = a[row * (n / 2) + (col + 1) % (n / 2)];
            apl[i] = (prod[i] + b[i]) % t;
            mpl[i] = mulmod(a[i], b[i]);
            mixed[i] = (prod[i] + a[i]) % t;
        }
        auto dec = [&](const Ciphertext& c) { return encoder.decode_new(decryptor.decrypt_new(c)); };
        Plaintext pa = encoder.encode_new(a), pb = encoder.encode_new(b);
        Ciphertext ca = encryptor.encrypt_asymmetric_new(pa), cb = encryptor.encrypt_symmetric_new(pb, false);
        std::printf("form ntt=%d cf=%llu L=%zu\n", ca.is_ntt_form() ? 1 : 0, (unsigned long long)ca.correction_factor(), ca.coeff_modulus_size());
        check("decrypt_asymmetric", dec(ca), a);
        check("decrypt_symmetric", dec(cb), b);
        check("add", dec(ev.add_new(ca, cb)), sum);
        Ciphertext m3 = ev.multiply_new(ca, cb);
        check("multiply", dec(m3), prod);
        Ciphertext m = ev.relinearize_new(m3, rk);
        check("relinearize", dec(m), prod);
        Ciphertext low = ev.mod_switch_to_next_new(m);
        std::printf("mod_switch L=%zu cf=%llu\n", low.coeff_modulus_size(), (unsigned long long)low.correction_factor());
        check("mod_switch_to_next", dec(low), prod);
        Ciphertext low2 = ev.mod_switch_to_next_new(low);
        check("mod_switch_to_next(2)", dec(low2), prod);
        check("rotate_rows", dec(ev.rotate_rows_new(ca, 1, gk)), rot);
        check("add_plain(after mod switch)", dec(ev.add_plain_new(low, pb)), apl);
        check("multiply_plain", dec(ev.multiply_plain_new(ca, pb)), mpl);
        // operands with different correction factors: the mod-switched product and a mod_switch_to'd fresh ciphertext
        Ciphertext a_low = ev.mod_switch_to_new(ca, low.parms_id());
        std::printf("correction factors %llu %llu\n", (unsigned long long)low.correction_factor(), (unsigned long long)a_low.correction_factor());
        Ciphertext mix = ev.add_new(low, a_low);
        check("add(balanced correction factors)", dec(mix), mixed);
        // genuinely different factors: a product of two mod-switched ciphertexts carries c^2, the fresh operand c
        Ciphertext sq = ev.relinearize_new(ev.multiply_new(low, a_low), rk);
        std::printf("correction factors %llu %llu\n", (unsigned long long)sq.correction_factor(), (unsigned long long)a_low.correction_factor());
        failures += sq.correction_factor() == a_low.correction_factor();
        std::vector<uint64_t> want2(n), want3(n);
        for (size_t i = 0; i < n; i++) { want2[i] = (mulmod(prod[i], a[i]) + a[i]) % t; want3[i] = (mulmod(prod[i], a[i]) + t - a[i]) % t; }
        check("add(different correction factors)", dec(ev.add_new(sq, a_low)), want2);
        check("sub(different correction factors)", dec(ev.sub_new(sq, a_low)), want3);
        // wire format carries the correction factor
        std::stringstream ss;
        low.save(ss, context);
        Ciphertext back = Ciphertext::load_new(ss, context);
        std::printf("serialized_cf_equal %d\n", back.correction_factor() == low.correction_factor() ? 1 : 0);
        failures += back.correction_factor() != low.correction_factor();
        check("decrypt(loaded)", dec(back), prod);
        std::printf(failures ? "FAIL\n" : "OK\n");
        MemoryPool::Destroy();
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
