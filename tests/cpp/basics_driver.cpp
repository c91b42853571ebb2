// C++ driver for tests/test_gpu_cpp_api.py::test_basics_cpp_api, in the spirit of examples/1_bfv_basics.cu and 3_levels.cu:
// SEAL's default parameters (CoeffModulus::bfv_default), context qualifiers, the invariant noise budget along a computation
// (fresh > after a multiplication > ... > 0, at which point decryption stops being correct), modulus switching down the chain.
#include <cstdio>
#include <sstream>
#include <random>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

int main() {
    try {
        const size_t n = 4096;
        const uint64_t t = 1024;                                     // the example's non-batching plain modulus
        EncryptionParameters params(SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::bfv_default(n));
        params.set_plain_modulus(t);
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Classical128, 0x1b);
        ContextDataPointer key = context->key_context_data().value(), first = context->first_context_data().value();
        std::printf("chain %zu primes, key level %zu bits, first level %zu bits\n", key->parms().coeff_modulus().size(), key->total_coeff_modulus_bit_count(),
                    first->total_coeff_modulus_bit_count());
        std::printf("qualifiers batching %d fast_plain_lift %d descending %d security %d\n", first->qualifiers().using_batching ? 1 : 0,
                    first->qualifiers().using_fast_plain_lift ? 1 : 0, first->qualifiers().using_descending_modulus_chain ? 1 : 0,
                    static_cast<int>(first->qualifiers().security_level));
        context->to_device_inplace();
        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false));
        Decryptor decryptor(context, keygen.secret_key());
        Evaluator ev(context);
        RelinKeys rk = keygen.create_relin_keys(false);

        // x = 6 as the constant polynomial; evaluate (x^2 + 1) and (x + 1)^2, then their product 4x^4 + 8x^3 + 8x^2 + 8x + 4 = 4 (x^2+1)(x+1)^2
        const uint64_t x = 6;
        Ciphertext cx = encryptor.encrypt_asymmetric_new(encoder.encode_polynomial_new({x}));
        const size_t fresh = decryptor.invariant_noise_budget(cx);
        Plaintext one = encoder.encode_polynomial_new({1}), four = encoder.encode_polynomial_new({4});
        Ciphertext a = ev.add_plain_new(ev.square_new(cx), one);      // x^2 + 1, three polynomials
        const size_t after_square = decryptor.invariant_noise_budget(a);
        ev.relinearize_inplace(a, rk);
        const size_t after_relin = decryptor.invariant_noise_budget(a);
        Ciphertext b = ev.add_plain_new(cx, one);
        ev.square_inplace(b);
        ev.relinearize_inplace(b, rk);
        Ciphertext r = ev.multiply_new(a, b);
        ev.relinearize_inplace(r, rk);
        ev.multiply_plain_inplace(r, four);
        const size_t final_budget = decryptor.invariant_noise_budget(r);
        const uint64_t want = (4 * x * x * x * x + 8 * x * x * x + 8 * x * x + 8 * x + 4) % t;
        const uint64_t got = encoder.decode_polynomial_new(decryptor.decrypt_new(r))[0];
        std::printf("noise budget fresh %zu after_square %zu after_relinearize %zu final %zu\n", fresh, after_square, after_relin, final_budget);
        std::printf("polynomial got %llu want %llu\n", (unsigned long long)got, (unsigned long long)want);
        bool ok = got == want && fresh > after_square && after_square >= after_relin && after_relin > final_budget && final_budget > 0;

        // 3_levels: the budget shrinks with the modulus, the value survives
        Ciphertext low = ev.mod_switch_to_next_new(cx);
        const size_t low_budget = decryptor.invariant_noise_budget(low);
        std::printf("mod_switch level %zu budget %zu value %llu\n", context->get_context_data(low.parms_id()).value()->chain_index(), low_budget,
                    (unsigned long long)encoder.decode_polynomial_new(decryptor.decrypt_new(low))[0]);
        ok = ok && low_budget < fresh && low_budget > 0 && encoder.decode_polynomial_new(decryptor.decrypt_new(low))[0] == x;

        // exhaust the budget: repeated squaring ends at 0 bits and a wrong decryption
        Ciphertext s = cx;
        size_t budget = fresh, steps = 0;
        while (budget > 0 && steps < 12) { ev.square_inplace(s); ev.relinearize_inplace(s, rk); budget = decryptor.invariant_noise_budget(s); steps++; }
        std::printf("budget exhausted after %zu squarings (budget %zu)\n", steps, budget);
        ok = ok && budget == 0 && steps >= 2 && steps < 12;
        // utils::Slice / ConstSlice views of a ciphertext (ciphertext.h:211-252, utils/box.h:262-308): c0 + c1 built by hand through
        // the views equals Evaluator::add of two ciphertexts that share c0 and have complementary halves
        {
            Ciphertext u = cx, v = cx;
            utils::Slice<uint64_t> u1 = u.poly(1);
            const Ciphertext& cu = u;
            utils::ConstSlice<uint64_t> c0 = cu.poly(0), comp = cu.poly_component(1, 1), both = cu.polys(0, 2);
            const size_t degree = u.poly_modulus_degree(), limbs = u.coeff_modulus_size();
            bool views = u1.size() == degree * limbs && u1.on_device() && c0.size() == u1.size() && comp.size() == degree && both.size() == 2 * u1.size() &&
                         comp.raw_pointer() == u1.raw_pointer() + degree && both.raw_pointer() == c0.raw_pointer() && u1.slice(degree, 2 * degree).raw_pointer() == comp.raw_pointer();
            v.poly(1).set_zero();                              // v = (c0, 0): decrypts to the phase of c0 alone, not to x
            v.poly(1).copy_from_slice(cu.const_poly(1));       // ... and back: v == u again
            views = views && encoder.decode_polynomial_new(decryptor.decrypt_new(v))[0] == x;
            std::vector<uint64_t> host(degree);
            utils::Slice<uint64_t>(host.data(), degree, false).copy_from_slice(comp);    // device -> host through the views
            Ciphertext h = cx.to_host();
            views = views && host[5] == h.poly_component(1, 1)[5] && !h.poly(0).on_device();
            std::printf("slice views %d\n", (int)views);
            ok = ok && views;
        }
        // batch_utils collectors, Ciphertext::resize / reconfigure_like with the reference's argument meaning, Modulus::const_ratio as a slice, operator<<
        {
            std::vector<Ciphertext> batch{cx, cx, cx};
            const auto c1s = batch_utils::rcollect_const_poly(batch, 1);
            const auto both = batch_utils::rcollect_polys(batch, 0, 2);
            const auto ptrs = batch_utils::collect_const_pointer(batch);
            const auto c0s = batch_utils::pcollect_const_poly(ptrs, 0);
            const size_t d = cx.poly_modulus_degree() * cx.coeff_modulus_size();
            bool surf = c1s.size() == 3 && c1s[2].size() == d && c1s[1].raw_pointer() == batch[1].poly(1).raw_pointer() && both[0].size() == 2 * d && c0s[2].raw_pointer() == batch[2].data().raw_pointer();
            surf = surf && batch_utils::rcollect_as_const(both)[1].raw_pointer() == both[1].raw_pointer() && batch_utils::clone(batch).size() == 3 && batch_utils::rcollect_const_reference(batch)[0].size() == 2 * d;
            Ciphertext grown = cx;
            grown.resize(context, cx.parms_id(), 3);                                   // defaults: the old polynomials are kept, the new one is zero
            surf = surf && grown.polynomial_count() == 3 && grown.polys(0, 2).to_vector() == cx.data().to_vector();
            bool zero = true;
            for (uint64_t w : grown.poly(2).to_vector()) zero = zero && w == 0;
            surf = surf && zero && !grown.is_transparent();
            Ciphertext raw = cx;
            raw.resize(context, cx.parms_id(), 3, false, true);                        // the new polynomial is left as allocated; the old ones are still there
            surf = surf && raw.polys(0, 2).to_vector() == cx.data().to_vector();
            Ciphertext like;
            like.data() = utils::DynamicArray(0, true);
            like.scale() = 0.5;
            like.reconfigure_like(context, cx, 2);
            surf = surf && like.parms_id() == cx.parms_id() && like.polynomial_count() == 2 && like.is_ntt_form() == cx.is_ntt_form() && like.scale() == cx.scale() && like.is_transparent();
            bool threw = false;
            try { like.resize(context, cx.parms_id(), 1); } catch (const std::invalid_argument&) { threw = true; }
            const Modulus q0 = context->first_context_data().value()->parms().coeff_modulus()[0];
            const unsigned __int128 ratio = (static_cast<unsigned __int128>(q0.const_ratio()[1]) << 64) | q0.const_ratio()[0];
            surf = surf && threw && q0.const_ratio().size() == 3 && ratio == (~static_cast<unsigned __int128>(0)) / q0.value();     // floor(2^128 / q) for an odd q
            // EncryptionParameters::plain_modulus(): the reference's pointer style (->, *) and the value style both compile and agree
            const EncryptionParameters& kp = context->key_context_data().value()->parms();
            const Modulus& by_ref = kp.plain_modulus();
            surf = surf && kp.plain_modulus()->value() == kp.plain_modulus().value() && (*kp.plain_modulus()).bit_count() == by_ref.bit_count() && !kp.plain_modulus().is_null() &&
                   by_ref.value() == kp.plain_modulus_host().value();
            std::stringstream text;
            text << q0;
            surf = surf && text.str() == "Modulus(" + std::to_string(q0.value()) + ")";
            std::printf("collectors, resize, reconfigure_like, const_ratio %d\n", (int)surf);
            ok = ok && surf;
        }
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
