// C++ driver for tests/test_gpu_cpp_api.py: exercises troy::Evaluator (the host-side mirror of the
// reference API) the way the reference's own tests do -- build a context, move it to the device, run
// evaluator methods -- and dumps raw result words for a bit-exact comparison with the oracle.
// usage: evaluator_driver <ckks|bfv> <N> <t> <out.bin> <bits...>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

static void fill_uniform(uint64_t seed, uint64_t bound, uint64_t* out, size_t n) {   // same generator as oracle/troy_oracle.c
    uint64_t s = seed;
    for (size_t i = 0; i < n; i++) {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = (uint64_t)(((unsigned __int128)z * bound) >> 64);
    }
}

static Ciphertext random_ct(const HeContextPointer& ctx, const ParmsID& pid, uint64_t seed, size_t pcount, bool ntt, double scale) {
    const auto& q = ctx->get_context_data(pid).value()->parms().coeff_modulus();
    size_t n = ctx->get_context_data(pid).value()->parms().poly_modulus_degree(), L = q.size();
    utils::DynamicArray d(pcount * L * n, false);
    for (size_t p = 0; p < pcount; p++)
        for (size_t l = 0; l < L; l++) fill_uniform(seed * 1000003 + p * 101 + l, q[l].value(), d.raw_pointer() + (p * L + l) * n, n);
    return Ciphertext::from_members(pcount, L, n, pid, scale, ntt, 1, 0, std::move(d));
}

static RelinKeys random_relin_keys(const HeContextPointer& ctx, uint64_t seed) {
    const auto& kp = ctx->key_context_data().value()->parms();
    const auto& q = kp.coeff_modulus();
    size_t n = kp.poly_modulus_degree(), K = q.size(), L = K - 1;
    std::vector<PublicKey> vec;
    for (size_t j = 0; j < L; j++) {
        utils::DynamicArray d(2 * K * n, false);
        for (size_t c = 0; c < 2; c++)
            for (size_t l = 0; l < K; l++) fill_uniform(seed * 7919 + j * 257 + c * 31 + l, q[l].value(), d.raw_pointer() + (c * K + l) * n, n);
        vec.emplace_back(Ciphertext::from_members(2, K, n, ctx->key_parms_id(), 1.0, true, 1, 0, std::move(d)));
    }
    std::vector<std::vector<PublicKey>> keys;
    keys.push_back(std::move(vec));
    return RelinKeys(KSwitchKeys(ctx->key_parms_id(), std::move(keys)));
}

static void dump(std::ofstream& f, const Ciphertext& c) {
    std::vector<uint64_t> v = c.data().to_vector();
    uint64_t hdr[4] = {c.polynomial_count(), c.coeff_modulus_size(), c.poly_modulus_degree(), (uint64_t)c.is_ntt_form()};
    f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
    f.write(reinterpret_cast<const char*>(v.data()), v.size() * 8);
}

template <typename F>
static bool throws_invalid_argument(F&& f) {
    try { f(); } catch (const std::invalid_argument&) { return true; } catch (...) { return false; }
    return false;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::cerr << "usage: evaluator_driver <ckks|bfv> <N> <t> <out.bin> <bits...>\n"; return 2; }
    const bool ckks = std::strcmp(argv[1], "ckks") == 0, bgv = std::strcmp(argv[1], "bgv") == 0;
    const bool ntt = ckks || bgv;            // CKKS and BGV ciphertexts live in NTT form
    const size_t n = std::strtoull(argv[2], nullptr, 10);
    const uint64_t t = std::strtoull(argv[3], nullptr, 10);
    std::vector<size_t> bits;
    for (int i = 5; i < argc; i++) bits.push_back(std::strtoull(argv[i], nullptr, 10));

    EncryptionParameters parms(ckks ? SchemeType::CKKS : bgv ? SchemeType::BGV : SchemeType::BFV);
    parms.set_poly_modulus_degree(n);
    parms.set_coeff_modulus(CoeffModulus::create(n, bits));
    if (!ckks) parms.set_plain_modulus(t);
    HeContextPointer context = HeContext::create(parms, true, SecurityLevel::Nil, 0x123);
    if (!context->parameters_set()) { std::cerr << "parameters not set\n"; return 3; }
    Evaluator evaluator(context);
    ParmsID first = context->first_parms_id();
    const double scale = 1099511627776.0;   // 2^40

    int failures = 0;
    Ciphertext a = random_ct(context, first, 11, 2, ntt, ckks ? scale : 1.0);
    Ciphertext b = random_ct(context, first, 29, 2, ntt, ckks ? scale : 1.0);
    RelinKeys rk = random_relin_keys(context, 7);

    // device / host duality: the evaluator refuses host operands and an un-moved context
    if (!throws_invalid_argument([&] { evaluator.add_new(a, b); })) { std::cerr << "FAIL: host context accepted\n"; failures++; }
    context->to_device_inplace();
    if (!throws_invalid_argument([&] { evaluator.add_new(a, b); })) { std::cerr << "FAIL: host operands accepted\n"; failures++; }
    a.to_device_inplace(); b.to_device_inplace(); rk.to_device_inplace();

    std::ofstream f(argv[4], std::ios::binary);
    Ciphertext prod = evaluator.multiply_new(a, b);
    dump(f, prod);
    Ciphertext relin = evaluator.relinearize_new(prod, rk);
    dump(f, relin);
    Ciphertext next = ckks ? evaluator.rescale_to_next_new(relin) : evaluator.mod_switch_to_next_new(relin);
    dump(f, next);
    dump(f, evaluator.add_new(a, b));
    dump(f, evaluator.sub_new(a, b));
    dump(f, evaluator.negate_new(a));
    dump(f, ntt ? evaluator.transform_from_ntt_new(a) : evaluator.transform_to_ntt_new(a));
    // in-place flavours must agree with the _new flavours
    Ciphertext p2 = a;
    evaluator.multiply_inplace(p2, b);
    evaluator.relinearize_inplace(p2, rk);
    dump(f, p2);
    // add with different sizes (3 + 2 polynomials): evaluator_translate.cu:100-116
    Ciphertext a2 = a;
    a2.scale() = prod.scale();   // CKKS add requires equal scales ([Evaluator::translate_inplace] Arguments have different scales.)
    dump(f, evaluator.add_new(prod, a2));
    dump(f, evaluator.sub_new(a2, prod));
    if (ckks && !throws_invalid_argument([&] { evaluator.add_new(prod, a); })) { std::cerr << "FAIL: scale mismatch accepted\n"; failures++; }
    f.close();

    // metadata
    if (ckks) {
        if (std::fabs(prod.scale() - scale * scale) > 1.0) { std::cerr << "FAIL: product scale\n"; failures++; }
        if (next.parms_id() == relin.parms_id() || next.coeff_modulus_size() != relin.coeff_modulus_size() - 1) { std::cerr << "FAIL: rescale level\n"; failures++; }
    }
    // error behaviour mirrored from the reference
    if (!throws_invalid_argument([&] { evaluator.multiply_new(a, next); })) { std::cerr << "FAIL: parms mismatch accepted\n"; failures++; }
    if (ntt) {
        if (!throws_invalid_argument([&] { Ciphertext c = evaluator.transform_from_ntt_new(a); evaluator.multiply_new(c, c); })) { std::cerr << "FAIL: non-NTT CKKS/BGV multiply accepted\n"; failures++; }
        if (!throws_invalid_argument([&] { evaluator.transform_to_ntt_new(a); })) { std::cerr << "FAIL: double NTT accepted\n"; failures++; }
    }
    if (!ckks) {
        if (!throws_invalid_argument([&] { evaluator.rescale_to_next_new(a); })) { std::cerr << "FAIL: BFV/BGV rescale accepted\n"; failures++; }
    }
    if (bgv && (next.correction_factor() == 1 || relin.correction_factor() != 1)) { std::cerr << "FAIL: BGV correction factor bookkeeping\n"; failures++; }
    {
        Ciphertext last = a;
        if (ckks) last.scale() = 1048576.0;   // 2^20: a scale that is still within bounds at the last level (is_scale_within_bounds, evaluator_utils.h:307-323)
        while (last.parms_id() != context->last_parms_id()) evaluator.mod_switch_to_next_inplace(last);
        if (!throws_invalid_argument([&] { evaluator.mod_switch_to_next_new(last); })) { std::cerr << "FAIL: end of chain accepted\n"; failures++; }
    }
    bool oor = false;
    Ciphertext prod_small = prod, a_small = a;
    if (ckks) { prod_small.scale() = 2.0; a_small.scale() = 2.0; }   // keep the 4-polynomial product's scale within bounds (evaluator.cu:140-143)
    try { Ciphertext c4 = evaluator.multiply_new(prod_small, a_small); evaluator.relinearize_new(c4, rk); } catch (const std::out_of_range&) { oor = true; } catch (...) {}
    if (!oor) { std::cerr << "FAIL: missing relin key power not reported as out_of_range\n"; failures++; }

    MemoryPool::Destroy();
    std::cout << (failures ? "FAILED" : "OK") << std::endl;
    return failures ? 1 : 0;
}
