// C++ driver for tests/test_gpu_cpp_api.py::test_ring2k_cpp_api: examples/13_ring2k.cu -- elements of Z_{2^k} through
// PolynomialEncoderRing2k<T>: scale_up -> encrypt -> multiply by a centralized plaintext -> bfv_decrypt_without_scaling_down ->
// scale_down gives the negacyclic product mod 2^k; for T = uint32_t, uint64_t and unsigned __int128 (N=16384, six 60-bit primes).
#include <cstdio>
#include <cstring>
#include <random>
#include <sstream>

#include "../../troy-nova_amd/troy/matmul.h"
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

// y = x * w + s over Z_{2^k} through MatmulHelper's ring-2^k forms (tests of app/matmul with the Ring2k adapters), packed or not
template <typename T>
static bool run_matmul(const HeContextPointer& he, const KeyGenerator& keygen, const Encryptor& encryptor, const Decryptor& decryptor, const Evaluator& evaluator,
                       size_t plain_bits, bool pack_lwe, const char* name) {
    linear::PolynomialEncoderRing2k<T> encoder(he, plain_bits);
    const T mask = encoder.t_mask();
    const size_t M = 9, R = 40, Nn = 17, n = encoder.slot_count();
    std::mt19937_64 gen(sizeof(T) * 7 + plain_bits + (pack_lwe ? 1 : 0));
    auto rnd = [&]() { T v = static_cast<T>(gen()); if (sizeof(T) == 16) v = (v << 32 << 32) | static_cast<T>(gen()); return static_cast<T>(v & mask); };
    std::vector<T> x(M * R), w(R * Nn), sb(M * Nn), want(M * Nn, 0);
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd();
    for (auto& v : sb) v = rnd();
    for (size_t i = 0; i < M; i++) for (size_t k = 0; k < R; k++) for (size_t j = 0; j < Nn; j++) want[i * Nn + j] = static_cast<T>((want[i * Nn + j] + x[i * R + k] * w[k * Nn + j]) & mask);
    for (size_t i = 0; i < M * Nn; i++) want[i] = static_cast<T>((want[i] + sb[i]) & mask);
    linear::MatmulHelper helper(M, R, Nn, n, linear::MatmulObjective::EncryptLeft, pack_lwe);
    {
        std::stringstream text;
        text << helper;
        const std::string want_text = "MatmulHelper(batch_size=9, input_dims=40, output_dims=17, slot_count=" + std::to_string(n) + ", objective=EncryptLeft, pack_lwe=" + (pack_lwe ? "1" : "0") + ")";
        if (text.str() != want_text) { std::printf("operator<< gives %s\n", text.str().c_str()); return false; }
    }
    linear::Plain2d we = helper.encode_weights_ring2k(encoder, w.data(), std::nullopt);
    linear::Cipher2d xe = helper.encrypt_inputs_ring2k(encryptor, encoder, x.data(), std::nullopt);
    std::stringstream wire;
    xe.save(wire, he);
    xe = linear::Cipher2d::load_new(wire, he);
    linear::Cipher2d ye = helper.matmul(evaluator, xe, we);
    {
        linear::Cipher2d yf = helper.matmul_fly_ring2k(encoder, evaluator, xe, w.data(), std::nullopt);
        size_t fly_bad = 0;
        for (size_t r = 0; r < ye.data().size(); r++)
            for (size_t c = 0; c < ye[r].size(); c++) fly_bad += yf[r][c].data().to_vector() != ye[r][c].data().to_vector();
        if (fly_bad) { std::printf("matmul_fly_ring2k differs from matmul in %zu ciphertexts\n", fly_bad); return false; }
    }
    if (pack_lwe) {
        GaloisKeys autokey = keygen.create_automorphism_keys(false);
        ye = helper.pack_outputs(evaluator, autokey, ye);
    }
    linear::Plain2d se = helper.encode_outputs_ring2k(encoder, sb.data(), ye[0][0].parms_id());
    ye.add_plain_inplace(evaluator, se);
    std::stringstream ywire;
    helper.serialize_outputs(evaluator, ye, ywire);
    linear::Cipher2d yl = helper.deserialize_outputs(evaluator, ywire);
    const std::vector<T> got = helper.decrypt_outputs_ring2k(encoder, decryptor, yl);
    size_t bad = 0;
    for (size_t i = 0; i < got.size(); i++) bad += got[i] != want[i];
    std::printf("matmul %s k=%zu pack_lwe %d mismatches %zu of %zu\n", name, plain_bits, pack_lwe ? 1 : 0, bad, got.size());
    return bad == 0;
}

// test/app/bfv_ring2k.cu:119-370 replayed: scale_up -> scale_down and centralize -> decentralize give the message back, for one source and for a batch of 16
// sources of 0..15 elements; the plaintexts keep only the source's coefficients (coeff_count = source size, data = coeff_modulus_size * coeff_count);
// first and second level; plus the slice forms on device-resident sources / destinations and decentralize's correction factor
template <typename T>
static bool run_forms(const std::vector<size_t>& q_bits, const std::vector<size_t>& t_bits, const char* name) {
    bool all = true;
    for (size_t k : t_bits) {
        const size_t n = 32, B = 16;
        EncryptionParameters parms(SchemeType::BFV);
        parms.set_plain_modulus(PlainModulus::batching(n, 30));          // unused by the encoder
        parms.set_poly_modulus_degree(n);
        parms.set_coeff_modulus(CoeffModulus::create(n, q_bits));
        HeContextPointer he = HeContext::create(parms, true, SecurityLevel::Nil, 0x123);
        linear::PolynomialEncoderRing2k<T> encoder(he, k);
        he->to_device_inplace();
        encoder.to_device_inplace();
        const T mask = encoder.t_mask();
        std::mt19937_64 gen(sizeof(T) * 1000 + k);
        auto sample = [&](size_t count) { std::vector<T> v(count); for (auto& x : v) { T r = static_cast<T>(gen()); if (sizeof(T) == 16) r = (r << 32 << 32) | static_cast<T>(gen()); x = static_cast<T>(r & mask); } return v; };
        const ParmsID second = he->first_context_data_pointer()->next_context_data_pointer()->parms_id();
        auto shape = [&](const Plaintext& p, size_t cc) { return p.coeff_count() == cc && p.poly_modulus_degree() == n && p.data().size() == p.coeff_modulus_size() * cc; };
        size_t bad = 0;
        for (const std::optional<ParmsID>& level : {std::optional<ParmsID>(std::nullopt), std::optional<ParmsID>(second)}) {
            std::vector<T> m = sample(10);
            Plaintext up = encoder.scale_up_new(m, level), ce = encoder.centralize_new(m, level);
            bad += !shape(up, 10) || !shape(ce, 10) || encoder.scale_down_new(up) != m || encoder.decentralize_new(ce) != m;
            if (level.has_value()) bad += !(up.parms_id() == second);
            // a correction factor divides the decoded value (bfv_ring2k.cu:913-924)
            const T cf = static_cast<T>(sample(1)[0] | 1);
            std::vector<T> scaled(m.size());
            for (size_t i = 0; i < m.size(); i++) scaled[i] = static_cast<T>((m[i] * cf) & mask);
            bad += encoder.decentralize_new(encoder.centralize_new(scaled, level), cf) != m;
            // batches of 0 .. 15 elements
            std::vector<utils::Array<T>> message(B);
            for (size_t i = 0; i < B; i++) { message[i] = utils::Array<T>::from_vector(sample(i)); message[i].to_device_inplace(); }
            bad += !message[B - 1].on_device() || !message[B - 1].const_reference().on_device() || message[B - 1].to_host().to_vector() != message[B - 1].to_vector();   // really device-resident sources
            std::vector<Plaintext> ups(B), ces(B);
            encoder.scale_up_slice_batched(batch_utils::rcollect_const_reference<utils::Array<T>, T>(message), level, batch_utils::collect_pointer(ups));
            encoder.centralize_slice_batched(batch_utils::rcollect_const_reference<utils::Array<T>, T>(message), level, batch_utils::collect_pointer(ces));
            for (size_t i = 0; i < B; i++)
                bad += !shape(ups[i], i) || !shape(ces[i], i) || encoder.scale_down_new(ups[i]) != message[i].to_vector() || encoder.decentralize_new(ces[i]) != message[i].to_vector();
            // the batch equals the single calls word for word
            for (size_t i = 1; i < B; i++)
                bad += ups[i].data().to_vector() != encoder.scale_up_new(message[i].to_vector(), level).data().to_vector() ||
                       ces[i].data().to_vector() != encoder.centralize_slice_new(message[i].const_reference(), level).data().to_vector();
            // device-resident source and destination: elements as raw bytes in device memory
            std::vector<T> full = sample(n);
            const size_t words = (n * sizeof(T) + 7) / 8;
            std::vector<uint64_t> raw(words, 0);
            std::memcpy(raw.data(), full.data(), n * sizeof(T));
            utils::DynamicArray dev(words, true), back(words, true);
            dev.copy_from(raw.data(), words, false);
            const utils::ConstSlice<T> src(reinterpret_cast<const T*>(dev.raw_pointer()), n, true);
            Plaintext fu = encoder.scale_up_slice_new(src, level), fc = encoder.centralize_slice_new(src, level);
            bad += !shape(fu, n) || fu.data().to_vector() != encoder.scale_up_new(full, level).data().to_vector();
            encoder.decentralize_slice(fc, utils::Slice<T>(reinterpret_cast<T*>(back.raw_pointer()), n, true));
            std::vector<T> got(n);
            const std::vector<uint64_t> rb = back.to_vector();
            std::memcpy(got.data(), rb.data(), n * sizeof(T));
            const utils::Array<T> down = encoder.scale_down_slice_new(fu);
            bad += got != full || !down.on_device() || down.to_vector() != full || encoder.decentralize_slice_new(fc).to_vector() != full;
        }
        {   // utils::Array<T> on the device (box.h:300-560): clone / to_host / copy_from_slice in every host-device combination, zero-initialised construction
            const std::vector<T> v = sample(7);
            utils::Array<T> d = utils::Array<T>::from_vector(std::vector<T>(v)).to_device(), z(7, true), h(7, false);
            utils::Array<T> c = d.clone();
            h.copy_from_slice(d.const_reference());
            z.copy_from_slice(h.const_reference());
            utils::Array<T> z2(7, true);
            z2.copy_from_slice(d.const_slice(0, 7));
            bad += !d.on_device() || c.to_vector() != v || h.to_vector() != v || z.to_vector() != v || z2.to_vector() != v || utils::Array<T>(5, true).to_vector() != std::vector<T>(5, 0) ||
                   d.const_slice(2, 5).to_vector() != std::vector<T>(v.begin() + 2, v.begin() + 5);
            d.to_host_inplace();
            bad += d.on_device() || d[3] != v[3];
        }
        bool threw = false;
        try { encoder.decentralize_new(encoder.centralize_new(sample(4), std::nullopt), static_cast<T>(2)); } catch (const std::exception&) { threw = true; }
        std::printf("forms %s k=%zu mismatches %zu even_correction_rejected %d\n", name, k, bad, threw ? 1 : 0);
        all = all && bad == 0 && threw;
    }
    return all;
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
        ok = run_matmul<uint64_t>(he, keygen, encryptor, decryptor, evaluator, 64, false, "uint64") && ok;
        ok = run_matmul<uint64_t>(he, keygen, encryptor, decryptor, evaluator, 50, true, "uint64") && ok;
        ok = run_matmul<uint32_t>(he, keygen, encryptor, decryptor, evaluator, 32, true, "uint32") && ok;
        ok = run_matmul<u128>(he, keygen, encryptor, decryptor, evaluator, 100, false, "uint128") && ok;
        ok = run_forms<uint32_t>({40, 40, 40}, {32, 20, 17}, "uint32") && ok;
        ok = run_forms<uint64_t>({40, 40, 40, 40}, {64, 50, 33}, "uint64") && ok;
        ok = run_forms<u128>({60, 60, 60, 60, 60, 60}, {128, 100, 65}, "uint128") && ok;
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
