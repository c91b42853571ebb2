// C++ driver for tests/test_gpu_cpp_api.py::test_every_call_uses_the_pool_it_is_given: the idea of the reference's test/multithread.cu:1250-1340
// (SharedContextMultiPools: test_troublesome_pools).  One context, keys and Evaluator on a context pool; then the CONTEXT pool AND the GLOBAL pool are set to deny
// (MemoryPool::deny, utils/memory_pool.h:100): from there on any allocation that does not come from the pool handed to the call throws.  Four host threads, each
// with a pool of its own, run the Evaluator / Encryptor / Decryptor / encoder / KeyGenerator surface with that pool: every result must report that pool, must be
// correct, and nothing may throw.  usage: pools_driver <bfv|bgv|ckks>
#include <atomic>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;
using cd = std::complex<double>;

static std::mutex print_mutex;
static std::atomic<size_t> failures{0};
static void report(size_t thread, const char* what, const std::string& why) {
    std::lock_guard<std::mutex> lock(print_mutex);
    std::printf("thread %zu %s: %s\n", thread, what, why.c_str());
    failures++;
}

int main(int argc, char** argv) {
    try {
        const std::string sch = argc > 1 ? argv[1] : "bfv";
        const SchemeType scheme = sch == "bgv" ? SchemeType::BGV : sch == "ckks" ? SchemeType::CKKS : SchemeType::BFV;
        const bool ckks = scheme == SchemeType::CKKS;
        const size_t n = 4096;
        MemoryPoolHandle context_pool = MemoryPool::create(0);
        EncryptionParameters params(scheme);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {60, 40, 40, 60}));
        if (!ckks) params.set_plain_modulus(PlainModulus::batching(n, 20));
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0x123);
        context->to_device_inplace(context_pool);
        const double scale = std::pow(2.0, 30);
        const uint64_t t = ckks ? 0 : params.plain_modulus().value();
        std::unique_ptr<BatchEncoder> benc;
        std::unique_ptr<CKKSEncoder> cenc;
        if (ckks) cenc.reset(new CKKSEncoder(context)); else benc.reset(new BatchEncoder(context));
        KeyGenerator keygen(context, context_pool);
        Encryptor encryptor(context);
        encryptor.set_public_key(keygen.create_public_key(false, context_pool));
        encryptor.set_secret_key(keygen.secret_key());
        Decryptor decryptor(context, keygen.secret_key(), context_pool);
        Evaluator ev(context);
        RelinKeys rk = keygen.create_relin_keys(false, 2, context_pool);
        GaloisKeys gk = keygen.create_galois_keys(false, context_pool);
        GaloisKeys ak = keygen.create_automorphism_keys(false, context_pool);
        {   // expand the secret-key powers and every lazily built per-level handle before the pools are closed (the reference does the same, multithread.cu:1263-1268)
            Ciphertext c = encryptor.encrypt_zero_asymmetric_new(std::nullopt, nullptr, context_pool);
            if (ckks) c.scale() = scale;
            ev.square_inplace(c, context_pool);
            decryptor.decrypt_new(c, context_pool);
        }
        utils::stream_sync();
        context_pool->deny();
        MemoryPool::GlobalPool()->deny();

        auto body = [&](size_t id) {
            MemoryPoolHandle pool = MemoryPool::create(0);
            std::mt19937_64 gen(100 + id);
            const char* at = "start";
            try {
                auto good = [&](const char* what, MemoryPoolHandle got) { if (got != pool) report(id, what, "result is not in the pool given to the call"); };
                // ---- messages, encoders -------------------------------------------------------------------------------------------------
                std::vector<uint64_t> m1, m2;
                std::vector<cd> z1, z2;
                Plaintext p1, p2;
                if (ckks) {
                    std::uniform_real_distribution<double> U(-4.0, 4.0);
                    z1.resize(cenc->slot_count()); z2.resize(cenc->slot_count());
                    for (auto& v : z1) v = {U(gen), U(gen)};
                    for (auto& v : z2) v = {U(gen), U(gen)};
                    at = "encode_complex64_simd"; p1 = cenc->encode_complex64_simd_new(z1, std::nullopt, scale, pool); p2 = cenc->encode_complex64_simd_new(z2, std::nullopt, scale, pool);
                } else {
                    m1.resize(benc->slot_count()); m2.resize(benc->slot_count());
                    for (auto& v : m1) v = gen() % t;
                    for (auto& v : m2) v = gen() % t;
                    at = "encode"; p1 = benc->encode_new(m1, pool); p2 = benc->encode_new(m2, pool);
                }
                good("encode", p1.pool());
                auto decode_check = [&](const char* what, const Ciphertext& c, auto&& expect_u, auto&& expect_z, double tol = 1e-2) {
                    good(what, c.pool());
                    Plaintext d = decryptor.decrypt_new(c, pool);
                    good("decrypt_new", d.pool());
                    if (ckks) {
                        const std::vector<cd> got = cenc->decode_complex64_simd_new(d, pool);
                        for (size_t i = 0; i < got.size(); i++) if (!(std::abs(got[i] - expect_z(i)) < tol)) { report(id, what, "wrong value"); return; }
                    } else {
                        const std::vector<uint64_t> got = benc->decode_new(d, pool);
                        for (size_t i = 0; i < got.size(); i++) if (got[i] != expect_u(i)) { report(id, what, "wrong value"); return; }
                    }
                };
                // ---- encryption ---------------------------------------------------------------------------------------------------------
                at = "encrypt_asymmetric_new"; Ciphertext c1 = encryptor.encrypt_asymmetric_new(p1, nullptr, pool);
                at = "encrypt_symmetric_new"; Ciphertext c2 = encryptor.encrypt_symmetric_new(p2, false, nullptr, pool);
                decode_check("encrypt_asymmetric_new", c1, [&](size_t i) { return m1[i]; }, [&](size_t i) { return z1[i]; });
                decode_check("encrypt_symmetric_new", c2, [&](size_t i) { return m2[i]; }, [&](size_t i) { return z2[i]; });
                at = "encrypt_zero_asymmetric_new"; good("encrypt_zero_asymmetric_new", encryptor.encrypt_zero_asymmetric_new(std::nullopt, nullptr, pool).pool());
                at = "encrypt_zero_symmetric_new"; good("encrypt_zero_symmetric_new", encryptor.encrypt_zero_symmetric_new(false, std::nullopt, nullptr, pool).pool());
                auto mulm = [&](uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) % t); };
                // ---- evaluator ----------------------------------------------------------------------------------------------------------
                at = "negate_new"; decode_check("negate_new", ev.negate_new(c1, pool), [&](size_t i) { return (t - m1[i]) % t; }, [&](size_t i) { return -z1[i]; });
                at = "add_new"; decode_check("add_new", ev.add_new(c1, c2, pool), [&](size_t i) { return (m1[i] + m2[i]) % t; }, [&](size_t i) { return z1[i] + z2[i]; });
                at = "sub_new"; decode_check("sub_new", ev.sub_new(c1, c2, pool), [&](size_t i) { return (m1[i] + t - m2[i]) % t; }, [&](size_t i) { return z1[i] - z2[i]; });
                at = "multiply_new"; Ciphertext m3 = ev.multiply_new(c1, c2, pool);
                decode_check("multiply_new", m3, [&](size_t i) { return mulm(m1[i], m2[i]); }, [&](size_t i) { return z1[i] * z2[i]; });
                at = "relinearize_new"; Ciphertext r2 = ev.relinearize_new(m3, rk, pool);
                decode_check("relinearize_new", r2, [&](size_t i) { return mulm(m1[i], m2[i]); }, [&](size_t i) { return z1[i] * z2[i]; });
                at = "square_new"; decode_check("square_new", ev.relinearize_new(ev.square_new(c1, pool), rk, pool), [&](size_t i) { return mulm(m1[i], m1[i]); }, [&](size_t i) { return z1[i] * z1[i]; });
                if (ckks) { at = "rescale_to_next_new"; decode_check("rescale_to_next_new", ev.rescale_to_next_new(r2, pool), [&](size_t) { return 0ull; }, [&](size_t i) { return z1[i] * z2[i]; }); }
                at = "mod_switch_to_next_new"; decode_check("mod_switch_to_next_new", ev.mod_switch_to_next_new(c1, pool), [&](size_t i) { return m1[i]; }, [&](size_t i) { return z1[i]; });
                at = "add_plain_new"; decode_check("add_plain_new", ev.add_plain_new(c1, p2, pool), [&](size_t i) { return (m1[i] + m2[i]) % t; }, [&](size_t i) { return z1[i] + z2[i]; });
                at = "sub_plain_new"; decode_check("sub_plain_new", ev.sub_plain_new(c1, p2, pool), [&](size_t i) { return (m1[i] + t - m2[i]) % t; }, [&](size_t i) { return z1[i] - z2[i]; });
                at = "multiply_plain_new"; decode_check("multiply_plain_new", ev.multiply_plain_new(c1, p2, pool), [&](size_t i) { return mulm(m1[i], m2[i]); }, [&](size_t i) { return z1[i] * z2[i]; });
                if (!ckks) {
                    at = "transform_plain_to_ntt_new"; Plaintext pn = ev.transform_plain_to_ntt_new(p2, c1.parms_id(), pool);
                    good("transform_plain_to_ntt_new", pn.pool());
                    at = "transform_to_ntt_new"; Ciphertext cn = c1.is_ntt_form() ? c1.clone(pool) : ev.transform_to_ntt_new(c1, pool);
                    good("transform_to_ntt_new", cn.pool());
                    at = "multiply_plain_new(ntt)"; Ciphertext mp = ev.multiply_plain_new(cn, pn, pool);
                    if (scheme == SchemeType::BFV) { at = "transform_from_ntt_new"; mp = ev.transform_from_ntt_new(mp, pool); }
                    decode_check("multiply_plain_new(ntt)", mp, [&](size_t i) { return mulm(m1[i], m2[i]); }, [&](size_t i) { return z1[i]; });
                    const size_t half = m1.size() / 2;
                    at = "rotate_rows_new"; decode_check("rotate_rows_new", ev.rotate_rows_new(c1, 1, gk, pool), [&](size_t i) { return m1[(i / half) * half + (i % half + 1) % half]; }, [&](size_t i) { return z1[i]; });
                    at = "rotate_columns_new"; decode_check("rotate_columns_new", ev.rotate_columns_new(c1, gk, pool), [&](size_t i) { return m1[(i + half) % (2 * half)]; }, [&](size_t i) { return z1[i]; });
                } else {
                    const size_t slots = z1.size();
                    at = "rotate_vector_new"; decode_check("rotate_vector_new", ev.rotate_vector_new(c1, 1, gk, pool), [&](size_t) { return 0ull; }, [&](size_t i) { return z1[(i + 1) % slots]; });
                    at = "complex_conjugate_new"; decode_check("complex_conjugate_new", ev.complex_conjugate_new(c1, gk, pool), [&](size_t) { return 0ull; }, [&](size_t i) { return std::conj(z1[i]); });
                }
                // ---- LWE ------------------------------------------------------------------------------------------------------------------
                at = "extract_lwe_new"; LWECiphertext lwe = ev.extract_lwe_new(c1, 0, pool);
                at = "assemble_lwe_new"; good("assemble_lwe_new", ev.assemble_lwe_new(lwe, pool).pool());
                at = "pack_lwe_ciphertexts_new"; good("pack_lwe_ciphertexts_new", ev.pack_lwe_ciphertexts_new(std::vector<const LWECiphertext*>{&lwe, &lwe}, ak, pool).pool());
                // ---- key generation into the thread's pool --------------------------------------------------------------------------------
                at = "create_relin_keys"; RelinKeys rk2 = keygen.create_relin_keys(false, 2, pool);
                decode_check("relinearize_new(own keys)", ev.relinearize_new(m3, rk2, pool), [&](size_t i) { return mulm(m1[i], m2[i]); }, [&](size_t i) { return z1[i] * z2[i]; });
                at = "done";
            } catch (const std::exception& e) {
                report(id, at, std::string("EXCEPTION ") + e.what());
            }
            utils::stream_sync();
        };
        std::vector<std::thread> th;
        for (size_t i = 0; i < 4; i++) th.emplace_back(body, i);
        for (auto& x : th) x.join();
        context_pool->deny(false);
        MemoryPool::GlobalPool()->deny(false);
        std::printf("failures %zu\n", failures.load());
        std::printf(failures.load() ? "FAIL\n" : "OK\n");
        return failures.load() ? 1 : 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
