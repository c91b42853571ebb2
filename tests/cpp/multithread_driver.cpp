// C++ driver for tests/test_gpu_cpp_api.py::test_multithread_cpp_api (the reference's test/test_multithread.cu scenario): several
// host threads share one HeContext, one set of keys and the global memory pool, each on its own per-thread stream, and run
// encrypt -> multiply -> relinearize -> mod-switch -> decrypt loops concurrently; every result must decrypt correctly.
// With TROY_COMBINE=1 in the environment the same program runs with call combining (troy.h): one shared stream, the multiply /
// relinearize calls of concurrent threads batched, everything else (encrypt, add, mod-switch, decrypt) queued between the batches.
// usage: multithread_driver <threads> <iterations>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;

int main(int argc, char** argv) {
    try {
        const size_t threads = argc > 1 ? std::strtoull(argv[1], nullptr, 0) : 4, iterations = argc > 2 ? std::strtoull(argv[2], nullptr, 0) : 10;
        const size_t n = 8192;
        EncryptionParameters params(SchemeType::BFV);
        params.set_poly_modulus_degree(n);
        params.set_coeff_modulus(CoeffModulus::create(n, {50, 50, 50}));
        params.set_plain_modulus(PlainModulus::batching(n, 20));
        const uint64_t t = params.plain_modulus().value();
        HeContextPointer context = HeContext::create(params, true, SecurityLevel::Nil, 0x7ead);
        context->to_device_inplace();
        BatchEncoder encoder(context);
        KeyGenerator keygen(context);
        const PublicKey pk = keygen.create_public_key(false);
        const RelinKeys rk = keygen.create_relin_keys(false);
        const GaloisKeys gk = keygen.create_galois_keys(false);
        Decryptor decryptor(context, keygen.secret_key());
        std::atomic<size_t> bad{0}, done{0};
        auto work = [&](size_t id) {
            try {
                Encryptor encryptor(context);
                encryptor.set_public_key(pk);
                Evaluator ev(context);
                std::mt19937_64 gen(1000 + id);
                for (size_t it = 0; it < iterations; it++) {
                    std::vector<uint64_t> a(n), b(n);
                    for (size_t i = 0; i < n; i++) { a[i] = gen() % t; b[i] = gen() % t; }
                    Ciphertext ca = encryptor.encrypt_asymmetric_new(encoder.encode_new(a)), cb = encryptor.encrypt_asymmetric_new(encoder.encode_new(b));
                    Ciphertext m = ev.multiply_new(ca, cb);
                    ev.relinearize_inplace(m, rk);
                    ev.add_inplace(m, ca);
                    ev.mod_switch_to_next_inplace(m);
                    const std::vector<uint64_t> got = encoder.decode_new(decryptor.decrypt_new(m));
                    for (size_t i = 0; i < n; i++) {
                        const uint64_t want = (uint64_t)((((unsigned __int128)a[i] * b[i]) + a[i]) % t);
                        if (got[i] != want) { bad++; break; }
                    }
                    // a rotation (key switch by a Galois key): the rows of the 2 x n/2 slot matrix move left by 3
                    const std::vector<uint64_t> rot = encoder.decode_new(decryptor.decrypt_new(ev.rotate_rows_new(ca, 3, gk)));
                    const size_t half = n / 2;
                    for (size_t i = 0; i < n; i++) {
                        const size_t row = i / half, col = i % half;
                        if (rot[i] != a[row * half + (col + 3) % half]) { bad++; break; }
                    }
                    done++;
                }
            } catch (const std::exception& e) {
                std::printf("thread %zu EXCEPTION %s\n", id, e.what());
                bad++;
            }
        };
        std::vector<std::thread> pool;
        for (size_t i = 0; i < threads; i++) pool.emplace_back(work, i);
        for (auto& th : pool) th.join();
        std::printf("threads %zu iterations %zu completed %zu wrong %zu\n", threads, iterations, done.load(), bad.load());
        std::printf("combining %d combined_calls %llu\n", combining::enabled() ? 1 : 0, (unsigned long long)combining::stats().calls);
        const bool ok = bad.load() == 0 && done.load() == threads * iterations;
        std::printf(ok ? "OK\n" : "FAIL\n");
        MemoryPool::Destroy();
        return ok ? 0 : 1;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
