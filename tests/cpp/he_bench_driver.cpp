// troybench-shaped C++ driver (the reference's bench tool: test/bench/he_operations.cu:57-135, :695-735 "MultiplyRelinearize" and
// :811-840 "RescaleToNext"): BASELINE config 3 -- CKKS N = 16384, 6 x 50-bit -- timed THROUGH troy::Evaluator, the reference's seam:
//   * single objects: multiply_new + relinearize_new + rescale_to_next_new per op, stream synchronised after every call as the tool does
//     (he_operations.cu:711-720), and the fused Evaluator::multiply_relinearize_rescale_new;
//   * batches of 64 / 256 / 1024 ciphertext pairs: multiply_batched + relinearize_batched + rescale_to_next_batched, and
//     multiply_relinearize_rescale_batched;  1 and 4 host threads (the tool's -c option), every thread with its own operands.
// Every configuration first checks that the fused method is BIT-IDENTICAL to the three calls (payload, parms_id, scale) and that the result
// decrypts to the slot-wise product.  Output: one `key value` line per measurement (tests/test_gpu_cpp_api.py, bench.py other_configs.cpp_api).
//   he_bench_driver [check|bench|threads|single|stress] [repeat]   |   he_bench_driver devices [threads] [repeat]  (the tool's -c N -mp -md mode, below)
#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "../../troy-nova_amd/troy/troy.h"

using namespace troy;
using cd = std::complex<double>;
using clk = std::chrono::steady_clock;

static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct World {
    HeContextPointer context;
    std::unique_ptr<CKKSEncoder> encoder;
    std::unique_ptr<KeyGenerator> keygen;
    std::unique_ptr<Encryptor> encryptor;
    std::unique_ptr<Decryptor> decryptor;
    std::unique_ptr<Evaluator> evaluator;
    RelinKeys rk;
    double scale = std::pow(2.0, 40);
};

static bool same_ct(const Ciphertext& a, const Ciphertext& b) {
    if (a.parms_id() != b.parms_id() || a.scale() != b.scale() || a.is_ntt_form() != b.is_ntt_form() || a.polynomial_count() != b.polynomial_count() ||
        a.coeff_modulus_size() != b.coeff_modulus_size()) return false;
    Ciphertext ha = a.to_host(), hb = b.to_host();
    return ha.data().size() == hb.data().size() && std::memcmp(ha.data().raw_pointer(), hb.data().raw_pointer(), ha.data().size() * 8) == 0;
}

// ---- the reference tool's multi-device mode (`troybench -c N -mp -md`: test/bench/he_operations.cu:33-34, :87-88, :139-147; test/test_multithread.cu:18-37;
// readme.md:179-202; scripts/compare_bench.sh:12-13): thread i works in its OWN pool on device i % device_count, one context per device (moved there with
// to_device_inplace(pool)), every context's KeyGenerator built from the SAME secret key cloned to its device.  On a one-GPU box this degrades to N pools on
// device 0 (the tool's -mp mode).  Checks: all contexts share the secret key; on every pool the fused call equals the three calls and decrypts to the product
// (with a decryptor of ANOTHER device's context where there is one -- the keys are the same); then single-object and *_batched throughput per device and aggregate.
//   he_bench_driver devices [threads = 8] [repeat = 20]
static int run_devices(size_t threads, size_t repeat) {
    const size_t n = 16384, D = std::max<size_t>(1, utils::device_count());
    const size_t contexts = std::min(D, threads);
    EncryptionParameters params(SchemeType::CKKS);
    params.set_poly_modulus_degree(n);
    params.set_coeff_modulus(CoeffModulus::create(n, {50, 50, 50, 50, 50, 50}));
    const double scale = std::pow(2.0, 40);
    struct Dev {
        MemoryPoolHandle pool;
        HeContextPointer context;
        std::unique_ptr<CKKSEncoder> encoder;
        std::unique_ptr<KeyGenerator> keygen;
        std::unique_ptr<Encryptor> encryptor;
        std::unique_ptr<Decryptor> decryptor;
        std::unique_ptr<Evaluator> evaluator;
        RelinKeys rk;
    };
    std::vector<MemoryPoolHandle> pools(threads);
    for (size_t i = 0; i < threads; i++) pools[i] = MemoryPool::create(i % D);
    // the secret key comes from a key generator on the default device (readme.md:186-190)
    HeContextPointer he0 = HeContext::create(params, true, SecurityLevel::Classical128, 0x123);
    he0->to_device_inplace();
    KeyGenerator keygen0(he0);
    const SecretKey secret_key = keygen0.secret_key().clone();
    std::vector<Dev> dev(contexts);
    for (size_t d = 0; d < contexts; d++) {
        Dev& x = dev[d];
        x.pool = pools[d];                       // pools[d] lives on device d (d < D)
        x.context = HeContext::create(params, true, SecurityLevel::Classical128, 0x123 + d);
        x.context->to_device_inplace(x.pool);
        x.encoder.reset(new CKKSEncoder(x.context));
        x.keygen.reset(new KeyGenerator(x.context, secret_key.to_host().to_device(x.pool), x.pool));      // through the host: no peer access between devices is assumed
        x.encryptor.reset(new Encryptor(x.context));
        x.encryptor->set_public_key(x.keygen->create_public_key(false, x.pool));
        x.decryptor.reset(new Decryptor(x.context, x.keygen->secret_key()));
        x.evaluator.reset(new Evaluator(x.context));
        x.rk = x.keygen->create_relin_keys(false, 2, x.pool);
    }
    {   // all contexts share the secret key (test_multithread.cu:58-73)
        const std::vector<uint64_t> s0 = dev[0].keygen->secret_key().to_host().data().to_vector();
        int same = 1;
        for (size_t d = 1; d < contexts; d++) same &= dev[d].keygen->secret_key().to_host().data().to_vector() == s0 ? 1 : 0;
        std::printf("devices %zu\npools %zu\ncontexts %zu\ndevices_same_secret_key %d\n", D, threads, contexts, same);
    }
    std::atomic<int> ok{1};
    std::atomic<size_t> ready{0};
    std::atomic<bool> go{false};
    std::vector<double> single3(threads), single1(threads), batched3(threads), batched1(threads);
    const size_t B = 64, reps_single = std::max<size_t>(50, 800 / threads), reps_batched = std::max<size_t>(repeat, 8);
    for (int phase = 0; phase < 2; phase++) {     // 0: single objects, 1: batches of 64
        ready.store(0); go.store(false);
        auto body = [&](size_t i) {
            try {
                Dev& x = dev[i % contexts];
                MemoryPoolHandle pool = pools[i];
                const Evaluator& ev = *x.evaluator;
                std::mt19937_64 gen(11 + i);
                std::uniform_real_distribution<double> U(-1.0, 1.0);
                const size_t slots = x.encoder->slot_count();
                std::vector<cd> z1(slots), z2(slots);
                for (auto& v : z1) v = cd(U(gen), U(gen));
                for (auto& v : z2) v = cd(U(gen), U(gen));
                Ciphertext c1 = x.encryptor->encrypt_asymmetric_new(x.encoder->encode_complex64_simd_new(z1, std::nullopt, scale, pool), pool);
                Ciphertext c2 = x.encryptor->encrypt_asymmetric_new(x.encoder->encode_complex64_simd_new(z2, std::nullopt, scale, pool), pool);
                if (phase == 0) {
                    // parity on THIS pool: fused == three calls; the product decrypts (through the decryptor of the next device's context: same secret key)
                    Ciphertext m3 = ev.multiply_new(c1, c2, pool);
                    ev.relinearize_inplace(m3, x.rk, pool);
                    ev.rescale_to_next_inplace(m3, pool);
                    Ciphertext m1 = ev.multiply_relinearize_rescale_new(c1, c2, x.rk, pool);
                    if (!same_ct(m1, m3)) ok.store(0);
                    Dev& y = dev[(i + 1) % contexts];
                    Ciphertext moved = &y == &x ? m1.clone(pool) : m1.to_host().to_device(y.pool);
                    std::vector<cd> got = y.encoder->decode_complex64_simd_new(y.decryptor->decrypt_new(moved, y.pool), y.pool);
                    double e = 0;
                    for (size_t k = 0; k < slots; k++) e = std::max(e, std::abs(got[k] - z1[k] * z2[k]));
                    if (!(e < 1e-4)) ok.store(0);
                }
                auto once_single = [&](int fused) {
                    if (fused) { Ciphertext t = ev.multiply_relinearize_rescale_new(c1, c2, x.rk, pool); }
                    else { Ciphertext t = ev.multiply_new(c1, c2, pool); Ciphertext r = ev.relinearize_new(t, x.rk, pool); Ciphertext s = ev.rescale_to_next_new(r, pool); }
                    troyn_sync_current_stream();
                };
                std::vector<Ciphertext> A, Bv, d, t1, t2;
                std::vector<const Ciphertext*> pa, pb, pt1, pt2; std::vector<Ciphertext*> pd, q1, q2;
                if (phase == 1) {
                    A.assign(B, c1); Bv.assign(B, c2); d.resize(B); t1.resize(B); t2.resize(B);
                    for (size_t k = 0; k < B; k++) { pa.push_back(&A[k]); pb.push_back(&Bv[k]); pd.push_back(&d[k]); q1.push_back(&t1[k]); q2.push_back(&t2[k]); pt1.push_back(&t1[k]); pt2.push_back(&t2[k]); }
                }
                auto once_batched = [&](int fused) {
                    if (fused) ev.multiply_relinearize_rescale_batched(pa, pb, x.rk, pd, pool);
                    else { ev.multiply_batched(pa, pb, q1, pool); ev.relinearize_batched(pt1, x.rk, q2, pool); ev.rescale_to_next_batched(pt2, pd, pool); }
                    troyn_sync_current_stream();
                };
                for (int fused = 0; fused < 2; fused++) {
                    auto once = [&] { if (phase == 0) once_single(fused); else once_batched(fused); };
                    { size_t it = 0; for (auto w0 = clk::now(); secs(w0, clk::now()) < 0.05 || it < 4; it++) once(); }
                    ready.fetch_add(1);
                    while (ready.load() < (size_t)(fused + 1) * threads) std::this_thread::yield();      // all threads start each timed loop together
                    const size_t reps = phase == 0 ? reps_single : reps_batched;
                    auto t0 = clk::now();
                    for (size_t r = 0; r < reps; r++) once();
                    const double dt = secs(t0, clk::now());
                    const double rate = (double)(phase == 0 ? reps : reps * B) / dt;
                    (phase == 0 ? (fused ? single1 : single3) : (fused ? batched1 : batched3))[i] = rate;
                }
                if (phase == 1) {   // the batched results are the single-object results
                    Ciphertext s = ev.multiply_relinearize_rescale_new(c1, c2, x.rk, pool);
                    if (!same_ct(d[0], s) || !same_ct(d[B - 1], s)) ok.store(0);
                }
            } catch (const std::exception& e) { std::printf("devices thread %zu EXCEPTION %s\n", i, e.what()); ok.store(0); ready.fetch_add(4); }
        };
        std::vector<std::thread> th;
        for (size_t i = 0; i < threads; i++) th.emplace_back(body, i);
        for (auto& t : th) t.join();
    }
    // a thread's rate is measured over its own loop; the loops start together, so per-device and aggregate figures are sums
    auto report = [&](const char* name, const std::vector<double>& v) {
        double total = 0;
        std::vector<double> per(D, 0.0);
        for (size_t i = 0; i < threads; i++) { total += v[i]; per[i % D] += v[i]; }
        std::printf("devices_%s_ops_per_s %.1f\n", name, total);
        for (size_t d = 0; d < D; d++) std::printf("devices_%s_device%zu_ops_per_s %.1f\n", name, d, per[d]);
    };
    report("single_three_calls", single3); report("single_fused", single1);
    report("batch64_three_calls", batched3); report("batch64_fused", batched1);
    std::printf("devices_identical %d\n", ok.load());
    std::printf(ok.load() ? "OK\n" : "FAIL\n");
    dev.clear(); pools.clear();
    MemoryPool::Destroy();
    return ok.load() ? 0 : 1;
}

// ---- the pool's reuse rules (troy.h MemoryPool): a block another LIVE thread released is not handed out while fresh memory is to be had -- unless the pool is
// above its high-water mark, where it synchronises the device once and reuses; a block released while such a wait drains keeps its owner's tag (round-5 fix)
//   he_bench_driver pool
static int run_pool() {
    const size_t MB = size_t(1) << 20, block = 48 * MB;
    int ok = 1;
    for (int capped = 0; capped < 2; capped++) {
        MemoryPoolHandle pool = MemoryPool::create(0);
        if (capped) pool->set_high_water_bytes(64 * MB);
        const uint64_t mallocs0 = MemoryPool::device_allocations();
        std::atomic<int> stage{0};
        std::thread a([&] {      // thread A allocates and releases a block, then stays alive
            void* p = pool->allocate(block);
            pool->release(p);
            stage.store(1);
            while (stage.load() != 2) std::this_thread::yield();
        });
        while (stage.load() != 1) std::this_thread::yield();
        void* q = pool->allocate(block);     // thread B (this one): A's block is foreign and A is alive
        const size_t held = pool->held_bytes();
        const uint64_t mallocs = MemoryPool::device_allocations() - mallocs0;
        pool->release(q);
        void* r = pool->allocate(block);     // B's own block comes straight back
        const uint64_t mallocs2 = MemoryPool::device_allocations() - mallocs0;
        pool->release(r);
        stage.store(2);
        a.join();
        std::printf("pool_%s_held_MB %zu\npool_%s_device_allocations %llu\n", capped ? "capped" : "uncapped", held / MB, capped ? "capped" : "uncapped", (unsigned long long)mallocs);
        ok &= capped ? (held == block && mallocs == 1) : (held == 2 * block && mallocs == 2);      // capped: A's block reused after one device-wide wait
        ok &= mallocs2 == mallocs ? 1 : 0;
    }
    // test/multithread.cu:23-55 (MultithreadTest.DeviceAllocate): 64 threads allocate at the same time, 4 rounds; no two live arrays share an address
    {
        size_t clashes = 0;
        for (int round = 0; round < 4; round++) {
            std::vector<utils::DynamicArray> arrays(64);
            std::vector<std::thread> th;
            for (size_t i = 0; i < 64; i++) th.emplace_back([&arrays, i] { arrays[i] = utils::DynamicArray(64, true); });
            for (auto& x : th) x.join();
            for (size_t i = 0; i < 64; i++)
                for (size_t j = i + 1; j < 64; j++) clashes += arrays[i].raw_pointer() == arrays[j].raw_pointer() || !arrays[i].raw_pointer();
        }
        std::printf("pool_concurrent_allocate_clashes %zu\n", clashes);
        ok &= clashes == 0;
    }
    std::printf(ok ? "OK\n" : "FAIL\n");
    MemoryPool::Destroy();
    return ok ? 0 : 1;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "pool") == 0) {
        try { return run_pool(); } catch (const std::exception& e) { std::printf("EXCEPTION %s\n", e.what()); return 1; }
    }
    if (argc > 1 && std::strcmp(argv[1], "devices") == 0) {
        try {
            return run_devices(argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 8, argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 20);
        } catch (const std::exception& e) { std::printf("EXCEPTION %s\n", e.what()); return 1; }
    }
    const bool threads_only = argc > 1 && std::strcmp(argv[1], "threads") == 0;      // only the N-thread single-object sweep
    const bool single_only = argc > 1 && std::strcmp(argv[1], "single") == 0;        // only the one-thread single-object loops (profiling)
    const bool bench = threads_only || single_only || (argc > 1 && std::strcmp(argv[1], "bench") == 0);
    const size_t repeat = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 20;
    try {
        const size_t n = 16384;
        World w;
        EncryptionParameters params(SchemeType::CKKS);
        params.set_poly_modulus_degree(n);
        // `check60`: the parity part on the usual CKKS shape {60,50,50,50,50,60} -- wide first and special primes: integer kernels for those limbs, per-class fused chain
        const bool wide_chain = argc > 1 && std::strcmp(argv[1], "check60") == 0;
        params.set_coeff_modulus(wide_chain ? CoeffModulus::create(n, {60, 50, 50, 50, 50, 60}) : CoeffModulus::create(n, {50, 50, 50, 50, 50, 50}));
        w.context = HeContext::create(params, true, SecurityLevel::Classical128, 0x123);
        w.context->to_device_inplace();
        w.encoder.reset(new CKKSEncoder(w.context));
        w.keygen.reset(new KeyGenerator(w.context));
        w.encryptor.reset(new Encryptor(w.context));
        w.encryptor->set_public_key(w.keygen->create_public_key(false));
        w.decryptor.reset(new Decryptor(w.context, w.keygen->secret_key()));
        w.evaluator.reset(new Evaluator(w.context));
        w.rk = w.keygen->create_relin_keys(false);
        const Evaluator& ev = *w.evaluator;

        const size_t slots = w.encoder->slot_count();
        std::mt19937_64 gen(11);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        auto fresh = [&](std::vector<cd>& z) {
            z.resize(slots);
            for (auto& v : z) v = cd(U(gen), U(gen));
            return w.encryptor->encrypt_asymmetric_new(w.encoder->encode_complex64_simd_new(z, std::nullopt, w.scale));
        };
        std::vector<cd> z1, z2;
        Ciphertext c1 = fresh(z1), c2 = fresh(z2);

        // ---- parity of the fused method with the three calls + semantics --------------------------------------------------------------
        Ciphertext m3 = ev.multiply_new(c1, c2);
        ev.relinearize_inplace(m3, w.rk);
        ev.rescale_to_next_inplace(m3);
        Ciphertext m1 = ev.multiply_relinearize_rescale_new(c1, c2, w.rk);
        std::printf("fused_single_identical %d\n", same_ct(m1, m3) ? 1 : 0);
        {
            std::vector<cd> got = w.encoder->decode_complex64_simd_new(w.decryptor->decrypt_new(m1));
            double e = 0;
            for (size_t i = 0; i < slots; i++) e = std::max(e, std::abs(got[i] - z1[i] * z2[i]));
            std::printf("fused_single_error %.3e\n", e);
        }
        Ciphertext ip = c1;
        ev.multiply_relinearize_rescale_inplace(ip, c2, w.rk);
        std::printf("fused_inplace_identical %d\n", same_ct(ip, m3) ? 1 : 0);
        {
            // a batch of distinct pairs (16: above BATCH_OP_THRESHOLD, a multiple of 8 = the XCD-grouped order of the inner product)
            const size_t B = 16;
            std::vector<Ciphertext> a(B), b(B), d3(B), d1(B), t1(B), t2(B);
            std::vector<cd> z;
            for (size_t i = 0; i < B; i++) { a[i] = fresh(z); b[i] = fresh(z); }
            std::vector<const Ciphertext*> pa, pb, pt1, pt2; std::vector<Ciphertext*> pd3, pd1, q1, q2;
            for (size_t i = 0; i < B; i++) { pa.push_back(&a[i]); pb.push_back(&b[i]); pd3.push_back(&d3[i]); pd1.push_back(&d1[i]); q1.push_back(&t1[i]); q2.push_back(&t2[i]); pt1.push_back(&t1[i]); pt2.push_back(&t2[i]); }
            ev.multiply_batched(pa, pb, q1);
            ev.relinearize_batched(pt1, w.rk, q2);
            ev.rescale_to_next_batched(pt2, pd3);
            ev.multiply_relinearize_rescale_batched(pa, pb, w.rk, pd1);
            int ok = 1;
            for (size_t i = 0; i < B; i++) {
                ok &= same_ct(d1[i], d3[i]) ? 1 : 0;
                Ciphertext s = ev.multiply_relinearize_rescale_new(a[i], b[i], w.rk);
                ok &= same_ct(d1[i], s) ? 1 : 0;
            }
            std::printf("fused_batched_identical %d\n", ok);
            // a non-uniform batch (one operand at a lower level) takes the per-object path and still equals the three calls
            Ciphertext lo_a = ev.mod_switch_to_next_new(a[0]), lo_b = ev.mod_switch_to_next_new(b[0]);
            pa[3] = &lo_a; pb[3] = &lo_b;
            ev.multiply_relinearize_rescale_batched(pa, pb, w.rk, pd1);
            Ciphertext r = ev.multiply_new(lo_a, lo_b); ev.relinearize_inplace(r, w.rk); ev.rescale_to_next_inplace(r);
            std::printf("fused_mixed_levels_identical %d\n", (same_ct(d1[3], r) && same_ct(d1[2], d3[2])) ? 1 : 0);
        }
        // the reference's error behaviour survives the fusion
        {
            int caught = 0;
            try { Ciphertext c1c = ev.transform_from_ntt_new(c1); ev.multiply_relinearize_rescale_new(c1c, c2, w.rk); } catch (const std::invalid_argument&) { caught++; }
            Ciphertext bottom = c1;
            while (bottom.parms_id() != w.context->last_parms_id()) ev.mod_switch_to_next_inplace(bottom);
            try { ev.multiply_relinearize_rescale_new(bottom, bottom, w.rk); } catch (const std::invalid_argument&) { caught++; }
            std::printf("fused_errors %d\n", caught);
        }
        {
            // ---- call combining (troy.h): T threads of single-object calls on DISTINCT operands; every result equals the uncombined call's -------
            const size_t T = 8, rounds = 4;
            std::vector<Ciphertext> a(T), b(T), want3(T), want1(T);
            std::vector<cd> z;
            for (size_t t = 0; t < T; t++) {
                a[t] = fresh(z); b[t] = fresh(z);
                // two shapes in flight: three of the eight threads work one level down (their calls form batches of their own)
                if (t % 3 == 2) { ev.mod_switch_to_next_inplace(a[t]); ev.mod_switch_to_next_inplace(b[t]); }
                want3[t] = ev.multiply_new(a[t], b[t]); ev.relinearize_inplace(want3[t], w.rk); ev.rescale_to_next_inplace(want3[t]);
                want1[t] = ev.multiply_relinearize_rescale_new(a[t], b[t], w.rk);
            }
            troyn_sync_current_stream();
            combining::reset_stats();
            combining::set_enabled(true);
            const unsigned window_before = combining::window_us();
            combining::set_window_us(2000);   // generous: this is a correctness run (threads start raggedly)
            std::atomic<int> ok{1}, caught{0};
            auto body = [&](size_t t) {
                for (size_t r = 0; r < rounds; r++) {
                    Ciphertext m = ev.multiply_new(a[t], b[t]);
                    Ciphertext l = ev.relinearize_new(m, w.rk);
                    Ciphertext s = ev.rescale_to_next_new(l);
                    Ciphertext f = ev.multiply_relinearize_rescale_new(a[t], b[t], w.rk);
                    troyn_sync_current_stream();
                    if (!same_ct(s, want3[t]) || !same_ct(f, want1[t])) ok.store(0);
                    // an argument error stays with the thread that made it (checks run before the rendezvous)
                    if (t == 3 && r == 1) {
                        try { Ciphertext cf = ev.transform_from_ntt_new(a[t]); ev.multiply_new(cf, b[t]); } catch (const std::invalid_argument&) { caught.fetch_add(1); }
                    }
                }
            };
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++) th.emplace_back(body, t);
            for (auto& x : th) x.join();
            combining::set_enabled(false);
            combining::set_window_us(window_before);
            const combining::Stats st = combining::stats();
            std::printf("combined_identical %d\n", ok.load());
            std::printf("combined_errors %d\n", caught.load());
            std::printf("combined_calls %llu\n", (unsigned long long)st.calls);
            std::printf("combined_batches %llu\n", (unsigned long long)st.batches);
            std::printf("combined_largest_batch %llu\n", (unsigned long long)st.largest_batch);
            // one thread alone is never combined (and stays asynchronous)
            combining::reset_stats();
            combining::set_enabled(true);
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
            Ciphertext alone = ev.multiply_relinearize_rescale_new(a[0], b[0], w.rk);
            combining::set_enabled(false);
            std::printf("combined_alone_calls %llu\n", (unsigned long long)combining::stats().calls);
            std::printf("combined_alone_identical %d\n", same_ct(alone, want1[0]) ? 1 : 0);
        }
        if (wide_chain) { std::printf("OK\n"); MemoryPool::Destroy(); return 0; }
        if (argc > 1 && std::strcmp(argv[1], "stress") == 0) {
            // ---- call combining under ragged arrival: 24 threads, random op kinds (three calls / fused / only the multiply), two levels, random pauses,
            // occasional stream waits, threads that start late and leave early; every result is compared with the uncombined one -------------------------
            const size_t T = 24, ops = repeat > 20 ? repeat : 120;
            std::vector<Ciphertext> a(2 * T), b(2 * T), want3(2 * T), want1(2 * T), wantm(2 * T);
            std::vector<cd> z;
            for (size_t t = 0; t < 2 * T; t++) {
                a[t] = fresh(z); b[t] = fresh(z);
                if (t >= T) { ev.mod_switch_to_next_inplace(a[t]); ev.mod_switch_to_next_inplace(b[t]); }   // the second half one level down
                wantm[t] = ev.multiply_new(a[t], b[t]);
                want3[t] = ev.relinearize_new(wantm[t], w.rk); ev.rescale_to_next_inplace(want3[t]);
                want1[t] = ev.multiply_relinearize_rescale_new(a[t], b[t], w.rk);
            }
            troyn_sync_current_stream();
            combining::reset_stats();
            combining::set_enabled(true);
            std::atomic<size_t> bad{0}, done{0};
            auto body = [&](size_t t) {
                try {
                    std::mt19937_64 g(77 + t);
                    if (t % 5 == 4) std::this_thread::sleep_for(std::chrono::milliseconds(3));     // late starters
                    const size_t mine = t % 7 == 6 ? ops / 3 : ops;                               // early leavers
                    for (size_t i = 0; i < mine; i++) {
                        const size_t k = (g() & 1) ? t : t + T;                                    // level
                        const unsigned kind = (unsigned)(g() % 3);
                        if (kind == 0) {
                            Ciphertext s = ev.rescale_to_next_new(ev.relinearize_new(ev.multiply_new(a[k], b[k]), w.rk));
                            if (!same_ct(s, want3[k])) bad++;
                        } else if (kind == 1) {
                            Ciphertext f = ev.multiply_relinearize_rescale_new(a[k], b[k], w.rk);
                            if (!same_ct(f, want1[k])) bad++;
                        } else {
                            Ciphertext m = ev.multiply_new(a[k], b[k]);
                            if (g() % 4 == 0) troyn_sync_current_stream();
                            if (!same_ct(m, wantm[k])) bad++;
                        }
                        if (g() % 3 == 0) std::this_thread::sleep_for(std::chrono::microseconds(g() % 200));
                        done++;
                    }
                } catch (const std::exception& e) { std::printf("stress thread %zu EXCEPTION %s\n", t, e.what()); bad++; }
            };
            std::vector<std::thread> th;
            for (size_t t = 0; t < T; t++) th.emplace_back(body, t);
            for (auto& x : th) x.join();
            combining::set_enabled(false);
            const combining::Stats st = combining::stats();
            std::printf("stress_ops %zu\nstress_wrong %zu\nstress_combined_calls %llu\nstress_batches %llu\nstress_largest_batch %llu\n", done.load(), bad.load(),
                        (unsigned long long)st.calls, (unsigned long long)st.batches, (unsigned long long)st.largest_batch);
            std::printf(bad.load() == 0 ? "OK\n" : "FAIL\n");
            MemoryPool::Destroy();
            return bad.load() == 0 ? 0 : 1;
        }
        if (!bench) { std::printf("OK\n"); MemoryPool::Destroy(); return 0; }

        // ---- single objects (the tool's loop: every call followed by a stream synchronisation) -----------------------------------------
        for (int fused = 0; fused < 2 && !threads_only; fused++) {
            const size_t reps = 400;
            // the clocks need 20-25 ms of load to come up after an idle gap (tools/ramp_probe.py): 50 ms of the loop's own work first.  Rounds 3
            // and earlier warmed up with 10 calls (~1.5 ms) and timed 200 ops, i.e. mostly the ramp.
            for (auto w0 = clk::now(); secs(w0, clk::now()) < 0.05;) { Ciphertext t = ev.multiply_relinearize_rescale_new(c1, c2, w.rk); troyn_sync_current_stream(); }
            troyn_sync_current_stream();
            auto t0 = clk::now();
            for (size_t i = 0; i < reps; i++) {
                if (fused) { Ciphertext t = ev.multiply_relinearize_rescale_new(c1, c2, w.rk); troyn_sync_current_stream(); }
                else {
                    Ciphertext t = ev.multiply_new(c1, c2); troyn_sync_current_stream();
                    Ciphertext r = ev.relinearize_new(t, w.rk); troyn_sync_current_stream();
                    Ciphertext s = ev.rescale_to_next_new(r); troyn_sync_current_stream();
                }
            }
            const double dt = secs(t0, clk::now());
            std::printf("single_%s_us_per_op %.2f\n", fused ? "fused" : "three_calls", dt / reps * 1e6);
        }

        if (single_only) { std::printf("OK\n"); MemoryPool::Destroy(); return 0; }
        // ---- single objects from N host threads: the reference tool's -c N mode (he_operations.cu:85, :364-380: N threads, each on its own
        // stream, sharing context and keys; the reference has NO batched multiply / relinearize / rescale, :119-135, so this is what an
        // unmodified caller gets).  Every thread owns its operands; the three calls of an op are queued without a synchronisation in between
        // and the stream is synchronised once per op (the result is consumed by the caller).
        for (int combine = 0; combine < 2; combine++)
        for (size_t threads : {(size_t)1, (size_t)4, (size_t)16, (size_t)64}) {
            if (combine && threads == 1) continue;
            combining::set_enabled(combine != 0);
            combining::reset_stats();
            for (int fused = 0; fused < 2; fused++) {
                const size_t reps = std::max<size_t>(50, (combine ? 6400 : 1600) / threads);
                std::atomic<size_t> ready{0};
                std::atomic<bool> go{false};
                std::atomic<uint64_t> phase_ns[4] = {};   // where a caller's time goes: the three calls and the wait (three-call loop)
                auto body = [&](size_t) {
                    Ciphertext a = c1.clone(), b = c2.clone();
                    auto once = [&] {
                        if (fused) { Ciphertext t = ev.multiply_relinearize_rescale_new(a, b, w.rk); troyn_sync_current_stream(); }
                        else {
                            auto p0 = clk::now();
                            Ciphertext t = ev.multiply_new(a, b);
                            auto p1 = clk::now();
                            Ciphertext r = ev.relinearize_new(t, w.rk);
                            auto p2 = clk::now();
                            Ciphertext s = ev.rescale_to_next_new(r);
                            auto p3 = clk::now();
                            troyn_sync_current_stream();
                            auto p4 = clk::now();
                            phase_ns[0] += (uint64_t)(secs(p0, p1) * 1e9); phase_ns[1] += (uint64_t)(secs(p1, p2) * 1e9);
                            phase_ns[2] += (uint64_t)(secs(p2, p3) * 1e9); phase_ns[3] += (uint64_t)(secs(p3, p4) * 1e9);
                        }
                    };
                    for (auto w0 = clk::now(); secs(w0, clk::now()) < 0.05;) once();
                    ready.fetch_add(1);
                    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                    for (size_t r = 0; r < reps; r++) once();
                };
                std::vector<std::thread> th;
                for (size_t t = 0; t < threads; t++) th.emplace_back(body, t);
                while (ready.load() < threads) std::this_thread::yield();
                auto t0 = clk::now();
                go.store(true, std::memory_order_release);
                for (auto& x : th) x.join();
                const double mx = secs(t0, clk::now());
                std::printf("single_threads%zu_%s%s_ops_per_s %.1f\n", threads, fused ? "fused" : "three_calls", combine ? "_combined" : "", (double)(threads * reps) / mx);
                if (!fused) {
                    const double calls = (double)threads * (double)reps + 1e-9;   // the warm-up loop's calls are in the sums too: shares, not absolutes
                    const double tot = (double)(phase_ns[0] + phase_ns[1] + phase_ns[2] + phase_ns[3]) + 1e-9;
                    (void)calls;
                    std::printf("single_threads%zu_three_calls%s_time_shares multiply %.2f relinearize %.2f rescale %.2f wait %.2f\n", threads, combine ? "_combined" : "",
                                (double)phase_ns[0] / tot, (double)phase_ns[1] / tot, (double)phase_ns[2] / tot, (double)phase_ns[3] / tot);
                }
            }
            if (combine) {
                const combining::Stats st = combining::stats();
                std::printf("single_threads%zu_combined_mean_batch %.2f\n", threads, st.batches ? (double)st.calls / (double)st.batches : 0.0);
                std::printf("single_threads%zu_combined_gather_us_per_batch %.1f\n", threads, st.batches ? (double)st.gather_ns / 1e3 / (double)(st.batches + st.uncombined) : 0.0);
                std::printf("single_threads%zu_combined_execute_us_per_batch %.1f\n", threads, st.batches ? (double)st.execute_ns / 1e3 / (double)st.batches : 0.0);
                std::printf("single_threads%zu_combined_uncombined_calls %llu\n", threads, (unsigned long long)st.uncombined);
                std::printf("single_threads%zu_combined_caller_us_between_calls %.1f (no wait in between), %.1f (with a stream wait)\n", threads,
                            st.between_calls ? (double)st.between_ns / 1e3 / (double)st.between_calls : 0.0,
                            st.between_wait_calls ? (double)st.between_wait_ns / 1e3 / (double)st.between_wait_calls : 0.0);
                std::printf("single_threads%zu_combined_arrival_spread_us %.1f\n", threads, (double)st.spread_ns / 1e3 / (double)std::max<uint64_t>(1, st.batches + st.uncombined));
                std::printf("single_threads%zu_combined_window_expired %llu of %llu leaders, mean target %.1f\n", threads, (unsigned long long)st.window_expired,
                            (unsigned long long)(st.batches + st.uncombined), (double)st.target_sum / (double)std::max<uint64_t>(1, st.batches + st.uncombined));
            }
            combining::set_enabled(false);
            MemoryPool::GlobalPool()->release_unused();
        }

        if (threads_only) { std::printf("OK\n"); MemoryPool::Destroy(); return 0; }
        // ---- batched forms ----------------------------------------------------------------------------------------------------------------
        const size_t maxB = 1024, threads_max = 4;
        // operands: one contiguous block per thread (what *_batched returns, so `contiguous()` uses them in place), made by transform round trips
        std::vector<std::vector<Ciphertext>> A(threads_max), Bv(threads_max);
        for (size_t t = 0; t < threads_max; t++) {
            std::vector<Ciphertext> sa(maxB, c1), sb(maxB, c2);
            std::vector<const Ciphertext*> pa, pb; std::vector<Ciphertext*> qa, qb;
            A[t].resize(maxB); Bv[t].resize(maxB);
            for (size_t i = 0; i < maxB; i++) { pa.push_back(&sa[i]); pb.push_back(&sb[i]); qa.push_back(&A[t][i]); qb.push_back(&Bv[t][i]); }
            ev.negate_batched(pa, qa);       // adjacent windows of one buffer
            ev.negate_batched(pb, qb);
        }
        for (size_t threads : {(size_t)1, (size_t)4}) {
            for (size_t B : {(size_t)64, (size_t)256, (size_t)1024}) {
                for (int fused = 0; fused < 2; fused++) {
                    // every thread warms up, then all start their timed loops together; the wall clock runs from that common start to the last finish
                    const size_t reps = std::max<size_t>(repeat, 40 * 1024 / B / threads);
                    std::atomic<size_t> ready{0};
                    std::atomic<bool> go{false};
                    auto body = [&](size_t t) {
                        std::vector<const Ciphertext*> pa, pb; std::vector<Ciphertext> d(B), t1(B), t2(B); std::vector<Ciphertext*> pd, q1, q2; std::vector<const Ciphertext*> pt1, pt2;
                        for (size_t i = 0; i < B; i++) { pa.push_back(&A[t][i]); pb.push_back(&Bv[t][i]); pd.push_back(&d[i]); q1.push_back(&t1[i]); q2.push_back(&t2[i]); pt1.push_back(&t1[i]); pt2.push_back(&t2[i]); }
                        auto once = [&] {
                            if (fused) ev.multiply_relinearize_rescale_batched(pa, pb, w.rk, pd);
                            else { ev.multiply_batched(pa, pb, q1); ev.relinearize_batched(pt1, w.rk, q2); ev.rescale_to_next_batched(pt2, pd); }
                        };
                        // the clocks need 20-25 ms of load to come up after an idle gap: 50 ms of warm-up -- and at least 8 ops: the pool settles after a
                        // few (an op's result block is released while the next op already holds its own, and a request may take a block up to twice its
                        // size), and a hipMalloc of gigabytes inside the timed loop costs milliseconds (tens, when the memory was last used by another process)
                        { size_t it = 0; for (auto w0 = clk::now(); secs(w0, clk::now()) < 0.05 || it < 8; it++) { once(); troyn_sync_current_stream(); } }
                        ready.fetch_add(1);
                        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
                        // a stream wait behind every op, as the reference's tool does (he_operations.cu:711-720).  The methods themselves no longer
                        // wait (round 4); without this wait 4 threads x batch 1024 queue ten ops deep each and the three-call chain drops from 198 k
                        // to 134 k ops/s (kernels of four saturating launch sequences interleave on the GPU), every other shape gains 0-4 %
                        for (size_t r = 0; r < reps; r++) { once(); troyn_sync_current_stream(); }
                    };
                    std::vector<std::thread> th;
                    for (size_t t = 0; t < threads; t++) th.emplace_back(body, t);
                    while (ready.load() < threads) std::this_thread::yield();
                    const uint64_t mallocs0 = MemoryPool::device_allocations();
                    auto t0 = clk::now();
                    go.store(true, std::memory_order_release);
                    for (auto& x : th) x.join();
                    const double mx = secs(t0, clk::now());
                    if (MemoryPool::device_allocations() != mallocs0)
                        std::printf("note batched_%s_threads%zu_batch%zu: %llu device allocations inside the timed loop\n", fused ? "fused" : "three_calls", threads, B,
                                    (unsigned long long)(MemoryPool::device_allocations() - mallocs0));
                    const size_t repeat_used = reps;
                    std::printf("batched_%s_threads%zu_batch%zu_ops_per_s %.1f\n", fused ? "fused" : "three_calls", threads, B, (double)(threads * B * repeat_used) / mx);
                    // the blocks this configuration's threads cached (tens of GB at batch 1024) are of no use to the next one: give them back, or the
                    // device fills up over the sweep and a later timed loop pays for the pool's out-of-memory path (seen as 134 k instead of 200 k)
                    MemoryPool::GlobalPool()->release_unused();
                }
            }
        }
        std::printf("OK\n");
        A.clear(); Bv.clear();
        MemoryPool::Destroy();
        return 0;
    } catch (const std::exception& e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 1;
    }
}
