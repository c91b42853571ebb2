// behzbench -- development harness of the second-generation BEHZ kernels (csrc/behz2_kernels.hpp): builds the tables for a chain
// of L primes, runs behz2_lift_kernel / behz2_floor_kernel on random residues, checks sampled coefficients against a step-by-step
// host evaluation of the same conversions (128-bit %), and times the kernels with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I troy-nova_amd/csrc tools/behzbench/behzbench.hip -o tools/behzbench/behzbench
//   ./behzbench [log_n=15] [L=10] [qbits=50] [batch=64] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "behz2_kernels.hpp"

using namespace troyn;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <typename T> static T* upload(const std::vector<T>& v) {
    T* d; CK(hipMalloc(&d, v.size() * sizeof(T))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}

template <int L, bool S> static void launch_lift(unsigned grid, unsigned ch, const Behz2Dev& c, const u64* in, u64* out) {
    hipLaunchKernelGGL((behz2_lift_kernel<L, S>), dim3(grid), dim3(256), 0, 0, ch, c, in, out);
}
template <int L, bool S> static void launch_floor(unsigned grid, unsigned ch, const Behz2Dev& c, const u64* a, const u64* b, u64* out) {
    hipLaunchKernelGGL((behz2_floor_kernel<L, S>), dim3(grid), dim3(256), 0, 0, ch, c, a, b, out);
}
#ifndef BENCH_L
#define BENCH_L 10
#endif

int main(int argc, char** argv) {
    const unsigned log_n = argc > 1 ? atoi(argv[1]) : 15;
    const size_t L = BENCH_L;
    const unsigned qbits = argc > 2 ? atoi(argv[2]) : 50;
    const size_t batch = argc > 3 ? atoi(argv[3]) : 64;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const unsigned n = 1u << log_n;
    const u64 t = 65537;   // any plain modulus
    const u64 mt = (u64)1 << 32;
    std::vector<u64> q = host::get_primes(2 * (u64)n, qbits, L);
    std::vector<u64> primes = host::get_primes(2 * (u64)n, 61, L + 3);
    const u64 m_sk = primes[0];
    std::vector<u64> B(primes.begin() + 2, primes.begin() + 2 + L);
    std::vector<u64> bsk = B; bsk.push_back(m_sk);
    bool smallq = true;
    for (u64 v : q) if (v >= ((u64)1 << 50)) smallq = false;
    printf("N=%u L=%zu qbits=%u batch=%zu smallq=%d\n", n, L, qbits, batch, (int)smallq);

    std::vector<u64> blob; Behz2Offsets o;
    if (!behz2_build_tables(q, B, m_sk, t, smallq, blob, o)) { fprintf(stderr, "tables failed\n"); return 2; }
    // input scalings (the same constants troyn_behz_create keeps)
    std::vector<u64> inv_punc(L, 1);
    for (size_t i = 0; i < L; i++) if (L > 1) host::invmod(host::product_mod(q, i, q[i]), q[i], inv_punc[i]);
    std::vector<ulonglong2> s_mt(L), s_t(L);
    std::vector<DevModulus> qm(L);
    for (size_t i = 0; i < L; i++) {
        host::Shoup a = host::shoup(host::mulmod(mt % q[i], inv_punc[i], q[i]), q[i]); s_mt[i] = make_ulonglong2(a.operand, a.quotient);
        host::Shoup b = host::shoup(host::mulmod(t % q[i], inv_punc[i], q[i]), q[i]); s_t[i] = make_ulonglong2(b.operand, b.quotient);
        std::memset(&qm[i], 0, sizeof(DevModulus));
        qm[i].q = q[i]; host::BarrettRatio r = host::barrett_ratio(q[i]); qm[i].ratio_lo = r.lo; qm[i].ratio_hi = r.hi;
    }
    u64* d_blob = upload(blob);
    Behz2Dev c; std::memset(&c, 0, sizeof(c));
    c.L = (unsigned)L; c.n = n; c.rs = o.rs;
    c.q_mods = upload(qm); c.q_mt_inv_punc = upload(s_mt); c.q_t_inv_punc = upload(s_t);
    c.lift_mt = (const u32*)(d_blob + o.lift_mt); c.lift_rows = (const u32*)(d_blob + o.lift_rows); c.lift_rc = d_blob + o.lift_rc;
    c.fa_rows = (const u32*)(d_blob + o.fa_rows); c.fa_rc = d_blob + o.fa_rc;
    c.fb_cols = (const u32*)(d_blob + o.fb_cols); c.fb_rc = d_blob + o.fb_rc;

    std::mt19937_64 rng(7);
    const size_t lift_items = batch * 2, floor_items = batch * 3;
    std::vector<u64> hq(floor_items * L * n), hb(floor_items * (L + 1) * n);
    for (size_t it = 0; it < floor_items; it++) {
        for (size_t i = 0; i < L; i++) for (unsigned x = 0; x < n; x++) hq[(it * L + i) * n + x] = rng() % q[i];
        for (size_t b = 0; b <= L; b++) for (unsigned x = 0; x < n; x++) hb[(it * (L + 1) + b) * n + x] = rng() % bsk[b];
    }
    // corner values in the first coefficients of item 0
    for (size_t i = 0; i < L; i++) { hq[i * n + 0] = q[i] - 1; hq[i * n + 1] = 0; hq[i * n + 2] = 1; }
    for (size_t b = 0; b <= L; b++) { hb[b * n + 0] = bsk[b] - 1; hb[b * n + 1] = 0; hb[b * n + 2] = 1; }
    u64 *d_q = upload(hq), *d_b = upload(hb), *d_lift, *d_floor;
    CK(hipMalloc(&d_lift, lift_items * (L + 1) * n * sizeof(u64)));
    CK(hipMalloc(&d_floor, floor_items * L * n * sizeof(u64)));
    const unsigned ch = (n + 255) / 256;
    auto run_lift = [&] { if (smallq) launch_lift<BENCH_L, true>((unsigned)(lift_items * ch), ch, c, d_q, d_lift); else launch_lift<BENCH_L, false>((unsigned)(lift_items * ch), ch, c, d_q, d_lift); };
    auto run_floor = [&] { if (smallq) launch_floor<BENCH_L, true>((unsigned)(floor_items * ch), ch, c, d_q, d_b, d_floor); else launch_floor<BENCH_L, false>((unsigned)(floor_items * ch), ch, c, d_q, d_b, d_floor); };
    run_lift(); run_floor();
    CK(hipDeviceSynchronize());
    std::vector<u64> ol(lift_items * (L + 1) * n), of(floor_items * L * n);
    CK(hipMemcpy(ol.data(), d_lift, ol.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(of.data(), d_floor, of.size() * 8, hipMemcpyDeviceToHost));

    // ---- host evaluation, step by step (evaluator.cu:50-116 order) ----
    using host::mulmod;
    auto M = [&](u128 v, u64 m) { return (u64)(v % m); };
    std::vector<u64> inv_mt(L + 1), inv_q(L + 1), Binv(L, 1);
    for (size_t b = 0; b <= L; b++) { host::invmod(mt % bsk[b], bsk[b], inv_mt[b]); host::invmod(host::product_mod(q, SIZE_MAX, bsk[b]), bsk[b], inv_q[b]); }
    for (size_t b = 0; b < L; b++) if (L > 1) host::invmod(host::product_mod(B, b, B[b]), B[b], Binv[b]);
    u64 inv_q_mt, invB_msk;
    host::invmod(host::product_mod(q, SIZE_MAX, mt), mt, inv_q_mt);
    host::invmod(host::product_mod(B, SIZE_MAX, m_sk), m_sk, invB_msk);
    size_t bad = 0, checked = 0;
    auto sample = [&](size_t items, auto fn) {
        for (size_t it : {(size_t)0, items / 2, items - 1}) for (unsigned x = 0; x < n; x += (x < 8 ? 1 : 997)) fn(it, x);
    };
    sample(lift_items, [&](size_t it, unsigned x) {
        std::vector<u64> y(L);
        for (size_t i = 0; i < L; i++) y[i] = mulmod(mulmod(hq[(it * L + i) * n + x], mt % q[i], q[i]), inv_punc[i], q[i]);
        u128 s = 0;
        for (size_t i = 0; i < L; i++) s += (u128)y[i] * host::product_mod(q, i, mt);
        const u64 r_mt = M((u128)M(s, mt) * ((mt - inv_q_mt) % mt), mt);
        for (size_t b = 0; b <= L; b++) {
            const u64 p = bsk[b];
            u128 a = 0;
            for (size_t i = 0; i < L; i++) a += (u128)y[i] * host::product_mod(q, i, p);
            u64 temp = r_mt; if (temp >= mt / 2) temp += p - mt;
            const u64 mad = M((u128)temp * host::product_mod(q, SIZE_MAX, p) + M(a, p), p);
            const u64 e = mulmod(mad, inv_mt[b], p);
            checked++;
            if (e != ol[(it * (L + 1) + b) * n + x]) { if (bad++ < 5) printf("lift mismatch item %zu x %u b %zu: %llu vs %llu\n", it, x, b, ol[(it * (L + 1) + b) * n + x], e); }
        }
    });
    sample(floor_items, [&](size_t it, unsigned x) {
        std::vector<u64> y(L), r(L + 1), z(L);
        for (size_t i = 0; i < L; i++) y[i] = mulmod(mulmod(hq[(it * L + i) * n + x], t % q[i], q[i]), inv_punc[i], q[i]);
        for (size_t b = 0; b <= L; b++) {
            const u64 p = bsk[b];
            u128 a = 0;
            for (size_t i = 0; i < L; i++) a += (u128)y[i] * host::product_mod(q, i, p);
            const u64 xb = mulmod(hb[(it * (L + 1) + b) * n + x], t % p, p);
            r[b] = mulmod((xb + p - M(a, p)) % p, inv_q[b], p);
        }
        u128 h = 0;
        for (size_t b = 0; b < L; b++) { z[b] = mulmod(r[b], Binv[b], B[b]); h += (u128)z[b] * host::product_mod(B, b, m_sk); }
        const u64 alpha = mulmod((M(h, m_sk) + m_sk - r[L]) % m_sk, invB_msk, m_sk);
        for (size_t j = 0; j < L; j++) {
            u128 g = 0;
            for (size_t b = 0; b < L; b++) g += (u128)z[b] * host::product_mod(B, b, q[j]);
            const u64 pb = host::product_mod(B, SIZE_MAX, q[j]);
            u64 e;
            if (alpha > m_sk / 2) e = M((u128)M(g, q[j]) + (u128)mulmod((m_sk - alpha) % q[j], pb, q[j]), q[j]);
            else e = M((u128)M(g, q[j]) + (u128)mulmod(alpha % q[j], (q[j] - pb) % q[j], q[j]), q[j]);
            checked++;
            if (e != of[(it * L + j) * n + x]) { if (bad++ < 10) printf("floor mismatch item %zu x %u j %zu: %llu vs %llu\n", it, x, j, of[(it * L + j) * n + x], e); }
        }
    });
    printf("checked %zu values, %zu mismatches\n", checked, bad);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto fn, double bytes) {
        for (int i = 0; i < 3; i++) fn();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; i++) fn();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-6s %9.1f us/launch  %7.2f TB/s algorithmic\n", name, 1e3 * ms / reps, bytes / (1e-3 * ms / reps) / 1e12);
    };
    time_it("lift", run_lift, (double)lift_items * (2 * L + 1) * n * 8);
    time_it("floor", run_floor, (double)floor_items * (3 * L + 1) * n * 8);
    return bad ? 1 : 0;
}
