// ksbench.hip -- standalone A/B harness for the key-switch / NTT kernels at the headline shape
// (CKKS N = 16384, K = 6 limbs of 50 bits, L = 5).  Development tool: it instantiates only the kernels
// under study (seconds to build instead of the minutes libtroyn.so takes) and checks every variant
// bit-for-bit against the shipped kernel of the same name before timing it.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DKSBENCH_LEAN] -o ksbench ksbench.hip     (LEAN: shipped kernels only, 15 s)
//   ./ksbench [batch] [reps] [variant ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../troy-nova_amd/csrc/host_math.hpp"
#include "../../troy-nova_amd/csrc/ntt_kernels.hpp"
#include "../../troy-nova_amd/csrc/ksmac_kernels.hpp"
#include "ksmac3_experiment.hpp"
#include "ksmac4_experiment.hpp"
#include "ksmac5_experiment.hpp"
#if __has_include("ksmac_base.hpp")
#include "ksmac_base.hpp"      // frozen copy of the committed kernel (tools/ksbench/freeze_base.sh): same-run A/B
#define KSBENCH_HAVE_BASE 1
#endif

using namespace troyn;

// plain forward whole-limb transform with the last wave-private exchange as a DPP register <-> lane transpose (A/B only)
__global__ __launch_bounds__(1024, 1) void ntt_fwd14_shfl_kernel(NttArgs a) {
    __shared__ u64 lds[ntt_lds_words(14)];
    ntt_pass_body<ArithF64, 14, 0, 14, 14, 4, false, true, true, false, 0, 0, false, true>(a, nullptr, lds, blockIdx.x, threadIdx.x);
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned LOGN = 14, N = 1u << LOGN, K = 6, L = 5;

__global__ void center_kernel(const u64* in, double* out, size_t count, unsigned n, const u64* moduli, unsigned nmod) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        const u64 q = moduli[(i / n) % nmod], d = in[i];
        out[i] = (double)d; (void)q;
    }
}

__global__ void fill_kernel(u64* out, size_t count, unsigned n, const u64* moduli, unsigned nmod, unsigned mod_of_row_div, u64 seed) {
    // row r of n words gets modulus moduli[(r / mod_of_row_div) % nmod]  (mod_of_row_div = 1: [..][nmod][n] layouts)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        u64 z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const size_t row = i / n;
        out[i] = z % moduli[(row / mod_of_row_div) % nmod];
    }
}

static DevModulus plan_modulus(const host::NttTable& t, unsigned log_n) {
    DevModulus m;
    std::memset(&m, 0, sizeof(m));
    const u64 q = t.q;
    m.q = q;
    host::BarrettRatio r = host::barrett_ratio(q);
    m.ratio_lo = r.lo; m.ratio_hi = r.hi;
    m.inv_n_op = t.inv_degree.operand; m.inv_n_quo = t.inv_degree.quotient;
    m.pd = (double)q; m.inv_pd = 1.0 / (double)q;
    m.inv_n_d = (double)t.inv_degree.operand; m.inv_n_pd = (double)t.inv_degree.operand * (1.0 / (double)q);
    const size_t n = (size_t)1 << log_n;
    const u64 nw = host::mulmod(t.inv[n - 1].operand, t.inv_degree.operand, q);
    m.inv_n_w_d = (double)nw; m.inv_n_w_pd = (double)nw * (1.0 / (double)q);
    return m;
}

struct Bench {
    size_t B;
    DevModulus* d_mods;
    double* d_fwd;
    u64* d_moduli;
    u64 *digits, *target, *out_ref, *out;
    u64* keys[L];
    KeyPtrs kp;
    NttArgs args;
};

static float time_launch(const std::function<void()>& f, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // the clocks ramp for tens of milliseconds after an idle gap (a host copy is enough): one warm-up launch made whatever was timed
    // first after a gap read 13 % slow
    for (int i = 0; i < reps; i++) f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const size_t B = argc > 1 ? (size_t)atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    std::vector<std::string> want;
    for (int i = 3; i < argc; i++) want.push_back(argv[i]);
    auto wanted = [&](const char* name) {
        if (want.empty()) return true;
        for (auto& w : want) if (w == name) return true;
        return false;
    };

    std::vector<u64> q = host::get_primes(2ull * N, 50, K);
    std::vector<DevModulus> mods(K);
    std::vector<double> fwd((size_t)K * N), fwdp((size_t)K * N), r1((size_t)K * (N >> 10) * 32), r2((size_t)K * N);
    for (unsigned i = 0; i < K; i++) {
        host::NttTable t = host::make_ntt_table(LOGN, q[i], 0);
        mods[i] = plan_modulus(t, LOGN);
        const double inv_p = 1.0 / (double)q[i];
        for (size_t x = 0; x < N; x++) { fwd[(size_t)i * N + x] = (double)t.fwd[x].operand; fwdp[(size_t)i * N + x] = (double)t.fwd[x].operand * inv_p; }
        for (unsigned th = 0; th < (N >> 10); th++)
            for (unsigned s = 1; s < 32; s++) {
                unsigned lvl = 31 - __builtin_clz(s), g = s - (1u << lvl);
                r1[((size_t)i * (N >> 10) + th) * 32 + s] = (double)t.fwd[(((N >> 10) + th) << lvl) + g].operand;
            }
        for (unsigned T = 0; T < (N >> 5); T++)
            for (unsigned s = 1; s < 32; s++) {
                unsigned lvl = 31 - __builtin_clz(s), g = s - (1u << lvl);
                r2[(size_t)i * N + ksm_perm(T * 32 + s)] = (double)t.fwd[(((N >> 5) + T) << lvl) + g].operand;
            }
    }
    Bench b;
    b.B = B;
    CHECK(hipMalloc(&b.d_mods, K * sizeof(DevModulus)));
    CHECK(hipMalloc(&b.d_fwd, fwd.size() * sizeof(double)));
    CHECK(hipMalloc(&b.d_moduli, K * sizeof(u64)));
    CHECK(hipMemcpy(b.d_mods, mods.data(), K * sizeof(DevModulus), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b.d_fwd, fwd.data(), fwd.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(b.d_moduli, q.data(), K * sizeof(u64), hipMemcpyHostToDevice));
    const size_t dig_words = B * L * N, out_words = B * 2 * (L + 1) * N, key_words = 2ull * K * N;
    CHECK(hipMalloc(&b.digits, dig_words * 8));
    CHECK(hipMalloc(&b.target, dig_words * 8));
    CHECK(hipMalloc(&b.out_ref, out_words * 8));
    CHECK(hipMalloc(&b.out, out_words * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, b.digits, dig_words, N, b.d_moduli, L, 1u, 0x1111ull);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, b.target, dig_words, N, b.d_moduli, L, 1u, 0x2222ull);
    std::memset(&b.kp, 0, sizeof(b.kp));
    for (unsigned j = 0; j < L; j++) {
        CHECK(hipMalloc(&b.keys[j], key_words * 8));
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, b.keys[j], key_words, N, b.d_moduli, K, 1u, 0x3333ull + j);
        b.kp.p[j] = b.keys[j];
    }
    double *d_fwdp, *d_r1, *d_r2, *d_keys;
    CHECK(hipMalloc(&d_fwdp, fwdp.size() * 8)); CHECK(hipMemcpy(d_fwdp, fwdp.data(), fwdp.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_r1, r1.size() * 8)); CHECK(hipMemcpy(d_r1, r1.data(), r1.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_r2, r2.size() * 8)); CHECK(hipMemcpy(d_r2, r2.data(), r2.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_keys, (size_t)L * key_words * 8));
    CHECK(hipDeviceSynchronize());

    NttArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = b.digits; a.out = b.out_ref;
    a.mods = b.d_mods; a.tw = b.d_fwd;
    a.in_bstride = (long long)L * N; a.in_pstride = 0; a.in_cstride = N;
    a.out_bstride = 2ll * (L + 1) * N; a.out_pstride = (long long)(L + 1) * N; a.out_cstride = N;
    a.pcount = 1; a.ncomp = L + 1;
    a.table_start = 0; a.table_count = K; a.mode = 1; a.decomp = L;
    a.reduce_input = 1; a.stream_loads = 0; a.skip_diag = 1;
    a.ext0 = b.target; a.ext0_bstride = (long long)L * N; a.ext0_cstride = N;
    a.batch = (unsigned)B; a.key_pstride = (long long)K * N;
    a.xcd_groups = (B % 8 == 0) ? L + 1 : 0;
    const unsigned blocks = (unsigned)(B * (L + 1));

    // reference: the shipped kernel shape of round 1 (1024 threads x 16 coefficients)
    float t_ref = time_launch([&] { hipLaunchKernelGGL((ks_mac_kernel<ArithF64, 14, 4>), dim3(blocks), dim3(1024), 0, 0, a, b.kp); }, reps);
    printf("%-28s %9.1f us  (reference)\n", "ks_mac<14,4> 1024thr", t_ref);
    std::vector<u64> ref(out_words), got(out_words);
    CHECK(hipMemcpy(ref.data(), b.out_ref, out_words * 8, hipMemcpyDeviceToHost));

    auto run_variant = [&](const char* name, const std::function<void(const NttArgs&)>& launch) {
        if (!wanted(name)) return;
        NttArgs v = a;
        v.out = b.out;
        CHECK(hipMemset(b.out, 0xff, out_words * 8));
        float t = time_launch([&] { launch(v); }, reps);
        CHECK(hipMemcpy(got.data(), b.out, out_words * 8, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < out_words; i++) if (got[i] != ref[i]) { if (!bad) first = i; bad++; }
        printf("%-28s %9.1f us  %s", name, t, bad ? "MISMATCH" : "bit-exact");
        if (bad) printf(" (%zu words, first at %zu: got %llu want %llu)", bad, first, got[first], ref[first]);
        printf("\n");
    };

#ifndef KSBENCH_LEAN
    run_variant("ks_mac<14,5> 512thr", [&](const NttArgs& v) {
        hipLaunchKernelGGL((ks_mac_kernel<ArithF64, 14, 5>), dim3(blocks), dim3(512), 0, 0, v, b.kp); });
#endif
    KsMacArgs ka;
    std::memset(&ka, 0, sizeof(ka));
    ka.digits = b.digits; ka.dig_bstride = (long long)L * N; ka.dig_cstride = N;
    ka.diag = b.target; ka.diag_bstride = (long long)L * N; ka.diag_cstride = N;
    ka.out_bstride = 2ll * (L + 1) * N; ka.out_pstride = (long long)(L + 1) * N; ka.out_cstride = N;
    ka.mods = b.d_mods; ka.tw = b.d_fwd; ka.tw_r1 = d_r1; ka.tw_r2 = d_r2;
    ka.keys = d_keys; ka.key_jstride = 2ll * K * N; ka.key_pstride = (long long)K * N;
    ka.L = L; ka.table_start = 0; ka.table_count = K; ka.batch = (unsigned)B; ka.grouped = (B % 8 == 0) ? 1 : 0;
    if (getenv("KSB_ORDER")) ka.grouped = (unsigned)atoi(getenv("KSB_ORDER"));      // 3: bands of two rows (B must be a multiple of 128)
    float t_prep = time_launch([&] { hipLaunchKernelGGL(ksmac_prepare_keys_kernel, dim3(1024), dim3(256), 0, 0, b.kp, L, 2 * K, N, d_keys, (const ulonglong2*)nullptr, (const DevModulus*)nullptr, 0u, (double*)nullptr); }, reps);
    printf("%-28s %9.1f us\n", "prepare_keys", t_prep);
    run_variant("ksmac2", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out;
        hipLaunchKernelGGL((ksmac2_kernel<14, false>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    double* d_digf;
    CHECK(hipMalloc(&d_digf, dig_words * 8));
    hipLaunchKernelGGL(center_kernel, dim3(4096), dim3(256), 0, 0, b.digits, d_digf, dig_words, N, b.d_moduli, L);
#ifdef KSBENCH_HAVE_BASE
    {   // the fused chain's form: the NTT-form digit of row k is the product diag (.) diag_b formed while loading; base vs working copy
        u64* t2; CHECK(hipMalloc(&t2, dig_words * 8));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, t2, dig_words, N, b.d_moduli, L, 1u, 0x4444ull);
        std::vector<u64> r0(out_words), r1(out_words);
        for (int rep = 0; rep < 2; rep++) {
            KsMacArgs kv = ka; kv.digits = (const u64*)d_digf; kv.diag_b = t2;
            kv.out = b.out_ref;
            float tb = time_launch([&] { hipLaunchKernelGGL((ksmac2_base_kernel<14, true>), dim3(blocks * 2), dim3(KSB_THREADS), 0, 0, kv); }, reps);
            kv.out = b.out;
            float tn = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); }, reps);
            CHECK(hipMemcpy(r0.data(), b.out_ref, out_words * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(r1.data(), b.out, out_words * 8, hipMemcpyDeviceToHost));
            printf("%-28s %9.1f us   working copy %9.1f us  %s\n", "BASE fused form (diag_b)", tb, tn, r0 == r1 ? "identical" : "MISMATCH");
            if (rep == 1) {
                // tensor terms folded into the epilogue (KsMacArgs::ten_a; the keys' qk^-1 factor does not change the timing): full, operand loads from one line, no products
                u64* big; CHECK(hipMalloc(&big, 4 * dig_words * 8));      // [a0 | a1 | b0 | b1], each [item][L][N]
                for (int q = 0; q < 4; q++) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, big + q * dig_words, dig_words, N, b.d_moduli, L, 1u, 0x5555ull + q);
                KsMacArgs kt = kv; kt.diag = big + dig_words; kt.diag_b = big + 3 * dig_words;
                kt.ten_a = big; kt.ten_b = big + 2 * dig_words; kt.ten_bstride = ka.diag_bstride; kt.ten_pstride = (long long)dig_words;
                kt.diag_keys = d_keys;     // any [L][2][N] doubles will do for the timing
                { KsMacArgs k0 = kt; k0.ten_a = nullptr; k0.ten_b = nullptr;
                  float t_none = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, k0); }, reps);
                  printf("same operands, no tensor terms %9.1f us\n", t_none); }
                float t_full = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 0, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kt); }, reps);
                float t_l = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 128, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kt); }, reps);
                float t_p = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 256, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kt); }, reps);
                float t_lp = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 384, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kt); }, reps);
                float t_9 = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 512, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kt); }, reps);
                printf("tensor folded, a1 / b1 from one line %9.1f us\n", t_9);
                for (unsigned ord : {1u, 3u}) { KsMacArgs ko = kt; ko.grouped = ord; KsMacArgs kn = ko; kn.ten_a = nullptr; kn.ten_b = nullptr;
                  float t_o = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 0, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, ko); }, reps);
                  float t_n = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kn); }, reps);
                  printf("order %u: tensor folded %9.1f us   without %9.1f us\n", ord, t_o, t_n); }
                printf("tensor folded: full %9.1f us   loads from one line %9.1f us   no products %9.1f us   neither %9.1f us\n", t_full, t_l, t_p, t_lp);
                {   // third generation (wave-specialised, persistent): word-for-word against ksmac2's TEN instantiation, both orders
                    int ncu = 256; { hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount / 8 * 8; }
                    if (getenv("KSB_GRID")) ncu = atoi(getenv("KSB_GRID"));
                    for (unsigned ord : {3u, 1u}) {
                        KsMacArgs ko = kt; ko.grouped = ord;
                        const unsigned grid_rows = ord == 3 ? (L + 2) / 2 * 2 : L + 1;
                        const unsigned total_vb = (unsigned)(B * grid_rows * 2);
                        ko.out = b.out_ref;
                        CHECK(hipMemset(b.out_ref, 0xee, out_words * 8));
                        float t2 = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 0, false, true>), dim3(total_vb), dim3(KSM_THREADS), 0, 0, ko); }, reps);
                        ko.out = b.out;
                        CHECK(hipMemset(b.out, 0xff, out_words * 8));
                        float t3 = time_launch([&] { hipLaunchKernelGGL((ksmac3_kernel<14, true, false, 1>), dim3(ncu), dim3(KSM3_THREADS), 0, 0, ko, total_vb); }, reps);
                        CHECK(hipMemcpy(r0.data(), b.out_ref, out_words * 8, hipMemcpyDeviceToHost));
                        CHECK(hipMemcpy(r1.data(), b.out, out_words * 8, hipMemcpyDeviceToHost));
                        size_t bad = 0, first = 0;
                        for (size_t i = 0; i < out_words; i++) if (r0[i] != r1[i]) { if (!bad) first = i; bad++; }
                        printf("order %u: ksmac2 TEN %9.1f us   ksmac3 (grid %d) %9.1f us   %s", ord, t2, ncu, t3, bad ? "MISMATCH" : "identical");
                        if (bad) printf(" (%zu words, first at %zu: got %llu want %llu)", bad, first, (unsigned long long)r1[first], (unsigned long long)r0[first]);
                        printf("\n");
                        CHECK(hipMemset(b.out, 0xff, out_words * 8));
                        float t4 = time_launch([&] { hipLaunchKernelGGL((ksmac4_kernel<14, true, false, 1>), dim3(ncu), dim3(KSM3_THREADS), 0, 0, ko, total_vb); }, reps);
                        CHECK(hipMemcpy(r1.data(), b.out, out_words * 8, hipMemcpyDeviceToHost));
                        bad = 0; first = 0;
                        for (size_t i = 0; i < out_words; i++) if (r0[i] != r1[i]) { if (!bad) first = i; bad++; }
                        printf("order %u: ksmac4 (two-stage pipeline, key prefetch %d) %9.1f us   %s", ord, KSM4_KEY_AHEAD, t4, bad ? "MISMATCH" : "identical");
                        if (bad) printf(" (%zu words, first at %zu: got %llu want %llu)", bad, first, (unsigned long long)r1[first], (unsigned long long)r0[first]);
                        printf("\n");
                    }
                }
                CHECK(hipFree(big));
            }
        }
        // restore the reference output of the plain form for the variants below
        hipLaunchKernelGGL((ks_mac_kernel<ArithF64, 14, 4>), dim3(blocks), dim3(1024), 0, 0, a, b.kp);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(t2));
    }
    for (int rep = 0; rep < 2; rep++) {
    run_variant("BASE ksmac2 f64 digits", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out; kv.digits = (const u64*)d_digf;
        hipLaunchKernelGGL((ksmac2_base_kernel<14, true>), dim3(blocks * 2), dim3(KSB_THREADS), 0, 0, kv); });
    run_variant("ksmac2 f64 digits", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out; kv.digits = (const u64*)d_digf;
        hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    }
#endif
    run_variant("ksmac2 f64 digits", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out; kv.digits = (const u64*)d_digf;
        hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    {   // 16 coefficients per thread, four waves per SIMD (ksmac5_experiment.hpp) against ksmac2's NODIAG instantiation: every digit transformed, no epilogue
        double* d_keys5; CHECK(hipMalloc(&d_keys5, (size_t)L * key_words * 8));
        // its round-2 / round-3 twiddle tables (layouts in the kernel's comments)
        std::vector<double> t2((size_t)K * 2 * 8 * 128, 0.0), t3((size_t)K * 2 * 8 * 1024, 0.0);
        for (unsigned i = 0; i < K; i++) {
            auto tw = [&](int beta, unsigned hh, unsigned hi) { return fwd[(size_t)i * N + (N >> (beta + 1)) + (hh << (12 - beta)) + hi]; };
            for (unsigned hh = 0; hh < 2; hh++) for (unsigned w = 0; w < 8; w++) {
                for (unsigned j = 0; j < 16; j++) {
                    const unsigned hi6 = (w << 4) | j;
                    double* v = &t2[(((size_t)(i * 2 + hh) * 8 + w) * 16 + j) * 8];
                    v[1] = tw(5, hh, hi6);
                    for (unsigned g = 0; g < 2; g++) v[2 + g] = tw(4, hh, (hi6 << 1) | g);
                    for (unsigned g = 0; g < 4; g++) v[4 + g] = tw(3, hh, (hi6 << 2) | g);
                }
                for (unsigned ln = 0; ln < 64; ln++) {
                    const unsigned hi4 = (w << 6) | ln;
                    double* blk = &t3[((size_t)(i * 2 + hh) * 8 + w) * 1024];
                    auto put = [&](unsigned slot, double val) { blk[((slot >> 1) * 64 + ln) * 2 + (slot & 1)] = val; };
                    for (unsigned g = 0; g < 2; g++) put(g, tw(2, hh, (hi4 << 1) | g));
                    for (unsigned g = 0; g < 4; g++) put(2 + g, tw(1, hh, (hi4 << 2) | g));
                    for (unsigned g = 0; g < 8; g++) put(6 + g, tw(0, hh, (hi4 << 3) | g));
                }
            }
        }
        double *d_t2, *d_t3;
        CHECK(hipMalloc(&d_t2, t2.size() * 8)); CHECK(hipMemcpy(d_t2, t2.data(), t2.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipMalloc(&d_t3, t3.size() * 8)); CHECK(hipMemcpy(d_t3, t3.data(), t3.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ksm5_prepare_keys_kernel, dim3(1024), dim3(256), 0, 0, b.kp, L, 2 * K, N, d_keys5);
        std::vector<u64> r0(out_words), r1(out_words);
        for (int rep = 0; rep < 2; rep++) {
            KsMacArgs k2 = ka; k2.digits = (const u64*)d_digf; k2.diag = nullptr; k2.out = b.out_ref; k2.grouped = 1;
            float t2 = time_launch([&] { hipLaunchKernelGGL((ksmac2_kernel<14, true, 0, false, false, false, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, k2); }, reps);
            KsMacArgs k5 = k2; k5.keys = d_keys5; k5.out = b.out; k5.tw_r1 = d_t2; k5.tw_r2 = d_t3;
            CHECK(hipMemset(b.out, 0xff, out_words * 8));
            float t5 = time_launch([&] { hipLaunchKernelGGL((ksmac5_kernel<14>), dim3(blocks * 2), dim3(KSM5_THREADS), 0, 0, k5); }, reps);
            CHECK(hipMemcpy(r0.data(), b.out_ref, out_words * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(r1.data(), b.out, out_words * 8, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < out_words; i++) if (r0[i] != r1[i]) { if (!bad) first = i; bad++; }
            printf("item order: ksmac2 NODIAG %9.1f us   ksmac5 (16 coefficients x 512 threads, 4 waves / SIMD) %9.1f us   %s", t2, t5, bad ? "MISMATCH" : "identical");
            if (bad) printf(" (%zu words, first at %zu: got %llu want %llu)", bad, first, (unsigned long long)r1[first], (unsigned long long)r0[first]);
            printf("\n");
        }
        // restore the reference output of the plain form
        hipLaunchKernelGGL((ks_mac_kernel<ArithF64, 14, 4>), dim3(blocks), dim3(1024), 0, 0, a, b.kp);
        CHECK(hipDeviceSynchronize());
        CHECK(hipFree(d_keys5));
    }
#ifdef KSM_PHASE_PROFILE
    {   // shader cycles of wave 0 of every workgroup per phase (s_memtime), one launch
        unsigned long long* d_prof; CHECK(hipMalloc(&d_prof, 9 * 8)); CHECK(hipMemset(d_prof, 0, 9 * 8));
        KsMacArgs kv = ka; kv.out = b.out; kv.digits = (const u64*)d_digf; kv.prof = d_prof;
        hipLaunchKernelGGL((ksmac2_kernel<14, true>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv);
        unsigned long long hp[9]; CHECK(hipMemcpy(hp, d_prof, 72, hipMemcpyDeviceToHost));
        const char* names[9] = {"load+layer0", "round0", "exchange0 (2 barriers)", "round1", "exchange1", "round2", "mac", "diag digit", "epilogue"};
        double tot = 0; for (int i = 0; i < 9; i++) tot += (double)hp[i];
        for (int i = 0; i < 9; i++) printf("phase %-24s %10.0f cycles per workgroup  %5.1f %%\n", names[i], (double)hp[i] / (blocks * 2), 100.0 * hp[i] / tot);
        printf("phase total %.0f cycles per workgroup\n", tot / (blocks * 2));
    }
#endif
#ifndef KSBENCH_LEAN
    run_variant("ksmac2 1wg/cu", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out;
        hipLaunchKernelGGL((ksmac2_kernel<14, false>), dim3(blocks * 2), dim3(KSM_THREADS), 30000, 0, kv); });
    run_variant("ksmac2 prio", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out;
        hipLaunchKernelGGL((ksmac2_kernel<14, false, 64>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    run_variant("ksmac2 row-major", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out; kv.grouped = 2;
        hipLaunchKernelGGL((ksmac2_kernel<14, false>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    run_variant("ksmac2 ungrouped", [&](const NttArgs& v) {
        KsMacArgs kv = ka; kv.out = v.out; kv.grouped = 0;
        hipLaunchKernelGGL((ksmac2_kernel<14, false>), dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); });
    // ---- plain forward / inverse NTT of B*(L+1)*L/2 limb-polynomials: register-block size A/B (profiles/r02_ntt_ab.txt) ----
    if (wanted("ntt")) {
        const size_t lp = B * 15;                     // limb-polynomials per launch
        u64 *nin, *nout;
        CHECK(hipMalloc(&nin, lp * N * 8)); CHECK(hipMalloc(&nout, lp * N * 8));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, nin, lp * N, N, b.d_moduli, L, 1u, 0x7777ull);
        NttArgs na;
        std::memset(&na, 0, sizeof(na));
        na.in = nin; na.out = nout; na.mods = b.d_mods; na.tw = b.d_fwd;
        na.in_cstride = na.out_cstride = N; na.in_pstride = na.out_pstride = (long long)L * N; na.in_bstride = na.out_bstride = (long long)L * N;
        na.pcount = 1; na.ncomp = L; na.table_start = 0; na.table_count = L; na.mode = 0; na.stream_loads = 1;
        std::vector<u64> r4(lp * N), r5(lp * N);
        float t4 = time_launch([&] { hipLaunchKernelGGL((ntt_pass_kernel<ArithF64, 14, 0, 14, 14, 4, false, true, true, 0>), dim3((unsigned)lp), dim3(1024), 0, 0, na); }, reps);
        CHECK(hipMemcpy(r4.data(), nout, lp * N * 8, hipMemcpyDeviceToHost));
        float t5 = time_launch([&] { hipLaunchKernelGGL((ntt_pass_kernel<ArithF64, 14, 0, 14, 14, 5, false, true, true, 0>), dim3((unsigned)lp), dim3(512), 0, 0, na); }, reps);
        CHECK(hipMemcpy(r5.data(), nout, lp * N * 8, hipMemcpyDeviceToHost));
        const double gb = 16.0 * N * lp / 1e3;
        std::vector<u64> rs(lp * N);
        float ts = time_launch([&] { hipLaunchKernelGGL(ntt_fwd14_shfl_kernel, dim3((unsigned)lp), dim3(1024), 0, 0, na); }, reps);
        CHECK(hipMemcpy(rs.data(), nout, lp * N * 8, hipMemcpyDeviceToHost));
        float t4b = time_launch([&] { hipLaunchKernelGGL((ntt_pass_kernel<ArithF64, 14, 0, 14, 14, 4, false, true, true, 0>), dim3((unsigned)lp), dim3(1024), 0, 0, na); }, reps);
        printf("%-28s %9.1f us  %7.1f GB/s  %s\n", "ntt fwd 16/thr, DPP quad transpose for the wave-private exchange", ts, gb / ts, r4 == rs ? "bit-exact" : "MISMATCH");
        printf("%-28s %9.1f us  %7.1f GB/s  (LDS exchange, repeated)\n", "ntt fwd 16/thr x1024 (3 xchg)", t4b, gb / t4b);
        printf("%-28s %9.1f us  %7.1f GB/s\n", "ntt fwd 16/thr x1024 (3 xchg)", t4, gb / t4);
        printf("%-28s %9.1f us  %7.1f GB/s  %s\n", "ntt fwd 32/thr x512 (2 xchg)", t5, gb / t5, r4 == r5 ? "bit-exact" : "MISMATCH");
    }
    // ablations (timing only; results are wrong by design)
    auto run_abl = [&](const char* name, auto kern) {
        if (!wanted("abl")) return;
        KsMacArgs kv = ka; kv.out = b.out;
        float t = time_launch([&] { hipLaunchKernelGGL(kern, dim3(blocks * 2), dim3(KSM_THREADS), 0, 0, kv); }, reps);
        printf("%-28s %9.1f us\n", name, t);
    };
    run_abl("abl: no digit loads", ksmac2_kernel<14, false, 1>);
    run_abl("abl: no key loads", ksmac2_kernel<14, false, 2>);
    run_abl("abl: no LDS", ksmac2_kernel<14, false, 4>);
    run_abl("abl: no butterflies", ksmac2_kernel<14, false, 8>);
    run_abl("abl: no mac", ksmac2_kernel<14, false, 16>);
    run_abl("abl: no twiddle loads", ksmac2_kernel<14, false, 32>);
    run_abl("abl: no loads at all", ksmac2_kernel<14, false, 1 | 2 | 32>);
    run_abl("abl: alu only", ksmac2_kernel<14, false, 1 | 2 | 4 | 32>);
    run_abl("abl: mem only", ksmac2_kernel<14, false, 8 | 16>);
#endif
    return 0;
}
