// Host-side exactness check of the FP64 modular arithmetic used by the gfx950 NTT fast path
// (troy-nova_amd/csrc/dev_math_f64.hpp) against the CPU oracle: runs the SAME per-element
// functions layer by layer with the kernel's re-centring schedule (every 4 layers; inverse also
// after 2) and compares bit-for-bit with orc_ntt_forward / orc_ntt_inverse.
//   g++ -O2 -std=c++17 -ffp-contract=off -mfma -I troy-nova_amd/csrc tools/fp64_check.cpp oracle/troy_oracle.c ...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define __forceinline__
typedef unsigned long long u64_;
namespace troyn { typedef unsigned long long u64; }
#define TROYN_NO_HIP
#include "../oracle/troy_oracle.h"

// minimal copy-free include of the arithmetic: dev_math_f64.hpp pulls dev_math.hpp (HIP); provide the pieces it needs
namespace troyn {
#define TROYN_HD inline
constexpr u64 F64_MODULUS_LIMIT = 1ull << 50;
constexpr double F64_TWO52 = 4503599627370496.0;
struct F64Mod { double p, inv_p; };
}
#define TROYN_F64_BODY_ONLY
#include "fp64_body.inc"

using namespace troyn;

static int check(unsigned log_n, u64 q, unsigned seed, int adversarial) {
    const size_t n = (size_t)1 << log_n;
    orc_ntt_tables* t = orc_ntt_tables_create(log_n, q);
    if (!t) { printf("no tables\n"); return 1; }
    std::vector<uint64_t> in(n), ref(n);
    orc_fill_uniform(seed, adversarial ? 0 : q, in.data(), n);
    if (adversarial) for (size_t i = 0; i < n; i++) in[i] = (in[i] & 1) ? q - 1 - (in[i] % 3) : (in[i] % 4);
    const F64Mod m{(double)q, 1.0 / (double)q};
    const double p = m.p;
    int bad = 0;
    // ---- forward ----
    ref = in;
    const orc_ntt_tables* tt = t;
    orc_ntt_forward(ref.data(), 1, 1, log_n, &tt, 1, 0, 0);
    std::vector<double> x(n);
    for (size_t i = 0; i < n; i++) x[i] = f64_corr(f64_from_u64(in[i]), m);
    double maxabs = 0;
    for (unsigned layer = 0; layer < log_n; layer++) {
        const size_t mm = (size_t)1 << layer, gap = n >> (layer + 1);
        for (size_t g = 0; g < mm; g++) {
            const double w = (double)t->root_powers[mm + g].operand, wp = w * m.inv_p;
            for (size_t j = 0; j < gap; j++) {
                const size_t a = 2 * g * gap + j, b = a + gap;
                const double r = f64_mulc(x[b], w, wp, p);
                const double xa = x[a];
                x[a] = xa + r; x[b] = xa - r;
                if (__builtin_fabs(x[a]) > maxabs) maxabs = __builtin_fabs(x[a]);
                if (__builtin_fabs(x[b]) > maxabs) maxabs = __builtin_fabs(x[b]);
            }
        }
        if ((layer + 1) % 4 == 0) for (size_t i = 0; i < n; i++) x[i] = f64_corr(x[i], m);
    }
    for (size_t i = 0; i < n; i++) if (f64_canon(x[i], m) != ref[i]) { if (bad < 3) printf("fwd mismatch i=%zu\n", i); bad++; }
    printf("logn=%u q=%llu fwd: max|x|/p = %.3f  mismatches=%d\n", log_n, (u64)q, maxabs / p, bad);
    // ---- inverse (input = canonical NTT-form ref) ----
    std::vector<uint64_t> inv_ref = ref;
    orc_ntt_inverse(inv_ref.data(), 1, 1, log_n, &tt, 1, 0, 0);
    for (size_t i = 0; i < n; i++) x[i] = f64_corr(f64_from_u64(ref[i]), m);
    maxabs = 0;
    int bad2 = 0;
    for (unsigned layer = 0; layer < log_n; layer++) {
        const size_t gap = (size_t)1 << layer, mm = n >> (layer + 1);
        for (size_t g = 0; g < mm; g++) {
            const double w = (double)t->inv_root_powers[n - 2 * mm + 1 + g].operand, wp = w * m.inv_p;
            for (size_t j = 0; j < gap; j++) {
                const size_t a = 2 * g * gap + j, b = a + gap;
                const double u = x[a], v = x[b];
                x[a] = u + v;
                x[b] = f64_mulc(u - v, w, wp, p);
                if (__builtin_fabs(x[a]) > maxabs) maxabs = __builtin_fabs(x[a]);
                if (__builtin_fabs(u - v) > maxabs) maxabs = __builtin_fabs(u - v);
            }
        }
        if ((layer + 1) % 4 == 2) {   // mid-block re-centring of the sums
            for (size_t g = 0; g < mm; g++) for (size_t j = 0; j < gap; j++) { const size_t a = 2 * g * gap + j; x[a] = f64_corr(x[a], m); }
        }
        if ((layer + 1) % 4 == 0) for (size_t i = 0; i < n; i++) x[i] = f64_corr(x[i], m);
    }
    const double ninv = (double)t->inv_degree_modulo.operand, ninv_p = ninv * m.inv_p;
    for (size_t i = 0; i < n; i++) {
        const double r = f64_mulc(x[i], ninv, ninv_p, p);
        if (f64_canon(r, m) != inv_ref[i]) { if (bad2 < 3) printf("inv mismatch i=%zu\n", i); bad2++; }
        if (inv_ref[i] >= q) { printf("oracle INTT output not canonical at %zu\n", i); bad2++; }
    }
    printf("logn=%u q=%llu inv: max|x|/p = %.3f  mismatches=%d\n", log_n, (u64)q, maxabs / p, bad2);
    orc_ntt_tables_destroy(t);
    return bad + bad2;
}

int main() {
    int bad = 0;
    const u64 primes[] = {1125899903107073ull, 1125899904679937ull, 1099510824961ull, 1032193ull, 1125899906826241ull /* largest 50-bit prime = 1 mod 2^16 */};
    for (u64 q : primes) {
        if (q >= F64_MODULUS_LIMIT) { printf("skip %llu\n", q); continue; }
        for (unsigned ln : {5u, 13u, 14u, 15u}) {
            if ((q - 1) % (2ull << ln)) continue;
            bad += check(ln, q, 1234 + ln, 0);
            bad += check(ln, q, 99 + ln, 1);
        }
    }
    printf(bad ? "FAILED\n" : "ALL EXACT\n");
    return bad != 0;
}
