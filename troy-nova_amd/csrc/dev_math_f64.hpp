// dev_math_f64.hpp -- exact modular arithmetic on the FP64 vector ALU for moduli p < 2^50.
//
// Why: on gfx950 a 32x32 integer multiply (v_mul_lo/hi_u32, v_mad_u64_u32) and a 53x53-bit
// v_fma_f64 both issue at ~4.3 cycles per wave-instruction (tools/ubench/alu_rates.hip).  A 64-bit
// Shoup butterfly costs 10 integer multiplies + ~17 carry/select ops; the same butterfly on
// integer-valued doubles costs 6 FP64 ops + 2 adds.  This is still the vector ALU (no MFMA) and
// every operation below is EXACT integer arithmetic carried in doubles, so canonical results are
// bit-identical to the reference's Barrett/Shoup results:
//   * all values are integers of magnitude < 2^53, hence exactly representable;
//   * mulc(): h = fl(y*w), l = fma(y,w,-h) is the error-free product (y*w = h + l exactly);
//     q = rint(fl(y*wp)) with wp = fl(w * fl(1/p)) (relative error <= 1.5 * 2^-52) is within
//     0.5 + 1.5*|y|*2^-52 of y*w/p; h - q*p is an integer below 2^53 so fma(-q,p,h) is exact, and
//     r = (h - q*p) + l = y*w - q*p exactly, |r| <= (0.5 + 1.5*|y|*2^-52) * p.
//   * corr(): x - rint(x/p)*p, exact for |x| < 2^53, result |x| <= 0.5p + 1.
// Bounds used by the NTT kernels (p < 2^50): forward blocks of 4 Cooley-Tukey layers start with
// |x| <= 0.5p+1 and stay below 5.2p; inverse blocks of 4 Gentleman-Sande layers re-centre the sums
// after 2 layers and stay below 4.5p; both are < 2^53 = 8 * 2^50 (tools/fp64_check.cpp prints the
// maxima actually reached).
#pragma once
#include "dev_math.hpp"

#if defined(__HIPCC__)
#define TROYN_HD __host__ __device__ __forceinline__
#else
#define TROYN_HD inline
#endif

// Exactness rests on SEPARATE roundings of fl(y*w) and fma(y, w, -h): a compiler that contracts `a * b + c` into an FMA on its own
// would change h and break the error-free product.  The build passes -ffp-contract=off (csrc/Makefile); the pragmas below make the
// property travel with this header for any translation unit that includes it without that flag (tools/, one-line hipcc builds).
// Every FMA this file wants is written as __builtin_fma.
#pragma clang fp contract(off)

namespace troyn {

constexpr u64 F64_MODULUS_LIMIT = 1ull << 50;   // FP64 path is used only for p < 2^50
constexpr double F64_TWO52 = 4503599627370496.0;

struct F64Mod {
    double p;      // modulus
    double inv_p;  // fl(1/p)
};

TROYN_HD double f64_bits_to_double(u64 b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
TROYN_HD u64 f64_double_to_bits(double d) { u64 b; __builtin_memcpy(&b, &d, 8); return b; }

// integer v < 2^52 -> double (exact)
TROYN_HD double f64_from_u64(u64 v) { return f64_bits_to_double(v | 0x4330000000000000ull) - F64_TWO52; }
// integer-valued double 0 <= x < 2^52 -> u64 (exact)
TROYN_HD u64 f64_to_u64(double x) { return f64_double_to_bits(x + F64_TWO52) & 0x000FFFFFFFFFFFFFull; }

// re-centre: |result| <= 0.5p + 1
TROYN_HD double f64_corr(double x, const F64Mod& m) {
    const double k = __builtin_rint(x * m.inv_p);
    return __builtin_fma(-k, m.p, x);
}

// y * w mod p (lazy, signed): w integer in [0,p), wp = fl(w/p)
TROYN_HD double f64_mulc(double y, double w, double wp, double p) {
    const double q = __builtin_rint(y * wp);
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double t = __builtin_fma(-q, p, h);
    return t + l;
}

// y * w mod p (lazy, signed) with the quotient estimated from the rounded product itself: q = rint(fl(y*w) * fl(1/p)) is
// within 0.5 + 1.5 |y| 2^-52 of y*w/p (the bound of f64_mulc) and needs no per-operand w/p.  w any integer in [0, p).
TROYN_HD double f64_mulq(double y, double w, double inv_p, double p) {
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double q = __builtin_rint(h * inv_p);
    return __builtin_fma(-q, p, h) + l;
}

// canonical representative in [0, p) as u64
TROYN_HD u64 f64_canon(double x, const F64Mod& m) {
    x = f64_corr(x, m);
    x = (x < 0.0) ? x + m.p : x;
    return f64_to_u64(x);
}

}  // namespace troyn
