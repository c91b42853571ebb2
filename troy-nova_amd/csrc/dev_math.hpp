// dev_math.hpp -- 64-bit modular arithmetic for gfx950 integer ALUs (no MFMA: this is not a
// dense FP contraction).  Value semantics follow the reference's scalar layer:
//   Barrett-64 / Barrett-128   modulus.h:22-78
//   Shoup multiply (lazy/full) utils/uint_small_mod.h:130-148
//   add/sub with ONE correction utils/uint_small_mod.h:54-72
// so every canonical result is bit-identical to the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace troyn {

typedef unsigned long long u64;  // matches ulonglong2 element type
typedef unsigned __int128 u128;

// Per-modulus constants resident in HBM (read through the scalar/vector caches).
struct DevModulus {
    u64 q;
    u64 ratio_lo;   // floor(2^128/q) low word   (const_ratio[0])
    u64 ratio_hi;   // floor(2^128/q) high word  (const_ratio[1])
    u64 inv_n_op;   // N^-1 mod q                (NTTTables::inv_degree_modulo().operand)
    u64 inv_n_quo;  //                            (.quotient)
    // FP64 fast path (valid when q < 2^50, see dev_math_f64.hpp)
    double pd;        // (double) q
    double inv_pd;    // fl(1 / q)
    double inv_n_d;   // (double) N^-1 mod q
    double inv_n_pd;  // fl(inv_n_d / q)
    double inv_n_w_d;   // (N^-1 * w_last) mod q, w_last = the single twiddle of the final inverse layer
    double inv_n_w_pd;  // fl(inv_n_w_d / q)
    u64 pad_;
};
static_assert(sizeof(DevModulus) == 96, "DevModulus layout");

__device__ __forceinline__ u64 mul_hi(u64 a, u64 b) { return __umul64hi(a, b); }
// full 64 x 64 -> 128 product in one piece: the compiler builds it from four v_mad_u64_u32; asking for a*b and
// __umul64hi(a, b) separately costs seven multiplier instructions instead
__device__ __forceinline__ void mul128(u64 a, u64 b, u64& lo, u64& hi) {
    const u128 p = (u128)a * b;
    lo = (u64)p; hi = (u64)(p >> 64);
}

// multiply_uint64operand_mod_lazy: result in [0, 2q) for any 64-bit x
__device__ __forceinline__ u64 shoup_lazy(u64 x, u64 w, u64 wq, u64 q) { return w * x - mul_hi(x, wq) * q; }

// multiply_uint64operand_mod: canonical
__device__ __forceinline__ u64 shoup_mul(u64 x, u64 w, u64 wq, u64 q) {
    u64 r = shoup_lazy(x, w, wq, q);
    return r >= q ? r - q : r;
}

// Modulus::reduce (Barrett-64 with const_ratio[1])
__device__ __forceinline__ u64 barrett64(u64 x, u64 q, u64 ratio_hi) {
    u64 r = x - mul_hi(x, ratio_hi) * q;
    return r >= q ? r - q : r;
}

// Modulus::reduce_uint128_limbs, same word-level sequence as modulus.h:44-78
__device__ __forceinline__ u64 barrett128(u64 in0, u64 in1, u64 q, u64 r0, u64 r1) {
    u64 carry = mul_hi(in0, r0);
    u64 t2lo, t2hi;
    mul128(in0, r1, t2lo, t2hi);
    u64 tmp1 = t2lo + carry;
    u64 tmp3 = t2hi + (tmp1 < t2lo ? 1ull : 0ull);
    mul128(in1, r0, t2lo, t2hi);
    u64 tmp1b = tmp1 + t2lo;
    carry = t2hi + (tmp1b < tmp1 ? 1ull : 0ull);
    u64 quot = in1 * r1 + tmp3 + carry;
    u64 r = in0 - quot * q;
    return r >= q ? r - q : r;
}

__device__ __forceinline__ u64 mul_mod(u64 a, u64 b, const DevModulus& m) {
    u64 lo, hi;
    mul128(a, b, lo, hi);
    return barrett128(lo, hi, m.q, m.ratio_lo, m.ratio_hi);
}

__device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 q) { u64 s = a + b; return s >= q ? s - q : s; }
__device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 q) { u64 d = a - b; return a < b ? d + q : d; }
__device__ __forceinline__ u64 neg_mod(u64 a, u64 q) { return a == 0 ? 0 : q - a; }

// 128-bit accumulate helper for lazy dot products
__device__ __forceinline__ void mac128(u64& lo, u64& hi, u64 a, u64 b) {
    u128 acc = ((u128)hi << 64) | lo;
    acc += (u128)a * b;
    lo = (u64)acc; hi = (u64)(acc >> 64);
}

}  // namespace troyn
