// ring2k_kernels.hpp -- the Z_{2^k} polynomial encoder of the reference's BFV ring-2^k application (src/app/bfv_ring2k.cu): a
// plaintext modulus t = 2^k, k up to 128, handled outside the RNS context (the context's own plain modulus is not used).
//   scale_up     m -> round(Q/t * m) mod q_l              (encode for encryption)
//   centralize   m -> centred lift of m mod q_l            (encode for multiplication with a ciphertext)
//   scale_down   phase mod Q -> m mod t                    (decode: the BEHZ rounding of SEAL's rns.cpp with t = 2^k and an auxiliary prime gamma)
//   decentralize x mod Q -> x mod t                        (the inverse of centralize: an exact base conversion {q_l} -> {2^k} with a floating-point quotient estimate)
// Elements are 4, 8 or 16 bytes wide; every computation runs on 128-bit words and masks with 2^k - 1 at the end, which equals the
// reference's arithmetic in uint32 / uint64 / uint128 (reduction mod 2^k commutes with the wrap-around of the narrower types).
#pragma once
#include "poly_kernels.hpp"

namespace troyn {

typedef unsigned __int128 u128;

struct Ring2kDev {
    const DevModulus* mods;
    const ulonglong2* q_div_t_mod_q;      // [L] Shoup pairs of floor(Q / 2^k) mod q_l
    const ulonglong2* gamma_t_mod_q;      // [L] Shoup pairs of gamma * 2^k mod q_l
    const ulonglong2* inv_punctured;      // [L] Shoup pairs of (Q/q_l)^-1 mod q_l
    const u64* punctured_mod_gamma;       // [L] (Q/q_l) mod gamma
    const u64* punctured_mod_t;           // [L][2] (Q/q_l) mod 2^k as (lo, hi)
    DevModulus gamma;
    ulonglong2 neg_inv_q_mod_gamma;       // Shoup pair
    u64 q_mod_t[2], t_half[2], mask[2], neg_inv_q_mod_t[2], inv_gamma_mod_t[2];
    unsigned L, n, t_bits, elem_bytes;
};

__device__ __forceinline__ u128 ring2k_make(u64 lo, u64 hi) { return ((u128)hi << 64) | lo; }

__device__ __forceinline__ u128 ring2k_load(const void* src, size_t i, unsigned elem_bytes) {
    if (elem_bytes == 4) return (u128) reinterpret_cast<const unsigned*>(src)[i];
    if (elem_bytes == 8) return (u128) reinterpret_cast<const u64*>(src)[i];
    const u64* p = reinterpret_cast<const u64*>(src) + 2 * i;
    return ring2k_make(p[0], p[1]);
}

__device__ __forceinline__ void ring2k_store(void* dst, size_t i, unsigned elem_bytes, u128 v) {
    if (elem_bytes == 4) reinterpret_cast<unsigned*>(dst)[i] = (unsigned)v;
    else if (elem_bytes == 8) reinterpret_cast<u64*>(dst)[i] = (u64)v;
    else { u64* p = reinterpret_cast<u64*>(dst) + 2 * i; p[0] = (u64)v; p[1] = (u64)(v >> 64); }
}

__device__ __forceinline__ u64 ring2k_reduce(u128 x, const DevModulus& m) { return barrett128((u64)x, (u64)(x >> 64), m.q, m.ratio_lo, m.ratio_hi); }

// device_scale_up / device_scale_up_uint128 (bfv_ring2k.cu:197-297): out [L][n], coefficients >= count are zero
__global__ __launch_bounds__(256) void ring2k_scale_up_kernel(Ring2kDev c, const void* src, unsigned count, u64* out) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.n) return;
    if (j >= count) { for (unsigned l = 0; l < c.L; l++) out[(size_t)l * c.n + j] = 0; return; }
    const u128 x = ring2k_load(src, j, c.elem_bytes);
    const u128 qt = ring2k_make(c.q_mod_t[0], c.q_mod_t[1]), th = ring2k_make(c.t_half[0], c.t_half[1]);
    // v = floor((Q_mod_t * x + t/2) / 2^k): a 256-bit product, of which bits [k, k + 128) are kept
    const u64 a0 = (u64)x, a1 = (u64)(x >> 64), b0 = (u64)qt, b1 = (u64)(qt >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u64 w[4];
    w[0] = (u64)p00;
    u128 mid = (p00 >> 64) + (u64)p01 + (u64)p10;
    w[1] = (u64)mid;
    u128 hi = (mid >> 64) + (p01 >> 64) + (p10 >> 64) + (u64)p11;
    w[2] = (u64)hi;
    w[3] = (u64)((hi >> 64) + (p11 >> 64));
    {   // + t_half
        u128 s = (u128)w[0] + (u64)th; w[0] = (u64)s;
        s = (u128)w[1] + (u64)(th >> 64) + (u64)(s >> 64); w[1] = (u64)s;
        s = (u128)w[2] + (u64)(s >> 64); w[2] = (u64)s;
        w[3] += (u64)(s >> 64);
    }
    u128 v;
    if (c.elem_bytes <= 8) {
        // the reference keeps v in the element type: T v = (...) >> k, then reduce(u + v) in 64-bit arithmetic (:207-208)
        const u128 prod = ring2k_make(w[0], w[1]);
        v = (u128)(u64)(c.elem_bytes == 4 ? (u64)(unsigned)(prod >> c.t_bits) : (u64)(prod >> c.t_bits));
    } else {
        // drop the low word, shift the remaining 192 bits by k - 64 (k > 64), keep 128 bits (:262-266)
        const unsigned sh = c.t_bits - 64;
        const u128 lo192 = ring2k_make(w[1], w[2]);
        v = sh == 64 ? ring2k_make(w[2], w[3]) : ((lo192 >> sh) | ((u128)w[3] << (128 - sh)));
    }
    for (unsigned l = 0; l < c.L; l++) {
        const DevModulus m = c.mods[l];
        const ulonglong2 d = c.q_div_t_mod_q[l];
        const u64 x64 = c.elem_bytes == 16 ? ring2k_reduce(x, m) : barrett64((u64)x, m.q, m.ratio_hi);
        const u64 u = shoup_mul(x64, d.x, d.y, m.q);
        out[(size_t)l * c.n + j] = c.elem_bytes == 16 ? ring2k_reduce((u128)u + v, m) : barrett64(u + (u64)v, m.q, m.ratio_hi);
    }
}

// device_centralize (:488-506)
__global__ __launch_bounds__(256) void ring2k_centralize_kernel(Ring2kDev c, const void* src, unsigned count, u64* out) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.n) return;
    if (j >= count) { for (unsigned l = 0; l < c.L; l++) out[(size_t)l * c.n + j] = 0; return; }
    const u128 x = ring2k_load(src, j, c.elem_bytes);
    const u128 th = ring2k_make(c.t_half[0], c.t_half[1]), mask = ring2k_make(c.mask[0], c.mask[1]);
    for (unsigned l = 0; l < c.L; l++) {
        const DevModulus m = c.mods[l];
        if (x > th) out[(size_t)l * c.n + j] = neg_mod(ring2k_reduce((u128)(0 - x) & mask, m), m.q);
        else out[(size_t)l * c.n + j] = ring2k_reduce(x, m);
    }
}

// PolynomialEncoderRNSHelper::scale_down (:659-735), one thread per coefficient: in [L][n] (coefficient form, mod Q) -> elements
__global__ __launch_bounds__(256) void ring2k_scale_down_kernel(Ring2kDev c, const u64* in, void* dst) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    const u128 mask = ring2k_make(c.mask[0], c.mask[1]);
    u128 on_t = 0;
    u64 glo = 0, ghi = 0;
    for (unsigned l = 0; l < c.L; l++) {
        const DevModulus m = c.mods[l];
        const ulonglong2 gt = c.gamma_t_mod_q[l], ip = c.inv_punctured[l];
        const u64 scaled = shoup_mul(in[(size_t)l * c.n + i], gt.x, gt.y, m.q);                 // 1. times gamma * t
        const u64 y = (ip.x == 1) ? barrett64(scaled, m.q, m.ratio_hi) : shoup_mul(scaled, ip.x, ip.y, m.q);   // fast base conversion, step 1
        mac128(glo, ghi, y, c.punctured_mod_gamma[l]);                                          // 2-1 towards {gamma}
        on_t += (u128)y * ring2k_make(c.punctured_mod_t[2 * l], c.punctured_mod_t[2 * l + 1]);  // 3-1 towards {t} (wraps mod 2^128)
    }
    u64 on_gamma = barrett128(glo, ghi, c.gamma.q, c.gamma.ratio_lo, c.gamma.ratio_hi);
    on_gamma = shoup_mul(on_gamma, c.neg_inv_q_mod_gamma.x, c.neg_inv_q_mod_gamma.y, c.gamma.q);                 // 2-2
    on_t = (on_t * ring2k_make(c.neg_inv_q_mod_t[0], c.neg_inv_q_mod_t[1])) & mask;                              // 3-2
    const u128 ig = ring2k_make(c.inv_gamma_mod_t[0], c.inv_gamma_mod_t[1]);
    // 4. subtract the centred remainder mod gamma, divide by gamma mod t
    const u128 r = (on_gamma > (c.gamma.q >> 1)) ? ((on_t + c.gamma.q - on_gamma) * ig) & mask : ((on_t - on_gamma) * ig) & mask;
    ring2k_store(dst, i, c.elem_bytes, r);
}

// PolynomialEncoderRNSHelper::decentralize (:752-911; steps 1 and 2 of the reference in one pass, one thread per coefficient): in [L][n] (coefficient form) -> elements.
//   y_l = x_l * (Q/q_l)^-1 mod q_l;   v = round(sum_l y_l / q_l) (doubles, summed in the order l = 0 .. L-1 as the reference does);
//   out = (sum_l y_l * ((Q/q_l) mod 2^k) - v * (Q mod 2^k)) * fix  mod 2^k,   fix = correction_factor^-1 mod 2^k (1 when there is none)
__global__ __launch_bounds__(256) void ring2k_decentralize_kernel(Ring2kDev c, const u64* in, void* dst, u64 fix_lo, u64 fix_hi) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.n) return;
    const u128 mask = ring2k_make(c.mask[0], c.mask[1]);
    u128 acc = 0;
    double v = 0.0;
    for (unsigned l = 0; l < c.L; l++) {
        const DevModulus m = c.mods[l];
        const ulonglong2 ip = c.inv_punctured[l];
        const u64 x = in[(size_t)l * c.n + i];
        const u64 y = (ip.x == 1) ? barrett64(x, m.q, m.ratio_hi) : shoup_mul(x, ip.x, ip.y, m.q);
        v += (double)y / (double)m.q;
        acc += (u128)y * ring2k_make(c.punctured_mod_t[2 * l], c.punctured_mod_t[2 * l + 1]);
    }
    const u128 rounded = (u128)(u64)round(v);                      // v < L <= 64
    u128 r = (acc - rounded * ring2k_make(c.q_mod_t[0], c.q_mod_t[1])) & mask;
    r = (r * ring2k_make(fix_lo, fix_hi)) & mask;
    ring2k_store(dst, i, c.elem_bytes, r);
}

}  // namespace troyn
