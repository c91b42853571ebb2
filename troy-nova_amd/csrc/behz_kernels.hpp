// behz_kernels.hpp -- BEHZ RNS base conversions for the BFV multiply (integer-ALU bound:
// O(L * |Bsk|) 64x64->128 multiply-accumulates per coefficient; no MFMA).
//
// Replace fgk/rns_tool.cu:7-100 (kernel_fast_b_conv_m_tilde_sm_mrq) and :147-286
// (kernel_fast_floor_fast_b_conv_sk).  The reference kernels spill their per-coefficient
// intermediate vector to an AoS scratch buffer in global memory (fgk/rns_tool.cu:45,55);
// here the vector lives in registers (array length is a template bound) and every input is
// read, and every output written, exactly once, coalesced along the coefficient index.
// Value semantics follow the host functions utils/rns_tool.cu:762-790 (fast_b_conv_sk),
// :870-905 (sm_mrq), :973-988 (fast_floor), :1083-1094 (fast_b_conv_m_tilde) and
// utils/rns_base.cu:350-380 (fast_convert_array).
#pragma once
#include "dev_math.hpp"

namespace troyn {

struct BehzDev {
    unsigned L, Bn, Bsk, n;
    u64 t;                                  // plain modulus value
    const DevModulus* q_mods;               // [L]
    const DevModulus* bsk_mods;             // [Bsk]  (B primes then m_sk)
    DevModulus m_tilde;                     // 2^32 (non-prime Modulus, rns_tool.cu:83)
    const ulonglong2* q_inv_punc;           // [L]   (q/q_i)^-1 mod q_i            (Shoup)
    const ulonglong2* q_mt_inv_punc;        // [L]   m_tilde * (q/q_i)^-1 mod q_i  (Shoup): lift, scalar factor folded in
    const ulonglong2* q_t_inv_punc;         // [L]   t * (q/q_i)^-1 mod q_i        (Shoup): floor, scalar factor folded in
    const ulonglong2* t_inv_prod_q_mod_bsk; // [Bsk] t * q^-1 mod p_b              (Shoup)
    const u64* q_to_bsk;                    // [Bsk][L] (q/q_i) mod p_b
    const u64* q_to_mt;                     // [L]   (q/q_i) mod m_tilde
    ulonglong2 neg_inv_prod_q_mod_mt;       // -q^-1 mod m_tilde                    (Shoup wrt m_tilde)
    const ulonglong2* prod_q_mod_bsk;       // [Bsk] q mod p_b                      (Shoup)
    const ulonglong2* inv_mt_mod_bsk;       // [Bsk] m_tilde^-1 mod p_b             (Shoup)
    const ulonglong2* inv_prod_q_mod_bsk;   // [Bsk] q^-1 mod p_b                   (Shoup)
    const ulonglong2* B_inv_punc;           // [Bn]  (B/p_b)^-1 mod p_b             (Shoup)
    const u64* B_to_q;                      // [L][Bn] (B/p_b) mod q_i
    const u64* B_to_msk;                    // [Bn]  (B/p_b) mod m_sk
    ulonglong2 inv_prod_B_mod_msk;          // B^-1 mod m_sk                        (Shoup)
    const ulonglong2* prod_B_mod_q;         // [L]   B mod q_i                      (Shoup)
    const ulonglong2* neg_prod_B_mod_q;     // [L]   -B mod q_i                     (Shoup)
};

// The conversion matrices and per-modulus constants are never written by a kernel and every access has a
// wave-uniform index: reading them through the constant address space turns ~300 dependent vector loads per
// coefficient into scalar loads (SGPR operands of the multiplies).
typedef const u64 __attribute__((address_space(4)))* cu64p;
typedef u64 u64x2c __attribute__((ext_vector_type(2)));
typedef const u64x2c __attribute__((address_space(4)))* cu64x2p;
typedef const DevModulus __attribute__((address_space(4)))* cmodp;
template <typename T> __device__ __forceinline__ cu64p as_c64(const T* p) { return (cu64p)(unsigned long long)p; }
template <typename T> __device__ __forceinline__ cu64x2p as_c128(const T* p) { return (cu64x2p)(unsigned long long)p; }
__device__ __forceinline__ cmodp as_cmod(const DevModulus* p) { return (cmodp)(unsigned long long)p; }
__device__ __forceinline__ ulonglong2 ld_pair(cu64x2p p, unsigned i) { const u64x2c v = p[i]; return make_ulonglong2(v.x, v.y); }
__device__ __forceinline__ DevModulus ld_mod(cmodp p, unsigned i) {
    DevModulus m;
    m.q = p[i].q; m.ratio_lo = p[i].ratio_lo; m.ratio_hi = p[i].ratio_hi;
    return m;
}

// BaseConverter::fast_convert_array step 1: x * inv_punc mod q (barrett when the operand is 1,
// utils/rns_base.cu:358-366)
__device__ __forceinline__ u64 conv_scale(u64 x, const ulonglong2 op, const DevModulus& m) {
    return (op.x == 1) ? barrett64(x, m.q, m.ratio_hi) : shoup_mul(x, op.x, op.y, m.q);
}

// BEHZ steps (1)-(2): in [items][L][N] (coefficient form, base q) -> out [items][Bsk][N].
// one thread per coefficient; MAXL bounds L so y[] stays in VGPRs.
template <int MAXL>
__global__ __launch_bounds__(256) void behz_lift_kernel(unsigned chunks, BehzDev c, const u64* in, u64* out) {
    const unsigned n = c.n, L = c.L, Bsk = c.Bsk;
    const size_t item = blockIdx.x / chunks;
    const u64* ip = in + item * (size_t)L * n;
    u64* op = out + item * (size_t)Bsk * n;
    const u64 mt = c.m_tilde.q;
    const cmodp q_mods = as_cmod(c.q_mods), bsk_mods = as_cmod(c.bsk_mods);
    const cu64x2p q_mt_inv_punc = as_c128(c.q_mt_inv_punc), prod_q_mod_bsk = as_c128(c.prod_q_mod_bsk), inv_mt_mod_bsk = as_c128(c.inv_mt_mod_bsk);
    const cu64p q_to_mt = as_c64(c.q_to_mt), q_to_bsk = as_c64(c.q_to_bsk);
    for (unsigned x = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; x < n; x += chunks * blockDim.x) {
        u64 y[MAXL];
#pragma unroll
        for (int i = 0; i < MAXL; ++i) {
            if (i < (int)L) {
                // multiply_scalar_p by m_tilde, then the first step of fast_convert_array: one Shoup multiply by m_tilde * inv_punc
                const ulonglong2 f = ld_pair(q_mt_inv_punc, i);
                y[i] = shoup_mul(ip[(size_t)i * n + x], f.x, f.y, q_mods[i].q);
            } else y[i] = 0;
        }
        // q -> {m_tilde}
        u64 lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < MAXL; ++i) if (i < (int)L) mac128(lo, hi, y[i], q_to_mt[i]);
        const u64 in_mt = barrett128(lo, hi, mt, c.m_tilde.ratio_lo, c.m_tilde.ratio_hi);
        const u64 r_mt = shoup_mul(in_mt, c.neg_inv_prod_q_mod_mt.x, c.neg_inv_prod_q_mod_mt.y, mt);
        const u64 mt_half = mt >> 1;
        for (unsigned b = 0; b < Bsk; ++b) {
            const DevModulus mb = ld_mod(bsk_mods, b);
            lo = 0; hi = 0;
            const cu64p row = q_to_bsk + (size_t)b * L;
#pragma unroll
            for (int i = 0; i < MAXL; ++i) if (i < (int)L) mac128(lo, hi, y[i], row[i]);
            const u64 in_b = barrett128(lo, hi, mb.q, mb.ratio_lo, mb.ratio_hi);
            u64 temp = r_mt;
            if (temp >= mt_half) temp += mb.q - mt;
            const ulonglong2 pq = ld_pair(prod_q_mod_bsk, b), im = ld_pair(inv_mt_mod_bsk, b);
            const u64 mad = add_mod(shoup_mul(temp, pq.x, pq.y, mb.q), barrett64(in_b, mb.q, mb.ratio_hi), mb.q);
            op[(size_t)b * n + x] = shoup_mul(mad, im.x, im.y, mb.q);
        }
    }
}

// BEHZ steps (6)-(8): in_q [items][L][N], in_bsk [items][Bsk][N] (coefficient form) -> out [items][L][N]
template <int MAXB>
__global__ __launch_bounds__(256) void behz_floor_kernel(unsigned chunks, BehzDev c, const u64* in_q, const u64* in_bsk, u64* out) {
    const unsigned n = c.n, L = c.L, Bsk = c.Bsk, Bn = c.Bn;
    const size_t item = blockIdx.x / chunks;
    const u64* qp = in_q + item * (size_t)L * n;
    const u64* bp = in_bsk + item * (size_t)Bsk * n;
    u64* op = out + item * (size_t)L * n;
    const cmodp q_mods = as_cmod(c.q_mods), bsk_mods = as_cmod(c.bsk_mods);
    const cu64x2p q_t_inv_punc = as_c128(c.q_t_inv_punc), inv_prod_q_mod_bsk = as_c128(c.inv_prod_q_mod_bsk), t_inv_prod_q_mod_bsk = as_c128(c.t_inv_prod_q_mod_bsk), B_inv_punc = as_c128(c.B_inv_punc),
                  prod_B_mod_q = as_c128(c.prod_B_mod_q), neg_prod_B_mod_q = as_c128(c.neg_prod_B_mod_q);
    const cu64p q_to_bsk = as_c64(c.q_to_bsk), B_to_q = as_c64(c.B_to_q), B_to_msk = as_c64(c.B_to_msk);
    for (unsigned x = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; x < n; x += chunks * blockDim.x) {
        u64 y[MAXB];   // first the scaled q residues (L <= MAXB), later the scaled B residues
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            if (i < (int)L) {
                const ulonglong2 f = ld_pair(q_t_inv_punc, i);               // step (6) times t, folded into the conversion's scaling
                y[i] = shoup_mul(qp[(size_t)i * n + x], f.x, f.y, q_mods[i].q);
            } else y[i] = 0;
        }
        // step (7) fast_floor into Bsk
        u64 r[MAXB];
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b < (int)Bsk) {
                const DevModulus mb = ld_mod(bsk_mods, b);
                u64 lo = 0, hi = 0;
                const cu64p row = q_to_bsk + (size_t)b * L;
#pragma unroll
                for (int i = 0; i < MAXB; ++i) if (i < (int)L) mac128(lo, hi, y[i], row[i]);
                const u64 f = barrett128(lo, hi, mb.q, mb.ratio_lo, mb.ratio_hi);
                // (x t - f) q^-1 = x (t q^-1) - f q^-1 mod p_b
                const ulonglong2 iq = ld_pair(inv_prod_q_mod_bsk, b), tq = ld_pair(t_inv_prod_q_mod_bsk, b);
                r[b] = sub_mod(shoup_mul(bp[(size_t)b * n + x], tq.x, tq.y, mb.q), shoup_mul(f, iq.x, iq.y, mb.q), mb.q);
            } else r[b] = 0;
        }
        // step (8) fast_b_conv_sk: B -> q and B -> {m_sk}
        const DevModulus msk = ld_mod(bsk_mods, Bn);
        u64 r_sk = 0;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
            if (b == (int)Bn) r_sk = r[b];
            if (b < (int)Bn) y[b] = conv_scale(r[b], ld_pair(B_inv_punc, b), ld_mod(bsk_mods, b));
        }
        u64 lo = 0, hi = 0;
#pragma unroll
        for (int b = 0; b < MAXB; ++b) if (b < (int)Bn) mac128(lo, hi, y[b], B_to_msk[b]);
        const u64 h = barrett128(lo, hi, msk.q, msk.ratio_lo, msk.ratio_hi);
        const u64 alpha_sk = shoup_mul(h + (msk.q - r_sk), c.inv_prod_B_mod_msk.x, c.inv_prod_B_mod_msk.y, msk.q);
        const bool neg = alpha_sk > (msk.q >> 1);
        const u64 alpha_use = neg ? neg_mod(alpha_sk, msk.q) : alpha_sk;
        for (unsigned i = 0; i < L; ++i) {
            const DevModulus md = ld_mod(q_mods, i);
            lo = 0; hi = 0;
            const cu64p row = B_to_q + (size_t)i * Bn;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) if (b < (int)Bn) mac128(lo, hi, y[b], row[b]);
            const u64 g = barrett128(lo, hi, md.q, md.ratio_lo, md.ratio_hi);
            const ulonglong2 f = neg ? ld_pair(prod_B_mod_q, i) : ld_pair(neg_prod_B_mod_q, i);
            op[(size_t)i * n + x] = add_mod(shoup_mul(alpha_use, f.x, f.y, md.q), g, md.q);      // g is canonical (Barrett-128 output)
        }
    }
}

}  // namespace troyn
