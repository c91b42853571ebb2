// bgv_kernels.hpp -- the BGV-only element-wise steps.  BGV shares the NTT, dyadic, key-switch inner
// product and Galois kernels with BFV/CKKS; what differs is how a ciphertext is divided by its last prime (the correction
// is computed mod t, so that the plaintext is only multiplied by q_last^-1 mod t) and how a phase becomes a plaintext.
#pragma once
#include "poly_kernels.hpp"

namespace troyn {

// Step 1 of RNSTool::mod_t_and_divide_q_last_ntt (utils/rns_tool.cu:1602-1640) and of the key switch's BGV tail
// (kernel_ski_util5_merged_step1, evaluator_keyswitching_core.cu:436-476), which are the same computation:
//   k = -(c_last mod t) * q_last^-1 mod t;  delta_j = (k mod q_j) * q_last + (c_last mod q_j)  mod q_j
//   last_intt [items][..] coefficient form of the dropped limb (item stride last_stride);  delta [items][Lout][N]
__global__ __launch_bounds__(POLY_BLOCK) void bgv_delta_kernel(unsigned chunks, const DevModulus* mods, unsigned Lout, unsigned n, DevModulus t, u64 inv_last_mod_t,
                                                               u64 q_last, const u64* last_intt, size_t last_stride, u64* delta) {
    const unsigned j = blk_row(chunks) % Lout;
    const size_t item = blk_row(chunks) / Lout;
    const DevModulus qj = mods[j];
    const u64* src = last_intt + item * last_stride;
    u64* dst = delta + (item * Lout + j) * (size_t)n;
    for (unsigned x = blk_col(chunks); x < n; x += chunks * blockDim.x) {
        const u64 c = src[x];
        u64 k = neg_mod(barrett64(c, t.q, t.ratio_hi), t.q);
        if (inv_last_mod_t != 1) k = mul_mod(k, inv_last_mod_t, t);
        const u64 d = mul_mod(barrett64(k, qj.q, qj.ratio_hi), q_last, qj);
        dst[x] = add_mod(d, barrett64(c, qj.q, qj.ratio_hi), qj.q);
    }
}

// Step 2 (:1642-1670 / kernel_ski_util5_merged_step2 :506-538): dest_j (op)= (src_j - NTT(delta_j)) * q_last^-1 mod q_j.
//   src  [batch][P][src_rows][N] (rows 0..Lout-1 used), delta [batch][P][Lout][N] NTT form, dest [batch][P][Lout][N]
//   assign_method < 0: plain overwrite (mod switch); otherwise SwitchKeyDestinationAssignMethod with P = 2
__global__ __launch_bounds__(POLY_BLOCK) void bgv_finish_kernel(unsigned chunks, const DevModulus* mods, unsigned Lout, unsigned src_rows, unsigned n, const u64* src,
                                                                const u64* delta, const ulonglong2* inv_last, int assign_method, u64* dest, const u64* addend,
                                                                size_t addend_bstride) {
    const unsigned j = blk_row(chunks) % Lout;
    const size_t kb = blk_row(chunks) / Lout;          // item * P + poly
    const unsigned k = (unsigned)(kb & 1);
    const DevModulus md = mods[j];
    const ulonglong2 f = inv_last[j];
    const bool add_inplace = assign_method == 0 || (k == 0 && assign_method == 2);
    const u64* sp = src + (kb * src_rows + j) * (size_t)n;
    const u64* dl = delta + (kb * Lout + j) * (size_t)n;
    u64* dp = dest + (kb * Lout + j) * (size_t)n;
    for (unsigned x = blk_col(chunks); x < n; x += chunks * blockDim.x) {
        u64 d = shoup_mul(sub_mod(sp[x], dl[x], md.q), f.x, f.y, md.q);
        if (add_inplace) d = add_mod(dp[x], d, md.q);
        if (addend) d = add_mod(d, addend[(kb >> 1) * addend_bstride + ((size_t)k * Lout + j) * n + x], md.q);
        dp[x] = d;
    }
}

// RNSTool::decrypt_mod_t = BaseConverter::exact_convey_array q -> {t} (utils/rns_base.cu:467-490, :531-559), then the
// correction-factor inverse (scaling_variant::decentralize, utils/scaling_variant.cu:415-431).
//   y_i = x_i * (q/q_i)^-1 mod q_i;  v = round(sum_i y_i / q_i) in double precision, summed in limb order as the reference
//   does (the result depends on it);  out = (sum_i y_i * (q/q_i mod t) - v * (q mod t)) * fix mod t
struct BgvDecryptArgs {
    const DevModulus* mods;
    const ulonglong2* inv_punctured;     // [L] Shoup pairs mod q_i
    const u64* punctured_mod_t;          // [L]
    DevModulus t;
    u64 q_mod_t, fix;
    unsigned L, n;
};

__global__ __launch_bounds__(256) void bgv_decrypt_mod_t_kernel(BgvDecryptArgs a, const u64* phase, u64* dest) {
    const unsigned x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= a.n) return;
    const u64* ph = phase + (size_t)blockIdx.y * a.L * a.n;
    double v = 0.0;
    u64 acc_lo = 0, acc_hi = 0;
    for (unsigned i = 0; i < a.L; i++) {
        const DevModulus m = a.mods[i];
        const ulonglong2 op = a.inv_punctured[i];
        const u64 raw = ph[(size_t)i * a.n + x];
        const u64 y = (op.x == 1) ? barrett64(raw, m.q, m.ratio_hi) : shoup_mul(raw, op.x, op.y, m.q);
        v += (double)y / (double)m.q;
        // 128-bit accumulation of y * (q/q_i mod t): L * 2^61 * 2^61 < 2^128
        const u64 w = a.punctured_mod_t[i];
        const u64 lo = y * w, hi = mul_hi(y, w);
        acc_lo += lo;
        acc_hi += hi + (acc_lo < lo);
    }
    const u64 rounded = (u64)__builtin_round(v);
    const u64 sum = barrett128(acc_lo, acc_hi, a.t.q, a.t.ratio_lo, a.t.ratio_hi);
    u64 r = sub_mod(sum, mul_mod(rounded, a.q_mod_t, a.t), a.t.q);
    if (a.fix != 1) r = mul_mod(r, a.fix, a.t);
    dest[(size_t)blockIdx.y * a.n + x] = r;
}

// utils::multiply_scalar mod t on plaintext coefficients (add_plain multiplies the plaintext by the ciphertext's correction
// factor, evaluator_translate_plain.cu:80-82)
__global__ __launch_bounds__(256) void scalar_mod_t_kernel(DevModulus t, u64 scalar, const u64* in, u64* out, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = mul_mod(barrett64(in[i], t.q, t.ratio_hi), scalar, t);
}

}  // namespace troyn
