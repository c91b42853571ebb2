// poly_kernels.hpp -- streaming (HBM-bound) RNS polynomial kernels: element-wise ops, dyadic
// tensor products, the key-switch inner product and the mod-switch / rescale element steps.
//
// All of them are pure streaming kernels: one thread owns TWO adjacent coefficients so every
// lane issues 16-byte (dwordx4) loads/stores, consecutive lanes touch consecutive addresses,
// and the modulus constants of a limb are wave-uniform.  Grids are sized by the data
// (>> 256 workgroups for any batch worth timing).
#pragma once
#include "dev_math.hpp"
#include "ntt_kernels.hpp"   // KeyPtrs

namespace troyn {

constexpr int POLY_BLOCK = 256;

// 1-D launch geometry shared by every streaming kernel: block = (row, chunk), `chunks` blocks per
// row of N coefficients (grid.y is limited to 65535, a batched launch can exceed that).
__device__ __forceinline__ unsigned blk_row(unsigned chunks) { return blockIdx.x / chunks; }
__device__ __forceinline__ unsigned blk_col(unsigned chunks) { return (blockIdx.x % chunks) * blockDim.x + threadIdx.x; }

struct u64x2 { u64 a, b; };
__device__ __forceinline__ u64x2 ld2(const u64* p) { ulonglong2 v = *reinterpret_cast<const ulonglong2*>(p); return u64x2{v.x, v.y}; }
__device__ __forceinline__ void st2(u64* p, u64 a, u64 b) { *reinterpret_cast<ulonglong2*>(p) = make_ulonglong2(a, b); }

enum { EW_ADD = 0, EW_SUB = 1, EW_NEG = 2, EW_MULS = 3, EW_MUL = 4, EW_MOD = 5, EW_MULOP = 6 };

// data [count][nmod][N]; grid.x covers N/2 pairs, grid.y = count*nmod limb-polynomials
// (utils/poly_small_mod.cu kernel_add_ps / kernel_sub_ps / kernel_negate_ps /
//  kernel_multiply_scalar_ps / kernel_dyadic_product_ps / kernel_modulo_ps / kernel_multiply_uint64operand_ps)
// EW_MOD: Modulus::reduce of ANY 64-bit word; EW_MULOP: b holds one (operand, quotient) pair per limb of the slice
// (MultiplyUint64Operand, utils/uint_small_mod.h:92-139), the input may be any 64-bit word, the result is canonical.
template <int OP>
__global__ __launch_bounds__(POLY_BLOCK) void elementwise_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod,
                                                                 unsigned n, const u64* a, const u64* b, u64 scalar, u64* out) {
    const unsigned limb = blk_row(chunks);
    const DevModulus md = mods[mod_start + limb % nmod];
    const size_t base = (size_t)limb * n;
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        const u64x2 va = ld2(a + base + i);
        u64 r0, r1;
        if constexpr (OP == EW_ADD) { const u64x2 vb = ld2(b + base + i); r0 = add_mod(va.a, vb.a, md.q); r1 = add_mod(va.b, vb.b, md.q); }
        else if constexpr (OP == EW_SUB) { const u64x2 vb = ld2(b + base + i); r0 = sub_mod(va.a, vb.a, md.q); r1 = sub_mod(va.b, vb.b, md.q); }
        else if constexpr (OP == EW_NEG) { r0 = neg_mod(va.a, md.q); r1 = neg_mod(va.b, md.q); }
        else if constexpr (OP == EW_MULS) { r0 = mul_mod(va.a, scalar, md); r1 = mul_mod(va.b, scalar, md); }
        else if constexpr (OP == EW_MOD) { r0 = barrett64(va.a, md.q, md.ratio_hi); r1 = barrett64(va.b, md.q, md.ratio_hi); }
        else if constexpr (OP == EW_MULOP) {
            const u64x2 w = ld2(b + 2 * (limb % nmod));     // wave-uniform (operand, quotient)
            r0 = shoup_mul(va.a, w.a, w.b, md.q); r1 = shoup_mul(va.b, w.a, w.b, md.q);
        }
        else { const u64x2 vb = ld2(b + base + i); r0 = mul_mod(va.a, vb.a, md); r1 = mul_mod(va.b, vb.b, md); }
        st2(out + base + i, r0, r1);
    }
}

// fgk/dyadic_convolute.cu:6-41 kernel_dyadic_convolute, restructured: one thread computes ALL
// pa+pb-1 outputs of its two coefficients, so each input word is read from HBM exactly once
// (the reference re-reads a_i, b_j once per output polynomial).  PA, PB compile-time (2,2 / 3,2..).
template <int PA, int PB>
__global__ __launch_bounds__(POLY_BLOCK) void dyadic_convolute_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod,
                                                                      unsigned n, const u64* a, const u64* b, u64* out) {
    constexpr int PO = PA + PB - 1;
    const unsigned limb = blk_row(chunks) % nmod;
    const size_t item = blk_row(chunks) / nmod;
    const DevModulus md = mods[mod_start + limb];
    const size_t pc = (size_t)nmod * n;
    const u64* ap = a + item * PA * pc + (size_t)limb * n;
    const u64* bp = b + item * PB * pc + (size_t)limb * n;
    u64* op = out + item * PO * pc + (size_t)limb * n;
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        u64x2 va[PA], vb[PB];
#pragma unroll
        for (int p = 0; p < PA; ++p) va[p] = ld2(ap + p * pc + i);
#pragma unroll
        for (int p = 0; p < PB; ++p) vb[p] = ld2(bp + p * pc + i);
#pragma unroll
        for (int o = 0; o < PO; ++o) {
            u64 acc0 = 0, acc1 = 0;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int r = o - p;
                if (r >= 0 && r < PB) {
                    acc0 = add_mod(acc0, mul_mod(va[p].a, vb[r].a, md), md.q);
                    acc1 = add_mod(acc1, mul_mod(va[p].b, vb[r].b, md), md.q);
                }
            }
            st2(op + o * pc + i, acc0, acc1);
        }
    }
}

// generic (any pa, pb) version with the reference's loop structure
__global__ __launch_bounds__(POLY_BLOCK) void dyadic_convolute_generic_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod,
                                                                              unsigned n, const u64* a, unsigned pa, const u64* b, unsigned pb, u64* out) {
    const unsigned po = pa + pb - 1;
    const unsigned limb = blk_row(chunks) % nmod;
    const size_t item = blk_row(chunks) / nmod;
    const DevModulus md = mods[mod_start + limb];
    const size_t pc = (size_t)nmod * n;
    const u64* ap = a + item * pa * pc + (size_t)limb * n;
    const u64* bp = b + item * pb * pc + (size_t)limb * n;
    u64* op = out + item * po * pc + (size_t)limb * n;
    for (unsigned i = blk_col(chunks); i < n; i += chunks * blockDim.x) {
        for (unsigned o = 0; o < po; ++o) {
            const unsigned i_start = (o + 1 > pb) ? (o + 1 - pb) : 0;
            const unsigned i_end = (pa - 1 < o) ? pa - 1 : o;
            u64 acc = 0;
            for (unsigned p = i_start; p <= i_end; ++p)
                acc = add_mod(acc, mul_mod(ap[p * pc + i], bp[(o - p) * pc + i], md), md.q);
            op[o * pc + i] = acc;
        }
    }
}

// fgk/dyadic_convolute.cu:92-114 kernel_dyadic_square
__global__ __launch_bounds__(POLY_BLOCK) void dyadic_square_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod,
                                                                   unsigned n, const u64* a, u64* out) {
    const unsigned limb = blk_row(chunks) % nmod;
    const size_t item = blk_row(chunks) / nmod;
    const DevModulus md = mods[mod_start + limb];
    const size_t pc = (size_t)nmod * n;
    const u64* ap = a + item * 2 * pc + (size_t)limb * n;
    u64* op = out + item * 3 * pc + (size_t)limb * n;
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        const u64x2 c0 = ld2(ap + i), c1 = ld2(ap + pc + i);
        u64 x0 = mul_mod(c0.a, c1.a, md), x1 = mul_mod(c0.b, c1.b, md);
        st2(op + i, mul_mod(c0.a, c0.a, md), mul_mod(c0.b, c0.b, md));
        st2(op + pc + i, add_mod(x0, x0, md.q), add_mod(x1, x1, md.q));
        st2(op + 2 * pc + i, mul_mod(c1.a, c1.a, md), mul_mod(c1.b, c1.b, md));
    }
}

// ---- key switching ---------------------------------------------------------------------

// KeyPtrs / KS_MAX_KEYS: ntt_kernels.hpp

// fgk/switch_key.cu:83-136 kernel_accumulate_products.
//   temp_ntt [batch][L+1][L][N]   decomposed digits in NTT form (row i under q_key(i))
//   keys[j]  -> [2][K][N]
//   poly_prod [batch][2][L+1][N]
// grid.y = batch*(L+1); a thread owns two coefficients of ONE row i and produces both key
// components from a single pass over its L digits; the <digit, key> sum is accumulated in
// 128 bits and reduced once (a sum of L <= 64 products of 61-bit residues is < 2^128, and the
// reference's per-term Barrett reduction yields the same canonical value).
// diag != nullptr: the digit of row i under its own modulus (j == i < L) is the untouched NTT-form input limb
// (the reference's host path does the same, evaluator_keyswitching_core.cu:851-852), read from diag[item][j].
__global__ __launch_bounds__(POLY_BLOCK) void ks_accumulate_kernel(unsigned chunks, const DevModulus* mods, unsigned K, unsigned L, unsigned n,
                                                                   const u64* temp_ntt, KeyPtrs keys, u64* poly_prod,
                                                                   const u64* diag, size_t diag_bstride) {
    const unsigned i = blk_row(chunks) % (L + 1);
    const size_t item = blk_row(chunks) / (L + 1);
    const unsigned key_index = (i == L) ? K - 1 : i;
    const DevModulus md = mods[key_index];
    const u64* tp = temp_ntt + (item * (L + 1) + i) * (size_t)L * n;
    const size_t key_poly = (size_t)K * n;
    u64* pp = poly_prod + item * 2 * (size_t)(L + 1) * n + (size_t)i * n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        u64 lo00 = 0, hi00 = 0, lo01 = 0, hi01 = 0, lo10 = 0, hi10 = 0, lo11 = 0, hi11 = 0;
        for (unsigned j = 0; j < L; ++j) {
            const u64x2 d = (diag && j == i) ? ld2(diag + item * diag_bstride + (size_t)j * n + x) : ld2(tp + (size_t)j * n + x);
            const u64* kj = keys.p[j] + (size_t)key_index * n + x;
            const u64x2 k0 = ld2(kj), k1 = ld2(kj + key_poly);
            mac128(lo00, hi00, d.a, k0.a); mac128(lo01, hi01, d.b, k0.b);
            mac128(lo10, hi10, d.a, k1.a); mac128(lo11, hi11, d.b, k1.b);
        }
        st2(pp + x, barrett128(lo00, hi00, md.q, md.ratio_lo, md.ratio_hi), barrett128(lo01, hi01, md.q, md.ratio_lo, md.ratio_hi));
        st2(pp + (size_t)(L + 1) * n + x, barrett128(lo10, hi10, md.q, md.ratio_lo, md.ratio_hi),
            barrett128(lo11, hi11, md.q, md.ratio_lo, md.ratio_hi));
    }
}

// evaluator_keyswitching_core.cu:570-598 device_ski_util6_merged.
//   last_intt [batch][2][N]       INTT of the special-prime row of poly_prod
//   temp_last [batch][2][L][N]
// grid.y = batch*2*L : thread -> two coefficients of (item, k, j)
__global__ __launch_bounds__(POLY_BLOCK) void ks_util6_kernel(unsigned chunks, const DevModulus* mods, unsigned K, unsigned L, unsigned n,
                                                              const u64* last_intt, size_t last_stride, u64* temp_last) {
    const unsigned j = blk_row(chunks) % L;
    const size_t kb = blk_row(chunks) / L;  // item*2 + k
    const DevModulus qk = mods[K - 1];
    const DevModulus qi = mods[j];
    const u64 qk_half = qk.q >> 1;
    const u64 fix = qi.q - barrett64(qk_half, qi.q, qi.ratio_hi);
    const bool need_reduce = qk.q > qi.q;
    const u64* src = last_intt + kb * last_stride;   // last_stride = N (compact) or (L+1)*N (row L of a full INTT)
    u64* dst = temp_last + (kb * L + j) * (size_t)n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 v = ld2(src + x);
        u64 t0 = barrett64(v.a + qk_half, qk.q, qk.ratio_hi), t1 = barrett64(v.b + qk_half, qk.q, qk.ratio_hi);
        if (need_reduce) { t0 = barrett64(t0, qi.q, qi.ratio_hi); t1 = barrett64(t1, qi.q, qi.ratio_hi); }
        st2(dst + x, t0 + fix, t1 + fix);
    }
}

// evaluator_keyswitching_core.cu:625-658 device_ski_util7_merged.
//   prod      [batch][2][prod_rows][N] (rows 0..L-1 used; NTT form: poly_prod, else its INTT)
//   temp_last [batch][2][L][N]
//   dest      [batch][2][L][N]
//   inv_qk    Shoup pairs (operand, quotient) of q_special^-1 mod q_j, j < L (RNSTool::inv_q_last_mod_q of the key level)
__global__ __launch_bounds__(POLY_BLOCK) void ks_util7_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned prod_rows, unsigned n,
                                                              const u64* prod, const u64* temp_last, const ulonglong2* inv_qk,
                                                              int is_ckks, int assign_method, u64* dest,
                                                              const u64* addend, size_t addend_bstride,
                                                              const u64* last_intt = nullptr, size_t last_stride = 0, unsigned K = 0) {
    const unsigned j = blk_row(chunks) % L;
    const size_t kb = blk_row(chunks) / L;   // item*2 + k
    const unsigned k = kb & 1;
    const DevModulus md = mods[j];
    const ulonglong2 f = inv_qk[j];
    // coefficient form: ski_util6's rounding fix is formed here from the special-prime row (last_intt) instead of being read back
    // from temp_last -- the same word sequence as ks_util6_kernel
    const DevModulus qk = mods[last_intt ? K - 1 : j];
    const u64 qk_half = qk.q >> 1;
    const u64 fix = md.q - barrett64(qk_half, md.q, md.ratio_hi);
    const bool need_reduce = qk.q > md.q;
    const u64* ls = last_intt ? last_intt + kb * last_stride : nullptr;
    const u64 lift = is_ckks ? (md.q << 2) : (md.q << 1);
    const bool add_inplace = (assign_method == 0) || (k == 0 && assign_method == 2);
    const u64* pp = prod + (kb * prod_rows + j) * (size_t)n;
    const u64* tl = temp_last + (kb * L + j) * (size_t)n;
    u64* dp = dest + (kb * L + j) * (size_t)n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 p = ld2(pp + x);
        u64x2 t;
        if (ls) {
            const u64x2 v = ld2(ls + x);
            u64 t0 = barrett64(v.a + qk_half, qk.q, qk.ratio_hi), t1 = barrett64(v.b + qk_half, qk.q, qk.ratio_hi);
            if (need_reduce) { t0 = barrett64(t0, md.q, md.ratio_hi); t1 = barrett64(t1, md.q, md.ratio_hi); }
            t.a = t0 + fix; t.b = t1 + fix;
        } else t = ld2(tl + x);
        u64 d0 = shoup_mul(p.a + lift - t.a, f.x, f.y, md.q);
        u64 d1 = shoup_mul(p.b + lift - t.b, f.x, f.y, md.q);
        if (add_inplace) {
            const u64x2 o = ld2(dp + x);
            d0 = add_mod(o.a, d0, md.q); d1 = add_mod(o.b, d1, md.q);
        }
        if (addend) {   // relinearize_internal's trailing add_inplace_ps (evaluator_keyswitching.cu:143), fused
            const u64x2 o = ld2(addend + (kb >> 1) * addend_bstride + ((size_t)k * L + j) * n + x);
            d0 = add_mod(d0, o.a, md.q); d1 = add_mod(d1, o.b, md.q);
        }
        st2(dp + x, d0, d1);
    }
}

// ---- modulus switching -------------------------------------------------------------------

// utils/rns_tool.cu:374-402 device_divide_and_round_q_last (BFV, coefficient form).
//   in [items][L][N] -> out [items][L-1][N]; grid.y = items*(L-1)
//   inv_last: Shoup pairs of q_{L-1}^-1 mod q_i for this level
__global__ __launch_bounds__(POLY_BLOCK) void divide_round_q_last_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned n,
                                                                         const u64* in, const ulonglong2* inv_last, u64* out) {
    const unsigned i = blk_row(chunks) % (L - 1);
    const size_t item = blk_row(chunks) / (L - 1);
    const DevModulus md = mods[i];
    const DevModulus ql = mods[L - 1];
    const ulonglong2 f = inv_last[i];
    const u64 half = ql.q >> 1;
    const u64 half_mod = barrett64(half, md.q, md.ratio_hi);
    const u64* last = in + (item * L + (L - 1)) * (size_t)n;
    const u64* src = in + (item * L + i) * (size_t)n;
    u64* dst = out + (item * (L - 1) + i) * (size_t)n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 vl = ld2(last + x), vi = ld2(src + x);
        u64 t0 = barrett64(add_mod(vl.a, half, ql.q), md.q, md.ratio_hi);
        u64 t1 = barrett64(add_mod(vl.b, half, ql.q), md.q, md.ratio_hi);
        t0 = sub_mod(t0, half_mod, md.q); t1 = sub_mod(t1, half_mod, md.q);
        st2(dst + x, shoup_mul(sub_mod(vi.a, t0, md.q), f.x, f.y, md.q), shoup_mul(sub_mod(vi.b, t1, md.q), f.x, f.y, md.q));
    }
}

// utils/rns_tool.cu:523-550 device_divide_and_round_q_last_ntt_step1.
//   last_intt [items][N] (INTT of the last limb) -> temp [items][L-1][N]; grid.y = items*(L-1)
__global__ __launch_bounds__(POLY_BLOCK) void rescale_step1_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned n,
                                                                   const u64* last_intt, u64* temp) {
    const unsigned i = blk_row(chunks) % (L - 1);
    const size_t item = blk_row(chunks) / (L - 1);
    const DevModulus md = mods[i];
    const DevModulus ql = mods[L - 1];
    const u64 half = ql.q >> 1;
    const u64 half_mod = barrett64(half, md.q, md.ratio_hi);
    const bool need_reduce = md.q < ql.q;
    const u64* src = last_intt + item * (size_t)n;
    u64* dst = temp + (item * (L - 1) + i) * (size_t)n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 v = ld2(src + x);
        u64 t0 = add_mod(v.a, half, ql.q), t1 = add_mod(v.b, half, ql.q);
        if (need_reduce) { t0 = barrett64(t0, md.q, md.ratio_hi); t1 = barrett64(t1, md.q, md.ratio_hi); }
        st2(dst + x, sub_mod(t0, half_mod, md.q), sub_mod(t1, half_mod, md.q));
    }
}

// utils/rns_tool.cu:607-627 device_divide_and_round_q_last_ntt_step2.
//   in [items][L][N], temp [items][L-1][N] (NTT form) -> out [items][L-1][N]
__global__ __launch_bounds__(POLY_BLOCK) void rescale_step2_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned n,
                                                                   const u64* in, const u64* temp, const ulonglong2* inv_last, u64* out) {
    const unsigned i = blk_row(chunks) % (L - 1);
    const size_t item = blk_row(chunks) / (L - 1);
    const DevModulus md = mods[i];
    const ulonglong2 f = inv_last[i];
    const u64 lift = md.q << 2;
    const u64* src = in + (item * L + i) * (size_t)n;
    const u64* tp = temp + (item * (L - 1) + i) * (size_t)n;
    u64* dst = out + (item * (L - 1) + i) * (size_t)n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 v = ld2(src + x), t = ld2(tp + x);
        // add_uint64_mod(x, 4q) = x + 3q ; sub_uint64_mod(., t) never borrows ; Shoup multiply canonicalises
        u64 d0 = add_mod(v.a, lift, md.q), d1 = add_mod(v.b, lift, md.q);
        d0 = sub_mod(d0, t.a, md.q); d1 = sub_mod(d1, t.b, md.q);
        st2(dst + x, shoup_mul(d0, f.x, f.y, md.q), shoup_mul(d1, f.x, f.y, md.q));
    }
}

// strided limb copy: evaluator_modswitch.cu:147-171 kernel_mod_switch_drop_to, and the gathers
// the fused pipelines need ([items][L_in][N] -> [items][L_out][N], first L_out limbs, or one limb).
__global__ __launch_bounds__(POLY_BLOCK) void copy_limbs_kernel(unsigned chunks, unsigned n, const u64* in, size_t in_item_stride, unsigned in_first,
                                                                u64* out, size_t out_item_stride, unsigned nlimbs) {
    const unsigned l = blk_row(chunks) % nlimbs;
    const size_t item = blk_row(chunks) / nlimbs;
    const u64* src = in + item * in_item_stride + (size_t)(in_first + l) * n;
    u64* dst = out + item * out_item_stride + (size_t)l * n;
    for (unsigned x = blk_col(chunks) * 2; x < n; x += chunks * blockDim.x * 2) {
        const u64x2 v = ld2(src + x);
        st2(dst + x, v.a, v.b);
    }
}

}  // namespace troyn
