// crypto_kernels.hpp -- the callers either side of the evaluator hot path: the context PRNG's
// samplers, BFV plaintext scaling (encrypt side) and BEHZ decrypt_scale_and_round (decrypt side).
//
// These are small launches (N or L*N coefficients); they exist so that key generation, encryption and decryption
// stay on the GPU next to the evaluator kernels and reproduce the reference's results word for word:
//   * samplers: utils/random_generator.cu (ternary :318-336, centered binomial :374-385,:421-440, uniform
//     :475-481); one AES-128-CTR block per thread, blocks numbered from the generator's counter.
//   * bfv_scale_up_kernel: fgk/translate_plain.cu:6-75 (multiply_translate_plain).
//   * bfv_decrypt_round_kernel: utils/rns_tool.cu:1189-1268 (decrypt_scale_and_round, fused device form), with the
//     per-coefficient scratch of the reference (temp / fast_convert_temp in global memory) kept in registers.
#pragma once
#include "aes128.hpp"
#include "poly_kernels.hpp"

namespace troyn {

// out [nmod][n]; one AES block = 16 coefficients (one byte each, byte % 3; 2 -> q-1).  One thread per COEFFICIENT: the 16 threads of a block each
// encrypt the same counter (the cipher is ~9 us of latency either way) and store one word per limb, lane-consecutive -- a thread per block wrote
// 16 nmod strided words behind its cipher: 31.5 -> 10 us for one N = 16384, 6-limb polynomial (round 5).
__global__ __launch_bounds__(256) void sample_ternary_kernel(AesRoundKeys key, u64 counter, const DevModulus* mods, unsigned nmod, unsigned n, u64* out) {
    const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const unsigned blk = j >> 4, k = j & 15u;
    u64 w[2];
    aes128_encrypt_counter(key, counter + blk, (counter + blk < counter) ? 1ull : 0ull, w[0], w[1]);
    const unsigned byte = (unsigned)((w[k >> 3] >> ((k & 7) * 8)) & 0xff);
    const unsigned v = byte % 3;
    for (unsigned i = 0; i < nmod; i++) out[(size_t)i * n + j] = (v == 2) ? mods[i].q - 1 : (u64)v;
}

__device__ __forceinline__ int cbd_from_u64(u64 v) {
    // utils/random_generator.cu:374-385: 21 bits minus 21 bits
    const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    const unsigned pos = lo & 0x1fffffu;                                     // bytes 0,1 and the low 5 bits of byte 2
    const unsigned neg = ((lo >> 24) & 0xffu) | ((hi & 0xffu) << 8) | (((hi >> 8) & 0x1fu) << 16);   // bytes 3,4, 5 bits of byte 5
    return __popc(pos) - __popc(neg);
}

// out [nmod][n]; thread = one AES block = 2 coefficients
__global__ __launch_bounds__(256) void sample_cbd_kernel(AesRoundKeys key, u64 counter, const DevModulus* mods, unsigned nmod, unsigned n, u64* out) {
    const unsigned blk = blockIdx.x * blockDim.x + threadIdx.x;
    if ((size_t)blk * 2 >= n) return;
    u64 w[2];
    aes128_encrypt_counter(key, counter + blk, (counter + blk < counter) ? 1ull : 0ull, w[0], w[1]);
    for (unsigned k = 0; k < 2 && blk * 2 + k < n; k++) {
        const int v = cbd_from_u64(w[k]);
        const unsigned j = blk * 2 + k;
        for (unsigned i = 0; i < nmod; i++) out[(size_t)i * n + j] = (v >= 0) ? (u64)v : mods[i].q - (u64)(-v);
    }
}

// out [nmod][n] = the AES-CTR word stream reduced per limb (fill_uint64s + modulo_inplace_p); thread = 2 words
__global__ __launch_bounds__(256) void sample_uniform_kernel(AesRoundKeys key, u64 counter, const DevModulus* mods, unsigned nmod, unsigned n, u64* out) {
    const size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)nmod * n;
    if (blk * 2 >= total) return;
    u64 w[2];
    aes128_encrypt_counter(key, counter + blk, (counter + blk < counter) ? 1ull : 0ull, w[0], w[1]);
    for (unsigned k = 0; k < 2 && blk * 2 + k < total; k++) {
        const size_t pos = blk * 2 + k;
        const DevModulus md = mods[pos / n];
        out[pos] = barrett64(w[k], md.q, md.ratio_hi);
    }
}

// batched forms for many independent ciphertexts in one launch (grid.y = item):
//   centered binomial: item i draws from the SAME generator at counter + i * counter_stride (what i sequential
//   encryptions would have consumed); uniform: item i has its own generator (key i), counter 0
constexpr int SAMPLE_MULTI_KEYS = 16;     // 16 x 176 B of round keys fit the kernel-argument block
struct AesRoundKeysMulti { AesRoundKeys k[SAMPLE_MULTI_KEYS]; };

__global__ __launch_bounds__(256) void sample_cbd_strided_kernel(AesRoundKeys key, u64 counter, u64 counter_stride, const DevModulus* mods, unsigned nmod,
                                                                unsigned n, u64* out) {
    const unsigned blk = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned item = blockIdx.y;
    if ((size_t)blk * 2 >= n) return;
    u64 w[2];
    const u64 ctr = counter + (u64)item * counter_stride + blk;
    aes128_encrypt_counter(key, ctr, 0ull, w[0], w[1]);
    u64* op = out + (size_t)item * nmod * n;
    for (unsigned k = 0; k < 2 && blk * 2 + k < n; k++) {
        const int v = cbd_from_u64(w[k]);
        const unsigned j = blk * 2 + k;
        for (unsigned i = 0; i < nmod; i++) op[(size_t)i * n + j] = (v >= 0) ? (u64)v : mods[i].q - (u64)(-v);
    }
}

__global__ __launch_bounds__(256) void sample_uniform_multi_kernel(AesRoundKeysMulti keys, const DevModulus* mods, unsigned nmod, unsigned n, u64* out) {
    const size_t blk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned item = blockIdx.y;
    const size_t total = (size_t)nmod * n;
    if (blk * 2 >= total) return;
    u64 w[2];
    aes128_encrypt_counter(keys.k[item], (u64)blk, 0ull, w[0], w[1]);
    u64* op = out + (size_t)item * total;
    for (unsigned k = 0; k < 2 && blk * 2 + k < total; k++) {
        const size_t pos = blk * 2 + k;
        const DevModulus md = mods[pos / n];
        op[pos] = barrett64(w[k], md.q, md.ratio_hi);
    }
}

struct ScaleUpArgs {
    unsigned L, n, plain_coeff_count, subtract;
    const DevModulus* mods;
    const ulonglong2* delta;       // [L] Shoup pair of floor(q/t) mod q_j   (coeff_div_plain_modulus)
    DevModulus t;                  // plain modulus with its Barrett ratio
    u64 q_mod_t, threshold;        // q mod t, (t+1)/2
    const u64* plain; long long plain_bstride;
    const u64* from;  long long from_bstride;     // nullable
    u64* dest;        long long dest_bstride;
};

// floor(((hi:lo)) / t) for a numerator below t*t (quotient < 2^64): Barrett-128 quotient + one correction
__device__ __forceinline__ u64 div128_small_quotient(u64 in0, u64 in1, const DevModulus& t) {
    u64 carry = mul_hi(in0, t.ratio_lo);
    u64 t2lo = in0 * t.ratio_hi, t2hi = mul_hi(in0, t.ratio_hi);
    u64 tmp1 = t2lo + carry;
    u64 tmp3 = t2hi + (tmp1 < t2lo ? 1ull : 0ull);
    t2lo = in1 * t.ratio_lo; t2hi = mul_hi(in1, t.ratio_lo);
    u64 tmp1b = tmp1 + t2lo;
    carry = t2hi + (tmp1b < tmp1 ? 1ull : 0ull);
    u64 quot = in1 * t.ratio_hi + tmp3 + carry;
    const u64 r = in0 - quot * t.q;
    return r >= t.q ? quot + 1 : quot;
}

// one polynomial [L][n] per item: dest = (from or 0) +/- round(q/t * m); grid = items * L rows
__global__ __launch_bounds__(POLY_BLOCK) void bfv_scale_up_kernel(unsigned chunks, ScaleUpArgs a) {
    const unsigned j = blk_row(chunks) % a.L;
    const size_t item = blk_row(chunks) / a.L;
    const DevModulus md = a.mods[j];
    const ulonglong2 delta = a.delta[j];
    const u64* pl = a.plain + item * a.plain_bstride;
    const u64* fr = a.from ? a.from + item * a.from_bstride + (size_t)j * a.n : nullptr;
    u64* de = a.dest + item * a.dest_bstride + (size_t)j * a.n;
    for (unsigned i = blk_col(chunks); i < a.n; i += chunks * blockDim.x) {
        u64 v;
        if (i < a.plain_coeff_count) {
            const u64 m = pl[i];
            u64 lo = m * a.q_mod_t, hi = mul_hi(m, a.q_mod_t);
            const u64 lo2 = lo + a.threshold;
            hi += (lo2 < lo) ? 1ull : 0ull;
            const u64 fix = div128_small_quotient(lo2, hi, a.t);
            const u64 scaled = add_mod(shoup_mul(m, delta.x, delta.y, md.q), barrett64(fix, md.q, md.ratio_hi), md.q);
            if (fr) v = a.subtract ? sub_mod(fr[i], scaled, md.q) : add_mod(fr[i], scaled, md.q);
            else v = a.subtract ? neg_mod(scaled, md.q) : scaled;
        } else {
            v = fr ? fr[i] : 0;
        }
        de[i] = v;
    }
}

struct DecryptArgs {
    unsigned L, n;
    const DevModulus* mods;                 // base q
    DevModulus t, gamma;
    const ulonglong2* prod_t_gamma_mod_q;   // [L]
    const ulonglong2* q_inv_punc;           // [L]
    const u64* q_to_t;                      // [L] (q/q_i) mod t
    const u64* q_to_gamma;                  // [L] (q/q_i) mod gamma
    ulonglong2 neg_inv_q_mod_t, neg_inv_q_mod_gamma, inv_gamma_mod_t;
};

// phase [items][L][n] -> dest [items][n]; one thread per coefficient
__global__ __launch_bounds__(256) void bfv_decrypt_round_kernel(unsigned chunks, DecryptArgs c, const u64* phase, u64* dest) {
    const size_t item = blockIdx.x / chunks;
    const u64* ph = phase + item * (size_t)c.L * c.n;
    u64* de = dest + item * (size_t)c.n;
    for (unsigned x = (blockIdx.x % chunks) * blockDim.x + threadIdx.x; x < c.n; x += chunks * blockDim.x) {
        u64 lo_t = 0, hi_t = 0, lo_g = 0, hi_g = 0;
#pragma unroll 4
        for (unsigned i = 0; i < c.L; ++i) {
            const DevModulus md = c.mods[i];
            const ulonglong2 ptg = c.prod_t_gamma_mod_q[i];
            u64 y = shoup_mul(ph[(size_t)i * c.n + x], ptg.x, ptg.y, md.q);          // |gamma*t|_qi * ct(s)
            const ulonglong2 ip = c.q_inv_punc[i];
            y = (ip.x == 1) ? barrett64(y, md.q, md.ratio_hi) : shoup_mul(y, ip.x, ip.y, md.q);
            mac128(lo_t, hi_t, y, c.q_to_t[i]);
            mac128(lo_g, hi_g, y, c.q_to_gamma[i]);
        }
        u64 vt = barrett128(lo_t, hi_t, c.t.q, c.t.ratio_lo, c.t.ratio_hi);
        u64 vg = barrett128(lo_g, hi_g, c.gamma.q, c.gamma.ratio_lo, c.gamma.ratio_hi);
        vt = shoup_mul(vt, c.neg_inv_q_mod_t.x, c.neg_inv_q_mod_t.y, c.t.q);
        vg = shoup_mul(vg, c.neg_inv_q_mod_gamma.x, c.neg_inv_q_mod_gamma.y, c.gamma.q);
        const u64 gamma_div_2 = c.gamma.q >> 1;
        u64 d;
        if (vg > gamma_div_2) d = add_mod(vt, barrett64(c.gamma.q - vg, c.t.q, c.t.ratio_hi), c.t.q);
        else d = sub_mod(vt, barrett64(vg, c.t.q, c.t.ratio_hi), c.t.q);
        de[x] = d ? shoup_mul(d, c.inv_gamma_mod_t.x, c.inv_gamma_mod_t.y, c.t.q) : 0;
    }
}

// ---- ciphertext x plaintext -------------------------------------------------------------

// scaling_variant::centralize, fast-plain-lift case (utils/scaling_variant.cu:226-275): plain [items][count] mod t ->
// dest [items][L][n]: m >= (t+1)/2 ? m + (q_l - t) : m, zero beyond plain_coeff_count
__global__ __launch_bounds__(POLY_BLOCK) void plain_centralize_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned n, u64 t,
                                                                      const u64* plain, unsigned plain_coeff_count, long long plain_bstride, u64* dest) {
    const unsigned l = blk_row(chunks) % L;
    const size_t item = blk_row(chunks) / L;
    // fast plain lift (t < q_l): m + (q_l - t).  Otherwise (scaling_variant.cu:344-349, multiply_plain_normal_no_fast_plain_lift then
    // decompose_array): the multi-precision m + (Q - t) reduced mod q_l = (m mod q_l) + (q_l - (t mod q_l)) mod q_l, as Q = 0 mod q_l
    const DevModulus md = mods[l];
    const bool fast = t < md.q;
    const u64 t_red = fast ? t : barrett64(t, md.q, md.ratio_hi);
    const u64 inc = md.q - t_red, threshold = (t + 1) >> 1;
    const u64* pl = plain + item * plain_bstride;
    u64* de = dest + (item * L + l) * (size_t)n;
    for (unsigned i = blk_col(chunks); i < n; i += chunks * blockDim.x) {
        u64 v = 0;
        if (i < plain_coeff_count) {
            const u64 m = pl[i];
            if (fast) v = (m >= threshold) ? m + inc : m;
            else { const u64 r = barrett64(m, md.q, md.ratio_hi); v = (m >= threshold) ? add_mod(r, inc == md.q ? 0 : inc, md.q) : r; }
        }
        de[i] = v;
    }
}

// fgk/dyadic_convolute.cu:152-195 kernel_dyadic_broadcast_product_ps: out[item][p][l] = ct[item][p][l] (.) pt[item][l]
__global__ __launch_bounds__(POLY_BLOCK) void dyadic_broadcast_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned n,
                                                                      unsigned pcount, const u64* ct, const u64* pt, long long pt_bstride, u64* out) {
    const unsigned l = blk_row(chunks) % nmod;
    const size_t item = blk_row(chunks) / nmod;
    const DevModulus md = mods[mod_start + l];
    const size_t pc = (size_t)nmod * n;
    const u64* cp = ct + item * pcount * pc + (size_t)l * n;
    const u64* pp = pt + item * pt_bstride + (size_t)l * n;
    u64* op = out + item * pcount * pc + (size_t)l * n;
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        const u64x2 w = ld2(pp + i);
        for (unsigned p = 0; p < pcount; ++p) {
            const u64x2 c = ld2(cp + p * pc + i);
            st2(op + p * pc + i, mul_mod(c.a, w.a, md), mul_mod(c.b, w.b, md));
        }
    }
}

// Evaluator::multiply_plain_ntt_accumulate (evaluator_multiply_plain.cu:258-307; the reference kernel
// kernel_dyadic_broadcast_product_accumulate_bps makes every thread walk the whole batch).  Here the (ct, pt, dst)
// triples are grouped by destination on the host; one thread owns two coefficients of one (destination, poly, limb)
// and walks that destination's terms with a 128-bit lazy sum, so every destination word is written exactly once
// and every ct / pt word of the group is read exactly once.
//   tab: [ct_ptr x count][pt_ptr x count][dst_ptr x groups][start x (groups+1)]
__global__ __launch_bounds__(POLY_BLOCK) void plain_mac_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned n,
                                                               unsigned pcount, const u64* tab, unsigned count, unsigned groups, int set_zero) {
    const unsigned row = blk_row(chunks);
    const unsigned l = row % nmod;
    const unsigned p = (row / nmod) % pcount;
    const unsigned g = row / (nmod * pcount);
    const DevModulus md = mods[mod_start + l];
    const u64* const* cts = reinterpret_cast<const u64* const*>(tab);
    const u64* const* pts = reinterpret_cast<const u64* const*>(tab + count);
    u64* dst = reinterpret_cast<u64* const*>(tab + 2 * (size_t)count)[g];
    const u64* starts = tab + 2 * (size_t)count + groups;
    const unsigned k0 = (unsigned)starts[g], k1 = (unsigned)starts[g + 1];
    const size_t coff = ((size_t)p * nmod + l) * n, poff = (size_t)l * n;
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        u64 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, r0 = 0, r1 = 0;
        unsigned pending = 0;
        for (unsigned k = k0; k < k1; ++k) {
            const u64x2 c = ld2(cts[k] + coff + i), w = ld2(pts[k] + poff + i);
            mac128(lo0, hi0, c.a, w.a); mac128(lo1, hi1, c.b, w.b);
            if (++pending == 32) {      // 32 products of 61-bit residues stay below 2^128
                r0 = add_mod(r0, barrett128(lo0, hi0, md.q, md.ratio_lo, md.ratio_hi), md.q);
                r1 = add_mod(r1, barrett128(lo1, hi1, md.q, md.ratio_lo, md.ratio_hi), md.q);
                lo0 = hi0 = lo1 = hi1 = 0; pending = 0;
            }
        }
        r0 = add_mod(r0, barrett128(lo0, hi0, md.q, md.ratio_lo, md.ratio_hi), md.q);
        r1 = add_mod(r1, barrett128(lo1, hi1, md.q, md.ratio_lo, md.ratio_hi), md.q);
        if (!set_zero) { const u64x2 d = ld2(dst + coff + i); r0 = add_mod(r0, d.a, md.q); r1 = add_mod(r1, d.b, md.q); }
        st2(dst + coff + i, r0, r1);
    }
}

// Round 2: the same accumulation with both polynomials of a destination in one thread and the term loop unrolled by four.  The
// launch is bound by streaming the plaintexts (every weight is used once per destination): one thread now reads a weight word
// once for the PC polynomials (the first kernel read it once per polynomial, from different workgroups), bypasses the caches for
// it (the ciphertext words are re-read by every destination of the group and should stay cached), and has the loads of four
// terms in flight before their multiply-accumulates start.
// PACKED (layout experiment, tools/plain_mac_ab.py; TROYN_PLAIN_MAC=packed): the weights of a destination are ONE block [limb][chunk of 512 words][term][512]
// whose base is the plaintext pointer of the group's first term, so a workgroup's 4 KiB touches of T different plaintexts become one contiguous
// stream of T x 4 KiB.  Not the reference's object layout (a Plain2d is a vector of stand-alone plaintexts): measured, profiles/r04_plain_mac_ab.txt.
template <int PC, bool PACKED = false>
__global__ __launch_bounds__(POLY_BLOCK) void plain_mac2_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned n,
                                                                const u64* tab, unsigned count, unsigned groups, int set_zero) {
    const unsigned row = blk_row(chunks);
    const unsigned l = row % nmod;
    const unsigned g = row / nmod;
    const DevModulus md = mods[mod_start + l];
    const u64* const* cts = reinterpret_cast<const u64* const*>(tab);
    const u64* const* pts = reinterpret_cast<const u64* const*>(tab + count);
    u64* dst = reinterpret_cast<u64* const*>(tab + 2 * (size_t)count)[g];
    const u64* starts = tab + 2 * (size_t)count + groups;
    const unsigned k0 = (unsigned)starts[g], k1 = (unsigned)starts[g + 1];
    const size_t pstride = (size_t)nmod * n, loff = (size_t)l * n;
    auto ldw = [](const u64* p) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
        return u64x2{v.x, v.y};
    };
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        u64 lo[PC][2], hi[PC][2], r[PC][2];
#pragma unroll
        for (int p = 0; p < PC; ++p) { lo[p][0] = lo[p][1] = hi[p][0] = hi[p][1] = 0; r[p][0] = r[p][1] = 0; }
        auto fold = [&]() {      // 32 products of 61-bit residues stay below 2^128
#pragma unroll
            for (int p = 0; p < PC; ++p) {
                r[p][0] = add_mod(r[p][0], barrett128(lo[p][0], hi[p][0], md.q, md.ratio_lo, md.ratio_hi), md.q);
                r[p][1] = add_mod(r[p][1], barrett128(lo[p][1], hi[p][1], md.q, md.ratio_lo, md.ratio_hi), md.q);
                lo[p][0] = lo[p][1] = hi[p][0] = hi[p][1] = 0;
            }
        };
        unsigned k = k0, pending = 0;
        // packed weights: word (k, l, i) of the group at base + ((l * n/512 + i/512) * T + (k - k0)) * 512 + i % 512; expressed as a per-term pointer
        // so that the loads below stay `pw + loff + i`
        const u64* const wpk = PACKED ? pts[k0] + ((size_t)(l * (n >> 9) + (i >> 9)) * (k1 - k0)) * 512u + (i & 511u) - (loff + i) : nullptr;
        auto wptr = [&](unsigned kk) -> const u64* { return PACKED ? wpk + (size_t)(kk - k0) * 512u : pts[kk]; };
        // the operand pointers of the next four terms are fetched (scalar loads) while this iteration's words are in flight
        const u64 *pw[4], *pc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned kk = k + u < k1 ? k + u : k1 - 1; pw[u] = wptr(kk); pc[u] = cts[kk]; }
        for (; k + 4 <= k1; k += 4) {
            u64x2 w[4], c[4][PC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                w[u] = ldw(pw[u] + loff + i);
#pragma unroll
                for (int p = 0; p < PC; ++p) c[u][p] = ld2(pc[u] + p * pstride + loff + i);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const unsigned kk = k + 4 + u < k1 ? k + 4 + u : k1 - 1; pw[u] = wptr(kk); pc[u] = cts[kk]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int p = 0; p < PC; ++p) { mac128(lo[p][0], hi[p][0], c[u][p].a, w[u].a); mac128(lo[p][1], hi[p][1], c[u][p].b, w[u].b); }
            pending += 4;
            if (pending == 32) { fold(); pending = 0; }
        }
        for (; k < k1; ++k) {
            const u64x2 w = ldw(wptr(k) + loff + i);
#pragma unroll
            for (int p = 0; p < PC; ++p) {
                const u64x2 c = ld2(cts[k] + p * pstride + loff + i);
                mac128(lo[p][0], hi[p][0], c.a, w.a); mac128(lo[p][1], hi[p][1], c.b, w.b);
            }
        }
        fold();   // at most 28 + 3 pending products
#pragma unroll
        for (int p = 0; p < PC; ++p) {
            u64 r0 = r[p][0], r1 = r[p][1];
            if (!set_zero) { const u64x2 d = ld2(dst + p * pstride + loff + i); r0 = add_mod(r0, d.a, md.q); r1 = add_mod(r1, d.b, md.q); }
            st2(dst + p * pstride + loff + i, r0, r1);
        }
    }
}

// ND destinations per workgroup (round 4): ND consecutive destinations share their ciphertext operands term by term (a matmul row: every output
// column multiplies the same input ciphertexts by its own weights; the host checks it), so a ciphertext word is loaded once for both.  The launch
// streams every weight once from HBM whatever the tiling; what the grouping divides by ND is the ciphertext traffic from L2 to the CUs (two thirds of
// the vector-memory bytes of the one-destination kernel).
template <int PC, int ND>
__global__ __launch_bounds__(POLY_BLOCK) void plain_mac2_multi_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned n,
                                                                      const u64* tab, unsigned count, unsigned groups, int set_zero) {
    const unsigned row = blk_row(chunks);
    const unsigned l = row % nmod;
    const unsigned gp = row / nmod;                       // group of ND destinations
    const DevModulus md = mods[mod_start + l];
    const u64* const* cts = reinterpret_cast<const u64* const*>(tab);
    const u64* const* pts = reinterpret_cast<const u64* const*>(tab + count);
    u64* const* dsts = reinterpret_cast<u64* const*>(tab + 2 * (size_t)count) + (size_t)ND * gp;
    const u64* starts = tab + 2 * (size_t)count + groups + (size_t)ND * gp;
    const unsigned k0 = (unsigned)starts[0], terms = (unsigned)starts[1] - k0;   // every destination of the group has `terms` terms, laid out back to back
    const size_t pstride = (size_t)nmod * n, loff = (size_t)l * n;
    auto ldw = [](const u64* p) {
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        const v2 v = __builtin_nontemporal_load(reinterpret_cast<const v2*>(p));
        return u64x2{v.x, v.y};
    };
    for (unsigned i = blk_col(chunks) * 2; i < n; i += chunks * blockDim.x * 2) {
        u64 lo[ND][PC][2], hi[ND][PC][2], r[ND][PC][2];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int p = 0; p < PC; ++p) { lo[d][p][0] = lo[d][p][1] = hi[d][p][0] = hi[d][p][1] = 0; r[d][p][0] = r[d][p][1] = 0; }
        auto fold = [&]() {      // 32 products of 61-bit residues stay below 2^128
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int p = 0; p < PC; ++p) {
                    r[d][p][0] = add_mod(r[d][p][0], barrett128(lo[d][p][0], hi[d][p][0], md.q, md.ratio_lo, md.ratio_hi), md.q);
                    r[d][p][1] = add_mod(r[d][p][1], barrett128(lo[d][p][1], hi[d][p][1], md.q, md.ratio_lo, md.ratio_hi), md.q);
                    lo[d][p][0] = lo[d][p][1] = hi[d][p][0] = hi[d][p][1] = 0;
                }
        };
        constexpr int U = ND == 2 ? 4 : 2;    // terms in flight per step
        unsigned j = 0, pending = 0;
        for (; j + U <= terms; j += U) {
            u64x2 w[U][ND], c[U][PC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u64* pc = cts[k0 + j + u];
#pragma unroll
                for (int d = 0; d < ND; ++d) w[u][d] = ldw(pts[k0 + (unsigned)d * terms + j + u] + loff + i);
#pragma unroll
                for (int p = 0; p < PC; ++p) c[u][p] = ld2(pc + p * pstride + loff + i);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < PC; ++p)
#pragma unroll
                    for (int d = 0; d < ND; ++d) { mac128(lo[d][p][0], hi[d][p][0], c[u][p].a, w[u][d].a); mac128(lo[d][p][1], hi[d][p][1], c[u][p].b, w[u][d].b); }
            pending += U;
            if (pending == 32) { fold(); pending = 0; }
        }
        for (; j < terms; ++j) {
            u64x2 w[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) w[d] = ldw(pts[k0 + (unsigned)d * terms + j] + loff + i);
#pragma unroll
            for (int p = 0; p < PC; ++p) {
                const u64x2 c = ld2(cts[k0 + j] + p * pstride + loff + i);
#pragma unroll
                for (int d = 0; d < ND; ++d) { mac128(lo[d][p][0], hi[d][p][0], c.a, w[d].a); mac128(lo[d][p][1], hi[d][p][1], c.b, w[d].b); }
            }
        }
        fold();
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int p = 0; p < PC; ++p) {
                u64 a0 = r[d][p][0], a1 = r[d][p][1];
                u64* dp = dsts[d] + p * pstride + loff + i;
                if (!set_zero) { const u64x2 dv = ld2(dp); a0 = add_mod(a0, dv.a, md.q); a1 = add_mod(a1, dv.b, md.q); }
                st2(dp, a0, a1);
            }
    }
}

// ---- Galois automorphisms ---------------------------------------------------------------
// GaloisTool::apply_ps (utils/galois.cu:168-185): coefficient form, X -> X^g: out[i*g mod N] = +/- in[i]
// GaloisTool::apply_ntt_ps (:24-41 table + gather): NTT form, out[i] = in[table(i)],
//   table(i) = bitrev_logN(((g * bitrev_{logN+1}(i + N)) >> 1) mod N) -- computed on the fly, no table upload.
// data [rows][n], one limb-polynomial per row; row % nmod selects the modulus (negation in coefficient form).
__global__ __launch_bounds__(POLY_BLOCK) void galois_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned log_n,
                                                            unsigned g, int is_ntt_form, const u64* in, u64* out, u64 q_single) {
    const unsigned n = 1u << log_n, mask = n - 1;
    const size_t row = blk_row(chunks);
    const u64 q = q_single ? q_single : mods[mod_start + row % nmod].q;    // q_single: one polynomial modulo the plain modulus (GaloisTool::apply)
    const u64* ip = in + row * n;
    u64* op = out + row * n;
    for (unsigned i = blk_col(chunks); i < n; i += chunks * blockDim.x) {
        if (is_ntt_form) {
            const unsigned reversed = __brev(i + n) >> (31 - log_n);               // (log_n + 1)-bit reversal
            const unsigned index_raw = (unsigned)(((u64)g * reversed) >> 1) & mask;
            const unsigned src = __brev(index_raw) >> (32 - log_n);
            op[i] = ip[src];
        } else {
            const u64 index_raw = (u64)i * g;
            const u64 v = ip[i];
            op[index_raw & mask] = ((index_raw >> log_n) & 1) ? neg_mod(v, q) : v;
        }
    }
}

// ---- RLWE packing (Evaluator::pack_rlwe_ciphertexts, evaluator_lwes.cu:315-681) ------------------------
// negacyclic_shift_ps (utils/poly_small_mod.cu:927-944): out[(shift + k) mod N] = ((shift + k) & N) ? -in[k] : in[k], shift in [0, 2N).
// Written as a gather so stores stay coalesced: the source of output index k is j = (k - shift) mod N.
__device__ __forceinline__ u64 shifted_coeff(const u64* poly, unsigned k, unsigned shift, unsigned log_n, u64 q) {
    const unsigned mask = (1u << log_n) - 1;
    const unsigned j = (k - shift) & mask;
    const u64 v = poly[j];
    return (((shift + j) >> log_n) & 1) ? neg_mod(v, q) : v;
}

__global__ __launch_bounds__(POLY_BLOCK) void negacyclic_shift_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned log_n,
                                                                      unsigned shift, const u64* in, u64* out) {
    const unsigned n = 1u << log_n;
    const size_t row = blk_row(chunks);
    const u64 q = mods[mod_start + row % nmod].q;
    for (unsigned k = blk_col(chunks); k < n; k += chunks * blockDim.x) out[row * n + k] = shifted_coeff(in + row * n, k, shift, log_n, q);
}

// ntt_multiply_inv_degree (utils/ntt.cu:93-107): x * N^-1 * scalar mod q (lazy Shoup by N^-1, then a Barrett product: canonical)
__global__ __launch_bounds__(POLY_BLOCK) void multiply_inv_degree_kernel(unsigned chunks, const DevModulus* mods, unsigned mod_start, unsigned nmod, unsigned n,
                                                                         u64 scalar, const u64* in, u64* out) {
    const size_t row = blk_row(chunks);
    const DevModulus m = mods[mod_start + row % nmod];
    for (unsigned k = blk_col(chunks); k < n; k += chunks * blockDim.x)
        out[row * n + k] = mul_mod(shoup_lazy(in[row * n + k], m.inv_n_op, m.inv_n_quo, m.q), scalar, m);
}

// The first step of pack_rlwe_ciphertexts (:361-381 / :599-640): slot s of the working set takes source ciphertext src[s]
// (device pointer, polynomial count pcount, coefficient form) scaled by N^-1 * mul and shifted by `shift`; absent slots are zero.
__global__ __launch_bounds__(POLY_BLOCK) void pack_prepare_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned log_n, unsigned pcount,
                                                                  u64 mul, unsigned shift, const u64* const* src, u64* out) {
    const unsigned n = 1u << log_n, mask = n - 1;
    const size_t row = blk_row(chunks);                       // (slot, poly, limb)
    const size_t slot = row / ((size_t)pcount * L);
    const DevModulus m = mods[row % L];
    const u64* sp = src[slot];
    u64* op = out + row * n;
    if (!sp) {
        for (unsigned k = blk_col(chunks); k < n; k += chunks * blockDim.x) op[k] = 0;
        return;
    }
    sp += (row % ((size_t)pcount * L)) * n;
    for (unsigned k = blk_col(chunks); k < n; k += chunks * blockDim.x) {
        const unsigned j = (k - shift) & mask;
        const u64 v = mul_mod(shoup_lazy(sp[j], m.inv_n_op, m.inv_n_quo, m.q), mul, m);
        op[k] = (((shift + j) >> log_n) & 1) ? neg_mod(v, m.q) : v;
    }
}

// One layer of the packing tree (:441-477 / :654-697), everything except the key switch, for adjacent (even, odd) pairs:
//   temp = negacyclic_shift(odd, shift);  odd' = even - temp;  even' = even + temp;  result = even' + apply_galois(odd', g)
// apply_galois = coefficient permutation X -> X^g of (c0, c1) followed by a key switch of the permuted c1 whose output is
// added to (perm c0, 0).  This kernel writes out[pair] = (even'.c0 + perm(odd'.c0), even'.c1) and target[pair] = perm(odd'.c1);
// troyn_switch_key(target -> out, AddInplace) completes the layer.  The permutation is taken as a gather (g_inv = g^-1 mod 2N):
// output coefficient k of perm(x) is +/- x[i], i = k * g_inv mod 2N folded into [0, N).
__global__ __launch_bounds__(POLY_BLOCK) void pack_layer_kernel(unsigned chunks, const DevModulus* mods, unsigned L, unsigned log_n, unsigned shift,
                                                                unsigned g_inv, const u64* in, u64* out, u64* target) {
    const unsigned n = 1u << log_n, mask2 = 2 * n - 1;
    const size_t row = blk_row(chunks);                       // (pair, poly, limb)
    const size_t pair = row / (2 * (size_t)L);
    const unsigned c = (unsigned)((row / L) % 2), l = (unsigned)(row % L);
    const u64 q = mods[l].q;
    const u64* even = in + ((2 * pair) * 2 * L + (size_t)c * L + l) * n;
    const u64* odd = in + ((2 * pair + 1) * 2 * L + (size_t)c * L + l) * n;
    u64* op = out + row * n;
    u64* tp = target + (pair * L + l) * n;
    for (unsigned k = blk_col(chunks); k < n; k += chunks * blockDim.x) {
        const u64 sum = add_mod(even[k], shifted_coeff(odd, k, shift, log_n, q), q);
        const unsigned raw = (unsigned)(((u64)k * g_inv) & mask2);
        const unsigned i = raw & (n - 1);
        u64 diff = sub_mod(even[i], shifted_coeff(odd, i, shift, log_n, q), q);
        if (raw >> log_n) diff = neg_mod(diff, q);            // i*g = k + N (mod 2N): the coefficient arrives negated
        if (c == 0) op[k] = add_mod(sum, diff, q);
        else { op[k] = sum; tp[k] = diff; }
    }
}

// extract_lwe (evaluator_lwes.cu:15-22): c0[l] = rlwe_c0[l][term] for `count` (ciphertext, term) pairs given as device arrays
__global__ void extract_lwe_c0_kernel(unsigned L, unsigned n, const u64* const* c0_polys, const unsigned* terms, u64* out, unsigned count) {
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * L) return;
    const unsigned item = idx / L, l = idx % L;
    out[idx] = c0_polys[item][(size_t)l * n + terms[item]];
}

// Collect `count` device buffers of `words` words each into one contiguous block (the batched host API stages scattered
// operands with ONE launch instead of `count` copies).  src = device array of pointers; 16-byte accesses.
__global__ __launch_bounds__(256) void gather_kernel(const u64* const* src, size_t words, u64* out) {
    const u64* sp = src[blockIdx.y];
    u64* op = out + (size_t)blockIdx.y * words;
    const size_t pairs = words / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2*>(op)[i] = reinterpret_cast<const ulonglong2*>(sp)[i];
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) op[words - 1] = sp[words - 1];
}

// the same for up to 64 buffers with the pointers in the kernel arguments: no table upload, nothing to wait for on the host (the call
// combining layer of the C++ mirror stages the operands of <= 64 concurrent single-object calls with it)
struct GatherPtrs { const u64* p[64]; };
__global__ __launch_bounds__(256) void gather_small_kernel(GatherPtrs src, size_t words, u64* out) {
    const u64* sp = src.p[blockIdx.y];
    u64* op = out + (size_t)blockIdx.y * words;
    const size_t pairs = words / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2*>(op)[i] = reinterpret_cast<const ulonglong2*>(sp)[i];
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) op[words - 1] = sp[words - 1];
}

// the inverse: windows of one contiguous block back to `count` separate buffers (results of a combined batch go to the callers' own arrays)
struct ScatterPtrs { u64* p[64]; };
__global__ __launch_bounds__(256) void scatter_small_kernel(const u64* in, ScatterPtrs dst, size_t words) {
    u64* op = dst.p[blockIdx.y];
    const u64* sp = in + (size_t)blockIdx.y * words;
    const size_t pairs = words / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2*>(op)[i] = reinterpret_cast<const ulonglong2*>(sp)[i];
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) op[words - 1] = sp[words - 1];
}
__global__ __launch_bounds__(256) void scatter_kernel(const u64* in, u64* const* dst, size_t words) {
    u64* op = dst[blockIdx.y];
    const u64* sp = in + (size_t)blockIdx.y * words;
    const size_t pairs = words / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<ulonglong2*>(op)[i] = reinterpret_cast<const ulonglong2*>(sp)[i];
    if ((words & 1) && blockIdx.x == 0 && threadIdx.x == 0) op[words - 1] = sp[words - 1];
}

}  // namespace troyn
