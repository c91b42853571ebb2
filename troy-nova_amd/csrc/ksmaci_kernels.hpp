// ksmaci_kernels.hpp -- fused key-switch inner product for output rows whose modulus is 2^50 or wider (integer arithmetic), in the
// mould of ksmac2_kernel (ksmac_kernels.hpp): tiles of 2^13 outputs, 256 threads x 32 coefficients, two workgroups per CU.
//
// Replaces kernel_set_accumulate + ntt + kernel_accumulate_products (reference fgk/switch_key.cu:6-54, :83-154, driven from
// evaluator_keyswitching_core.cu:904-919) for the rows the exact-FP64 policy cannot take (the reference treats every modulus alike; its
// bench tool defaults to {60,40,40,60}, test/bench/he_operations.cu:22-24):
//     out[k][c] = sum_j  NTT_{q_key(k)}( digit_j mod q_key(k) ) (.) key_j[c][k]          c = 0, 1
// The first-generation kernel (ks_mac_kernel<ArithU64>, ntt_kernels.hpp) needs 187 registers at 512 threads x 16 coefficients, i.e. ONE
// workgroup (2 waves per SIMD, all in the same phase) per CU, whole-limb tiles only (N = 16384: 1024 threads under the 128-register cap):
// 0.58 VALU-busy at N = 8192 (profiles/r04 mixed PMC pass).  Here:
//   * the same tile / thread / exchange structure as ksmac2 (one workgroup barrier pair + one wave-private exchange per digit; the
//     first Cooley-Tukey layer(s) of N = 16384 / 32768 applied while loading), two independent workgroups per CU;
//   * Harvey butterflies with the three-product Shoup quotient (ArithU64: lazy [0, 8q), q < 2^61), words cross the LDS as they are
//     (no re-centring as in the FP64 form);
//   * round-0 twiddles are workgroup-uniform (operand, quotient) pairs in SGPRs; round-1 / round-2 pairs come from per-round copies of
//     the table (tw_r1: 512 bytes per value of the index bits above bit 9; tw_r2: lane-interleaved, 2 KB of consecutive bytes per wave
//     load) in groups of four, one group ahead of the butterflies;
//   * keys are prepared once per call as (key, floor(key 2^64 / q)) pairs in the accumulators' register layout, ONLY for the rows this
//     kernel takes (ksmaci_prepare_keys_kernel): a <digit, key> term is a lazy Shoup product on a lazy sum, one conditional subtract;
//   * digits of limbs whose modulus is not above the row's are used as they are (no Barrett reduction per word);
//   * epilogues in the coalesced layout: plain (coefficient-form target), DG (NTT-form target: the diagonal digit is the target's own
//     limb), TEN (fused multiply -> relinearize -> rescale chain: Q = P qk^-1 + tensor term, keys prepared times qk^-1).
// Results are canonical residues, bit-identical to the reference (every step is exact modular arithmetic; lazy ranges: ArithU64).
#pragma once
#include "ksmac_kernels.hpp"

namespace troyn {

struct KsMacIArgs {
    const u64* digits; long long dig_bstride, dig_cstride;      // coefficient-form digits [item][j][N], canonical under q_j
    const u64* diag;   long long diag_bstride, diag_cstride;     // EPI 1: NTT-form target limbs [item][j][N]
    const u64* ten_a; const u64* ten_b; long long ten_bstride, ten_pstride;     // EPI 2: the two input ciphertexts [item][2][limbs][N]
    u64* out;          long long out_bstride, out_pstride, out_cstride;         // [item][2][L+1][N]: (item, component, row)
    const DevModulus* mods;
    const ulonglong2* tw;       // [K][N]  forward twiddles (operand, quotient), reference table order
    const ulonglong2* tw_r1;    // [K][N/1024][32]
    const ulonglong2* tw_r2;    // [K][N]  lane-interleaved (ksm_perm)
    const ulonglong2* keys;     // prepared (key, quotient) pairs [L][2][slots][N], ksm_perm order; slot = rank of the row in row_mask
    long long key_jstride, key_pstride;
    const ulonglong2* diag_keys;   // EPI 1 / 2: [slot][2][N] natural order: key_k under modulus k (EPI 2: times qk^-1) with its quotient
    unsigned L;                 // digits = data limbs; rows = L + 1
    unsigned table_start, table_count;    // row k uses modulus table_start + (k == L ? table_count - 1 : k)
    unsigned batch;
    unsigned grouped;           // 1: the workgroups of an item are dealt to one XCD (batch % 8 == 0)
    unsigned long long row_mask;   // the rows this launch covers (never 0)
};

typedef u64 ksmi_tw_mem __attribute__((ext_vector_type(2)));

// (key, floor(key 2^64 / q)) pairs of the rows in row_mask, permuted to the accumulators' layout.
//   keys[j] -> [2][K][N] u64 (KSwitchKeys layout)  ==>  out [j][2][slots][N] pairs
// scale != nullptr (fused chain): data rows (r < scale_rows) are multiplied by scale[r] = qk^-1 mod q_r first.
// diag_out: [slot][2][N] pairs in natural order, the block (key k, modulus k) of every data row in the mask.
static __global__ __launch_bounds__(256) void ksmaci_prepare_keys_kernel(KeyPtrs keys, unsigned L, unsigned K, unsigned n, unsigned long long row_mask,
                                                                         ulonglong2* out, const ulonglong2* scale, const DevModulus* mods, unsigned scale_rows,
                                                                         ulonglong2* diag_out) {
    const unsigned slots = (unsigned)__builtin_popcountll(row_mask);
    const size_t per_key = (size_t)2 * slots * n, total = (size_t)L * per_key;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned j = (unsigned)(p / per_key);
        const size_t w = p % per_key;
        const unsigned c = (unsigned)(w / ((size_t)slots * n));
        const unsigned sl = (unsigned)((w / n) % slots);
        const unsigned i = (unsigned)(w % n);
        const unsigned k = nth_set_bit(row_mask, sl);
        const unsigned mrow = (k == L) ? K - 1 : k;
        const DevModulus md = mods[mrow];
        u64 v = keys.p[j][((size_t)c * K + mrow) * n + i];
        if (scale && mrow < scale_rows && k < L) { const ulonglong2 f = scale[mrow]; v = shoup_mul(v, f.x, f.y, md.q); }
        u64 e = v * md.ratio_hi + mul_hi(v, md.ratio_lo);            // floor(v floor(2^128/q) / 2^64): the quotient or up to two below it (v < q)
        u128 rem = ((u128)v << 64) - (u128)e * md.q;
        while (rem >= md.q) { ++e; rem -= md.q; }
        const ulonglong2 pr = make_ulonglong2(v, e);
        out[(((size_t)j * 2 + c) * slots + sl) * n + ksm_perm(i)] = pr;
        if (diag_out && k < L && j == k) diag_out[((size_t)sl * 2 + c) * n + i] = pr;
    }
}

// butterflies of register bit RB for the twiddle groups [G0, G0 + NG): tw[i] belongs to group G0 + i
template <int RB, int G0, int NG>
__device__ __forceinline__ void ksmi_layer(u64 (&x)[32], const ulonglong2* tw, const ArithU64::Mod& md) {
    static_for<0, NG>([&](auto gc) {
        constexpr int g = G0 + decltype(gc)::value;
        const ulonglong2 w = tw[decltype(gc)::value];
        static_for<0, (1 << RB)>([&](auto oc) {
            constexpr int R0 = (g << (RB + 1)) | decltype(oc)::value, R1 = R0 | (1 << RB);
            ArithU64::fwd(x[R0], x[R1], w, md);
        });
    });
}

// A 5-layer register round whose 31 twiddle pairs sit in a 32-slot vector (slot (1 << lvl) + g; slot 0 unused), fetched four slots at a
// time, KSMI_TW_AHEAD groups ahead of the butterflies that use them (ring of KSMI_TW_AHEAD + 1 buffers of four pairs).  Groups 0 .. KSMI_TW_AHEAD - 1
// (slots 0 .. 4 KSMI_TW_AHEAD - 1) were requested by the caller into tw[0 ..] before the LDS exchange that feeds the round.
#ifndef KSMI_TW_AHEAD
#define KSMI_TW_AHEAD 1
#endif
constexpr int KSMI_TW_BUFS = KSMI_TW_AHEAD + 1;
template <class LD>
__device__ __forceinline__ void ksmi_round5(u64 (&x)[32], ulonglong2 (&tw)[KSMI_TW_BUFS][4], LD&& ld, const ArithU64::Mod& md) {
    // group g (slots 4g .. 4g+3) serves: g0 -> layers rb 4 (slot 1) and rb 3 (slots 2, 3); g1 -> rb 2; g2, g3 -> rb 1; g4 .. g7 -> rb 0
    static_for<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g + KSMI_TW_AHEAD < 8) {
            constexpr int gn = g + KSMI_TW_AHEAD;
            static_for<0, 4>([&](auto ic) { tw[gn % KSMI_TW_BUFS][decltype(ic)::value] = ld(4 * gn + decltype(ic)::value); });
        }
        ulonglong2 (&t)[4] = tw[g % KSMI_TW_BUFS];
        if constexpr (g == 0) { ksmi_layer<4, 0, 1>(x, t + 1, md); ksmi_layer<3, 0, 2>(x, t + 2, md); }
        else if constexpr (g == 1) ksmi_layer<2, 0, 4>(x, t, md);
        else if constexpr (g <= 3) ksmi_layer<1, 4 * (g - 2), 4>(x, t, md);
        else ksmi_layer<0, 4 * (g - 4), 4>(x, t, md);
        __builtin_amdgcn_sched_barrier(0);
    });
}

#ifndef KSMI_LOAD_WINDOW
#define KSMI_LOAD_WINDOW 6       // register pairs of the half / quarter tile loaders in flight (two / four 16-byte loads each)
#endif
#ifndef KSMI_KEY_AHEAD
#define KSMI_KEY_AHEAD 1       // (1 vs 2 vs 3 measured: profiles/r05_ksmaci_variants.txt; 2 spills 16 more bytes per lane)
#endif

// EPI: 0 = every digit in the loop (coefficient-form target), 1 = DG (NTT-form target), 2 = TEN (fused chain)
template <int LOGN, int EPI>
__global__ __launch_bounds__(KSM_THREADS, 2) void ksmaci_kernel(KsMacIArgs a) {
    using A = ArithU64;
    static_assert(LOGN >= 13 && LOGN <= 15, "ksmaci covers N = 8192, 16384 and 32768");
    constexpr bool SPLIT = LOGN == 14;        // half tiles: one Cooley-Tukey layer applied while loading
    constexpr bool SPLIT4 = LOGN == 15;       // quarter tiles: two layers applied while loading
    constexpr unsigned N = 1u << LOGN;
    constexpr int HALVES = 1 << (LOGN - KSM_TB);
    __shared__ __attribute__((aligned(16))) u64 lds[KSM_LDS_WORDS];

    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    unsigned b, k, h;
    {
        const unsigned nrows = (unsigned)__builtin_popcountll(a.row_mask);
        const unsigned G = nrows * HALVES;
        unsigned g;
        if (a.grouped) {
            const unsigned per = 8u * G, r = blockIdx.x % per;
            g = r / 8u; b = (blockIdx.x / per) * 8u + (r % 8u);
        } else {
            g = blockIdx.x % G; b = blockIdx.x / G;
        }
        h = g % HALVES;
        k = g / HALVES;
    }
    const unsigned slot = k;                                      // rank of the row in the mask = slot of its prepared keys
    k = nth_set_bit(a.row_mask, k);
    const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;    // row of the key / modulus slot
    const unsigned mi = a.table_start + mrow;
    const A::Mod md = A::make(a.mods[mi]);

    typedef const ksmi_tw_mem __attribute__((address_space(4)))* ctp;
    const ctp tws = (ctp)(unsigned long long)(a.tw + (size_t)mi * N);       // scalar (wave-uniform) twiddle fetches
    auto tw_s = [&](unsigned idx) { const ksmi_tw_mem v = tws[idx]; return make_ulonglong2(v.x, v.y); };

    auto at = [](const void* ubase, unsigned byte_off) { return reinterpret_cast<const char*>(ubase) + byte_off; };
    const ulonglong2* r1u = a.tw_r1 + ((size_t)mi * (N >> 10) + h * (KSM_THREADS >> 5)) * 32;   // index bits above bit 9 = T >> 5
    const ulonglong2* r2u = a.tw_r2 + (size_t)mi * N + (size_t)h * (KSM_THREADS * 32);
    unsigned r1off = (t >> 5) * 512u;                           // bytes: 32 pairs per value of t >> 5
    unsigned r2off = wave * 32768u + lane * 32u;                // bytes: blocks of 2048 pairs per wave, two pairs (slots 2m, 2m+1) per lane
    unsigned slice_off = wave * 16384u + lane * 16u;            // bytes: u64 rows in the coalesced layout (two words per lane)

    u64 acc0[32], acc1[32];
    static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = 0; acc1[decltype(rc)::value] = 0; });

    // LDS addresses (padded words), as in ksmac2: round 0 holds registers r = b0 | b9<<1 | R3<<2 of tile index b0 | t<<1 | b9<<9 | R3<<10;
    // round 1 holds bits [5,10); round 2 holds bits [0,5).
    unsigned p0 = ksm_phys(t << 1);
    unsigned p1 = ksm_phys((t & 31u) | ((t >> 5) << 10));
    unsigned p2 = ksm_phys(t << 5);
    unsigned pt = ksm_phys(wave * 2048u + lane * 2u);

    const ulonglong2* kbase = a.keys + (size_t)slot * N + (size_t)h * (KSM_THREADS * 32);
    const u64* dig_item = a.digits + (long long)b * a.dig_bstride;
    const bool epi_row = EPI != 0 && k < a.L;      // data row of an instantiation whose epilogue takes the diagonal digit
    const unsigned steps = epi_row ? a.L - 1 : a.L;

    for (unsigned step = 0; step < steps; ++step) {
        const unsigned it = !epi_row ? step : (step < k ? step : step + 1);     // digit of this step
        u64 x[32];
        // nothing below depends on the digit except the input and the key: keep the compiler from hoisting twiddle loads / address
        // arithmetic out of the digit loop (and spilling them)
        asm volatile("" : "+v"(r1off), "+v"(r2off), "+v"(slice_off));
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(pt));
        const u64* gin_u = ksm_uniform(dig_item + (long long)it * a.dig_cstride);
        const unsigned gin_off = t << 4;
        // a digit of a modulus above this row's is reduced (Modulus::reduce); the others are canonical under this row's modulus already
        const bool need_reduce = a.mods[a.table_start + it].q > md.q;
        auto load_phase = [&](auto reduce_c) {
            constexpr bool RED = decltype(reduce_c)::value;
            auto red = [&](u64 v) { return RED ? barrett64(v, md.q, md.ratio_hi) : v; };
            if constexpr (SPLIT) {
                const ulonglong2 w1 = tw_s(1);
                constexpr int W0 = KSMI_LOAD_WINDOW;
                ulonglong2 ru[16], rv[16];
                auto request = [&](auto ic) {
                    constexpr int i = decltype(ic)::value;     // i = b9 | R3<<1
                    ru[i] = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                    rv[i] = ksm_gload<ulonglong2>(gin_u + 8192 + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, W0>([&](auto ic) { request(ic); });
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    // u < q, w v in [0, 4q): upper half of the outputs (h = 1) takes u + 4q - w v, both in [0, 5q]
                    const u64 u0 = red(ru[i].x), u1 = red(ru[i].y);
                    const u64 m0 = A::shoup_lazy3(red(rv[i].x), w1.x, w1.y, md.neg_q), m1 = A::shoup_lazy3(red(rv[i].y), w1.x, w1.y, md.neg_q);
                    x[2 * i] = h ? u0 + md.four_q - m0 : u0 + m0;
                    x[2 * i + 1] = h ? u1 + md.four_q - m1 : u1 + m1;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i + W0 < 16) request(std::integral_constant<int, i + W0>{});
                });
            } else if constexpr (SPLIT4) {
                // quarter tile h = 2 hA + hB of a 2^15-point transform: with (a, b, c, d) = x[i], x[i+N/4], x[i+N/2], x[i+3N/4],
                //   layer 0:  u = a +- w1 c,  v = b +- w1 d     (minus for the upper half hA = 1)
                //   layer 1:  x = u +- wB v,  wB = tw[2 + hA]    (minus for the odd quarter hB = 1)
                const ulonglong2 w1 = tw_s(1), wB = tw_s(2 + (h >> 1));
                const bool nA = (h >> 1) != 0, nB = (h & 1) != 0;
                constexpr int W4 = (KSMI_LOAD_WINDOW + 1) / 2;
                ulonglong2 ra[16], rb_[16], rc[16], rd[16];
                auto request = [&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr unsigned off = ((i & 1) << 9) + ((i >> 1) << 10);
                    ra[i] = ksm_gload<ulonglong2>(gin_u + off, gin_off);
                    rb_[i] = ksm_gload<ulonglong2>(gin_u + 8192 + off, gin_off);
                    rc[i] = ksm_gload<ulonglong2>(gin_u + 16384 + off, gin_off);
                    rd[i] = ksm_gload<ulonglong2>(gin_u + 24576 + off, gin_off);
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, W4>([&](auto ic) { request(ic); });
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    auto two_layers = [&](u64 av, u64 bv, u64 cv, u64 dv) {
                        const u64 mc = A::shoup_lazy3(red(cv), w1.x, w1.y, md.neg_q), mdv = A::shoup_lazy3(red(dv), w1.x, w1.y, md.neg_q);
                        const u64 ra_ = red(av), rb2 = red(bv);
                        const u64 u = nA ? ra_ + md.four_q - mc : ra_ + mc;          // [0, 5q]
                        const u64 v = nA ? rb2 + md.four_q - mdv : rb2 + mdv;        // [0, 5q]: any 64-bit word may enter the Shoup product
                        const u64 mv = A::shoup_lazy3(v, wB.x, wB.y, md.neg_q);
                        const u64 uc = A::csub4(u, md);                               // [0, 4q)
                        return nB ? uc + md.four_q - mv : uc + mv;                    // [0, 8q)
                    };
                    x[2 * i] = two_layers(ra[i].x, rb_[i].x, rc[i].x, rd[i].x);
                    x[2 * i + 1] = two_layers(ra[i].y, rb_[i].y, rc[i].y, rd[i].y);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (i + W4 < 16) request(std::integral_constant<int, i + W4>{});
                });
            } else {
                ulonglong2 rw[16];
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    rw[i] = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                });
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    x[2 * i] = red(rw[i].x);
                    x[2 * i + 1] = red(rw[i].y);
                });
            }
        };
        if (need_reduce) load_phase(std::true_type{}); else load_phase(std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 0: tile bits 12, 11, 10 = register bits 4, 3, 2; twiddles are workgroup-uniform (SGPR pairs) ----------
        static_for<0, 3>([&](auto lc) {
            constexpr int li = decltype(lc)::value;
            constexpr int bit = 12 - li, rb = 4 - li;
            static_for<0, (1 << li)>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const unsigned idx = (N >> (bit + 1)) + (h << (12 - bit)) + g;
                const ulonglong2 w = tw_s(idx);
                static_for<0, (1 << rb)>([&](auto oc) {
                    constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                    A::fwd(x[R0], x[R1], w, md);
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 0 -> 1 -------------------------------------------------------------------------------
        ulonglong2 tw[KSMI_TW_BUFS][4];
        auto ld1 = [&](int s) { return ksm_gload<ulonglong2>(r1u + s, r1off); };
        auto ld2 = [&](int s) { return ksm_gload<ulonglong2>(r2u + ((s >> 1) * 128 + (s & 1)), r2off); };
        __syncthreads();     // every wave has finished reading its slice of the previous digit
        static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr unsigned off = ksm_phys(((i & 1) << 9) | ((i >> 1) << 10));
            *reinterpret_cast<ulonglong2*>(&lds[p0 + off]) = make_ulonglong2(x[2 * i], x[2 * i + 1]);
        });
        // the first twiddle group(s) of round 1 travel while the exchange completes (x is dead here)
        static_for<1, 4 * KSMI_TW_AHEAD>([&](auto ic) { constexpr int sl = decltype(ic)::value; tw[sl / 4][sl % 4] = ld1(sl); });
        tw[0][0] = tw[0][1];
        __syncthreads();
        static_for<0, 32>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            x[R] = lds[p1 + 34 * R];
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 1: tile bits 9..5 = register bits 4..0 ----------------------------------------------------
        ksmi_round5(x, tw, ld1, md);
        // ---- exchange 1 -> 2: stays inside groups of 32 consecutive threads (tile bits [10,13) = t >> 5 on both sides) -------
        static_for<0, 32>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            lds[p1 + 34 * R] = x[R];
        });
        static_for<1, 4 * KSMI_TW_AHEAD>([&](auto ic) { constexpr int sl = decltype(ic)::value; tw[sl / 4][sl % 4] = ld2(sl); });
        tw[0][0] = tw[0][1];
        __builtin_amdgcn_wave_barrier();
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[p2 + 2 * m]);
            x[2 * m] = v.x; x[2 * m + 1] = v.y;
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 2: tile bits 4..0 = register bits 4..0, lane-interleaved twiddle pairs ----------------------
        ksmi_round5(x, tw, ld2, md);
        // ---- multiply-accumulate with key `it` straight from the registers: lazy Shoup products (any 64-bit digit word may enter) --------
        {
            const ulonglong2* k0 = ksm_uniform(kbase + (long long)it * a.key_jstride);
            const ulonglong2* k1 = ksm_uniform(k0 + a.key_pstride);
            constexpr int AHEAD = KSMI_KEY_AHEAD;
            ulonglong2 y0a[16], y0b[16], y1a[16], y1b[16];      // component 0 / 1, coefficients 2m (a) and 2m + 1 (b)
            auto request = [&](auto mc) {
                constexpr int m = decltype(mc)::value;
                y0a[m] = ksm_gload<ulonglong2>(k0 + m * 128, r2off);
                y0b[m] = ksm_gload<ulonglong2>(k0 + m * 128 + 1, r2off);
                y1a[m] = ksm_gload<ulonglong2>(k1 + m * 128, r2off);
                y1b[m] = ksm_gload<ulonglong2>(k1 + m * 128 + 1, r2off);
            };
            static_for<0, AHEAD>([&](auto mc) { request(mc); });
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m + AHEAD < 16) request(std::integral_constant<int, m + AHEAD>{});
                A::mac_shoup(acc0[2 * m], x[2 * m], y0a[m].x, y0a[m].y, md);
                A::mac_shoup(acc1[2 * m], x[2 * m], y1a[m].x, y1a[m].y, md);
                A::mac_shoup(acc0[2 * m + 1], x[2 * m + 1], y0b[m].x, y0b[m].y, md);
                A::mac_shoup(acc1[2 * m + 1], x[2 * m + 1], y1b[m].x, y1b[m].y, md);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- results: both accumulators cross the wave's own LDS slice into the coalesced layout (lazy words, [0, 4q)) --------------------
    u64* go = a.out + (long long)b * a.out_bstride + (long long)k * a.out_cstride + (size_t)h * (KSM_THREADS * 32);
    // (no workgroup barrier: after a digit's first exchange every LDS access of a wave stays inside the wave's own slice)
    auto cross = [&](u64 (&acc)[32]) {
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            *reinterpret_cast<ulonglong2*>(&lds[p2 + 2 * m]) = make_ulonglong2(acc[2 * m], acc[2 * m + 1]);
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[pt + ksm_phys(m * 128u)]);
            acc[2 * m] = v.x; acc[2 * m + 1] = v.y;
        });
        __builtin_amdgcn_wave_barrier();
    };
    cross(acc0);
    cross(acc1);
    __builtin_amdgcn_sched_barrier(0);
    auto store_pair = [&](int m, u64 q0x, u64 q0y, u64 q1x, u64 q1y) {
        nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + m * 128, slice_off))), q0x, q0y);
        nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + a.out_pstride + m * 128, slice_off))), q1x, q1y);
    };
    if (!epi_row) {
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            store_pair(m, A::final_fwd(acc0[2 * m], md), A::final_fwd(acc0[2 * m + 1], md), A::final_fwd(acc1[2 * m], md), A::final_fwd(acc1[2 * m + 1], md));
        });
        return;
    }
    if constexpr (EPI != 0) {
        // data row: the diagonal digit (EPI 1: the target's own limb k; EPI 2: a1 (.) b1 of limb k) times key_k under modulus k, natural order,
        // and (EPI 2) the tensor terms c0 = a0 b0, c1 = a0 b1 + a1 b0 -- the keys carry the factor qk^-1, so the row leaves as Q = P qk^-1 + c
        const ulonglong2* dk0 = ksm_uniform(a.diag_keys + (size_t)slot * 2 * N + (size_t)h * (KSM_THREADS * 32));
        const ulonglong2* dk1 = ksm_uniform(dk0 + N);
        const u64* dg = nullptr; const u64* ta0 = nullptr; const u64* tb0 = nullptr; const u64* ta1 = nullptr; const u64* tb1 = nullptr;
        if constexpr (EPI == 1) dg = ksm_uniform(a.diag + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + (size_t)h * (KSM_THREADS * 32));
        else {
            const size_t toff = (size_t)b * a.ten_bstride + (size_t)k * N + (size_t)h * (KSM_THREADS * 32);
            ta0 = ksm_uniform(a.ten_a + toff); tb0 = ksm_uniform(a.ten_b + toff);
            ta1 = ksm_uniform(a.ten_a + toff + a.ten_pstride); tb1 = ksm_uniform(a.ten_b + toff + a.ten_pstride);
        }
        constexpr int W = 2;
        ulonglong2 xd[16], xa0[EPI == 2 ? 16 : 1], xb0[EPI == 2 ? 16 : 1], xb1[EPI == 2 ? 16 : 1];
        ulonglong2 y0a[16], y0b[16], y1a[16], y1b[16];
        const unsigned pair_off = wave * 32768u + lane * 32u;     // diagonal key pairs in natural order: two pairs per lane
        auto request = [&](auto ic) {
            constexpr int m = decltype(ic)::value;
            if constexpr (EPI == 1) xd[m] = ksm_gload<ulonglong2>(dg + m * 128, slice_off);
            else {
                xd[m] = ksm_gload<ulonglong2>(ta1 + m * 128, slice_off);
                xb1[m] = ksm_gload<ulonglong2>(tb1 + m * 128, slice_off);
                xa0[m] = ksm_gload<ulonglong2>(ta0 + m * 128, slice_off);
                xb0[m] = ksm_gload<ulonglong2>(tb0 + m * 128, slice_off);
            }
            y0a[m] = ksm_gload<ulonglong2>(dk0 + m * 128, pair_off);
            y0b[m] = ksm_gload<ulonglong2>(dk0 + m * 128 + 1, pair_off);
            y1a[m] = ksm_gload<ulonglong2>(dk1 + m * 128, pair_off);
            y1b[m] = ksm_gload<ulonglong2>(dk1 + m * 128 + 1, pair_off);
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, W>([&](auto ic) { request(ic); });
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            __builtin_amdgcn_sched_barrier(0);
            u64 q0x = acc0[2 * m], q0y = acc0[2 * m + 1], q1x = acc1[2 * m], q1y = acc1[2 * m + 1];      // [0, 4q)
            u64 dx, dy;
            if constexpr (EPI == 1) { dx = xd[m].x; dy = xd[m].y; }
            else { dx = A::prod(xd[m].x, xb1[m].x, md); dy = A::prod(xd[m].y, xb1[m].y, md); }
            A::mac_shoup(q0x, dx, y0a[m].x, y0a[m].y, md); A::mac_shoup(q1x, dx, y1a[m].x, y1a[m].y, md);
            A::mac_shoup(q0y, dy, y0b[m].x, y0b[m].y, md); A::mac_shoup(q1y, dy, y1b[m].x, y1b[m].y, md);
            q0x = A::final_fwd(q0x, md); q0y = A::final_fwd(q0y, md); q1x = A::final_fwd(q1x, md); q1y = A::final_fwd(q1y, md);
            if constexpr (EPI == 2) {
                q0x = add_mod(q0x, A::prod(xa0[m].x, xb0[m].x, md), md.q);
                q0y = add_mod(q0y, A::prod(xa0[m].y, xb0[m].y, md), md.q);
                q1x = add_mod(q1x, add_mod(A::prod(xa0[m].x, xb1[m].x, md), A::prod(xd[m].x, xb0[m].x, md), md.q), md.q);
                q1y = add_mod(q1y, add_mod(A::prod(xa0[m].y, xb1[m].y, md), A::prod(xd[m].y, xb0[m].y, md), md.q), md.q);
            }
            store_pair(m, q0x, q0y, q1x, q1y);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m + W < 16) request(std::integral_constant<int, m + W>{});
        });
    }
}

}  // namespace troyn
