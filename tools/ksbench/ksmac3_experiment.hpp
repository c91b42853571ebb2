// ksmac3_kernels.hpp -- key-switch inner product, third generation: wave-specialised, persistent workgroups.
//
// Same function as ksmac2_kernel (ksmac_kernels.hpp; reference fgk/switch_key.cu:6-54, :83-154 driven from
// evaluator_keyswitching_core.cu:904-919):
//     out[k][c] = sum_j  NTT_{q_key(k)}( digit_j mod q_key(k) ) (.) key_j[c][k]          c = 0, 1
// and the same tiles, register rounds, twiddle / key layouts and arithmetic.  What changes is WHO does what.
//
// ksmac2 runs two independent 256-thread workgroups per CU; every wave walks load -> layer 0 -> round 0 -> exchange -> round 1 ->
// exchange -> round 2 -> multiply-accumulate for each digit, in order, with one vmcnt stream: 40 % of its wave cycles wait
// (profiles/r03_bench_ksmac_counters.json), mostly for the digit rows (one exposed HBM latency per digit), for the two workgroup
// barriers of exchange 0 and for whatever of the key / twiddle latency the 64 spare registers cannot cover.
//
// Here ONE 512-thread workgroup owns a CU for the whole launch and its waves have two roles (one of each per SIMD):
//   * waves 0-3, PRODUCERS: no accumulators, so the 32 sixteen-byte loads of a digit are ALL in flight at once (128 VGPRs); they
//     apply layer 0 and round 0 and leave the tile in LDS buffer (step & 1).  They run one step ahead of the consumers, across
//     tile boundaries, so a digit's HBM latency is never on the consumers' path.  Fused chain (EPI = 1): they also form the
//     epilogue addend of the tile -- tensor terms a0 b0, a0 b1 + a1 b0 and the diagonal digit (a1 b1) key_kk (what ksmac2's TEN
//     epilogue did with cold operand rows while the whole workgroup waited) -- a quarter of the tile per step, and park it as
//     doubles in the tile's own output rows; EPI = 2 (separate key switch, NTT-form target): the diagonal digit's term alone.
//   * waves 4-7, CONSUMERS: rounds 1 and 2, the multiply-accumulate against the prepared keys, the 2 x 32 accumulators; their
//     epilogue crosses the accumulators through their own slice, adds the parked addend (two rows, L2-warm, written by the same
//     CU a few steps earlier) and stores canonical words.
//   * ONE s_barrier per step replaces the two per digit: the producers fill buffer (n+1)&1 while the consumers work on buffer n&1.
// Results are canonical residues of exact integer arithmetic: bit-identical to ksmac2 (tests: every key-switch test of the suite
// runs through both, tools/ksbench compares them word for word).
#pragma once
#include "../../troy-nova_amd/csrc/ksmac_kernels.hpp"

namespace troyn {

constexpr int KSM3_THREADS = 2 * KSM_THREADS;      // 4 producer + 4 consumer waves
constexpr unsigned KSM3_LDS_BYTES = 2u * KSM_LDS_WORDS * 8u;

// (item, row, tile) of virtual block vb: the workgroup orders of ksmac2 (KsMacArgs::grouped), with vb in the place of blockIdx.x
template <int HALVES>
__device__ __forceinline__ bool ksm3_decode(const KsMacArgs& a, unsigned nrows, unsigned vb, unsigned& b, unsigned& k, unsigned& h) {
    if (a.grouped == 3) {
        constexpr unsigned ITEMS = 64u / (2u * HALVES);
        const unsigned bands = (nrows + 1u) / 2u, per = bands * 64u;
        const unsigned xcd = vb & 7u, sq = vb >> 3, r = sq % per, r2 = r & 63u;
        k = 2u * (r >> 6) + (r2 % (2u * HALVES)) / HALVES;
        h = r2 % HALVES;
        b = ((sq / per) * 8u + xcd) * ITEMS + r2 / (2u * HALVES);
        if (k >= nrows) return false;
    } else if (a.grouped == 2) {
        const unsigned per = 8u * HALVES, r = vb % per, q = vb / per;
        h = r / 8u; b = (q % (a.batch / 8u)) * 8u + (r % 8u); k = q / (a.batch / 8u);
    } else {
        const unsigned G = nrows * HALVES;
        unsigned g;
        if (a.grouped) {
            const unsigned per = 8u * G, r = vb % per;
            g = r / 8u; b = (vb / per) * 8u + (r % 8u);
        } else {
            g = vb % G; b = vb / G;
        }
        k = g / HALVES; h = g % HALVES;
    }
    if (a.row_mask) k = nth_set_bit(a.row_mask, k);
    return true;
}

// EPI: 0 no epilogue addend (coefficient-form target, or a target whose diagonal digit is not wanted), 1 fused chain (ten_a / ten_b /
// diag_keys: Q = P qk^-1 + tensor terms, diagonal digit a1 (.) b1), 2 separate key switch on an NTT-form target (diag + diag_keys)
// a.digits never holds the diagonal digit's row for the data rows of EPI != 0 (it is skipped like in ksmac2's TEN / DG forms).
template <int LOGN, bool DIGF64, bool WIDE, int EPI>
__global__ __launch_bounds__(KSM3_THREADS, 2) void ksmac3_kernel(KsMacArgs a, unsigned total_vb) {
    static_assert(LOGN == 14, "ksmac3: half tiles of N = 16384 (the other sizes stay on ksmac2)");
    constexpr unsigned N = 1u << LOGN;
    constexpr int HALVES = 1 << (LOGN - KSM_TB);
    __shared__ __attribute__((aligned(16))) u64 lds[2 * KSM_LDS_WORDS];

    const unsigned role = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));     // 0 producer, 1 consumer (wave-uniform)
    const unsigned t = threadIdx.x & 255u, lane = t & 63u, wave = t >> 6;
    const unsigned nrows = a.row_mask ? (unsigned)__builtin_popcountll(a.row_mask) : a.L + 1;
    const unsigned nW = gridDim.x >> 3, xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3;

    auto at = [](const void* ubase, unsigned byte_off) { return reinterpret_cast<const char*>(ubase) + byte_off; };
    unsigned slice_off = wave * 16384u + lane * 16u;      // bytes: the coalesced layout (16-byte chunk m of a wave's 16 KiB at + m KiB)
    unsigned n = 0;                                        // step counter of this workgroup (both roles count alike)

    if (role == 0) {
        // =========================================== PRODUCERS ===========================================
        unsigned p0 = ksm_phys(t << 1);
        for (unsigned qi = 0;; ++qi) {
            const unsigned vb = ((qi * nW + wi) << 3) | xcd;
            if (vb >= total_vb) break;
            unsigned b, k, h;
            if (!ksm3_decode<HALVES>(a, nrows, vb, b, k, h)) continue;
            const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
            const unsigned mi = a.table_start + mrow;
            const DevModulus dm = a.mods[mi];
            const F64Mod fm{dm.pd, dm.inv_pd};
            const double p = fm.p, inv_p = fm.inv_p;
            typedef const double __attribute__((address_space(4)))* cdp;
            const cdp tws = (cdp)(unsigned long long)(a.tw + (size_t)mi * N);
            const u64* dig_item = a.digits + (long long)b * a.dig_bstride;
            const bool epi_row = EPI != 0 && k < a.L;
            const unsigned steps = epi_row ? a.L - 1 : a.L;
            u64* go = a.out + (long long)b * a.out_bstride + (long long)k * a.out_cstride + (size_t)h * (KSM_THREADS * 32);
            auto dig_in = [&](u64 raw) -> double {
                if constexpr (DIGF64) return f64_bits_to_double(raw);
                else if constexpr (WIDE) return f64_from_u64(barrett64(raw, dm.q, dm.ratio_hi));
                else return f64_from_u64(raw);
            };
            auto mac2z = [&](double& e0, double& e1, double v, double y0, double y1) {      // the term of ksmac2's mac2, from zero
                const double h0 = v * y0, h1 = v * y1;
                const double l0 = __builtin_fma(v, y0, -h0), l1 = __builtin_fma(v, y1, -h1);
                const double q0 = __builtin_rint(h0 * inv_p), q1 = __builtin_rint(h1 * inv_p);
                e0 = __builtin_fma(-q0, p, h0) + l0;
                e1 = __builtin_fma(-q1, p, h1) + l1;
            };
            // the epilogue addend is formed in four passes of four 16-byte chunks per lane; pass q belongs to step q * steps / 4
            unsigned pass = 0;
            for (unsigned step = 0; step < steps; ++step) {
                const unsigned it = !epi_row ? step : (step < k ? step : step + 1);
                asm volatile("" : "+v"(p0), "+v"(slice_off));
                u64* buf = lds + (n & 1u) * KSM_LDS_WORDS;
                // ---- every load of the digit in flight at once ------------------------------------------------------
                const u64* gin_u = ksm_uniform(dig_item + (long long)it * a.dig_cstride);
                const unsigned gin_off = t << 4;
                ulonglong2 ru[16], rv[16];
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;     // i = b9 | R3<<1
                    ru[i] = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                    rv[i] = ksm_gload<ulonglong2>(gin_u + 8192 + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                });
                __builtin_amdgcn_sched_barrier(0);
                // ---- epilogue addend, pass `pass`: its operand loads queue behind the digit's -------------------------
                const bool do_pass = EPI != 0 && epi_row && pass < 4u && (pass * steps) / 4u == step;
                ulonglong2 xa0[4], xb0[4], xa1[4], xb1[4];
                double2 y0[4], y1[4];
                const size_t toff = (size_t)b * a.ten_bstride + (size_t)k * N + (size_t)h * (KSM_THREADS * 32);
                auto epi_request = [&](unsigned ps) {
                    const unsigned m0 = ps * 4u;
                    if constexpr (EPI == 1) {
                        const u64* ta0 = ksm_uniform(a.ten_a + toff + m0 * 128u);
                        const u64* tb0 = ksm_uniform(a.ten_b + toff + m0 * 128u);
                        const u64* ta1 = ksm_uniform(a.ten_a + toff + a.ten_pstride + m0 * 128u);
                        const u64* tb1 = ksm_uniform(a.ten_b + toff + a.ten_pstride + m0 * 128u);
                        const double* dk0 = ksm_uniform(a.diag_keys + (size_t)k * 2 * N + (size_t)h * (KSM_THREADS * 32) + m0 * 128u);
                        const double* dk1 = ksm_uniform(dk0 + N);
                        static_for<0, 4>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            xa1[c] = ksm_gload<ulonglong2>(ta1 + c * 128, slice_off);
                            xb1[c] = ksm_gload<ulonglong2>(tb1 + c * 128, slice_off);
                            y0[c] = ksm_gload<double2>(dk0 + c * 128, slice_off);
                            y1[c] = ksm_gload<double2>(dk1 + c * 128, slice_off);
                            xa0[c] = ksm_gload<ulonglong2>(ta0 + c * 128, slice_off);
                            xb0[c] = ksm_gload<ulonglong2>(tb0 + c * 128, slice_off);
                        });
                    } else if constexpr (EPI == 2) {
                        const u64* dg = ksm_uniform(a.diag + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + (size_t)h * (KSM_THREADS * 32) + m0 * 128u);
                        const double* dk0 = ksm_uniform(a.diag_keys + (size_t)k * 2 * N + (size_t)h * (KSM_THREADS * 32) + m0 * 128u);
                        const double* dk1 = ksm_uniform(dk0 + N);
                        static_for<0, 4>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            xa1[c] = ksm_gload<ulonglong2>(dg + c * 128, slice_off);
                            y0[c] = ksm_gload<double2>(dk0 + c * 128, slice_off);
                            y1[c] = ksm_gload<double2>(dk1 + c * 128, slice_off);
                        });
                    }
                };
                auto epi_finish = [&](unsigned ps) {
                    u64* eo = go + ps * 4u * 128u;
                    static_for<0, 4>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        double e0x, e0y, e1x, e1y;
                        if constexpr (EPI == 1) {
                            const double a0x = f64_from_u64(xa0[c].x), a0y = f64_from_u64(xa0[c].y), b0x = f64_from_u64(xb0[c].x), b0y = f64_from_u64(xb0[c].y);
                            const double a1x = f64_from_u64(xa1[c].x), a1y = f64_from_u64(xa1[c].y), b1x = f64_from_u64(xb1[c].x), b1y = f64_from_u64(xb1[c].y);
                            // diagonal digit d = a1 (.) b1 re-centred, times the key of digit k under modulus k (natural order copy)
                            const double dx = f64_corr(f64_mulq(f64_corr(a1x, fm), b1x, inv_p, p), fm), dy = f64_corr(f64_mulq(f64_corr(a1y, fm), b1y, inv_p, p), fm);
                            mac2z(e0x, e1x, dx, y0[c].x, y1[c].x);
                            mac2z(e0y, e1y, dy, y0[c].y, y1[c].y);
                            // tensor terms; canonical factors below p: each product is within (-0.875 p, 0.875 p)
                            e0x += f64_mulq(a0x, b0x, inv_p, p);
                            e0y += f64_mulq(a0y, b0y, inv_p, p);
                            e1x += f64_mulq(a0x, b1x, inv_p, p) + f64_mulq(a1x, b0x, inv_p, p);
                            e1y += f64_mulq(a0y, b1y, inv_p, p) + f64_mulq(a1y, b0y, inv_p, p);
                        } else {
                            const double dx = f64_corr(f64_from_u64(xa1[c].x), fm), dy = f64_corr(f64_from_u64(xa1[c].y), fm);
                            mac2z(e0x, e1x, dx, y0[c].x, y1[c].x);
                            mac2z(e0y, e1y, dy, y0[c].y, y1[c].y);
                        }
                        // |e0| <= 1.6 p, |e1| <= 2.5 p: parked as doubles in the tile's own output rows (plain stores: the line stays in this XCD's L2)
                        *reinterpret_cast<double2*>(const_cast<char*>(at(eo + c * 128, slice_off))) = make_double2(e0x, e0y);
                        *reinterpret_cast<double2*>(const_cast<char*>(at(eo + a.out_pstride + c * 128, slice_off))) = make_double2(e1x, e1y);
                    });
                };
                if constexpr (EPI != 0) { if (do_pass) epi_request(pass); }
                __builtin_amdgcn_sched_barrier(0);
                // ---- layer 0 (half tile: u +- w v) ---------------------------------------------------------------------
                double x[32];
                {
                    const double w1 = tws[1];
                    const double sgn = h ? -1.0 : 1.0;
                    static_for<0, 16>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const double u0 = dig_in(ru[i].x), u1 = dig_in(ru[i].y), v0 = dig_in(rv[i].x), v1 = dig_in(rv[i].y);
                        x[2 * i] = f64_corr(__builtin_fma(sgn, f64_mulq(v0, w1, inv_p, p), u0), fm);
                        x[2 * i + 1] = f64_corr(__builtin_fma(sgn, f64_mulq(v1, w1, inv_p, p), u1), fm);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
                // ---- round 0: tile bits 12, 11, 10 = register bits 4, 3, 2; twiddles are workgroup-uniform ----------
                static_for<0, 3>([&](auto lc) {
                    constexpr int li = decltype(lc)::value;
                    constexpr int bit = 12 - li, rb = 4 - li;
                    static_for<0, (1 << li)>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        const unsigned idx = (N >> (bit + 1)) + (h << (12 - bit)) + g;
                        const double w = tws[idx];
                        static_for<0, (1 << rb)>([&](auto oc) {
                            constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                            const double r = f64_mulq(x[R1], w, inv_p, p);
                            const double u = x[R0];
                            x[R0] = u + r; x[R1] = u - r;
                        });
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
                // ---- hand-over: buffer n & 1 (the consumers left it before the previous barrier) ------------------------
                static_for<0, 16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr unsigned off = ksm_phys(((i & 1) << 9) | ((i >> 1) << 10));
                    *reinterpret_cast<double2*>(&buf[p0 + off]) = make_double2(f64_corr(x[2 * i], fm), f64_corr(x[2 * i + 1], fm));
                });
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (EPI != 0) {
                    if (do_pass) { epi_finish(pass); ++pass; }
                    // short chains: the passes that do not get a step of their own
                    while (epi_row && pass < 4u && (pass * steps) / 4u == step) { epi_request(pass); epi_finish(pass); ++pass; }
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                ++n;
            }
        }
    } else {
        // =========================================== CONSUMERS ===========================================
        unsigned p1 = ksm_phys((t & 31u) | ((t >> 5) << 10));                       // + 34 * R
        unsigned p2 = ksm_phys(t << 5);                                             // + R
        unsigned pt = ksm_phys(wave * 2048u + lane * 2u);                            // transposed pairs: + ksm_phys(128 m)
        unsigned r1off = (t >> 5) * 256u;
        for (unsigned qi = 0;; ++qi) {
            const unsigned vb = ((qi * nW + wi) << 3) | xcd;
            if (vb >= total_vb) break;
            unsigned b, k, h;
            if (!ksm3_decode<HALVES>(a, nrows, vb, b, k, h)) continue;
            const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
            const unsigned mi = a.table_start + mrow;
            const DevModulus dm = a.mods[mi];
            const F64Mod fm{dm.pd, dm.inv_pd};
            const double p = fm.p, inv_p = fm.inv_p;
            const double* r1u = a.tw_r1 + ((size_t)mi * (N >> 10) + h * (KSM_THREADS >> 5)) * 32;
            const double* r2u = a.tw_r2 + (size_t)mi * N + (size_t)h * (KSM_THREADS * 32);
            const double* kbase = a.keys + (size_t)mrow * N + (size_t)h * (KSM_THREADS * 32);
            const bool epi_row = EPI != 0 && k < a.L;
            const unsigned steps = epi_row ? a.L - 1 : a.L;
            u64* go = a.out + (long long)b * a.out_bstride + (long long)k * a.out_cstride + (size_t)h * (KSM_THREADS * 32);

            double acc0[32], acc1[32];
            static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = 0.0; acc1[decltype(rc)::value] = 0.0; });
            auto mac2 = [&](double& a0, double& a1, double v, double y0, double y1) {
                const double h0 = v * y0, h1 = v * y1;
                const double l0 = __builtin_fma(v, y0, -h0), l1 = __builtin_fma(v, y1, -h1);
                const double q0 = __builtin_rint(h0 * inv_p), q1 = __builtin_rint(h1 * inv_p);
                a0 += __builtin_fma(-q0, p, h0) + l0;
                a1 += __builtin_fma(-q1, p, h1) + l1;
            };
            for (unsigned step = 0; step < steps; ++step) {
                const unsigned it = !epi_row ? step : (step < k ? step : step + 1);
                asm volatile("" : "+v"(r1off), "+v"(slice_off));
                asm volatile("" : "+v"(p1), "+v"(p2), "+v"(pt));
                u64* buf = lds + (n & 1u) * KSM_LDS_WORDS;
                double x[32];
                double ta[8], tb[8];
                // the first 16 twiddle slots of round 1 travel while this wave waits for the producers
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + 2 * q, r1off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + 8 + 2 * q, r1off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                static_for<0, 32>([&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    x[R] = f64_bits_to_double(buf[p1 + 34 * R]);
                });
                __builtin_amdgcn_sched_barrier(0);
                // ---- round 1: tile bits 9..5 = register bits 4..0 ----------------------------------------------------
                ksm_round5<false>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r1u + 2 * q, r1off); }, [] {}, inv_p, p);
                // ---- exchange 1 -> 2: inside the wave's own slice (ksmac2) -------------------------------------------
                static_for<0, 32>([&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    buf[p1 + 34 * R] = f64_double_to_bits(f64_corr(x[R], fm));
                });
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + 128 * q, slice_off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + 128 * (4 + q), slice_off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
                __builtin_amdgcn_wave_barrier();
                static_for<0, 16>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const double2 v = *reinterpret_cast<const double2*>(&buf[p2 + 2 * m]);
                    x[2 * m] = v.x; x[2 * m + 1] = v.y;
                });
                __builtin_amdgcn_sched_barrier(0);
                // ---- round 2: tile bits 4..0, lane-interleaved twiddle vectors -----------------------------------------
                ksm_round5<false>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r2u + 128 * q, slice_off); }, [] {}, inv_p, p);
                // ---- multiply-accumulate with key `it` straight from the registers ---------------------------------------
                {
                    const double* k0 = ksm_uniform(kbase + (long long)it * a.key_jstride);
                    const double* k1 = ksm_uniform(k0 + a.key_pstride);
                    constexpr int AHEAD = KSM_KEY_AHEAD;
                    double2 y0[16], y1[16];
                    static_for<0, AHEAD>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        y0[m] = ksm_gload<double2>(k0 + m * 128, slice_off);
                        y1[m] = ksm_gload<double2>(k1 + m * 128, slice_off);
                    });
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (m + AHEAD < 16) {
                            y0[m + AHEAD] = ksm_gload<double2>(k0 + (m + AHEAD) * 128, slice_off);
                            y1[m + AHEAD] = ksm_gload<double2>(k1 + (m + AHEAD) * 128, slice_off);
                        }
                        const double v0 = f64_corr(x[2 * m], fm), v1 = f64_corr(x[2 * m + 1], fm);
                        mac2(acc0[2 * m], acc1[2 * m], v0, y0[m].x, y1[m].x);
                        mac2(acc0[2 * m + 1], acc1[2 * m + 1], v1, y0[m].y, y1[m].y);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
                if ((step & 7u) == 7u)
                    static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = f64_corr(acc0[decltype(rc)::value], fm); acc1[decltype(rc)::value] = f64_corr(acc1[decltype(rc)::value], fm); });
                ++n;
            }
            // ---- epilogue: through the wave's own slice of the buffer it consumed last (its until the next barrier) -------
            u64* buf = lds + ((n - 1u) & 1u) * KSM_LDS_WORDS;
            if (EPI != 0 && epi_row) {
                // data row with a parked addend: both accumulators cross the slice in place (re-centred doubles), then one sweep
                // adds the addend (16-byte loads, rolling window) and stores canonical words over it
                constexpr int W = 4;
                double2 e0[16], e1[16];
                auto request = [&](auto ic) {
                    constexpr int m = decltype(ic)::value;
                    e0[m] = ksm_gload<double2>(go + m * 128, slice_off);
                    e1[m] = ksm_gload<double2>(go + a.out_pstride + m * 128, slice_off);
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, W>([&](auto ic) { request(ic); });
                __builtin_amdgcn_sched_barrier(0);
                auto cross = [&](double (&acc)[32]) {
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        *reinterpret_cast<double2*>(&buf[p2 + 2 * m]) = make_double2(f64_corr(acc[2 * m], fm), f64_corr(acc[2 * m + 1], fm));
                    });
                    __builtin_amdgcn_wave_barrier();
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        const double2 v = *reinterpret_cast<const double2*>(&buf[pt + ksm_phys(m * 128u)]);
                        acc[2 * m] = v.x; acc[2 * m + 1] = v.y;
                    });
                    __builtin_amdgcn_wave_barrier();
                };
                cross(acc0);
                cross(acc1);
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 16>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    const double q0x = acc0[2 * m] + e0[m].x, q0y = acc0[2 * m + 1] + e0[m].y;      // |.| <= p/2 + 1 + 2.5 p
                    const double q1x = acc1[2 * m] + e1[m].x, q1y = acc1[2 * m + 1] + e1[m].y;
                    nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + m * 128, slice_off))), f64_canon(q0x, fm), f64_canon(q0y, fm));
                    nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + a.out_pstride + m * 128, slice_off))), f64_canon(q1x, fm), f64_canon(q1y, fm));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (m + W < 16) request(std::integral_constant<int, m + W>{});
                });
            } else {
                static_for<0, 2>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        const ulonglong2 v = make_ulonglong2(f64_canon(c ? acc1[2 * m] : acc0[2 * m], fm), f64_canon(c ? acc1[2 * m + 1] : acc0[2 * m + 1], fm));
                        *reinterpret_cast<ulonglong2*>(&buf[p2 + 2 * m]) = v;
                    });
                    __builtin_amdgcn_wave_barrier();
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&buf[pt + ksm_phys(m * 128u)]);
                        nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + (long long)c * a.out_pstride + m * 128, slice_off))), v.x, v.y);
                    });
                    __builtin_amdgcn_wave_barrier();
                });
            }
        }
    }
}

}  // namespace troyn
