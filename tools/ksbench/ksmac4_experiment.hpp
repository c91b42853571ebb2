// ksmac4_experiment.hpp -- A/B experiment (tools/ksbench only, not part of the library): the digit body of ksmac2 cut in TWO
// halves of about equal vector-ALU work that run as a two-stage pipeline on the two waves of every SIMD.
//
//   waves 0-3 (FRONT): digit loads (all 32 sixteen-byte loads of the next digit requested behind the last twiddle load of round 1,
//                      i.e. a whole phase ahead: no accumulators, so 128 VGPRs are free for them), layer 0, round 0, exchange 0
//                      through buffer X0 (among the four front waves), round 1, hand-over through buffer X1
//   waves 4-7 (BACK):  round 2, multiply-accumulate (2 x 32 accumulators), epilogue (ksmac2's, crossing through X0)
// Two workgroup barriers per step (B1: X0 written / the back waves are between round 2 and the multiply-accumulate; B2: X1 written /
// consumed).  One persistent 512-thread workgroup per CU; tiles are dealt like ksmac2's workgroup orders (ksm3_decode).
// ksmac3_experiment.hpp (producers that only load + round 0, consumers with 10 layers + MAC) lost 27 % to ksmac2: a SIMD needs two
// compute-heavy waves; this variant balances the halves (front ~48, back ~40 FP64 operations per coefficient and digit).
#pragma once
#include "ksmac3_experiment.hpp"

namespace troyn {

#ifndef KSM4_KEY_AHEAD
#define KSM4_KEY_AHEAD 4
#endif
#ifndef KSM4_EARLY
#define KSM4_EARLY 16      // register pairs of the next digit requested behind round 1's last twiddle load; the others after the hand-over
#endif

template <int LOGN, bool DIGF64, bool WIDE, int EPI>
__global__ __launch_bounds__(KSM3_THREADS, 2) void ksmac4_kernel(KsMacArgs a, unsigned total_vb) {
    static_assert(LOGN == 14, "half tiles of N = 16384");
    static_assert(EPI == 0 || EPI == 1, "experiment: fused chain (1) or no diagonal digit (0)");
    constexpr unsigned N = 1u << LOGN;
    constexpr int HALVES = 1 << (LOGN - KSM_TB);
    __shared__ __attribute__((aligned(16))) u64 lds[2 * KSM_LDS_WORDS + 2];
    u64* const X0 = lds;
    u64* const X1 = lds + KSM_LDS_WORDS;
    unsigned* const flag = reinterpret_cast<unsigned*>(lds + 2 * KSM_LDS_WORDS);     // front waves that have read X0, ever

    const unsigned role = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const unsigned t = threadIdx.x & 255u, lane = t & 63u, wave = t >> 6;
    const unsigned nrows = a.row_mask ? (unsigned)__builtin_popcountll(a.row_mask) : a.L + 1;
    const unsigned nW = gridDim.x >> 3, xcd = blockIdx.x & 7u, wi = blockIdx.x >> 3;
    if (threadIdx.x == 0) *flag = 0;
    __syncthreads();

    auto at = [](const void* ubase, unsigned byte_off) { return reinterpret_cast<const char*>(ubase) + byte_off; };
    unsigned slice_off = wave * 16384u + lane * 16u;
    auto tile_of = [&](unsigned qi, unsigned& b, unsigned& k, unsigned& h, bool& end) -> unsigned {      // steps of tile qi (0: padding tile)
        const unsigned vb = ((qi * nW + wi) << 3) | xcd;
        end = vb >= total_vb;
        if (end) return 0;
        if (!ksm3_decode<HALVES>(a, nrows, vb, b, k, h)) return 0;
        return (EPI != 0 && k < a.L) ? a.L - 1 : a.L;
    };
#define KSM4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (role == 0) {
        // =========================================== FRONT ===========================================
        struct Desc { unsigned qi, step, steps, b, k, h; bool valid; };
        auto seek = [&](unsigned qi) {
            Desc d; d.step = 0; d.valid = false; d.b = d.k = d.h = 0; d.steps = 0;
            for (;; ++qi) {
                bool end;
                d.steps = tile_of(qi, d.b, d.k, d.h, end);
                if (end) break;
                if (d.steps) { d.valid = true; break; }
            }
            d.qi = qi;
            return d;
        };
        auto next = [&](Desc d) { if (d.step + 1 < d.steps) { ++d.step; return d; } return seek(d.qi + 1); };
        unsigned p0 = ksm_phys(t << 1);
        unsigned p1 = ksm_phys((t & 31u) | ((t >> 5) << 10));
        unsigned r1off = (t >> 5) * 256u;
        ulonglong2 ru[16], rv[16];
        auto issue = [&](const Desc& d, auto lo, auto hi) {
            const bool epi_row = EPI != 0 && d.k < a.L;
            const unsigned it = !epi_row ? d.step : (d.step < d.k ? d.step : d.step + 1);
            const u64* gin_u = ksm_uniform(a.digits + (long long)d.b * a.dig_bstride + (long long)it * a.dig_cstride);
            const unsigned gin_off = t << 4;
            static_for<decltype(lo)::value, decltype(hi)::value>([&](auto ic) {
                constexpr int i = decltype(ic)::value;     // i = b9 | R3<<1
                ru[i] = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
                rv[i] = ksm_gload<ulonglong2>(gin_u + 8192 + (((i & 1) << 9) + ((i >> 1) << 10)), gin_off);
            });
        };
        constexpr std::integral_constant<int, 0> I0{};
        constexpr std::integral_constant<int, KSM4_EARLY> IE{};
        constexpr std::integral_constant<int, 16> I16{};
        Desc d = seek(0);
        if (d.valid) issue(d, I0, I16);
        while (d.valid) {
            const unsigned k = d.k, h = d.h;
            const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
            const unsigned mi = a.table_start + mrow;
            const DevModulus dm = a.mods[mi];
            const F64Mod fm{dm.pd, dm.inv_pd};
            const double p = fm.p, inv_p = fm.inv_p;
            typedef const double __attribute__((address_space(4)))* cdp;
            const cdp tws = (cdp)(unsigned long long)(a.tw + (size_t)mi * N);
            const double* r1u = a.tw_r1 + ((size_t)mi * (N >> 10) + h * (KSM_THREADS >> 5)) * 32;
            auto dig_in = [&](u64 raw) -> double {
                if constexpr (DIGF64) return f64_bits_to_double(raw);
                else if constexpr (WIDE) return f64_from_u64(barrett64(raw, dm.q, dm.ratio_hi));
                else return f64_from_u64(raw);
            };
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(r1off));
            double x[32];
            // ---- layer 0 on the digit requested one phase ago ----------------------------------------------------------
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
            // ---- round 0 ----------------------------------------------------------------------------------------------
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
            // ---- exchange 0 through X0 (front waves only; every front wave read its X0 words before the previous B2) -------
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr unsigned off = ksm_phys(((i & 1) << 9) | ((i >> 1) << 10));
                *reinterpret_cast<double2*>(&X0[p0 + off]) = make_double2(f64_corr(x[2 * i], fm), f64_corr(x[2 * i + 1], fm));
            });
            double ta[8], tb[8];
            static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + 2 * q, r1off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
            static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + 8 + 2 * q, r1off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
            KSM4_BARRIER();      // B1
            static_for<0, 32>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                x[R] = f64_bits_to_double(X0[p1 + 34 * R]);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_sched_barrier(0);
            const Desc dn = next(d);
            // ---- round 1; the next digit's loads queue behind the round's last twiddle load -----------------------------
            ksm_round5<false>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r1u + 2 * q, r1off); }, [&] { if (dn.valid) issue(dn, I0, IE); }, inv_p, p);
            // ---- hand-over through X1 (the back waves read the previous digit out of it before B1) -------------------------
            static_for<0, 32>([&](auto rc) {
                constexpr int R = decltype(rc)::value;
                X1[p1 + 34 * R] = f64_double_to_bits(f64_corr(x[R], fm));
            });
            if constexpr (KSM4_EARLY < 16) { if (dn.valid) issue(dn, IE, I16); }
            KSM4_BARRIER();      // B2
            d = dn;
        }
        KSM4_BARRIER();
        KSM4_BARRIER();
    } else {
        // =========================================== BACK ===========================================
        unsigned p2 = ksm_phys(t << 5);
        unsigned pt = ksm_phys(wave * 2048u + lane * 2u);
        unsigned S = 0;      // data steps of this workgroup
        for (unsigned qi = 0;; ++qi) { unsigned b, k, h; bool end; const unsigned st = tile_of(qi, b, k, h, end); if (end) break; S += st; }
        unsigned n = 0;
        KSM4_BARRIER();      // B1 of the first global step (the front waves fill X0)
        for (unsigned qi = 0;; ++qi) {
            unsigned b, k, h; bool end;
            const unsigned steps = tile_of(qi, b, k, h, end);
            if (end) break;
            if (!steps) continue;
            const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
            const unsigned mi = a.table_start + mrow;
            const DevModulus dm = a.mods[mi];
            const F64Mod fm{dm.pd, dm.inv_pd};
            const double p = fm.p, inv_p = fm.inv_p;
            const double* r2u = a.tw_r2 + (size_t)mi * N + (size_t)h * (KSM_THREADS * 32);
            const double* kbase = a.keys + (size_t)mrow * N + (size_t)h * (KSM_THREADS * 32);
            const bool epi_row = EPI != 0 && k < a.L;
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
                asm volatile("" : "+v"(slice_off), "+v"(p2), "+v"(pt));
                double x[32];
                double ta[8], tb[8];
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + 128 * q, slice_off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
                static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + 128 * (4 + q), slice_off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
                KSM4_BARRIER();      // B2 of the global step that produced this digit
                static_for<0, 16>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    const double2 v = *reinterpret_cast<const double2*>(&X1[p2 + 2 * m]);
                    x[2 * m] = v.x; x[2 * m + 1] = v.y;
                });
                __builtin_amdgcn_sched_barrier(0);
                ksm_round5<false>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r2u + 128 * q, slice_off); }, [] {}, inv_p, p);
                const double* k0 = ksm_uniform(kbase + (long long)it * a.key_jstride);
                const double* k1 = ksm_uniform(k0 + a.key_pstride);
                constexpr int AHEAD = KSM4_KEY_AHEAD;
                double2 y0[16], y1[16];
                static_for<0, AHEAD>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    y0[m] = ksm_gload<double2>(k0 + m * 128, slice_off);
                    y1[m] = ksm_gload<double2>(k1 + m * 128, slice_off);
                });
                KSM4_BARRIER();      // B1 of the next global step (X1 is free again; the front waves start round 1 of the next digit)
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
                if ((step & 7u) == 7u)
                    static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = f64_corr(acc0[decltype(rc)::value], fm); acc1[decltype(rc)::value] = f64_corr(acc1[decltype(rc)::value], fm); });
                ++n;
            }
            // ---- epilogue (ksmac2's), crossing through this wave's slice of X0: the front waves have read the X0 words of the digit they
            // are working on (flag) and write X0 again only after B2 -----------------------------------------------------------------
            {
                const unsigned need = 4u * (n + 1u < S ? n + 1u : S);
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(1);
            }
            if (EPI == 1 && epi_row) {
                const size_t toff = (size_t)b * a.ten_bstride + (size_t)k * N + (size_t)h * (KSM_THREADS * 32);
                const u64* ta0 = ksm_uniform(a.ten_a + toff);
                const u64* tb0 = ksm_uniform(a.ten_b + toff);
                const u64* ta1 = ksm_uniform(a.ten_a + toff + a.ten_pstride);
                const u64* tb1 = ksm_uniform(a.ten_b + toff + a.ten_pstride);
                const double* dk0 = ksm_uniform(a.diag_keys + (size_t)k * 2 * N + (size_t)h * (KSM_THREADS * 32));
                const double* dk1 = ksm_uniform(dk0 + N);
                constexpr int W = KSM_TEN_WINDOW;
                ulonglong2 xa0[16], xb0[16], xa1[16], xb1[16];
                double2 y0[16], y1[16];
                auto request = [&](auto ic) {
                    constexpr int m = decltype(ic)::value;
                    xa1[m] = ksm_gload<ulonglong2>(ta1 + m * 128, slice_off);
                    xb1[m] = ksm_gload<ulonglong2>(tb1 + m * 128, slice_off);
                    y0[m] = ksm_gload<double2>(dk0 + m * 128, slice_off);
                    y1[m] = ksm_gload<double2>(dk1 + m * 128, slice_off);
                    xa0[m] = ksm_gload<ulonglong2>(ta0 + m * 128, slice_off);
                    xb0[m] = ksm_gload<ulonglong2>(tb0 + m * 128, slice_off);
                };
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, W>([&](auto ic) { request(ic); });
                __builtin_amdgcn_sched_barrier(0);
                auto cross = [&](double (&acc)[32]) {
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        *reinterpret_cast<double2*>(&X0[p2 + 2 * m]) = make_double2(f64_corr(acc[2 * m], fm), f64_corr(acc[2 * m + 1], fm));
                    });
                    __builtin_amdgcn_wave_barrier();
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        const double2 v = *reinterpret_cast<const double2*>(&X0[pt + ksm_phys(m * 128u)]);
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
                    const double a0x = f64_from_u64(xa0[m].x), a0y = f64_from_u64(xa0[m].y), b0x = f64_from_u64(xb0[m].x), b0y = f64_from_u64(xb0[m].y);
                    const double a1x = f64_from_u64(xa1[m].x), a1y = f64_from_u64(xa1[m].y), b1x = f64_from_u64(xb1[m].x), b1y = f64_from_u64(xb1[m].y);
                    double q0x = acc0[2 * m], q0y = acc0[2 * m + 1], q1x = acc1[2 * m], q1y = acc1[2 * m + 1];
                    const double dx = f64_corr(f64_mulq(f64_corr(a1x, fm), b1x, inv_p, p), fm), dy = f64_corr(f64_mulq(f64_corr(a1y, fm), b1y, inv_p, p), fm);
                    mac2(q0x, q1x, dx, y0[m].x, y1[m].x);
                    mac2(q0y, q1y, dy, y0[m].y, y1[m].y);
                    q0x += f64_mulq(a0x, b0x, inv_p, p);
                    q0y += f64_mulq(a0y, b0y, inv_p, p);
                    q1x += f64_mulq(a0x, b1x, inv_p, p) + f64_mulq(a1x, b0x, inv_p, p);
                    q1y += f64_mulq(a0y, b1y, inv_p, p) + f64_mulq(a1y, b0y, inv_p, p);
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
                        *reinterpret_cast<ulonglong2*>(&X0[p2 + 2 * m]) = v;
                    });
                    __builtin_amdgcn_wave_barrier();
                    static_for<0, 16>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&X0[pt + ksm_phys(m * 128u)]);
                        nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + (long long)c * a.out_pstride + m * 128, slice_off))), v.x, v.y);
                    });
                    __builtin_amdgcn_wave_barrier();
                });
            }
        }
        KSM4_BARRIER();      // B2 of the last global step
    }
#undef KSM4_BARRIER
}

}  // namespace troyn
