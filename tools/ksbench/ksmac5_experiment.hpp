// ksmac5_experiment.hpp -- A/B experiment (tools/ksbench only): the key-switch inner product with 16 coefficients per thread and FOUR waves per
// SIMD.  Same tiles as ksmac2 (2^13 outputs of one (item, row), half tiles at N = 16384 with layer 0 applied while loading), but 512 threads per
// workgroup, two workgroups per CU, 128 VGPRs per thread: 2 x 16 accumulators + 16 coefficients.  Four register rounds (3 | 4 | 3 | 3 layers):
// exchange 0 crosses waves (two workgroup barriers), exchanges 1 and 2 stay inside a wave (a wave owns a contiguous block of 1024 indices after
// exchange 0), and the twiddles of round 1 are wave-uniform (scalar loads).  The question it answers: do twice as many, half as large waves hide
// what ksmac2's two 256-register waves per SIMD cannot?  (Round 1's kernel had this shape at 1024 threads and spilled 28 registers.)
// Plain form only (no diagonal digit, no fused epilogue): compared with ksmac2's NODIAG instantiation on double digits.
#pragma once
#include "../../troy-nova_amd/csrc/ksmac_kernels.hpp"

namespace troyn {

constexpr int KSM5_THREADS = 512;
#ifndef KSM5_LOADW
#define KSM5_LOADW 4
#endif
#ifndef KSM5_AHEAD
#define KSM5_AHEAD 2
#endif
__host__ __device__ constexpr unsigned ksm5_phys(unsigned w) { return w + 2u * (w >> 5); }

// prepared keys for the 16-coefficient layout: blocks of 1024 words = 64 lanes x 16 registers stored as [m = reg/2][lane][reg%2]
__host__ __device__ constexpr unsigned ksm5_perm(unsigned i) {
    return (i & ~1023u) | ((((i >> 1) & 7u) * 64u + ((i >> 4) & 63u)) * 2u) | (i & 1u);
}
static __global__ __launch_bounds__(256) void ksm5_prepare_keys_kernel(KeyPtrs keys, unsigned L, unsigned rows_per_key, unsigned n, double* out) {
    const size_t pairs_per_key = (size_t)rows_per_key * (n / 2), total = (size_t)L * pairs_per_key;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned j = (unsigned)(p / pairs_per_key);
        const size_t q = p % pairs_per_key, row = q / (n / 2);
        const unsigned i = (unsigned)(q % (n / 2)) * 2u;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(keys.p[j] + row * n + i);
        *reinterpret_cast<double2*>(out + ((size_t)j * rows_per_key + row) * n + ksm5_perm(i)) = make_double2(f64_from_u64(v.x), f64_from_u64(v.y));
    }
}

template <int LOGN>
__global__ __launch_bounds__(KSM5_THREADS, 4) void ksmac5_kernel(KsMacArgs a) {
    static_assert(LOGN == 14, "experiment: N = 16384 half tiles");
    constexpr unsigned N = 1u << LOGN;
    constexpr int HALVES = 2;
    __shared__ __attribute__((aligned(16))) u64 lds[KSM_LDS_WORDS];
    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    // workgroup -> (item, row, half): item-major on one XCD (KsMacArgs::grouped == 1) or plain
    unsigned b, k, h;
    {
        const unsigned G = (a.L + 1) * HALVES;
        unsigned g;
        if (a.grouped) { const unsigned per = 8u * G, r = blockIdx.x % per; g = r / 8u; b = (blockIdx.x / per) * 8u + (r % 8u); }
        else { g = blockIdx.x % G; b = blockIdx.x / G; }
        k = g / HALVES; h = g % HALVES;
    }
    const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
    const unsigned mi = a.table_start + mrow;
    const DevModulus dm = a.mods[mi];
    const F64Mod fm{dm.pd, dm.inv_pd};
    const double p = fm.p, inv_p = fm.inv_p;
    typedef const double __attribute__((address_space(4)))* cdp;
    const cdp tws = (cdp)(unsigned long long)(a.tw + (size_t)mi * N);
    const unsigned wave_u = (unsigned)__builtin_amdgcn_readfirstlane((int)wave);

    double acc0[16], acc1[16];
    static_for<0, 16>([&](auto rc) { acc0[decltype(rc)::value] = 0.0; acc1[decltype(rc)::value] = 0.0; });
    auto mac2 = [&](double& a0, double& a1, double v, double y0, double y1) {
        const double h0 = v * y0, h1 = v * y1;
        const double l0 = __builtin_fma(v, y0, -h0), l1 = __builtin_fma(v, y1, -h1);
        const double q0 = __builtin_rint(h0 * inv_p), q1 = __builtin_rint(h1 * inv_p);
        a0 += __builtin_fma(-q0, p, h0) + l0;
        a1 += __builtin_fma(-q1, p, h1) + l1;
    };
    // twiddle of the butterfly at tile bit BETA whose higher tile bits (within the tile) are `hi`
    auto tw_index = [&](int beta, unsigned hi) { return (N >> (beta + 1)) + (h << (12 - beta)) + hi; };

    // LDS positions (padded words)
    unsigned p0 = ksm5_phys(t << 1);                                   // exchange 0 writes: + phys(R3 << 10), 16-byte pairs
    unsigned p1 = ksm5_phys(lane | (wave << 10));                      // round-1 layout: + phys(R << 6)
    unsigned p2 = ksm5_phys((lane & 3u) | ((lane >> 2) << 6) | (wave << 10));     // round-2 layout: + phys(R << 2)
    unsigned p3 = ksm5_phys((lane << 4) | (wave << 10));               // round-3 layout: + R
    unsigned pc = ksm5_phys((lane << 1) | (wave << 10));               // coalesced pairs: + phys(m << 7)
    const double* kbase = a.keys + (size_t)mrow * N + (size_t)h * 8192u;
    const u64* dig_item = a.digits + (long long)b * a.dig_bstride;
    unsigned slice_off = wave * 8192u + lane * 16u;                    // bytes inside the tile: wave's 1024 words, 16 bytes per lane

    for (unsigned it = 0; it < a.L; ++it) {
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(slice_off));
        double x[16];
        // ---- load + layer 0: registers r = b0 | R3 << 1, tile index b0 | t << 1 | R3 << 10 -----------------------------------------
        {
            const u64* gin_u = ksm_uniform(dig_item + (long long)it * a.dig_cstride);
            const unsigned gin_off = t << 4;
            const double w1 = tws[1];
            const double sgn = h ? -1.0 : 1.0;
            ulonglong2 ru[8], rv[8];
            auto request = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                ru[i] = ksm_gload<ulonglong2>(gin_u + (i << 10), gin_off);
                rv[i] = ksm_gload<ulonglong2>(gin_u + 8192 + (i << 10), gin_off);
            };
            constexpr int W0 = KSM5_LOADW;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, W0>([&](auto ic) { request(ic); });
            static_for<0, 8>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                const double u0 = f64_bits_to_double(ru[i].x), u1 = f64_bits_to_double(ru[i].y), v0 = f64_bits_to_double(rv[i].x), v1 = f64_bits_to_double(rv[i].y);
                x[2 * i] = f64_corr(__builtin_fma(sgn, f64_mulq(v0, w1, inv_p, p), u0), fm);
                x[2 * i + 1] = f64_corr(__builtin_fma(sgn, f64_mulq(v1, w1, inv_p, p), u1), fm);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i + W0 < 8) request(std::integral_constant<int, i + W0>{});
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 0: tile bits 12, 11, 10 = register bits 3, 2, 1 (workgroup-uniform twiddles) ----------------------------------------
        static_for<0, 3>([&](auto lc) {
            constexpr int li = decltype(lc)::value;
            constexpr int bit = 12 - li, rb = 3 - li;
            static_for<0, (1 << li)>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const double w = tws[tw_index(bit, g)];
                static_for<0, (1 << rb)>([&](auto oc) {
                    constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                    const double r = f64_mulq(x[R1], w, inv_p, p);
                    const double u = x[R0];
                    x[R0] = u + r; x[R1] = u - r;
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 0 (workgroup-wide) -------------------------------------------------------------------------------------------------
        __syncthreads();
        static_for<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            *reinterpret_cast<double2*>(&lds[p0 + ksm5_phys((unsigned)i << 10)]) = make_double2(f64_corr(x[2 * i], fm), f64_corr(x[2 * i + 1], fm));
        });
        __syncthreads();
        static_for<0, 16>([&](auto rc) { constexpr int R = decltype(rc)::value; x[R] = f64_bits_to_double(lds[p1 + ksm5_phys((unsigned)R << 6)]); });
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 1: tile bits 9, 8, 7, 6 = register bits 3..0; the higher bits are the wave id: scalar twiddles ------------------------
        static_for<0, 4>([&](auto lc) {
            constexpr int li = decltype(lc)::value;
            constexpr int bit = 9 - li, rb = 3 - li;
            static_for<0, (1 << li)>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const double w = tws[tw_index(bit, (wave_u << li) | g)];
                static_for<0, (1 << rb)>([&](auto oc) {
                    constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                    const double r = f64_mulq(x[R1], w, inv_p, p);
                    const double u = x[R0];
                    x[R0] = u + r; x[R1] = u - r;
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 1 (inside the wave's block of 1024): registers become tile bits 5, 4, 3, 2 -------------------------------------------
        static_for<0, 16>([&](auto rc) { constexpr int R = decltype(rc)::value; lds[p1 + ksm5_phys((unsigned)R << 6)] = f64_double_to_bits(f64_corr(x[R], fm)); });
        __builtin_amdgcn_wave_barrier();
        static_for<0, 16>([&](auto rc) { constexpr int R = decltype(rc)::value; x[R] = f64_bits_to_double(lds[p2 + ksm5_phys((unsigned)R << 2)]); });
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 2: tile bits 5, 4, 3 = register bits 3, 2, 1; the twiddles depend on tile bits 6..12 = (wave, lane >> 2): a vector of 8 slots per
        // group of four lanes (slot 1 | 2, 3 | 4..7), KsMacArgs::tw_r1 here = [modulus][half][wave][lane >> 2][8] ------------------------------
        {
            const double* r2u = a.tw_r1 + ((size_t)(mi * 2 + h) * 8 + wave_u) * 128;
            const unsigned r2off = (lane >> 2) * 64u;
            double tq[8];
            static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + 2 * q, r2off); tq[2 * q] = v.x; tq[2 * q + 1] = v.y; });
            static_for<0, 3>([&](auto lc) {
                constexpr int li = decltype(lc)::value;
                constexpr int rb = 3 - li;
                static_for<0, (1 << li)>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const double w = tq[(1 << li) + g];
                    static_for<0, (1 << rb)>([&](auto oc) {
                        constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                        const double r = f64_mulq(x[R1], w, inv_p, p);
                        const double u = x[R0];
                        x[R0] = u + r; x[R1] = u - r;
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- exchange 2 (inside the wave): registers become tile bits 3, 2, 1, 0 -----------------------------------------------------------
        static_for<0, 16>([&](auto rc) { constexpr int R = decltype(rc)::value; lds[p2 + ksm5_phys((unsigned)R << 2)] = f64_double_to_bits(f64_corr(x[R], fm)); });
        __builtin_amdgcn_wave_barrier();
        static_for<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const double2 v = *reinterpret_cast<const double2*>(&lds[p3 + 2 * m]);
            x[2 * m] = v.x; x[2 * m + 1] = v.y;
        });
        __builtin_amdgcn_sched_barrier(0);
        // ---- round 3: tile bits 2, 1, 0 = register bits 2, 1, 0 (register bit 3 = tile bit 3 is above them); 14 per-lane twiddles in the
        // lane-interleaved table KsMacArgs::tw_r2 here = [modulus][half][wave][q][lane][2], slots 0, 1 | 2..5 | 6..13, fetched layer by layer ------
        {
            const double* r3u = a.tw_r2 + ((size_t)(mi * 2 + h) * 8 + wave_u) * 1024;
            const unsigned r3off = lane * 16u;
            static_for<0, 3>([&](auto lc) {
                constexpr int li = decltype(lc)::value;
                constexpr int rb = 2 - li;
                constexpr int NG = 2 << li, S0 = NG - 2;            // groups of this layer, first slot
                double tq[NG];
                static_for<0, NG / 2>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r3u + (S0 / 2 + q) * 128, r3off); tq[2 * q] = v.x; tq[2 * q + 1] = v.y; });
                static_for<0, NG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const double w = tq[g];
                    static_for<0, (1 << rb)>([&](auto oc) {
                        constexpr int R0 = (g << (rb + 1)) | decltype(oc)::value, R1 = R0 | (1 << rb);
                        const double r = f64_mulq(x[R1], w, inv_p, p);
                        const double u = x[R0];
                        x[R0] = u + r; x[R1] = u - r;
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // ---- multiply-accumulate: 16 consecutive coefficients per thread, prepared keys [m][lane][2] --------------------------------------
        {
            const double* k0 = ksm_uniform(kbase + (long long)it * a.key_jstride);
            const double* k1 = ksm_uniform(k0 + a.key_pstride);
            constexpr int AHEAD = KSM5_AHEAD;
            double2 y0[8], y1[8];
            static_for<0, AHEAD>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                y0[m] = ksm_gload<double2>(k0 + m * 128, slice_off);
                y1[m] = ksm_gload<double2>(k1 + m * 128, slice_off);
            });
            static_for<0, 8>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m + AHEAD < 8) {
                    y0[m + AHEAD] = ksm_gload<double2>(k0 + (m + AHEAD) * 128, slice_off);
                    y1[m + AHEAD] = ksm_gload<double2>(k1 + (m + AHEAD) * 128, slice_off);
                }
                const double v0 = f64_corr(x[2 * m], fm), v1 = f64_corr(x[2 * m + 1], fm);
                mac2(acc0[2 * m], acc1[2 * m], v0, y0[m].x, y1[m].x);
                mac2(acc0[2 * m + 1], acc1[2 * m + 1], v1, y0[m].y, y1[m].y);
            });
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((it & 7u) == 7u)
            static_for<0, 16>([&](auto rc) { acc0[decltype(rc)::value] = f64_corr(acc0[decltype(rc)::value], fm); acc1[decltype(rc)::value] = f64_corr(acc1[decltype(rc)::value], fm); });
    }
    // ---- canonical results through the wave's block, 16-byte coalesced stores ---------------------------------------------------------------
    u64* go = a.out + (long long)b * a.out_bstride + (long long)k * a.out_cstride + (size_t)h * 8192u;
    static_for<0, 2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            *reinterpret_cast<ulonglong2*>(&lds[p3 + 2 * m]) = make_ulonglong2(f64_canon(c ? acc1[2 * m] : acc0[2 * m], fm), f64_canon(c ? acc1[2 * m + 1] : acc0[2 * m + 1], fm));
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[pc + ksm5_phys((unsigned)m << 7)]);
            nt_store2(reinterpret_cast<u64*>(reinterpret_cast<char*>(go + (long long)c * a.out_pstride + m * 128) + slice_off), v.x, v.y);
        });
        __builtin_amdgcn_wave_barrier();
    });
}

}  // namespace troyn
