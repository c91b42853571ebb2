// ntt_kernels.hpp -- negacyclic NTT / INTT for gfx950 (CDNA4), wave64, LDS-staged.
//
// Replaces the device branches of fgk::ntt_grouped::ntt / intt (reference
// fgk/ntt_grouped.cu:159-344, :489-685: 8 radix-2 layers per launch through a 4 KiB shared
// tile, i.e. 2 full HBM round trips for N >= 8192 and a global twiddle load per butterfly per
// layer).  Design here:
//   * one workgroup owns a TILE of 2^TB coefficients of one limb-polynomial; for
//     N <= 16384 the tile is the whole limb (single pass: each coefficient is read from HBM
//     once and written once); N = 32768.. uses two passes (strided columns, then contiguous
//     chunks) because one limb (256 KiB) exceeds the 160 KiB LDS.
//   * every thread keeps E = 2^EB coefficients in VGPRs and runs EB radix-2 Harvey layers on
//     them (a radix-2^EB register block) between LDS exchanges, so a 14-layer transform makes
//     3 LDS round trips instead of 14, and twiddles are fetched once per register block.
//   * butterflies use the reference's lazy ranges ([0,4q) forward, [0,2q) inverse) and the
//     same final corrections, so stored words are bit-identical.
//   * LDS index padding (1 word per 32) keeps the power-of-two strides of the exchanges off
//     the same banks.
#pragma once
#include "dev_math.hpp"
#include <type_traits>

namespace troyn {

struct NttArgs {
    const u64* in;
    u64* out;
    const DevModulus* mods;   // [n_moduli]
    const ulonglong2* tw;     // [n_moduli][N] forward or inverse table (operand, quotient)
    long long in_bstride, in_pstride, in_cstride;     // element strides: batch, polynomial, component
    long long out_bstride, out_pstride, out_cstride;
    unsigned pcount, ncomp;
    unsigned table_start, table_count, mode, decomp;  // NTTTableIndexer (utils/ntt.h:105-124)
    unsigned reduce_input;   // 1: Barrett-reduce inputs mod the limb's modulus while loading
};

__device__ __forceinline__ unsigned ntt_table_index(const NttArgs& a, unsigned k, unsigned j) {
    unsigned idx;
    if (a.mode == 1) idx = (k == a.decomp) ? a.table_count - 1 : k;        // KeySwitchingSetProducts
    else if (a.mode == 2) idx = (j == a.decomp) ? a.table_count - 1 : j;   // KeySwitchingSkipFinals
    else idx = j;                                                           // Componentwise
    return a.table_start + idx;
}

constexpr int NTT_PAD_SHIFT = 5;
__host__ __device__ constexpr unsigned ntt_lds_words(int tb) { return (1u << tb) + ((1u << tb) >> NTT_PAD_SHIFT); }
__device__ __forceinline__ unsigned lds_phys(unsigned loc) { return loc + (loc >> NTT_PAD_SHIFT); }

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E_) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E_>(f);
    }
}

// One pass over the transform bits [LOGN-LO-G, LOGN-LO) of a 2^LOGN-point transform.
//   tile = all 2^G values of those bits x 2^C consecutive low indices, C = TB - G;
//   local index = (mid << C) | low, transform bits = local bits [C, TB).
//   forward: layers LO .. LO+G-1 (Cooley-Tukey, high bit first)
//   inverse: the matching Gentleman-Sande layers, low bit first
// Round r keeps local bits [S, S+EB) in registers and runs the butterflies of the transform
// bits [BLO, BHI] that fall inside that window.
// FIRST/LAST mark the first/last pass of the whole transform (prologue / final correction).
template <int LOGN, int LO, int G, int TB, int EB, bool INV, bool FIRST, bool LAST>
__global__ __launch_bounds__(1 << (TB - EB)) void ntt_pass_kernel(NttArgs a) {
    constexpr int C = TB - G;
    constexpr int E = 1 << EB;
    constexpr unsigned N = 1u << LOGN;
    constexpr int TILE_BITS = LOGN - TB;               // tiles per limb-polynomial = 2^TILE_BITS
    constexpr int NLB = LOGN - LO - G - C;             // low-block bits in the tile id
    static_assert(G >= 1 && C >= 0 && NLB >= 0 && TB >= EB, "bad NTT pass shape");
    constexpr int ROUNDS = (G + EB - 1) / EB;

    __shared__ u64 lds[ROUNDS > 1 ? ntt_lds_words(TB) : 1];

    const unsigned t = threadIdx.x;
    unsigned bid = blockIdx.x;
    const unsigned tile = bid & ((1u << TILE_BITS) - 1); bid >>= TILE_BITS;
    const unsigned j = bid % a.ncomp; bid /= a.ncomp;
    const unsigned k = bid % a.pcount;
    const unsigned b = bid / a.pcount;
    const unsigned top = tile >> NLB;
    const unsigned lb = tile & ((1u << NLB) - 1);

    const unsigned mi = ntt_table_index(a, k, j);
    const DevModulus md = a.mods[mi];
    const u64 q = md.q, two_q = md.q << 1;
    const ulonglong2* __restrict__ tw = a.tw + (size_t)mi * N;
    const u64* __restrict__ gin = a.in + (long long)b * a.in_bstride + (long long)k * a.in_pstride + (long long)j * a.in_cstride;
    u64* __restrict__ gout = a.out + (long long)b * a.out_bstride + (long long)k * a.out_pstride + (long long)j * a.out_cstride;

    auto gindex = [&](unsigned loc) -> unsigned {
        return (top << (LOGN - LO)) | ((loc >> C) << (LOGN - LO - G)) | (lb << C) | (loc & ((1u << C) - 1));
    };

    u64 x[E];

    static_for<0, ROUNDS>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        // transform bits handled this round, and the register window [S, S+EB)
        constexpr int BHI = INV ? ((C + (r + 1) * EB - 1 < TB - 1) ? C + (r + 1) * EB - 1 : TB - 1) : TB - 1 - r * EB;
        constexpr int BLO = INV ? C + r * EB : ((TB - (r + 1) * EB > C) ? TB - (r + 1) * EB : C);
        constexpr int S = INV ? ((BLO < TB - EB) ? BLO : TB - EB) : ((TB - (r + 1) * EB > 0) ? TB - (r + 1) * EB : 0);
        static_assert(S >= 0 && S + EB <= TB && BLO >= S && BHI < S + EB && BLO <= BHI, "bad round window");
        const unsigned tlow = t & ((1u << S) - 1);
        const unsigned locbase = tlow | ((t >> S) << (S + EB));

        if constexpr (r == 0) {
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                u64 v = gin[gindex(locbase | ((unsigned)R << S))];
                if (FIRST && a.reduce_input) v = barrett64(v, q, md.ratio_hi);
                x[R] = v;
            });
        } else {
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                x[R] = lds[lds_phys(locbase | ((unsigned)R << S))];
            });
        }

        constexpr int NLAYERS = BHI - BLO + 1;
        static_for<0, NLAYERS>([&](auto lc) {
            // forward: highest bit first; inverse: lowest bit first
            constexpr int bit = INV ? BLO + decltype(lc)::value : BHI - decltype(lc)::value;
            constexpr int rb = bit - S;            // register bit
            constexpr int kk = TB - 1 - bit;       // layer inside the tile
            constexpr int l = LO + kk;             // global (forward-numbered) layer of this bit
            static_for<0, (E >> (rb + 1))>([&](auto hc) {
                constexpr int hi = decltype(hc)::value;
                const unsigned loc0 = locbase | ((unsigned)(hi << (rb + 1)) << S);
                const unsigned grp = (top << kk) + (loc0 >> (bit + 1));
                const ulonglong2 w = INV ? tw[N - (2u << l) + 1 + grp] : tw[(1u << l) + grp];
                static_for<0, (1 << rb)>([&](auto oc) {
                    constexpr int R0 = (hi << (rb + 1)) | decltype(oc)::value;
                    constexpr int R1 = R0 | (1 << rb);
                    if constexpr (!INV) {
                        u64 u = x[R0];
                        u = u >= two_q ? u - two_q : u;
                        const u64 v = shoup_lazy(x[R1], w.x, w.y, q);
                        x[R0] = u + v;
                        x[R1] = u + two_q - v;
                    } else {
                        const u64 u = x[R0], v = x[R1];
                        const u64 s = u + v;
                        x[R0] = s >= two_q ? s - two_q : s;
                        x[R1] = shoup_lazy(u + two_q - v, w.x, w.y, q);
                    }
                });
            });
        });

        if constexpr (r == ROUNDS - 1) {
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                u64 v = x[R];
                if constexpr (LAST) {
                    v = v >= two_q ? v - two_q : v;
                    v = v >= q ? v - q : v;
                    if constexpr (INV) v = shoup_lazy(v, md.inv_n_op, md.inv_n_quo, q);
                }
                gout[gindex(locbase | ((unsigned)R << S))] = v;
            });
        } else {
            if constexpr (r > 0) __syncthreads();   // all reads of the previous exchange are done
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                lds[lds_phys(locbase | ((unsigned)R << S))] = x[R];
            });
            __syncthreads();
        }
    });
}

// Generic fallback for any 2 <= N: one workgroup per limb-polynomial, radix-2 layer by layer.
// N <= 4096 runs out of LDS; larger N (not covered by an optimised instantiation) works in
// place in global memory.  Used for small test rings (the reference's tests use N = 32).
__global__ __launch_bounds__(256) void ntt_generic_kernel(NttArgs a, unsigned log_n, int inverse) {
    __shared__ u64 lds[4096];
    const unsigned n = 1u << log_n;
    unsigned bid = blockIdx.x;
    const unsigned j = bid % a.ncomp; bid /= a.ncomp;
    const unsigned k = bid % a.pcount;
    const unsigned b = bid / a.pcount;
    const unsigned mi = ntt_table_index(a, k, j);
    const DevModulus md = a.mods[mi];
    const u64 q = md.q, two_q = md.q << 1;
    const ulonglong2* __restrict__ tw = a.tw + (size_t)mi * n;
    const u64* gin = a.in + (long long)b * a.in_bstride + (long long)k * a.in_pstride + (long long)j * a.in_cstride;
    u64* gout = a.out + (long long)b * a.out_bstride + (long long)k * a.out_pstride + (long long)j * a.out_cstride;
    const bool use_lds = n <= 4096;
    u64* work = use_lds ? lds : gout;
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        u64 v = gin[i];
        if (a.reduce_input) v = barrett64(v, q, md.ratio_hi);
        work[i] = v;
    }
    __syncthreads();
    const unsigned half = n >> 1;
    for (unsigned layer = 0; layer < log_n; ++layer) {
        const unsigned gap_power = inverse ? layer : (log_n - layer - 1);
        const unsigned gap = 1u << gap_power;
        const unsigned m = inverse ? (n >> (layer + 1)) : (1u << layer);
        for (unsigned i = threadIdx.x; i < half; i += blockDim.x) {
            const unsigned grp = i >> gap_power;
            const unsigned xi = (grp << (gap_power + 1)) + (i & (gap - 1));
            const unsigned yi = xi + gap;
            if (!inverse) {
                const ulonglong2 w = tw[m + grp];
                u64 u = work[xi];
                u = u >= two_q ? u - two_q : u;
                const u64 v = shoup_lazy(work[yi], w.x, w.y, q);
                work[xi] = u + v;
                work[yi] = u + two_q - v;
            } else {
                const ulonglong2 w = tw[n - (m << 1) + 1 + grp];
                const u64 u = work[xi], v = work[yi];
                const u64 s = u + v;
                work[xi] = s >= two_q ? s - two_q : s;
                work[yi] = shoup_lazy(u + two_q - v, w.x, w.y, q);
            }
        }
        __syncthreads();
    }
    for (unsigned i = threadIdx.x; i < n; i += blockDim.x) {
        u64 v = work[i];
        v = v >= two_q ? v - two_q : v;
        v = v >= q ? v - q : v;
        if (inverse) v = shoup_lazy(v, md.inv_n_op, md.inv_n_quo, q);
        gout[i] = v;
    }
}

}  // namespace troyn
