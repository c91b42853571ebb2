// behz2_lift_pass1.hpp -- BEHZ steps (1)-(3) of one operand of a BFV multiply at N = 32768 as ONE launch (round 5).
//
// At the two-pass ring sizes the operand used to make three trips through HBM before tensor_core_kernel: the strided first forward pass
// over its L rows of base q (read L, write L), behz2_lift_kernel (read L, write NB + 1) and the strided first pass over the lifted rows
// (read NB + 1, write NB + 1).  The lift works on one coefficient of all rows, the strided pass on the 8 coefficients {base + k N/8} of one
// row.  Here a 256-thread workgroup owns 32 consecutive bases x 8 strides:
//   phase A: thread (k, j) lifts coefficient base_j + k N/8 exactly as behz2_lift_kernel does (20 registers of split residues) and parks the
//            re-centred doubles of its L input words and NB + 1 lifted words in LDS ([row][k][j], 2 KB per row);
//   phase B: the (L + NB + 1) x 32 octets are dealt to the threads; each runs layers 0-2 on its 8 words and stores the first-pass words.
// Giving a thread the 8 strided coefficients of all rows instead (no LDS) needs 16 L registers of split residues next to the partial sums:
// measured at the compiler, 255 registers and 1.4 KB of scratch per lane at L = 10.
// HBM traffic per operand: L + (L + NB + 1) rows instead of 3 L + 3 (NB + 1); same words at the same addresses as the three launches
// (tensor_core_kernel reads them unchanged); every global access is a run of 32 consecutive words (256 bytes).
//
// Values: the lift is behz2_lift_kernel's (fgk/rns_tool.cu:7-100 kernel_fast_b_conv_m_tilde_sm_mrq, utils/rns_tool.cu:762-790, :870-905),
// the butterflies are ntt_pass_body<ArithF64, 15, 0, 3, 12, 4, false, true, false> (fgk/ntt_grouped.cu layers 0-2): canonical word ->
// re-centred double -> three Cooley-Tukey layers -> raw double bits.  FP64 policy only: every q_i and every auxiliary prime below 2^50.
#pragma once
#include "behz2_kernels.hpp"
#include "dev_math_f64.hpp"
#include "ntt_kernels.hpp"

namespace troyn {

struct LiftPass1Args {
    const u64* in;              // [items][L][N] coefficient form, base q
    u64* out_q;                 // [items][L][N] first-pass words of base q
    u64* out_bsk;               // [items][NB+1][N] first-pass words of the lifted rows
    const double* tw_q;         // forward twiddles of the plan of base q  [L][N]
    const double* tw_aux;       // forward twiddles of the auxiliary plan   [NB+1][N]
    const DevModulus* q_mods;   // [L]
    const DevModulus* aux_mods; // [NB+1]
};

constexpr unsigned LIFT_PASS1_MAX_ROWS = 31;      // 2 KB of LDS per row, 64 KB of dynamic LDS without an attribute

template <int L>
__global__ __launch_bounds__(256) void behz2_lift_pass1_kernel(Behz2Dev c, LiftPass1Args a) {
    constexpr int SHQ = 25, GROUP = 64;
    constexpr unsigned LOGN = 15, N = 1u << LOGN, SEG = N / 8, CHUNKS = SEG / 32;
    extern __shared__ u64 lift_lds[];              // [L + NB + 1][8][32] re-centred doubles
    const unsigned NB = c.NB, t = threadIdx.x;
    const size_t item = blockIdx.x / CHUNKS;
    const unsigned base0 = (blockIdx.x % CHUNKS) * 32u;
    {
        // ---- phase A: one coefficient of all rows ----
        const unsigned x = (t >> 5) * SEG + base0 + (t & 31u);
        const u64* ip = a.in + item * (size_t)L * N;
        const cmodp q_mods = as_cmod(c.q_mods);
        const cu64x2p scale = as_c128(c.q_mt_inv_punc);
        const cu32p mtrow = as_c32(c.lift_mt), rows = as_c32(c.lift_rows);
        const cu64p rcs = as_c64(c.lift_rc);
        const unsigned Lp = c.rs >> 1;
        u32 ylo[L], yhi[L];
        u32 r_mt = 0;
#pragma unroll
        for (int i = 0; i < L; ++i) {
            const u64 xv = __builtin_nontemporal_load(ip + (size_t)i * N + x);
            const ulonglong2 f = ld_pair(scale, i);
            const u64 y = shoup_mul(xv, f.x, f.y, q_mods[i].q);
            ylo[i] = (u32)y & ((1u << SHQ) - 1);
            yhi[i] = (u32)(y >> SHQ);
            r_mt += (u32)y * mtrow[i];
            lift_lds[i * 256 + t] = f64_double_to_bits(f64_corr(f64_from_u64(xv), F64Mod{q_mods[i].pd, q_mods[i].inv_pd}));
        }
        const bool neg = r_mt >= 0x80000000u;
        const cmodp am = as_cmod(a.aux_mods);
#pragma unroll 1
        for (unsigned b = 0; b <= NB; ++b) {
            const cu32p row = rows + (size_t)b * c.rs;
            const cu64p rc = rcs + (size_t)b * BEHZ2_RC;
            u128 v = behz2_dot<L, SHQ, 32, GROUP>(ylo, yhi, row, row + Lp);
            v += (u128)rc[B2_C0] * r_mt + (neg ? rc[B2_C1] : 0ull);
            const u64 w = behz2_reduce(v, rc[B2_P], rc[B2_RLO], rc[B2_RHI]);
            lift_lds[(L + b) * 256 + t] = f64_double_to_bits(f64_corr(f64_from_u64(w), F64Mod{am[b].pd, am[b].inv_pd}));
        }
    }
    __syncthreads();
    // ---- phase B: layers 0-2 on the octets; a wave covers two rows, so the modulus constants and twiddles are per-lane loads ----
    const unsigned nrows = L + NB + 1;
    u64* oq = a.out_q + item * (size_t)L * N;
    u64* ob = a.out_bsk + item * (size_t)(NB + 1) * N;
    for (unsigned w = t; w < nrows * 32u; w += 256u) {
        const unsigned row = w >> 5, j = w & 31u;
        const bool isq = row < (unsigned)L;
        const unsigned r = isq ? row : row - L;
        const DevModulus* md = (isq ? a.q_mods : a.aux_mods) + r;
        const double* tw = (isq ? a.tw_q : a.tw_aux) + (size_t)r * N;
        u64* out = (isq ? oq : ob) + (size_t)r * N + base0 + j;
        const double p = md->pd, inv_p = md->inv_pd;
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = f64_bits_to_double(lift_lds[row * 256 + k * 32 + j]);
        auto bf = [&](int r0, int r1, double tw_w) {
            const double rr = f64_mulq(x[r1], tw_w, inv_p, p);
            const double u = x[r0];
            x[r0] = u + rr; x[r1] = u - rr;
        };
        const double w1 = tw[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) bf(k, k + 4, w1);
        const double w2 = tw[2], w3 = tw[3];
        bf(0, 2, w2); bf(1, 3, w2); bf(4, 6, w3); bf(5, 7, w3);
        const double w4 = tw[4], w5 = tw[5], w6 = tw[6], w7 = tw[7];
        bf(0, 1, w4); bf(2, 3, w5); bf(4, 5, w6); bf(6, 7, w7);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(f64_double_to_bits(x[k]), out + k * SEG);
    }
}

}  // namespace troyn
