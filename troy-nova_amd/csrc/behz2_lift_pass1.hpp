// behz2_lift_pass1.hpp -- N = 32768 BFV multiply: the BEHZ conversions fused with the strided transform pass next to them (round 5).
// (1) BEHZ steps (1)-(3) of one operand as ONE launch; (2) below: the last inverse pass of the product + the floor as one launch.
//
// At the two-pass ring sizes the operand used to make three trips through HBM before tensor_core_kernel: the strided first forward pass
// over its L rows of base q (read L, write L), behz2_lift_kernel (read L, write NB + 1) and the strided first pass over the lifted rows
// (read NB + 1, write NB + 1).  The lift works on one coefficient of all rows, the strided pass on the 8 coefficients {base + k N/8} of one
// row.  Here a 256-thread workgroup owns 32 consecutive bases x 8 strides:
//   thread (k, j) reads coefficient base_j + k N/8 of the L input rows and parks their re-centred doubles in LDS ([row][k][j], 2 KB per row);
//   the L x 32 octets are dealt to the threads, each runs layers 0-2 on its 8 words and stores the first-pass words of base q;
//   thread (k, j) lifts its coefficient exactly as behz2_lift_kernel does (behz2_lift_one, 2 L registers of split residues) into the same LDS rows;
//   the (NB + 1) x 32 octets of the lifted rows follow.  One base in LDS at a time: max(L, NB + 1) rows, 26 KB at cfg4 (both bases together: 45 KB and
//   half the workgroups per CU, -1 %).
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

template <int L, int T = BEHZ2_FUSED_THREADS>
__global__ __launch_bounds__(T) void behz2_lift_pass1_kernel(Behz2Dev c, LiftPass1Args a) {
    constexpr unsigned LOGN = 15, N = 1u << LOGN, SEG = N / 8, BASES = T / 8, CHUNKS = SEG / BASES;
    // LDS holds ONE base at a time ([rows][8][BASES] re-centred doubles): the rows of base q leave before the lifted rows arrive, so a workgroup
    // needs max(L, NB + 1) rows instead of L + NB + 1 and twice as many workgroups share a CU (two more barriers)
    extern __shared__ u64 lift_lds[];
    const unsigned NB = c.NB, t = threadIdx.x;
    const size_t item = blockIdx.x / CHUNKS;
    const unsigned base0 = (blockIdx.x % CHUNKS) * BASES;
    // layers 0-2 on the octets of `nrows` rows in LDS; a wave covers several rows, so the modulus constants and twiddles are per-lane loads
    auto octets = [&](unsigned nrows, const DevModulus* mods, const double* twt, u64* outp) {
        for (unsigned w = t; w < nrows * BASES; w += T) {
            const unsigned row = w / BASES, j = w % BASES;
            const DevModulus* md = mods + row;
            const double* tw = twt + (size_t)row * N;
            u64* out = outp + (size_t)row * N + base0 + j;
            const double p = md->pd, inv_p = md->inv_pd;
            double x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) x[k] = f64_bits_to_double(lift_lds[row * T + k * BASES + j]);
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
    };
    // ---- one coefficient of all rows (behz2_lift_one: the same arithmetic as behz2_lift_kernel) ----
    const unsigned x = (t / BASES) * SEG + base0 + (t % BASES);
    const u64* ip = a.in + item * (size_t)L * N;
    const cmodp qm = as_cmod(a.q_mods);
    u64 xv[L];
#pragma unroll
    for (int i = 0; i < L; ++i) {
        xv[i] = __builtin_nontemporal_load(ip + (size_t)i * N + x);
        lift_lds[i * T + t] = f64_double_to_bits(f64_corr(f64_from_u64(xv[i]), F64Mod{qm[i].pd, qm[i].inv_pd}));
    }
    __syncthreads();
    octets(L, a.q_mods, a.tw_q, a.out_q + item * (size_t)L * N);
    __syncthreads();
    behz2_lift_one<L, true, true>(c, [&](int i) { return xv[i]; }, [&](unsigned b, double w) { lift_lds[b * T + t] = f64_double_to_bits(w); });      // already re-centred (behz2_reduce_f64)
    __syncthreads();
    octets(NB + 1, a.aux_mods, a.tw_aux, a.out_bsk + item * (size_t)(NB + 1) * N);
}

// ---- BEHZ steps (5)-(8) tail: last inverse pass of both bases + kernel_fast_floor_fast_b_conv_sk as ONE launch ----
// The product leaves tensor_core_kernel after the first inverse pass (12 layers inside 4096-word blocks).  The last pass (layers 2, 1, 0:
// octets {base + k N/8}, ntt_pass_body<ArithF64, 15, 0, 3, 12, 4, true, false, true>) used to write the L + NB + 1 rows in coefficient form
// and behz2_floor_kernel read them back.  Same workgroup shape as above, phases swapped:
//   phase A: the (L + NB + 1) x 32 octets are dealt to the threads: load_mid, three Gentleman-Sande layers (the last one folded with N^-1),
//            N^-1 on the sum outputs, canonical words into LDS ([row][k][j]);
//   phase B: thread (k, j) runs behz2_floor_one on coefficient base_j + k N/8 with its residues read from LDS and stores the L output words.
// The coefficient-form product never reaches HBM: per product polynomial L + NB + 1 rows read and L written instead of 3 (L + NB + 1) + L.
struct FloorPass2Args {
    const u64* in_q;            // [items][L][N] words after the first inverse pass, base q
    const u64* in_bsk;          // [items][NB+1][N] the same for the auxiliary rows
    u64* out;                   // [items][L][N] coefficient form, base q
    const double* tw_q;         // inverse twiddles of the plan of base q  [L][N]
    const double* tw_aux;       // inverse twiddles of the auxiliary plan   [NB+1][N]
    const DevModulus* q_mods;   // [L]
    const DevModulus* aux_mods; // [NB+1]
};

template <int L, int T = BEHZ2_FUSED_THREADS>
__global__ __launch_bounds__(T) void behz2_floor_pass2_kernel(Behz2Dev c, FloorPass2Args a) {
    constexpr unsigned LOGN = 15, N = 1u << LOGN, SEG = N / 8, BASES = T / 8, CHUNKS = SEG / BASES;
    extern __shared__ u64 lift_lds[];              // [L + NB + 1][8][32] canonical words
    const unsigned NB = c.NB, t = threadIdx.x;
    const size_t item = blockIdx.x / CHUNKS;
    const unsigned base0 = (blockIdx.x % CHUNKS) * BASES;
    const unsigned nrows = L + NB + 1;
    const u64* iq = a.in_q + item * (size_t)L * N;
    const u64* ib = a.in_bsk + item * (size_t)(NB + 1) * N;
    for (unsigned w = t; w < nrows * BASES; w += T) {
        const unsigned row = w / BASES, j = w % BASES;
        const bool isq = row < (unsigned)L;
        const unsigned r = isq ? row : row - L;
        const DevModulus* md = (isq ? a.q_mods : a.aux_mods) + r;
        const double* tw = (isq ? a.tw_q : a.tw_aux) + (size_t)r * N;
        const u64* in = (isq ? iq : ib) + (size_t)r * N + base0 + j;
        const F64Mod m{md->pd, md->inv_pd};
        double x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = f64_corr(f64_bits_to_double(__builtin_nontemporal_load(in + k * SEG)), m);     // ArithF64::load_mid
        auto bf = [&](int r0, int r1, double tw_w) {
            const double u = x[r0], v = x[r1];
            x[r0] = u + v;
            x[r1] = f64_mulq(u - v, tw_w, m.inv_p, m.p);
        };
        // inverse table: layer l (forward numbering) starts at N - 2^(l+1) + 1
        const double w4 = tw[N - 7], w5 = tw[N - 6], w6 = tw[N - 5], w7 = tw[N - 4];
        bf(0, 1, w4); bf(2, 3, w5); bf(4, 5, w6); bf(6, 7, w7);
        const double w2 = tw[N - 3], w3 = tw[N - 2];
        bf(0, 2, w2); bf(1, 3, w2); bf(4, 6, w3); bf(5, 7, w3);
        const double nw = md->inv_n_w_d;            // ArithF64::inv_fold: the difference outputs take w N^-1 in one product
#pragma unroll
        for (int k = 0; k < 4; ++k) bf(k, k + 4, nw);
        const double ninv = md->inv_n_d, ninv_p = md->inv_n_pd;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double y = k < 4 ? f64_mulc(x[k], ninv, ninv_p, m.p) : x[k];          // final_inv / final_fwd
            lift_lds[row * T + k * BASES + j] = f64_canon(y, m);
        }
    }
    __syncthreads();
    const unsigned x = (t / BASES) * SEG + base0 + (t % BASES);
    u64* op = a.out + item * (size_t)L * N;
    behz2_floor_one<L, true, true>(c, [&](int i) { return lift_lds[i * T + t]; }, [&](unsigned b) { return lift_lds[(L + b) * T + t]; },
                                   [&](int jj, u64 wv) { __builtin_nontemporal_store(wv, op + (size_t)jj * N + x); });
}

}  // namespace troyn
