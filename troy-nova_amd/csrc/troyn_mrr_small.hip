// troyn_mrr_small.hip -- single objects through the fused multiply -> relinearize -> rescale entry at N = 8192 / 16384 (round 5).
//
// A launch of a few limb-polynomials takes the two-pass form of the transforms (ntt_launch.inl: 4 workgroups per limb and pass instead of one
// CU per 16384-point transform).  In that form the tail of the chain was six launches: special rows INTT (pass A, pass B) -> LAST_LIMB (A, B)
// -> TAIL_RESCALE (A, B).  The three strided passes in the middle -- the last two inverse layers of the special rows, the last two inverse
// layers of the dropped limb with the key switch's rounding fix, the first two forward layers of the L - 1 output limbs with both rounding
// fixes -- all work on the SAME quartets {i, i + N/4, i + N/2, i + 3N/4}: one thread can run them back to back with T_s and T_l in registers.
// The chain's tail becomes three launches: pass A over the rows {L - 1, special} together (they only depend on the inner product), this
// kernel, TAIL_RESCALE's pass B.  T_s and T_l never reach memory.
// Arithmetic: ArithF64's own functions in the order ntt_pass_body applies them (load_mid, inv / inv_fold, final_inv / final_fwd + round_half,
// last_out, tail_in, fwd, store_mid) with the constants of ntt_io_fused -- the words handed to TAIL_RESCALE's pass B are the ones its pass A wrote.
// Reference: evaluator_keyswitching_core.cu:570-658 (ski_util6/7), utils/rns_tool.cu:523-627 (divide_and_round_q_last_ntt).
#include "launch.hpp"

namespace troyn {

// the special rows share `la`'s tables: only their address and modulus travel (two NttArgs + 32 bytes of kernel arguments)
struct QuartetSpecial { const u64* in; long long in_bstride, in_pstride; unsigned mod; };

template <int LOGN, bool JPAR>
__global__ __launch_bounds__(256) void mrr_quartet_kernel(QuartetSpecial qs, NttArgs la, NttArgs ta) {
    using A = ArithF64;
    constexpr unsigned N = 1u << LOGN, Q = N / 4;
    typedef const double __attribute__((address_space(4)))* ctw;
    // one thread per (output limb j, item, polynomial, quartet): the two inverse tails are recomputed by the L - 1 threads that share a quartet
    // (8 loads and ~70 FP64 operations, from L2) -- a launch of a single ciphertext is latency-bound, and one limb per thread is the shorter chain
    // (JPAR = false: one thread per quartet loops over the output limbs -- a quarter of the workgroups and no recomputation: several host threads
    // with a stream each keep the GPU busy, and then total work counts, not the length of one chain)
    const unsigned gid = (JPAR ? blockIdx.x % ta.xcd_groups : blockIdx.x) * 256u + threadIdx.x;      // xcd_groups: workgroups per limb
    const unsigned j0 = JPAR ? blockIdx.x / ta.xcd_groups : 0u, j1 = JPAR ? j0 + 1u : ta.ncomp;
    const unsigned i = gid % Q, g = gid / Q, k = g & 1u, b = g >> 1;
    auto inverse_tail = [&](const u64* base, long long bstride, long long pstride, unsigned mi, const A::Mod& md, double (&v)[4]) {
        // pass B of an inverse transform in its two-pass form: layers 1 and 0 (Gentleman-Sande, the last one folded with N^-1)
        const u64* in = base + (long long)b * bstride + (long long)k * pstride;
        const ctw tw = (ctw)(unsigned long long)(reinterpret_cast<const double*>(la.tw) + (size_t)mi * N);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = A::load_mid(in[i + kk * Q], md);
        A::inv(v[0], v[1], A::tw_from_mem(tw[N - 3], md), md);
        A::inv(v[2], v[3], A::tw_from_mem(tw[N - 2], md), md);
        A::inv_fold(v[0], v[2], md);
        A::inv_fold(v[1], v[3], md);
    };
    // T_s = (s + qk/2) mod qk at the four coefficients (the plain inverse transform's NTT_FLAG_STORE_ROUND_HALF epilogue)
    u64 ts[4], tl[4];
    {
        const unsigned mi = qs.mod;
        const A::Mod md = A::make(la.mods[mi]);
        double v[4];
        inverse_tail(qs.in, qs.in_bstride, qs.in_pstride, mi, md, v);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u64 c = kk >= 2 ? A::final_fwd(v[kk], md) : A::final_inv(v[kk], md);      // the folded layer scaled its difference outputs already
            ts[kk] = f64_double_to_bits(A::round_half(f64_from_u64(c), md));
        }
    }
    // T_l = (l + ql/2) mod ql, l = INTT(Q_{L-1}) - r(s) qk^-1 (NTT_FUSED_LAST_LIMB's epilogue)
    {
        const unsigned mi = la.table_start;
        const A::Mod md = A::make(la.mods[mi]);
        NttIo io;
        ntt_io_fused(io, la, b, k, 0, mi);
        double v[4];
        inverse_tail(la.in, la.in_bstride, la.in_pstride, mi, md, v);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tl[kk] = A::template last_out<false>(io, v[kk], kk >= 2, ts[kk], md);
    }
    // the L - 1 output limbs: r_j(s) qk^-1 + f_j(l) enters ONE forward transform; its first two layers here (pass A of NTT_FUSED_TAIL_RESCALE)
    for (unsigned j = j0; j < j1; ++j) {
        const unsigned mi = ta.table_start + j;
        const A::Mod md = A::make(ta.mods[mi]);
        NttIo io;
        ntt_io_fused(io, ta, b, k, j, mi);
        const ctw tw = (ctw)(unsigned long long)(reinterpret_cast<const double*>(ta.tw) + (size_t)mi * N);
        double y[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) y[kk] = A::template tail_in<false>(io, ts[kk], tl[kk], md);
        const A::tw_t w1 = A::tw_from_mem(tw[1], md);
        A::fwd(y[0], y[2], w1, md);
        A::fwd(y[1], y[3], w1, md);
        A::fwd(y[0], y[1], A::tw_from_mem(tw[2], md), md);
        A::fwd(y[2], y[3], A::tw_from_mem(tw[3], md), md);
        u64* out = ta.out + (long long)b * ta.out_bstride + (long long)k * ta.out_pstride + (long long)j * ta.out_cstride;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) __builtin_nontemporal_store(A::store_mid(y[kk], md), out + i + kk * Q);
    }
}

// The same idea for the separate calls (troyn_switch_key / troyn_relinearize in NTT form, troyn_divide_and_round_q_last_ntt): there the tail is
// INTT of one row (special prime / dropped limb; passes A, B) -> forward transform of the data limbs whose loader forms the rounding fix from that
// row (ski_util6_merged :570-598 / divide_and_round_q_last_ntt step 1, utils/rns_tool.cu:523-550; passes A, B).  Pass B of the inverse and pass A of
// the forward transform share their quartets: one launch, the coefficient-form row never reaches memory.  LM = NTT_LOAD_KS_ROUND / NTT_LOAD_RESCALE.
template <class A, int LOGN, int LM>
__global__ __launch_bounds__(256) void mrr_quartet_load_kernel(NttArgs iv, NttArgs fw) {
    constexpr unsigned N = 1u << LOGN, Q = N / 4;
    typedef const typename A::tw_mem __attribute__((address_space(4)))* ctw;
    const unsigned gid = (blockIdx.x % fw.xcd_groups) * 256u + threadIdx.x, j = blockIdx.x / fw.xcd_groups;      // xcd_groups: workgroups per limb
    const unsigned i = gid % Q, g = gid / Q, k = g % fw.pcount, b = g / fw.pcount;
    u64 c[4];
    {
        const unsigned mi = iv.table_start;
        const typename A::Mod md = A::make(iv.mods[mi]);
        const u64* in = iv.in + (long long)b * iv.in_bstride + (long long)k * iv.in_pstride;
        const ctw tw = (ctw)(unsigned long long)(reinterpret_cast<const typename A::tw_mem*>(iv.tw) + (size_t)mi * N);
        typename A::elem v[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) v[kk] = A::load_mid(in[i + kk * Q], md);
        A::inv(v[0], v[1], A::tw_from_mem(tw[N - 3], md), md);
        A::inv(v[2], v[3], A::tw_from_mem(tw[N - 2], md), md);
        if constexpr (A::FOLD_NINV) {
            A::inv_fold(v[0], v[2], md);
            A::inv_fold(v[1], v[3], md);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c[kk] = kk >= 2 ? A::final_fwd(v[kk], md) : A::final_inv(v[kk], md);
        } else {
            // integer policy: the last layer is an ordinary butterfly and every output takes the (lazy) N^-1 multiply, as ntt_pass_body stores it
            const typename A::tw_t w0 = A::tw_from_mem(tw[N - 1], md);
            A::inv(v[0], v[2], w0, md);
            A::inv(v[1], v[3], w0, md);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) c[kk] = A::final_inv(v[kk], md);
        }
    }
    const unsigned mi = ntt_table_index(fw, k, j);
    const typename A::Mod md = A::make(fw.mods[mi]);
    const NttIo io = ntt_io_make(fw, b, k, j, mi, nullptr);
    const ctw tw = (ctw)(unsigned long long)(reinterpret_cast<const typename A::tw_mem*>(fw.tw) + (size_t)mi * N);
    typename A::elem y[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) y[kk] = A::template load_io<LM>(io, c[kk], fw.reduce_input != 0, md);
    const typename A::tw_t w1 = A::tw_from_mem(tw[1], md);
    A::fwd(y[0], y[2], w1, md);
    A::fwd(y[1], y[3], w1, md);
    A::fwd(y[0], y[1], A::tw_from_mem(tw[2], md), md);
    A::fwd(y[2], y[3], A::tw_from_mem(tw[3], md), md);
    u64* out = fw.out + (long long)b * fw.out_bstride + (long long)k * fw.out_pstride + (long long)j * fw.out_cstride;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) __builtin_nontemporal_store(A::store_mid(y[kk], md), out + i + kk * Q);
}

// iv: pass-A words of the row to invert (in, strides, table_start = its modulus, tw = inverse tables); fw: the forward launch's arguments with
// out = where its pass B reads; groups = batch * fw.pcount; f64: the arithmetic policy of BOTH transforms (one class per small launch)
template <class A>
static void launch_quartet_load_t(unsigned log_n, const dim3 grid, const NttArgs& iv, const NttArgs& t, hipStream_t s) {
    const dim3 block(256);
    const bool ks = t.load_mode == NTT_LOAD_KS_ROUND;
    if (log_n == 14) {
        if (ks) hipLaunchKernelGGL((mrr_quartet_load_kernel<A, 14, NTT_LOAD_KS_ROUND>), grid, block, 0, s, iv, t);
        else hipLaunchKernelGGL((mrr_quartet_load_kernel<A, 14, NTT_LOAD_RESCALE>), grid, block, 0, s, iv, t);
    } else {
        if (ks) hipLaunchKernelGGL((mrr_quartet_load_kernel<A, 13, NTT_LOAD_KS_ROUND>), grid, block, 0, s, iv, t);
        else hipLaunchKernelGGL((mrr_quartet_load_kernel<A, 13, NTT_LOAD_RESCALE>), grid, block, 0, s, iv, t);
    }
}
void launch_mrr_quartet_load(unsigned log_n, size_t groups, const NttArgs& iv, const NttArgs& fw, hipStream_t s, bool f64) {
    if (log_n != 13 && log_n != 14) return;
    const unsigned blocks = (unsigned)(groups * ((1u << log_n) / 4) / 256);
    NttArgs t = fw;
    t.xcd_groups = blocks;
    if (f64) launch_quartet_load_t<ArithF64>(log_n, dim3(blocks * fw.ncomp), iv, t, s);
    else launch_quartet_load_t<ArithU64>(log_n, dim3(blocks * fw.ncomp), iv, t, s);
}

// sp / la: the pass-A words of the special rows / of limb L - 1 (in, strides, table_start = their modulus, tw = inverse tables; la with the
// constants of step (4)); ta: step (5)'s arguments (out = where pass B reads, tw = forward tables).  batch * 2 polynomials.
void launch_mrr_quartet(unsigned log_n, size_t batch, const NttArgs& sp, const NttArgs& la, const NttArgs& ta, hipStream_t s, bool limb_parallel) {
    if (log_n != 13 && log_n != 14) return;
    const unsigned blocks = (unsigned)(batch * 2 * ((1u << log_n) / 4) / 256);
    NttArgs t = ta;
    t.xcd_groups = blocks;        // (the field is free in this kernel: workgroups per output limb)
    const QuartetSpecial qs{sp.in, sp.in_bstride, sp.in_pstride, sp.table_start};
    const dim3 grid(limb_parallel ? blocks * ta.ncomp : blocks), block(256);
    if (log_n == 14) {
        if (limb_parallel) hipLaunchKernelGGL((mrr_quartet_kernel<14, true>), grid, block, 0, s, qs, la, t);
        else hipLaunchKernelGGL((mrr_quartet_kernel<14, false>), grid, block, 0, s, qs, la, t);
    } else {
        if (limb_parallel) hipLaunchKernelGGL((mrr_quartet_kernel<13, true>), grid, block, 0, s, qs, la, t);
        else hipLaunchKernelGGL((mrr_quartet_kernel<13, false>), grid, block, 0, s, qs, la, t);
    }
}

}  // namespace troyn
