// troyn_mrr_small.hip -- a few ciphertexts through the fused multiply -> relinearize -> rescale entry and the separate calls, N = 8192 .. 32768 (round 5).
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

// The strided pass of a two-pass transform works on P = 2^G1 words {i + k N/P} (G1 = 2: quartets at N = 8192 / 16384, G1 = 3: octets at
// N = 32768); word k sits at index bits [LOGN - G1, LOGN) = k.  Forward layer l (l = 0 .. G1-1) pairs the words that differ in bit G1-1-l of k
// with twiddle (1 << l) + (k >> (G1 - l)); the inverse runs l = G1-1 .. 0 with twiddle N - (2 << l) + 1 + (k >> (G1 - l)) -- ntt_pass_body's
// numbering.  Everything below calls the policy's own functions in ntt_pass_body's order.
template <class A, int LOGN, int G1>
struct StridedPass {
    static constexpr unsigned N = 1u << LOGN, P = 1u << G1, Q = N >> G1;
    typedef const typename A::tw_mem __attribute__((address_space(4)))* ctw;
    static __device__ __forceinline__ ctw table(const void* tw, unsigned mi) {
        return (ctw)(unsigned long long)(reinterpret_cast<const typename A::tw_mem*>(tw) + (size_t)mi * N);
    }
    // pass B of an inverse transform: load_mid, the last G1 Gentleman-Sande layers (FP64: the final one folded with N^-1), then the words
    // ntt_pass_body would store: canonical under the FP64 policy, the (lazy) N^-1 multiply under the integer policy
    static __device__ __forceinline__ void inverse_tail(const u64* in, unsigned i, ctw tw, const typename A::Mod& md, u64 (&c)[P]) {
        typename A::elem v[P];
#pragma unroll
        for (unsigned k = 0; k < P; ++k) v[k] = A::load_mid(in[i + k * Q], md);
#pragma unroll
        for (int l = G1 - 1; l >= 1; --l) {
#pragma unroll
            for (unsigned k = 0; k < P; ++k)
                if (!(k & (1u << (G1 - 1 - l)))) A::inv(v[k], v[k | (1u << (G1 - 1 - l))], A::tw_from_mem(tw[N - (2u << l) + 1 + (k >> (G1 - l))], md), md);
        }
        if constexpr (A::FOLD_NINV) {
#pragma unroll
            for (unsigned k = 0; k < P / 2; ++k) A::inv_fold(v[k], v[k + P / 2], md);
#pragma unroll
            for (unsigned k = 0; k < P; ++k) c[k] = k >= P / 2 ? A::final_fwd(v[k], md) : A::final_inv(v[k], md);     // the folded layer scaled its difference outputs
        } else {
            const typename A::tw_t w0 = A::tw_from_mem(tw[N - 1], md);
#pragma unroll
            for (unsigned k = 0; k < P / 2; ++k) A::inv(v[k], v[k + P / 2], w0, md);
#pragma unroll
            for (unsigned k = 0; k < P; ++k) c[k] = A::final_inv(v[k], md);
        }
    }
    // pass A of a forward transform: the first G1 Cooley-Tukey layers, store_mid
    static __device__ __forceinline__ void forward_head(typename A::elem (&y)[P], ctw tw, const typename A::Mod& md, u64* out, unsigned i) {
#pragma unroll
        for (int l = 0; l < G1; ++l) {
#pragma unroll
            for (unsigned k = 0; k < P; ++k)
                if (!(k & (1u << (G1 - 1 - l)))) A::fwd(y[k], y[k | (1u << (G1 - 1 - l))], A::tw_from_mem(tw[(1u << l) + (k >> (G1 - l))], md), md);
        }
#pragma unroll
        for (unsigned k = 0; k < P; ++k) __builtin_nontemporal_store(A::store_mid(y[k], md), out + i + k * Q);
    }
};

// the special rows share `la`'s tables: only their address and modulus travel (two NttArgs + 32 bytes of kernel arguments)
struct QuartetSpecial { const u64* in; long long in_bstride, in_pstride; unsigned mod; };

template <int LOGN, int G1, bool JPAR>
__global__ __launch_bounds__(256) void mrr_quartet_kernel(QuartetSpecial qs, NttArgs la, NttArgs ta) {
    using A = ArithF64;
    using SP = StridedPass<A, LOGN, G1>;
    constexpr unsigned P = SP::P, Q = SP::Q;
    // one thread per (output limb j, item, polynomial, quartet): the two inverse tails are recomputed by the L - 1 threads that share a quartet
    // (2 P loads and ~70 FP64 operations, from L2) -- a launch of a few ciphertexts is latency-bound, and one limb per thread is the shorter chain
    // (JPAR = false: one thread per quartet loops over the output limbs -- fewer workgroups and no recomputation; A/B runs)
    const unsigned gid = (JPAR ? blockIdx.x % ta.xcd_groups : blockIdx.x) * 256u + threadIdx.x;      // xcd_groups: workgroups per limb
    const unsigned j0 = JPAR ? blockIdx.x / ta.xcd_groups : 0u, j1 = JPAR ? j0 + 1u : ta.ncomp;
    const unsigned i = gid % Q, g = gid / Q, k = g & 1u, b = g >> 1;
    // T_s = (s + qk/2) mod qk at the P coefficients (the plain inverse transform's NTT_FLAG_STORE_ROUND_HALF epilogue)
    u64 ts[P], tl[P];
    {
        const unsigned mi = qs.mod;
        const A::Mod md = A::make(la.mods[mi]);
        u64 c[P];
        SP::inverse_tail(qs.in + (long long)b * qs.in_bstride + (long long)k * qs.in_pstride, i, SP::table(la.tw, mi), md, c);
#pragma unroll
        for (unsigned kk = 0; kk < P; ++kk) ts[kk] = f64_double_to_bits(A::round_half(f64_from_u64(c[kk]), md));
    }
    // T_l = (l + ql/2) mod ql, l = INTT(Q_{L-1}) - r(s) qk^-1 (NTT_FUSED_LAST_LIMB's epilogue on the canonical word: last_out's own scaling is skipped)
    {
        const unsigned mi = la.table_start;
        const A::Mod md = A::make(la.mods[mi]);
        NttIo io;
        ntt_io_fused(io, la, b, k, 0, mi);
        u64 c[P];
        SP::inverse_tail(la.in + (long long)b * la.in_bstride + (long long)k * la.in_pstride, i, SP::table(la.tw, mi), md, c);
#pragma unroll
        for (unsigned kk = 0; kk < P; ++kk) tl[kk] = A::template last_out<false>(io, f64_from_u64(c[kk]), true, ts[kk], md);
    }
    // the L - 1 output limbs: r_j(s) qk^-1 + f_j(l) enters ONE forward transform; its first layers here (pass A of NTT_FUSED_TAIL_RESCALE)
    for (unsigned j = j0; j < j1; ++j) {
        const unsigned mi = ta.table_start + j;
        const A::Mod md = A::make(ta.mods[mi]);
        NttIo io;
        ntt_io_fused(io, ta, b, k, j, mi);
        double y[P];
#pragma unroll
        for (unsigned kk = 0; kk < P; ++kk) y[kk] = A::template tail_in<false>(io, ts[kk], tl[kk], md);
        SP::forward_head(y, SP::table(ta.tw, mi), md, ta.out + (long long)b * ta.out_bstride + (long long)k * ta.out_pstride + (long long)j * ta.out_cstride, i);
    }
}

// The same idea for the separate calls (troyn_switch_key / troyn_relinearize in NTT form, troyn_divide_and_round_q_last_ntt): there the tail is
// INTT of one row (special prime / dropped limb; passes A, B) -> forward transform of the data limbs whose loader forms the rounding fix from that
// row (ski_util6_merged :570-598 / divide_and_round_q_last_ntt step 1, utils/rns_tool.cu:523-550; passes A, B).  Pass B of the inverse and pass A of
// the forward transform share their words: one launch, the coefficient-form row never reaches memory.  LM = NTT_LOAD_KS_ROUND / NTT_LOAD_RESCALE.
template <class A, int LOGN, int G1, int LM>
__global__ __launch_bounds__(256) void mrr_quartet_load_kernel(NttArgs iv, NttArgs fw) {
    using SP = StridedPass<A, LOGN, G1>;
    constexpr unsigned P = SP::P, Q = SP::Q;
    const unsigned gid = (blockIdx.x % fw.xcd_groups) * 256u + threadIdx.x, j = blockIdx.x / fw.xcd_groups;      // xcd_groups: workgroups per limb
    const unsigned i = gid % Q, g = gid / Q, k = g % fw.pcount, b = g / fw.pcount;
    u64 c[P];
    {
        const unsigned mi = iv.table_start;
        SP::inverse_tail(iv.in + (long long)b * iv.in_bstride + (long long)k * iv.in_pstride, i, SP::table(iv.tw, mi), A::make(iv.mods[mi]), c);
    }
    const unsigned mi = ntt_table_index(fw, k, j);
    const typename A::Mod md = A::make(fw.mods[mi]);
    const NttIo io = ntt_io_make(fw, b, k, j, mi, nullptr);
    typename A::elem y[P];
#pragma unroll
    for (unsigned kk = 0; kk < P; ++kk) y[kk] = A::template load_io<LM>(io, c[kk], fw.reduce_input != 0, md);
    SP::forward_head(y, SP::table(fw.tw, mi), md, fw.out + (long long)b * fw.out_bstride + (long long)k * fw.out_pstride + (long long)j * fw.out_cstride, i);
}

// iv: pass-A words of the row to invert (in, strides, table_start = its modulus, tw = inverse tables); fw: the forward launch's arguments with
// out = where its pass B reads; groups = batch * fw.pcount; f64: the arithmetic policy of BOTH transforms (one class per small launch)
template <class A, int LOGN, int G1>
static void launch_quartet_load_t(const dim3 grid, const NttArgs& iv, const NttArgs& t, hipStream_t s) {
    if (t.load_mode == NTT_LOAD_KS_ROUND) hipLaunchKernelGGL((mrr_quartet_load_kernel<A, LOGN, G1, NTT_LOAD_KS_ROUND>), grid, dim3(256), 0, s, iv, t);
    else hipLaunchKernelGGL((mrr_quartet_load_kernel<A, LOGN, G1, NTT_LOAD_RESCALE>), grid, dim3(256), 0, s, iv, t);
}
static unsigned strided_words(unsigned log_n) { return log_n == 15 ? 8u : 4u; }
void launch_mrr_quartet_load(unsigned log_n, size_t groups, const NttArgs& iv, const NttArgs& fw, hipStream_t s, bool f64) {
    if (log_n < 13 || log_n > 15) return;
    const unsigned blocks = (unsigned)(groups * ((1u << log_n) / strided_words(log_n)) / 256);
    NttArgs t = fw;
    t.xcd_groups = blocks;
    const dim3 grid(blocks * fw.ncomp);
    if (f64) {
        if (log_n == 13) launch_quartet_load_t<ArithF64, 13, 2>(grid, iv, t, s);
        else if (log_n == 14) launch_quartet_load_t<ArithF64, 14, 2>(grid, iv, t, s);
        else launch_quartet_load_t<ArithF64, 15, 3>(grid, iv, t, s);
    } else {
        if (log_n == 13) launch_quartet_load_t<ArithU64, 13, 2>(grid, iv, t, s);
        else if (log_n == 14) launch_quartet_load_t<ArithU64, 14, 2>(grid, iv, t, s);
        else launch_quartet_load_t<ArithU64, 15, 3>(grid, iv, t, s);
    }
}

// sp / la: the pass-A words of the special rows / of limb L - 1 (in, strides, table_start = their modulus, tw = inverse tables; la with the
// constants of step (4)); ta: step (5)'s arguments (out = where pass B reads, tw = forward tables).  batch * 2 polynomials.
template <int LOGN, int G1>
static void launch_quartet_t(const QuartetSpecial& qs, const NttArgs& la, const NttArgs& t, unsigned blocks, hipStream_t s, bool limb_parallel) {
    if (limb_parallel) hipLaunchKernelGGL((mrr_quartet_kernel<LOGN, G1, true>), dim3(blocks * t.ncomp), dim3(256), 0, s, qs, la, t);
    else hipLaunchKernelGGL((mrr_quartet_kernel<LOGN, G1, false>), dim3(blocks), dim3(256), 0, s, qs, la, t);
}
void launch_mrr_quartet(unsigned log_n, size_t batch, const NttArgs& sp, const NttArgs& la, const NttArgs& ta, hipStream_t s, bool limb_parallel) {
    if (log_n < 13 || log_n > 15) return;
    const unsigned blocks = (unsigned)(batch * 2 * ((1u << log_n) / strided_words(log_n)) / 256);
    NttArgs t = ta;
    t.xcd_groups = blocks;        // (the field is free in this kernel: workgroups per output limb)
    const QuartetSpecial qs{sp.in, sp.in_bstride, sp.in_pstride, sp.table_start};
    if (log_n == 13) launch_quartet_t<13, 2>(qs, la, t, blocks, s, limb_parallel);
    else if (log_n == 14) launch_quartet_t<14, 2>(qs, la, t, blocks, s, limb_parallel);
    else launch_quartet_t<15, 3>(qs, la, t, blocks, s, limb_parallel);
}

}  // namespace troyn
