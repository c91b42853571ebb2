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
#include "dev_math_f64.hpp"
#include <type_traits>

namespace troyn {

constexpr int KS_MAX_KEYS = 64;  // HE_COEFF_MOD_COUNT_MAX (utils/constants.h:9)
struct KeyPtrs { const u64* p[KS_MAX_KEYS]; };

struct NttArgs {
    const u64* in;
    u64* out;
    const DevModulus* mods;   // [n_moduli]
    const void* tw;           // [n_moduli][N] forward or inverse table: (operand, quotient) u64 pairs, or w as double
    long long in_bstride, in_pstride, in_cstride;     // element strides: batch, polynomial, component
    long long out_bstride, out_pstride, out_cstride;
    unsigned pcount, ncomp;
    unsigned table_start, table_count, mode, decomp;  // NTTTableIndexer (utils/ntt.h:105-124)
    unsigned reduce_input;   // 1: Barrett-reduce inputs mod the limb's modulus while loading
    unsigned stream_loads;   // 1: inputs are read exactly once (non-temporal loads); 0: several blocks re-read them (keep in L2)
    // ---- fused element-wise prologue / epilogue (see NttLoad / NttStore below) ----
    unsigned load_mode, store_mode;
    unsigned aux_mod;          // modulus index of the "other" prime (special prime / dropped prime)
    unsigned skip_diag;        // 1: blocks with k == j < decomp produce nothing (the consumer reads the untouched input limb)
    unsigned flags;            // bit0: CKKS lift (4q instead of 2q); bits 1-2: SwitchKeyDestinationAssignMethod
    const u64* ext0;           // store: key-switch products / rescale input
    long long ext0_bstride, ext0_pstride, ext0_cstride;
    const u64* ext1;           // store: optional addend (relinearize's c0, c1)
    long long ext1_bstride, ext1_pstride, ext1_cstride;
    const ulonglong2* inv_table;   // store: Shoup pair of (dropped prime)^-1 mod q_j, indexed by component j
    unsigned batch;                // ks_mac_kernel: number of items (workgroup -> (row, item) mapping)
    unsigned xcd_groups;           // fused tail / rescale launches: batch * pcount groups whose ncomp limbs share one input row (0: off)
    long long key_pstride;         // ks_mac_kernel: elements between the two polynomials of a key (K*N)
    unsigned long long ks_row_mask; // ks_mac_kernel: 0 = all decomp + 1 output rows; else the launch covers the rows whose bit is set (mixed chains:
                                   // the rows of moduli >= 2^50 take this kernel, the others ksmac2_kernel)
    const u64* key_quo;            // ks_mac_kernel, integer policy: Shoup quotients floor(key 2^64 / q) of the keys, [j][2][K][N] (ks_key_quotients_kernel);
    long long key_quo_jstride;     // non-null: <digit, key> terms are lazy Shoup products instead of Barrett-128 reductions (elements between keys j)
    // ---- fused multiply -> relinearize -> rescale chain (IOM 3..5, NttFused below) ----
    const u64* mul_a; const u64* mul_b;       // the two input ciphertexts [item][2][limbs][N] (NTT form)
    long long mul_bstride, mul_pstride;       // element strides: item, polynomial (limb stride = N)
    unsigned mul_limb0;                       // limb of a / b that component j = 0 of the launch works on
    const u64* in2;                           // second coefficient-form input row (INTT of the dropped limb / of the special-prime row)
    long long in2_bstride, in2_pstride;
    unsigned aux2_mod;                        // modulus index of the dropped prime q_{L-1}
    const ulonglong2* inv_table2;             // Shoup pairs of q_{L-1}^-1 mod q_j
    unsigned fused_mode;                      // 0, or the NttFused variant to launch
    // ---- NTT_LOAD_CENTRALIZE: the input row is a plaintext modulo t, shared by the ncomp limbs (in_cstride = 0) ----
    u64 cz_t;                                 // plain modulus (t < every modulus of the launch: the fast plain lift)
    unsigned cz_count;                        // coefficients the plaintext holds; the rest of the row is zero and is not read
};

// Fused prologues: what a coefficient looks like when it enters the transform.
enum NttLoad {
    NTT_LOAD_PLAIN = 0,
    // evaluator_keyswitching_core.cu:570-598 (ski_util6_merged): input = INTT of the special-prime row (same for
    // every component j); value = ((x + qk/2) mod qk) mod q_j + (q_j - (qk/2 mod q_j))
    NTT_LOAD_KS_ROUND = 1,
    // utils/rns_tool.cu:523-550 (divide_and_round_q_last_ntt step 1): input = INTT of the last limb;
    // value = ((x + ql/2) mod ql) mod q_i - (ql/2 mod q_i)
    NTT_LOAD_RESCALE = 2,
    // utils/scaling_variant.cu:326-357 (scaling_variant::centralize, fast plain lift t < q_j) in front of Evaluator::transform_plain_to_ntt
    // (evaluator_transform_ntt.cu:35-70): input = plaintext coefficient m in [0, t), zero beyond the plaintext's length;
    // value = m + (q_j - t) when m >= (t + 1) / 2, else m
    NTT_LOAD_CENTRALIZE = 3,
};
constexpr int NTT_IOM_CENTRALIZE = 8;      // kernel variant (template parameter IOM of ntt_pass_kernel) that loads through NTT_LOAD_CENTRALIZE
// Fused epilogues: what happens to a canonical NTT output y before it is stored.
enum NttStore {
    NTT_STORE_PLAIN = 0,
    // evaluator_keyswitching_core.cu:625-658 (ski_util7_merged): out = (prod + lift - y) * qk^-1 mod q_j, then
    // overwrite / accumulate into the destination, plus relinearize's trailing add (evaluator_keyswitching.cu:143)
    NTT_STORE_KS_FINISH = 1,
    // utils/rns_tool.cu:607-627 (divide_and_round_q_last_ntt step 2): out = (in + 4q - y) * ql^-1 mod q_i
    NTT_STORE_RESCALE = 2,
};

// Kernels of the fused CKKS multiply -> relinearize -> rescale chain (troyn_ckks_multiply_relinearize_rescale).  With c = a (x) b
// the dyadic tensor product (c0 = a0 b0, c1 = a0 b1 + a1 b0, c2 = a1 b1), P the key-switch inner product of c2, s = INTT of P's
// special-prime row, r_j(s) the rounding fix of the key switch and f_j(l) the rounding fix of the rescale, linearity of the
// transforms (all arithmetic is exact mod q_j) gives
//     relin_j   = (P_j - NTT_j(r_j(s))) qk^-1 + c_j                         (evaluator_keyswitching_core.cu:570-658 + :143)
//     l         = INTT(relin_{L-1}) = INTT(P_{L-1} qk^-1 + c_{L-1}) - r_{L-1}(s) qk^-1
//     out_j     = (relin_j - NTT_j(f_j(l))) ql^-1                           (utils/rns_tool.cu:523-627)
//               = (P_j qk^-1 + c_j - NTT_j(r_j(s) qk^-1 + f_j(l))) ql^-1
// i.e. ONE forward transform per output limb instead of two, and c is never materialised.  Since round 3 the inner product kernel hands
// over Q_j = P_j qk^-1 + c_j for the data rows (its keys are prepared times qk^-1 and its epilogue adds the tensor terms, KsMacArgs::ten_a):
// the kernels below read one row where they used to read P_j and four rows of a and b.
constexpr unsigned NTT_FLAG_STORE_F64 = 16u;         // NttArgs::flags: NTT_FUSED_MULPAIR stores its canonical outputs as doubles (the digits ksmac2 reads)
constexpr unsigned NTT_FLAG_STORE_ROUND_HALF = 8u;   // NttArgs::flags: a plain inverse transform stores T = (x + q/2) mod q (FP64 policy: as doubles; integer policy: u64 words)
// chains with moduli of 2^50 and more (round 5): a T row written by an integer kernel holds u64 words, one written by an FP64 kernel doubles
constexpr unsigned NTT_FLAG_TS_U64 = 32u;            // NttArgs::flags: the rows T_s = (s + qk/2) mod qk (special prime) hold u64 words
constexpr unsigned NTT_FLAG_TL_U64 = 64u;            // NttArgs::flags: the rows T_l = (l + ql/2) mod ql (dropped prime) hold u64 words

enum NttFused {
    NTT_FUSED_MULPAIR = 3,       // inverse: input word = a1 (.) b1 (the product c2 formed while loading)
    NTT_FUSED_LAST_LIMB = 4,     // inverse: input = Q = P qk^-1 + c_k at limb L-1 (from ksmac2); stored word = result - r(s) qk^-1   (= l above)
    NTT_FUSED_TAIL_RESCALE = 5,  // forward: input = r_j(s) qk^-1 + f_j(l); stored word = (Q_kj - y) ql^-1
    // FP64 policy only: the same two kernels for chains in which the special and / or the dropped prime is 2^50 or wider -- a T row may then hold
    // u64 words (NTT_FLAG_TS_U64 / NTT_FLAG_TL_U64), reduced in FP64 while loading.  Separate instantiations: the all-FP64 chain's kernels are
    // tuned to their register budgets and stay as they are.
    NTT_FUSED_TAIL_RESCALE_W = 6,
    NTT_FUSED_LAST_LIMB_W = 7,
};

struct NttIo {
    unsigned load_mode, store_mode;
    // loader constants
    u64 aux_q, aux_ratio_hi, aux_half;   // the other prime, its Barrett word, floor(aux/2)
    u64 q, ratio_hi, fix;                // this limb's prime; fix = q - (aux_half mod q)  (KS)  or  (aux_half mod q) (rescale)
    bool aux_bigger;
    bool aux_wide;                       // aux >= 2^50: the FP64 policy reduces the word with integer arithmetic first
    // storer
    const u64* ext0; const u64* ext1; u64* dest;
    ulonglong2 inv; u64 lift; bool add_inplace;
    // the same constants as doubles, for the FP64 policy (all moduli < 2^50, exact integers)
    double aux_qd, aux_half_d, hm_d;     // other prime, floor(aux/2), (aux_half mod q)
    double inv_d, inv_pd;                // (dropped prime)^-1 mod q and fl(inv/q)
    // fused chain: the dropped prime q_{L-1} next to the special prime
    double aux2_qd, aux2_half_d, hm2_d, inv2_d;
    const u64* in2;                      // second input row of this (item, polynomial)
    const u64 *a0, *a1, *b0, *b1;        // this limb of the two input ciphertexts
    unsigned poly;                       // output polynomial k: c_0 = a0 b0, c_1 = a0 b1 + a1 b0
    // fused chain, integer forms / mixed chains: the constants above as words, and what the T rows hold
    u64 hm_u, hm2_u;                     // (qk/2 mod q), (ql/2 mod q)
    u64 aux2_q;                          // the dropped prime ql
    bool aux2_bigger;                    // ql > q
    ulonglong2 inv2;                     // Shoup pair of ql^-1 mod q
    bool ts_u64, tl_u64;                 // NTT_FLAG_TS_U64 / NTT_FLAG_TL_U64
    // NTT_LOAD_CENTRALIZE
    u64 cz_thr, cz_inc;                  // (t + 1) / 2 and q - t
    unsigned cz_count;
};

__device__ __forceinline__ NttIo ntt_io_make(const NttArgs& a, unsigned b, unsigned k, unsigned j, unsigned mi, u64* gout) {
    NttIo io;
    io.load_mode = a.load_mode; io.store_mode = a.store_mode;
    const DevModulus md = a.mods[mi];
    io.q = md.q; io.ratio_hi = md.ratio_hi;
    io.aux_q = 0; io.aux_ratio_hi = 0; io.aux_half = 0; io.fix = 0; io.aux_bigger = false; io.aux_wide = false;
    io.aux_qd = 0.0; io.aux_half_d = 0.0; io.hm_d = 0.0;
    io.cz_thr = 0; io.cz_inc = 0; io.cz_count = 0;
    if (a.load_mode == NTT_LOAD_CENTRALIZE) {
        io.cz_thr = (a.cz_t + 1) >> 1; io.cz_inc = md.q - a.cz_t; io.cz_count = a.cz_count;
    } else if (a.load_mode != NTT_LOAD_PLAIN) {
        const DevModulus ax = a.mods[a.aux_mod];
        io.aux_q = ax.q; io.aux_ratio_hi = ax.ratio_hi; io.aux_half = ax.q >> 1;
        const u64 half_mod = barrett64(io.aux_half, md.q, md.ratio_hi);
        io.fix = (a.load_mode == NTT_LOAD_KS_ROUND) ? md.q - half_mod : half_mod;
        io.aux_bigger = ax.q > md.q;
        io.aux_wide = (ax.q >> 50) != 0;
        io.aux_qd = ax.pd; io.aux_half_d = (double)io.aux_half; io.hm_d = (double)half_mod;
    }
    io.ext0 = nullptr; io.ext1 = nullptr; io.dest = gout; io.inv = make_ulonglong2(0, 0); io.lift = 0; io.add_inplace = false;
    io.inv_d = 0.0; io.inv_pd = 0.0;
    if (a.store_mode != NTT_STORE_PLAIN) {
        io.ext0 = a.ext0 + (long long)b * a.ext0_bstride + (long long)k * a.ext0_pstride + (long long)j * a.ext0_cstride;
        if (a.ext1) io.ext1 = a.ext1 + (long long)b * a.ext1_bstride + (long long)k * a.ext1_pstride + (long long)j * a.ext1_cstride;
        io.inv = a.inv_table[j];
        io.inv_d = (double)io.inv.x; io.inv_pd = io.inv_d * md.inv_pd;
        io.lift = (a.store_mode == NTT_STORE_RESCALE || (a.flags & 1u)) ? (md.q << 2) : (md.q << 1);
        const unsigned assign = (a.flags >> 1) & 3u;
        io.add_inplace = (a.store_mode == NTT_STORE_KS_FINISH) && (assign == 0u || (k == 0u && assign == 2u));
    }
    return io;
}

// constants and operand rows of the fused chain for workgroup (item b, polynomial k, component j) under modulus mi
__device__ __forceinline__ void ntt_io_fused(NttIo& io, const NttArgs& a, unsigned b, unsigned k, unsigned j, unsigned mi) {
    const DevModulus md = a.mods[mi];
    io.q = md.q; io.ratio_hi = md.ratio_hi;
    io.poly = k;
    io.in2 = nullptr; io.a0 = io.a1 = io.b0 = io.b1 = nullptr; io.ext0 = nullptr; io.ext1 = nullptr;
    io.aux_qd = io.aux_half_d = io.hm_d = io.inv_d = io.inv_pd = 0.0;
    io.aux2_qd = io.aux2_half_d = io.hm2_d = io.inv2_d = 0.0;
    io.hm_u = io.hm2_u = 0; io.aux_q = io.aux2_q = 0; io.aux_bigger = io.aux2_bigger = false;
    io.inv = io.inv2 = make_ulonglong2(0, 0);
    io.ts_u64 = (a.flags & NTT_FLAG_TS_U64) != 0; io.tl_u64 = (a.flags & NTT_FLAG_TL_U64) != 0;
    if (a.mul_a) {
        const u64* pa = a.mul_a + (long long)b * a.mul_bstride;
        const u64* pb = a.mul_b + (long long)b * a.mul_bstride;
        io.a0 = pa; io.a1 = pa + a.mul_pstride; io.b0 = pb; io.b1 = pb + a.mul_pstride;    // + limb * N added by the caller (N is a template constant there)
    }
    if (a.in2) io.in2 = a.in2 + (long long)b * a.in2_bstride + (long long)k * a.in2_pstride;
    if (a.ext0) io.ext0 = a.ext0 + (long long)b * a.ext0_bstride + (long long)k * a.ext0_pstride + (long long)j * a.ext0_cstride;
    if (a.inv_table) {
        // special prime qk: rounding fix r(s) = ((s + qk/2) mod qk) - (qk/2 mod q), and qk^-1 mod q
        const DevModulus ax = a.mods[a.aux_mod];
        const u64 half = ax.q >> 1;
        io.hm_u = barrett64(half, md.q, md.ratio_hi); io.aux_q = ax.q; io.aux_bigger = ax.q > md.q;
        io.aux_qd = ax.pd; io.aux_half_d = (double)half; io.hm_d = (double)io.hm_u;
        io.inv = a.inv_table[j]; io.inv_d = (double)io.inv.x;
    }
    if (a.inv_table2) {
        const DevModulus ax = a.mods[a.aux2_mod];
        const u64 half = ax.q >> 1;
        io.hm2_u = barrett64(half, md.q, md.ratio_hi); io.aux2_q = ax.q; io.aux2_bigger = ax.q > md.q;
        io.aux2_qd = ax.pd; io.aux2_half_d = (double)half; io.hm2_d = (double)io.hm2_u;
        io.inv2 = a.inv_table2[j]; io.inv2_d = (double)io.inv2.x;
    }
}

// raw input word -> the word the transform should see (still subject to A::load_first)
template <int LM>
__device__ __forceinline__ u64 ntt_io_load(const NttIo& io, u64 raw) {
    if constexpr (LM == NTT_LOAD_KS_ROUND) {
        u64 t = barrett64(raw + io.aux_half, io.aux_q, io.aux_ratio_hi);
        if (io.aux_bigger) t = barrett64(t, io.q, io.ratio_hi);
        return t + io.fix;
    }
    if constexpr (LM == NTT_LOAD_RESCALE) {
        u64 t = add_mod(raw, io.aux_half, io.aux_q);
        if (io.aux_bigger) t = barrett64(t, io.q, io.ratio_hi);   // q_i < q_last
        return sub_mod(t, io.fix, io.q);
    }
    if constexpr (LM == NTT_LOAD_CENTRALIZE) return raw >= io.cz_thr ? raw + io.cz_inc : raw;      // canonical: m + q - t < q
    return raw;
}

// canonical transform output y at limb-local index idx -> stored word
// Operands of the epilogue are passed by value (prod = ext0 word, addend = ext1 word, destv = old destination word)
// so that the caller can request all of them before any arithmetic starts.
template <int SM>
__device__ __forceinline__ u64 ntt_io_store(const NttIo& io, u64 y, u64 prod, u64 addend, u64 destv) {
    if constexpr (SM == NTT_STORE_KS_FINISH) {
        u64 d = shoup_mul(prod + io.lift - y, io.inv.x, io.inv.y, io.q);
        if (io.add_inplace) d = add_mod(destv, d, io.q);
        if (io.ext1) d = add_mod(d, addend, io.q);
        return d;
    }
    if constexpr (SM == NTT_STORE_RESCALE) {
        u64 d = add_mod(prod, io.lift, io.q);    // add_uint64_mod(x, 4q): x + 3q
        d = sub_mod(d, y, io.q);
        return shoup_mul(d, io.inv.x, io.inv.y, io.q);
    }
    return y;
}


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
// lds_phys(base | X) == lds_phys(base) + lds_off(X) whenever base has zeros in every bit position X can set (the
// low-5-bit parts are then disjoint, so the padding term splits without a carry).  All LDS accesses of a round are
// therefore one per-thread base plus compile-time offsets (immediate operands of ds_read / ds_write).
__host__ __device__ constexpr unsigned lds_off(unsigned x) { return x + (x >> NTT_PAD_SHIFT); }

// Streaming (non-temporal) accesses for the polynomial data: every word is touched once per kernel, while the
// twiddle tables (same size as one limb, shared by every workgroup of that modulus) should stay in the XCD's L2.
#ifdef TROYN_NO_NT
__device__ __forceinline__ u64 nt_load(const u64* p) { return *p; }
__device__ __forceinline__ ulonglong2 nt_load2(const u64* p) { return *reinterpret_cast<const ulonglong2*>(p); }
__device__ __forceinline__ void nt_store(u64* p, u64 v) { *p = v; }
__device__ __forceinline__ void nt_store2(u64* p, u64 a, u64 b) { *reinterpret_cast<ulonglong2*>(p) = make_ulonglong2(a, b); }
#else
typedef u64 u64x2_native __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u64 nt_load(const u64* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ ulonglong2 nt_load2(const u64* p) {
    const u64x2_native v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_native*>(p));
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void nt_store(u64* p, u64 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void nt_store2(u64* p, u64 a, u64 b) {
    u64x2_native v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<u64x2_native*>(p));
}
#endif

// store / load of one word at (wave-uniform base) + (32-bit lane byte offset): the scalar-base addressing form, so that E strided
// words of a thread cost one offset register instead of E 64-bit addresses
typedef char __attribute__((address_space(1)))* nt_gcp;
__device__ __forceinline__ void nt_store_at(u64* ubase, unsigned byte_off, u64 v) {
    typedef u64 __attribute__((address_space(1)))* gp;
    __builtin_nontemporal_store(v, (gp)((nt_gcp)(unsigned long long)ubase + byte_off));
}
__device__ __forceinline__ ulonglong2 ld2_at(const u64* ubase, unsigned byte_off, bool nontemporal = false) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    typedef const v2 __attribute__((address_space(1)))* gp;
    const gp p = (gp)((nt_gcp)(unsigned long long)ubase + byte_off);
    const v2 v = nontemporal ? __builtin_nontemporal_load(p) : *p;
    return make_ulonglong2(v.x, v.y);
}
__device__ __forceinline__ void nt_store2_at(u64* ubase, unsigned byte_off, u64 a, u64 b) {
    typedef u64 v2 __attribute__((ext_vector_type(2)));
    typedef v2 __attribute__((address_space(1)))* gp;
    v2 v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, (gp)((nt_gcp)(unsigned long long)ubase + byte_off));
}
__device__ __forceinline__ u64 nt_ld_at(const u64* ubase, unsigned byte_off) {
    typedef const u64 __attribute__((address_space(1)))* gp;
    return __builtin_nontemporal_load((gp)((nt_gcp)(unsigned long long)ubase + byte_off));
}
__device__ __forceinline__ u64 ld_at(const u64* ubase, unsigned byte_off) {
    typedef const u64 __attribute__((address_space(1)))* gp;
    return *(gp)((nt_gcp)(unsigned long long)ubase + byte_off);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E_, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E_) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E_>(f);
    }
}

// index of the (n+1)-th set bit of mask (row subsets of the key-switch inner product kernels)
__host__ __device__ inline unsigned nth_set_bit(unsigned long long mask, unsigned n) {
    for (unsigned i = 0; i < 64; i++)
        if ((mask >> i) & 1ull) { if (n == 0) return i; --n; }
    return 0;
}

// ---- arithmetic policies -------------------------------------------------------------------
// ArithU64: 64-bit Harvey/Shoup butterflies in the integer ALU, any modulus < 2^61 (the reference's
//           lazy ranges: [0,4q) forward, [0,2q) inverse; fgk/ntt_grouped.cu:204-233, :540-572).
// ArithF64: the same butterflies on integer-valued doubles for moduli < 2^50 (dev_math_f64.hpp).
typedef u64 u64x2_mem __attribute__((ext_vector_type(2)));
struct ArithU64 {
    using elem = u64;
    using tw_t = ulonglong2;
    using tw_mem = u64x2_mem;     // (operand, quotient) as stored in the table

    static constexpr bool MID_FIX = false;
    static constexpr bool FOLD_NINV = false;
    // Round 2: the butterflies estimate the Shoup quotient from the three high 32 x 32 partial products only (shoup_lazy3: never
    // above floor(x wq / 2^64), at most 2 below), which saves the low product, its carry chain and the register moves around them --
    // the product lands in [0, 4q) instead of [0, 2q), so the lazy ranges are twice the reference's: [0, 8q) forward, [0, 4q) inverse
    // (q < 2^61, 8q < 2^64).  Stored results are canonical, hence unchanged.
    struct Mod { u64 q, two_q, ratio_hi, ninv_op, ninv_quo, ratio_lo, four_q, neg_four_q, neg_q; };
    static __device__ __forceinline__ Mod make(const DevModulus& d) { return Mod{d.q, d.q << 1, d.ratio_hi, d.inv_n_op, d.inv_n_quo, d.ratio_lo, d.q << 2, 0ull - (d.q << 2), 0ull - d.q}; }
    static __device__ __forceinline__ u64 final_fwd(elem v, const Mod& m) {     // [0, 8q) -> [0, q)
        v = v >= m.four_q ? v - m.four_q : v;
        v = v >= m.two_q ? v - m.two_q : v;
        return v >= m.q ? v - m.q : v;
    }
    // neg_q = 2^64 - q: w x - qh q = w x + qh neg_q (mod 2^64) -- a 64-bit add is one v_lshl_add_u64, a subtract is a carry chain
    static __device__ __forceinline__ u64 shoup_lazy3(u64 x, u64 w, u64 wq, u64 neg_q) {   // any 64-bit x -> [0, 4q)
        const unsigned x0 = (unsigned)x, x1 = (unsigned)(x >> 32), w0 = (unsigned)wq, w1 = (unsigned)(wq >> 32);
        const u64 qh = (u64)x1 * w1 + __umulhi(x1, w0) + __umulhi(x0, w1);
        return w * x + qh * neg_q;
    }
    static __device__ __forceinline__ u64 csub4(u64 v, const Mod& m) {   // [0, 8q) -> [0, 4q)
        const u64 t = v + m.neg_four_q;
        return (long long)t < 0 ? v : t;
    }
    // <digit, key> accumulation of the fused key switch (ks_mac_kernel): canonical in, canonical accumulator
    static __device__ __forceinline__ elem mac_zero() { return 0; }
    static __device__ __forceinline__ elem mac_in(u64 canonical, const Mod&) { return canonical; }
    static __device__ __forceinline__ u64 mac_to_lds(elem x, const Mod& m) { return final_fwd(x, m); }
    static __device__ __forceinline__ void mac(elem& acc, elem v, u64 key, const Mod& m) {
        acc = add_mod(acc, barrett128(v * key, mul_hi(v, key), m.q, m.ratio_lo, m.ratio_hi), m.q);
    }
    static __device__ __forceinline__ void mac_fix(elem&, const Mod&) {}
    static __device__ __forceinline__ u64 mac_final(elem acc, const Mod& m) { return final_fwd(acc, m); }     // canonical stays canonical; the Shoup form leaves [0, 4q)
    // the same term with the key's Shoup quotient kq = floor(key 2^64 / q) (exact): a lazy product in [0, 4q) on a lazy sum in [0, 4q)
    // (8q < 2^64) -- 3 + 2 multiplier instructions instead of the 4 + ~12 of a 128-bit product and its Barrett reduction
    static constexpr bool HAS_MAC_SHOUP = true;
    static __device__ __forceinline__ void mac_shoup(elem& acc, elem v, u64 key, u64 kq, const Mod& m) {
        acc = csub4(acc + shoup_lazy3(v, key, kq, m.neg_q), m);
    }
    static __device__ __forceinline__ tw_t tw_from_mem(const tw_mem v, const Mod&) { return make_ulonglong2(v.x, v.y); }
    static __device__ __forceinline__ elem load_first(u64 raw, bool reduce, const Mod& m) { return reduce ? barrett64(raw, m.q, m.ratio_hi) : raw; }
    // fused prologue / epilogue (NttLoad / NttStore): integer forms of the reference kernels
    template <int LM> static __device__ __forceinline__ elem load_io(const NttIo& io, u64 raw, bool reduce, const Mod& m) {
        return load_first(ntt_io_load<LM>(io, raw), reduce, m);
    }
    template <int SM> static __device__ __forceinline__ u64 store_prep(elem x, const Mod& m) { return final_fwd(x, m); }
    template <int SM> static __device__ __forceinline__ u64 store_io(const NttIo& io, u64 word, u64 prod, u64 addend, u64 destv, const Mod&) {
        return ntt_io_store<SM>(io, word, prod, addend, destv);
    }
    static __device__ __forceinline__ elem load_mid(u64 raw, const Mod&) { return raw; }
    static __device__ __forceinline__ u64 store_mid(elem x, const Mod&) { return x; }
    static __device__ __forceinline__ elem from_lds(u64 raw) { return raw; }
    static __device__ __forceinline__ u64 to_lds(elem x, const Mod&) { return x; }
    static __device__ __forceinline__ elem mid_fix(elem x, const Mod&) { return x; }
    static __device__ __forceinline__ void fwd(elem& a, elem& b, const tw_t w, const Mod& m) {
        const u64 u = csub4(a, m);
        const u64 v = shoup_lazy3(b, w.x, w.y, m.neg_q);
        a = u + v;
        b = u + m.four_q - v;
    }
    static __device__ __forceinline__ void inv(elem& a, elem& b, const tw_t w, const Mod& m) {
        const u64 u = a, v = b;
        a = csub4(u + v, m);
        b = shoup_lazy3(u + m.four_q - v, w.x, w.y, m.neg_q);
    }
    static __device__ __forceinline__ u64 final_inv(elem v, const Mod& m) {
        return shoup_lazy(final_fwd(v, m), m.ninv_op, m.ninv_quo, m.q);   // the reference's lazy N^-1 multiply
    }
    static __device__ __forceinline__ void inv_fold(elem&, elem&, const Mod&) {}   // FOLD_NINV is false: never used
    // register hand-over between a forward and an inverse transform (tensor_core_kernel): canonical residues
    static __device__ __forceinline__ elem keep(elem x, const Mod& m) { return final_fwd(x, m); }
    static __device__ __forceinline__ elem prod(elem x, elem y, const Mod& m) {
        u64 lo, hi;
        mul128(x, y, lo, hi);
        return barrett128(lo, hi, m.q, m.ratio_lo, m.ratio_hi);
    }
    static __device__ __forceinline__ elem sum(elem x, elem y, const Mod& m) { return add_mod(x, y, m.q); }
    static __device__ __forceinline__ elem inv_in(elem x, const Mod&) { return x; }
    // ---- fused multiply -> relinearize -> rescale chain (NttFused), integer forms: every value a canonical residue ----
    static __device__ __forceinline__ elem prod_in(u64 x, u64 y, const Mod& m) { return prod(x, y, m); }
    static __device__ __forceinline__ elem from_canon(u64 x) { return x; }
    // a word of a T row (T = (x + aux/2) mod aux, stored by the producer as u64 or as an exact double) as a residue of this limb's modulus
    static __device__ __forceinline__ u64 t_residue(u64 raw, bool is_u64, bool aux_bigger, const Mod& m) {
        const u64 t = is_u64 ? raw : f64_to_u64(f64_bits_to_double(raw));
        return aux_bigger ? barrett64(t, m.q, m.ratio_hi) : t;
    }
    // input of the fused tail + rescale transform: r_j(s) qk^-1 + f_j(l)
    template <bool TW> static __device__ __forceinline__ elem tail_in(const NttIo& io, u64 raw_s, u64 raw_l, const Mod& m) {
        const u64 rs = sub_mod(t_residue(raw_s, io.ts_u64, io.aux_bigger, m), io.hm_u, m.q);
        const u64 fl = sub_mod(t_residue(raw_l, io.tl_u64, io.aux2_bigger, m), io.hm2_u, m.q);
        return add_mod(shoup_mul(rs, io.inv.x, io.inv.y, m.q), fl, m.q);
    }
    // its stored word: (Q - y) ql^-1
    static __device__ __forceinline__ u64 tail_out(const NttIo& io, u64 prod_word, elem y, const Mod& m) {
        return shoup_mul(sub_mod(prod_word, final_fwd(y, m), m.q), io.inv2.x, io.inv2.y, m.q);
    }
    // stored word of the dropped limb's inverse transform: T_l = (l + ql/2) mod ql with l = INTT(Q) - r(s) qk^-1
    template <bool TW> static __device__ __forceinline__ u64 last_out(const NttIo& io, elem xr, bool, u64 raw_s, const Mod& m) {
        const u64 ys = final_inv(xr, m);
        const u64 rs = sub_mod(t_residue(raw_s, io.ts_u64, io.aux_bigger, m), io.hm_u, m.q);
        const u64 l = sub_mod(ys, shoup_mul(rs, io.inv.x, io.inv.y, m.q), m.q);
        return add_mod(l, m.q >> 1, m.q);
    }
};

struct ArithF64 {
    static constexpr bool HAS_MAC_SHOUP = false;
    using elem = double;
    using tw_t = double2;
    using tw_mem = double;        // only w is stored (8 bytes per twiddle); w/p is rebuilt as w * fl(1/p)
    static constexpr bool MID_FIX = true;   // inverse blocks of 4 layers re-centre their sums after 2
    // the final inverse layer multiplies its difference output by (w * N^-1) directly, so only the sum outputs
    // still need the N^-1 multiply at the end (8 of 16 per thread)
    static constexpr bool FOLD_NINV = true;
    struct Mod { F64Mod m; double ninv, ninv_p, nw, nw_p; };
    static __device__ __forceinline__ Mod make(const DevModulus& d) { return Mod{F64Mod{d.pd, d.inv_pd}, d.inv_n_d, d.inv_n_pd, d.inv_n_w_d, d.inv_n_w_pd}; }
    static __device__ __forceinline__ void inv_fold(elem& a, elem& b, const Mod& m) {
        const double u = a, v = b;
        a = u + v;
        b = f64_mulq(u - v, m.nw, m.m.inv_p, m.m.p);
    }
    // <digit, key> accumulation: |v| <= 0.5p+1, key in [0,p).  The quotient is estimated from the product itself
    // (q = rint(fl(v*y) * fl(1/p)), off by < 0.2 from v*y/p), so each term is an exact integer of magnitude <= 0.69p;
    // the accumulator is re-centred every 8 terms (mac_fix), i.e. it stays below 6.1p < 2^53.
    static __device__ __forceinline__ elem mac_zero() { return 0.0; }
    static __device__ __forceinline__ elem mac_in(u64 canonical, const Mod& m) { return f64_corr(f64_from_u64(canonical), m.m); }
    static __device__ __forceinline__ u64 mac_to_lds(elem x, const Mod& m) { return f64_double_to_bits(f64_corr(x, m.m)); }
    static __device__ __forceinline__ void mac(elem& acc, elem v, u64 key, const Mod& m) {
        const double y = f64_from_u64(key);
        const double h = v * y;
        const double l = __builtin_fma(v, y, -h);
        const double q = __builtin_rint(h * m.m.inv_p);
        acc += __builtin_fma(-q, m.m.p, h) + l;
    }
    static __device__ __forceinline__ void mac_fix(elem& acc, const Mod& m) { acc = f64_corr(acc, m.m); }
    static __device__ __forceinline__ u64 mac_final(elem acc, const Mod& m) { return f64_canon(acc, m.m); }
    static __device__ __forceinline__ tw_t tw_from_mem(const tw_mem w, const Mod& m) { return make_double2(w, w * m.m.inv_p); }
    static __device__ __forceinline__ elem load_first(u64 raw, bool, const Mod& m) { return f64_corr(f64_from_u64(raw), m.m); }
    // fused prologue: T = (x + aux/2) mod aux, value = T - (aux/2 mod p)  (KS_ROUND adds p - (..), the same residue)
    template <int LM> static __device__ __forceinline__ elem load_io(const NttIo& io, u64 raw, bool, const Mod& m) {
        if constexpr (LM == NTT_LOAD_PLAIN) return f64_corr(f64_from_u64(raw), m.m);
        else if constexpr (LM == NTT_LOAD_CENTRALIZE) return f64_corr(f64_from_u64(ntt_io_load<LM>(io, raw)), m.m);
        else {
            if (io.aux_wide) {
                // the dropped prime has 50 bits or more ({40,40,60}: a wide special / last prime over narrow data limbs): the word does not
                // fit a double, so T and its residue mod p are formed with the integer policy's arithmetic (ntt_io_load) and only the
                // canonical residue becomes a double
                u64 t = barrett64(raw + io.aux_half, io.aux_q, io.aux_ratio_hi);
                t = barrett64(t, io.q, io.ratio_hi);
                return f64_corr(f64_from_u64(t) - io.hm_d, m.m);
            }
            double t = f64_from_u64(raw) + io.aux_half_d;          // x < aux: t < 1.5 aux < 2^51, exact
            t = (t >= io.aux_qd) ? t - io.aux_qd : t;
            return f64_corr(t - io.hm_d, m.m);
        }
    }
    // fused epilogue: the transform output stays a centred double through the wave transpose
    template <int SM> static __device__ __forceinline__ u64 store_prep(elem x, const Mod& m) {
        if constexpr (SM == NTT_STORE_PLAIN) return final_fwd(x, m);
        else return f64_double_to_bits(f64_corr(x, m.m));
    }
    template <int SM> static __device__ __forceinline__ u64 store_io(const NttIo& io, u64 word, u64 prod, u64 addend, u64 destv, const Mod& m) {
        if constexpr (SM == NTT_STORE_PLAIN) return word;
        else {
            // (prod - y) * inv mod p  [+ dest] [+ addend]; |prod - y| <= 1.5p + 1, every term an exact integer
            const double y = f64_bits_to_double(word);
            double d = f64_mulc(f64_from_u64(prod) - y, io.inv_d, io.inv_pd, m.m.p);
            if constexpr (SM == NTT_STORE_KS_FINISH) {
                if (io.add_inplace) d += f64_from_u64(destv);
                if (io.ext1) d += f64_from_u64(addend);
            }
            return f64_canon(d, m.m);
        }
    }
    // ---- fused chain (NttFused): every value an exact integer; products by f64_mulq (|result| <= 0.69 p for a re-centred factor) ----
    // canonical x, y < p: no re-centring needed in front of the product, |result| <= 0.875 p
    static __device__ __forceinline__ elem prod_in(u64 x, u64 y, const Mod& m) {
        return f64_mulq(f64_from_u64(x), f64_from_u64(y), m.m.inv_p, m.m.p);
    }
    // c_k at one coefficient: c_0 = a0 b0, c_1 = a0 b1 + a1 b0
    static __device__ __forceinline__ elem tensor_term(unsigned k, u64 a0, u64 a1, u64 b0, u64 b1, const Mod& m) {
        return k == 0 ? prod_in(a0, b0, m) : prod_in(a0, b1, m) + prod_in(a1, b0, m);
    }
    // rounding fix of a coefficient x of a dropped prime `aux` (x < aux < 2^50): ((x + aux/2) mod aux) - (aux/2 mod p), re-centred
    static __device__ __forceinline__ elem round_fix(u64 x, double aux_q, double aux_half, double hm, const Mod& m) {
        double t = f64_from_u64(x) + aux_half;
        t = (t >= aux_q) ? t - aux_q : t;
        return f64_corr(t - hm, m.m);
    }
    // T = (x + floor(p/2)) mod p of a canonical x as an exact double: the modulus-independent half of a rounding fix
    // (ski_util6 / divide_and_round_q_last step 1), stored by the producer so that every consumer limb only subtracts its constant
    static __device__ __forceinline__ double round_half(double x_canon, const Mod& m) {
        const double t = x_canon + 0.5 * (m.m.p - 1.0);
        return t >= m.m.p ? t - m.m.p : t;
    }
    static __device__ __forceinline__ elem round_fix_t(u64 t_bits, double hm, const Mod& m) { return f64_corr(f64_bits_to_double(t_bits) - hm, m.m); }
    // the same for a T row that may hold u64 words of a prime of 2^50 or more (NTT_FUSED_*_W): T < 2^61 = hi 2^30 + lo, hi 2^30 is exact in a
    // double, so (hi 2^30 mod p) comes out of one quotient estimate and one exact fma; |.| <= p/2 + 2^30 before the constant is subtracted
    static __device__ __forceinline__ elem round_fix_tw(u64 raw, bool is_u64, double hm, const Mod& m) {
        double t;
        if (is_u64) {
            const double h = (double)(unsigned)(raw >> 30) * 1073741824.0;
            const double lo = (double)((unsigned)raw & 0x3fffffffu);
            t = __builtin_fma(-__builtin_rint(h * m.m.inv_p), m.m.p, h) + lo;
        } else t = f64_bits_to_double(raw);
        return f64_corr(t - hm, m.m);
    }
    static __device__ __forceinline__ elem from_canon(u64 x) { return f64_from_u64(x); }
    template <bool TW> static __device__ __forceinline__ elem tail_in(const NttIo& io, u64 raw_s, u64 raw_l, const Mod& m) {
        const elem rs = TW ? round_fix_tw(raw_s, io.ts_u64, io.hm_d, m) : round_fix_t(raw_s, io.hm_d, m);       // both rows arrive as T = (x + aux/2) mod aux
        const elem fl = TW ? round_fix_tw(raw_l, io.tl_u64, io.hm2_d, m) : round_fix_t(raw_l, io.hm2_d, m);
        return scale_by(rs, io.inv_d, m) + fl;      // |x| <= 1.2 p: a 4-layer block from here stays below 7.7 p < 2^53
    }
    static __device__ __forceinline__ u64 tail_out(const NttIo& io, u64 prod_word, elem y, const Mod& m) {
        // the product of a re-centred factor is within (-0.7 p, 0.7 p): one conditional add canonicalises it
        return canon_small(scale_by(f64_corr(f64_from_u64(prod_word) - y, m.m), io.inv2_d, m), m);
    }
    template <bool TW> static __device__ __forceinline__ u64 last_out(const NttIo& io, elem xr, bool scaled, u64 raw_s, const Mod& m) {
        const elem ys = scaled ? xr : f64_mulc(xr, m.ninv, m.ninv_p, m.m.p);          // scaled: the folded final layer applied N^-1 already
        const elem rs = TW ? round_fix_tw(raw_s, io.ts_u64, io.hm_d, m) : round_fix_t(raw_s, io.hm_d, m);
        // stored as T_l = (l + ql/2) mod ql (double): what the rescale's rounding fix of every remaining limb starts from
        double lc = f64_corr(ys - scale_by(rs, io.inv_d, m), m.m);
        lc = lc < 0.0 ? lc + m.m.p : lc;
        return f64_double_to_bits(round_half(lc, m));
    }
    static __device__ __forceinline__ u64 canon_small(elem x, const Mod& m) { return f64_to_u64(x < 0.0 ? x + m.m.p : x); }   // |x| < p
    static __device__ __forceinline__ elem scale_by(elem x, double inv_d, const Mod& m) { return f64_mulq(x, inv_d, m.m.inv_p, m.m.p); }   // |x| <= p
    static __device__ __forceinline__ elem load_mid(u64 raw, const Mod& m) { return f64_corr(f64_bits_to_double(raw), m.m); }
    static __device__ __forceinline__ u64 store_mid(elem x, const Mod&) { return f64_double_to_bits(x); }
    static __device__ __forceinline__ elem from_lds(u64 raw) { return f64_bits_to_double(raw); }
    static __device__ __forceinline__ u64 to_lds(elem x, const Mod& m) { return f64_double_to_bits(f64_corr(x, m.m)); }
    static __device__ __forceinline__ elem mid_fix(elem x, const Mod& m) { return f64_corr(x, m.m); }
    // f64_mulq: the quotient comes from the rounded product itself, so no w/p has to be formed per twiddle (w.y is never read and
    // the compiler drops its multiply)
    static __device__ __forceinline__ void fwd(elem& a, elem& b, const tw_t w, const Mod& m) {
        const double r = f64_mulq(b, w.x, m.m.inv_p, m.m.p);
        const double u = a;
        a = u + r;
        b = u - r;
    }
    static __device__ __forceinline__ void inv(elem& a, elem& b, const tw_t w, const Mod& m) {
        const double u = a, v = b;
        a = u + v;
        b = f64_mulq(u - v, w.x, m.m.inv_p, m.m.p);
    }
    static __device__ __forceinline__ u64 final_fwd(elem v, const Mod& m) { return f64_canon(v, m.m); }
    static __device__ __forceinline__ u64 final_inv(elem v, const Mod& m) { return f64_canon(f64_mulc(v, m.ninv, m.ninv_p, m.m.p), m.m); }
    // register hand-over between a forward and an inverse transform (tensor_core_kernel): re-centred doubles; a product of two
    // re-centred factors is an exact integer of magnitude <= 0.69 p, a sum of two is re-centred again before the inverse butterflies
    static __device__ __forceinline__ elem keep(elem x, const Mod& m) { return f64_corr(x, m.m); }
    static __device__ __forceinline__ elem prod(elem x, elem y, const Mod& m) { return f64_mulq(x, y, m.m.inv_p, m.m.p); }
    static __device__ __forceinline__ elem sum(elem x, elem y, const Mod&) { return x + y; }
    static __device__ __forceinline__ elem inv_in(elem x, const Mod& m) { return f64_corr(x, m.m); }
};

// Which local index bits identify the wave (thread bits >= 6) when the register window starts at bit S.
// If two consecutive rounds have the same set, a wave reads back exactly the LDS words it wrote itself
// (register <-> lane transpose), so that exchange needs no workgroup barrier at all.
__host__ __device__ constexpr unsigned ntt_wave_bits(int S, int EB, int TB) {
    unsigned m = 0;
    for (int p = 6; p < TB - EB; ++p) m |= 1u << (p < S ? p : p + EB);
    return m;
}

// One pass over the transform bits [LOGN-LO-G, LOGN-LO) of a 2^LOGN-point transform.
//   tile = all 2^G values of those bits x 2^C consecutive low indices, C = TB - G;
//   local index = (mid << C) | low, transform bits = local bits [C, TB).
//   forward: layers LO .. LO+G-1 (Cooley-Tukey, high bit first)
//   inverse: the matching Gentleman-Sande layers, low bit first
// Round r keeps local bits [S, S+EB) in registers and runs the butterflies of the transform
// bits [BLO, BHI] that fall inside that window.
// FIRST/LAST mark the first/last pass of the whole transform (prologue / final correction).
//
// KSMAC (ks_mac_kernel): the workgroup owns output row k of one item of the key switch and loops over the L digits:
// digit j is reduced mod q_key(k) while loading, transformed, and multiplied into two register accumulators with
// the matching limbs of key j (fgk/switch_key.cu:6-54 + :83-154 in one pass); the (L+1)*L transformed digits never
// reach HBM.  Digit k of row k < L is the untouched NTT-form input limb (evaluator_keyswitching_core.cu:851-852).
// IOM selects the fused element-wise prologue / epilogue at compile time: 0 plain, 1 key-switch tail
// (NTT_LOAD_KS_ROUND on the first pass, NTT_STORE_KS_FINISH on the last), 2 rescale.
__host__ __device__ constexpr bool ROUNDS_OK(int G, int EB) { return (G + EB - 1) / EB > 1; }

// REGIO (tensor_core_kernel): 1 = a forward last pass leaves its E consecutive outputs per thread in xio (A::keep form) instead of
// storing them, 2 = an inverse first pass takes its E consecutive inputs per thread from xio (A::inv_in form) instead of loading them.
// HALF: the LDS tile holds 32-bit words (half the bytes): every exchange moves the low halves, then the high halves of its E words
// (three barriers instead of one).  A whole-limb N = 16384 tile then takes 66 KB instead of 132 KB and TWO 1024-thread workgroups
// share a CU, so one can load / store while the other computes -- what N = 8192 gets for free.
template <class A, int LOGN, int LO, int G, int TB, int EB, bool INV, bool FIRST, bool LAST, bool KSMAC, int IOM, int REGIO = 0, bool HALF = false, bool SHFL = false>
__device__ __forceinline__ void ntt_pass_body(const NttArgs& a, const KeyPtrs* keys, u64* lds, unsigned bid, unsigned t, typename A::elem* xio = nullptr) {
    constexpr int C = TB - G;
    static_assert(!HALF || (!KSMAC && REGIO == 0), "half-word LDS tiles: plain and fused transform kernels only");
    static_assert(REGIO == 0 || (!KSMAC && IOM == 0 && C == 0 && (G + EB - 1) / EB > 1 && (REGIO == 1 ? (!INV && LAST) : (INV && FIRST))), "register hand-over: last forward / first inverse pass on whole tiles");
    constexpr int E = 1 << EB;
    constexpr unsigned N = 1u << LOGN;
    constexpr int TILE_BITS = LOGN - TB;               // tiles per limb-polynomial = 2^TILE_BITS
    constexpr int NLB = LOGN - LO - G - C;             // low-block bits in the tile id
    static_assert(G >= 1 && C >= 0 && NLB >= 0 && TB >= EB, "bad NTT pass shape");
    constexpr int ROUNDS = (G + EB - 1) / EB;
    static_assert(!KSMAC || (!INV && FIRST && LAST && C == 0 && LO == 0 && ROUNDS > 1), "KSMAC needs a whole-limb forward tile");
    using elem = typename A::elem;
    using tw_t = typename A::tw_t;

    unsigned tile, j, k, b;
    if constexpr (KSMAC) {
        // row-major over the launch: consecutive workgroups (dealt round-robin to the 8 XCDs) work on the same
        // output row, so that row's 2*L key limbs (L2-sized) stay hot in every XCD while the digits stream through
        tile = 0; j = 0;
        const unsigned R = a.xcd_groups;        // rows co-scheduled per XCD (0 / 1: plain row-major)
        if (R > 1) {
            // R rows of the same item land on one XCD back to back (workgroup ids 8 apart), so the item's digits are fetched once
            // for the R rows, while the launch still walks the rows R at a time and those rows' keys (R * 2L limbs) stay in L2
            const unsigned per = 8u * R, phase = a.batch * R;
            const unsigned q = bid % phase, r = q % per;
            k = (bid / phase) * R + r / 8u;
            b = (q / per) * 8u + (r % 8u);
        } else {
            b = bid % a.batch; k = bid / a.batch;
        }
        if (a.ks_row_mask) k = nth_set_bit(a.ks_row_mask, k);
    } else {
        tile = bid & ((1u << TILE_BITS) - 1); bid >>= TILE_BITS;
        unsigned g;   // (batch, poly) group
        if (IOM != 0 && TILE_BITS == 0 && a.xcd_groups) {
            // The ncomp limbs of a group read the SAME input row (fused key-switch tail / rescale).  Workgroups are
            // dealt round-robin to the 8 XCDs, so place a group's limbs 8 apart: they land on one XCD, back to back,
            // and the shared row is fetched from HBM once instead of ncomp times.
            const unsigned per = 8u * a.ncomp, full = (a.xcd_groups / 8u) * per;
            if (bid < full) { const unsigned r = bid % per; j = r / 8u; g = (bid / per) * 8u + (r % 8u); }
            else { const unsigned r = bid - full; j = r % a.ncomp; g = (a.xcd_groups / 8u) * 8u + r / a.ncomp; }
        } else {
            j = bid % a.ncomp; g = bid / a.ncomp;
        }
        k = g % a.pcount;
        b = g / a.pcount;
    }
    const unsigned top = tile >> NLB;
    const unsigned lb = tile & ((1u << NLB) - 1);

    if (!KSMAC && a.skip_diag && k == j && k < a.decomp) return;   // consumer reads the original limb (see ks_accumulate_kernel)
    const unsigned mi = KSMAC ? a.table_start + ((k == a.decomp) ? a.table_count - 1 : k) : ntt_table_index(a, k, j);
    const typename A::Mod md = A::make(a.mods[mi]);
    // twiddle tables are never written by a kernel: read them through the constant address space so that
    // wave-uniform fetches become scalar loads and stay out of the vector-memory queue
    typedef typename A::tw_mem tw_mem;
    typedef const tw_mem __attribute__((address_space(4)))* ctw_ptr;
    unsigned long long tw_addr = (unsigned long long)(reinterpret_cast<const tw_mem*>(a.tw) + (size_t)mi * N);
    ctw_ptr twc = (ctw_ptr)tw_addr;
    auto tw_load = [&](unsigned idx) -> tw_t { const tw_mem v = twc[idx]; return A::tw_from_mem(v, md); };
    const u64* __restrict__ gin = a.in + (long long)b * a.in_bstride + (KSMAC ? 0ll : (long long)k * a.in_pstride + (long long)j * a.in_cstride);
    u64* __restrict__ gout = a.out + (long long)b * a.out_bstride + (KSMAC ? (long long)k * a.out_cstride : (long long)k * a.out_pstride + (long long)j * a.out_cstride);

    auto gindex = [&](unsigned loc) -> unsigned {
        return (top << (LOGN - LO)) | ((loc >> C) << (LOGN - LO - G)) | (lb << C) | (loc & ((1u << C) - 1));
    };
    // gindex(x | y) = gindex(x) + gpart(y) for disjoint bit sets: the part of the index that a register number R contributes
    // is a compile-time constant
    auto gpart = [](unsigned loc) constexpr -> unsigned { return ((loc >> C) << (LOGN - LO - G)) | (loc & ((1u << C) - 1)); };
    constexpr int LM = (FIRST && !KSMAC && !INV) ? (IOM == 1 ? (int)NTT_LOAD_KS_ROUND : IOM == 2 ? (int)NTT_LOAD_RESCALE : IOM == NTT_IOM_CENTRALIZE ? (int)NTT_LOAD_CENTRALIZE : (int)NTT_LOAD_PLAIN) : (int)NTT_LOAD_PLAIN;
    constexpr int SM = (LAST && !KSMAC && !INV) ? (IOM == 1 ? (int)NTT_STORE_KS_FINISH : IOM == 2 ? (int)NTT_STORE_RESCALE : (int)NTT_STORE_PLAIN) : (int)NTT_STORE_PLAIN;
    constexpr bool FUSED = IOM >= 3 && IOM <= 7;     // NttFused: kernels of the multiply -> relinearize -> rescale chain (loaders act in the first pass
                                         // of a transform, epilogues in its last pass: one kernel for N <= 16384, two for N = 32768)
    constexpr bool F_MULPAIR = IOM == NTT_FUSED_MULPAIR && FIRST;
    constexpr bool F_LAST = IOM == NTT_FUSED_LAST_LIMB || IOM == NTT_FUSED_LAST_LIMB_W, F_TR = IOM == NTT_FUSED_TAIL_RESCALE || IOM == NTT_FUSED_TAIL_RESCALE_W;
    constexpr bool TW = IOM == NTT_FUSED_LAST_LIMB_W || IOM == NTT_FUSED_TAIL_RESCALE_W;     // FP64 policy: the T rows may hold u64 words
    constexpr bool F_LAST_LD = F_LAST && FIRST, F_LAST_ST = F_LAST && LAST;
    constexpr bool F_TR_LD = F_TR && FIRST, F_TR_ST = F_TR && LAST;
    // coefficient-form key-switch tail (BFV): the inverse transform of a data row ends with ski_util6_merged + ski_util7_merged
    // (evaluator_keyswitching_core.cu:570-658) -- the rounding fix is formed from the INTT of the special-prime row (in2) at the same
    // coefficient, then (this + lift - fix) qk^-1 [+ dest] [+ addend] is stored
    constexpr bool F_KC_ST = INV && LAST && IOM == 1 && !KSMAC;
    static_assert(!FUSED || !KSMAC, "fused chain: transform kernels");
    static_assert(!TW || std::is_same<A, ArithF64>::value, "the _W variants exist for the FP64 policy (the integer forms read either kind of T row)");
    static_assert(!FUSED || (INV == (IOM == NTT_FUSED_MULPAIR || F_LAST)), "fused chain: transform direction");
    NttIo io;
    if constexpr (FUSED) { io.load_mode = NTT_LOAD_PLAIN; io.store_mode = NTT_STORE_PLAIN; ntt_io_fused(io, a, b, k, j, mi); }
    else if constexpr (LM != NTT_LOAD_PLAIN || SM != NTT_STORE_PLAIN) io = ntt_io_make(a, b, k, j, mi, gout);
    else if constexpr (F_KC_ST) { io = ntt_io_make(a, b, k, j, mi, gout); io.in2 = a.in2 + (long long)b * a.in2_bstride + (long long)k * a.in2_pstride; }
    else { io.load_mode = NTT_LOAD_PLAIN; io.store_mode = NTT_STORE_PLAIN; }
    // this workgroup's limb of the two input ciphertexts (fused chain)
    const long long mul_off = FUSED ? (long long)(a.mul_limb0 + j) * N : 0ll;

    elem x[E];
    elem acc0[KSMAC ? E : 1], acc1[KSMAC ? E : 1];
    const u64* __restrict__ key0 = nullptr;   // this iteration's key, component 0, limb mi
    const u64* __restrict__ kq0 = nullptr;    // its Shoup quotients (NttArgs::key_quo), or null
    if constexpr (KSMAC) {
        static_for<0, E>([&](auto Rc) { acc0[decltype(Rc)::value] = A::mac_zero(); acc1[decltype(Rc)::value] = A::mac_zero(); });
    }
    const u64* const gin0 = gin;
    auto one_digit = [&](unsigned it) {
    if constexpr (KSMAC) {
        // neither the twiddles nor the LDS addresses depend on the digit: without these the compiler hoists every
        // twiddle load (with its w/p product) and every index computation out of the digit loop and spills them
        asm volatile("" : "+v"(t));
        unsigned tw_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tw_addr);
        unsigned tw_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(tw_addr >> 32));
        asm volatile("" : "+s"(tw_lo), "+s"(tw_hi));
        twc = (ctw_ptr)(((unsigned long long)tw_hi << 32) | tw_lo);
        gin = gin0 + (long long)it * a.in_cstride;
        key0 = keys->p[it] + (size_t)(mi - a.table_start) * N;
        kq0 = a.key_quo ? a.key_quo + (long long)it * a.key_quo_jstride + (size_t)(mi - a.table_start) * N : nullptr;
        if (a.skip_diag && it == k) {
            // own digit: already in NTT form under this modulus; read it in the accumulators' (transposed) layout
            const unsigned lane = t & 63u, wbase = (t >> 6) * (64u * E);
            const u64* dg = a.ext0 + (long long)b * a.ext0_bstride + (long long)k * a.ext0_cstride;
            static_for<0, E / 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const unsigned idx = wbase + m * 128u + lane * 2u;
                const ulonglong2 v = nt_load2(dg + idx);
                const ulonglong2 k0 = *reinterpret_cast<const ulonglong2*>(key0 + idx);
                const ulonglong2 k1 = *reinterpret_cast<const ulonglong2*>(key0 + a.key_pstride + idx);
                const elem v0 = A::mac_in(v.x, md), v1 = A::mac_in(v.y, md);
                if constexpr (A::HAS_MAC_SHOUP) { if (kq0) {
                    const ulonglong2 q0 = *reinterpret_cast<const ulonglong2*>(kq0 + idx), q1 = *reinterpret_cast<const ulonglong2*>(kq0 + a.key_pstride + idx);
                    A::mac_shoup(acc0[2 * m], v0, k0.x, q0.x, md); A::mac_shoup(acc0[2 * m + 1], v1, k0.y, q0.y, md);
                    A::mac_shoup(acc1[2 * m], v0, k1.x, q1.x, md); A::mac_shoup(acc1[2 * m + 1], v1, k1.y, q1.y, md);
                    __builtin_amdgcn_sched_barrier(0);
                    return;
                } }
                A::mac(acc0[2 * m], v0, k0.x, md); A::mac(acc0[2 * m + 1], v1, k0.y, md);
                A::mac(acc1[2 * m], v0, k1.x, md); A::mac(acc1[2 * m + 1], v1, k1.y, md);
                __builtin_amdgcn_sched_barrier(0);   // keep the key loads of later pairs from piling up in registers
            });
            return;
        }
    }
    // Exchange of E words through the tile: word i of this thread goes to wa(i), its new word i comes from ra(i); `sync` is the
    // barrier that separates the writes from the reads (workgroup, or wave when the exchange stays inside the wave's own slice).
    // No barrier is needed in front: a thread writes exactly the words it read itself in its previous exchange.
    unsigned* const lds32 = reinterpret_cast<unsigned*>(lds);
    // `after_write` runs between the (first) write phase and its barrier: loads that should be in flight during the exchange
    auto xchg = [&](auto wa, auto ra, u64 (&w)[E], auto sync, auto after_write) {
        if constexpr (!HALF) {
            static_for<0, E>([&](auto ic) { lds[wa(ic)] = w[decltype(ic)::value]; });
            after_write();
            sync();
            static_for<0, E>([&](auto ic) { w[decltype(ic)::value] = lds[ra(ic)]; });
        } else {
            // the incoming low halves take the place of the outgoing ones (dead once written): no extra registers
            static_for<0, E>([&](auto ic) { lds32[wa(ic)] = (unsigned)w[decltype(ic)::value]; });
            after_write();
            sync();
            static_for<0, E>([&](auto ic) { constexpr int i = decltype(ic)::value; w[i] = (w[i] & 0xffffffff00000000ull) | lds32[ra(ic)]; });
            sync();
            static_for<0, E>([&](auto ic) { lds32[wa(ic)] = (unsigned)(w[decltype(ic)::value] >> 32); });
            sync();
            static_for<0, E>([&](auto ic) { constexpr int i = decltype(ic)::value; w[i] = ((u64)lds32[ra(ic)] << 32) | (unsigned)w[i]; });
        }
    };
    auto nothing = [] {};
    auto wave_sync = [] { __builtin_amdgcn_wave_barrier(); };
    auto wg_sync = [] { __syncthreads(); };
    (void)wg_sync;
    static_for<0, ROUNDS>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        // transform bits handled this round, and the register window [S, S+EB)
        constexpr int BHI = INV ? ((C + (r + 1) * EB - 1 < TB - 1) ? C + (r + 1) * EB - 1 : TB - 1) : TB - 1 - r * EB;
        constexpr int BLO = INV ? C + r * EB : ((TB - (r + 1) * EB > C) ? TB - (r + 1) * EB : C);
        constexpr int S = INV ? ((BLO < TB - EB) ? BLO : TB - EB) : ((TB - (r + 1) * EB > 0) ? TB - (r + 1) * EB : 0);
        static_assert(S >= 0 && S + EB <= TB && BLO >= S && BHI < S + EB && BLO <= BHI, "bad round window");
        const unsigned tlow = t & ((1u << S) - 1);
        const unsigned locbase = tlow | ((t >> S) << (S + EB));
        const unsigned pbase = lds_phys(locbase);
        // the twiddle index only sees bits above the butterfly bit; when S >= 6 those are wave-uniform
        const unsigned twbase = (S >= 6) ? ((unsigned)__builtin_amdgcn_readfirstlane((int)(t >> S)) << (S + EB)) : locbase;

        // window of the previous / next round (for the wave-private exchange test)
        constexpr int S_PREV = (r == 0) ? S : (INV ? ((C + (r - 1) * EB < TB - EB) ? C + (r - 1) * EB : TB - EB)
                                                   : ((TB - r * EB > 0) ? TB - r * EB : 0));
        constexpr int S_NEXT = (r == ROUNDS - 1) ? S : (INV ? ((C + (r + 1) * EB < TB - EB) ? C + (r + 1) * EB : TB - EB)
                                                            : ((TB - (r + 2) * EB > 0) ? TB - (r + 2) * EB : 0));
        constexpr bool PRIVATE_IN = (r > 0) && ntt_wave_bits(S_PREV, EB, TB) == ntt_wave_bits(S, EB, TB);
        constexpr bool PRIVATE_OUT = (r < ROUNDS - 1) && ntt_wave_bits(S, EB, TB) == ntt_wave_bits(S_NEXT, EB, TB);
        (void)PRIVATE_IN;

        if constexpr (REGIO == 2 && r == 0) {
            static_assert(r != 0 || REGIO != 2 || S == 0, "register hand-over: E consecutive coefficients per thread");
            static_for<0, E>([&](auto Rc) { x[decltype(Rc)::value] = xio[decltype(Rc)::value]; });
        } else if constexpr (r == 0 && INV && S == 0 && C == 0 && ROUNDS > 1) {
            // Mirror image of the forward store transpose: a thread starts with E consecutive coefficients.  Loading
            // them directly makes every load instruction touch 64 different 128-byte lines; instead the wave loads
            // its 64*E consecutive words with 16 bytes per lane, parks them in its own LDS slice (the words it will
            // overwrite itself in the first exchange) and picks its coefficients from there.
            const unsigned lane = t & 63u, wbase = (t >> 6) * (64u * E);
            const unsigned pown = lds_phys(wbase + lane * E), pidx = lds_phys(wbase + lane * 2u);   // own E words / 16-byte pairs
            const unsigned gbase = gindex(wbase);
            u64 wv[E];
            static_for<0, E / 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const unsigned idx = m * 128u + lane * 2u;
                if constexpr (F_MULPAIR) {
                    // c2 = a1 (.) b1 formed while loading (the tensor product is never written to HBM)
                    const ulonglong2 va = ld2_at(io.a1 + mul_off + gbase + m * 128u, lane * 16u, true), vb = ld2_at(io.b1 + mul_off + gbase + m * 128u, lane * 16u, true);
                    wv[2 * m] = A::to_lds(A::prod_in(va.x, vb.x, md), md);
                    wv[2 * m + 1] = A::to_lds(A::prod_in(va.y, vb.y, md), md);
                } else if constexpr (F_LAST_LD) {
                    // Q = P qk^-1 + c_k at the dropped limb, as ksmac2 left it (KsMacArgs::ten_a)
                    const ulonglong2 vp = nt_load2(gin + gbase + idx);
                    wv[2 * m] = A::to_lds(A::from_canon(vp.x), md);
                    wv[2 * m + 1] = A::to_lds(A::from_canon(vp.y), md);
                } else {
                    // (uniform row + pair offset) + one 32-bit lane offset: no 64-bit address per load
                    const ulonglong2 v = ld2_at(gin + gbase + m * 128u, lane * 16u, a.stream_loads != 0);
                    wv[2 * m] = ntt_io_load<LM>(io, v.x);
                    wv[2 * m + 1] = ntt_io_load<LM>(io, v.y);
                }
            });
            xchg([&](auto ic) { constexpr int i = decltype(ic)::value; return pidx + lds_off((i / 2) * 128u) + (i & 1); },
                 [&](auto ic) { return pown + (unsigned)decltype(ic)::value; }, wv, wave_sync, nothing);
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                const u64 raw = wv[R];
                if constexpr (F_MULPAIR || F_LAST_LD) x[R] = A::from_lds(raw);     // re-centred when it was parked
                else if constexpr (FIRST) x[R] = A::load_first(raw, a.reduce_input != 0, md);
                else x[R] = A::load_mid(raw, md);
            });
        } else if constexpr (r == 0) {
            const unsigned lb0 = gindex(locbase) * 8u;     // one 32-bit lane offset; the register number only moves the (uniform) base
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                constexpr unsigned GR = gpart((unsigned)R << S);
                u64 raw;
                if constexpr (KSMAC) raw = a.stream_loads ? nt_load(gin + gindex(locbase | ((unsigned)R << S))) : gin[gindex(locbase | ((unsigned)R << S))];
                else if constexpr (LM == NTT_LOAD_CENTRALIZE) raw = ((lb0 >> 3) + GR < io.cz_count) ? ld_at(gin + GR, lb0) : 0ull;      // a plaintext shorter than N: zeros are not read
                else raw = a.stream_loads ? nt_ld_at(gin + GR, lb0) : ld_at(gin + GR, lb0);
                if constexpr (KSMAC) x[R] = A::load_first(raw, true, md);
                else if constexpr (F_TR_LD) {
                    // r_j(s) qk^-1 + f_j(l): the rounding fixes of the key switch and of the rescale enter ONE transform
                    const u64 raw2 = ld_at(io.in2 + GR, lb0);
                    x[R] = A::template tail_in<TW>(io, raw, raw2, md);
                }
                else if constexpr (FIRST) x[R] = A::template load_io<LM>(io, raw, a.reduce_input != 0, md);
                else x[R] = A::load_mid(raw, md);
            });
        } else {
            // exchange with the previous round: the words leave from that round's register window (S_PREV) and arrive in this one's
            const unsigned pprev = lds_phys((t & ((1u << S_PREV) - 1)) | ((t >> S_PREV) << (S_PREV + EB)));
            u64 wv[E];
            static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; wv[R] = A::to_lds(x[R], md); });
            auto wa = [&](auto ic) { return pprev + lds_off((unsigned)decltype(ic)::value << S_PREV); };
            auto ra = [&](auto ic) { return pbase + lds_off((unsigned)decltype(ic)::value << S); };
            if constexpr (SHFL && PRIVATE_IN && !INV && S_PREV == 2 && S == 0 && EB == 4 && std::is_same<A, ArithF64>::value) {
                // Measured alternative (tools/ksbench `ntt`, profiles/r03_ntt_ab.txt; not used by the library): the wave-private exchange in
                // front of the last forward round as a register <-> lane transpose.  Index bits (0, 1) sit in lane bits (0, 1) and must
                // become register bits, index bits (4, 5) sit in register bits (2, 3) and must become lane bits (0, 1): two DPP quad_perm
                // swaps (lane bit 0 <-> register bit 2, lane bit 1 <-> register bit 3) per 64-bit word pair, no LDS.
                double y[E];
                static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; y[R] = A::from_lds(wv[R]); });
                auto dpp64 = [](double v, auto ctrl) {
                    constexpr int CT = decltype(ctrl)::value;
                    const u64 bits = f64_double_to_bits(v);
                    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)bits, CT, 0xF, 0xF, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(bits >> 32), CT, 0xF, 0xF, false);
                    return f64_bits_to_double(((u64)(unsigned)hi << 32) | (unsigned)lo);
                };
                static_for<0, 2>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;                       // lane bit st <-> register bit 2 + st
                    constexpr int CT = st == 0 ? 0xB1 : 0x4E;                     // quad_perm [1,0,3,2] / [2,3,0,1]
                    const bool odd = ((t >> st) & 1u) != 0;
                    static_for<0, E>([&](auto Rc) {
                        constexpr int R = decltype(Rc)::value;
                        if constexpr (((R >> (2 + st)) & 1) == 0) {
                            constexpr int R1 = R | (1 << (2 + st));
                            const double pa = dpp64(y[R], std::integral_constant<int, CT>{}), pb = dpp64(y[R1], std::integral_constant<int, CT>{});
                            const double na = odd ? pb : y[R], nb = odd ? y[R1] : pa;
                            y[R] = na; y[R1] = nb;
                        }
                    });
                });
                static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; x[(R & 3) << 2 | (R >> 2)] = y[R]; });
            } else {
            // RAW: readers are the writer's own wave (LDS executes a wave's accesses in order) or other waves
            if constexpr (PRIVATE_IN) xchg(wa, ra, wv, wave_sync, nothing); else xchg(wa, ra, wv, wg_sync, nothing);
            static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; x[R] = A::from_lds(wv[R]); });
            }
        }

        constexpr int NLAYERS = BHI - BLO + 1;
        unsigned tw_dep = 0;      // see below
        static_for<0, NLAYERS>([&](auto lc) {
            // forward: highest bit first; inverse: lowest bit first
            constexpr int li = decltype(lc)::value;
            constexpr int bit = INV ? BLO + li : BHI - li;
            constexpr int rb = bit - S;            // register bit
            constexpr int kk = TB - 1 - bit;       // layer inside the tile
            constexpr int l = LO + kk;             // global (forward-numbered) layer of this bit
            if constexpr (KSMAC || (HALF && !INV)) __builtin_amdgcn_sched_barrier(0);   // layer by layer: bounds the live twiddles
            if constexpr (HALF && INV && r == 0 && std::is_same<A, ArithF64>::value && (decltype(lc)::value == 1 || decltype(lc)::value == 2)) {
                // first inverse round, per-lane twiddles: the 8 twiddles of layer 0 and the 7 of layers 1..3 do not fit next to the 16
                // coefficients in 64 registers.  An opaque zero derived from a layer-0 result is added to the later layers' table index, so
                // their loads cannot be hoisted above layer 0 (a scheduling barrier here costs more registers than it saves).
                const unsigned lo = (unsigned)f64_double_to_bits(x[0]);
                asm volatile("v_and_b32 %0, 0, %1" : "=v"(tw_dep) : "v"(lo));
            }
            static_for<0, (E >> (rb + 1))>([&](auto hc) {
                constexpr int hi = decltype(hc)::value;
                const unsigned loc0 = twbase | ((unsigned)(hi << (rb + 1)) << S);
                const unsigned grp = (top << kk) + (loc0 >> (bit + 1));
#ifdef TROYN_ABLATE_NO_TWIDDLE
                const tw_t w = tw_load(1u + (grp & 1u));
#else
                const tw_t w = tw_load((INV ? N - (2u << l) + 1 + grp : (1u << l) + grp) + tw_dep);
#endif
                static_for<0, (1 << rb)>([&](auto oc) {
                    constexpr int R0 = (hi << (rb + 1)) | decltype(oc)::value;
                    constexpr int R1 = R0 | (1 << rb);
#ifdef TROYN_ABLATE_NO_BUTTERFLY
                    x[R0] = x[R0] + x[R1]; (void)w;
#else
                    if constexpr (!INV) A::fwd(x[R0], x[R1], w, md);
                    else {
                        if constexpr (A::FOLD_NINV && LAST && l == 0) { A::inv_fold(x[R0], x[R1], md); (void)w; }
                        else A::inv(x[R0], x[R1], w, md);
                        if constexpr (A::MID_FIX && NLAYERS == 4 && li == 1) x[R0] = A::mid_fix(x[R0], md);
                    }
#endif
                });
            });
        });

        if constexpr (REGIO == 1 && r == ROUNDS - 1) {
            static_assert(r != ROUNDS - 1 || REGIO != 1 || S == 0, "register hand-over: E consecutive coefficients per thread");
            static_for<0, E>([&](auto Rc) { xio[decltype(Rc)::value] = A::keep(x[decltype(Rc)::value], md); });
        } else if constexpr (KSMAC && r == ROUNDS - 1) {
            // transpose inside the wave's own LDS slice (see the store path below), then multiply-accumulate with
            // 16-byte coalesced key loads
            const unsigned lane = t & 63u, wbase = (t >> 6) * (64u * E);
            const unsigned pown = lds_phys(wbase + lane * E), pidx = lds_phys(wbase + lane * 2u);   // own E words / 16-byte pairs
            // key loads are software-pipelined one pair ahead: the first pair is requested before the transpose, pair
            // m+1 before the arithmetic of pair m (two pairs = 16 VGPRs in flight; more would spill)
            const u64* kp0 = key0 + wbase + lane * 2u;
            const u64* kp1 = kp0 + a.key_pstride;
            ulonglong2 kc0 = *reinterpret_cast<const ulonglong2*>(kp0), kc1 = *reinterpret_cast<const ulonglong2*>(kp1);
            // Shoup quotients of the same words (integer policy, NttArgs::key_quo): pipelined like the keys
            const bool shoup = A::HAS_MAC_SHOUP && kq0 != nullptr;
            const u64* qp0 = shoup ? kq0 + wbase + lane * 2u : kp0;
            const u64* qp1 = qp0 + a.key_pstride;
            ulonglong2 qc0 = make_ulonglong2(0, 0), qc1 = make_ulonglong2(0, 0);
            if constexpr (A::HAS_MAC_SHOUP) { if (shoup) { qc0 = *reinterpret_cast<const ulonglong2*>(qp0); qc1 = *reinterpret_cast<const ulonglong2*>(qp1); } }
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                lds[pown + R] = A::mac_to_lds(x[R], md);
            });
            __builtin_amdgcn_wave_barrier();
            static_for<0, E / 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                ulonglong2 kn0 = kc0, kn1 = kc1, qn0 = qc0, qn1 = qc1;
                if constexpr (m + 1 < E / 2) {
                    kn0 = *reinterpret_cast<const ulonglong2*>(kp0 + (m + 1) * 128u);
                    kn1 = *reinterpret_cast<const ulonglong2*>(kp1 + (m + 1) * 128u);
                    if constexpr (A::HAS_MAC_SHOUP) { if (shoup) {
                        qn0 = *reinterpret_cast<const ulonglong2*>(qp0 + (m + 1) * 128u);
                        qn1 = *reinterpret_cast<const ulonglong2*>(qp1 + (m + 1) * 128u);
                    } }
                }
                const elem v0 = A::from_lds(lds[pidx + lds_off(m * 128u)]), v1 = A::from_lds(lds[pidx + lds_off(m * 128u) + 1]);
                bool done = false;
                if constexpr (A::HAS_MAC_SHOUP) { if (shoup) {
                    A::mac_shoup(acc0[2 * m], v0, kc0.x, qc0.x, md); A::mac_shoup(acc0[2 * m + 1], v1, kc0.y, qc0.y, md);
                    A::mac_shoup(acc1[2 * m], v0, kc1.x, qc1.x, md); A::mac_shoup(acc1[2 * m + 1], v1, kc1.y, qc1.y, md);
                    done = true;
                } }
                if (!done) {
                    A::mac(acc0[2 * m], v0, kc0.x, md); A::mac(acc0[2 * m + 1], v1, kc0.y, md);
                    A::mac(acc1[2 * m], v0, kc1.x, md); A::mac(acc1[2 * m + 1], v1, kc1.y, md);
                }
                kc0 = kn0; kc1 = kn1; qc0 = qn0; qc1 = qn1;
                __builtin_amdgcn_sched_barrier(0);   // keep the key loads of later pairs from piling up in registers
            });
            // the next digit's first exchange overwrites every wave's slice
            __syncthreads();
        } else if constexpr (r == ROUNDS - 1 && !INV && LAST && S == 0 && C == 0 && ROUNDS > 1) {
            // A thread ends with E consecutive coefficients, a wave with 64*E.  Transpose them inside the wave's own
            // LDS slice (exactly the words this wave read in the last exchange, so no other wave is disturbed)
            // and store 16 bytes per lane to consecutive addresses: every store instruction writes 1 KiB of
            // consecutive bytes instead of touching 64 different 128-byte lines.
            const unsigned lane = t & 63u, wbase = (t >> 6) * (64u * E);
            const unsigned pown = lds_phys(wbase + lane * E), pidx = lds_phys(wbase + lane * 2u);   // own E words / 16-byte pairs
            u64 wv[E];
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                if constexpr (F_TR_ST) wv[R] = A::to_lds(x[R], md);       // stays a re-centred double through the transpose
                else wv[R] = A::template store_prep<SM>(x[R], md);
            });
            const unsigned gbase = gindex(wbase);
            // epilogue operands: request every word first (16 bytes per lane, all in flight together), then compute
            ulonglong2 e0[SM != NTT_STORE_PLAIN ? E / 2 : 1], e1[SM == NTT_STORE_KS_FINISH ? E / 2 : 1], ed[SM == NTT_STORE_KS_FINISH ? E / 2 : 1];
            auto request_operands = [&] {
                if constexpr (SM != NTT_STORE_PLAIN && !HALF) {     // half-word tiles: 64 registers per thread, operands are loaded pair by pair below
                    static_for<0, E / 2>([&](auto mc) {
                        constexpr int m = decltype(mc)::value;
                        e0[m] = *reinterpret_cast<const ulonglong2*>(io.ext0 + gbase + m * 128u + lane * 2u);
                    });
                    if constexpr (SM == NTT_STORE_KS_FINISH) {
                        static_for<0, E / 2>([&](auto mc) { e1[decltype(mc)::value] = make_ulonglong2(0, 0); ed[decltype(mc)::value] = make_ulonglong2(0, 0); });
                        if (io.ext1) static_for<0, E / 2>([&](auto mc) {
                            constexpr int m = decltype(mc)::value;
                            e1[m] = *reinterpret_cast<const ulonglong2*>(io.ext1 + gbase + m * 128u + lane * 2u);
                        });
                        if (io.add_inplace) static_for<0, E / 2>([&](auto mc) {
                            constexpr int m = decltype(mc)::value;
                            ed[m] = *reinterpret_cast<const ulonglong2*>(io.dest + gbase + m * 128u + lane * 2u);
                        });
                    }
                }
            };
            xchg([&](auto ic) { return pown + (unsigned)decltype(ic)::value; },
                 [&](auto ic) { constexpr int i = decltype(ic)::value; return pidx + lds_off((i / 2) * 128u) + (i & 1); }, wv, wave_sync, request_operands);
            static_for<0, E / 2>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const unsigned idx = m * 128u + lane * 2u;
                u64 v0 = wv[2 * m], v1 = wv[2 * m + 1];
                if constexpr (F_TR_ST) {
                    // (Q_kj - y) ql^-1 with Q_kj = P_j qk^-1 + c_kj as ksmac2 left it: relinearize's divide-and-add happened there, the
                    // rescale's divide happens here
                    const unsigned boff = (gbase + idx) * 8u;
                    const ulonglong2 pr = ld2_at(io.ext0, boff, true);
                    v0 = A::tail_out(io, pr.x, A::from_lds(v0), md);
                    v1 = A::tail_out(io, pr.y, A::from_lds(v1), md);
                }
                if constexpr (SM != NTT_STORE_PLAIN && HALF) {
                    const unsigned boff = (gbase + idx) * 8u;
                    const ulonglong2 z = make_ulonglong2(0, 0);
                    const ulonglong2 o0 = ld2_at(io.ext0, boff);
                    ulonglong2 o1 = z, od = z;
                    if constexpr (SM == NTT_STORE_KS_FINISH) {
                        if (io.ext1) o1 = ld2_at(io.ext1, boff);
                        if (io.add_inplace) od = ld2_at(io.dest, boff);
                    }
                    v0 = A::template store_io<SM>(io, v0, o0.x, o1.x, od.x, md);
                    v1 = A::template store_io<SM>(io, v1, o0.y, o1.y, od.y, md);
                } else if constexpr (SM != NTT_STORE_PLAIN) {
                    constexpr int mk = (SM == NTT_STORE_KS_FINISH) ? m : 0;
                    v0 = A::template store_io<SM>(io, v0, e0[m].x, e1[mk].x, ed[mk].x, md);
                    v1 = A::template store_io<SM>(io, v1, e0[m].y, e1[mk].y, ed[mk].y, md);
                }
                nt_store2(gout + gbase + idx, v0, v1);
            });
        } else if constexpr (r == ROUNDS - 1) {
            const unsigned gi0 = gindex(locbase), gb0 = gi0 * 8u;
            static_for<0, E>([&](auto Rc) {
                constexpr int R = decltype(Rc)::value;
                u64 v;
                constexpr unsigned GR = gpart((unsigned)R << S);     // this register's share of the index
                const unsigned gi = gi0 + GR;
                if constexpr (F_LAST_ST) {
                    // l = INTT(P qk^-1 + c) - r(s) qk^-1: the INTT of relinearize's last limb without ever forming that limb
                    constexpr bool scaled = A::FOLD_NINV && ((R >> (EB - 1)) & 1);      // the folded final layer applied N^-1 already
                    v = A::template last_out<TW>(io, x[R], scaled, io.in2[gi], md);
                } else if constexpr (F_KC_ST) {
                    u64 pw;
                    if constexpr (A::FOLD_NINV && ((R >> (EB - 1)) & 1)) pw = A::final_fwd(x[R], md);          // already scaled by N^-1
                    else pw = A::final_inv(x[R], md);
                    const elem fixv = A::template load_io<NTT_LOAD_KS_ROUND>(io, io.in2[gi], false, md);
                    v = A::template store_io<NTT_STORE_KS_FINISH>(io, A::store_mid(fixv, md), pw, io.ext1 ? io.ext1[gi] : 0, io.add_inplace ? io.dest[gi] : 0, md);
                } else if constexpr (LAST) {
                    if constexpr (INV && A::FOLD_NINV && ((R >> (EB - 1)) & 1)) v = A::final_fwd(x[R], md);   // already scaled
                    else if constexpr (INV) v = A::final_inv(x[R], md);
                    else if constexpr (SM == NTT_STORE_PLAIN) v = A::final_fwd(x[R], md);
                    else v = A::template store_io<SM>(io, A::template store_prep<SM>(x[R], md), io.ext0[gi],
                                                      (SM == NTT_STORE_KS_FINISH && io.ext1) ? io.ext1[gi] : 0,
                                                      (SM == NTT_STORE_KS_FINISH && io.add_inplace) ? io.dest[gi] : 0, md);
                    if constexpr (INV && IOM == 0 && std::is_same<A, ArithF64>::value) {
                        if (a.flags & NTT_FLAG_STORE_ROUND_HALF) v = f64_double_to_bits(ArithF64::round_half(f64_from_u64(v), md));
                    }
                    if constexpr (INV && IOM == 0 && std::is_same<A, ArithU64>::value) {
                        if (a.flags & NTT_FLAG_STORE_ROUND_HALF) v = add_mod(v, md.q >> 1, md.q);      // the same T as a u64 word (NTT_FLAG_TS_U64)
                    }
                    if constexpr (INV && IOM == NTT_FUSED_MULPAIR && std::is_same<A, ArithF64>::value) {
                        // the digits of the key switch are consumed as doubles by ksmac2: convert once here instead of once per output row there
                        if (a.flags & NTT_FLAG_STORE_F64) v = f64_double_to_bits(f64_from_u64(v));
                    }
                } else v = A::store_mid(x[R], md);
                nt_store_at(gout + GR, gb0, v);
            });
        }   // other rounds: the words stay in registers; the next round starts with the exchange
        (void)PRIVATE_OUT;
    });
    };   // one_digit
    if constexpr (KSMAC) {
        for (unsigned it = 0; it < a.decomp; ++it) {
            one_digit(it);
            if ((it & 7u) == 7u) static_for<0, E>([&](auto Rc) { A::mac_fix(acc0[decltype(Rc)::value], md); A::mac_fix(acc1[decltype(Rc)::value], md); });
        }
    }
    else one_digit(0u);
    if constexpr (KSMAC) {
        const unsigned lane = t & 63u, wbase = (t >> 6) * (64u * E);
        static_for<0, E / 2>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const unsigned idx = wbase + m * 128u + lane * 2u;
            nt_store2(gout + idx, A::mac_final(acc0[2 * m], md), A::mac_final(acc0[2 * m + 1], md));
            nt_store2(gout + a.out_pstride + idx, A::mac_final(acc1[2 * m], md), A::mac_final(acc1[2 * m + 1], md));
        });
    }
}

template <class A, int LOGN, int LO, int G, int TB, int EB, bool INV, bool FIRST, bool LAST, int IOM, bool HALF = false>
__global__ __launch_bounds__(1 << (TB - EB), HALF ? (2 << (TB - EB - 8)) : 1) void ntt_pass_kernel(NttArgs a) {
    __shared__ u64 lds[(G + EB - 1) / EB > 1 ? (HALF ? (ntt_lds_words(TB) + 1) / 2 : ntt_lds_words(TB)) : 1];
    ntt_pass_body<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, false, IOM, 0, HALF>(a, nullptr, lds, blockIdx.x, threadIdx.x);
}

// Tensor product of two 2-component ciphertexts between the transforms (BEHZ steps (3)-(5), evaluator.cu:56-93): one workgroup takes
// the same 2^TB-word tile of one limb of a0, a1, b0, b1, runs the last forward pass of each, forms d0 = a0 b0, d1 = a0 b1 + a1 b0,
// d2 = a1 b1 in registers -- a thread ends a forward pass and starts an inverse pass with the same E consecutive coefficients -- and
// runs the first inverse pass of d0, d1, d2.  The NTT-form operands and the NTT-form product never reach HBM.
//   TB == LOGN (N <= 8192, whole-limb tiles): the passes are the whole transforms; 4 reads + 3 writes of a limb replace the forward
//     transforms (4+4), the dyadic kernel (4+3) and the inverse transforms (3+3).
//   TB <  LOGN (N >= 32768, two-pass transforms): the operands arrive after their strided first pass and the product leaves before
//     its strided last pass; the same 4+3 replace the second forward pass, the dyadic kernel and the first inverse pass.
//   fa, fb: forward arguments over [item][2][ncomp][N], id: inverse arguments over [item][3][ncomp][N]
template <class A, int LOGN, int TB, int EB, int MINB = 1>
__global__ __launch_bounds__(1 << (TB - EB), MINB) void tensor_core_kernel(NttArgs fa, NttArgs fb, NttArgs id) {
    constexpr int G1 = LOGN - TB, E = 1 << EB;
    constexpr bool WHOLE = G1 == 0;
    __shared__ u64 lds[ntt_lds_words(TB)];
    using elem = typename A::elem;
    const unsigned t = threadIdx.x;
    const unsigned tile = blockIdx.x & ((1u << G1) - 1), lj = blockIdx.x >> G1;
    const unsigned j = lj % fa.ncomp, b = lj / fa.ncomp;
    auto bid = [&](unsigned pcount, unsigned k) { return (((b * pcount + k) * fa.ncomp + j) << G1) | tile; };
    const typename A::Mod md = A::make(fa.mods[ntt_table_index(fa, 0, j)]);
    // order chosen for register pressure: at most three held polynomials (3 * 2E registers) next to a transform in flight
    elem a0[E], b0[E], a1[E], w[E];
    if (fa.in == fb.in && fa.in_bstride == fb.in_bstride) {
        // squaring (Evaluator::square on BFV multiplies a ciphertext with itself, evaluator.cu:147-160): two forward transforms
        // instead of four; d0 = a0^2, d1 = 2 a0 a1, d2 = a1^2
        ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fa, nullptr, lds, bid(2, 0), t, a0);
        __syncthreads();
        static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; w[R] = A::inv_in(A::prod(a0[R], a0[R], md), md); });
        ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 0), t, w);
        __syncthreads();
        ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fa, nullptr, lds, bid(2, 1), t, a1);
        __syncthreads();
        static_for<0, E>([&](auto Rc) {
            constexpr int R = decltype(Rc)::value;
            const elem c = A::prod(a0[R], a1[R], md);
            a0[R] = A::inv_in(A::sum(c, c, md), md);
            a1[R] = A::inv_in(A::prod(a1[R], a1[R], md), md);
        });
        ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 1), t, a0);
        __syncthreads();
        ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 2), t, a1);
        return;
    }
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fa, nullptr, lds, bid(2, 0), t, a0);
    __syncthreads();   // the next transform's first exchange overwrites words other waves read in this one's last round
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fb, nullptr, lds, bid(2, 0), t, b0);
    __syncthreads();
    static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; w[R] = A::inv_in(A::prod(a0[R], b0[R], md), md); });
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 0), t, w);          // d0 = a0 b0
    __syncthreads();
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fa, nullptr, lds, bid(2, 1), t, a1);
    __syncthreads();
    static_for<0, E>([&](auto Rc) { constexpr int R = decltype(Rc)::value; b0[R] = A::prod(a1[R], b0[R], md); });       // a1 b0
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, false, WHOLE, true, false, 0, 1>(fb, nullptr, lds, bid(2, 1), t, w);          // b1
    __syncthreads();
    static_for<0, E>([&](auto Rc) {
        constexpr int R = decltype(Rc)::value;
        a0[R] = A::inv_in(A::sum(b0[R], A::prod(a0[R], w[R], md), md), md);   // d1 = a1 b0 + a0 b1
        a1[R] = A::inv_in(A::prod(a1[R], w[R], md), md);                       // d2 = a1 b1
    });
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 1), t, a0);
    __syncthreads();
    ntt_pass_body<A, LOGN, G1, TB, TB, EB, true, true, WHOLE, false, 0, 2>(id, nullptr, lds, bid(3, 2), t, a1);
}

// Fused key-switch inner product: grid = (L+1) rows x batch items, one whole-limb workgroup each.
//   a.in   digits in coefficient form [item][j][N] (in_bstride, in_cstride), a.ext0 the NTT-form input (diagonal)
//   keys   L pointers to u64[2][K][N];  a.out poly_prod [item][2][L+1][N] (out_bstride, out_pstride = poly, out_cstride = row)
template <class A, int LOGN, int EB>
__global__ __launch_bounds__(1 << (LOGN - EB)) void ks_mac_kernel(NttArgs a, KeyPtrs keys) {
    __shared__ u64 lds[ntt_lds_words(LOGN)];
    ntt_pass_body<A, LOGN, 0, LOGN, LOGN, EB, false, true, true, true, 0>(a, &keys, lds, blockIdx.x, threadIdx.x);
}

// Generic fallback for any 2 <= N: one workgroup per limb-polynomial, radix-2 layer by layer.
// N <= 4096 runs out of LDS; larger N (not covered by an optimised instantiation) works in
// place in global memory.  Used for small test rings (the reference's tests use N = 32).
static __global__ __launch_bounds__(256) void ntt_generic_kernel(NttArgs a, unsigned log_n, int inverse) {
    __shared__ u64 lds[4096];
    const unsigned n = 1u << log_n;
    unsigned bid = blockIdx.x;
    const unsigned j = bid % a.ncomp; bid /= a.ncomp;
    const unsigned k = bid % a.pcount;
    const unsigned b = bid / a.pcount;
    const unsigned mi = ntt_table_index(a, k, j);
    const DevModulus md = a.mods[mi];
    const u64 q = md.q, two_q = md.q << 1;
    const ulonglong2* __restrict__ tw = reinterpret_cast<const ulonglong2*>(a.tw) + (size_t)mi * n;
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
