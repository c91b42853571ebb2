// ksmac_kernels.hpp -- fused key-switch inner product for whole-limb rings, second generation.
//
// Replaces kernel_set_accumulate + ntt + kernel_accumulate_products (reference fgk/switch_key.cu:6-54, :83-154,
// driven from evaluator_keyswitching_core.cu:904-919): for every item and every output row k <= L
//     out[k][c] = sum_j  NTT_{q_key(k)}( digit_j mod q_key(k) ) (.) key_j[c][k]          c = 0, 1
// with the transformed digits living only in registers / LDS.
//
// Why a second kernel (profiles/r01_bench_v7_summary.txt): the first one (ks_mac_kernel in
// ntt_kernels.hpp) runs ONE 1024-thread workgroup per CU at N = 16384.  Its 2 x 16 accumulators + 16 coefficients
// per thread do not fit the 128-VGPR cap of that shape (28 registers spilled, ~0.85 GB of scratch traffic per
// 512-item launch), its phases (load, butterflies, LDS exchange, key loads) serialise behind workgroup barriers,
// the keys are converted u64 -> f64 again for every item and the last butterfly rounds fetch their twiddles with one
// 128-byte line per lane.
//
// Design here (gfx950, wave64, FP64-exact arithmetic of dev_math_f64.hpp, moduli < 2^50):
//   * A workgroup owns a TILE of 2^13 outputs of one (item, row): the whole limb at N = 8192, one HALF of the
//     outputs at N = 16384 (one QUARTER at N = 32768, two layers applied while loading) -- after the first Cooley-Tukey
//     layer the two halves of a negacyclic NTT are independent
//     transforms, so a half-tile workgroup applies layer 0 while loading (u +- w*v, the twiddle multiply is the
//     only duplicated work, ~8 % more FP64 ops) and then runs a 13-layer transform in 68 KiB of LDS.  Two
//     independent workgroups share a CU: one computes while the other waits on HBM / LDS / a barrier.
//   * 256 threads x 32 coefficients: 3 register rounds (3|4 + 5 + 5 layers) = two LDS exchanges per digit; the
//     accumulators (2 x 32 doubles) and the coefficients use 192 of the 256 VGPRs available at 2 waves per SIMD.
//   * Round-2 twiddles come from a lane-interleaved copy of the table (every load instruction reads 1 KiB of
//     consecutive bytes), round-1 twiddles from a 256-byte vector per wave-half, round-0 twiddles are scalar loads.
//   * Evaluation keys are prepared once per call (ksmac_prepare_keys_kernel): converted to exact doubles and
//     permuted to the accumulators' register layout, so the multiply-accumulate reads its digit operand straight
//     from the registers the last butterfly layer left it in (no transpose through LDS per digit) with 16-byte
//     coalesced key loads, and spends 7 FP64 operations per term instead of 9.
//   * LDS index padding of 2 words per 32 keeps every access of the last round a conflict-free 16-byte access.
// Results are canonical residues, bit-identical to the reference (exactness argument: dev_math_f64.hpp; the growth
// bound of a 5-layer block that starts from |x| <= p/2 + 1 is 7.7 p < 2^53 for p < 2^50).
#pragma once
#include "ntt_kernels.hpp"

namespace troyn {

struct KsMacArgs {
    const u64* digits; long long dig_bstride, dig_cstride;      // coefficient-form digits [item][j][N]: canonical u64, or
                                                                 // (DIGF64) the same integers stored as doubles
    const u64* diag;   long long diag_bstride, diag_cstride;     // NTT-form input limbs [item][j][N]; nullptr: no diagonal shortcut
    const u64* diag_b;                                           // non-null: the NTT-form input is the product diag (.) diag_b (same strides),
                                                                 // formed while loading (fused multiply -> relinearize chain)
    u64* out;          long long out_bstride, out_pstride, out_cstride;   // [item][2][L+1][N]: (item, component, row)
    // fused multiply -> relinearize -> rescale chain: the two input ciphertexts a, b [item][2][limbs][N] (NTT form).  Non-null: the
    // keys of the data rows were prepared times qk^-1 (ksmac_prepare_keys_kernel), and a data row k < L leaves the kernel as
    //   Q_0 = P_0 qk^-1 + a0 (.) b0,   Q_1 = P_1 qk^-1 + a0 (.) b1 + a1 (.) b0      (limb k of a and b)
    // i.e. with relinearize's division by the special prime and its trailing add (evaluator_keyswitching_core.cu:641-656,
    // evaluator_keyswitching.cu:143) already applied to the inner product P; the special row stays P.  The memory-bound kernels
    // behind this one then read one row where they read P and four rows of a and b.
    const u64* ten_a; const u64* ten_b; long long ten_bstride, ten_pstride;
    const double* diag_keys;   // with ten_a (TEN) or diag (DG): [k][2][N] the two components of key k under modulus k (times qk^-1), NATURAL order: the diagonal
                               // digit a1 (.) b1 of a data row is multiplied-accumulated in the epilogue's coalesced layout
    const DevModulus* mods;
    const double* tw;       // [K][N]  forward twiddles w, reference table order
    const double* tw_r1;    // [K][N/1024][32]  round-1 vector per value of the index bits above bit 9
    const double* tw_r2;    // [K][N]  round-2 vectors, lane-interleaved (ksmac_r2_slot)
    const double* keys;     // prepared keys [L][2][KEYROWS][N] (ksmac_prepare_keys_kernel)
    long long key_jstride, key_pstride;   // elements between keys j / between the two components
    unsigned L;             // digits = data limbs; rows = L + 1
    unsigned table_start, table_count;    // row k uses modulus table_start + (k == L ? table_count - 1 : k)
    unsigned batch;
    unsigned grouped;       // 1: the (L+1) * HALVES workgroups of an item are dealt to one XCD (batch % 8 == 0)
    unsigned long long row_mask;   // 0: all L + 1 output rows; else the launch covers the rows whose bit is set (mixed chains: rows of moduli < 2^50)
    unsigned long long* prof;   // development only (tools/ksbench -DKSM_PHASE_PROFILE): per-phase shader cycles of wave 0 of every workgroup, summed
    // digit-parallel form for small batches (SPLITJ instantiations): the grid is L times larger, workgroup (tile, js) transforms and
    // multiplies ONE digit js and leaves its two accumulators as re-centred doubles in part[js] (strides of `out` inside a slot);
    // ksmac_split_reduce_kernel adds the L slots (and the diagonal digit / tensor terms of the NTT-form / fused callers) and stores
    // canonical words.  split_skip_diag: slot js == k of a data row k is not produced (the reducer forms that term itself).
    double* part; long long part_jstride;
    unsigned split_skip_diag;
    // The digit-parallel form reads the caller's keys AS THEY ARE (u64 [2][K][N] per digit, KSwitchKeys layout): with a handful of items a key word is
    // used once or twice, so converting all of them first (ksmac_prepare_keys_kernel: 7 us + a launch for 16 MB of traffic at cfg3) costs more than
    // converting the words a workgroup actually touches.  raw_pstride = K N.  The digit is crossed to the coalesced layout before the
    // multiply-accumulate (one exchange through the wave's slice, the one the epilogue no longer needs).  split_scale (fused chain): Shoup pairs of
    // qk^-1 mod q_r for the data rows -- applied by the reducer to the slot sum instead of to every key word.
    KeyPtrs raw;
    long long raw_pstride;
    const ulonglong2* split_scale;
    unsigned no_load_corr;   // half tiles, fused chain: the chain passed ksmac_no_load_corr_ok -> the NLC instantiation
};

constexpr int KSM_TB = 13;                  // tile bits
constexpr int KSM_THREADS = 1 << (KSM_TB - 5);
__host__ __device__ constexpr unsigned ksm_phys(unsigned w) { return w + 2u * (w >> 5); }     // LDS word -> padded word
constexpr unsigned KSM_LDS_WORDS = (1u << KSM_TB) + 2u * ((1u << KSM_TB) >> 5);

// position of natural index i (within one limb) in the prepared-key / round-2 layouts: blocks of 2048 words =
// 64 lanes x 32 registers are stored as [m = reg/2][lane][reg%2], so that lane-consecutive 16-byte loads are contiguous
__host__ __device__ constexpr unsigned ksm_perm(unsigned i) {
    return (i & ~2047u) | ((((i >> 1) & 15u) * 64u + ((i >> 5) & 63u)) * 2u) | (i & 1u);
}

// keys[j] -> [2][K][N] u64 (the reference's KSwitchKeys layout)  ==>  prepared [j][2][K][N] doubles, permuted.
// scale != nullptr (fused chain): the rows of the data moduli r < scale_rows (row index within a component) are multiplied by
// scale[r] = qk^-1 mod q_r (Shoup pair) first -- exact integer arithmetic, so the inner product comes out as P qk^-1 mod q_r.
static __global__ __launch_bounds__(256) void ksmac_prepare_keys_kernel(KeyPtrs keys, unsigned L, unsigned rows_per_key, unsigned n, double* out,
                                                                        const ulonglong2* scale, const DevModulus* mods, unsigned scale_rows, double* diag_out) {
    const size_t pairs_per_key = (size_t)rows_per_key * (n / 2);
    const size_t total = (size_t)L * pairs_per_key;
    const unsigned K = rows_per_key / 2;
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const unsigned j = (unsigned)(p / pairs_per_key);
        const size_t q = p % pairs_per_key;
        const size_t row = q / (n / 2);
        const unsigned i = (unsigned)(q % (n / 2)) * 2u;
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(keys.p[j] + row * n + i);
        const unsigned r = (unsigned)(row % K);
        if (scale && r < scale_rows) {
            const ulonglong2 f = scale[r];
            const u64 qr = mods[r].q;
            v.x = shoup_mul(v.x, f.x, f.y, qr); v.y = shoup_mul(v.y, f.x, f.y, qr);
        }
        double2 d = make_double2(f64_from_u64(v.x), f64_from_u64(v.y));
        *reinterpret_cast<double2*>(out + ((size_t)j * rows_per_key + row) * n + ksm_perm(i)) = d;
        // the block (key j, modulus j) again in natural order for the epilogue of the fused chain's data rows (KsMacArgs::diag_keys)
        if (diag_out && r == j) *reinterpret_cast<double2*>(diag_out + ((size_t)j * 2 + row / K) * n + i) = d;
    }
}

// A pointer every lane agrees on, pinned to scalar registers: loads through (uniform base + 32-bit lane offset + immediate)
// then use the scalar-base addressing form instead of a 64-bit address per lane (and per loop iteration).
template <class P>
__device__ __forceinline__ P* ksm_uniform(P* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (P*)(((unsigned long long)hi << 32) | lo);
}
// load a V from (uniform global base) + (32-bit lane byte offset): global address space, scalar-base addressing form
template <class V> struct ksm_native;
template <> struct ksm_native<double2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct ksm_native<ulonglong2> { typedef u64 type __attribute__((ext_vector_type(2))); };
template <class V>
__device__ __forceinline__ V ksm_gload(const void* ubase, unsigned byte_off) {
    typedef typename ksm_native<V>::type NV;
    typedef const char __attribute__((address_space(1)))* gcp;
    typedef const NV __attribute__((address_space(1)))* gvp;
    const NV v = *(gvp)((gcp)(unsigned long long)ubase + byte_off);
    V r; r.x = v.x; r.y = v.y;
    return r;
}

// butterflies of register bit RB for the twiddle groups [G0, G0 + NG): tw[i] belongs to group G0 + i
template <int RB, int G0, int NG, bool NOBF = false>
__device__ __forceinline__ void ksm_layer(double (&x)[32], const double* tw, double inv_p, double p) {
    static_for<0, NG>([&](auto gc) {
        constexpr int g = G0 + decltype(gc)::value;
        const double w = tw[decltype(gc)::value];
        static_for<0, (1 << RB)>([&](auto oc) {
            constexpr int R0 = (g << (RB + 1)) | decltype(oc)::value, R1 = R0 | (1 << RB);
            const double r = NOBF ? w : f64_mulq(x[R1], w, inv_p, p);
            const double u = x[R0];
            x[R0] = u + r; x[R1] = u - r;
        });
    });
}

// A 5-layer register round whose 31 twiddles sit in a 32-slot vector (slot (1 << lvl) + g; slot 0 unused), fetched 8
// slots at a time one step ahead of the butterflies that use them.  ld(q) returns slots 2q, 2q+1.
// first/second: slots 0..7 and 8..15, already requested by the caller (before the LDS exchange that feeds the round).
template <bool NOBF = false, class LD, class HK>
__device__ __forceinline__ void ksm_round5(double (&x)[32], double (&ta)[8], double (&tb)[8], LD&& ld, HK&& after_last_load, double inv_p, double p) {
    ksm_layer<4, 0, 1, NOBF>(x, ta + 1, inv_p, p);
    ksm_layer<3, 0, 2, NOBF>(x, ta + 2, inv_p, p);
    ksm_layer<2, 0, 4, NOBF>(x, ta + 4, inv_p, p);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 4>([&](auto qc) { const double2 v = ld(8 + decltype(qc)::value); ta[2 * decltype(qc)::value] = v.x; ta[2 * decltype(qc)::value + 1] = v.y; });
    ksm_layer<1, 0, 8, NOBF>(x, tb, inv_p, p);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 4>([&](auto qc) { const double2 v = ld(12 + decltype(qc)::value); tb[2 * decltype(qc)::value] = v.x; tb[2 * decltype(qc)::value + 1] = v.y; });
    after_last_load();      // every load of the round is on its way: whatever is issued here queues behind them
    ksm_layer<0, 0, 8, NOBF>(x, ta, inv_p, p);
    __builtin_amdgcn_sched_barrier(0);
    ksm_layer<0, 8, 8, NOBF>(x, tb, inv_p, p);
    __builtin_amdgcn_sched_barrier(0);
}

#ifdef KSM_PHASE_PROFILE
#define KSM_MARK(ph) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[ph] += now_ - prof_t; prof_t = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define KSM_MARK(ph) do { } while (0)
#endif

// ABL: development-only ablation mask (tools/ksbench), 0 in the library.  bit0 digit loads / bit1 key loads / bit5 twiddle loads / bit7 tensor operand loads all hit one cache line, bit8 no tensor products,
// bit2 no LDS exchange, bit3 no butterflies, bit4 no multiply-accumulate (results are wrong by design)
// TEN: the instantiation of the fused chain (KsMacArgs::ten_a): data rows leave as Q = P qk^-1 + c; kept out of the other instantiations,
// whose digit loop otherwise pays for the epilogue's registers (scratch 32 -> 48 bytes, relinearize -4 %)
// DG: the same epilogue without the tensor terms, for the separate key switch on an NTT-form target (KsMacArgs::diag + diag_keys): the
// diagonal digit is multiplied-accumulated in the coalesced layout, the loop drops its diagonal path (scratch 36 -> 12 bytes)
// NODIAG: coefficient-form target (BFV): there is no diagonal digit, and the instantiation does not carry the loop's diagonal path
// SPLITJ: one digit per workgroup (KsMacArgs::part), for launches that would otherwise leave most of the chip idle while 12 workgroups
// walk their L digits one after the other (a single ciphertext: 85 us of a 150 us multiply + relinearize + rescale)
// NLC (half tiles): no re-centring between layer 0 (applied while loading) and round 0 -- exact when every digit modulus is within ~1.2x of the
// row's modulus (the host checks the growth bound of the four layers, ksmac_no_load_corr_ok): -3 of ~107 FP64 operations per coefficient and digit
template <int LOGN, bool DIGF64, int ABL = 0, bool WIDE = false, bool TEN = false, bool DG = false, bool NODIAG = false, bool SPLITJ = false, bool NLC = false>
#ifndef KSM_WAVES_PER_SIMD
#define KSM_WAVES_PER_SIMD 2
#endif
#ifndef KSM_KEY_AHEAD
#define KSM_KEY_AHEAD 3
#endif
#ifndef KSM_LOAD_WINDOW
#define KSM_LOAD_WINDOW 12
#endif
#ifndef KSM_TEN_WINDOW
#define KSM_TEN_WINDOW 2      // register pairs (six 16-byte loads each) in flight in the epilogue of the fused chain's data rows
#endif
__global__ __launch_bounds__(KSM_THREADS, KSM_WAVES_PER_SIMD) void ksmac2_kernel(KsMacArgs a) {
    constexpr auto abl = [](int bit) constexpr { return ((ABL >> bit) & 1) != 0; };
    static_assert(LOGN >= 13 && LOGN <= 15, "ksmac2 covers N = 8192, 16384 and 32768");
    constexpr bool SPLIT = LOGN == 14;        // half tiles: one Cooley-Tukey layer applied while loading
    constexpr bool SPLIT4 = LOGN == 15;       // quarter tiles: two layers applied while loading
    constexpr unsigned N = 1u << LOGN;
    constexpr int HALVES = 1 << (LOGN - KSM_TB);
    __shared__ __attribute__((aligned(16))) u64 lds[KSM_LDS_WORDS];

    const unsigned t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if constexpr (abl(6)) {
        // two workgroups share a CU: the one in the upper LDS slot yields VALU issue to the other, so that the pair does not
        // run its compute and memory phases in lock-step
        const unsigned lds_base = __builtin_amdgcn_s_getreg((7 << 11) | 6);     // HW_REG_LDS_ALLOC.LDS_BASE
        if (lds_base) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
    }
    // ---- workgroup -> (item, row, half) ------------------------------------------------------------------
    static_assert(!SPLITJ || (NODIAG && !TEN && !DG), "the digit-parallel form has no diagonal path of its own");
    const unsigned split_js = SPLITJ ? blockIdx.x % a.L : 0u;                  // the one digit of this workgroup
    const unsigned bx = SPLITJ ? blockIdx.x / a.L : blockIdx.x;
    unsigned b, k, h;
    {
        const unsigned nrows = a.row_mask ? (unsigned)__builtin_popcountll(a.row_mask) : a.L + 1;
        const unsigned G = nrows * HALVES;
        unsigned g;
        if (a.grouped == 3) {
            // bands of two rows x 64 / (2 HALVES) items per XCD: the 64 workgroup slots of an XCD hold one band, whose keys (2 rows x 2L
            // limbs, 2.6 MB at cfg3) stay in that XCD's L2 while the items stream through; an item's digits are fetched once per band
            // (order 1 fetches each digit once but every key line per workgroup: 7.8 MB of keys cycle through 4 MB of L2).
            // Odd row count: the workgroups of the missing row exit.  Measured -4..-6 % on the launch (profiles/r03_ksmac_ab.txt).
            constexpr unsigned ITEMS = 64u / (2u * HALVES);
            const unsigned bands = (nrows + 1u) / 2u, per = bands * 64u;
            const unsigned xcd = bx & 7u, sq = bx >> 3, r = sq % per, r2 = r & 63u;
            k = 2u * (r >> 6) + (r2 % (2u * HALVES)) / HALVES;
            h = r2 % HALVES;
            b = ((sq / per) * 8u + xcd) * ITEMS + r2 / (2u * HALVES);
            if (k >= nrows) return;
        } else if (a.grouped == 2) {
            // row-major: the whole chip works on one output row at a time, so that row's 2L key limbs (1.3 MB at cfg3) stay
            // in every XCD's L2; the halves of an item sit 8 workgroups apart = on the same XCD, back to back
            const unsigned per = 8u * HALVES, r = bx % per, q = bx / per;
            h = r / 8u; b = (q % (a.batch / 8u)) * 8u + (r % 8u); k = q / (a.batch / 8u);
        } else {
        if (a.grouped) {
            const unsigned per = 8u * G, r = bx % per;
            g = r / 8u; b = (bx / per) * 8u + (r % 8u);
        } else {
            g = bx % G; b = bx / G;
        }
        k = g / HALVES; h = g % HALVES;
        }
        if (a.row_mask) k = nth_set_bit(a.row_mask, k);
    }
    const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;    // row of the key / modulus slot
    const unsigned mi = a.table_start + mrow;
    const DevModulus dm = a.mods[mi];
    const F64Mod fm{dm.pd, dm.inv_pd};
    const double p = fm.p, inv_p = fm.inv_p;

    typedef const double __attribute__((address_space(4)))* cdp;
    const cdp tws = (cdp)(unsigned long long)(a.tw + (size_t)mi * N);       // scalar (wave-uniform) twiddle fetches

    // every global address below is (workgroup-uniform base) + (32-bit per-thread byte offset) + immediate, so the loads use
    // the scalar-base addressing form and no 64-bit address lives in VGPRs
    auto at = [](const void* ubase, unsigned byte_off) { return reinterpret_cast<const char*>(ubase) + byte_off; };
    const double* r1u = a.tw_r1 + ((size_t)mi * (N >> 10) + h * (KSM_THREADS >> 5)) * 32;   // index bits above bit 9 = T >> 5
    const double* r2u = a.tw_r2 + (size_t)mi * N + (size_t)h * (KSM_THREADS * 32);
    unsigned r1off = (t >> 5) * 256u, slice_off = wave * 16384u + lane * 16u;                // bytes

    double acc0[32], acc1[32];
    static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = 0.0; acc1[decltype(rc)::value] = 0.0; });

    // LDS addresses (padded words).  Round 0 holds registers r = b0 | b9<<1 | R3<<2 of tile index
    // b0 | t<<1 | b9<<9 | R3<<10; round 1 holds bits [5,10); round 2 holds bits [0,5).
    unsigned p0 = ksm_phys(t << 1);                                             // + ksm_phys(b9<<9 | R3<<10): multiples of 32 words
    unsigned p1 = ksm_phys((t & 31u) | ((t >> 5) << 10));                       // + 34 * R
    unsigned p2 = ksm_phys(t << 5);                                             // + R  (R < 32: no pad crossed)
    unsigned pt = ksm_phys(wave * 2048u + lane * 2u);                            // transposed pairs: + ksm_phys(128 m)

    const double* kbase = a.keys + (size_t)mrow * N + (size_t)h * (KSM_THREADS * 32);
    const u64* dig_item = a.digits + (long long)b * a.dig_bstride;

    // one <digit, key> term per register:  acc += v * key  (mod p, lazy: |term| <= 0.69 p)
    auto mac2 = [&](double& a0, double& a1, double v, double y0, double y1) {
        const double h0 = v * y0, h1 = v * y1;
        const double l0 = __builtin_fma(v, y0, -h0), l1 = __builtin_fma(v, y1, -h1);
        const double q0 = __builtin_rint(h0 * inv_p), q1 = __builtin_rint(h1 * inv_p);
        a0 += __builtin_fma(-q0, p, h0) + l0;
        a1 += __builtin_fma(-q1, p, h1) + l1;
    };
    // multiply-accumulate the 32 registers v[] with key j, two registers per 16-byte key load; the key loads run AHEAD
    // pairs ahead of the arithmetic (8 VGPRs per pair in flight).  v is re-centred here (|v| <= p/2 + 1).
    auto mac_all = [&](double (&v)[32], unsigned j) {
        const double* k0 = ksm_uniform(kbase + (long long)j * a.key_jstride);
        const double* k1 = ksm_uniform(k0 + a.key_pstride);
        constexpr int AHEAD = KSM_KEY_AHEAD;
        const unsigned koff = abl(1) ? (slice_off & 16u) : slice_off;
        double2 y0[16], y1[16];
        static_for<0, AHEAD>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            y0[m] = ksm_gload<double2>(k0 + (abl(1) ? 0 : m * 128), koff);
            y1[m] = ksm_gload<double2>(k1 + (abl(1) ? 0 : m * 128), koff);
        });
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
#ifndef KSM_MAC_NO_FENCE      // experiment (tools/ksmac_variants.sh): let the compiler place the key loads of the window itself
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (m + AHEAD < 16) {
                y0[m + AHEAD] = ksm_gload<double2>(k0 + (abl(1) ? 0 : (m + AHEAD) * 128), koff);
                y1[m + AHEAD] = ksm_gload<double2>(k1 + (abl(1) ? 0 : (m + AHEAD) * 128), koff);
            }
            if constexpr (abl(4)) { acc0[2 * m] += v[2 * m] + y0[m].x + y1[m].x; acc1[2 * m + 1] += v[2 * m + 1] + y0[m].y + y1[m].y; return; }
            const double v0 = f64_corr(v[2 * m], fm), v1 = f64_corr(v[2 * m + 1], fm);
            mac2(acc0[2 * m], acc1[2 * m], v0, y0[m].x, y1[m].x);
            mac2(acc0[2 * m + 1], acc1[2 * m + 1], v1, y0[m].y, y1[m].y);
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    // a digit word as it enters the transform
    // (WIDE: a chain with limbs of 2^50 and more -- the digit of such a limb is reduced with integer arithmetic first, Modulus::reduce)
    auto dig_in = [&](u64 raw) -> double {
        if constexpr (DIGF64) return f64_bits_to_double(raw);
        else if constexpr (WIDE) {
            // raw < 2^61 = hi 2^30 + lo.  hi 2^30 is exact in a double (31 significant bits), so (hi 2^30 mod p) comes out of one quotient estimate and
            // one exact fma; the result is raw mod p up to sign, |.| <= p/2 + 2^30 -- every consumer below re-centres.  Round 3 reduced the word with
            // Modulus::reduce (64-bit integer multiplies): 376 / 396 bytes of scratch per lane at N = 16384 / 32768, the temporaries of 64 reductions
            // in flight next to the load window.
            const double h = (double)(unsigned)(raw >> 30) * 1073741824.0;
            const double lo = (double)((unsigned)raw & 0x3fffffffu);
            return __builtin_fma(-__builtin_rint(h * inv_p), p, h) + lo;
        }
        else return f64_from_u64(raw);
    };

#ifdef KSM_PHASE_PROFILE
    unsigned long long prof_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long prof_t = __builtin_readcyclecounter();
#endif
    // fused chain, data row: the diagonal digit (a1 (.) b1 of limb k, NTT form) is left to the epilogue, which loads a1 and b1 anyway
    // (the sum is exact, its order is free)
    static_assert(!(TEN && DG), "one epilogue");
    const bool ten_row = (TEN || DG) && k < a.L;      // data row of an instantiation whose epilogue takes the diagonal digit
    if constexpr (SPLITJ) { if (a.split_skip_diag && k < a.L && split_js == k) return; }      // the reducer forms the diagonal digit's term
    const unsigned steps = SPLITJ ? 1u : (ten_row ? a.L - 1 : a.L);
    for (unsigned step = 0; step < steps; ++step) {
        const unsigned it = SPLITJ ? split_js : (!ten_row ? step : (step < k ? step : step + 1));     // digit of this step
        double x[32];
        // nothing below depends on the digit except the input and the key: without these the compiler hoists every
        // twiddle load (and its w/p product) out of the digit loop and spills them
        asm volatile("" : "+v"(r1off), "+v"(slice_off));
        // the same for the LDS bases: the ds_read2 / ds_write offsets reach 2 KB, so the compiler derives ~6 more base registers from
        // p1; hoisted out of the digit loop they are spilled, and every reload after the exchange barrier comes with s_waitcnt vmcnt(0),
        // i.e. it waits for the round's twiddle loads that were issued to travel under the exchange (round 3: -5.6 % on the launch).
        // Re-derived per digit they cost one v_add each.
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(pt));
        if (!TEN && !DG && !NODIAG && a.diag && it == k) {      // (TEN: the diagonal digit of a data row is handled in the epilogue, the special row has none)
            // the digit of row k under its own modulus is the NTT-form input limb (evaluator_keyswitching_core.cu:851-852):
            // coalesced load, transpose through the wave's own LDS slice into the accumulators' layout
            const u64* dg = ksm_uniform(a.diag + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + (size_t)h * (KSM_THREADS * 32));
            const u64* dgb = a.diag_b ? ksm_uniform(a.diag_b + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + (size_t)h * (KSM_THREADS * 32)) : nullptr;
            // two batches of 8 (+ 8) loads that are all in flight before the first is used; the product of the fused chain is formed in the
            // coalesced layout, so there is ONE transposition (round 3: the second operand used to be loaded, transposed and multiplied
            // one 16-byte load at a time, each behind its own s_waitcnt vmcnt(0))
            auto diag_half = [&](auto hc, auto with_b) {
                constexpr int hb = decltype(hc)::value;
                constexpr bool WB = decltype(with_b)::value;
                ulonglong2 va[8], vb[WB ? 8 : 1];
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 8>([&](auto ic) { va[decltype(ic)::value] = ksm_gload<ulonglong2>(dg + (hb * 8 + decltype(ic)::value) * 128, slice_off); });
                if constexpr (WB) static_for<0, 8>([&](auto ic) { vb[decltype(ic)::value] = ksm_gload<ulonglong2>(dgb + (hb * 8 + decltype(ic)::value) * 128, slice_off); });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 8>([&](auto ic) {
                    constexpr int j = decltype(ic)::value, m = hb * 8 + j;
                    double2 pr = make_double2(f64_from_u64(va[j].x), f64_from_u64(va[j].y));
                    if constexpr (WB) {
                        pr.x = f64_mulq(f64_corr(pr.x, fm), f64_from_u64(vb[j].x), inv_p, p);
                        pr.y = f64_mulq(f64_corr(pr.y, fm), f64_from_u64(vb[j].y), inv_p, p);
                    }
                    *reinterpret_cast<double2*>(&lds[pt + ksm_phys(m * 128u)]) = pr;
                });
            };
            if (dgb) { diag_half(std::integral_constant<int, 0>{}, std::true_type{}); diag_half(std::integral_constant<int, 1>{}, std::true_type{}); }
            else static_for<0, 16>([&](auto mc) {       // plain form: the limb streams through the slice as it arrives
                constexpr int m = decltype(mc)::value;
                const ulonglong2 v = ksm_gload<ulonglong2>(dg + m * 128, slice_off);
                *reinterpret_cast<double2*>(&lds[pt + ksm_phys(m * 128u)]) = make_double2(f64_from_u64(v.x), f64_from_u64(v.y));
            });
            __builtin_amdgcn_wave_barrier();
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const double2 v = *reinterpret_cast<const double2*>(&lds[p2 + 2 * m]);
                x[2 * m] = v.x; x[2 * m + 1] = v.y;
            });
            KSM_MARK(7);
        } else {
        // ---- load (+ layer 0 for a half tile), two steps of 8 register pairs ---------------------------------
        const u64* gin_u = ksm_uniform(dig_item + (long long)it * a.dig_cstride);
        const unsigned gin_off = abl(0) ? ((t << 4) & 16u) : (t << 4);
        if constexpr (SPLIT) {
            const double w1 = tws[1];
            const double sgn = h ? -1.0 : 1.0;       // upper half of the outputs: u - w*v
            // a rolling window of loads: KSM_LOAD_WINDOW register pairs (two 16-byte loads each) are requested up front, and every pair
            // that is consumed makes room for the request of another one -- ONE exposed memory latency per digit instead of one per
            // batch (the first version requested 8 pairs, consumed them, requested the other 8)
            constexpr int W0 = KSM_LOAD_WINDOW;
            ulonglong2 ru[16], rv[16];
            auto request = [&](auto ic) {
                constexpr int i = decltype(ic)::value;     // i = b9 | R3<<1
                ru[i] = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)) * (abl(0) ? 0 : 1), gin_off);
                rv[i] = ksm_gload<ulonglong2>(gin_u + 8192 + (((i & 1) << 9) + ((i >> 1) << 10)) * (abl(0) ? 0 : 1), gin_off);
            };
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, W0>([&](auto ic) { request(ic); });
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                __builtin_amdgcn_sched_barrier(0);
                const double u0 = dig_in(ru[i].x), u1 = dig_in(ru[i].y), v0 = dig_in(rv[i].x), v1 = dig_in(rv[i].y);
                // 0 <= u, v < 2^50: one re-centring after the layer instead of one per input
                if constexpr (NLC) {       // |x| <= max q_j + 0.875 p; three more layers stay below 2^53 (checked on the host for this chain)
                    x[2 * i] = __builtin_fma(sgn, f64_mulq(v0, w1, inv_p, p), u0);
                    x[2 * i + 1] = __builtin_fma(sgn, f64_mulq(v1, w1, inv_p, p), u1);
                } else {
                    x[2 * i] = f64_corr(__builtin_fma(sgn, f64_mulq(v0, w1, inv_p, p), u0), fm);
                    x[2 * i + 1] = f64_corr(__builtin_fma(sgn, f64_mulq(v1, w1, inv_p, p), u1), fm);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i + W0 < 16) request(std::integral_constant<int, i + W0>{});
            });
        } else if constexpr (SPLIT4) {
            // quarter tile h = 2 hA + hB of a 2^15-point transform: with (a, b, c, d) = x[i], x[i+N/4], x[i+N/2], x[i+3N/4],
            //   layer 0:  u = a + sA w1 c,  v = b + sA w1 d          (sA = -1 for the upper half hA = 1)
            //   layer 1:  x = u + sB wB v,  wB = tw[2 + hA]           (sB = -1 for the odd quarter hB = 1)
            // the three twiddle products are the work the four quarter-tile workgroups duplicate
            const double w1 = tws[1], wB = tws[2 + (h >> 1)];
            const double sA = (h >> 1) ? -1.0 : 1.0, sB = (h & 1) ? -1.0 : 1.0;
            // rolling window of loads as above: W4 register pairs (four 16-byte loads each) in flight
            constexpr int W4 = KSM_LOAD_WINDOW / 3;
            ulonglong2 ra[16], rb_[16], rc[16], rd[16];
            auto request = [&](auto ic) {
                constexpr int i = decltype(ic)::value;     // i = b9 | R3<<1
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
                // raw inputs below 2^50: |u|, |v| <= 3.3 p, |x| <= 5 p before the re-centring
                const double u0 = __builtin_fma(sA, f64_mulq(dig_in(rc[i].x), w1, inv_p, p), dig_in(ra[i].x));
                const double u1 = __builtin_fma(sA, f64_mulq(dig_in(rc[i].y), w1, inv_p, p), dig_in(ra[i].y));
                const double v0 = __builtin_fma(sA, f64_mulq(dig_in(rd[i].x), w1, inv_p, p), dig_in(rb_[i].x));
                const double v1 = __builtin_fma(sA, f64_mulq(dig_in(rd[i].y), w1, inv_p, p), dig_in(rb_[i].y));
                x[2 * i] = f64_corr(__builtin_fma(sB, f64_mulq(v0, wB, inv_p, p), u0), fm);
                x[2 * i + 1] = f64_corr(__builtin_fma(sB, f64_mulq(v1, wB, inv_p, p), u1), fm);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (i + W4 < 16) request(std::integral_constant<int, i + W4>{});
            });
        } else {
            static_for<0, 16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const ulonglong2 v = ksm_gload<ulonglong2>(gin_u + (((i & 1) << 9) + ((i >> 1) << 10)) * (abl(0) ? 0 : 1), gin_off);
                x[2 * i] = f64_corr(dig_in(v.x), fm);
                x[2 * i + 1] = f64_corr(dig_in(v.y), fm);
            });
        }
        __builtin_amdgcn_sched_barrier(0);
        KSM_MARK(0);
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
                    const double r = abl(3) ? w : f64_mulq(x[R1], w, inv_p, p);
                    const double u = x[R0];
                    x[R0] = u + r; x[R1] = u - r;
                });
            });
        });
        __builtin_amdgcn_sched_barrier(0);
        KSM_MARK(1);
        // ---- exchange 0 -> 1 -------------------------------------------------------------------------------
        double ta[8], tb[8];
        if constexpr (!abl(2)) __syncthreads();     // every wave has finished reading its slice of the previous digit
        if constexpr (!abl(2)) static_for<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr unsigned off = ksm_phys(((i & 1) << 9) | ((i >> 1) << 10));
            double2 v = make_double2(f64_corr(x[2 * i], fm), f64_corr(x[2 * i + 1], fm));
            *reinterpret_cast<double2*>(&lds[p0 + off]) = v;
        });
        // the first 16 twiddle slots of round 1 travel while the exchange completes (x is dead here)
        static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + (abl(5) ? 0 : 2 * q), r1off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
        static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r1u + (abl(5) ? 0 : 8 + 2 * q), r1off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
        if constexpr (!abl(2)) __syncthreads();
        if constexpr (!abl(2)) static_for<0, 32>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            x[R] = f64_bits_to_double(lds[p1 + 34 * R]);
        });
        __builtin_amdgcn_sched_barrier(0);
        KSM_MARK(2);
        // ---- round 1: tile bits 9..5 = register bits 4..0 ----------------------------------------------------
        ksm_round5<abl(3)>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r1u + (abl(5) ? 0 : 2 * q), r1off); }, [] {}, inv_p, p);
        KSM_MARK(3);
        // ---- exchange 1 -> 2 -------------------------------------------------------------------------------
        if constexpr (!abl(2)) static_for<0, 32>([&](auto rc) {
            constexpr int R = decltype(rc)::value;
            lds[p1 + 34 * R] = f64_double_to_bits(f64_corr(x[R], fm));
        });
        static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + (abl(5) ? 0 : 128 * q), abl(5) ? (slice_off & 16u) : slice_off); ta[2 * q] = v.x; ta[2 * q + 1] = v.y; });
        static_for<0, 4>([&](auto qc) { constexpr int q = decltype(qc)::value; const double2 v = ksm_gload<double2>(r2u + (abl(5) ? 0 : 128 * (4 + q)), abl(5) ? (slice_off & 16u) : slice_off); tb[2 * q] = v.x; tb[2 * q + 1] = v.y; });
        // this exchange stays inside groups of 32 consecutive threads (tile bits [10,13) = t >> 5 on both sides): a wave reads only
        // what it wrote itself, LDS executes a wave's accesses in order -- no workgroup barrier
        if constexpr (!abl(2)) __builtin_amdgcn_wave_barrier();
        if constexpr (!abl(2)) static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const double2 v = *reinterpret_cast<const double2*>(&lds[p2 + 2 * m]);
            x[2 * m] = v.x; x[2 * m + 1] = v.y;
        });
        __builtin_amdgcn_sched_barrier(0);
        KSM_MARK(4);
        // ---- round 2: tile bits 4..0 = register bits 4..0, lane-interleaved twiddle vectors --------------------
        ksm_round5<abl(3)>(x, ta, tb, [&](int q) { return ksm_gload<double2>(r2u + (abl(5) ? 0 : 128 * q), abl(5) ? (slice_off & 16u) : slice_off); }, [] {}, inv_p, p);
        KSM_MARK(5);
        }
        // ---- multiply-accumulate with key `it` straight from the registers ---------------------------------------
        if constexpr (SPLITJ) {
            // digit-parallel form: cross the digit to the coalesced layout (register pair m <-> words m*128 + lane*2 of the wave's 2048), then
            // the caller's own key words, 16 bytes per lane, converted on the fly; the accumulators stay in that layout for the store
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                *reinterpret_cast<double2*>(&lds[p2 + 2 * m]) = make_double2(f64_corr(x[2 * m], fm), f64_corr(x[2 * m + 1], fm));
            });
            const u64* k0 = ksm_uniform(a.raw.p[it] + (size_t)mrow * N + (size_t)h * (KSM_THREADS * 32));
            const u64* k1 = ksm_uniform(k0 + a.raw_pstride);
            constexpr int AHEAD = 4;
            ulonglong2 y0[16], y1[16];
            static_for<0, AHEAD>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                y0[m] = ksm_gload<ulonglong2>(k0 + m * 128, slice_off);
                y1[m] = ksm_gload<ulonglong2>(k1 + m * 128, slice_off);
            });
            __builtin_amdgcn_wave_barrier();
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (m + AHEAD < 16) {
                    y0[m + AHEAD] = ksm_gload<ulonglong2>(k0 + (m + AHEAD) * 128, slice_off);
                    y1[m + AHEAD] = ksm_gload<ulonglong2>(k1 + (m + AHEAD) * 128, slice_off);
                }
                const double2 v = *reinterpret_cast<const double2*>(&lds[pt + ksm_phys(m * 128u)]);      // re-centred when it was written
                mac2(acc0[2 * m], acc1[2 * m], v.x, f64_from_u64(y0[m].x), f64_from_u64(y1[m].x));
                mac2(acc0[2 * m + 1], acc1[2 * m + 1], v.y, f64_from_u64(y0[m].y), f64_from_u64(y1[m].y));
            });
            __builtin_amdgcn_sched_barrier(0);
        } else
        mac_all(x, it);
        KSM_MARK(6);
        if ((step & 7u) == 7u)
            static_for<0, 32>([&](auto rc) { acc0[decltype(rc)::value] = f64_corr(acc0[decltype(rc)::value], fm); acc1[decltype(rc)::value] = f64_corr(acc1[decltype(rc)::value], fm); });
    }

#ifdef KSM_PHASE_PROFILE
    const unsigned long long prof_t0 = prof_t;
#endif
    // ---- canonical results, transposed through the wave's own LDS slice, 16-byte coalesced stores --------------
    u64* go = a.out + (long long)b * a.out_bstride + (long long)k * a.out_cstride + (size_t)h * (KSM_THREADS * 32);
    if constexpr (TEN) { if (ten_row) {
        // fused chain, data row: Q_c = P_c qk^-1 (the keys carry the factor) + tensor term c.  Both accumulators cross the wave's slice
        // (in place: 16 register pairs out, 16 pairs of the coalesced layout back) as re-centred doubles; then ONE sweep over the 16
        // pairs loads a0, b0, a1, b1 and the two key components of the diagonal digit (natural order) with 16-byte loads in a rolling
        // window of W pairs, adds the diagonal digit's term d (.) key with d = a1 (.) b1, adds the tensor terms and stores both
        // components.  The operand rows are cold (HBM) and the workgroup has nothing else to do while they arrive, so every row is read
        // once: with the diagonal digit in the loop (a1, b1 read there and again here) the launch was 0.18 ms slower, with one pass
        // per component (a0, b0 twice as well) 0.3 ms.
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
            xa1[m] = ksm_gload<ulonglong2>(ta1 + ((abl(7) || abl(9)) ? 0 : m * 128), (abl(7) || abl(9)) ? (slice_off & 16u) : slice_off);
            xb1[m] = ksm_gload<ulonglong2>(tb1 + ((abl(7) || abl(9)) ? 0 : m * 128), (abl(7) || abl(9)) ? (slice_off & 16u) : slice_off);
            y0[m] = ksm_gload<double2>(dk0 + (abl(7) ? 0 : m * 128), abl(7) ? (slice_off & 16u) : slice_off);
            y1[m] = ksm_gload<double2>(dk1 + (abl(7) ? 0 : m * 128), abl(7) ? (slice_off & 16u) : slice_off);
            xa0[m] = ksm_gload<ulonglong2>(ta0 + (abl(7) ? 0 : m * 128), abl(7) ? (slice_off & 16u) : slice_off);
            xb0[m] = ksm_gload<ulonglong2>(tb0 + (abl(7) ? 0 : m * 128), abl(7) ? (slice_off & 16u) : slice_off);
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, W>([&](auto ic) { request(ic); });
        __builtin_amdgcn_sched_barrier(0);
        auto cross = [&](double (&acc)[32]) {
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                *reinterpret_cast<double2*>(&lds[p2 + 2 * m]) = make_double2(f64_corr(acc[2 * m], fm), f64_corr(acc[2 * m + 1], fm));
            });
            __builtin_amdgcn_wave_barrier();
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const double2 v = *reinterpret_cast<const double2*>(&lds[pt + ksm_phys(m * 128u)]);
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
            double q0x = acc0[2 * m], q0y = acc0[2 * m + 1], q1x = acc1[2 * m], q1y = acc1[2 * m + 1];      // |.| <= p/2 + 1
            if constexpr (abl(8)) {
                q0x += a0x + b0x + y0[m].x; q0y += a0y + b0y + y0[m].y; q1x += b1x + a1x + y1[m].x; q1y += b1y + a1y + y1[m].y;
            } else {
            // diagonal digit: d = a1 (.) b1 re-centred (what the loop's loader produced), times the key of digit k under modulus k
            const double dx = f64_corr(f64_mulq(f64_corr(a1x, fm), b1x, inv_p, p), fm), dy = f64_corr(f64_mulq(f64_corr(a1y, fm), b1y, inv_p, p), fm);
            mac2(q0x, q1x, dx, y0[m].x, y1[m].x);
            mac2(q0y, q1y, dy, y0[m].y, y1[m].y);
            // tensor terms; canonical factors below p: each product is within (-0.875 p, 0.875 p) (ArithF64::prod_in)
            q0x += f64_mulq(a0x, b0x, inv_p, p);
            q0y += f64_mulq(a0y, b0y, inv_p, p);
            q1x += f64_mulq(a0x, b1x, inv_p, p) + f64_mulq(a1x, b0x, inv_p, p);
            q1y += f64_mulq(a0y, b1y, inv_p, p) + f64_mulq(a1y, b0y, inv_p, p);
            }
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + m * 128, slice_off))), f64_canon(q0x, fm), f64_canon(q0y, fm));
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + a.out_pstride + m * 128, slice_off))), f64_canon(q1x, fm), f64_canon(q1y, fm));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m + W < 16) request(std::integral_constant<int, m + W>{});
        });
    } }
    if constexpr (DG) { if (ten_row) {
        // separate key switch, NTT-form target, data row: both accumulators cross the slice in place, then one sweep adds the diagonal
        // digit's term (the target's limb k itself, coalesced 16-byte loads, times the natural-order copy of key k under modulus k)
        const u64* dg = ksm_uniform(a.diag + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + (size_t)h * (KSM_THREADS * 32));
        const double* dk0 = ksm_uniform(a.diag_keys + (size_t)k * 2 * N + (size_t)h * (KSM_THREADS * 32));
        const double* dk1 = ksm_uniform(dk0 + N);
        constexpr int W = 4;
        ulonglong2 xd[16];
        double2 y0[16], y1[16];
        auto request = [&](auto ic) {
            constexpr int m = decltype(ic)::value;
            xd[m] = ksm_gload<ulonglong2>(dg + m * 128, slice_off);
            y0[m] = ksm_gload<double2>(dk0 + m * 128, slice_off);
            y1[m] = ksm_gload<double2>(dk1 + m * 128, slice_off);
        };
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, W>([&](auto ic) { request(ic); });
        __builtin_amdgcn_sched_barrier(0);
        auto cross = [&](double (&acc)[32]) {
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                *reinterpret_cast<double2*>(&lds[p2 + 2 * m]) = make_double2(f64_corr(acc[2 * m], fm), f64_corr(acc[2 * m + 1], fm));
            });
            __builtin_amdgcn_wave_barrier();
            static_for<0, 16>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                const double2 v = *reinterpret_cast<const double2*>(&lds[pt + ksm_phys(m * 128u)]);
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
            double q0x = acc0[2 * m], q0y = acc0[2 * m + 1], q1x = acc1[2 * m], q1y = acc1[2 * m + 1];      // |.| <= p/2 + 1
            const double dx = f64_corr(f64_from_u64(xd[m].x), fm), dy = f64_corr(f64_from_u64(xd[m].y), fm);
            mac2(q0x, q1x, dx, y0[m].x, y1[m].x);
            mac2(q0y, q1y, dy, y0[m].y, y1[m].y);
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + m * 128, slice_off))), f64_canon(q0x, fm), f64_canon(q0y, fm));
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + a.out_pstride + m * 128, slice_off))), f64_canon(q1x, fm), f64_canon(q1y, fm));
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (m + W < 16) request(std::integral_constant<int, m + W>{});
        });
    } }
    if constexpr (SPLITJ) {
        // slot js, same place inside it; the accumulators are in the coalesced layout already
        go = reinterpret_cast<u64*>(a.part) + (long long)split_js * a.part_jstride + (go - a.out);
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + m * 128, slice_off))), f64_double_to_bits(f64_corr(acc0[2 * m], fm)), f64_double_to_bits(f64_corr(acc0[2 * m + 1], fm)));
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + a.out_pstride + m * 128, slice_off))), f64_double_to_bits(f64_corr(acc1[2 * m], fm)), f64_double_to_bits(f64_corr(acc1[2 * m + 1], fm)));
        });
        return;
    }
    if (!ten_row)
    static_for<0, 2>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 v = make_ulonglong2(f64_canon(c ? acc1[2 * m] : acc0[2 * m], fm), f64_canon(c ? acc1[2 * m + 1] : acc0[2 * m + 1], fm));
            *reinterpret_cast<ulonglong2*>(&lds[p2 + 2 * m]) = v;
        });
        __builtin_amdgcn_wave_barrier();
        static_for<0, 16>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&lds[pt + ksm_phys(m * 128u)]);
            nt_store2(reinterpret_cast<u64*>(const_cast<char*>(at(go + (long long)c * a.out_pstride + m * 128, slice_off))), v.x, v.y);
        });
        __builtin_amdgcn_wave_barrier();
    });
#ifdef KSM_PHASE_PROFILE
    if (a.prof && t == 0) {
        prof_acc[8] = __builtin_readcyclecounter() - prof_t0;
        for (int i = 0; i < 9; i++) atomicAdd(a.prof + i, prof_acc[i]);
    }
#endif
}

// Second stage of the digit-parallel form: out[c][k] = canon( sum_js part[js][c][k]  +  the terms ksmac2's TEN / DG epilogues add ).
// EPI 0: nothing more (coefficient-form target).  EPI 1 (fused chain, data rows): the diagonal digit (a1 (.) b1) key_kk and the tensor terms
// a0 b0 / a0 b1 + a1 b0 (KsMacArgs::ten_a, ten_b, diag_keys; keys prepared times qk^-1).  EPI 2 (NTT-form target): the diagonal digit
// (limb k of the target) times key_kk.  One thread per pair of coefficients; exact integer arithmetic, hence the same canonical words.
template <int EPI>
__global__ __launch_bounds__(256) void ksmac_split_reduce_kernel(KsMacArgs a, unsigned n) {
    const unsigned chunks = n / 512u;
    const unsigned ch = blockIdx.x % chunks, row = blockIdx.x / chunks;
    const unsigned nrows = a.L + 1, k = row % nrows, b = row / nrows;
    const unsigned i = ch * 512u + threadIdx.x * 2u;
    const unsigned mrow = (k == a.L) ? a.table_count - 1 : k;
    const DevModulus dm = a.mods[a.table_start + mrow];
    const F64Mod fm{dm.pd, dm.inv_pd};
    const double p = fm.p, inv_p = fm.inv_p;
    const bool data_row = EPI != 0 && k < a.L;
    const long long base = (long long)b * a.out_bstride + (long long)k * a.out_cstride + i;
    double s0x = 0.0, s0y = 0.0, s1x = 0.0, s1y = 0.0;
    for (unsigned js = 0; js < a.L; ++js) {
        if (data_row && js == k) continue;
        const double* ps = a.part + (long long)js * a.part_jstride + base;
        const double2 v0 = *reinterpret_cast<const double2*>(ps), v1 = *reinterpret_cast<const double2*>(ps + a.out_pstride);
        s0x += v0.x; s0y += v0.y; s1x += v1.x; s1y += v1.y;          // |slot| <= p/2 + 1: the sum of L <= 15 slots stays below 2^53
    }
    s0x = f64_corr(s0x, fm); s0y = f64_corr(s0y, fm); s1x = f64_corr(s1x, fm); s1y = f64_corr(s1y, fm);
    auto mac2 = [&](double& a0, double& a1, double v, double y0, double y1) {
        const double h0 = v * y0, h1 = v * y1;
        const double l0 = __builtin_fma(v, y0, -h0), l1 = __builtin_fma(v, y1, -h1);
        const double q0 = __builtin_rint(h0 * inv_p), q1 = __builtin_rint(h1 * inv_p);
        a0 += __builtin_fma(-q0, p, h0) + l0;
        a1 += __builtin_fma(-q1, p, h1) + l1;
    };
    if (data_row) {
        // the diagonal digit's key block (key k, modulus k) straight from the caller's keys
        const u64* dk = a.raw.p[k] + (size_t)k * n + i;
        const ulonglong2 r0 = *reinterpret_cast<const ulonglong2*>(dk), r1 = *reinterpret_cast<const ulonglong2*>(dk + a.raw_pstride);
        const double2 y0 = make_double2(f64_from_u64(r0.x), f64_from_u64(r0.y)), y1 = make_double2(f64_from_u64(r1.x), f64_from_u64(r1.y));
        if constexpr (EPI == 1) {
            const size_t toff = (size_t)b * a.ten_bstride + (size_t)k * n + i;
            const ulonglong2 xa0 = *reinterpret_cast<const ulonglong2*>(a.ten_a + toff), xb0 = *reinterpret_cast<const ulonglong2*>(a.ten_b + toff);
            const ulonglong2 xa1 = *reinterpret_cast<const ulonglong2*>(a.ten_a + toff + a.ten_pstride), xb1 = *reinterpret_cast<const ulonglong2*>(a.ten_b + toff + a.ten_pstride);
            const double a0x = f64_from_u64(xa0.x), a0y = f64_from_u64(xa0.y), b0x = f64_from_u64(xb0.x), b0y = f64_from_u64(xb0.y);
            const double a1x = f64_from_u64(xa1.x), a1y = f64_from_u64(xa1.y), b1x = f64_from_u64(xb1.x), b1y = f64_from_u64(xb1.y);
            const double dx = f64_corr(f64_mulq(f64_corr(a1x, fm), b1x, inv_p, p), fm), dy = f64_corr(f64_mulq(f64_corr(a1y, fm), b1y, inv_p, p), fm);
            mac2(s0x, s1x, dx, y0.x, y1.x);
            mac2(s0y, s1y, dy, y0.y, y1.y);
            // relinearize's division by the special prime on the whole inner product (the batched form carries qk^-1 in its prepared keys)
            const double sc = f64_from_u64(a.split_scale[k].x);
            s0x = f64_mulq(f64_corr(s0x, fm), sc, inv_p, p); s0y = f64_mulq(f64_corr(s0y, fm), sc, inv_p, p);
            s1x = f64_mulq(f64_corr(s1x, fm), sc, inv_p, p); s1y = f64_mulq(f64_corr(s1y, fm), sc, inv_p, p);
            s0x += f64_mulq(a0x, b0x, inv_p, p);
            s0y += f64_mulq(a0y, b0y, inv_p, p);
            s1x += f64_mulq(a0x, b1x, inv_p, p) + f64_mulq(a1x, b0x, inv_p, p);
            s1y += f64_mulq(a0y, b1y, inv_p, p) + f64_mulq(a1y, b0y, inv_p, p);
        } else if constexpr (EPI == 2) {
            const ulonglong2 xd = *reinterpret_cast<const ulonglong2*>(a.diag + (long long)b * a.diag_bstride + (long long)k * a.diag_cstride + i);
            mac2(s0x, s1x, f64_corr(f64_from_u64(xd.x), fm), y0.x, y1.x);
            mac2(s0y, s1y, f64_corr(f64_from_u64(xd.y), fm), y0.y, y1.y);
        }
    }
    u64* go = a.out + base;
    *reinterpret_cast<ulonglong2*>(go) = make_ulonglong2(f64_canon(s0x, fm), f64_canon(s0y, fm));
    *reinterpret_cast<ulonglong2*>(go + a.out_pstride) = make_ulonglong2(f64_canon(s1x, fm), f64_canon(s1y, fm));
}

}  // namespace troyn
