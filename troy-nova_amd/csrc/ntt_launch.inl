// ntt_launch.inl -- launch code of the NTT-family kernels (ntt_kernels.hpp), templated on the arithmetic policy.  Included by
// troyn_ntt_f64.hip and troyn_ntt_u64.hip, which instantiate it for one policy each so that the two sets of kernels compile in
// parallel; troyn.hip calls the non-template entry points declared in launch.hpp.
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <hip/hip_runtime.h>
#include "ntt_kernels.hpp"
#include "launch.hpp"

// TROYN_NTT_PART: 1 = the sizes N <= 8192 only, 2 = N >= 16384 only (one translation unit each, so that they compile in parallel)
#ifndef TROYN_NTT_PART
#define TROYN_NTT_PART 0
#endif
#define TROYN_NTT_SMALL (TROYN_NTT_PART != 2)
#define TROYN_NTT_LARGE (TROYN_NTT_PART != 1)
#ifndef TROYN_SMALL_EB
#define TROYN_SMALL_EB 3      // small launches at N = 16384 (two-pass form): 512 threads x 8 coefficients (single fused op 82 -> 78 us; 4: 82, 2: 81)
#endif

namespace troyn {

// IOM: compile-time variant of the kernel (0 plain, 1 key-switch tail, 2 rescale, 3..5 fused chain).  Whole-limb N = 16384 tiles of the
// FP64 policy can run with half-word LDS tiles (two workgroups per CU, ntt_pass_body HALF); they pay off where the kernel fits 64
// registers: the plain forward transform (+18 %, 3.8 -> 4.5 TB/s), the fused tail + rescale (+2 % on the headline) and, measured in round 3
// through the three-call path, the key-switch tail and the rescale (relinearize N = 16384 +2 %, multiply + relinearize + rescale as three calls
// +4 %).  The inverse variants need 76-82 registers; the plain one fits since the per-lane twiddles of its first round are loaded in three
// steps (ntt_pass_body: an opaque zero derived from the previous layer's result enters the table index, so the loads cannot be hoisted;
// scratch 68 -> 12 bytes, NTT + dyadic + INTT +2 %) and is on by default (mask 0x0127); the fused chain's MULPAIR / LAST_LIMB variants still
// spill 64 / 108 bytes in their loaders and stay on full-word tiles.  TROYN_NTT_HALF=<mask> (bit (INV ? 8 : 0) + IOM) selects variants for A/B runs.
// CUs of the current device (cached per host thread)
static unsigned ntt_cu_count() {
    static thread_local int cached_dev = -1;
    static thread_local unsigned cached = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev != cached_dev) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        cached = (unsigned)cus; cached_dev = dev;
    }
    return cached;
}


template <class A, int LOGN, int LO, int G, int TB, int EB, bool INV, bool FIRST, bool LAST, int IOM>
static void launch_variant(const NttArgs& a, dim3 grid, dim3 block, size_t extra_lds, const LaunchCtx& lc) {
    if constexpr (std::is_same<A, ArithF64>::value && LOGN == 14 && TB == 14 && LO == 0) {
        // bit (INV ? 8 : 0) + IOM selects the variant (plan option TROYN_NTT_HALF, -1: the default below)
        // half-word tiles buy a second workgroup per CU at the price of three barriers per exchange: with no more workgroups than CUs there is
        // nobody to share the CU with and the full-word tile is the faster one (a single ciphertext: three calls 174 -> 154 us per op); an
        // explicit TROYN_NTT_HALF is obeyed at every size
        const int half = lc.half_mask >= 0 ? lc.half_mask : (grid.x > ntt_cu_count() ? 0x0167 : 0);     // (bit 6: NTT_FUSED_TAIL_RESCALE_W follows bit 5)
        constexpr int half_bit = IOM == NTT_IOM_CENTRALIZE ? 0 : (INV ? 8 : 0) + IOM;                     // (the centralising loader follows the plain forward kernel)
        if ((half >> half_bit) & 1) { hipLaunchKernelGGL((ntt_pass_kernel<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, IOM, true>), grid, block, extra_lds, lc.s, a); return; }
    }
    hipLaunchKernelGGL((ntt_pass_kernel<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, IOM, false>), grid, block, extra_lds, lc.s, a);
}

template <class A, int LOGN, int LO, int G, int TB, int EB, bool INV, bool FIRST, bool LAST>
static void launch_pass(const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    const unsigned tiles = 1u << (LOGN - TB);
    dim3 grid((unsigned)(limb_polys * tiles)), block(1u << (TB - EB));
    constexpr int extra_lds = 0;
    // the fused prologue / epilogue is a compile-time variant of the forward kernels (no runtime branches per word)
    const unsigned lm = FIRST ? a.load_mode : 0u, sm = LAST ? a.store_mode : 0u;
    if constexpr (LOGN >= 13 && LOGN <= 15) {
        // kernels of the fused multiply -> relinearize -> rescale chain (NttFused): whole-limb at N <= 16384, both passes at N = 32768; both
        // policies since round 5 (chains with moduli of 2^50 and more run the chain per modulus class); the _W variants are FP64 kernels
        if constexpr (INV) {
            if (a.fused_mode == NTT_FUSED_MULPAIR) { launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_FUSED_MULPAIR>(a, grid, block, 0, lc); return; }
            if (a.fused_mode == NTT_FUSED_LAST_LIMB) { launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_FUSED_LAST_LIMB>(a, grid, block, 0, lc); return; }
            if constexpr (std::is_same<A, ArithF64>::value)
                if (a.fused_mode == NTT_FUSED_LAST_LIMB_W) { launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_FUSED_LAST_LIMB_W>(a, grid, block, 0, lc); return; }
        } else {
            if (a.fused_mode == NTT_FUSED_TAIL_RESCALE) { launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_FUSED_TAIL_RESCALE>(a, grid, block, 0, lc); return; }
            if constexpr (std::is_same<A, ArithF64>::value)
                if (a.fused_mode == NTT_FUSED_TAIL_RESCALE_W) { launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_FUSED_TAIL_RESCALE_W>(a, grid, block, 0, lc); return; }
        }
    }
    if constexpr (INV && LAST) {
        if (sm == NTT_STORE_KS_FINISH) {   // coefficient-form key-switch tail: the finish runs in the inverse transform's epilogue
            launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, 1>(a, grid, block, (size_t)extra_lds, lc);
            return;
        }
    }
    if constexpr (!INV && FIRST) {
        if (lm == NTT_LOAD_CENTRALIZE) {     // plaintext -> NTT form in one launch (troyn_plain_centralize_ntt)
            launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, NTT_IOM_CENTRALIZE>(a, grid, block, (size_t)extra_lds, lc);
            return;
        }
    }
    if constexpr (!INV) {
        if (lm == NTT_LOAD_KS_ROUND || sm == NTT_STORE_KS_FINISH) {
            launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, 1>(a, grid, block, (size_t)extra_lds, lc);
            return;
        }
        if (lm == NTT_LOAD_RESCALE || sm == NTT_STORE_RESCALE) {
            launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, 2>(a, grid, block, (size_t)extra_lds, lc);
            return;
        }
    }
    launch_variant<A, LOGN, LO, G, TB, EB, INV, FIRST, LAST, 0>(a, grid, block, (size_t)extra_lds, lc);
}

// single pass: whole limb in one tile
template <class A, int LOGN, int EB>
static void launch_single(const NttArgs& a, size_t lp, bool inv, const LaunchCtx& lc) {
    if (inv) launch_pass<A, LOGN, 0, LOGN, LOGN, EB, true, true, true>(a, lp, lc);
    else launch_pass<A, LOGN, 0, LOGN, LOGN, EB, false, true, true>(a, lp, lc);
}

// two passes: G1 strided layers (columns of 2^(TB-G1) consecutive words), then contiguous 2^TB chunks.
// The pass between them lives in `scratch` ([limb-polynomial][N], contiguous) when one is given, else in `out`: a fused
// epilogue that READS the old destination (key-switch tail with AddInplace / OverwriteExceptFirst) must not find the first
// pass's intermediate words there.
template <class A, int LOGN, int TB, int EB>
static void launch_two_pass(const NttArgs& a, size_t lp, bool inv, const LaunchCtx& lc, u64* scratch) {
    constexpr int G1 = LOGN - TB;
    NttArgs first = a, second = a;
    if (scratch) {
        first.out = scratch;
        first.out_cstride = (long long)1 << LOGN;
        first.out_pstride = (long long)a.ncomp << LOGN;
        first.out_bstride = (long long)a.pcount * a.ncomp << LOGN;
    }
    second.in = first.out;
    second.in_bstride = first.out_bstride; second.in_pstride = first.out_pstride; second.in_cstride = first.out_cstride;
    second.reduce_input = 0;
    if (!inv) {
        launch_pass<A, LOGN, 0, G1, TB, EB, false, true, false>(first, lp, lc);
        launch_pass<A, LOGN, G1, TB, TB, EB, false, false, true>(second, lp, lc);
    } else {
        launch_pass<A, LOGN, G1, TB, TB, EB, true, true, false>(first, lp, lc);
        launch_pass<A, LOGN, 0, G1, TB, EB, true, false, true>(second, lp, lc);
    }
}

template <class A>
static bool launch_ntt_optimised(unsigned log_n, const NttArgs& a, size_t lp, bool inverse, const LaunchCtx& lc, u64* scratch) {
    // N = 4096 / 8192: 8 coefficients per thread (EB = 3) doubles the waves per tile, so a CU holds 32 waves instead
    // of 16; measured 5-14 % faster than EB = 4 despite the extra LDS exchange.  N = 16384 needs EB = 4 to fit one
    // workgroup (1024 threads x 16 coefficients).
    switch (log_n) {
#if TROYN_NTT_SMALL
        case 10: launch_single<A, 10, 4>(a, lp, inverse, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 11: launch_single<A, 11, 4>(a, lp, inverse, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 12: launch_single<A, 12, 3>(a, lp, inverse, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 13:
            // (the same for N = 8192: 4 workgroups of 2048 words per limb and pass)
            if (lp * TROYN_SMALL_LP_FACTOR <= ntt_cu_count() && !lc.small_two_pass_off) launch_two_pass<A, 13, 11, TROYN_SMALL_EB>(a, lp, inverse, lc, scratch);
            else launch_single<A, 13, 3>(a, lp, inverse, lc);
            return true;
#endif
#if TROYN_NTT_LARGE
        case 14:
            // A whole-limb tile puts a 16384-point transform on ONE CU: 15-23 us however few limbs the launch has.  Launches that leave most
            // of the chip idle (a single ciphertext: 2-10 limb-polynomials) take the two-pass form of the larger rings instead -- 4 workgroups
            // per limb and pass, ~3x shorter; TROYN_NTT_SMALL_TWO_PASS=0 keeps the single pass (A/B runs, tests).  Results are the same words.
            if (lp * TROYN_SMALL_LP_FACTOR <= ntt_cu_count() && !lc.small_two_pass_off) launch_two_pass<A, 14, 12, TROYN_SMALL_EB>(a, lp, inverse, lc, scratch);
            else launch_single<A, 14, 4>(a, lp, inverse, lc);
            return true;
#endif
#if TROYN_NTT_LARGE
        case 15: launch_two_pass<A, 15, 12, 4>(a, lp, inverse, lc, scratch); return true;
#endif
#if TROYN_NTT_LARGE
        case 16: launch_two_pass<A, 16, 12, 4>(a, lp, inverse, lc, scratch); return true;
#endif
#if TROYN_NTT_LARGE
        case 17: launch_two_pass<A, 17, 12, 4>(a, lp, inverse, lc, scratch); return true;
#endif
        default: return false;
    }
}

template <class A>
static bool launch_ks_mac_t(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    switch (log_n) {
#if TROYN_NTT_SMALL
        case 10: hipLaunchKernelGGL((ks_mac_kernel<A, 10, 4>), dim3((unsigned)blocks), dim3(1u << 6), 0, lc.s, a, kp); return true;
#endif
#if TROYN_NTT_SMALL
        case 11: hipLaunchKernelGGL((ks_mac_kernel<A, 11, 4>), dim3((unsigned)blocks), dim3(1u << 7), 0, lc.s, a, kp); return true;
#endif
#if TROYN_NTT_SMALL
        case 12: hipLaunchKernelGGL((ks_mac_kernel<A, 12, 4>), dim3((unsigned)blocks), dim3(1u << 8), 0, lc.s, a, kp); return true;
#endif
        // (N = 8192 / 16384: ksmac2_kernel / ksmaci_kernel; the first-generation instantiations of those sizes left the library in round 5)
        default: return false;
    }
}

template <class A, int LOGN, int TB, int EB>
static void tensor_stage_t(int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    constexpr int G1 = LOGN - TB;
    if constexpr (G1 > 0) {
        if (stage == 0) { launch_pass<A, LOGN, 0, G1, TB, EB, false, true, false>(a, batch * a.pcount * a.ncomp, lc); return; }
        if (stage == 2) { launch_pass<A, LOGN, 0, G1, TB, EB, true, false, true>(a, batch * a.pcount * a.ncomp, lc); return; }
    }
    if (stage != 1) return;
    const dim3 grid((unsigned)((batch * a.ncomp) << G1)), block(1u << (TB - EB));
    if constexpr (G1 > 0) {
        // three polynomials are held in registers next to the transform in flight.  Default: 512-thread workgroups x 8 coefficients
        // (100-111 registers, no spills, two workgroups = 16 waves per CU).  TROYN_TENSOR_WGS=3 / 2: 256 threads x 16 coefficients with
        // three (168 registers, 11-28 spilled) / two (no spills) workgroups per CU -- measured 1.8 % / 3 % slower at N = 32768 L = 10.
        const int wgs = lc.tensor_wgs;
        if (wgs == 8) {   // 512-thread workgroups x 8 coefficients: 100-111 registers, no spills, two workgroups (16 waves) per CU
            hipLaunchKernelGGL((tensor_core_kernel<A, LOGN, TB, 3, 2>), grid, dim3(1u << (TB - 3)), 0, lc.s, a, b, d);
        } else if (wgs == 3) hipLaunchKernelGGL((tensor_core_kernel<A, LOGN, TB, EB, 3>), grid, block, 0, lc.s, a, b, d);
        else hipLaunchKernelGGL((tensor_core_kernel<A, LOGN, TB, EB, 2>), grid, block, 0, lc.s, a, b, d);
    } else hipLaunchKernelGGL((tensor_core_kernel<A, LOGN, TB, EB, 1>), grid, block, 0, lc.s, a, b, d);
}

template <class A>
static bool launch_tensor_class(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    switch (log_n) {
#if TROYN_NTT_SMALL
        case 10: tensor_stage_t<A, 10, 10, 4>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 11: tensor_stage_t<A, 11, 11, 4>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 12: tensor_stage_t<A, 12, 12, 3>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_SMALL
        case 13: tensor_stage_t<A, 13, 13, 3>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_LARGE
        case 14: tensor_stage_t<A, 14, 14, 4>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_LARGE
        case 15: tensor_stage_t<A, 15, 12, 4>(stage, a, b, d, batch, lc); return true;
#endif
#if TROYN_NTT_LARGE
        case 16: tensor_stage_t<A, 16, 12, 4>(stage, a, b, d, batch, lc); return true;
#endif
        default: return false;
    }
}

}  // namespace troyn
