// launch.hpp -- entry points of the translation units that hold the kernel instantiations (one per kernel family / arithmetic
// policy, so that they compile in parallel).  troyn.hip (the C-ABI) only calls these.
#pragma once
#include <hip/hip_runtime.h>
#include "ntt_kernels.hpp"
#include "ksmac_kernels.hpp"
#include "behz2_kernels.hpp"

// a launch of at most CUs / TROYN_SMALL_LP_FACTOR limb-polynomials counts as small (two-pass transforms at N = 8192 / 16384, merged tails, no split by class)
#ifndef TROYN_SMALL_LP_FACTOR
#define TROYN_SMALL_LP_FACTOR 2      // measured 8 | 2 | 1: eight ciphertexts at N = 16384, fused chain 100 | 75 | 75 us; 128: 473 | 469 | 481
#endif

namespace troyn {

// what a launch of the NTT family needs besides its arguments: the stream and the plan's A/B options that select kernel variants
// (read once when the plan is created, troyn.hip TroynOptions -- no environment access on the launch path)
struct LaunchCtx {
    hipStream_t s;
    int half_mask;             // TROYN_NTT_HALF: half-word LDS tile variants of the whole-limb N = 16384 FP64 transforms; -1 = default
    bool small_two_pass_off;   // TROYN_NTT_SMALL_TWO_PASS=0
    int tensor_wgs;            // TROYN_TENSOR_WGS (8 = default)
};

// optimised transforms of one arithmetic class (false: no kernel for this size -> ntt_generic); each class is instantiated in two
// translation units, N <= 8192 and N >= 16384
#define TROYN_DECL_NTT_UNIT(SUFFIX)                                                                                                   \
    bool launch_ntt_##SUFFIX(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, const LaunchCtx& lc, u64* scratch);          \
    bool launch_ks_mac_##SUFFIX(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc);                    \
    bool launch_tensor_##SUFFIX(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc);
TROYN_DECL_NTT_UNIT(f64_small) TROYN_DECL_NTT_UNIT(f64_large) TROYN_DECL_NTT_UNIT(u64_small) TROYN_DECL_NTT_UNIT(u64_large)
#undef TROYN_DECL_NTT_UNIT
void launch_ntt_generic(const NttArgs& a, unsigned log_n, bool inverse, size_t limb_polys, const LaunchCtx& lc);
// single objects through the fused chain at N = 8192 / 16384 (troyn_mrr_small.hip): one pass of a two-pass transform, and the strided passes of the
// chain's tail (special rows, dropped limb, output limbs) as one launch
void launch_ntt_f64_pass14(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
void launch_ntt_f64_pass13(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
void launch_ntt_u64_pass14(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
void launch_ntt_u64_pass13(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
void launch_ntt_f64_pass15(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
void launch_ntt_u64_pass15(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc);
inline void launch_ntt_f64_small_pass(unsigned log_n, int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (log_n == 13) launch_ntt_f64_pass13(which, a, limb_polys, lc); else if (log_n == 14) launch_ntt_f64_pass14(which, a, limb_polys, lc); else launch_ntt_f64_pass15(which, a, limb_polys, lc);
}
inline void launch_ntt_u64_small_pass(unsigned log_n, int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (log_n == 13) launch_ntt_u64_pass13(which, a, limb_polys, lc); else if (log_n == 14) launch_ntt_u64_pass14(which, a, limb_polys, lc); else launch_ntt_u64_pass15(which, a, limb_polys, lc);
}
void launch_mrr_quartet(unsigned log_n, size_t batch, const NttArgs& sp, const NttArgs& la, const NttArgs& ta, hipStream_t s, bool limb_parallel);
void launch_mrr_quartet_load(unsigned log_n, size_t groups, const NttArgs& iv, const NttArgs& fw, hipStream_t s, bool f64);
inline bool launch_ntt_f64(unsigned log_n, const NttArgs& a, size_t lp, bool inverse, const LaunchCtx& lc, u64* scratch) {
    return log_n <= 13 ? launch_ntt_f64_small(log_n, a, lp, inverse, lc, scratch) : launch_ntt_f64_large(log_n, a, lp, inverse, lc, scratch);
}
inline bool launch_ntt_u64(unsigned log_n, const NttArgs& a, size_t lp, bool inverse, const LaunchCtx& lc, u64* scratch) {
    return log_n <= 13 ? launch_ntt_u64_small(log_n, a, lp, inverse, lc, scratch) : launch_ntt_u64_large(log_n, a, lp, inverse, lc, scratch);
}
// first-generation fused key-switch inner product (ks_mac_kernel)
inline bool launch_ks_mac_f64(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    return log_n <= 13 ? launch_ks_mac_f64_small(log_n, a, kp, blocks, lc) : launch_ks_mac_f64_large(log_n, a, kp, blocks, lc);
}
inline bool launch_ks_mac_u64(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    return log_n <= 13 ? launch_ks_mac_u64_small(log_n, a, kp, blocks, lc) : launch_ks_mac_u64_large(log_n, a, kp, blocks, lc);
}
// tensor product fused with the transforms (tensor_core_kernel); stage 0 / 2: the strided passes of the two-pass sizes
inline bool launch_tensor_f64(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    return log_n <= 13 ? launch_tensor_f64_small(log_n, stage, a, b, d, batch, lc) : launch_tensor_f64_large(log_n, stage, a, b, d, batch, lc);
}
inline bool launch_tensor_u64(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    return log_n <= 13 ? launch_tensor_u64_small(log_n, stage, a, b, d, batch, lc) : launch_tensor_u64_large(log_n, stage, a, b, d, batch, lc);
}
// second-generation key-switch inner product (ksmac2_kernel, log_n = 13 / 14 / 15) and its key preparation
// digits_f64: the digit rows hold doubles (fused chain: NTT_FLAG_STORE_F64) instead of u64 words; wide_digits: some digit limb is 2^50 or
// wider (mixed chains, a.row_mask selects the rows of moduli < 2^50): digits are reduced with integer arithmetic while loading
void launch_ksmac2(unsigned log_n, size_t batch, unsigned rows, const KsMacArgs& a, hipStream_t s, bool digits_f64 = false, bool wide_digits = false);
void launch_ksmac2_split(unsigned log_n, size_t batch, const KsMacArgs& a, hipStream_t s, bool digits_f64, int epi);
// scale (optional): Shoup pairs of the factor the rows r < scale_rows of every key component are multiplied by (fused chain: qk^-1 mod q_r)
void launch_ksmac_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned polys, unsigned n, double* out, unsigned blocks, hipStream_t s,
                               const ulonglong2* scale = nullptr, const DevModulus* mods = nullptr, unsigned scale_rows = 0,
                               double* diag_out = nullptr);   // diag_out: [j][2][N], block (key j, modulus j) in natural order
// integer inner product for the rows of moduli >= 2^50 (ksmaci_kernel, log_n = 13 / 14 / 15) and its key preparation; a.row_mask = those rows
struct KsMacIArgs;
void launch_ksmaci(unsigned log_n, size_t batch, const KsMacIArgs& a, hipStream_t s, int epi);
void launch_ksmaci_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned K, unsigned n, unsigned long long row_mask, ulonglong2* out, unsigned blocks, hipStream_t s,
                                const ulonglong2* scale, const DevModulus* mods, unsigned scale_rows, ulonglong2* diag_out);
// second-generation BEHZ conversions (L = 1 .. 16)
// aux50: the auxiliary base holds primes below 2^50 (Behz2Dev::NB > L of them) instead of the reference's 61-bit primes
void launch_behz2_lift(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* src, u64* dst, bool aux50 = false);
bool launch_behz2_lift_pass1(unsigned L, size_t items, hipStream_t s, const Behz2Dev& c, const u64* src, u64* dst_q, u64* dst_bsk,
                             const double* tw_q, const double* tw_aux, const DevModulus* q_mods, const DevModulus* aux_mods);
bool launch_behz2_floor_pass2(unsigned L, size_t items, hipStream_t s, const Behz2Dev& c, const u64* in_q, const u64* in_bsk, u64* out,
                              const double* tw_q, const double* tw_aux, const DevModulus* q_mods, const DevModulus* aux_mods);
void launch_behz2_floor(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* in_q, const u64* in_bsk, u64* out, bool aux50 = false);

}  // namespace troyn
