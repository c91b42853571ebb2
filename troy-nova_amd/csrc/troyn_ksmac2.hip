// kernel instantiations of the second-generation key-switch inner product (ksmac_kernels.hpp)
#include "launch.hpp"

namespace troyn {

void launch_ksmac2(unsigned log_n, size_t batch, unsigned rows, const KsMacArgs& a, hipStream_t s, bool digits_f64, bool wide_digits) {
    const dim3 block(KSM_THREADS);
    // order 3 (bands of two rows): an odd row count is padded with workgroups that exit
    const size_t grid_rows = a.grouped == 3 ? (rows + 1) / 2 * 2 : rows;
#ifdef KSM_NLC_OFF      // A/B builds (tools/ksmac_variants.sh)
    const bool nlc = false;
#else
    const bool nlc = a.no_load_corr != 0;
#endif
#define KSMAC2_CASE(LOGN, TILES)                                                                                            \
    if (digits_f64 && a.ten_a && LOGN == 14 && nlc) hipLaunchKernelGGL((ksmac2_kernel<14, true, 0, false, true, false, false, false, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else if (digits_f64 && a.ten_a) hipLaunchKernelGGL((ksmac2_kernel<LOGN, true, 0, false, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else if (a.ten_a) hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, true, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); /* fused chain of a mixed chain: u64 digits, wide ones reduced while loading */ \
    else if (!digits_f64 && !wide_digits && !a.diag) hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, false, false, false, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else if (!digits_f64 && !wide_digits && a.diag && a.diag_keys) hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, false, false, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else if (digits_f64) hipLaunchKernelGGL((ksmac2_kernel<LOGN, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else if (wide_digits && a.diag && a.diag_keys) hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, true, false, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); /* mixed chain, NTT-form target: DG epilogue */ \
    else if (wide_digits) hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, true>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a); \
    else hipLaunchKernelGGL((ksmac2_kernel<LOGN, false>), dim3((unsigned)(batch * grid_rows * TILES)), block, 0, s, a);
    if (log_n == 15) { KSMAC2_CASE(15, 4) }
    else if (log_n == 14) { KSMAC2_CASE(14, 2) }
    else { KSMAC2_CASE(13, 1) }
#undef KSMAC2_CASE
}

// digit-parallel form (small batches): one workgroup per (item, row, tile, digit) leaves its accumulators in a.part, then the reducer adds the
// slots and the epilogue terms (epi: 0 none, 1 fused chain, 2 NTT-form target)
void launch_ksmac2_split(unsigned log_n, size_t batch, const KsMacArgs& a, hipStream_t s, bool digits_f64, int epi) {
    const dim3 block(KSM_THREADS);
    const size_t rows = a.L + 1;
#define KSMAC2_SPLIT_CASE(LOGN, TILES)                                                                                            \
    if (digits_f64) hipLaunchKernelGGL((ksmac2_kernel<LOGN, true, 0, false, false, false, true, true>), dim3((unsigned)(batch * rows * TILES * a.L)), block, 0, s, a); \
    else hipLaunchKernelGGL((ksmac2_kernel<LOGN, false, 0, false, false, false, true, true>), dim3((unsigned)(batch * rows * TILES * a.L)), block, 0, s, a);
    if (log_n == 15) { KSMAC2_SPLIT_CASE(15, 4) }
    else if (log_n == 14) { KSMAC2_SPLIT_CASE(14, 2) }
    else { KSMAC2_SPLIT_CASE(13, 1) }
#undef KSMAC2_SPLIT_CASE
    const unsigned n = 1u << log_n;
    const dim3 grid((unsigned)(batch * rows * (n / 512u)));
    if (epi == 1) hipLaunchKernelGGL(ksmac_split_reduce_kernel<1>, grid, dim3(256), 0, s, a, n);
    else if (epi == 2) hipLaunchKernelGGL(ksmac_split_reduce_kernel<2>, grid, dim3(256), 0, s, a, n);
    else hipLaunchKernelGGL(ksmac_split_reduce_kernel<0>, grid, dim3(256), 0, s, a, n);
}

void launch_ksmac_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned polys, unsigned n, double* out, unsigned blocks, hipStream_t s,
                               const ulonglong2* scale, const DevModulus* mods, unsigned scale_rows, double* diag_out) {
    hipLaunchKernelGGL(ksmac_prepare_keys_kernel, dim3(blocks), dim3(256), 0, s, kp, L, polys, n, out, scale, mods, scale_rows, diag_out);
}

}  // namespace troyn
