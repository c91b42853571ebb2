// kernel instantiations of the integer key-switch inner product for rows of moduli >= 2^50 (ksmaci_kernels.hpp)
#include "launch.hpp"
#include "ksmaci_kernels.hpp"

namespace troyn {

// epi: 0 every digit in the loop (coefficient-form target), 1 DG (NTT-form target), 2 TEN (fused chain)
void launch_ksmaci(unsigned log_n, size_t batch, const KsMacIArgs& a, hipStream_t s, int epi) {
    const dim3 block(KSM_THREADS);
    const size_t rows = (size_t)__builtin_popcountll(a.row_mask);
#define KSMACI_CASE(LOGN, TILES)                                                                                            \
    if (epi == 2) hipLaunchKernelGGL((ksmaci_kernel<LOGN, 2>), dim3((unsigned)(batch * rows * TILES)), block, 0, s, a);      \
    else if (epi == 1) hipLaunchKernelGGL((ksmaci_kernel<LOGN, 1>), dim3((unsigned)(batch * rows * TILES)), block, 0, s, a); \
    else hipLaunchKernelGGL((ksmaci_kernel<LOGN, 0>), dim3((unsigned)(batch * rows * TILES)), block, 0, s, a);
    if (log_n == 15) { KSMACI_CASE(15, 4) }
    else if (log_n == 14) { KSMACI_CASE(14, 2) }
    else { KSMACI_CASE(13, 1) }
#undef KSMACI_CASE
}

void launch_ksmaci_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned K, unsigned n, unsigned long long row_mask, ulonglong2* out, unsigned blocks, hipStream_t s,
                                const ulonglong2* scale, const DevModulus* mods, unsigned scale_rows, ulonglong2* diag_out) {
    hipLaunchKernelGGL(ksmaci_prepare_keys_kernel, dim3(blocks), dim3(256), 0, s, kp, L, K, n, row_mask, out, scale, mods, scale_rows, diag_out);
}

}  // namespace troyn
