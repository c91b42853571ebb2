// kernel instantiations of the second-generation key-switch inner product (ksmac_kernels.hpp)
#include "launch.hpp"

namespace troyn {

void launch_ksmac2(unsigned log_n, size_t batch, unsigned rows, const KsMacArgs& a, hipStream_t s) {
    if (log_n == 15) hipLaunchKernelGGL((ksmac2_kernel<15, false>), dim3((unsigned)(batch * rows * 4)), dim3(KSM_THREADS), 0, s, a);
    else if (log_n == 14) hipLaunchKernelGGL((ksmac2_kernel<14, false>), dim3((unsigned)(batch * rows * 2)), dim3(KSM_THREADS), 0, s, a);
    else hipLaunchKernelGGL((ksmac2_kernel<13, false>), dim3((unsigned)(batch * rows)), dim3(KSM_THREADS), 0, s, a);
}

void launch_ksmac_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned polys, unsigned n, double* out, unsigned blocks, hipStream_t s) {
    hipLaunchKernelGGL(ksmac_prepare_keys_kernel, dim3(blocks), dim3(256), 0, s, kp, L, polys, n, out);
}

}  // namespace troyn
