// kernel instantiations of the second-generation BEHZ conversions (behz2_kernels.hpp): one per base size and modulus class
#include "launch.hpp"

namespace troyn {

#define BEHZ2_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)

void launch_behz2_lift(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* src, u64* dst, bool aux50) {
    switch (L) {
#define X(N) case N: if (aux50) hipLaunchKernelGGL((behz2_lift_kernel<N, true, true>), dim3(grid), dim3(256), 0, s, chunks, c, src, dst); \
                     else if (smallq) hipLaunchKernelGGL((behz2_lift_kernel<N, true>), dim3(grid), dim3(256), 0, s, chunks, c, src, dst); \
                     else hipLaunchKernelGGL((behz2_lift_kernel<N, false>), dim3(grid), dim3(256), 0, s, chunks, c, src, dst); break;
        BEHZ2_CASES(X)
#undef X
        default: break;
    }
}

void launch_behz2_floor(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* in_q, const u64* in_bsk, u64* out, bool aux50) {
    switch (L) {
#define X(N) case N: if (aux50) hipLaunchKernelGGL((behz2_floor_kernel<N, true, true>), dim3(grid), dim3(256), 0, s, chunks, c, in_q, in_bsk, out); \
                     else if (smallq) hipLaunchKernelGGL((behz2_floor_kernel<N, true>), dim3(grid), dim3(256), 0, s, chunks, c, in_q, in_bsk, out); \
                     else hipLaunchKernelGGL((behz2_floor_kernel<N, false>), dim3(grid), dim3(256), 0, s, chunks, c, in_q, in_bsk, out); break;
        BEHZ2_CASES(X)
#undef X
        default: break;
    }
}

}  // namespace troyn
