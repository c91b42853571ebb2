// kernel instantiations of the ArithU64 policy, N <= 8192 (ntt_launch.inl)
#define TROYN_NTT_PART 1
#include "ntt_launch.inl"

namespace troyn {

bool launch_ntt_u64_small(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, hipStream_t s, u64* scratch) {
    return launch_ntt_optimised<ArithU64>(log_n, a, limb_polys, inverse, s, scratch);
}
bool launch_ks_mac_u64_small(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, hipStream_t s) {
    return launch_ks_mac_t<ArithU64>(log_n, a, kp, blocks, s);
}
bool launch_tensor_u64_small(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, hipStream_t s) {
    return launch_tensor_class<ArithU64>(log_n, stage, a, b, d, batch, s);
}

void launch_ntt_generic(const NttArgs& a, unsigned log_n, bool inverse, size_t limb_polys, hipStream_t s) {
    hipLaunchKernelGGL(ntt_generic_kernel, dim3((unsigned)limb_polys), dim3(256), 0, s, a, log_n, inverse ? 1 : 0);
}

}  // namespace troyn
