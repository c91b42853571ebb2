// kernel instantiations of the ArithU64 policy, N <= 8192 (ntt_launch.inl)
#define TROYN_NTT_PART 1
#include "ntt_launch.inl"

namespace troyn {

bool launch_ntt_u64_small(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, const LaunchCtx& lc, u64* scratch) {
    return launch_ntt_optimised<ArithU64>(log_n, a, limb_polys, inverse, lc, scratch);
}
bool launch_ks_mac_u64_small(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    return launch_ks_mac_t<ArithU64>(log_n, a, kp, blocks, lc);
}
bool launch_tensor_u64_small(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    return launch_tensor_class<ArithU64>(log_n, stage, a, b, d, batch, lc);
}

void launch_ntt_generic(const NttArgs& a, unsigned log_n, bool inverse, size_t limb_polys, const LaunchCtx& lc) {
    hipLaunchKernelGGL(ntt_generic_kernel, dim3((unsigned)limb_polys), dim3(256), 0, lc.s, a, log_n, inverse ? 1 : 0);
}

// single passes of the two-pass form of a small N = 8192 launch under the integer policy (see launch_ntt_f64_pass14)
void launch_ntt_u64_pass13(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (which == 0) launch_pass<ArithU64, 13, 2, 11, 11, TROYN_SMALL_EB, true, true, false>(a, limb_polys, lc);
    else launch_pass<ArithU64, 13, 2, 11, 11, TROYN_SMALL_EB, false, false, true>(a, limb_polys, lc);
}

}  // namespace troyn
