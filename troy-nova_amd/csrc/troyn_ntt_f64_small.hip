// kernel instantiations of the ArithF64 policy, N <= 8192 (ntt_launch.inl)
#define TROYN_NTT_PART 1
#include "ntt_launch.inl"

namespace troyn {

bool launch_ntt_f64_small(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, const LaunchCtx& lc, u64* scratch) {
    return launch_ntt_optimised<ArithF64>(log_n, a, limb_polys, inverse, lc, scratch);
}
bool launch_ks_mac_f64_small(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    return launch_ks_mac_t<ArithF64>(log_n, a, kp, blocks, lc);
}
bool launch_tensor_f64_small(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    return launch_tensor_class<ArithF64>(log_n, stage, a, b, d, batch, lc);
}

// single passes of the two-pass form of a small N = 8192 launch (see launch_ntt_f64_pass14)
void launch_ntt_f64_pass13(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (which == 0) launch_pass<ArithF64, 13, 2, 11, 11, TROYN_SMALL_EB, true, true, false>(a, limb_polys, lc);
    else launch_pass<ArithF64, 13, 2, 11, 11, TROYN_SMALL_EB, false, false, true>(a, limb_polys, lc);
}

}  // namespace troyn
