// kernel instantiations of the ArithU64 policy, N >= 16384 (ntt_launch.inl)
#define TROYN_NTT_PART 2
#include "ntt_launch.inl"

namespace troyn {

bool launch_ntt_u64_large(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, const LaunchCtx& lc, u64* scratch) {
    return launch_ntt_optimised<ArithU64>(log_n, a, limb_polys, inverse, lc, scratch);
}
bool launch_ks_mac_u64_large(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, const LaunchCtx& lc) {
    return launch_ks_mac_t<ArithU64>(log_n, a, kp, blocks, lc);
}
bool launch_tensor_u64_large(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, const LaunchCtx& lc) {
    return launch_tensor_class<ArithU64>(log_n, stage, a, b, d, batch, lc);
}

// single passes of the two-pass form of a small N = 16384 launch under the integer policy (see launch_ntt_f64_pass14)
void launch_ntt_u64_pass14(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (which == 0) launch_pass<ArithU64, 14, 2, 12, 12, TROYN_SMALL_EB, true, true, false>(a, limb_polys, lc);
    else launch_pass<ArithU64, 14, 2, 12, 12, TROYN_SMALL_EB, false, false, true>(a, limb_polys, lc);
}

// N = 32768: the two passes of launch_two_pass<A, 15, 12, 4> one at a time (troyn_mrr_small.hip runs the strided passes between them itself)
void launch_ntt_u64_pass15(int which, const NttArgs& a, size_t limb_polys, const LaunchCtx& lc) {
    if (which == 0) launch_pass<ArithU64, 15, 3, 12, 12, 4, true, true, false>(a, limb_polys, lc);
    else launch_pass<ArithU64, 15, 3, 12, 12, 4, false, false, true>(a, limb_polys, lc);
}

}  // namespace troyn
