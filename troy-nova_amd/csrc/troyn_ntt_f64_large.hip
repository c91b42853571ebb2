// kernel instantiations of the ArithF64 policy, N >= 16384 (ntt_launch.inl)
#define TROYN_NTT_PART 2
#include "ntt_launch.inl"

namespace troyn {

bool launch_ntt_f64_large(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, hipStream_t s, u64* scratch) {
    return launch_ntt_optimised<ArithF64>(log_n, a, limb_polys, inverse, s, scratch);
}
bool launch_ks_mac_f64_large(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, hipStream_t s) {
    return launch_ks_mac_t<ArithF64>(log_n, a, kp, blocks, s);
}
bool launch_tensor_f64_large(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, hipStream_t s) {
    return launch_tensor_class<ArithF64>(log_n, stage, a, b, d, batch, s);
}

}  // namespace troyn
