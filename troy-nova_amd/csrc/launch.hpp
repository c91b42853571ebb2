// launch.hpp -- entry points of the translation units that hold the kernel instantiations (one per kernel family / arithmetic
// policy, so that they compile in parallel).  troyn.hip (the C-ABI) only calls these.
#pragma once
#include <hip/hip_runtime.h>
#include "ntt_kernels.hpp"
#include "ksmac_kernels.hpp"
#include "behz2_kernels.hpp"

namespace troyn {

// optimised transforms of one arithmetic class (false: no kernel for this size -> ntt_generic)
bool launch_ntt_f64(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, hipStream_t s, u64* scratch);
bool launch_ntt_u64(unsigned log_n, const NttArgs& a, size_t limb_polys, bool inverse, hipStream_t s, u64* scratch);
void launch_ntt_generic(const NttArgs& a, unsigned log_n, bool inverse, size_t limb_polys, hipStream_t s);
// first-generation fused key-switch inner product (ks_mac_kernel)
bool launch_ks_mac_f64(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, hipStream_t s);
bool launch_ks_mac_u64(unsigned log_n, const NttArgs& a, const KeyPtrs& kp, size_t blocks, hipStream_t s);
// tensor product fused with the transforms (tensor_core_kernel); stage 0 / 2: the strided passes of the two-pass sizes
bool launch_tensor_f64(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, hipStream_t s);
bool launch_tensor_u64(unsigned log_n, int stage, const NttArgs& a, const NttArgs& b, const NttArgs& d, size_t batch, hipStream_t s);
// second-generation key-switch inner product (ksmac2_kernel, log_n = 13 / 14 / 15) and its key preparation
void launch_ksmac2(unsigned log_n, size_t batch, unsigned rows, const KsMacArgs& a, hipStream_t s);
void launch_ksmac_prepare_keys(const KeyPtrs& kp, unsigned L, unsigned polys, unsigned n, double* out, unsigned blocks, hipStream_t s);
// second-generation BEHZ conversions (L = 1 .. 16)
void launch_behz2_lift(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* src, u64* dst);
void launch_behz2_floor(unsigned L, bool smallq, unsigned grid, hipStream_t s, unsigned chunks, const Behz2Dev& c, const u64* in_q, const u64* in_bsk, u64* out);

}  // namespace troyn
