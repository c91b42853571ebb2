// kernel instantiations of the second-generation BEHZ conversions (behz2_kernels.hpp): one per base size and modulus class
#include "launch.hpp"
#include "behz2_lift_pass1.hpp"

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

// lift + first forward pass of both bases in one launch (behz2_lift_pass1.hpp): N = 32768, FP64 policy
bool launch_behz2_lift_pass1(unsigned L, size_t items, hipStream_t s, const Behz2Dev& c, const u64* src, u64* dst_q, u64* dst_bsk,
                             const double* tw_q, const double* tw_aux, const DevModulus* q_mods, const DevModulus* aux_mods) {
    const unsigned rows = L + c.NB + 1;
    if (rows > BEHZ2_FUSED_MAX_ROWS) return false;
    LiftPass1Args a{src, dst_q, dst_bsk, tw_q, tw_aux, q_mods, aux_mods};
    const dim3 grid((unsigned)(items * (4096u / (BEHZ2_FUSED_THREADS / 8)))), block(BEHZ2_FUSED_THREADS);
    const size_t lds = (size_t)(L > c.NB + 1 ? L : c.NB + 1) * BEHZ2_FUSED_THREADS * sizeof(u64);      // one base at a time
    switch (L) {
#define X(N) case N: hipLaunchKernelGGL((behz2_lift_pass1_kernel<N>), grid, block, lds, s, c, a); return true;
        BEHZ2_CASES(X)
#undef X
        default: return false;
    }
}

// last inverse pass of both bases + floor in one launch (behz2_lift_pass1.hpp): N = 32768, FP64 policy
bool launch_behz2_floor_pass2(unsigned L, size_t items, hipStream_t s, const Behz2Dev& c, const u64* in_q, const u64* in_bsk, u64* out,
                              const double* tw_q, const double* tw_aux, const DevModulus* q_mods, const DevModulus* aux_mods) {
    const unsigned rows = L + c.NB + 1;
    if (rows > BEHZ2_FUSED_MAX_ROWS) return false;
    FloorPass2Args a{in_q, in_bsk, out, tw_q, tw_aux, q_mods, aux_mods};
    const dim3 grid((unsigned)(items * (4096u / (BEHZ2_FUSED_THREADS / 8)))), block(BEHZ2_FUSED_THREADS);
    const size_t lds = (size_t)rows * BEHZ2_FUSED_THREADS * sizeof(u64);
    switch (L) {
#define X(N) case N: hipLaunchKernelGGL((behz2_floor_pass2_kernel<N>), grid, block, lds, s, c, a); return true;
        BEHZ2_CASES(X)
#undef X
        default: return false;
    }
}

}  // namespace troyn
