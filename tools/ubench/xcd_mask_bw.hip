// xcd_mask_bw.hip -- what a stream restricted to k of the 8 XCDs (hipExtStreamCreateWithCUMask) gets: which XCDs its workgroups land on and how much
// HBM bandwidth a streaming copy reaches there.  Question behind it: can the memory-bound kernels of the chain run on a few XCDs while the FP64-bound
// inner product keeps the others (and their L2s) to itself?
//   hipcc --offload-arch=gfx950 -O2 -o xcd_mask_bw xcd_mask_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
__global__ void census(unsigned* xcc_hist, unsigned* cu_hist) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));     // HW_REG_XCC_ID[3:0]
        atomicAdd(&xcc_hist[xcc & 15u], 1u);
    }
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / 16;
    uint4 *in, *out; unsigned* hist;
    CHECK(hipMalloc(&in, bytes)); CHECK(hipMalloc(&out, bytes)); CHECK(hipMalloc(&hist, 64 * 4));
    CHECK(hipMemset(in, 1, bytes));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("CUs %d\n", cus);
    // candidate bit layouts: (a) bit i = CU i with XCD = i % 8 (interleaved), (b) XCD = i / 32 (blocked)
    for (int layout = 0; layout < 2; layout++) {
        for (int k : {1, 2, 3, 4, 6, 8}) {
            std::vector<uint32_t> mask((cus + 31) / 32, 0);
            for (int i = 0; i < cus; i++) { const int xcd = layout == 0 ? i % 8 : i / (cus / 8); if (xcd < k) mask[i / 32] |= 1u << (i % 32); }
            hipStream_t s;
            if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("mask stream creation failed\n"); return 1; }
            CHECK(hipMemsetAsync(hist, 0, 64 * 4, s));
            hipLaunchKernelGGL(census, dim3(4096), dim3(64), 0, s, hist, hist + 16);
            unsigned h[16]; CHECK(hipMemcpyAsync(h, hist, 64, hipMemcpyDeviceToHost, s)); CHECK(hipStreamSynchronize(s));
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int w = 0; w < 3; w++) hipLaunchKernelGGL(copy_kernel, dim3(cus * 8), dim3(256), 0, s, in, out, n);
            CHECK(hipEventRecord(e0, s));
            const int reps = 5;
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(copy_kernel, dim3(cus * 8), dim3(256), 0, s, in, out, n);
            CHECK(hipEventRecord(e1, s)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("layout %s, %d XCD(s) in the mask: workgroups per XCC id [", layout == 0 ? "i%8" : "i/32", k);
            for (int x = 0; x < 8; x++) printf("%u ", h[x]);
            printf("]  copy %.2f TB/s (read + write)\n", 2.0 * bytes * reps / (ms * 1e-3) / 1e12);
            CHECK(hipStreamDestroy(s));
        }
    }
    return 0;
}
