// Sustained FP64 vector rate under the instruction mix of the exact-FP64 butterflies (dev_math_f64.hpp): what the chip delivers over
// tens of milliseconds (power management included), as the ceiling ksmac2_kernel's FP64 issue figure should be read against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I troy-nova_amd/csrc -o tools/ubench/fp64_sustained tools/ubench/fp64_sustained.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "dev_math_f64.hpp"
using namespace troyn;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// 16 butterflies (8 FP64 instructions each) per iteration on 32 registers; twiddles in registers; no memory traffic in the loop
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void butterfly_stream(double* out, int iters, double p, double inv_p, double w0) {
    double x[32];
    for (int i = 0; i < 32; i++) x[i] = (double)((threadIdx.x * 32 + i) * 2654435761u % 1000003u);
    double w = w0 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int b = 0; b < 16; b++) {
            const double r = f64_mulq(x[b + 16], w, inv_p, p);
            const double u = x[b];
            x[b] = u + r; x[b + 16] = u - r;
        }
#pragma unroll
        for (int b = 0; b < 16; b++) {     // a second layer pairing neighbours, so values stay bounded by re-centring one of them
            const double r = f64_mulq(x[2 * b + 1], w, inv_p, p);
            const double u = x[2 * b];
            x[2 * b] = u + r; x[2 * b + 1] = f64_corr(u - r, F64Mod{p, inv_p});
        }
    }
    double acc = 0;
    for (int i = 0; i < 32; i++) acc += x[i];
    if (acc == 123.456) out[threadIdx.x] = acc;
}

// pure FMA stream (32 independent chains)
__global__ __launch_bounds__(256) void fma_stream(double* out, int iters, double a, double b) {
    double x[32];
    for (int i = 0; i < 32; i++) x[i] = (double)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; i++) x[i] = __builtin_fma(x[i], a, b);
    }
    double acc = 0;
    for (int i = 0; i < 32; i++) acc += x[i];
    if (acc == 123.456) out[threadIdx.x] = acc;
}

template <class F>
static void run(const char* name, F launch, double lane_ops) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.2f ms  %6.2f T lane-ops/s  (%.3f of 39.3 T)\n", name, ms, lane_ops / (ms * 1e-3) / 1e12, lane_ops / (ms * 1e-3) / 39.3216e12);
    }
}

int main() {
    double* d;
    CHECK(hipMalloc(&d, 4096));
    const double p = 1125899903107073.0, inv_p = 1.0 / p;
    const int iters = 6000;
    {
        const int blocks = 256 * 8;     // 8 workgroups of 4 waves per CU = 8 waves per SIMD
        const double ops = (double)blocks * 256 * iters * (16 * 8 + 16 * 11);
        run("butterfly mix, 8 waves/SIMD", [&] { hipLaunchKernelGGL(butterfly_stream<1>, dim3(blocks), dim3(256), 0, 0, d, iters, p, inv_p, 12345.0); }, ops);
    }
    {
        const int blocks = 256 * 2;     // 2 workgroups per CU = 2 waves per SIMD (ksmac2's occupancy)
        const double ops = (double)blocks * 256 * iters * 4 * (16 * 8 + 16 * 11);
        run("butterfly mix, 2 waves/SIMD", [&] { hipLaunchKernelGGL(butterfly_stream<2>, dim3(blocks), dim3(256), 0, 0, d, iters * 4, p, inv_p, 12345.0); }, ops);
    }
    {
        const int blocks = 256 * 8;
        const double ops = (double)blocks * 256 * (iters * 8) * 32;
        run("v_fma_f64 only, 8 waves/SIMD", [&] { hipLaunchKernelGGL(fma_stream, dim3(blocks), dim3(256), 0, 0, d, iters * 8, 1.0000001, 0.5); }, ops);
    }
    return 0;
}
