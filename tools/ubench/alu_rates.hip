// Microbenchmark: issue rate of the integer / fp64 VALU ops that bound the 64-bit modular
// arithmetic on gfx950.  Each kernel runs a long dependent-free stream of one opcode from
// many waves; reports cycles per wave-instruction per SIMD (2.4 GHz assumed -> measured via
// wall time and the achieved ops/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

#define DEFINE_KERNEL(NAME, DECL, BODY)                                                   \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned a0, unsigned b0) { \
        DECL;                                                                             \
        for (int it = 0; it < ITERS; ++it) {                                              \
            _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) { BODY; }                  \
        }                                                                                 \
        unsigned acc = 0;                                                                 \
        _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) acc ^= (unsigned)x[u];         \
        if (acc == 0x12345678u) out[threadIdx.x] = acc;                                   \
    }

DEFINE_KERNEL(k_mul_lo, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mul_hi, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mul_u24, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mul_hi_u24, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mad_u24, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_add, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_add3, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mad_u64, unsigned long long x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0" : "+v"(x[u]) : "v"(b0) : "s10", "s11"))
DEFINE_KERNEL(k_lshl_add_u64, unsigned long long x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(x[u])))
DEFINE_KERNEL(k_fma_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[u]) : "v"((double)b0)))
DEFINE_KERNEL(k_mul_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[u]) : "v"((double)b0)))
DEFINE_KERNEL(k_add_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[u]) : "v"((double)b0)))
DEFINE_KERNEL(k_rndne_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_rndne_f64 %0, %0" : "+v"(x[u])))
DEFINE_KERNEL(k_floor_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_floor_f64 %0, %0" : "+v"(x[u])))
DEFINE_KERNEL(k_cmp_f64, double x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (double)(a0 + u + threadIdx.x),
              asm volatile("v_cmp_lt_f64 vcc, %0, %1" : "+v"(x[u]) : "v"((double)b0) : "vcc"))
DEFINE_KERNEL(k_fma_f32, float x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = (float)(a0 + u + threadIdx.x),
              asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[u]) : "v"((float)b0)))
DEFINE_KERNEL(k_cndmask, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_sub_co, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(x[u]) : "v"(b0) : "vcc"))
DEFINE_KERNEL(k_cndmask_e64, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_cmp_cndmask, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_cmp_le_u32 vcc, %1, %0\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[u]) : "v"(b0) : "vcc"))
DEFINE_KERNEL(k_cmp_only, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_cmp_le_u32 vcc, %1, %0" : "+v"(x[u]) : "v"(b0) : "vcc"))
DEFINE_KERNEL(k_min_u32, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_subb_co, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x[u]) : "v"(b0) : "vcc"))
DEFINE_KERNEL(k_ashr_and, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_ashrrev_i32 %0, 31, %0\n\tv_and_b32 %0, %0, %1" : "+v"(x[u]) : "v"(b0)))
DEFINE_KERNEL(k_mov, unsigned x[UNROLL]; for (int u = 0; u < UNROLL; ++u) x[u] = a0 + u + threadIdx.x,
              asm volatile("v_mov_b32 %0, %1" : "+v"(x[u]) : "v"(b0)))

template <typename K>
void run(const char* name, K kernel, unsigned* d_out) {
    const int blocks = 256 * 8, threads = 256;   // 8 blocks/CU -> 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double wave_instrs = (double)blocks * (threads / 64) * ITERS * UNROLL;
    const double per_simd = wave_instrs / (256.0 * 4.0);
    const double ns_per = ms * 1e6 / per_simd;
    printf("%-16s %8.3f ms  %7.3f ns/wave-instr/SIMD  = %6.2f cycles @2.4GHz  (%.1f Gops/s lanes)\n", name, ms, ns_per, ns_per * 2.4,
           wave_instrs * 64 / (ms * 1e-3) / 1e9);
}

int main() {
    unsigned* d_out;
    CHECK(hipMalloc(&d_out, 4096));
    run("v_add_u32", k_add, d_out);
    run("v_add3_u32", k_add3, d_out);
    run("v_mov_b32", k_mov, d_out);
    run("v_cndmask_b32", k_cndmask, d_out);
    run("v_cndmask_e64", k_cndmask_e64, d_out);
    run("cmp+cndmask", k_cmp_cndmask, d_out);
    run("v_cmp_le_u32", k_cmp_only, d_out);
    run("v_min_u32", k_min_u32, d_out);
    run("v_subb_co_u32", k_subb_co, d_out);
    run("ashr+and", k_ashr_and, d_out);
    run("v_sub_co_u32", k_sub_co, d_out);
    run("v_lshl_add_u64", k_lshl_add_u64, d_out);
    run("v_mul_lo_u32", k_mul_lo, d_out);
    run("v_mul_hi_u32", k_mul_hi, d_out);
    run("v_mad_u64_u32", k_mad_u64, d_out);
    run("v_mul_u32_u24", k_mul_u24, d_out);
    run("v_mul_hi_u32_u24", k_mul_hi_u24, d_out);
    run("v_mad_u32_u24", k_mad_u24, d_out);
    run("v_fma_f32", k_fma_f32, d_out);
    run("v_fma_f64", k_fma_f64, d_out);
    run("v_mul_f64", k_mul_f64, d_out);
    run("v_add_f64", k_add_f64, d_out);
    run("v_rndne_f64", k_rndne_f64, d_out);
    run("v_floor_f64", k_floor_f64, d_out);
    run("v_cmp_lt_f64", k_cmp_f64, d_out);
    return 0;
}
