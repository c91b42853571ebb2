// stream_pool.hip -- T host threads of "single-object ops" (7 dependent ~6 us kernels of 40 workgroups + one stream wait per op) on a POOL of Q explicit
// blocking streams (thread t -> stream (base + t * stride) % Q): how does ops/s depend on Q and on which streams the active threads hold?
// (the mirror's TROY_STREAMS mapping, troy/troy.cpp current_stream()).   hipcc --offload-arch=gfx950 -O2 -o stream_pool stream_pool.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) { }
    if (sink && threadIdx.x == 4096) *sink = 1;
}
static double run(const std::vector<hipStream_t>& pool, int threads, int base, int stride, int kernels_per_op, unsigned long long ticks) {
    const int per = 300;
    std::atomic<int> ready{0}; std::atomic<bool> go{false};
    auto body = [&](int t) {
        hipStream_t s = pool[(size_t)(base + t * stride) % pool.size()];
        auto op = [&] { for (int k = 0; k < kernels_per_op; k++) hipLaunchKernelGGL(spin, dim3(40), dim3(256), 0, s, ticks, (int*)nullptr); hipStreamSynchronize(s); };
        for (int i = 0; i < 30; i++) op();
        ready++;
        while (!go.load()) std::this_thread::yield();
        for (int i = 0; i < per; i++) op();
    };
    std::vector<std::thread> th;
    for (int t = 0; t < threads; t++) th.emplace_back(body, t);
    while (ready.load() < threads) std::this_thread::yield();
    auto t0 = std::chrono::steady_clock::now();
    go = true;
    for (auto& x : th) x.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return threads * (double)per / dt;
}
int main(int argc, char** argv) {
    hipFree(nullptr);
    const unsigned long long ticks = 600;     // s_memtime ticks at 100 MHz: ~6 us
    const int kpo = 7;
    for (int q : {1, 2, 3, 4, 5, 6, 8, 12, 16, 32}) {
        std::vector<hipStream_t> pool(q);
        for (auto& s : pool) hipStreamCreate(&s);
        std::printf("Q=%2d:", q);
        for (int threads : {1, 4, 16, 64}) std::printf("  T%-2d %7.0f", threads, run(pool, threads, 0, 1, kpo, ticks));
        // 4 threads on streams base..base+3 for every base: which quadruples share hardware queues?
        if (q >= 4) { std::printf("  | T4 by base:"); for (int b = 0; b < q && b < 8; b++) std::printf(" %6.0f", run(pool, 4, b, 1, kpo, ticks)); }
        std::printf("\n");
        for (auto& s : pool) hipStreamDestroy(s);
    }
    // non-blocking streams, same question
    for (int q : {4, 8, 16}) {
        std::vector<hipStream_t> pool(q);
        for (auto& s : pool) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        std::printf("Q=%2d non-blocking:", q);
        for (int threads : {1, 4, 16, 64}) std::printf("  T%-2d %7.0f", threads, run(pool, threads, 0, 1, kpo, ticks));
        std::printf("\n");
        for (auto& s : pool) hipStreamDestroy(s);
    }
    return 0;
}
