// stream_overlap.hip -- do small long-running kernels issued by N host threads on hipStreamPerThread overlap on the GPU?
// Each kernel: 12 workgroups x 256 threads spinning for ~50 us (the shape of a single-object key-switch inner product).
//   hipcc --offload-arch=gfx950 -O2 -o stream_overlap stream_overlap.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void spin(unsigned long long ticks, int* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < ticks) { }
    if (sink && threadIdx.x == 4096) *sink = 1;
}
int main() {
    hipFree(nullptr);
    const unsigned long long ticks = 100000;     // ~50 us at ~2 GHz (s_memtime ticks)
    for (int threads : {1, 2, 4, 8, 16, 32}) {
        const int per = 400;
        std::atomic<int> ready{0}; std::atomic<bool> go{false};
        auto body = [&] {
            for (int i = 0; i < 20; i++) hipLaunchKernelGGL(spin, dim3(12), dim3(256), 0, hipStreamPerThread, ticks, (int*)nullptr);
            hipStreamSynchronize(hipStreamPerThread);
            ready++;
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < per; i++) {
                hipLaunchKernelGGL(spin, dim3(12), dim3(256), 0, hipStreamPerThread, ticks, (int*)nullptr);
                if ((i & 3) == 3) hipStreamSynchronize(hipStreamPerThread);
            }
            hipStreamSynchronize(hipStreamPerThread);
        };
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++) th.emplace_back(body);
        while (ready.load() < threads) std::this_thread::yield();
        auto t0 = std::chrono::steady_clock::now();
        go = true;
        for (auto& x : th) x.join();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("threads %2d: %8.0f kernels/s, %.1f us per kernel per thread => %.1f kernels in flight on average\n", threads,
                    threads * (double)per / dt, dt / per * 1e6, threads * (double)per / dt * (dt / per) / 1.0 / threads * threads * 0 + (threads * (double)per / dt) / (1.0 / (dt / per)) );
    }
    return 0;
}
