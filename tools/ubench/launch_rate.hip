// launch_rate.hip -- how many kernel launches per second N host threads can issue on hipStreamPerThread (the reference's concurrency model:
// test/bench/he_operations.cu -c N).  Sets the ceiling of any single-object API: ops/s <= launches/s / launches per op.
//   hipcc --offload-arch=gfx950 -O2 -o launch_rate launch_rate.hip -lpthread ; ./launch_rate
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int* p) { if (p && threadIdx.x == 1024) *p = 1; }
int main() {
    hipFree(nullptr);
    for (int threads : {1, 2, 4, 8, 16, 32, 64}) {
        for (int grid : {1, 12}) {
            const int per = 20000 / threads + 2000;
            std::atomic<int> ready{0}; std::atomic<bool> go{false};
            auto body = [&] {
                for (int i = 0; i < 200; i++) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, hipStreamPerThread, (int*)nullptr);
                hipStreamSynchronize(hipStreamPerThread);
                ready++;
                while (!go.load()) std::this_thread::yield();
                for (int i = 0; i < per; i++) {
                    hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, hipStreamPerThread, (int*)nullptr);
                    if ((i & 7) == 7) hipStreamSynchronize(hipStreamPerThread);      // an op = 8 launches, then the caller consumes the result
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
            std::printf("threads %2d grid %2d: %8.0f launches/s  (%.2f us per launch per thread; 8 launches + 1 sync per op -> %.0f ops/s)\n", threads, grid,
                        threads * (double)per / dt, dt / per * 1e6, threads * (double)per / dt / 8);
        }
    }
    return 0;
}
