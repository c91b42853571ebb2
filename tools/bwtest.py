import torch, time
x = torch.empty(1<<28, dtype=torch.int64, device='cuda')  # 2 GiB
y = torch.empty_like(x)
for n in (1<<24, 1<<26, 1<<28):
    a, b = x[:n], y[:n]
    b.copy_(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): b.copy_(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(n * 8 / 2**20, "MiB copy:", round(2 * n * 8 / dt / 1e12, 2), "TB/s (read+write)")
    t0 = time.perf_counter()
    for _ in range(10): s = a.sum()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("   read-only sum:", round(n * 8 / dt / 1e12, 2), "TB/s")
