"""Fused CKKS multiply -> relinearize -> rescale vs the three separate calls at N = 32768 and N = 8192 (six 50-bit primes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
pkg = entry.load_package()
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev).manual_seed(1)
def res(shape, q, n):
    out = torch.empty(tuple(shape) + (len(q), n), dtype=torch.int64, device=dev)
    for l, m in enumerate(q):
        out[..., l, :] = torch.randint(0, m, tuple(shape) + (n,), dtype=torch.int64, device=dev, generator=gen)
    return out
for n, logn, bits, B in ((32768, 15, [50] * 6, 256), (8192, 13, [50] * 6, 2048)):
    q = pkg.capi.coeff_modulus_create(n, bits)
    L = len(q) - 1
    plan = pkg.Plan(dev, logn, q)
    a, b = res((B, 2), q[:L], n), res((B, 2), q[:L], n)
    keys = [res((2,), q, n) for _ in range(L)]
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=dev)
    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=dev)
    relin = torch.empty((B, 2, L, n), dtype=torch.int64, device=dev)
    def three():
        plan.dyadic_convolute(a, 2, b, 2, L, out=prod)
        plan.relinearize(L, prod, keys, out=relin, is_ckks=True, is_ntt_form=True)
        plan.divide_and_round_q_last_ntt(L, relin, 2, out=out)
    def fused():
        plan.ckks_multiply_relinearize_rescale(L, a, b, keys, out=out)
    for name, fn in (("three calls", three), ("fused", fused)):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
        print("N=%d L=%d batch %d %-12s %9.1f ops/s" % (n, L, B, name, B / t))
