#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS-RLWE hot path.

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  CKKS, N = 16384, CoeffModulus::create(16384, {50}x6)  ->  K = 6 key limbs, L = 5 data limbs.
  One "op"   = multiply (dyadic 2x2 -> 3) + relinearize (key-switch core) + rescale_to_next
               on one pair of ciphertexts.
  One "step" = that pipeline over a batch of B independent ciphertext pairs per GPU, inputs and
               evaluation keys already resident in HBM.
  value = whole-job ops/s = n_gpus * B * steps / time (max over ranks);  "scaling": "weak".

--workload cfg4 (BASELINE.json configs[3]): 1024 independent BFV N = 32768, L = 10 (K = 11) ciphertext
  multiplications (BEHZ multiply + relinearize), block-partitioned over the ranks with shard.shard_range
  ("scaling": "strong": the job is fixed, each rank evaluates its slice in chunks of --batch).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).  Under torchrun the ranks
come from the environment; `python bench.py --gpus N` without WORLD_SIZE starts the N rank processes itself,
BEFORE this process touches the GPU, and relays rank 0's JSON line.  Evaluation keys are broadcast once from
rank 0 before the timed region; the data path has no collective.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK = 39.3e12       # FP64 vector lane-operations per second: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz (78.6 TFLOP/s, FMA = 2)
CHECK_ITEMS = lambda B: sorted({0, 1, 7 % B, B // 2, B - 1})   # the fused kernels permute items over workgroups / XCDs


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["cfg3", "cfg4"], default="cfg3")
    ap.add_argument("--batch", type=int, default=0, help="ciphertext pairs per GPU per launch (cfg3: 1024; cfg4: chunk of 64)")
    ap.add_argument("--total", type=int, default=1024, help="cfg4: size of the fixed job that is sharded over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the N=8192 / N=32768 side measurements (rank 0, 1 GPU)")
    ap.add_argument("--unfused", action="store_true", help="cfg3: three library calls per op instead of the fused entry point")
    ap.add_argument("--dry-shard", action="store_true", help="print the partition of the job over --gpus ranks and exit (no GPU)")
    ap.add_argument("--dry-run", action="store_true", help="run the launcher / rendezvous / key broadcast / timing reduction with the gloo backend on "
                                                              "CPU tensors and no evaluation (exercises the N > 1 plumbing where there is no GPU)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------
# self-launcher: `python bench.py --gpus N` outside torchrun
# ------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """start n fresh rank processes (this process has not initialised HIP) and relay rank 0's output"""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank(s) failed: %s\n" % bad)
        return 1
    try:
        line = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")][-1]
        if json.loads(line).get("n_gpus") != n:
            sys.stderr.write("bench.py: asked for %d GPUs, the job reports %s\n" % (n, json.loads(line).get("n_gpus")))
            return 3
    except (IndexError, ValueError):
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 4
    return 0


def shard_plan(total, world):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    import importlib
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    return [list(shard.shard_range(total, r, world)) for r in range(world)]


# ------------------------------------------------------------------------------------------------------
def uniform_residues(torch, shape_prefix, moduli, n, device, gen):
    """[*shape_prefix][len(moduli)][n] uniform residues below each limb's modulus (synthetic payload)"""
    out = torch.empty(tuple(shape_prefix) + (len(moduli), n), dtype=torch.int64, device=device)
    for l, q in enumerate(moduli):
        out[..., l, :] = torch.randint(0, q, tuple(shape_prefix) + (n,), dtype=torch.int64, device=device, generator=gen)
    return out


def timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def profile_record(name):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def run_cfg3(args, torch, pkg, shard, entry, rank, world, device):
    n, log_n = 16384, 14
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    K, L = 6, 5
    B = args.batch or 1024
    plan = pkg.Plan(device, log_n, q)
    gen = torch.Generator(device=device).manual_seed(0x123 + rank)
    a = uniform_residues(torch, (B, 2), q[:L], n, device, gen)
    b = uniform_residues(torch, (B, 2), q[:L], n, device, gen)
    # relinearization keys: L keys of [2][K][N] uniform residues (timing-identical to genuine keys);
    # generated on rank 0 and broadcast once over RCCL (the only exchange of the batched path)
    kgen = torch.Generator(device=device).manual_seed(0xC0FFEE)
    keys = [uniform_residues(torch, (2,), q, n, device, kgen) for _ in range(L)]
    shard.broadcast_tensors(keys, src=0)

    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=device)
    relin = torch.empty((B, 2, L, n), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=device)
    fused = hasattr(plan, "ckks_multiply_relinearize_rescale") and not args.unfused

    def step3():
        plan.dyadic_convolute(a, 2, b, 2, L, out=prod)                       # Evaluator::multiply (CKKS)
        plan.relinearize(L, prod, keys, out=relin, is_ckks=True, is_ntt_form=True)  # Evaluator::relinearize
        plan.divide_and_round_q_last_ntt(L, relin, 2, out=out)               # Evaluator::rescale_to_next

    def step1():
        plan.ckks_multiply_relinearize_rescale(L, a, b, keys, out=out)       # the same three calls behind one entry point

    step = step1 if fused else step3
    for _ in range(args.warmup):
        step()
    lib = plan.lib
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 1))
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t0
    import ctypes as C
    ks_ms, ks_n = C.c_double(0.0), C.c_uint64(0)
    pkg.capi.check(lib.troyn_kernel_timer_read(0, C.byref(ks_ms), C.byref(ks_n)))
    pkg.capi.check(lib.troyn_kernel_timer_enable(0, 0))
    elapsed = shard.max_over_ranks(elapsed, device=device)
    value = world * B * args.steps / elapsed

    # ---- roofline of the kernel with the largest share of the timed step: the fused key-switch inner product ----
    # algorithmic bytes per launch (SURVEY.md 8d minimum traffic): per item the L coefficient-form digits and the
    # L NTT-form input limbs are read once (2 * 8*N*L), the two output polynomials of L+1 rows written once (16*N*(L+1)); the key set
    # (16*N*L*(L+1) bytes) is shared by the whole batch and counted once per launch.
    ks_launch_ms = ks_ms.value / max(1, ks_n.value)
    alg_bytes = B * (2 * 8.0 * n * L + 16.0 * n * (L + 1)) + 16.0 * n * L * (L + 1)
    achieved = alg_bytes / (ks_launch_ms * 1e-3) / 1e9 if ks_n.value else 0.0
    prof = profile_record("r02_ksmac_counters.json")      # written by tools/collect_traffic.py from the rocprofv3 PMC passes
    traffic = None
    valu = None
    if prof and prof.get("batch") == B:
        traffic = prof.get("traffic_bytes_per_launch")
        if prof.get("valu_lane_ops_per_launch"):
            rate = prof["valu_lane_ops_per_launch"] / (ks_launch_ms * 1e-3)
            valu = {"what": "the kernel is FP64-VALU-bound, not HBM-bound: exact 50-bit modular arithmetic on the FP64 vector ALU "
                            "(no MFMA); lane-operations per launch from SQ_INSTS_VALU x 64 (profiles/r02_ksmac_counters.json)",
                    "lane_ops_per_launch": prof["valu_lane_ops_per_launch"], "achieved_lane_ops_per_s": round(rate, 0),
                    "peak_lane_ops_per_s": FP64_VALU_PEAK, "frac": round(rate / FP64_VALU_PEAK, 4),
                    "simd_valu_busy_profiled": prof.get("simd_valu_busy")}
    roofline = {"bound": "hbm", "kernel": "ksmac2_kernel<14> (fused key-switch inner product: digit NTTs + <digit, key> accumulation), "
                                          "largest share of the timed step (%.0f %%)" % (100.0 * ks_ms.value / (elapsed * 1e3)),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "launch_ms": round(ks_launch_ms, 4), "launches_timed": int(ks_n.value),
                "algorithmic_bytes_per_launch": alg_bytes, "valu_fp64": valu,
                # whole pipeline against the chip's HBM peak, both key accountings of the verdict (keys per op / keys once per batch)
                "pipeline": {"bytes_per_op_keys_per_op": 18.0e6, "frac_keys_per_op": round(value / world * 18.0e6 / (HBM_PEAK_GBS * 1e9), 4),
                             "bytes_per_op_keys_amortised": 10.2e6, "frac_keys_amortised": round(value / world * 10.2e6 / (HBM_PEAK_GBS * 1e9), 4)}}

    # secondary: the plain forward NTT of the key-switch digit shape (round-1's probe), not part of the timed region
    if rank == 0 and world == 1:
        RB = 256
        digits = uniform_residues(torch, (RB, L + 1), q[:L], n, device, gen)
        t_ntt = timed(torch, lambda: plan.ntt(digits, L + 1, L, mode=pkg.IDX_KS_SET_PRODUCTS, decomp=L, table_count=K), 10)
        nb = 16.0 * n * RB * (L + 1) * L
        roofline["ntt_probe"] = {"kernel": "ntt_pass_kernel<ArithF64,14,fwd>, %d limb-polynomials" % (RB * (L + 1) * L), "launch_ms": round(t_ntt * 1e3, 4),
                                 "achieved": round(nb / t_ntt / 1e9, 1), "frac": round(nb / t_ntt / 1e9 / HBM_PEAK_GBS, 4)}
        del digits
        # what the memory system delivers on this box (not part of the timed region): a 1 GiB device-to-device copy and a read-only pass
        probe = torch.empty(1 << 27, dtype=torch.int64, device=device)
        sink = torch.empty_like(probe)
        t_copy = timed(torch, lambda: sink.copy_(probe), 5)
        t_read = timed(torch, lambda: probe.sum(), 5)
        roofline["memory_system"] = {"what": "measured on this box: 1 GiB copy (read + write) and read-only reduction; `frac` above stays against the 8 TB/s peak",
                                     "copy_GBps": round(2.0 * probe.numel() * 8 / t_copy / 1e9, 1), "read_GBps": round(probe.numel() * 8 / t_read / 1e9, 1)}
        del probe, sink
        torch.cuda.empty_cache()

    result = {
        "metric": "homomorphic mul+relinearize ops/sec (CKKS mul+relin+rescale), N=16384",
        "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64 (q < 2^50: exact FP64-carried butterflies, canonical u64 residues at every call boundary)", "data": "synthetic",
        "config": {"workload": "CKKS N=16384, 6x50-bit coeff modulus (K=6, L=5): multiply + relinearize + rescale_to_next, "
                               "batch of %d independent ciphertext pairs per GPU" % B,
                   "batch_per_gpu": B, "entry": "troyn_ckks_multiply_relinearize_rescale" if fused else "troyn_dyadic_convolute + troyn_relinearize + troyn_divide_and_round_q_last_ntt",
                   "parallelism": "batch-sharded x%d, keys broadcast once (RCCL)" % world},
        "roofline": roofline,
    }

    # ---- in-run parity + CPU baseline (rank 0, N = 1 only): the oracle is the checker / reported baseline ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        O = entry.load_oracle()
        ctx = O.Context("ckks", n, q)
        hk = [pkg.to_host(k) for k in keys]

        def one_op(c, ha, hb):
            e = c.ckks_multiply(L, ha, hb)
            e = c.relinearize(L, True, e, hk)
            return c.mod_switch_scale_to_next(L, e)

        items = CHECK_ITEMS(B)
        ops, t_cpu0 = 0, time.perf_counter()
        for i in items:
            exp = one_op(ctx, pkg.to_host(a[i]), pkg.to_host(b[i]))
            ops += 1
            if not np.array_equal(pkg.to_host(out[i]), exp):
                raise AssertionError("bench: GPU result of item %d differs from the CPU oracle" % i)
        ha, hb = pkg.to_host(a[0]), pkg.to_host(b[0])
        # (1) one thread: the reference's host path is strictly single-threaded per call
        while time.perf_counter() - t_cpu0 < args.cpu_seconds / 2:
            one_op(ctx, ha, hb)
            ops += 1
        t_cpu = time.perf_counter() - t_cpu0
        # (2) every host core evaluating independent ciphertexts (the reference's `troybench -H -c N` mode); the
        #     oracle calls release the GIL, each thread owns its context
        import threading
        nthreads = max(1, min(os.cpu_count() or 1, 128))
        counts = [0] * nthreads
        deadline = time.perf_counter() + args.cpu_seconds / 2

        def worker(i):
            c = O.Context("ckks", n, q)
            while True:
                one_op(c, ha, hb)
                counts[i] += 1
                if time.perf_counter() >= deadline:
                    break

        t_mt0 = time.perf_counter()
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        t_mt = time.perf_counter() - t_mt0
        result["cpu_baseline"] = {"value": round(ops / t_cpu, 3), "unit": "ops/s", "cores": 1, "kind": "port",
                                  "sample": "%d sequential mul+relin+rescale ops (items %s of the same workload, then item 0 repeated; %.1f s, "
                                            "oracle/troy_oracle.c, gcc -O3, 1 thread of %d host cores)" % (ops, items, t_cpu, os.cpu_count()),
                                  "all_cores": {"value": round(sum(counts) / t_mt, 2), "unit": "ops/s", "cores": nthreads,
                                                "sample": "%d ops by %d threads on independent copies in %.1f s" % (sum(counts), nthreads, t_mt)}}
        result["parity"] = "bit-exact vs CPU oracle (items %s of %d)" % (items, B)

    # ---- north_star's other sizes, same run (rank 0, 1 GPU): N = 8192 and N = 32768 -------------------------------
    if rank == 0 and world == 1 and not args.no_extra:
        del a, b, prod, relin, out
        torch.cuda.empty_cache()
        result["other_configs"] = extra_configs(torch, pkg, device)
    return result


def extra_configs(torch, pkg, device):
    """NTT + dyadic + INTT and relinearize throughput at N = 8192 (cfg2 shape) and N = 32768 (cfg4 shape); synthetic residues"""
    res = {}
    gen = torch.Generator(device=device).manual_seed(7)
    # the headline operation (CKKS multiply + relinearize + rescale, six 50-bit primes) at north_star's other two ring sizes
    for n, log_n, Bn in ((8192, 13, 2048), (32768, 15, 256)):
        q = pkg.capi.coeff_modulus_create(n, [50] * 6)
        L = 5
        plan = pkg.Plan(device, log_n, q)
        x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
        keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
        out = torch.empty((Bn, 2, L - 1, n), dtype=torch.int64, device=device)
        t = timed(torch, lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), 10)
        res["ckks_mul_relin_rescale_N%d" % n] = {"what": "CKKS N=%d, 6x50-bit (K=6, L=5): multiply + relinearize + rescale_to_next (fused entry)" % n,
                                                  "batch": Bn, "ops_per_s": round(Bn / t, 1)}
        del x, y, keys, out, plan
        torch.cuda.empty_cache()
    # N = 8192, 3 x 40-bit (L = 2 data limbs): BASELINE configs[1]
    n, Bn = 8192, 2048
    q = pkg.capi.coeff_modulus_create(n, [40, 40, 40])
    L = 2
    plan = pkg.Plan(device, 13, q)
    x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
    xn, yn = torch.empty_like(x), torch.empty_like(y)
    prod = torch.empty((Bn, 3, L, n), dtype=torch.int64, device=device)
    keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
    out2 = torch.empty((Bn, 2, L, n), dtype=torch.int64, device=device)

    def pipe():
        plan.ntt(x, 2, L, out=xn); plan.ntt(y, 2, L, out=yn)
        plan.dyadic_convolute(xn, 2, yn, 2, L, out=prod)
        plan.ntt(prod, 3, L, inverse=True)
    t = timed(torch, pipe, 10)
    alg = 168.0 * n * L * Bn                                # 4 NTT (16 B/coeff) + dyadic (56) + 3 INTT (16), per limb coefficient
    tr = timed(torch, lambda: plan.relinearize(L, prod, keys, out=out2, is_ckks=False, is_ntt_form=False), 10)
    behz = pkg.Behz(plan, L, 1032193)
    tm = timed(torch, lambda: behz.multiply(x, 2, y, 2, out=prod), 10)
    res["N8192_L2"] = {"what": "BFV N=8192, 3x40-bit (L=2): NTT(a), NTT(b) + dyadic 2x2->3 + INTT(3) as separate calls; relinearize (coefficient form); BEHZ multiply", "batch": Bn,
                       "ntt_dyadic_intt_ciphertexts_per_s": round(Bn / t, 1), "ntt_dyadic_intt_hbm_frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4),
                       "relinearize_ops_per_s": round(Bn / tr, 1), "multiply_ops_per_s": round(Bn / tm, 1)}
    del x, y, xn, yn, prod, keys, out2, plan, behz
    torch.cuda.empty_cache()
    # N = 16384, 6 x 50-bit (L = 5): the headline's ring size, the same two pipelines as separate calls
    n, Bn = 16384, 512
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    L = 5
    plan = pkg.Plan(device, 14, q)
    x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
    xn, yn = torch.empty_like(x), torch.empty_like(y)
    prod = torch.empty((Bn, 3, L, n), dtype=torch.int64, device=device)
    keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
    out2 = torch.empty((Bn, 2, L, n), dtype=torch.int64, device=device)
    t = timed(torch, lambda: (plan.ntt(x, 2, L, out=xn), plan.ntt(y, 2, L, out=yn), plan.dyadic_convolute(xn, 2, yn, 2, L, out=prod),
                              plan.ntt(prod, 3, L, inverse=True)), 10)
    alg = 168.0 * n * L * Bn
    tr = timed(torch, lambda: plan.relinearize(L, prod, keys, out=out2, is_ckks=True, is_ntt_form=True), 10)
    res["N16384_L5"] = {"what": "N=16384, 6x50-bit (L=5): NTT(a), NTT(b) + dyadic 2x2->3 + INTT(3) as separate calls; relinearize (NTT form)", "batch": Bn,
                        "ntt_dyadic_intt_ciphertexts_per_s": round(Bn / t, 1), "ntt_dyadic_intt_hbm_frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4),
                        "relinearize_ops_per_s": round(Bn / tr, 1)}
    del x, y, xn, yn, prod, keys, out2, plan
    torch.cuda.empty_cache()
    # N = 32768, 11 x 50-bit (L = 10): BASELINE configs[3]
    n, Bn = 32768, 64
    q = pkg.capi.coeff_modulus_create(n, [50] * 11)
    L = 10
    plan = pkg.Plan(device, 15, q)
    behz = pkg.Behz(plan, L, 1032193)
    x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
    xn, yn = torch.empty_like(x), torch.empty_like(y)
    prod = torch.empty((Bn, 3, L, n), dtype=torch.int64, device=device)
    keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
    out2 = torch.empty((Bn, 2, L, n), dtype=torch.int64, device=device)
    t = timed(torch, lambda: (plan.ntt(x, 2, L, out=xn), plan.ntt(y, 2, L, out=yn), plan.dyadic_convolute(xn, 2, yn, 2, L, out=prod),
                              plan.ntt(prod, 3, L, inverse=True)), 5)
    alg = (7 * 32.0 + 56.0) * n * L * Bn                    # two-pass transforms move every coefficient twice (32 B)
    tm = timed(torch, lambda: behz.multiply(x, 2, y, 2, out=prod), 5)
    tr = timed(torch, lambda: plan.relinearize(L, prod, keys, out=out2, is_ckks=False, is_ntt_form=False), 5)
    res["N32768_L10"] = {"what": "BFV N=32768, 11x50-bit (L=10): NTT+dyadic+INTT; BEHZ multiply; relinearize; multiply+relinearize", "batch": Bn,
                         "ntt_dyadic_intt_ciphertexts_per_s": round(Bn / t, 1), "ntt_dyadic_intt_hbm_frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4),
                         "multiply_ops_per_s": round(Bn / tm, 1), "relinearize_ops_per_s": round(Bn / tr, 1),
                         "multiply_relinearize_ops_per_s": round(Bn / (tm + tr), 1)}
    return res


def run_cfg4(args, torch, pkg, shard, entry, rank, world, device):
    """BASELINE configs[3]: a fixed job of --total BFV N=32768 L=10 multiply+relinearize ops, block-partitioned over the ranks"""
    n, log_n, t_plain = 32768, 15, 1032193
    q = pkg.capi.coeff_modulus_create(n, [50] * 11)
    K, L = 11, 10
    chunk = args.batch or 64
    lo, hi = shard.shard_range(args.total, rank, world)
    mine = hi - lo
    plan = pkg.Plan(device, log_n, q)
    behz = pkg.Behz(plan, L, t_plain)
    gen = torch.Generator(device=device).manual_seed(0x123)      # the same job on every world size: item i has the same payload
    kgen = torch.Generator(device=device).manual_seed(0xC0FFEE)
    keys = [uniform_residues(torch, (2,), q, n, device, kgen) for _ in range(L)]
    shard.broadcast_tensors(keys, src=0)
    nb = min(chunk, max(mine, 1))
    x = uniform_residues(torch, (nb, 2), q[:L], n, device, gen)
    y = uniform_residues(torch, (nb, 2), q[:L], n, device, gen)
    prod = torch.empty((nb, 3, L, n), dtype=torch.int64, device=device)
    out = torch.empty((nb, 2, L, n), dtype=torch.int64, device=device)

    def step():
        done = 0
        while done < mine:      # the rank's slice, `chunk` ciphertext pairs per launch (synthetic operands reused per chunk)
            c = min(nb, mine - done)
            behz.multiply(x[:c], 2, y[:c], 2, out=prod[:c])                                         # Evaluator::multiply (BFV, BEHZ)
            plan.relinearize(L, prod[:c], keys, out=out[:c], is_ckks=False, is_ntt_form=False)      # Evaluator::relinearize
            done += c

    for _ in range(args.warmup):
        step()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, device=device)
    value = args.total * args.steps / elapsed
    return {
        "metric": "homomorphic mul+relinearize ops/sec (BFV BEHZ multiply + relinearize), N=32768", "value": round(value, 1), "unit": "ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64 (q < 2^50: exact FP64-carried butterflies; 61-bit BEHZ base: integer butterflies)", "data": "synthetic",
        "config": {"workload": "BFV N=32768, 11x50-bit coeff modulus (K=11, L=10), t=1032193: %d independent multiply + relinearize ops per step, "
                               "block-partitioned over %d rank(s) (rank 0: items [%d, %d)), %d per launch" % (args.total, world, lo, hi, nb),
                   "total_ops_per_step": args.total, "parallelism": "batch-sharded x%d (shard.shard_range), keys broadcast once (RCCL)" % world},
        "roofline": None,
    }


def dry_run(args, rank, world):
    import torch
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    keys = [torch.full((2, 3, 8), 100 + j if rank == 0 else -1, dtype=torch.int64) for j in range(2)]
    shard.broadcast_tensors(keys, src=0)
    ok = all(int(k[0, 0, 0]) == 100 + j for j, k in enumerate(keys))
    lo, hi = shard.shard_range(args.total, rank, world)
    shard.barrier()
    elapsed = shard.max_over_ranks(0.001 * (rank + 1))
    covered = shard.sum_over_ranks(hi - lo)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "keys_broadcast_ok": ok, "max_elapsed": elapsed, "items_covered": covered,
                          "rank0_range": [lo, hi], "scaling": "strong" if args.workload == "cfg4" else "weak"}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if ok else 5


def main():
    args = parse_args()
    if args.dry_shard:
        print(json.dumps({"total": args.total, "world": args.gpus, "ranges": shard_plan(args.total, args.gpus)}))
        return 0
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("TROYN_BENCH_SPAWN")):
        return launch_ranks(args.gpus)         # nothing above touched the GPU (TROYN_BENCH_SPAWN: take this path at N = 1 too)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    if args.workload == "cfg4" and args.steps == 200:
        args.steps = 5                          # one step = the whole 1024-op job
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import __graft_entry__ as entry
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X")
    torch.cuda.set_device(local_rank)             # before the process group: RCCL binds its communicator to the current device
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    pkg = entry.load_package()
    import importlib
    shard = importlib.import_module("troy_nova_amd.shard")
    run = run_cfg3 if args.workload == "cfg3" else run_cfg4
    result = run(args, torch, pkg, shard, entry, rank, world, device)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
