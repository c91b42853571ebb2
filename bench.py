#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS-RLWE hot path.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  CKKS, N = 16384, CoeffModulus::create(16384, {50}x6)  ->  K = 6 key limbs, L = 5 data limbs.
  One "op"   = multiply (dyadic 2x2 -> 3) + relinearize (key-switch core) + rescale_to_next
               on one pair of ciphertexts.
  One "step" = that pipeline over a batch of B independent ciphertext pairs per GPU, inputs and
               evaluation keys already resident in HBM.
value = whole-job ops/s = n_gpus * B * steps / time (max over ranks).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL); the batch is sharded
(weak scaling: B per GPU), evaluation keys are broadcast once from rank 0 over RCCL before the
timed region, and the data path has no collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def uniform_residues(shape_prefix, moduli, n, device, gen):
    """[*shape_prefix][len(moduli)][n] uniform residues below each limb's modulus (synthetic payload)"""
    out = torch.empty(tuple(shape_prefix) + (len(moduli), n), dtype=torch.int64, device=device)
    for l, q in enumerate(moduli):
        out[..., l, :] = torch.randint(0, q, tuple(shape_prefix) + (n,), dtype=torch.int64, device=device, generator=gen)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="ciphertext pairs per GPU per step (2.7 GB of operands; a larger batch only amortises launch gaps)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    # ---- workload ---------------------------------------------------------------------------
    n, log_n = 16384, 14
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    K, L = 6, 5
    B = args.batch
    plan = pkg.Plan(device, log_n, q)
    gen = torch.Generator(device=device).manual_seed(0x123 + rank)
    a = uniform_residues((B, 2), q[:L], n, device, gen)
    b = uniform_residues((B, 2), q[:L], n, device, gen)
    # relinearization keys: L keys of [2][K][N] uniform residues (timing-identical to genuine keys);
    # generated on rank 0 and broadcast once over RCCL (the only exchange of the batched path)
    kgen = torch.Generator(device=device).manual_seed(0xC0FFEE)
    keys = [uniform_residues((2,), q, n, device, kgen) for _ in range(L)]
    shard.broadcast_tensors(keys, src=0)

    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=device)
    relin = torch.empty((B, 2, L, n), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=device)

    def step():
        plan.dyadic_convolute(a, 2, b, 2, L, out=prod)                       # Evaluator::multiply (CKKS)
        plan.relinearize(L, prod, keys, out=relin, is_ckks=True, is_ntt_form=True)  # Evaluator::relinearize
        plan.divide_and_round_q_last_ntt(L, relin, 2, out=out)               # Evaluator::rescale_to_next

    for _ in range(args.warmup):
        step()
    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, device=device)
    value = world * B * args.steps / elapsed

    # ---- roofline of the dominant kernel: the forward NTT of the key-switch digits -------------------
    # one launch transforms RB*(L+1)*L limb-polynomials; algorithmic bytes = 16*N per limb-polynomial
    # (8 read + 8 written).  Timed with events on the stream the kernel is launched on.
    RB = 256    # the roofline launch is a fixed shape (the one profiles/r01_ntt_traffic.json was collected on)
    digits = uniform_residues((RB, L + 1), q[:L], n, device, gen)
    reps = max(5, args.steps)
    plan.ntt(digits, L + 1, L, mode=pkg.IDX_KS_SET_PRODUCTS, decomp=L, table_count=K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        plan.ntt(digits, L + 1, L, mode=pkg.IDX_KS_SET_PRODUCTS, decomp=L, table_count=K)
    e1.record()
    torch.cuda.synchronize()
    ntt_ms = e0.elapsed_time(e1) / reps
    limb_polys = RB * (L + 1) * L
    alg_bytes = 16.0 * n * limb_polys
    achieved = alg_bytes / (ntt_ms * 1e-3) / 1e9
    # HBM traffic of that launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
    # --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as the microarchitecture guide prescribes for gfx950)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_ntt_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("grid_x") == limb_polys * 1024:
            traffic = tj["traffic_bytes_per_launch"]
    roofline = {"bound": "hbm", "kernel": "ntt_pass_kernel<ArithF64,14,fwd> (forward NTT of the key-switch digits, %d limb-polys/launch)" % limb_polys,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launch_ms": round(ntt_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}

    result = {
        "metric": "homomorphic mul+relinearize ops/sec (CKKS mul+relin+rescale), N=16384",
        "value": round(value, 1), "unit": "ops/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "CKKS N=16384, 6x50-bit coeff modulus (K=6, L=5): multiply + relinearize + rescale_to_next, "
                               "batch of %d independent ciphertext pairs per GPU" % B,
                   "batch_per_gpu": B, "parallelism": "batch-sharded x%d, keys broadcast once (RCCL)" % world},
        "roofline": roofline,
    }

    # ---- in-run parity + CPU baseline (rank 0, N = 1 only): the oracle is the checker / reported baseline ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        O = entry.load_oracle()
        ctx = O.Context("ckks", n, q)
        hk = [pkg.to_host(k) for k in keys]
        ha, hb = pkg.to_host(a[:1]), pkg.to_host(b[:1])
        got = pkg.to_host(out[:1])
        def one_op(c):
            e = c.ckks_multiply(L, ha[0], hb[0])
            e = c.relinearize(L, True, e, hk)
            return c.mod_switch_scale_to_next(L, e)

        # (1) one thread: the reference's host path is strictly single-threaded per call
        ops, t_cpu0 = 0, time.perf_counter()
        exp = None
        while True:
            e = one_op(ctx)
            exp = e if exp is None else exp
            ops += 1
            if time.perf_counter() - t_cpu0 >= args.cpu_seconds / 2 and ops >= 3:
                break
        t_cpu = time.perf_counter() - t_cpu0
        if not np.array_equal(got[0], exp):
            raise AssertionError("bench: GPU result of item 0 differs from the CPU oracle")
        # (2) every host core evaluating independent ciphertexts (the reference's `troybench -H -c N` mode); the
        #     oracle calls release the GIL, each thread owns its context
        import threading
        nthreads = max(1, min(os.cpu_count() or 1, 128))
        counts = [0] * nthreads
        deadline = time.perf_counter() + args.cpu_seconds / 2

        def worker(i):
            c = O.Context("ckks", n, q)
            while True:
                one_op(c)
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
                                  "sample": "%d sequential mul+relin+rescale ops on item 0 of the same workload (%.1f s, "
                                            "oracle/troy_oracle.c, gcc -O3, 1 thread of %d host cores)" % (ops, t_cpu, os.cpu_count()),
                                  "all_cores": {"value": round(sum(counts) / t_mt, 2), "unit": "ops/s", "cores": nthreads,
                                                "sample": "%d ops by %d threads on independent copies in %.1f s" % (sum(counts), nthreads, t_mt)}}
        result["parity"] = "bit-exact vs CPU oracle (item 0)"

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
