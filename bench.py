#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS-RLWE hot path.

Default workload (BASELINE.json configs[2], the configuration the metric is quoted on):
  CKKS, N = 16384, CoeffModulus::create(16384, {50}x6)  ->  K = 6 key limbs, L = 5 data limbs.
  One "op"   = multiply (dyadic 2x2 -> 3) + relinearize (key-switch core) + rescale_to_next
               on one pair of ciphertexts.
  One "step" = --inner (default 12) passes of that pipeline over a batch of B = 1024 independent ciphertext pairs per GPU,
               inputs and evaluation keys already resident in HBM (12 x 1024 ops per step: 20 driver steps time about a second).
  value = whole-job ops/s = n_gpus * B * inner * steps / time (max over ranks);  "scaling": "weak".
  The same line carries the three-call figure (Evaluator::multiply + relinearize + rescale_to_next as three library calls, the
  reference's own operator boundary) next to the fused entry's, the roofline of the dominant kernel, the CPU baseline, and
  `other_configs`: BASELINE configs[3] (cfg4) and configs[4] (cfg5) each with value + roofline + cpu_baseline + parity, north_star's
  other ring sizes, and the C++ mirror (troy::Evaluator) timed by tests/cpp/he_bench_driver.

--workload cfg4 (BASELINE.json configs[3]): 1024 independent BFV N = 32768, L = 10 (K = 11) ciphertext
  multiplications (BEHZ multiply + relinearize), block-partitioned over the ranks with shard.shard_range
  ("scaling": "strong": the job is fixed, each rank evaluates its slice in chunks of --batch).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL over xGMI).  Under torchrun the ranks
come from the environment; `python bench.py --gpus N` without WORLD_SIZE starts the N rank processes itself,
BEFORE this process touches the GPU, relays rank 0's JSON line, and ends every rank as soon as one of them fails.
Evaluation keys are broadcast once from rank 0 before the timed region; the data path has no collective.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# FP64 vector rate: not a row of the guide; derived from its chip table (256 CUs x 4 SIMDs, 2.4 GHz) and the 16 FP64 lanes a SIMD
# retires per clock (a wave64 FP64 instruction occupies its SIMD for 4 cycles, tools/ubench/alu_rates.hip): 39.3 T lane-ops/s = 78.6 TFLOP/s
FP64_VALU_PEAK = 256 * 4 * 16 * 2.4e9
CHECK_ITEMS = lambda B: sorted({0, 1, 7 % B, B // 2, B - 1})   # the fused kernels permute items over workgroups / XCDs
TIMER_KS, TIMER_TENSOR, TIMER_PLAIN_MAC, TIMER_FLOOR = 0, 1, 2, 3   # include/troyn.h TROYN_TIMER_*


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inner", type=int, default=0, help="passes over the batch per step (cfg3: 12; cfg4: 1 = the whole job once)")
    ap.add_argument("--workload", choices=["cfg3", "cfg4"], default="cfg3")
    ap.add_argument("--batch", type=int, default=0, help="ciphertext pairs per GPU per launch (cfg3: 1024; cfg4: chunk of 64)")
    ap.add_argument("--total", type=int, default=1024, help="cfg4: size of the fixed job that is sharded over the ranks")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-clock budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip other_configs (rank 0, 1 GPU)")
    ap.add_argument("--unfused", action="store_true", help="cfg3: three library calls per op instead of the fused entry point")
    ap.add_argument("--dry-shard", action="store_true", help="print the partition of the job over --gpus ranks and exit (no GPU)")
    ap.add_argument("--dry-run", action="store_true", help="run the launcher / rendezvous / key broadcast / timing reduction with the gloo backend on "
                                                              "CPU tensors and no evaluation (exercises the N > 1 plumbing where there is no GPU)")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="--dry-run only: this rank exits non-zero before the rendezvous (launcher watchdog test)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# self-launcher: `python bench.py --gpus N` outside torchrun
# ------------------------------------------------------------------------------------------------------
def _reserve_port():
    """a free rendezvous port; the socket stays open (SO_REUSEADDR) until the children have been started, which narrows the window in
    which another process could take the port"""
    s = socket.socket()
    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    s.bind(("127.0.0.1", 0))
    return s, s.getsockname()[1]


def launch_ranks(n, poll_s=0.2):
    """start n FRESH rank processes (this process has not initialised HIP; nothing is ever re-exec'ed), relay rank 0's output and
    watch every child: the first non-zero exit terminates the others, so a dead rank cannot leave rank 0 waiting in a collective"""
    holder, port = _reserve_port()
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        err = open(os.path.join("/tmp", "bench_rank%d_%d.err" % (r, os.getpid())), "w+")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err, text=True))
    holder.close()
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(poll_s)
    reader.join(timeout=10)
    text = out0[0] if out0 else ""
    sys.stdout.write(text or "")
    sys.stdout.flush()
    for r, err in enumerate(errs):
        err.seek(0)
        tail = err.read()[-2000:]
        err.close()
        try:
            os.unlink(err.name)
        except OSError:
            pass
        if tail.strip() and (failed or r == 0):
            sys.stderr.write("---- rank %d stderr ----\n%s\n" % (r, tail))
    if failed:
        sys.stderr.write("bench.py: rank(s) failed: %s; the remaining ranks were terminated\n" % failed)
        return 1
    try:
        line = [ln for ln in (text or "").splitlines() if ln.startswith("{")][-1]
        if json.loads(line).get("n_gpus") != n:
            sys.stderr.write("bench.py: asked for %d GPUs, the job reports %s\n" % (n, json.loads(line).get("n_gpus")))
            return 3
    except (IndexError, ValueError):
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 4
    return 0


# ------------------------------------------------------------------------------------------------------
# Output: the driver parses ONE line -- the LAST line of stdout.  It is the headline and stays below HEADLINE_LIMIT bytes (numbers and short
# names only; prose lives in DESIGN.md section 6).  Everything else the run measured is printed BEFORE it, one self-contained JSON line per block
# ({"extra": "<name>", ...}), each below EXTRA_LIMIT bytes.  (Round 5 printed one 23 KB line; the driver could not parse it.)
# ------------------------------------------------------------------------------------------------------
HEADLINE_LIMIT = 6144
EXTRA_LIMIT = 8192
HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                 "config", "roofline", "cpu_baseline", "parity")
ROOFLINE_KEYS = ("bound", "limiter", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_record", "launch_ms", "launches_timed", "items_per_launch",
                 "algorithmic_bytes_per_launch", "share_of_step")


def _pack(tag, block, limit=EXTRA_LIMIT):
    """one or more {"extra": tag, ...} dicts whose JSON stays below `limit`: top-level keys are packed greedily; a single value that is too large on its
    own is descended into (dict) or dropped with a marker -- an extra line never endangers the headline"""
    lines, cur = [], {"extra": tag}
    for k, v in block.items():
        piece = len(json.dumps({k: v})) + 2
        if piece + len(json.dumps({"extra": tag})) >= limit:
            if len(cur) > 1:
                lines.append(cur)
                cur = {"extra": tag}
            if isinstance(v, dict):
                lines.extend(_pack("%s.%s" % (tag, k), v, limit))
            else:
                lines.append({"extra": tag, k: "dropped: %d bytes" % piece})
            continue
        if len(json.dumps(cur)) + piece >= limit:
            lines.append(cur)
            cur = {"extra": tag, "continued": True}
        cur[k] = v
    if len(cur) > (2 if cur.get("continued") else 1):
        lines.append(cur)
    return lines


def split_output(result):
    """(extra lines, headline) of a run_* result.  The headline keeps HEADLINE_KEYS; of `roofline` the numeric fields, valu_fp64 as {frac, busy, record} and
    floor as four numbers + the record's file name; every other block becomes an extra line."""
    result = dict(result)
    extras = []
    other = result.pop("other_configs", None) or {}
    roof = dict(result.get("roofline") or {})
    head_roof = {k: roof.pop(k) for k in ROOFLINE_KEYS if k in roof}
    vf = roof.get("valu_fp64")
    if vf:
        head_roof["valu_fp64"] = {"frac": vf.get("frac"), "lane_ops_per_launch": vf.get("lane_ops_per_launch"), "peak_lane_ops_per_s": vf.get("peak_lane_ops_per_s"),
                                  "simd_valu_busy_profiled": vf.get("simd_valu_busy_profiled"), "record": vf.get("record")}
    fl = roof.get("floor")
    if fl:
        head_roof["floor"] = {k: fl.get(k) for k in ("lane_ops_per_pass", "nominal_ms_per_pass_at_full_issue", "measured_ms_per_pass", "issue_frac_profiled", "record")}
    if "pipeline" in roof:
        head_roof["pipeline"] = roof.pop("pipeline")
    if roof:
        extras.extend(_pack("roofline_detail", roof))
    head = {k: result.pop(k) for k in HEADLINE_KEYS if k in result}
    head["roofline"] = head_roof
    if result:                      # anything a run_* function added that is not a headline key
        extras.extend(_pack("more", result))
    small = {k: v for k, v in other.items() if k not in ("cfg4", "cfg5", "cpp_api", "single_object_latency_us")}
    if small:
        extras.extend(_pack("sizes", small))
    for k in ("cfg4", "cfg5", "cpp_api", "single_object_latency_us"):
        if k in other:
            v = other[k]
            if k == "cfg4":         # itself a bench result: same split, its headline becomes the block's first line
                sub_extras, sub_head = split_output(v)
                extras.extend(_pack("cfg4", sub_head))
                for e in sub_extras:
                    e["extra"] = "cfg4." + e["extra"]
                    extras.append(e)
            else:
                extras.extend(_pack(k, v))
    head["extras"] = sorted({e["extra"].split(".")[0] for e in extras})
    return extras, head


def emit(result, out=sys.stdout):
    extras, head = split_output(result)
    for e in extras:
        out.write(json.dumps(e) + "\n")
    line = json.dumps(head)
    if len(line) >= HEADLINE_LIMIT:
        raise RuntimeError("bench.py: the headline line is %d bytes (limit %d)" % (len(line), HEADLINE_LIMIT))
    out.write(line + "\n")
    out.flush()
    return line


def shard_plan(total, world):
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


def timed_broadcast(torch, shard, keys):
    """the one exchange of the batched path (evaluation keys, rank 0 -> all), outside the timed region; seconds on this rank"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    shard.broadcast_tensors(keys, src=0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def timed(torch, fn, reps):
    # the clocks need 20-25 ms of load to come up after any idle gap (tools/ramp_probe.py): warm up for at least 50 ms
    t0 = time.perf_counter()
    while True:
        fn()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 0.05:
            break
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def kernel_sources_sha(names):
    """stamp of the kernel sources a counter record belongs to (tools/collect_counters.py writes the same value)"""
    h = hashlib.sha256()
    for nm in names:
        with open(os.path.join(ROOT, "troy-nova_amd", "csrc", nm), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# everything the timed instantiation depends on: the kernel, its arithmetic, the helpers it takes from the transform header, the unit that
# picks the instantiation, and troyn.hip (workgroup order, argument block).  tools/collect_counters.py hashes the same list.
KSMAC_SOURCES = ("ksmac_kernels.hpp", "dev_math_f64.hpp", "dev_math.hpp", "ntt_kernels.hpp", "troyn_ksmac2.hip", "launch.hpp", "troyn.hip")


def counters_record(kernel_tag, batch, kernel_contains=None):
    """newest profiles/r*_<kernel_tag>_counters.json whose kernel sources are the ones being timed (a record of an older kernel is stale: dropped)"""
    sha = kernel_sources_sha(KSMAC_SOURCES)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_counters.json" % kernel_tag)), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("kernel_src_sha") == sha and rec.get("batch") == batch and (kernel_contains is None or kernel_contains in rec.get("kernel", "")):
            rec["_file"] = os.path.relpath(path, ROOT)
            return rec
    return None


def all_sources_sha():
    """stamp of every kernel source (tools/collect_valu.py writes the same value)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "troy-nova_amd", "csrc")
    for nm in sorted(os.listdir(d)):
        if nm.endswith((".hip", ".hpp", ".inl")):
            with open(os.path.join(d, nm), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def valu_record(tag):
    """newest profiles/r*_<tag>_valu.json (tools/collect_valu.py: per-kernel VALU issue accounting of a profiled run) measured on the current kernel sources"""
    sha = all_sources_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_valu.json" % tag)), reverse=True):
        try:
            rec = json.load(open(path))
        except (OSError, ValueError):
            continue
        if rec.get("sources_sha") == sha:
            rec["_file"] = os.path.relpath(path, ROOT)
            return rec
    return None


class KernelTimer:
    """the library's measurement hook (include/troyn.h troyn_kernel_timer_*): hipEvent pairs on the launch stream around one named launch"""

    def __init__(self, pkg, lib, region):
        self.pkg, self.lib, self.region = pkg, lib, region

    def __enter__(self):
        self.pkg.capi.check(self.lib.troyn_kernel_timer_enable(self.region, 1))
        return self

    def __exit__(self, *exc):
        self.pkg.capi.check(self.lib.troyn_kernel_timer_enable(self.region, 0))

    def read(self):
        import ctypes as C
        ms, n = C.c_double(0.0), C.c_uint64(0)
        self.pkg.capi.check(self.lib.troyn_kernel_timer_read(self.region, C.byref(ms), C.byref(n)))
        return ms.value, int(n.value)


def ksmac_alg_bytes(B, n, L, ntt_form, fused_chain=False):
    """algorithmic bytes of one key-switch inner-product launch (SURVEY.md 8d minimum traffic): per item the L
    coefficient-form digits (+ the L NTT-form input limbs when the operand is in NTT form) read once, the two output polynomials of L+1
    rows written once; the key set (16*N*L*(L+1) bytes) is shared by the whole batch and counted once per launch.
    fused_chain: the kernel of troyn_ckks_multiply_relinearize_rescale reads the L limbs of all four input polynomials a0, a1, b0, b1 once
    (the diagonal digit a1 (.) b1 and the tensor terms it adds to the data rows) instead of one NTT-form limb per digit"""
    operands = 4 if fused_chain else (1 if ntt_form else 0)
    return B * ((1 + operands) * 8.0 * n * L + 16.0 * n * (L + 1)) + 16.0 * n * L * (L + 1)


def run_cfg3(args, torch, pkg, shard, entry, rank, world, device):
    n, log_n = 16384, 14
    q = pkg.capi.coeff_modulus_create(n, [50] * 6)
    K, L = 6, 5
    B = args.batch or 1024
    inner = args.inner or 12
    plan = pkg.Plan(device, log_n, q)
    gen = torch.Generator(device=device).manual_seed(0x123 + rank)
    a = uniform_residues(torch, (B, 2), q[:L], n, device, gen)
    b = uniform_residues(torch, (B, 2), q[:L], n, device, gen)
    # relinearization keys: L keys of [2][K][N] uniform residues (timing-identical to genuine keys);
    # generated on rank 0 and broadcast once over RCCL (the only exchange of the batched path)
    kgen = torch.Generator(device=device).manual_seed(0xC0FFEE)
    keys = [uniform_residues(torch, (2,), q, n, device, kgen) for _ in range(L)]
    key_broadcast_s = timed_broadcast(torch, shard, keys)

    prod = torch.empty((B, 3, L, n), dtype=torch.int64, device=device)
    relin = torch.empty((B, 2, L, n), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, n), dtype=torch.int64, device=device)
    fused = hasattr(plan, "ckks_multiply_relinearize_rescale") and not args.unfused

    def pass3():
        plan.dyadic_convolute(a, 2, b, 2, L, out=prod)                       # Evaluator::multiply (CKKS)
        plan.relinearize(L, prod, keys, out=relin, is_ckks=True, is_ntt_form=True)  # Evaluator::relinearize
        plan.divide_and_round_q_last_ntt(L, relin, 2, out=out)               # Evaluator::rescale_to_next

    def pass1():
        plan.ckks_multiply_relinearize_rescale(L, a, b, keys, out=out)       # the same three calls behind one entry point

    one_pass = pass1 if fused else pass3

    def step():
        for _ in range(inner):
            one_pass()

    for _ in range(args.warmup):
        step()
    lib = plan.lib
    with KernelTimer(pkg, lib, TIMER_KS) as kt:
        shard.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        local_elapsed = time.perf_counter() - t0          # this rank's own work (the contract's clock runs to the closing barrier)
        shard.barrier()
        elapsed = time.perf_counter() - t0
        ks_ms, ks_n = kt.read()
    ranks = shard.rank_report(local_elapsed, args.steps, torch.cuda.get_device_name(device), key_broadcast_s, device=device)
    elapsed = shard.max_over_ranks(elapsed, device=device)
    value = world * B * inner * args.steps / elapsed

    # ---- roofline of the kernel with the largest share of the timed step: the fused key-switch inner product ----
    # The fused entry runs the whole batch as one launch sequence on the caller's stream (the default), so the launches timed inside the
    # region are the kernel's own.  If TROYN_MRR_CHUNK is set the batch runs as chunks on internal streams and a launch's event-bracketed
    # duration includes the time it shares the chip with another chunk's kernels (`in_region`); the kernel's own figure is then measured in
    # the same run right after the region, same buffers, with the chain on ONE stream (Plan.set_option).
    # `achieved` = algorithmic bytes per launch / the average of the kernel's own launch durations.
    in_region_ms = ks_ms / max(1, ks_n)
    in_region_items = max(1, round(B * inner * args.steps / max(1, ks_n)))
    excl_ms, excl_n = in_region_ms, ks_n
    if fused and in_region_items != B:
        plan.set_option("TROYN_MRR_CHUNK", "0")       # (the library reads its switches when the plan is created; per-plan changes go through set_option)
        try:
            one_pass()
            torch.cuda.synchronize()
            with KernelTimer(pkg, lib, TIMER_KS) as kt:
                for _ in range(max(inner, 8)):
                    one_pass()
                torch.cuda.synchronize()
                ms1, n1 = kt.read()
            excl_ms, excl_n = ms1 / max(1, n1), n1
        finally:
            plan.set_option("TROYN_MRR_CHUNK", os.environ.get("TROYN_MRR_CHUNK"))
    ks_launch_ms = excl_ms
    items_per_launch = B if (fused and in_region_items != B) else in_region_items
    alg_bytes = ksmac_alg_bytes(items_per_launch, n, L, True, fused_chain=fused)
    achieved = alg_bytes / (ks_launch_ms * 1e-3) / 1e9 if ks_launch_ms else 0.0
    prof = counters_record("ksmac", items_per_launch, "ksmac2_kernel<14")   # tools/profile_bench.sh + tools/collect_counters.py, separate rocprofv3 --pmc passes
    traffic, valu = None, None
    if prof:
        traffic = prof.get("traffic_bytes_per_launch")
        if prof.get("valu_lane_ops_per_launch"):
            rate = prof["valu_lane_ops_per_launch"] / (ks_launch_ms * 1e-3)
            valu = {"record": prof["_file"], "lane_ops_per_launch": prof["valu_lane_ops_per_launch"], "achieved_lane_ops_per_s": round(rate, 0),
                    "peak_lane_ops_per_s": FP64_VALU_PEAK, "frac": round(rate / FP64_VALU_PEAK, 4),
                    "simd_valu_busy_profiled": prof.get("simd_valu_busy"), "effective_clock_GHz_profiled": prof.get("effective_clock_GHz")}
    # `bound` / achieved / peak / frac are the contract's HBM figures (algorithmic bytes per launch / the launch duration measured in this run, against
    # 8 TB/s); what limits the kernel is FP64 vector issue: `limiter` + valu_fp64 (DESIGN.md section 6 has the prose)
    roofline = {"bound": "hbm", "limiter": "valu_fp64" if valu else None,
                "kernel": "ksmac2_kernel<14,TEN> (fused key-switch inner product)" if fused else "ksmac2_kernel<14,DG> (key-switch inner product)",
                "share_of_step": round(ks_launch_ms * (B / items_per_launch) * inner * args.steps / (elapsed * 1e3), 4),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_record": prof["_file"] if prof else None,
                "launch_ms": round(ks_launch_ms, 4), "launches_timed": excl_n, "items_per_launch": items_per_launch,
                "measured": "library kernel timer: HIP events on the launch stream around every launch" + (" of the timed region" if items_per_launch == in_region_items else "; same run and buffers, chain on one stream"),
                "in_region": None if items_per_launch == in_region_items else
                             {"launch_ms": round(in_region_ms, 4), "launches_timed": ks_n, "items_per_launch": in_region_items,
                              "note": "TROYN_MRR_CHUNK set: chunks in flight on internal streams, a launch shares the chip with another chunk's kernels"},
                "algorithmic_bytes_per_launch": alg_bytes, "valu_fp64": valu,
                # whole pipeline against the chip's HBM peak, both key accountings of SURVEY 8d (keys per op / keys once per batch)
                "pipeline": {"bytes_per_op_keys_per_op": 18.0e6, "frac_keys_per_op": round(value / world * 18.0e6 / (HBM_PEAK_GBS * 1e9), 4),
                             "bytes_per_op_keys_amortised": 10.2e6, "frac_keys_amortised": round(value / world * 10.2e6 / (HBM_PEAK_GBS * 1e9), 4)}}

    # The floor of this chain, written down so that it is not asked again: every kernel of the pass is bound by vector-ALU issue
    # (exact 50-bit modular arithmetic carried in FP64: ~107 FP64 instructions per coefficient and digit in the inner product, ~81-91 per coefficient in the
    # transforms) -- chain_valu is the per-kernel accounting of a profiled pass; five rebuilds of the inner product (rounds 3-4) and the scheduling / window
    # variants of round 5 (profiles/r05_ksmac_variants.txt) moved its launch by < 1.5 %.
    cv = valu_record("bench_chain") if (rank == 0 and fused) else None
    roofline["floor"] = {
        "record": cv["_file"] if cv else None,
        "lane_ops_per_pass": cv["per_pass"]["valu_lane_ops"] if cv else None,
        "nominal_ms_per_pass_at_full_issue": cv["per_pass"]["nominal_ms_at_full_issue"] if cv else None,
        "measured_ms_per_pass": round(elapsed / args.steps / inner * 1e3, 4),
        "issue_frac_profiled": cv["per_pass"]["issue_frac"] if cv else None,
        "kernels": [{k: r[k] for k in ("kernel", "avg_us", "time_share", "valu_lane_ops", "simd_valu_busy", "nominal_us_at_full_issue") if k in r} for r in cv["kernels"] if r.get("launches_per_pass", 1) > 0][:8] if cv else None,
        "experiments": ["profiles/r03_ksmac_ab.txt", "profiles/r04_ksmac_ab.txt", "profiles/r05_ksmac_variants.txt"]}

    # the reference's operator boundary: the same ops as three library calls (Evaluator::multiply / relinearize / rescale_to_next), same buffers
    three_call = None
    if fused:
        t3 = timed(torch, pass3, max(3, min(20, args.steps)))
        three_call = round(world * B / shard.max_over_ranks(t3, device=device), 1)

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
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "CKKS N=16384, 6x50-bit coeff modulus (K=6, L=5): multiply + relinearize + rescale_to_next (BASELINE configs[2])",
                   "batch_per_gpu": B, "passes_per_step": inner, "ops_per_step": world * B * inner,
                   "entry": "troyn_ckks_multiply_relinearize_rescale" if fused else "troyn_dyadic_convolute + troyn_relinearize + troyn_divide_and_round_q_last_ntt",
                   "three_call_ops_per_s": three_call,
                   "parallelism": "batch-sharded x%d, keys broadcast once (RCCL)" % world,
                   # what the backend saw, per rank (a first multi-GPU run that goes wrong must be readable from this line alone)
                   **ranks},
        "roofline": roofline,
    }

    # ---- in-run parity + CPU baseline (rank 0, N = 1 only): the oracle is the checker / reported baseline ----
    if rank == 0 and world > 1 and not args.no_cpu_baseline:
        # N > 1: rank 0 still checks items of ITS batch against the oracle (no CPU baseline: that is the N = 1 line's)
        import numpy as np
        O = entry.load_oracle()
        ctx = O.Context("ckks", n, q)
        hk = [pkg.to_host(k) for k in keys]
        if fused:
            one_pass()
        torch.cuda.synchronize()
        items = sorted({0, B - 1})
        for i in items:
            e = ctx.relinearize(L, True, ctx.ckks_multiply(L, pkg.to_host(a[i]), pkg.to_host(b[i])), hk)
            if not np.array_equal(pkg.to_host(out[i]), ctx.mod_switch_scale_to_next(L, e)):
                raise AssertionError("bench: GPU result of item %d (rank 0 of %d) differs from the CPU oracle" % (i, world))
        result["parity"] = "bit-exact vs CPU oracle (rank 0, items %s of %d)" % (items, B)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import numpy as np
        O = entry.load_oracle()
        ctx = O.Context("ckks", n, q)
        hk = [pkg.to_host(k) for k in keys]

        def one_op(c, ha, hb):
            e = c.ckks_multiply(L, ha, hb)
            e = c.relinearize(L, True, e, hk)
            return c.mod_switch_scale_to_next(L, e)

        if fused:
            one_pass()      # `out` holds the three-call results after the side measurement; both are checked
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

    # ---- the other BASELINE configurations and north_star's other sizes, same run (rank 0, 1 GPU) -------------------------------
    if rank == 0 and world == 1 and not args.no_extra:
        del a, b, prod, relin, out, plan
        torch.cuda.empty_cache()
        other = extra_configs(torch, pkg, device)
        sub = parse_args(["--workload", "cfg4", "--total", "256", "--steps", "3", "--warmup", "2", "--cpu-seconds", "4"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else []))
        other["cfg4"] = run_cfg4(sub, torch, pkg, shard, entry, 0, 1, device)
        other["cfg4"]["note"] = "BASELINE configs[3] on one GPU with a 256-op job per step (`bench.py --workload cfg4` runs the 1024-op job and shards it over --gpus ranks)"
        torch.cuda.empty_cache()
        other["cfg5"] = run_cfg5(args, torch, pkg, entry, device)
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info()
        other["cpp_api"] = run_cpp_api()
        other["cpp_api"]["device_memory_free_before_GB"] = round(free_b / 2**30, 1)
        other["single_object_latency_us"] = single_object_latency(torch, pkg, device)
        result["other_configs"] = other
    return result


def single_object_latency(torch, pkg, device):
    """ONE ciphertext through the C-ABI (ctypes layer, a stream wait after every call -- what an unmodified single-object caller sees): the fused entry and
    the three calls, for the reference bench tool's default chain (test/bench/he_operations.cu:22-24), its all-40-bit sibling and the headline chain with and
    without 60-bit primes.  A launch that cannot fill the chip is latency-bound: two-pass transforms, merged strided passes (troyn_mrr_small.hip), the
    two-launch inner product for chains with moduli >= 2^50."""
    res = {"what": "microseconds per call, batch 1, 300 calls after 30 warm-up calls, torch.cuda.synchronize() after each", "shapes": []}
    for n, bits, L in ((8192, [60, 40, 40, 60], 3), (8192, [40, 40, 40, 40], 3), (16384, [50] * 6, 5), (16384, [60, 50, 50, 50, 50, 60], 5)):
        gen = torch.Generator(device=device).manual_seed(7)
        q = pkg.capi.coeff_modulus_create(n, bits)
        plan = pkg.Plan(device, n.bit_length() - 1, q)
        x, y = uniform_residues(torch, (1, 2), q[:L], n, device, gen), uniform_residues(torch, (1, 2), q[:L], n, device, gen)
        keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
        out = torch.empty((1, 2, L - 1, n), dtype=torch.int64, device=device)
        prod = torch.empty((1, 3, L, n), dtype=torch.int64, device=device)
        rel = torch.empty((1, 2, L, n), dtype=torch.int64, device=device)

        def lat(f, reps=300):
            for _ in range(30):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
                torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / reps * 1e6, 1)
        res["shapes"].append({"n": n, "chain_bits": bits, "L": L,
                              "fused_mul_relin_rescale": lat(lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out)),
                              "multiply": lat(lambda: plan.dyadic_convolute(x, 2, y, 2, L, out=prod)),
                              "relinearize": lat(lambda: plan.relinearize(L, prod, keys, out=rel, is_ckks=True, is_ntt_form=True)),
                              "rescale": lat(lambda: plan.divide_and_round_q_last_ntt(L, rel, 2, out=out))})
        del plan, x, y, keys, out, prod, rel
    torch.cuda.empty_cache()
    return res


def extra_configs(torch, pkg, device):
    """NTT + dyadic + INTT and relinearize throughput at N = 8192 (cfg2 shape), 16384 and 32768 (cfg4 shape); synthetic residues"""
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

    def transform_pipeline(n, log_n, bits, L, Bn, key, what, ntt_form, two_pass=False, behz_t=0):
        q = pkg.capi.coeff_modulus_create(n, bits)
        K = len(q)
        plan = pkg.Plan(device, log_n, q)
        x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
        xn, yn = torch.empty_like(x), torch.empty_like(y)
        prod = torch.empty((Bn, 3, L, n), dtype=torch.int64, device=device)
        keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
        out2 = torch.empty((Bn, 2, L, n), dtype=torch.int64, device=device)

        def pipe():
            plan.ntt(x, 2, L, out=xn); plan.ntt(y, 2, L, out=yn)
            plan.dyadic_convolute(xn, 2, yn, 2, L, out=prod)
            plan.ntt(prod, 3, L, inverse=True)
        reps = 10 if n < 32768 else 5
        t = timed(torch, pipe, reps)
        # BASELINE's second figure: the transform alone against HBM -- algorithmic bytes 16 N per limb-polynomial, out of place, forward and inverse
        t_f = timed(torch, lambda: plan.ntt(x, 2, L, out=xn), reps)
        t_i = timed(torch, lambda: plan.ntt(xn, 2, L, out=yn, inverse=True), reps)
        ntt_bytes = 16.0 * n * L * 2 * Bn
        # 4 NTT (16 B/coeff) + dyadic (56) + 3 INTT (16) per limb coefficient; two-pass transforms move every coefficient twice
        alg = ((7 * 32.0 + 56.0) if two_pass else 168.0) * n * L * Bn
        with KernelTimer(pkg, plan.lib, TIMER_KS) as kt:
            tr = timed(torch, lambda: plan.relinearize(L, prod, keys, out=out2, is_ckks=ntt_form, is_ntt_form=ntt_form), reps)
            ks_ms, ks_n = kt.read()
        # relinearize against SURVEY 8d's bytes for one key switch: keys 16 L K N (per op / once per batch) + target 8 L N + destination 16 L N
        # (+ the c0, c1 read and the c2 INTT input of relinearize are left out: minimum traffic)
        ks_keys, ks_rest = 16.0 * L * K * n, 24.0 * L * n
        r = {"what": what, "batch": Bn, "ntt_dyadic_intt_ciphertexts_per_s": round(Bn / t, 1), "ntt_dyadic_intt_hbm_frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 4),
             "ntt_GBps": round(ntt_bytes / t_f / 1e9, 1), "ntt_hbm_frac": round(ntt_bytes / t_f / 1e9 / HBM_PEAK_GBS, 4),
             "intt_GBps": round(ntt_bytes / t_i / 1e9, 1), "intt_hbm_frac": round(ntt_bytes / t_i / 1e9 / HBM_PEAK_GBS, 4),
             "relinearize_ops_per_s": round(Bn / tr, 1),
             "relinearize_hbm_frac_keys_per_op": round((ks_keys + ks_rest) * Bn / tr / 1e9 / HBM_PEAK_GBS, 4),
             "relinearize_hbm_frac_keys_once_per_batch": round((ks_keys + ks_rest * Bn) / tr / 1e9 / HBM_PEAK_GBS, 4)}
        if ks_n:
            r["relinearize_inner_product_launch_ms"] = round(ks_ms / ks_n, 4)
            r["relinearize_inner_product_hbm_frac"] = round(ksmac_alg_bytes(Bn, n, L, ntt_form) / (ks_ms / ks_n * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if behz_t:
            behz = pkg.Behz(plan, L, behz_t)
            tm = timed(torch, lambda: behz.multiply(x, 2, y, 2, out=prod), reps)
            r["multiply_ops_per_s"] = round(Bn / tm, 1)
            r["multiply_relinearize_ops_per_s"] = round(Bn / (tm + tr), 1)
            del behz
        res[key] = r
        del x, y, xn, yn, prod, keys, out2, plan
        torch.cuda.empty_cache()

    transform_pipeline(8192, 13, [40, 40, 40], 2, 2048, "N8192_L2",
                       "BFV N=8192, 3x40-bit (L=2, BASELINE configs[1]): NTT(a), NTT(b) + dyadic 2x2->3 + INTT(3) as separate calls; relinearize (coefficient form); BEHZ multiply",
                       False, behz_t=1032193)
    transform_pipeline(16384, 14, [50] * 6, 5, 512, "N16384_L5",
                       "N=16384, 6x50-bit (L=5): NTT(a), NTT(b) + dyadic 2x2->3 + INTT(3) as separate calls; relinearize (NTT form)", True)
    transform_pipeline(32768, 15, [50] * 11, 10, 64, "N32768_L10",
                       "BFV N=32768, 11x50-bit (L=10): NTT+dyadic+INTT; BEHZ multiply; relinearize; multiply+relinearize", False, two_pass=True, behz_t=1032193)
    # the reference's default chain (test/bench/he_operations.cu:19-33): 60-bit primes take the integer policy
    transform_pipeline(8192, 13, [60, 40, 40, 60], 3, 1024, "N8192_60_40_40_60",
                       "N=8192 {60,40,40,60} (L=3, K=4; the reference bench tool's default log_q): transforms split by modulus class, relinearize (NTT form): 40-bit rows on "
                       "ksmac2, 60-bit rows on ksmaci_kernel", True)
    # the usual CKKS shape: wide first and special primes around 50-bit scaling primes (mixed arithmetic classes at N = 16384).  Since round 5 the fused
    # chain runs per modulus class for such chains (integer kernels for the wide limbs: ksmaci_kernel, integer forms of the fused transforms).
    transform_pipeline(16384, 14, [60, 50, 50, 50, 50, 60], 5, 512, "N16384_60_50_50_50_50_60",
                       "N=16384 {60,50,50,50,50,60} (L=5, K=6): transforms split by modulus class; relinearize (NTT form): rows of the 50-bit moduli on ksmac2 (wide digits "
                       "reduced in FP64 while loading), rows of the 60-bit moduli on the integer half-tile inner product (ksmaci_kernel)", True)
    for key, n, log_n, bits, L, Bn in (("N16384_60_50_50_50_50_60", 16384, 14, [60, 50, 50, 50, 50, 60], 5, 512), ("N8192_60_40_40_60", 8192, 13, [60, 40, 40, 60], 3, 1024)):
        q = pkg.capi.coeff_modulus_create(n, bits)
        plan = pkg.Plan(device, log_n, q)
        x, y = uniform_residues(torch, (Bn, 2), q[:L], n, device, gen), uniform_residues(torch, (Bn, 2), q[:L], n, device, gen)
        keys = [uniform_residues(torch, (2,), q, n, device, gen) for _ in range(L)]
        out = torch.empty((Bn, 2, L - 1, n), dtype=torch.int64, device=device)
        t = timed(torch, lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), 10)
        res[key]["ckks_mul_relin_rescale_ops_per_s"] = round(Bn / t, 1)
        res[key]["ckks_mul_relin_rescale_batch"] = Bn
        plan.set_option("TROYN_MRR_MIXED", "0")         # rounds 2-4: the three calls composed inside the entry
        t3 = timed(torch, lambda: plan.ckks_multiply_relinearize_rescale(L, x, y, keys, out=out), 10)
        res[key]["ckks_mul_relin_rescale_three_calls_ops_per_s"] = round(Bn / t3, 1)
        del x, y, keys, out, plan
        torch.cuda.empty_cache()
    return res


def cfg4_valu_block(nb, n, L, S):
    """SURVEY 8d's secondary ceiling for cfg4: per kernel, SIMD vector-ALU busy = SQ_ACTIVE_INST_VALU x 4 cycles / (launch cycles x 1024 SIMDs)
    and VALU instructions per coefficient, from the PMC passes of the cfg4 workload (tools/profile_cfg4.sh -> profiles/rNN_cfg4_valu.json).  Every VALU instruction of
    gfx950 (v_mad_u64_u32, v_mul_lo/hi_u32, v_fma_f64, adds) holds its SIMD for 4 cycles per wave64, so a kernel at busy ~1 can only get faster with fewer instructions."""
    rec = valu_record("cfg4")
    if not rec:
        return {"record": None, "note": "no profiles/r*_cfg4_valu.json for the current kernel sources (tools/profile_cfg4.sh)"}
    per_coeff = {"tensor_core_kernel<troyn::ArithU64": nb * S * n * 7.0,      # 4 operand + 3 product polynomials per auxiliary limb
                 "tensor_core_kernel<troyn::ArithF64": nb * L * n * 7.0,
                 "behz2_floor": nb * 3 * n * 1.0,                              # per coefficient of a result polynomial (all limbs; behz2_floor_pass2_kernel
                                                                               # includes the last inverse pass of its L + S rows)
                 "behz2_lift": nb * 2 * n * 1.0,                               # (behz2_lift_pass1_kernel: + the first forward pass of L + S rows)
                 "ksmac2_kernel": nb * (L + 1) * L * n * 1.0}                  # per coefficient and (row, digit)
    out = []
    for r in rec["kernels"][:10]:
        e = {k: r[k] for k in ("kernel", "avg_us", "time_share", "simd_valu_busy", "valu_lane_ops", "nominal_us_at_full_issue") if k in r}
        for key, denom in per_coeff.items():
            if key in r["kernel"]:
                e["valu_instructions_per_coefficient"] = round(r["valu_lane_ops"] / denom, 1)
        out.append(e)
    return {"record": rec["_file"], "what": "per kernel of one 64-pair chunk: SIMD VALU busy and instruction counts (integer: 4 v_mad_u64_u32 per base-conversion term with no "
                                          "carries, 20 instructions per Harvey butterfly; FP64: 8 per butterfly)",
            "per_chunk": rec.get("per_pass"), "kernels": out}


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
    S = behz.working_base_size      # limbs of the auxiliary base the multiply works in (11 + 1 primes below 2^50 here since round 5; the reference's base has 10 + 1 of 61 bits)
    gen = torch.Generator(device=device).manual_seed(0x123)      # the same job on every world size: item i has the same payload
    kgen = torch.Generator(device=device).manual_seed(0xC0FFEE)
    keys = [uniform_residues(torch, (2,), q, n, device, kgen) for _ in range(L)]
    key_broadcast_s = timed_broadcast(torch, shard, keys)
    nb = min(chunk, max(mine, 1))
    # the job's real operands: `mine` DISTINCT ciphertext pairs resident in HBM (1024 items = 2 x 5 GiB; the intermediates and results of a
    # chunk are reused, a caller would consume them before the next launch)
    x = torch.cat([uniform_residues(torch, (min(64, mine - i), 2), q[:L], n, device, gen) for i in range(0, max(mine, 1), 64)]) if mine else uniform_residues(torch, (1, 2), q[:L], n, device, gen)
    y = torch.cat([uniform_residues(torch, (min(64, mine - i), 2), q[:L], n, device, gen) for i in range(0, max(mine, 1), 64)]) if mine else uniform_residues(torch, (1, 2), q[:L], n, device, gen)
    prod = torch.empty((nb, 3, L, n), dtype=torch.int64, device=device)
    out = torch.empty((nb, 2, L, n), dtype=torch.int64, device=device)
    launches = [0]

    def step():
        done = 0
        while done < mine:      # the rank's slice, `chunk` ciphertext pairs per launch, every pair its own operands
            c = min(nb, mine - done)
            behz.multiply(x[done:done + c], 2, y[done:done + c], 2, out=prod[:c])                   # Evaluator::multiply (BFV, BEHZ)
            plan.relinearize(L, prod[:c], keys, out=out[:c], is_ckks=False, is_ntt_form=False)      # Evaluator::relinearize
            done += c
            launches[0] += 1

    for _ in range(args.warmup):
        step()
    lib = plan.lib
    launches[0] = 0
    with KernelTimer(pkg, lib, TIMER_KS) as kt_ks, KernelTimer(pkg, lib, TIMER_TENSOR) as kt_t, KernelTimer(pkg, lib, TIMER_FLOOR) as kt_f:
        shard.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        local_elapsed = time.perf_counter() - t0          # this rank's own work (the contract's clock runs to the closing barrier)
        shard.barrier()
        elapsed = time.perf_counter() - t0
        ks_ms, ks_n = kt_ks.read()
        tn_ms, tn_n = kt_t.read()
        fl_ms, fl_n = kt_f.read()
    ranks = shard.rank_report(local_elapsed, args.steps, torch.cuda.get_device_name(device), key_broadcast_s, device=device)
    ranks["items_per_rank"] = [int(v) for v in shard.gather_over_ranks(mine, device=device)]
    elapsed = shard.max_over_ranks(elapsed, device=device)
    value = args.total * args.steps / elapsed

    # ---- roofline: the launch with the largest share of the step.  Full chunks only (the job is a multiple of the chunk in every quoted run).
    share = lambda ms: 100.0 * ms / (elapsed * 1e3)
    ks_launch_ms = ks_ms / max(1, ks_n)
    ks_alg = ksmac_alg_bytes(nb, n, L, False)
    # tensor_core_kernel: per limb 4 operand rows read + 3 result rows written (8*N bytes each), L limbs of base q + S limbs of base Bsk per item;
    # the region brackets both bases' launches of a chunk, so per multiply call
    tn_call_ms = tn_ms / max(1, launches[0])
    tn_alg = nb * 7.0 * 8 * n * (L + S)
    fl_call_ms = fl_ms / max(1, fl_n)
    fl_alg = nb * 3 * 8.0 * n * (2 * L + S)     # last inverse pass + floor in one launch: per result polynomial L + S limbs read, L written
    kernels = {
        "ksmac2_kernel<15> (key-switch inner product, quarter tiles)": {"launch_ms": round(ks_launch_ms, 4), "share_of_step_pct": round(share(ks_ms), 1),
                                                                       "algorithmic_bytes_per_launch": ks_alg, "hbm_frac": round(ks_alg / (ks_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ks_n else None},
        "tensor_core_kernel (last forward pass + tensor product + first inverse pass; both bases on FP64 butterflies since the auxiliary primes are below 2^50)": {
            "ms_per_multiply_call": round(tn_call_ms, 4), "share_of_step_pct": round(share(tn_ms), 1), "algorithmic_bytes_per_call": tn_alg,
            "hbm_frac": round(tn_alg / (tn_call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tn_n else None},
        "behz2_floor_pass2_kernel<10> (last inverse pass of both bases + floor)": {"launch_ms": round(fl_call_ms, 4), "share_of_step_pct": round(share(fl_ms), 1), "algorithmic_bytes_per_launch": fl_alg,
                                   "hbm_frac": round(fl_alg / (fl_call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if fl_n else None},
    }
    dominant_is_ks = ks_ms >= tn_ms
    dom_alg, dom_ms = (ks_alg, ks_launch_ms) if dominant_is_ks else (tn_alg, tn_call_ms)
    achieved = dom_alg / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
    # SURVEY 8d bytes per op: multiply 18.4 MB (operands 2 x 5 MB read + product 7.5 MB written, rounded as the survey does) + relinearize 70.8 MB with
    # the keys per op, 13.1 MB with the keys once per launch
    per_gpu = value / world
    # counter bytes of the dominant launch (tools/profile_cfg4.sh: separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as the guide prescribes; sha-locked);
    # tensor_core_kernel is two launches per multiply (base q and base Bsk): the region brackets both, so both records are added
    rec4 = counters_record("cfg4_ksmac", nb) if dominant_is_ks else counters_record("cfg4_tensor_bsk", nb)
    traffic4 = rec4.get("traffic_bytes_per_launch") if rec4 else None
    if rec4 and not dominant_is_ks:
        rq = counters_record("cfg4_tensor_q", nb)
        traffic4 = (traffic4 + rq["traffic_bytes_per_launch"]) if (traffic4 and rq and rq.get("traffic_bytes_per_launch")) else None
    for tag, key in (("cfg4_ksmac", "ksmac2_kernel<15>"), ("cfg4_tensor_bsk", "tensor_core_kernel"), ("cfg4_tensor_q", "tensor_core_kernel")):
        r_ = counters_record(tag, nb)
        for kname, kd in kernels.items():
            if r_ and kname.startswith(key) and r_.get("traffic_bytes_per_launch"):
                kd["traffic_" + tag.split("_", 1)[1]] = r_["traffic_bytes_per_launch"]
    # `bound` = the contract's HBM figures of the dominant launch, timed in this run by the library's kernel timer; `limiter` = what binds it (FP64 butterflies
    # in ksmac2 / tensor_core_kernel, integer multiply-accumulates in the base conversions)
    roofline = {"bound": "hbm", "limiter": "valu_fp64" if dominant_is_ks else "valu_int",
                "kernel": "ksmac2_kernel<15>" if dominant_is_ks else "tensor_core_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic4, "traffic_record": rec4["_file"] if rec4 else None,
                "launch_ms": round(dom_ms, 4), "launches_timed": ks_n if dominant_is_ks else tn_n,
                "algorithmic_bytes_per_launch": dom_alg, "kernels": kernels,
                "valu_int": cfg4_valu_block(nb, n, L, S),
                "pipeline": {"bytes_per_op_keys_per_op": 89.2e6, "frac_keys_per_op": round(per_gpu * 89.2e6 / (HBM_PEAK_GBS * 1e9), 4),
                             "bytes_per_op_keys_once_per_launch": 18.4e6 + 13.1e6 + 57.7e6 / nb,
                             "frac_keys_once_per_launch": round(per_gpu * (18.4e6 + 13.1e6 + 57.7e6 / nb) / (HBM_PEAK_GBS * 1e9), 4)}}
    result = {
        "metric": "homomorphic mul+relinearize ops/sec (BFV BEHZ multiply + relinearize), N=32768", "value": round(value, 1), "unit": "ops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "BFV N=32768, 11x50-bit coeff modulus (K=11, L=10), t=1032193: %d independent multiply + relinearize ops per step, "
                               "block-partitioned over %d rank(s) (rank 0: items [%d, %d)), %d per launch" % (args.total, world, lo, hi, nb),
                   "total_ops_per_step": args.total, "operands": "%d distinct ciphertext pairs per rank resident in HBM" % mine,
                   "parallelism": "batch-sharded x%d (shard.shard_range), keys broadcast once (RCCL)" % world, **ranks},
        "roofline": roofline,
    }
    if rank == 0 and not args.no_cpu_baseline:
        import numpy as np
        O = entry.load_oracle()
        ctx = O.Context("bfv", n, q, t_plain)
        hk = [pkg.to_host(k) for k in keys]

        def one_op(ha, hb):
            return ctx.relinearize(L, False, ctx.bfv_multiply(L, ha, hb), hk)

        base = ((mine - 1) // nb) * nb                    # `out` holds the rank's LAST chunk: items base .. mine - 1
        items = sorted({0, min(1, mine - base - 1), mine - base - 1})      # (the last chunk may be ragged)
        ops, t0 = 0, time.perf_counter()
        for i in items:
            exp = one_op(pkg.to_host(x[base + i]), pkg.to_host(y[base + i]))
            ops += 1
            if not np.array_equal(pkg.to_host(out[i]), exp):
                raise AssertionError("bench cfg4: GPU result of item %d differs from the CPU oracle" % (base + i))
        ha, hb = pkg.to_host(x[0]), pkg.to_host(y[0])
        while time.perf_counter() - t0 < args.cpu_seconds and world == 1:
            one_op(ha, hb)
            ops += 1
        t_cpu = time.perf_counter() - t0
        result["parity"] = "bit-exact vs CPU oracle (items %s of the job: %s of its last chunk of %d)" % ([base + i for i in items], items, mine - base)
        result["cpu_baseline"] = {"value": round(ops / t_cpu, 3), "unit": "ops/s", "cores": 1, "kind": "port",
                                  "sample": "%d sequential BEHZ multiply + relinearize ops (items %s of the same workload, then item 0 repeated; %.1f s, "
                                            "oracle/troy_oracle.c, gcc -O3, 1 thread of %d host cores)" % (ops, items, t_cpu, os.cpu_count())}
    del x, y, prod, out, keys, behz, plan
    return result


def run_cfg5(args, torch, pkg, entry, device):
    """BASELINE configs[4]: BFV 512x512x512 matmul (examples/10_bfv_matmul.cu at that size), N=8192 {60,40,40,60}, t = 2^21, packed outputs.
    value = end-to-end latency encrypt -> matmul -> mod-switch -> pack -> add bias -> decrypt through MatmulHelper (tests/cpp/matmul_driver, a
    child process); roofline = the ct x pt multiply-accumulate launch of the same block shape, timed here by the library's kernel timer;
    cpu_baseline = the oracle's restatement of the reference's host branches for the same flow, per-object timings on a bounded sample multiplied out."""
    import ctypes as C
    res = {"metric": "BFV 512x512x512 packed matmul end-to-end latency (encrypt -> decrypt), N=8192", "unit": "ms", "higher_is_better": False, "data": "synthetic",
           "dtype": "u64 ({60,40,40,60}: 60-bit limb on integer butterflies, 40-bit limbs on exact-FP64 butterflies)"}
    drv = os.path.join(ROOT, "tests", "cpp", "matmul_driver")
    if os.path.exists(drv):
        r = subprocess.run([drv, "512", "512", "512", "20", "1", "1"], capture_output=True, text=True, timeout=900)
        lines = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
        if r.returncode == 0 and "ms" in lines:
            ms = dict(zip(lines["ms"][0::2], [float(v) for v in lines["ms"][1::2]]))
            rep = dict(zip(lines["ms_repeat"][0::2], [float(v) for v in lines["ms_repeat"][1::2]]))
            first = ms["encrypt_inputs"] + ms["matmul_first"] + ms["mod_switch"] + ms["pack"] + ms["add_bias"] + ms["decrypt"]
            # every phase at its steady state (second and later passes: buffers in the pool, the level's one-off constants built); first_call_latency_ms = the first pass
            steady = rep["encrypt_inputs"] + ms["matmul_repeat"] + rep.get("mod_switch", ms["mod_switch"]) + rep.get("pack", ms["pack"]) + rep.get("add_bias", ms["add_bias"]) + rep["decrypt"]
            # everything the example does (examples/10_bfv_matmul.cu:96-120: the encodings and the wire phases included), timed on COMPLETE passes of the flow
            # after a warm-up pass (ms_steady: no hipMalloc inside a phase, the caller keeps its wire buffers); first_pass_ms = the same phases on the cold pool
            flow = dict(zip(lines["ms_steady"][0::2], [float(v) for v in lines["ms_steady"][1::2]])) if "ms_steady" in lines else None
            tot = dict(zip(lines["ms_steady_total"][0::2], [float(v) for v in lines["ms_steady_total"][1::2]])) if "ms_steady_total" in lines else {}
            cold = (ms["encode_weights"] + ms["encode_bias"] + ms["encrypt_inputs"] + ms["inputs_wire"] + ms["matmul_first"] + ms["mod_switch"] +
                    ms["pack"] + ms["add_bias"] + ms["outputs_wire"] + ms["decrypt"])
            res.update({"value": round(steady, 3), "end_to_end_ms": round(sum(flow.values()), 3) if flow else None,
                        "end_to_end_phases_ms": flow, "end_to_end_passes": tot, "first_pass_ms": round(cold, 3),
                        "config": {"workload": "y = x*w + s, 512x512x512 over Z_{2^21}, MatmulHelper block %s, %s; encrypted inputs x plaintext weights, mod-switched and "
                                               "LWE-packed outputs" % ("x".join(lines["block"][:3]), " ".join(lines["objects"])),
                                   "value_phases": "encrypt_inputs + matmul + mod_switch + pack + add_bias + decrypt, steady state",
                                   "first_pass_phases_ms": ms, "steady_state_ms": rep, "first_call_latency_ms": round(first, 3)},
                        "parity": "all 262144 outputs equal the plain product mod 2^21; on-the-fly weight encoding gives word-identical ciphertexts" if "OK" in r.stdout else "FAILED"})
        else:
            res["error"] = (r.stdout + r.stderr)[-500:]
    else:
        res["error"] = "tests/cpp/matmul_driver is not built"

    # ---- roofline of the dominant launch: the ct x pt multiply-accumulate over the same block shape (32 x 512 weight plaintexts, one input block row) ----
    n, I, J, Bt, L = 8192, 32, 512, 1, 3
    q = pkg.capi.coeff_modulus_create(n, [60, 40, 40, 60])
    plan = pkg.Plan(device, 13, q)
    gen = torch.Generator(device=device).manual_seed(11)
    av = uniform_residues(torch, (Bt, I, 2), q[:L], n, device, gen)
    w = torch.empty((I, J, L, n), dtype=torch.int64, device=device)
    for l, m in enumerate(q[:L]):
        w[:, :, l, :].random_(0, m, generator=gen)
    out = torch.empty((Bt, J, 2, L, n), dtype=torch.int64, device=device)
    cts, pts, dsts = [], [], []
    for i in range(I):
        for j in range(J):
            for b in range(Bt):
                cts.append(av[b, i].data_ptr()); pts.append(w[i, j].data_ptr()); dsts.append(out[b, j].data_ptr())
    terms = len(cts)
    arr = lambda v: (C.c_void_p * terms)(*v)
    ca, pa, da = arr(cts), arr(pts), arr(dsts)
    nbytes = plan.lib.troyn_multiply_plain_accumulate_workspace_bytes(terms)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def mac():
        pkg.capi.check(plan.lib.troyn_multiply_plain_accumulate(plan.h, 0, L, 2, ca, pa, da, terms, 1, C.c_void_p(ws.data_ptr()), ws.numel(), stream))
    with KernelTimer(pkg, plan.lib, TIMER_PLAIN_MAC) as kt:
        t_call = timed(torch, mac, 10)
        k_ms, k_n = kt.read()
    launch_ms = k_ms / max(1, k_n)
    # each operand once: every weight plaintext (L limbs), every input ciphertext (2 L limbs) read once, every output ciphertext written once
    alg = terms * L * n * 8.0 + Bt * I * 2 * L * n * 8.0 + Bt * J * 2 * L * n * 8.0
    achieved = alg / (launch_ms * 1e-3) / 1e9 if k_n else 0.0
    res["roofline"] = {"bound": "hbm", "kernel": "plain_mac2_kernel<2> (multiply_plain_ntt_accumulate over %d terms: %d x %d weight plaintexts, %d input block row(s))" % (terms, I, J, Bt),
                       "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                       "launch_ms": round(launch_ms, 4), "launches_timed": k_n, "call_ms_with_pointer_table_upload": round(t_call * 1e3, 4),
                       "algorithmic_bytes_per_launch": alg}
    # parity of that launch: two destinations against the oracle's multiply_plain_ntt + add
    if not args.no_cpu_baseline:
        import numpy as np
        O = entry.load_oracle()
        ctx = O.Context("bfv", n, q, 1 << 21)
        mac()
        torch.cuda.synchronize()
        qa = np.array(q[:L], dtype=np.uint64)[None, :, None]
        for j in (0, J - 1):
            acc = np.zeros((2, L, n), dtype=np.uint64)
            for i in range(I):
                acc = (acc + ctx.multiply_plain_ntt(L, pkg.to_host(av[0, i]), pkg.to_host(w[i, j]))) % qa
            if not np.array_equal(pkg.to_host(out[0, j]), acc):
                raise AssertionError("bench cfg5: multiply-accumulate destination %d differs from the CPU oracle" % j)
        res["roofline"]["parity"] = "destinations 0 and %d of the timed launch bit-exact vs the CPU oracle" % (J - 1)
        # ---- CPU baseline of the packed flow: the oracle, one host thread, per-object timings multiplied out ----
        # the matmul core is timed IN FULL: all 32 x 512 = 16 384 multiply_plain_ntt + add terms (the same operands per term: the oracle's arithmetic
        # does not depend on the values), ~3 s of one host thread
        ct, pt = ctx.random_ct(1, 2, 3), ctx.random_ct(2, 1, 3)[0]
        acc = np.zeros_like(ct)
        t0 = time.perf_counter()
        for _ in range(terms):
            term = ctx.multiply_plain_ntt(3, ct, pt)
            acc = (acc + term) % qa
        t_core = time.perf_counter() - t0
        rng = O.Rng(3)
        sk = ctx.secret_key(rng)
        pk = ctx.public_key(rng, sk)
        msg = O.fill_uniform(9, 1 << 21, 8192)

        def cpu_timed(fn, reps):
            t0 = time.perf_counter()
            for _ in range(reps):
                o = fn()
            return (time.perf_counter() - t0) / reps, o
        t_enc, c = cpu_timed(lambda: ctx.encrypt_asymmetric_bfv(rng, pk, msg), 3)
        t_ntt, _ = cpu_timed(lambda: ctx.from_ntt(ctx.to_ntt(c, 2, 3), 2, 3), 3)            # one forward + one inverse transform of a ciphertext
        t_dec, _ = cpu_timed(lambda: ctx.decrypt_bfv(sk, c), 3)
        t_ms, low = cpu_timed(lambda: ctx.mod_switch_scale_to_next(3, c), 3)
        gkeys = {(8192 // 16) * (1 << (k + 1)) + 1: ctx.random_keys(50 + k, 2) for k in range(4)}
        t_pack, _ = cpu_timed(lambda: ctx.pack_rlwe_ciphertexts(2, [low] * 16, gkeys, 2 * 8192 - 15, 16, 1), 1)
        phases = {"encrypt_32_inputs": t_enc * 32, "ntt_32_inputs_intt_512_outputs": t_ntt / 2 * (32 + 512), "matmul_core_16384_terms_timed_in_full": t_core,
                  "mod_switch_512_outputs": t_ms * 512, "pack_32_groups_of_16": t_pack * 32, "decrypt_32_outputs": t_dec * 32}
        res["cpu_baseline"] = {"value": round(sum(phases.values()) * 1e3, 1), "unit": "ms", "cores": 1, "kind": "port",
                               "sample": "oracle (oracle/troy_oracle.c, gcc -O3, one host thread of %d cores): the matmul core in full (all %d multiply_plain_ntt + add "
                                         "terms, %.1f s); 3 encryptions / transform pairs / decryptions / modulus switches and one packing tree of 16, each multiplied out "
                                         "to the flow's object counts (src/app/matmul.cu:326-374 matmul + examples/10_bfv_matmul.cu)" % (os.cpu_count(), terms, t_core),
                               "composition": "matmul core measured; the remaining phases are per-object timings x object counts (an estimate, stated as such)",
                               "phases_ms": {k: round(v * 1e3, 1) for k, v in phases.items()}}
    del av, w, out, ws, plan
    torch.cuda.empty_cache()
    return res


def run_cpp_api():
    """BASELINE config 3 through the C++ mirror of the reference's API (troy::Evaluator), timed by the troybench-shaped driver in a child process"""
    drv = os.path.join(ROOT, "tests", "cpp", "he_bench_driver")
    if not os.path.exists(drv):
        return {"error": "tests/cpp/he_bench_driver is not built"}
    r = subprocess.run([drv, "bench", "10"], capture_output=True, text=True, timeout=900)
    kv = {ln.split()[0]: ln.split()[1] for ln in r.stdout.splitlines() if len(ln.split()) == 2}
    if r.returncode != 0 or "OK" not in r.stdout:
        return {"error": (r.stdout + r.stderr)[-500:]}
    out = {"what": "CKKS N=16384 6x50-bit through troy::Evaluator (tests/cpp/he_bench_driver.cpp, shaped like the reference's test/bench/he_operations.cu): "
                   "three_calls = multiply + relinearize + rescale_to_next (single objects: *_new with a stream synchronisation after every call as the tool does; "
                   "batches: *_batched), fused = Evaluator::multiply_relinearize_rescale{_new,_batched}; threads = host threads with their own operands; "
                   "single_threadsN_* = the tool's -c N mode (N threads x single objects, one stream wait per op); *_combined = the same loops with call combining on "
                   "(troy::combining, troy.h: one shared stream, the calls of concurrent threads run as one batched launch sequence; results bit-identical, see combined_identical)",
           "fused_identical_to_three_calls": all(kv.get(k) == "1" for k in ("fused_single_identical", "fused_inplace_identical", "fused_batched_identical", "fused_mixed_levels_identical")),
           "combined_identical": kv.get("combined_identical") == "1" and kv.get("combined_alone_identical") == "1"}
    for k, v in kv.items():
        if k.startswith(("single_", "batched_")):
            out[k] = float(v)
    # the reference tool's -c N -mp -md mode: thread i in MemoryPool::create(i % device_count()), one context per device, the same secret key everywhere
    # (on a one-GPU box: 8 pools on device 0); per-device and aggregate ops/s of single-object and batch-64 calls
    r2 = subprocess.run([drv, "devices", "8", "10"], capture_output=True, text=True, timeout=900)
    kv2 = {ln.split()[0]: ln.split()[1] for ln in r2.stdout.splitlines() if len(ln.split()) == 2}
    if r2.returncode == 0 and "OK" in r2.stdout:
        out["multi_device_mode"] = {"what": "he_bench_driver devices 8: 8 host threads, pool i on device i % devices, contexts = min(devices, 8) sharing one secret key; "
                                            "ops/s summed over threads (their timed loops start together)",
                                    "devices": int(kv2.get("devices", 0)), "pools": int(kv2.get("pools", 0)), "contexts": int(kv2.get("contexts", 0)),
                                    "same_secret_key": kv2.get("devices_same_secret_key") == "1", "identical": kv2.get("devices_identical") == "1"}
        for k, v in kv2.items():
            if k.startswith("devices_") and k.endswith("_ops_per_s"):
                out["multi_device_mode"][k[len("devices_"):]] = float(v)
    else:
        out["multi_device_mode"] = {"error": (r2.stdout + r2.stderr)[-300:]}
    return out


def dry_run(args, rank, world):
    import torch
    import importlib
    import __graft_entry__ as entry
    if rank == args.dry_fail_rank:
        sys.stderr.write("bench.py --dry-run: rank %d fails on purpose\n" % rank)
        return 7
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    keys = [torch.full((2, 3, 8), 100 + j if rank == 0 else -1, dtype=torch.int64) for j in range(2)]
    shard.broadcast_tensors(keys, src=0)
    ok = all(int(k[0, 0, 0]) == 100 + j for j, k in enumerate(keys))
    lo, hi = shard.shard_range(args.total, rank, world)
    shard.barrier()
    ranks = shard.rank_report(0.001 * (rank + 1), 1, "cpu:%d" % rank, 0.0005 * (rank + 1))
    elapsed = shard.max_over_ranks(0.001 * (rank + 1))
    covered = shard.sum_over_ranks(hi - lo)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "keys_broadcast_ok": ok, "max_elapsed": elapsed, "items_covered": covered,
                          "rank0_range": [lo, hi], "scaling": "strong" if args.workload == "cfg4" else "weak", "config": ranks}))
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0 if ok else 5


def main():
    args = parse_args()
    if args.dry_shard:
        print(json.dumps({"total": args.total, "world": args.gpus, "ranges": shard_plan(args.total, args.gpus)}))
        return 0
    # TROYN_BENCH_OVERSUBSCRIBE=1 (tests/test_bench_launcher.py, a one-GPU box): the N ranks share the visible device(s) and talk over gloo -- RCCL refuses two
    # ranks on one device.  It exercises the REAL N > 1 path (launcher, rendezvous, key broadcast, sharding, evaluation on the GPU, reductions, the line) where
    # no multi-GPU node is to be had; its timings mean nothing and the line says so (config.oversubscribed).
    oversub = os.environ.get("TROYN_BENCH_OVERSUBSCRIBE", "") not in ("", "0")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("TROYN_BENCH_SPAWN")):
        if not args.dry_run:
            import torch      # device_count() does not initialise HIP on this image: the parent still never touches the GPU
            visible = torch.cuda.device_count()
            if args.gpus > visible and not (oversub and visible >= 1):
                sys.stderr.write("bench.py: --gpus %d but only %d device(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
                                 "nothing was started\n" % (args.gpus, visible))
                return 2
        return launch_ranks(args.gpus)         # nothing above touched the GPU (TROYN_BENCH_SPAWN: take this path at N = 1 too)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    if args.workload == "cfg4" and args.steps == 20:
        args.steps = 5                          # one step = the whole 1024-op job
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import __graft_entry__ as entry
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X")
    if oversub:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)             # before the process group: RCCL binds its communicator to the current device
    device = torch.device("cuda", local_rank)
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if oversub else "nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    pkg = entry.load_package()
    import importlib
    shard = importlib.import_module("troy_nova_amd.shard")
    run = run_cfg3 if args.workload == "cfg3" else run_cfg4
    result = run(args, torch, pkg, shard, entry, rank, world, device)
    if oversub and world > 1:
        result["config"]["oversubscribed"] = "TEST MODE: %d ranks on %d visible device(s), collectives over gloo -- not a measurement" % (world, torch.cuda.device_count())
    if rank == 0:
        emit(result)
    if world > 1:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
