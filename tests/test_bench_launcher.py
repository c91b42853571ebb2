"""bench.py's multi-GPU entry on CPU: the partition it would use, the self-launcher (`--gpus N` without torchrun
starts N rank processes and relays rank 0's line) and its refusal of a world size that is not the one asked for."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def test_dry_shard_partitions_the_job():
    r = _run(["--gpus", "2", "--dry-shard", "--workload", "cfg4"])
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["ranges"] == [[0, 512], [512, 1024]]
    r = _run(["--gpus", "8", "--dry-shard", "--total", "1001"])
    parts = json.loads(r.stdout.strip().splitlines()[-1])["ranges"]
    assert parts[0][0] == 0 and parts[-1][1] == 1001 and all(parts[i][1] == parts[i + 1][0] for i in range(7))


def test_self_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--dry-run", "--workload", "cfg4"])
    assert r.returncode == 0, r.stderr
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["keys_broadcast_ok"] and j["items_covered"] == 1024 and j["rank0_range"] == [0, 512]
    assert j["max_elapsed"] == 0.002 and j["scaling"] == "strong"
    # what makes a first multi-GPU run diagnosable: the world size the BACKEND reports, every rank's own step time, the slowest rank,
    # the key broadcast time and what each rank ran on
    c = j["config"]
    assert c["ranks"] == 2 and c["ms_per_step_per_rank"] == [1.0, 2.0] and c["ms_per_step_min"] == 1.0 and c["ms_per_step_max"] == 2.0
    assert c["slowest_rank"] == 1 and c["devices"] == ["cpu:0", "cpu:1"]
    assert c["key_broadcast_s"] == 0.001 and c["key_broadcast_s_per_rank"] == [0.0005, 0.001]


def test_more_ranks_than_devices_is_refused_before_spawning():
    """no GPU in this container: any real (non --dry-run) multi-rank job asks for more devices than are visible"""
    import torch
    want = torch.cuda.device_count() + 1
    r = _run(["--gpus", str(max(want, 2))])
    assert r.returncode == 2 and "visible" in r.stderr and "nothing was started" in r.stderr, r.stderr


def test_launcher_watchdog_ends_the_job_when_a_rank_dies():
    """rank 1 exits before the rendezvous: the launcher terminates rank 0 (which waits in init_process_group) and reports the failure
    within seconds instead of the store / collective timeout"""
    import time
    t0 = time.time()
    r = _run(["--gpus", "2", "--dry-run", "--dry-fail-rank", "1"])
    assert r.returncode == 1 and "rank(s) failed" in r.stderr and "(1, 7)" in r.stderr, r.stderr
    assert "rank 1 fails on purpose" in r.stderr          # per-rank stderr attribution
    assert time.time() - t0 < 60


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4", "--dry-run"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


import pytest


@pytest.mark.gpu
def test_self_launcher_one_gpu():
    """the launcher path on real hardware: a fresh rank process is started before anything touches the GPU"""
    r = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--inner", "1", "--batch", "64", "--no-cpu-baseline", "--no-extra"], env_extra={"TROYN_BENCH_SPAWN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["roofline"]["launches_timed"] == 3


@pytest.mark.gpu
@pytest.mark.parametrize("workload,extra", [("cfg3", ["--batch", "16", "--inner", "1"]), ("cfg4", ["--total", "6", "--batch", "2"])])
def test_two_ranks_on_hardware_oversubscribed(workload, extra):
    """SURVEY 8e on a one-GPU box: TWO rank processes through the self-launcher, both on the visible device, collectives over gloo (RCCL refuses a duplicate
    device).  Everything but the transport is the real N > 1 path: fresh rank processes, rendezvous, key broadcast from rank 0, the block partition, evaluation on
    the GPU in both ranks at once, max-over-ranks, the per-rank block of the line -- and rank 0's results equal the oracle's.  (Timings mean nothing here.)"""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", workload, "--no-extra", "--cpu-seconds", "0"] + extra,
             env_extra={"TROYN_BENCH_OVERSUBSCRIBE": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    c = j["config"]
    assert j["n_gpus"] == 2 and c["ranks"] == 2 and len(c["ms_per_step_per_rank"]) == 2 and len(c["devices"]) == 2 and "oversubscribed" in c
    assert j["value"] > 0 and j["scaling"] == ("weak" if workload == "cfg3" else "strong")
    assert "bit-exact" in j.get("parity", ""), j.get("parity")
    if workload == "cfg4":
        assert c["items_per_rank"] == [3, 3] and c["total_ops_per_step"] == 6
    else:
        assert c["ops_per_step"] == 2 * 16
