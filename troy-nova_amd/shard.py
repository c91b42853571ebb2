"""Batch sharding across the GPUs of one node.

Independent ciphertext evaluations are block-partitioned over ranks (one process per GPU).  The data
path needs no collective; the only exchange is the one-time broadcast of evaluation keys from rank 0
(RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU tests) and the max-over-ranks of the
timing.  This mirrors the reference's model of one pool + context per device holding the SAME keys
(readme.md:179-202, test/test_multithread.cu:18-37).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """block partition of [0, total): ranks < total % world get one extra item"""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def broadcast_tensors(tensors, src=0):
    """one-time key / table broadcast; tensors are modified in place on non-src ranks"""
    if not is_distributed():
        return tensors
    for t in tensors:
        if dist.get_backend() == "gloo" and t.is_cuda:
            # gloo moves host memory: stage through a CPU copy (test modes only; the product path is nccl = RCCL, device to device over xGMI)
            h = t.cpu()
            dist.broadcast(h, src=src)
            if dist.get_rank() != src:
                t.copy_(h)
        else:
            dist.broadcast(t, src=src)
    return tensors


def _coll_device(device):
    """where the small reduction tensors live: the rank's GPU under nccl (= RCCL), the host under gloo (CPU tests; the oversubscribed hardware test of
    bench.py, where several ranks share one GPU and RCCL would refuse the duplicate device)"""
    return "cpu" if dist.get_backend() == "gloo" else device


def max_over_ranks(value, device="cpu"):
    if not is_distributed():
        return float(value)
    device = _coll_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not is_distributed():
        return float(value)
    device = _coll_device(device)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """every rank's value, in rank order (the first multi-GPU run must be diagnosable: which rank was slow, by how much)"""
    if not is_distributed():
        return [float(value)]
    device = _coll_device(device)
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def gather_objects(obj):
    """small python objects (device names) from every rank, in rank order"""
    if not is_distributed():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def backend_world_size():
    """the world size the BACKEND reports (not the one the launcher asked for)"""
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank_report(elapsed_s, steps, device_name, key_broadcast_s, device="cpu"):
    """the per-rank part of a bench line: spread of the step time over the ranks, who was slowest, what every rank ran on"""
    per = gather_over_ranks(elapsed_s / max(1, steps) * 1e3, device=device)
    names = gather_objects(device_name)
    bc = gather_over_ranks(key_broadcast_s, device=device)
    slow = max(range(len(per)), key=lambda r: per[r])
    return {"ranks": backend_world_size(), "ms_per_step_per_rank": [round(v, 4) for v in per], "ms_per_step_min": round(min(per), 4),
            "ms_per_step_max": round(max(per), 4), "slowest_rank": slow, "devices": names,
            "key_broadcast_s": round(max(bc), 4), "key_broadcast_s_per_rank": [round(v, 4) for v in bc]}


def barrier():
    if is_distributed():
        dist.barrier()
