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
        dist.broadcast(t, src=src)
    return tensors


def max_over_ranks(value, device="cpu"):
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if is_distributed():
        dist.barrier()
