"""N > 1 path on CPU: batch sharding + key broadcast + max-over-ranks with gloo, world_size 2.

What this covers is the PLUMBING of the multi-GPU path (troy-nova_amd/shard.py as bench.py uses it): the block partition, the one-time
key broadcast, the timing / count reductions and the barrier.  The product has no CPU path by design
(tests/test_capi_symbols.py::test_no_cpu_path), so the kernels cannot run here: test_two_rank_gloo uses a stand-in evaluation
(`torch.arange`), and test_sharded_job_equals_the_whole_job lets every rank evaluate its slice of a real CKKS multiply + relinearize +
rescale job with the CPU oracle (the checker; small ring) -- the union of the slices must be the unsharded job, item for item, with the
keys every rank used being the broadcast ones.  On a GPU box every rank runs exactly the single-GPU path on its slice, which the
`-m gpu` suite covers, and the data path has no collective to test."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # keys are produced on rank 0 only and broadcast once
    keys = [torch.full((2, 3, 8), 100 + j if rank == 0 else -1, dtype=torch.int64) for j in range(2)]
    shard.broadcast_tensors(keys, src=0)
    ok_keys = all(int(k[0, 0, 0]) == 100 + j for j, k in enumerate(keys))
    lo, hi = shard.shard_range(total, rank, world)
    # stand-in "evaluation": every rank handles its own slice of the batch with no exchange
    mine = torch.arange(lo, hi, dtype=torch.int64) * 2
    t = shard.max_over_ranks(1.0 + rank)
    n = shard.sum_over_ranks(hi - lo)
    shard.barrier()
    ret[rank] = (lo, hi, ok_keys, t, n, mine.tolist())
    dist.destroy_process_group()


def test_shard_range_partition():
    sys.path.insert(0, ROOT)
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 4, 8):
            parts = [shard.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert shard.max_over_ranks(3.5) == 3.5 and not shard.is_distributed()


def test_two_rank_gloo():
    world, total = 2, 11
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, total, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        r0, r1 = ret[0], ret[1]
    assert (r0[0], r0[1]) == (0, 6) and (r1[0], r1[1]) == (6, 11)
    assert r0[2] and r1[2], "keys must be identical on every rank after the broadcast"
    assert r0[3] == 2.0 and r1[3] == 2.0, "timing is the max over ranks"
    assert r0[4] == total and r1[4] == total
    assert r0[5] + r1[5] == [2 * i for i in range(total)]


def _oracle_job_item(O, ctx, L, n, q, item, keys):
    """item `item` of the job: its operands depend on the item index only (the same payload on every world size, as bench.py's cfg4 job), evaluated by the oracle"""
    import numpy as np
    a = np.stack([np.stack([O.fill_uniform(1000 * item + 10 * p + l, q[l], n) for l in range(L)]) for p in range(2)])
    b = np.stack([np.stack([O.fill_uniform(1000 * item + 500 + 10 * p + l, q[l], n) for l in range(L)]) for p in range(2)])
    return ctx.mod_switch_scale_to_next(L, ctx.relinearize(L, True, ctx.ckks_multiply(L, a, b), keys))


def _oracle_worker(rank, world, port, total, ret):
    sys.path.insert(0, ROOT)
    import importlib
    import numpy as np
    import __graft_entry__ as entry
    entry.load_package()
    shard = importlib.import_module("troy_nova_amd.shard")
    O = entry.load_oracle()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, L = 64, 2
    q = [int(v) for v in O.coeff_modulus_create(n, [30, 30, 30])]
    ctx = O.Context("ckks", n, q)
    # evaluation keys exist on rank 0 only; the others hold garbage until the one-time broadcast
    made = ctx.random_keys(77, L) if rank == 0 else [np.full((2, len(q), n), 12345, dtype=np.uint64) for _ in range(L)]
    keys_t = [torch.from_numpy(k.astype(np.int64)) for k in made]
    shard.broadcast_tensors(keys_t, src=0)
    keys = [k.numpy().astype(np.uint64) for k in keys_t]
    lo, hi = shard.shard_range(total, rank, world)
    mine = [_oracle_job_item(O, ctx, L, n, q, i, keys) for i in range(lo, hi)]
    covered = shard.sum_over_ranks(hi - lo)
    gathered = shard.gather_objects((lo, hi, [m.tobytes() for m in mine]))
    shard.barrier()
    if rank == 0:
        whole_keys = ctx.random_keys(77, L)
        whole = [_oracle_job_item(O, ctx, L, n, q, i, whole_keys).tobytes() for i in range(total)]
        joined = [blob for (_, _, blobs) in gathered for blob in blobs]
        ret["ok"] = joined == whole and covered == total and [g[0] for g in gathered] == [shard.shard_range(total, r, world)[0] for r in range(world)]
        ret["distinct"] = len(set(whole)) == total
    dist.destroy_process_group()


def test_sharded_job_equals_the_whole_job():
    world, total = 2, 5
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_oracle_worker, args=(r, world, port, total, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        assert ret["ok"], "the union of the ranks' slices (evaluated with the broadcast keys) must equal the unsharded job"
        assert ret["distinct"], "items of the job must differ from each other (or a permutation would go unnoticed)"
