"""N > 1 path of bench.py on CPU: two gloo ranks shard a global batch with no data-path collective and
agree on one table blob through a broadcast from rank 0 (the same torch.distributed calls that run over
RCCL/xGMI on the GPU box)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    shards = bench.shard_batches(2048, world)
    lo, hi = shards[rank]
    # every rank builds its own "tables"; rank 0's copy wins
    blob = torch.full((1000,), rank + 1, dtype=torch.uint8)
    n = torch.tensor([blob.numel()], dtype=torch.int64)
    dist.broadcast(n, src=0)
    dist.broadcast(blob, src=0)
    # throughput accounting: images are summed host-side, time is the max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out.put((rank, lo, hi, int(blob[0]), int(n[0]), float(t[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_table_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert [(r[1], r[2]) for r in res] == [(0, 1024), (1024, 2048)]      # disjoint, covering shards
    assert all(r[3] == 1 and r[4] == 1000 for r in res)                   # every rank holds rank 0's blob
    assert all(r[5] == 2.0 for r in res)                                  # elapsed = max over ranks


def test_shard_map_properties():
    import bench
    for world in (1, 2, 4, 8):
        s = bench.shard_batches(1024 * world, world)
        assert len(s) == world and s[0][0] == 0 and s[-1][1] == 1024 * world
        assert all(a[1] == b[0] for a, b in zip(s, s[1:])) and all(hi - lo == 1024 for lo, hi in s)
