"""N > 1 path on CPU: two processes over gloo exercise the sharding and the cross-rank
bookkeeping bench.py uses (the decode itself needs a GPU; ensembles never communicate)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from dabtools_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_streams(515, world, rank)
    seeds = [shard.stream_seed(2, s) for s in mine]
    shard.barrier()
    # pretend each stream yielded 196 ETI frames and rank r took (r + 1) seconds
    elapsed, frames = shard.aggregate(float(rank + 1), 196 * len(mine))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    q.put((rank, mine, seeds, elapsed, frames, gathered))
    dist.destroy_process_group()


def test_two_rank_sharding_and_aggregation():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    all_streams = []
    for rank, mine, seeds, elapsed, frames, gathered in res:
        assert elapsed == 2.0                      # MAX over ranks
        assert frames == 196 * 515                 # SUM over ranks
        assert seeds == [2000 + s for s in mine]
        assert gathered == [r[1] for r in res]
        all_streams += mine
    assert sorted(all_streams) == list(range(515))  # disjoint cover
    assert abs(len(res[0][1]) - len(res[1][1])) <= 1


def test_single_process_paths():
    sys.path.insert(0, ROOT)
    from dabtools_amd import shard
    assert shard.shard_streams(256, 1, 0) == list(range(256))
    assert [len(shard.shard_streams(2048, 8, r)) for r in range(8)] == [256] * 8
    assert shard.aggregate(1.5, 7) == (1.5, 7)
    shard.barrier(torch.device("cpu"))
