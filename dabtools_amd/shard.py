"""Multi-GPU sharding of independent ensembles.

Every DAB ensemble is decoded independently end to end, so the batch shards by stream with
no data-path collective: global stream s runs on rank s // streams_per_gpu.  The only
cross-rank traffic is the bench bookkeeping (a barrier, MAX of the elapsed time, SUM of the
ETI frame counts): bench.py carries it over its own pipes or a gloo group; the helpers below
do the same over whatever torch.distributed group is initialised (gloo in the CPU tests).
torch is imported only by those helpers: the sharding rules themselves need nothing.
"""


def shard_streams(total_streams, world, rank):
    """Global stream indices owned by `rank`: contiguous blocks, remainders to the low ranks."""
    base, rem = divmod(total_streams, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def stream_seed(config_id, global_stream):
    """Seed rule of SURVEY.md 8(d): seed = 1000 * config + stream."""
    return 1000 * config_id + global_stream


def barrier(device=None):
    import torch
    import torch.distributed as dist
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def aggregate(elapsed_s, frames, device=None):
    """(max elapsed over ranks, total frames over ranks)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), int(frames)
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    f = torch.tensor([frames], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(f.item())
