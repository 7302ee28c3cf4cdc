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


def cpu_budget():
    """(CPUs in this process's affinity mask, CFS quota of its cgroup in CPUs or None): what a container really grants -- os.cpu_count() is the
    machine's thread count (256 on the GPU boxes of this pool, whose containers get 16 CPUs' worth of time).  The same rule as the library's
    usable_cpus() (csrc/placement.hpp); pure Python so that a launcher can use it before anything touches the GPU."""
    import os
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    quota = None

    def take(q):
        nonlocal quota
        if q and q > 0 and (quota is None or q < quota):
            quota = q

    rel = ""
    try:
        for line in open("/proc/self/cgroup").read().splitlines():
            if line.startswith("0::"):
                rel = line[3:].rstrip("/")
    except OSError:
        pass
    path = "/sys/fs/cgroup" + rel
    while True:
        try:
            q, per = open(path + "/cpu.max").read().split()[:2]
            if q != "max":
                take(-(-int(q) // int(per)))
        except (OSError, ValueError):
            pass
        if len(path) <= len("/sys/fs/cgroup"):
            break
        path = path.rsplit("/", 1)[0]
    if quota is None:
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                take(-(-q // per))
        except (OSError, ValueError):
            pass
    return affinity, quota


def usable_cpus():
    """min(affinity mask, CFS quota), at least 1; DABHIP_CPUS=n overrides it."""
    import os
    v = os.environ.get("DABHIP_CPUS")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    affinity, quota = cpu_budget()
    return max(1, min(affinity, quota) if quota else affinity)


def host_threads_per_rank(world):
    """DABHIP_HOST_THREADS of each of `world` ranks sharing this container: half the usable CPUs dealt to the ranks, 2 .. 24."""
    return max(2, min(24, usable_cpus() // (2 * max(1, world))))
