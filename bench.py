#!/usr/bin/env python3
"""bench.py — ETI frames/s of the MI355X dab2eti hot path on synthetic Mode-I IQ.

One "step" = one pass of the whole hot path (sync scan -> OFDM FFT -> demap -> FIC decode ->
control plane -> MSC Viterbi -> ETI assembly) over one batch of B independent cu8 streams
that are already resident in HBM (BASELINE.json configs[2]: batch=256, canonical 12
sub-channel 1136 kbit/s ensemble, full MSC).  ETI frames stay in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run; every rank decodes its own
256 streams (weak scaling, no data-path collective: ensembles are independent).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT_BYTES_PER_TF = 311296 + 1245184        # cu8 read + complex64 spectra written
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
REALTIME_FPS = 1000.0 / 24.0


def make_streams(torch, dev, nstreams, ntf, ndistinct, rank):
    """ndistinct different synthetic ensembles generated on the host cores, tiled to nstreams on the device."""
    import dabtools_amd as dab

    from dabtools_amd import shard

    def gen(i):   # global stream index of this rank's i-th distinct ensemble -> seed rule of SURVEY.md 8(d)
        g = rank * nstreams + i
        cfg = dab.synth_preset(0, seed=shard.stream_seed(2, g), cif_count0=(97 * g) % 5000)
        return dab.synth_generate(cfg, ntf)

    with ThreadPoolExecutor(max_workers=min(8, ndistinct)) as ex:
        host = list(ex.map(gen, range(ndistinct)))
    base = [torch.from_numpy(h).to(dev) for h in host]
    tensors = [base[i] if i < ndistinct else base[i % ndistinct].clone() for i in range(nstreams)]
    return host, tensors


def cpu_baseline(host_stream, ntf):
    """The CPU restatement (oracle/, kind 'port') timed on one host core on a bounded sample
    of the same workload.  Checker code used only as the reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    sample_tf = min(ntf, 40)
    iq = host_stream[: sample_tf * 393216]
    t0 = time.perf_counter()
    eti, _ = oracle_lib.or_replay(iq)
    dt = time.perf_counter() - t0
    return {
        "value": len(eti) / dt, "unit": "ETI frames/s", "cores": 1, "kind": "port",
        "sample": "oracle/or_replay (scalar viterbi.c semantics, own fp64 DFT: libfftw3 absent) on 1 stream x %d TF of the same "
                  "ensemble: %d ETI frames in %.2f s on 1 of %d host cores" % (sample_tf, len(eti), dt, os.cpu_count()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--tfs", type=int, default=64, help="transmission frames per stream")
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic ensembles generated on the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import dabtools_amd as dab
    from dabtools_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    host, tensors = make_streams(torch, dev, args.streams, args.tfs, min(args.distinct, args.streams), rank)
    ptrs = [t.data_ptr() for t in tensors]
    sizes = [t.numel() for t in tensors]
    eng = dab.Engine(local_rank)

    def barrier():
        shard.barrier(dev)

    frames = 0
    for _ in range(args.warmup):
        frames = eng.decode_device(ptrs, sizes)
    barrier()
    t0 = time.perf_counter()
    fft_launches = fft_tfs = 0
    fft_ms = 0.0
    stage = {}
    for _ in range(args.steps):
        frames = eng.decode_device(ptrs, sizes)
        a, b, c = eng.fft_stats()
        fft_launches += a
        fft_tfs += b
        fft_ms += c
        for k, v in eng.stage_ms().items():
            stage[k] = stage.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, total_frames_per_step = shard.aggregate(elapsed, frames, dev)

    if rank == 0:
        value = total_frames_per_step * args.steps / elapsed
        achieved = FFT_BYTES_PER_TF * fft_tfs / (fft_ms * 1e-3) / 1e9 if fft_ms > 0 else 0.0
        out = {
            "metric": "ETI frames/s (24 ms each), Mode-I batch, synthetic IQ resident in HBM",
            "value": value, "unit": "ETI frames/s", "x_realtime": value / REALTIME_FPS,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic (%d distinct ensembles per GPU tiled to %d streams)" % (min(args.distinct, args.streams), args.streams),
            "config": {"workload": "BASELINE configs[2]: batch=%d synthetic Mode-I streams x %d TF per GPU, 12 sub-channels (6 UEP + 6 EEP) 1136 kbit/s full MSC"
                                   % (args.streams, args.tfs),
                       "streams_per_gpu": args.streams, "tf_per_stream": args.tfs, "eti_frames_per_step": total_frames_per_step,
                       "sharding": "independent ensembles, %d per GPU, no collective" % args.streams},
            "roofline": {"bound": "hbm", "kernel": "ofdm_fft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "launches": fft_launches, "tf_per_launch": fft_tfs / max(fft_launches, 1),
                         "avg_launch_ms": fft_ms / max(fft_launches, 1), "algorithmic_bytes_per_tf": FFT_BYTES_PER_TF},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items()},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host[0], args.tfs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
