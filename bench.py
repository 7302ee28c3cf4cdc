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
# HBM bytes per TF of ofdm_fft_kernel from rocprofv3 PMC passes (profiles/r01_k2_pmc_traffic.csv):
# (2 x FETCH_SIZE + WRITE_SIZE) KiB per 1024-TF launch = 2 x 165.8k + 1259.5k, FETCH_SIZE doubled as the
# gfx950 note in MI355X_MICROARCH.md (HBM) prescribes.  PMC counters cannot be read from inside this script.
FFT_PMC_BYTES_PER_TF = (2 * 165800 + 1259500) * 1024 // 1024   # KiB per 1024 TF == bytes per TF: 1,591,100
REALTIME_FPS = 1000.0 / 24.0


def make_streams(torch, dev, nstreams, ntf, ndistinct, rank, snr_db=1000.0, host_synth=False):
    """nstreams synthetic ensembles resident on the device.  Default: every stream its own ensemble (payload, CIF
    counter, noise), modulated by the device-side modulator (k_synth.hip).  --host-synth: the host generator,
    ndistinct ensembles tiled to nstreams (the round-1 recipe; slow beyond a few streams)."""
    import dabtools_amd as dab

    from dabtools_amd import shard

    def cfg_of(i):   # global stream index of this rank's i-th ensemble -> seed rule of SURVEY.md 8(d)
        g = rank * nstreams + i
        return dab.synth_preset(0, seed=shard.stream_seed(2, g), cif_count0=(97 * g) % 5000, snr_db=snr_db)

    if host_synth:
        with ThreadPoolExecutor(max_workers=min(8, ndistinct)) as ex:
            host = list(ex.map(lambda i: dab.synth_generate(cfg_of(i), ntf), range(ndistinct)))
        base = [torch.from_numpy(h).to(dev) for h in host]
        return [base[i] if i < ndistinct else base[i % ndistinct].clone() for i in range(nstreams)], ndistinct
    cfgs = [cfg_of(i) for i in range(nstreams)]
    tensors = [torch.empty(dab.synth_bytes(c, ntf), dtype=torch.uint8, device=dev) for c in cfgs]
    dab.synth_generate_device(cfgs, ntf, [t.data_ptr() for t in tensors], dev.index or 0)
    return tensors, nstreams


def payload_stats(dab, eng, first_global_stream, nstreams, ntf):
    """Decoded payload vs what the modulator sent, over the distinct streams of this rank (noisy configs)."""
    from dabtools_amd import shard
    frames = good = bit_err = bits = 0
    expected = nstreams * 4 * (ntf - 15)
    for b in range(nstreams):
        g = first_global_stream + b
        cfg = dab.synth_preset(0, seed=shard.stream_seed(2, g), cif_count0=(97 * g) % 5000)
        fib_index = {dab.synth_fibs(cfg, c).tobytes(): c for c in range(4 * ntf)}
        for e in eng.eti(b):
            frames += 1
            nst = int(e[5]) & 0x7f
            pos = 12 + 4 * nst
            cif = fib_index.get(e[pos:pos + 96].tobytes())
            if cif is None or nst != cfg.nsub:
                continue
            pos += 96
            wrong = 0
            for k in range(nst):
                want = dab.synth_payload(cfg, cif, k)
                wrong += int(np.unpackbits(np.bitwise_xor(e[pos:pos + want.size], want)).sum())
                bits += 8 * want.size
                pos += want.size
            bit_err += wrong
            good += int(wrong == 0)
    return {"streams_checked": nstreams, "frames_expected_if_locked": expected, "frames_out": frames, "error_free_frames": good,
            "payload_ber": (bit_err / bits) if bits else None}


def cpu_baseline(host_streams, ntf, nsample=10):
    """The CPU restatement (oracle/, kind 'port') timed on one host core on a bounded sample of the same
    workload, plus -- when oracle/_ref was built -- the REAL reference back end (dab_process_frame with the
    scalar viterbi.c and with ENABLE_SPIRAL_VITERBI) on the same demapped frames.  Checker code, used here
    only as the reported baseline."""
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    sample = [h[: ntf * 393216] for h in host_streams[:nsample]]     # ~10 s of scalar work at the default sizes
    neti, t0 = 0, time.perf_counter()
    for iq in sample:
        neti += len(oracle_lib.or_replay(iq)[0])
    dt = time.perf_counter() - t0
    out = {
        "value": neti / dt, "unit": "ETI frames/s", "cores": 1, "kind": "port",
        "sample": "oracle/or_replay (scalar viterbi.c semantics, own fp64 DFT: libfftw3 absent) on %d streams x %d TF of the same "
                  "workload: %d ETI frames in %.2f s on 1 of %d host cores" % (len(sample), ntf, neti, dt, os.cpu_count()),
    }
    # the reference's own back end (the front end needs libfftw3 and cannot be built): demapped TFs are produced
    # untimed by the oracle front end, then dab_process_frame of the real objects is timed
    O = oracle_lib.oracle()
    fic = np.zeros(9216, np.uint8)
    msc = np.zeros(221184, np.uint8)
    streams_tfs = []
    for iq in sample:
        S, tfs = O.or_sdr_new(), []
        for off in range(0, iq.size - 262144 + 1, 262144):
            if O.or_sdr_demod(S, oracle_lib._ptr(iq[off:off + 262144]), 262144, oracle_lib._ptr(fic), oracle_lib._ptr(msc)):
                tfs.append((fic.copy(), msc.copy()))
        O.or_sdr_free(S)
        streams_tfs.append(tfs)
    ntfs = sum(len(t) for t in streams_tfs)
    for key, sse in (("reference_backend_scalar", False), ("reference_backend_sse", True)):
        R = oracle_lib.ref(sse=sse)
        if R is None:
            continue
        devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(2)
        os.dup2(devnull, 2)                 # the reference prints its ensemble table to stderr
        try:
            n, dt = 0, 0.0
            for tfs in streams_tfs:
                H = R.refh_new()
                t0 = time.perf_counter()
                for f, m in tfs:
                    C.memmove(R.refh_tf_fic(H), oracle_lib._ptr(f), f.size)
                    C.memmove(R.refh_tf_msc(H), oracle_lib._ptr(m), m.size)
                    R.refh_process(H)
                dt += time.perf_counter() - t0
                n += R.refh_neti(H)
        finally:
            os.dup2(saved, 2)
            os.close(devnull)
            os.close(saved)
        out[key] = {"value": n / dt, "unit": "ETI frames/s", "cores": 1,
                    "sample": "real reference dab_process_frame (%s) on %d demapped TF: %d ETI frames in %.2f s; back end only"
                              % ("viterbi_spiral SSE2" if sse else "scalar viterbi.c", ntfs, n, dt)}
        # per-stage micro-timing: the reference's decoder alone, data Mbit/s on 4608-bit code words (192 kbit/s)
        nbits, reps = 4608, (40 if sse else 8)
        rng = np.random.default_rng(1)
        sym = np.where(rng.integers(0, 2, 4 * (nbits + 6)) > 0, 255 if sse else 129, 0 if sse else 127).astype(np.uint8)
        data = np.zeros(nbits // 8 + 8, np.uint8)
        H = R.refh_new()
        t0 = time.perf_counter()
        for _ in range(reps):
            R.refh_viterbi(H, oracle_lib._ptr(sym), oracle_lib._ptr(data), nbits)
        out[key]["viterbi_mbit_s"] = reps * nbits / (time.perf_counter() - t0) / 1e6
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--tfs", type=int, default=64, help="transmission frames per stream")
    ap.add_argument("--host-synth", action="store_true", help="modulate on the host (--distinct ensembles, tiled) instead of on the GPU")
    ap.add_argument("--distinct", type=int, default=16, help="with --host-synth: distinct ensembles generated on the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--snr", type=float, default=1000.0, help="AWGN SNR in dB over the 2.048 MHz band (BASELINE config 5: 5 dB); default: clean")
    ap.add_argument("--no-fused-variant", action="store_true", help="skip the extra timed pass with the fused OFDM stage")
    ap.add_argument("--subchannels", type=str, default="", help="extension: decode only these SubChIds, e.g. 5 or 1,9 (default: all = reference frames)")
    ap.add_argument("--soft", action="store_true", help="soft-decision decoding (extension; default: hard = reference semantics)")
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

    t_gen = time.perf_counter()
    tensors, ndistinct = make_streams(torch, dev, args.streams, args.tfs, min(args.distinct, args.streams), rank, args.snr, args.host_synth)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    ptrs = [t.data_ptr() for t in tensors]
    sizes = [t.numel() for t in tensors]
    eng = dab.Engine(local_rank)
    if args.soft:
        eng.set_soft(True)
    if args.subchannels:
        eng.set_subchannels([int(x) for x in args.subchannels.split(",")])
    eng.decode_device(ptrs[:2], [min(s, 20 * 393216) for s in sizes[:2]])   # loads the code objects (tiny, untimed, part of set-up)

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

    # Reported separately: the same job with K2 + K2b fused into one kernel that never writes the spectra.
    # Not HBM-bound, so it carries no roofline; `value` above is the default pipeline with the separate K2.
    fused = None
    if not args.no_fused_variant and not args.soft:
        eng.set_fused(True)
        eng.decode_device(ptrs, sizes)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            f2 = eng.decode_device(ptrs, sizes)
        barrier()
        e2, f2 = shard.aggregate(time.perf_counter() - t1, f2, dev)
        fused = {"value": f2 * args.steps / e2, "unit": "ETI frames/s", "ms_per_step": 1e3 * e2 / args.steps,
                 "stage_ms_per_step": {k: v for k, v in eng.stage_ms().items() if k in ("fft", "demap")},
                 "note": "dabhip_engine_set_fused(1): OFDM transform + demap in one kernel, 311,296 B read + 28,800 B written per TF, "
                         "identical ETI bytes; stage 'fft' is the fused kernel"}
        eng.set_fused(False)

    if rank == 0:
        value = total_frames_per_step * args.steps / elapsed
        achieved = FFT_BYTES_PER_TF * fft_tfs / (fft_ms * 1e-3) / 1e9 if fft_ms > 0 else 0.0
        out = {
            "metric": "ETI frames/s (24 ms each), Mode-I batch, synthetic IQ resident in HBM",
            "value": value, "unit": "ETI frames/s", "x_realtime": value / REALTIME_FPS,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": ("synthetic (%d distinct ensembles per GPU tiled to %d streams, host modulator)" % (ndistinct, args.streams)) if args.host_synth
                    else ("synthetic (%d distinct ensembles per GPU, device-side modulator, %.1f s)" % (args.streams, t_gen)),
            "config": {"workload": "BASELINE configs[2]: batch=%d synthetic Mode-I streams x %d TF per GPU, 12 sub-channels (6 UEP + 6 EEP) 1136 kbit/s full MSC"
                                   % (args.streams, args.tfs),
                       "streams_per_gpu": args.streams, "tf_per_stream": args.tfs, "eti_frames_per_step": total_frames_per_step,
                       "sharding": "independent ensembles, %d per GPU, no collective" % args.streams},
            "roofline": {"bound": "hbm", "kernel": "ofdm_fft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": FFT_PMC_BYTES_PER_TF * fft_tfs / max(fft_launches, 1),
                         "traffic_note": "bytes per launch; per-TF HBM bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                         "(profiles/r01_k2_pmc_traffic.csv, FETCH_SIZE x2 gfx950 correction) x TFs per launch",
                         "achieved_vs_measured_copy_ceiling": achieved / 6290.0,
                         # a kernel with K2's loads and stores and no transform (tools/k2_traffic_probe.cpp, profiles/r01_k2_traffic_probe.txt):
                         # 5317 GB/s with K2's workgroup shape and prefetch, 5696 GB/s at the finest granularity
                         "achieved_vs_traffic_only_probe": achieved / 5317.0,
                         "launches": fft_launches, "tf_per_launch": fft_tfs / max(fft_launches, 1),
                         "avg_launch_ms": fft_ms / max(fft_launches, 1), "algorithmic_bytes_per_tf": FFT_BYTES_PER_TF},
            "stage_ms_per_step": {k: v / args.steps for k, v in stage.items()},
        }
        if fused:
            out["fused_variant"] = fused
        if args.subchannels:
            out["config"]["subchannel_filter"] = args.subchannels
        if args.snr < 100.0:
            out["config"]["snr_db"] = args.snr
            out["config"]["decisions"] = "soft (4-bit)" if args.soft else "hard"
            out["payload"] = payload_stats(dab, eng, rank * args.streams, min(16, args.streams), args.tfs)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline([t.cpu().numpy() for t in tensors[:10]], args.tfs)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
