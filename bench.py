#!/usr/bin/env python3
"""bench.py — ETI frames/s of the MI355X dab2eti hot path on synthetic Mode-I IQ.

One "step" = one pass of the whole hot path (sync scan -> OFDM transform + demap -> FIC decode -> control plane -> MSC
Viterbi -> ETI assembly) over one batch of B independent cu8 streams that are already resident in HBM
(BASELINE.json configs[2]: batch=256, canonical 12 sub-channel 1136 kbit/s ensemble, full MSC).  ETI frames stay in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, every rank decodes its own 256 streams (weak scaling; ensembles are independent, so there is
no data-path collective and no RCCL traffic at all).  Two ways to get the N ranks:
  * `python bench.py --gpus N` by itself: this process starts N rank processes (before anything here touches the GPU)
    and plays barrier / reducer for them over their pipes;
  * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK / WORLD_SIZE in the environment): the
    ranks keep their books over a gloo group.
Either way: W warm-up steps, barrier + device sync, exactly K timed steps, device sync + barrier, MAX of the elapsed time
over ranks, SUM of the frames; rank 0 prints ONE JSON line.  `--dry-run` exercises launch, sharding and book-keeping
without a GPU (CPU test-suite).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FFT_BYTES_PER_TF = 311296 + 1245184        # cu8 read + complex64 spectra written
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
# HBM bytes per TF of ofdm_fft_kernel from rocprofv3 PMC passes (profiles/r02_k2_pmc_traffic.csv):
# (2 x FETCH_SIZE + WRITE_SIZE) KiB per 1024-TF launch = 2 x 161,196 + 1,245,184, FETCH_SIZE doubled as the gfx950 note in
# MI355X_MICROARCH.md (HBM) prescribes.  PMC counters cannot be read from inside this script.
FFT_PMC_BYTES_PER_TF = (2 * 161196 + 1245184) * 1024 // 1024   # KiB per 1024 TF == bytes per TF: 1,567,576 = 1.007 x algorithmic
REALTIME_FPS = 1000.0 / 24.0
# VALU issue peak: 256 CUs x 4 SIMDs, one wave-instruction per 4 cycles (a quad-cycle) at 2.4 GHz
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4
VIT_VALU_PER_STEP = 129.0                  # VALU wave-instructions per trellis step of viterbi_fused_kernel<1> (PMC, see roofline_viterbi)
# The same step priced in issue clocks per SIMD (profiles/r02_valu_rates.txt, tools/ubench/valu_rates.hip: wave64 instructions do not all
# cost a quad-cycle on gfx950): 64 v_add_u32 at 2.56 + 32 v_pk_max_u16 at 4.28 + 13 v_perm_b32 at 4.2 + ~20 others at ~4.2
VIT_ISSUE_CLOCKS_PER_STEP = 64 * 2.56 + 32 * 4.28 + 13 * 4.2 + 20 * 4.2
VIT_REC_BYTES_PER_WAVE_STEP = 512.0 * 1.45   # 64 lanes x 8 B written per step; read back: 16 of 64 bytes per lane and block = ~45 % of the sectors
HBM_STREAM_MIX_NOTE = "compare with roofline.stream_ceiling.copy, not with the 8 TB/s of the data sheet"


# ---- rank coordination ---------------------------------------------------------------------------------------------
class Single:
    rank, world = 0, 1

    def barrier(self):
        pass

    def gather(self, obj):
        return [obj]

    def close(self):
        pass


class Pipes:
    """Child of launch_ranks(): lines '@@<verb> <json>' on stdout, one reply line on stdin."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def _ask(self, verb, obj=None):
        sys.stdout.write("@@%s %s\n" % (verb, json.dumps(obj)))
        sys.stdout.flush()
        line = sys.stdin.readline()
        if not line:
            raise RuntimeError("bench launcher went away")
        return json.loads(line)

    def barrier(self):
        self._ask("barrier")

    def gather(self, obj):
        return self._ask("gather", obj)      # every rank gets the list (rank order)

    def close(self):
        pass


class Gloo:
    """Ranks started by torch.distributed.run: book-keeping over gloo (CPU); the data path has no collective."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def barrier(self):
        self.dist.barrier()

    def gather(self, obj):
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        self.dist.destroy_process_group()


def launch_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: N rank processes, this process their barrier and reducer.  Nothing in this
    process has touched torch or HIP."""
    n = args.gpus
    cores = os.cpu_count() or 8
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), DABHIP_BENCH_PIPES="1",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("DABHIP_HOST_THREADS", str(max(2, min(24, cores // (2 * n)))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdin=subprocess.PIPE,
                                      stdout=subprocess.PIPE, text=True, bufsize=1))
    lock = threading.Condition()
    pending = {}          # verb -> {rank: payload}
    failed = []

    def reader(r, p):
        for line in p.stdout:
            if not line.startswith("@@"):
                sys.stdout.write(line)           # rank 0's JSON line (and anything else a rank prints)
                sys.stdout.flush()
                continue
            verb, _, payload = line[2:].partition(" ")
            with lock:
                pending.setdefault(verb, {})[r] = json.loads(payload)
                if len(pending[verb]) == n:
                    got = pending.pop(verb)
                    reply = json.dumps([got[i] for i in range(n)] if verb == "gather" else True)
                    for q in procs:
                        try:
                            q.stdin.write(reply + "\n")
                            q.stdin.flush()
                        except BrokenPipeError:
                            pass
        if p.wait() != 0:
            failed.append(r)
            for q in procs:                      # a dead rank must not leave the others waiting at a barrier
                if q.poll() is None:
                    q.terminate()

    threads = [threading.Thread(target=reader, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return 1 if failed or any(p.returncode for p in procs) else 0


# ---- workload -----------------------------------------------------------------------------------------------------
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
    """Decoded payload vs what the modulator sent, over the first streams of this rank (noisy configs)."""
    import numpy as np
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


def cpu_baseline(tensors, ntf, nsample):
    """tools/cpu_baseline.py on the first streams of this very workload, as a CHILD process (one process per core needs
    fork, which a process that has initialised the GPU must not do)."""
    with tempfile.TemporaryDirectory(prefix="dabhip_bench_") as tmp:
        files = []
        for i, t in enumerate(tensors[:nsample]):
            path = os.path.join(tmp, "s%d.cu8" % i)
            t.cpu().numpy().tofile(path)
            files.append(path)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--tfs", str(ntf), "--iq"] + files
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if res.returncode != 0:
        return {"error": res.stderr[-400:]}
    return json.loads(res.stdout.strip().splitlines()[-1])


def workload_text(args):
    if args.snr < 100.0:
        return ("BASELINE configs[4]: batch=%d synthetic Mode-I streams x %d TF per GPU, AWGN %.1f dB, %s-decision Viterbi, 12 sub-channels "
                "(6 UEP + 6 EEP) 1136 kbit/s full MSC" % (args.streams, args.tfs, args.snr, "soft" if args.soft else "hard"))
    return ("BASELINE configs[2]: batch=%d synthetic Mode-I streams x %d TF per GPU, 12 sub-channels (6 UEP + 6 EEP) 1136 kbit/s full MSC"
            % (args.streams, args.tfs))


# ---- one rank -----------------------------------------------------------------------------------------------------
def run_rank(args, coord):
    rank, world = coord.rank, coord.world
    from dabtools_amd import shard
    mine = shard.shard_streams(world * args.streams, world, rank)          # global stream indices of this rank
    rank_info = {"rank": rank, "first_stream": mine[0], "last_stream": mine[-1], "streams": len(mine),
                 "first_seed": shard.stream_seed(2, mine[0]), "host_threads": os.environ.get("DABHIP_HOST_THREADS", "auto")}

    if args.dry_run:
        coord.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * args.steps * (1 + rank))                         # ranks finish at different times: MAX is what counts
        elapsed = time.perf_counter() - t0
        coord.barrier()
        frames = 4 * (args.tfs - 15) * len(mine)
        stage, fft, fused_off, extra = {}, None, None, {}
    else:
        import torch
        import dabtools_amd as dab
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0                                                 # test knob: all ranks share GPU 0 (exercises the N-rank path on a 1-GPU box)
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
        if args.two_kernel_ofdm:
            eng.set_fused(False)
        if args.no_parity_guard:
            eng.set_parity_guard(False)
        if args.subchannels:
            eng.set_subchannels([int(x) for x in args.subchannels.split(",")])
        eng.decode_device(ptrs[:2], [min(s, 20 * 393216) for s in sizes[:2]])   # loads the code objects (tiny, untimed, part of set-up)

        def barrier():
            torch.cuda.synchronize(dev)
            coord.barrier()
            torch.cuda.synchronize(dev)

        frames = 0
        for _ in range(args.warmup):
            frames = eng.decode_device(ptrs, sizes)
        barrier()
        t0 = time.perf_counter()
        stage = {}
        for _ in range(args.steps):
            frames = eng.decode_device(ptrs, sizes)
            for k, v in eng.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v
        barrier()
        elapsed = time.perf_counter() - t0
        stage = {k: v / args.steps for k, v in stage.items()}

        # Roofline: K2 = ofdm_fft_kernel by itself, on the same resident IQ and the frame list of the step just
        # timed, HIP events on the engine's stream around every launch.  The pipeline's default OFDM stage fuses K2 with the
        # demapper and never writes the spectra; that kernel is not HBM-bound and carries no roofline figure.
        fft = None
        fused_off = None
        extra = {}
        if rank == 0:
            flagged, decisions = eng.guard_stats()
            extra["parity_guard"] = {"on": not args.no_parity_guard and not args.soft, "decisions_per_step": decisions, "redecided_in_fp64_per_step": flagged,
                                     "note": "hard decisions whose fp32 margin lies inside the error band are re-decided in fp64 from the int8 samples "
                                             "(k_parity.hip); raw fp32 disagreement rate without it: profiles/r02_decision_audit.json"}
            fft = eng.fft_roofline(max(3, min(args.steps, 10)))
            if rank == 0:
                # what a bare streaming kernel with K2's read/write mix reaches on this device, over a footprint like K2's (untimed part)
                try:
                    extra["stream_ceiling"] = dab.stream_ceiling(local_rank, 4 << 30, 3)
                except dab.DabhipError as e:
                    extra["stream_ceiling"] = {"error": str(e)}
            if not args.soft and not args.no_parity_guard and not args.no_variants:
                # the same job accepting raw fp32 decisions (guard off: the fused kernel without the guard's test)
                eng.set_parity_guard(False)
                eng.decode_device(ptrs, sizes)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    f3 = eng.decode_device(ptrs, sizes)
                torch.cuda.synchronize(dev)
                e3 = time.perf_counter() - t1
                extra["parity_guard_off_variant"] = {"value": f3 * args.steps / e3, "unit": "ETI frames/s", "ms_per_step": 1e3 * e3 / args.steps,
                                                     "stage_ms_per_step": {k: v for k, v in eng.stage_ms().items() if k in ("fft", "demap")},
                                                     "note": "dabhip_engine_set_parity_guard(0): raw fp32 decisions (disagreement with exact arithmetic: 0 on this "
                                                             "clean workload, 2.6e-8 of the decisions at 5 dB); this rank only"}
                eng.set_parity_guard(True)
            if not args.soft and not args.two_kernel_ofdm and not args.no_variants:
                # the same job with the two-kernel OFDM stage (K2 writes the spectra, K2b reads them back), for comparison
                eng.set_fused(False)
                eng.decode_device(ptrs, sizes)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    f2 = eng.decode_device(ptrs, sizes)
                torch.cuda.synchronize(dev)
                e2 = time.perf_counter() - t1
                a, b, c = eng.fft_stats()
                fused_off = {"value": f2 * args.steps / e2, "unit": "ETI frames/s", "ms_per_step": 1e3 * e2 / args.steps,
                             "stage_ms_per_step": {k: v for k, v in eng.stage_ms().items() if k in ("fft", "demap")},
                             "k2_in_pipeline_avg_launch_ms": c / max(a, 1),
                             "note": "dabhip_engine_set_fused(0): K2 (cu8 -> complex64 spectra) + K2b (spectra -> bits) as two kernels, "
                                     "identical ETI bytes; this rank only, untimed against the other ranks"}
                eng.set_fused(True)
            extra["data"] = (("synthetic (%d distinct ensembles per GPU tiled to %d streams, host modulator)" % (ndistinct, args.streams)) if args.host_synth
                             else ("synthetic (%d distinct ensembles per GPU, device-side modulator, %.1f s)" % (args.streams, t_gen)))
            if args.snr < 100.0:
                eng.decode_device(ptrs, sizes)
                extra["payload"] = payload_stats(dab, eng, rank * args.streams, min(16, args.streams), args.tfs)
            if not args.no_cpu_baseline and world == 1:
                extra["cpu_baseline"] = cpu_baseline(tensors, args.tfs, args.cpu_sample)

    rank_info["host_ms_per_step"] = {k: round(stage[k], 4) for k in ("control", "host_worklist", "host_setup", "host_frames", "wall") if k in stage}
    rows = coord.gather({"info": rank_info, "elapsed": elapsed, "frames": frames})
    if rank == 0:
        elapsed_max = max(r["elapsed"] for r in rows)
        frames_step = sum(r["frames"] for r in rows)
        value = frames_step * args.steps / elapsed_max
        out = {
            "metric": "ETI frames/s (24 ms each), Mode-I batch, synthetic IQ resident in HBM",
            "value": value, "unit": "ETI frames/s", "x_realtime": value / REALTIME_FPS,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            # arithmetic of the path: fp64 sync estimators (K1), fp32 OFDM transform + differential demodulation,
            # packed-u16 integer add-compare-select, u8 ETI bytes
            "dtype": "f64 sync / f32 OFDM / u16 ACS / u8 ETI",
            "data": extra.get("data", "none (dry run)"),
            "config": {"workload": workload_text(args), "streams_per_gpu": args.streams, "tf_per_stream": args.tfs,
                       "eti_frames_per_step": frames_step,
                       "ofdm_stage": "K2 + K2b (two kernels)" if (args.two_kernel_ofdm or args.soft) else "fused transform + demap (default)",
                       "sharding": "independent ensembles, %d per GPU, stream s on rank s // %d, no collective" % (args.streams, args.streams)},
            "ranks": [dict(r["info"], elapsed_s=r["elapsed"], eti_frames_per_step=r["frames"]) for r in rows],
        }
        if args.dry_run:
            out["dry_run"] = True
        if fft:
            launches, tfs, ms = fft
            achieved = FFT_BYTES_PER_TF * tfs / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out["roofline"] = {
                "bound": "hbm", "kernel": "ofdm_fft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": FFT_PMC_BYTES_PER_TF * tfs / max(launches, 1),
                "traffic_note": "bytes per launch; per-TF HBM bytes from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                "(profiles/r02_k2_pmc_traffic.csv, FETCH_SIZE x2 gfx950 correction) x TFs per launch",
                "how": "dabhip_engine_fft_roofline: K2 alone over the frame list of the timed step (same resident IQ, %d TF per launch), "
                       "HIP events on the engine's stream; the step itself runs the fused transform + demap kernel" % (tfs // max(launches, 1)),
                "achieved_vs_measured_copy_ceiling": achieved / 6290.0,
                "stream_ceiling": dict(extra.get("stream_ceiling", {}), unit="GB/s",
                                       note="bare grid-stride kernels of this library on this device, 4 GiB buffers, best of three grid sizes (k_probe.hip): "
                                            "fill = write only, copy = 1 read : 1 written, k2_mix = 1 byte read per 4 written with K2's 16-byte nontemporal stores"),
                "achieved_vs_k2_mix_stream": (achieved / extra["stream_ceiling"]["k2_mix"]) if extra.get("stream_ceiling", {}).get("k2_mix") else None,
                "launches": launches, "tf_per_launch": tfs / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                "algorithmic_bytes_per_tf": FFT_BYTES_PER_TF}
        if stage:
            out["stage_ms_per_step"] = stage
            if stage.get("viterbi", 0) > 0 and not args.dry_run and not args.subchannels:
                # second roofline, for the stage with the most time after the OFDM kernel: the MSC Viterbi is bound by VALU issue and by
                # the survivor records.  Per trellis step and wave (64 code words): VIT_VALU_PER_STEP wave-instructions (rocprofv3 SQ_INSTS_VALU,
                # profiles/r01_sq_pmc_summary.csv: 2.77e9 per 2.14e7 wave-steps) and 512 B of records written + read back.
                steps_per_frame = 27336 if not args.soft else 27336
                wave_steps = frames_step / world * steps_per_frame / 64.0
                t = stage["viterbi"] * 1e-3
                out["roofline_viterbi"] = {
                    "kernel": "viterbi_fused_kernel", "bound": "valu issue / hbm (survivor records)",
                    "valu": {"achieved": VIT_VALU_PER_STEP * wave_steps / t / 1e9, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s",
                             "frac": VIT_VALU_PER_STEP * wave_steps / t / 1e9 / VALU_PEAK_GINST},
                    "issue_time": {"clocks_per_step": VIT_ISSUE_CLOCKS_PER_STEP, "frac": VIT_ISSUE_CLOCKS_PER_STEP * wave_steps / (256 * 4 * 2.4e9 * t),
                                   "note": "the step's instructions priced at their measured issue clocks (add 2.56, packed max 4.28, permute 4.2), "
                                           "over 1024 SIMDs at 2.4 GHz"},
                    "hbm": {"achieved": VIT_REC_BYTES_PER_WAVE_STEP * wave_steps / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": VIT_REC_BYTES_PER_WAVE_STEP * wave_steps / t / 1e9 / HBM_PEAK_GBS,
                            "achieved_vs_copy_stream": (VIT_REC_BYTES_PER_WAVE_STEP * wave_steps / t / 1e9 / extra["stream_ceiling"]["copy"]) if extra.get("stream_ceiling", {}).get("copy") else None,
                            "note": "64 B of survivor records per 8 steps and code word, written once; the chain-back reads the 16-byte part that holds its "
                                    "state's byte (about 45 % of the 32-byte sectors); " + HBM_STREAM_MIX_NOTE},
                    "trellis_steps_per_eti_frame": steps_per_frame, "avg_ms": stage["viterbi"]}
        if fused_off:
            out["two_kernel_ofdm_variant"] = fused_off
        if args.subchannels:
            out["config"]["subchannel_filter"] = args.subchannels
        if args.snr < 100.0:
            out["config"]["snr_db"] = args.snr
            out["config"]["decisions"] = "soft (4-bit)" if args.soft else "hard"
        for k in ("parity_guard", "parity_guard_off_variant", "payload", "cpu_baseline"):
            if k in extra:
                out[k] = extra[k]
        print(json.dumps(out))
        sys.stdout.flush()
    coord.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)     # the second full-size decode of a process still pays one-time runtime costs (a 7 ms host stall)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--tfs", type=int, default=64, help="transmission frames per stream")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: launch, sharding and rank book-keeping only")
    ap.add_argument("--host-synth", action="store_true", help="modulate on the host (--distinct ensembles, tiled) instead of on the GPU")
    ap.add_argument("--distinct", type=int, default=16, help="with --host-synth: distinct ensembles generated on the host")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=6, help="streams of the workload the one-core CPU baseline replays")
    ap.add_argument("--snr", type=float, default=1000.0, help="AWGN SNR in dB over the 2.048 MHz band (BASELINE config 5: 5 dB); default: clean")
    ap.add_argument("--two-kernel-ofdm", action="store_true", help="time the K2 + K2b OFDM stage instead of the fused default")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timed passes (two-kernel OFDM stage, parity guard off)")
    ap.add_argument("--no-parity-guard", action="store_true", help="time the pipeline with raw fp32 decisions (dabhip_engine_set_parity_guard(0))")
    ap.add_argument("--subchannels", type=str, default="", help="extension: decode only these SubChIds, e.g. 5 or 1,9 (default: all = reference frames)")
    ap.add_argument("--soft", action="store_true", help="soft-decision decoding (extension; default: hard = reference semantics)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))          # before any torch / HIP call in this process
    if os.environ.get("DABHIP_BENCH_PIPES") == "1" and world_env > 1:
        coord = Pipes(int(os.environ["RANK"]), world_env)
    elif world_env > 1:
        cores = os.cpu_count() or 8
        os.environ.setdefault("DABHIP_HOST_THREADS", str(max(2, min(24, cores // (2 * world_env)))))
        coord = Gloo()
    else:
        coord = Single()
    run_rank(args, coord)


if __name__ == "__main__":
    main()
