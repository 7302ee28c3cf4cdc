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
FUSED_BYTES_PER_TF = 311296 + 28800        # the fused OFDM stage: cu8 read + bit-packed decisions written
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec
REALTIME_FPS = 1000.0 / 24.0
SIMDS, XCDS = 256 * 4, 8
# Every roofline input that cannot be measured from inside this script (PMC counters, the effective clock) is read from the tracked
# profile of the round, produced by tools/refresh_profiles.sh (rocprofv3 passes over THIS script) -- never baked in here.
PROFILE_PMC = os.path.join(ROOT, "profiles", "r06_pmc_summary.csv")
# static instruction mix of the fused OFDM kernel's symbol loop, priced in issue cycles (tools/fused_isa_mix.sh; CPU only)
PROFILE_FUSED_MIX = os.path.join(ROOT, "profiles", "r06_fused_isa_mix.json")
# issue cost of a wave64 VALU instruction on gfx950 (MI355X_MICROARCH.md, per-instruction table: v_fma_f32 2 cycles; tools/ubench/valu_rates.hip,
# profiles/r03_valu_rates.txt: plain 32-bit VOP1/VOP2 2, every VOP3 / VOP3P / DPP / 64-bit form 4)
CYC_SIMPLE, CYC_VOP3 = 2.0, 4.0


def counter_busy(prof, kernel):
    """VALU-busy fraction straight from the counters: SQ_ACTIVE_INST_VALU counts quad-cycles (x 4 = SIMD cycles with a VALU instruction in flight),
    GRBM_GUI_ACTIVE / 8 XCDs = the kernel's clocks; both from the SAME pass (f1), summed over the kernel's dispatches."""
    return 4.0 * prof.cell("f1", kernel, "SQ_ACTIVE_INST_VALU") / (SIMDS * prof.cell("f1", kernel, "GRBM_GUI_ACTIVE") / XCDS)
VIT_ADDS_PER_STEP = 64                     # the v_add_u32 of one trellis step (k_decode.hip butterfly_pair: 4 per register pair x 16), all 2-cycle
MSC_STEPS_PER_FRAME = 4 * 3078 + 2 * 4614 + 3 * 1542 + 774 + 2 * 198      # 27,336: trellis steps of the 12 sub-channels of the canonical mix
FIC_STEPS = 774


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
    from dabtools_amd import shard                      # (pure Python: nothing here loads the library or touches the GPU)
    # each rank's host pool: its share of the CPUs this container really has (affinity mask and CFS quota, not os.cpu_count(): shard.cpu_budget)
    threads = shard.host_threads_per_rank(n)
    # Each rank sees ONLY its GPU (ROCR_VISIBLE_DEVICES, set here, before the child exists, so before anything of it touches a GPU): no context on
    # the other seven devices, no way to land on the wrong one.  The r-th entry of the list this launcher was given, or r.  Not when the caller
    # selects devices through HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (those index the ROCR list: combining them would select twice), and not
    # in the one-GPU rehearsal (DABHIP_BENCH_ONE_DEVICE=1); then the rank picks device LOCAL_RANK as before.
    pin = not any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")) and os.environ.get("DABHIP_BENCH_ONE_DEVICE") != "1"
    visible = [x for x in os.environ.get("ROCR_VISIBLE_DEVICES", "").split(",") if x.strip()]
    if visible and len(visible) < n:
        pin = False
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), DABHIP_BENCH_PIPES="1",
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("DABHIP_HOST_THREADS", str(threads))
        if pin:
            env["ROCR_VISIBLE_DEVICES"] = visible[r].strip() if visible else str(r)
            env["DABHIP_BENCH_DEVICE"] = "0"             # the one device the rank sees
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


# ---- tracked profile inputs ---------------------------------------------------------------------------------------------
class Profile:
    """profiles/rNN_pmc_summary.csv (tools/sq_pmc_summary.py): cell(pass, kernel, counter) = counter summed over the kernel's dispatches
    in that rocprofv3 pass; meta(key) = what the profiled command was."""

    def __init__(self, path):
        self.path = os.path.relpath(path, ROOT)
        self.cells, self.metas = {}, {}
        with open(path) as f:
            for line in f:
                if line.startswith("#") or line.startswith("pass,"):
                    continue
                pas, rest = line.rstrip("\n").split(",", 1)
                kernel, ndisp, counter, value = rest.rsplit(",", 3)     # kernel names may hold commas (template arguments)
                if pas == "meta":
                    self.metas[kernel] = float(value)
                else:
                    self.cells[(pas, kernel, counter)] = float(value)

    def cell(self, pas, kernel, counter):
        key = (pas, kernel, counter)
        if key not in self.cells:
            raise SystemExit("bench.py: %s has no cell (%s, %s, %s): re-run tools/refresh_profiles.sh" % ((self.path,) + key))
        return self.cells[key]

    def meta(self, key):
        if key not in self.metas:
            raise SystemExit("bench.py: %s has no meta row %s: re-run tools/refresh_profiles.sh" % (self.path, key))
        return self.metas[key]

    def ref(self, pas, kernel, counter):
        return "%s: %s / %s / %s" % (self.path, pas, kernel, counter)


def load_profile():
    if not os.path.exists(PROFILE_PMC):
        raise SystemExit("bench.py: %s is missing.  The roofline objects are computed from the tracked rocprofv3 profile of this round; produce it "
                         "on a GPU box with tools/refresh_profiles.sh (its own passes run with --profile-pass)." % os.path.relpath(PROFILE_PMC, ROOT))
    return Profile(PROFILE_PMC)


def wave_steps(frames, tf_slots):
    """Trellis steps x waves of one decode: every wave = 64 code words of one length (12 sub-channels per ETI frame, 4 FIC blocks per TF)."""
    return -(-frames // 64) * MSC_STEPS_PER_FRAME + -(-4 * tf_slots // 64) * FIC_STEPS


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
    from dabtools_amd import payload
    chk = payload.PayloadCheck()
    for b in range(nstreams):
        chk.add_stream(dab, payload.bench_cfg(dab, first_global_stream + b), ntf, eng.eti(b))
    return chk.result()


def cpu_baseline(tensors, ntf, nsample, ber_first_stream=-1, snr=1000.0):
    """tools/cpu_baseline.py on the first streams of this very workload, as a CHILD process (one process per core needs
    fork, which a process that has initialised the GPU must not do).  ber_first_stream >= 0: also the payload BER of the real
    reference back ends (scalar and ENABLE_SPIRAL_VITERBI SSE) on those captures' hard decisions (BASELINE configs[4])."""
    with tempfile.TemporaryDirectory(prefix="dabhip_bench_") as tmp:
        files = []
        for i, t in enumerate(tensors[:nsample]):
            path = os.path.join(tmp, "s%d.cu8" % i)
            t.cpu().numpy().tofile(path)
            files.append(path)
        cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--tfs", str(ntf)]
        if ber_first_stream >= 0:
            cmd += ["--ber-first-stream", str(ber_first_stream), "--snr", str(snr)]
        cmd += ["--iq"] + files
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    if res.returncode != 0:
        return {"error": res.stderr[-400:]}
    return json.loads(res.stdout.strip().splitlines()[-1])


def cli_leg(streams, tfs):
    """tools/cli_throughput.py as a CHILD process: the dab2eti-hip executable, capture files in, 6144-byte frames on stdout out (the reference's
    CLI contract, dab2eti.c:117-135), on a reduced sample of the workload so that the default run stays short; the full-size figures are tracked in
    profiles/r04_cli_throughput.json."""
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    with tempfile.TemporaryDirectory(prefix="dabhip_cli_", dir=d) as tmp:
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cli_throughput.py"), "--streams", str(streams), "--tfs", str(tfs), "--dir", tmp],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if res.returncode != 0:
        return {"error": res.stderr[-400:]}
    r = json.loads(res.stdout)
    sp = r["stream_pipeline"]
    return {"command": "dab2eti-hip --stream --segment-calls 12 cap0000.cu8 .. cap%04d.cu8 > /dev/null   (files in %s)" % (streams - 1, d),
            "value": sp["inside_the_process"]["steady_frames_per_s"], "unit": "ETI frames/s",
            "value_is": "steady state inside the process: frames of all segments after the first productive one / time until the last byte is written",
            "whole_process": {"seconds": sp["seconds"], "eti_frames_per_s": sp["eti_frames_per_s"], "setup_s": sp["inside_the_process"]["setup_s"],
                              "note": "process start, HIP initialisation, page-locking the five segment buffers and lock-in included"},
            "two_sessions_on_one_gpu": {"command": "dab2eti-hip --stream --devices 0,0 ...   (dabhip_multi_stream: the inputs dealt to two sessions, both on GPU 0 here)",
                                        "value": r["stream_pipeline_two_sessions_on_one_gpu"]["inside_the_process"]["steady_frames_per_s"],
                                        "whole_process_eti_frames_per_s": r["stream_pipeline_two_sessions_on_one_gpu"]["eti_frames_per_s"]},
            "one_batch": r["one_batch"], "one_stream_from_stdin": r["one_stream_from_stdin"],
            "stdout_bytes_equal_library_frames": bool(r["batch_stdout_equals_library_frames"] and r["stream_stdout_equals_library_frames"] and r["stdin_stdout_equals_library_frames"]),
            "streams": streams, "tf_per_stream": tfs, "full_size": "profiles/r04_cli_throughput.json (256 streams x 256 TF = 25.8 GB of captures: 567 k ETI frames/s steady state)"}


def workload_text(args):
    if args.snr < 100.0:
        return ("BASELINE configs[4]: batch=%d synthetic Mode-I streams x %d TF per GPU, AWGN %.1f dB, %s-decision Viterbi, 12 sub-channels "
                "(6 UEP + 6 EEP) 1136 kbit/s full MSC" % (args.streams, args.tfs, args.snr, "soft" if args.soft else "hard"))
    return ("BASELINE configs[2]: batch=%d synthetic Mode-I streams x %d TF per GPU, 12 sub-channels (6 UEP + 6 EEP) 1136 kbit/s full MSC"
            % (args.streams, args.tfs))


# ---- one rank -----------------------------------------------------------------------------------------------------
def h2d_inclusive(dab, device, tensors, sizes, frames_resident, args):
    """The same workload with the IQ in HOST memory, PCIe included (never `value`): page-locked buffers (dabhip_host_alloc) through
    (a) one dabhip_engine_decode(on_device = 0): upload, then decode; (b) a dabhip_stream session in segments of --h2d-segment-tfs TF
    with dabhip_stream_prefetch: segment k + 1 uploads while segment k decodes; plus (c) pageable memory through the staging ring."""
    import numpy as np
    B, nbytes = len(tensors), sizes[0]
    pinned = [dab.HostBuffer(nbytes) for _ in range(B)]
    for hb, t in zip(pinned, tensors):
        assert dab.lib().dabhip_device_copy(hb.ptr, t.data_ptr(), nbytes, 0) == 0
    out = {"pinned": True, "unit": "ETI frames/s", "workload_bytes": B * nbytes,
           "note": "IQ starts in page-locked HOST memory (dabhip_host_alloc); PCIe-bound: 128,394 B of IQ per ETI frame for a 64-TF capture "
                   "(98,304 in the steady state of a session); never `value`"}
    eng = dab.Engine(device)

    # the ETI leg: the frames come back into page-locked host memory too -- host -> host, as the CLI contract has it
    # (dab2eti.c:117-135); in the session their download (dabhip_stream_eti_fetch) runs beside the next segment's upload and decode
    eti_host = [dab.HostBuffer(max(frames_resident, 1) * 6144) for _ in range(2)]
    out["eti_leg"] = "included: all frames of a decode / segment downloaded into page-locked host memory (dabhip_engine_eti_fetch / dabhip_stream_eti_fetch)"

    def one_shot(ptrs, nb, reps=3):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            n = eng.decode_host_ptrs(ptrs, nb)
            assert eng.eti_fetch(eti_host[0].ptr, n) == n
            eng.eti_fetch_wait()
            dt = time.perf_counter() - t0
            st = eng.stage_ms()
            rec = {"value": n / dt, "ms": 1e3 * dt, "eti_frames": n, "h2d_ms": st["h2d"], "h2d_GBps": st["h2d_mbytes"] / max(st["h2d"], 1e-9),
                   "end_to_end_GBps": sum(nb) / dt / 1e9, "pinned_fraction": st["h2d_pinned_mbytes"] / max(st["h2d_mbytes"], 1e-9)}
            if best is None or rec["value"] > best["value"]:
                best = rec
        return best

    out["one_shot"] = one_shot([hb.ptr for hb in pinned], sizes)
    out["one_shot"]["identical_frame_count"] = out["one_shot"]["eti_frames"] == frames_resident
    npage = min(B, 32)                                       # pageable memory: a sample (the copies are made here, 0.8 GB)
    pageable = [np.array(hb.array[:nbytes], copy=True) for hb in pinned[:npage]]
    out["one_shot_pageable"] = dict(one_shot([a.ctypes.data for a in pageable], sizes[:npage], reps=2), streams=npage,
                                    note="pageable host memory: copied into a ring of page-locked staging buffers by the engine's host pool, piece by piece")
    del pageable
    eng.close()
    seg = args.h2d_segment_tfs * dab.TF_BYTES
    cuts = list(range(0, nbytes, seg)) + [nbytes]
    segs = [([hb.ptr + a for hb in pinned], [z - a] * B) for a, z in zip(cuts, cuts[1:])]
    best = None
    for _ in range(2):
        st = dab.Stream(B, device=device)
        per_seg = []
        t0 = time.perf_counter()
        st.prefetch_ptrs(*segs[0])
        for k in range(len(segs)):
            tk = time.perf_counter()                        # an iteration = hand over segment k + 1, decode segment k
            if k + 1 < len(segs):
                st.prefetch_ptrs(*segs[k + 1])
            n = st.feed_ptrs(*segs[k])                      # (its K4 waits for the download of segment k - 1, issued below one iteration ago)
            if k >= 1:
                st.eti_fetch_wait()                         # segment k - 1's frames have had this whole feed to arrive (at most two fetches may be outstanding)
            st.eti_fetch(eti_host[k & 1].ptr, n)            # segment k's frames on their way while segment k + 1 uploads and decodes
            per_seg.append((n, time.perf_counter() - tk))
        st.eti_fetch_wait()
        dt = time.perf_counter() - t0
        st.close()
        total = sum(n for n, _ in per_seg)
        first = max(3, -(-16 // args.h2d_segment_tfs) + 1)   # after lock-in, and with an upload running beside the decode (not the last)
        steady = [(n, t) for n, t in per_seg[first:-1] if n > 0]
        rec = {"value": total / dt, "ms": 1e3 * dt, "eti_frames": total, "end_to_end_GBps": B * nbytes / dt / 1e9, "segments": len(segs),
               "segment_tfs": args.h2d_segment_tfs, "identical_frame_count": total == frames_resident,
               "steady_state": ({"value": sum(n for n, _ in steady) / sum(t for _, t in steady), "ms_per_segment": 1e3 * sum(t for _, t in steady) / len(steady),
                                 "GBps": len(steady) * B * seg / sum(t for _, t in steady) / 1e9, "segments_counted": len(steady)} if steady else None)}
        if best is None or rec["value"] > best["value"]:
            best = rec
    out["session_prefetch"] = best
    # the headline of this object: the sustained host-fed rate (steady state of the session), with the one-shot figure beside it
    out["value"] = (best["steady_state"] or best)["value"]
    out["GBps"] = (best["steady_state"] or best).get("GBps", best["end_to_end_GBps"])
    for hb in pinned + eti_host:
        hb.free()
    return out


def steady_state(dab, torch, dev, tensors, args):
    """The same 256 x 64 TF as a SESSION in steady state: a one-shot capture spends its first 15 TF locking in and emits 4 (T - 15) frames (dab2eti's
    rule, dab.c / misc.c), a receiver that has been running emits 4 per TF.  Every stream is one linear buffer in HBM holding the 64-TF capture
    `reps + 3` times back to back (the capture is frame-aligned, so synchronisation and FIC lock are kept across the seams; the payload across a seam is
    not meaningful, the work is the same), and the session reads it in place (dabhip_stream_feed_resident: no copy), 64 TF further per feed.
    IQ and ETI stay in HBM, as for `value`."""
    reps = max(3, min(args.steps, 10))
    seg = tensors[0].numel()
    long = [t.repeat(reps + 3) for t in tensors]
    base = [t.data_ptr() for t in long]
    st = dab.Stream(len(long), device=dev.index or 0)
    first = st.feed_resident(base, [seg] * len(long))
    for k in (2, 3):
        n = st.feed_resident(base, [k * seg] * len(long))
    torch.cuda.synchronize(dev)
    stage = {}
    t0 = time.perf_counter()
    for k in range(4, reps + 4):
        n = st.feed_resident(base, [k * seg] * len(long))
        for kk, v in st.stage_ms().items():
            stage[kk] = stage.get(kk, 0.0) + v / reps
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    bad = sum(1 for b in range(len(long)) if st.status(b))
    st.close()
    del long
    return {"what": "a session in steady state, streams read in place (dabhip_stream_feed_resident): %d streams x %d TF per feed, every TF emits 4 frames "
                    "(the one-shot decode of `value` emits 4 (T - 15))" % (len(tensors), args.tfs),
            "value": n / dt, "unit": "ETI frames/s", "x_realtime": n / dt / REALTIME_FPS, "ms_per_segment": 1e3 * dt, "eti_frames_per_segment": n,
            "eti_frames_first_segment": first, "streams_flagged": bad, "segments_timed": reps, "stage_ms_per_segment": {k: round(v, 4) for k, v in stage.items()},
            "through_windows": "dabhip_stream_feed with device pointers copies each segment into the session's windows first: 15.1 ms per segment (profiles/r04_session_steady.json)"}


def single_ensemble(dab, torch, dev, eng, tensors, args):
    """BASELINE configs[1]: ONE Mode-I ensemble on one MI355X (the reference's only mode of use: one live ensemble, one demod thread,
    dab2eti.c:60-115,237).  (a) the batch entry with B = 1, IQ resident: a step is ~30 launches and two host hand-offs whatever the batch, and
    a code word's latency, not throughput, bounds the decoders -- so small decodes run one WAVE per code word (k_vitwave.hip);
    (b) a one-stream session fed one transmission frame (96 ms of signal) at a time from page-locked host memory, frames back on the host:
    what a live receiver sees per segment."""
    import numpy as np
    ptr, size = [tensors[0].data_ptr()], [tensors[0].numel()]
    call = eng.marshal(ptr, size)
    for _ in range(3):
        frames = eng.decode_marshalled(call)
    torch.cuda.synchronize(dev)
    reps, stage = 20, {}
    t0 = time.perf_counter()
    for _ in range(reps):
        frames = eng.decode_marshalled(call)
        for k, v in eng.stage_ms().items():
            stage[k] = stage.get(k, 0.0) + v
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / reps
    out = {"workload": "BASELINE configs[1]: one synthetic Mode-I ensemble x %d TF (12 sub-channels, 1136 kbit/s), IQ resident in HBM" % args.tfs,
           "value": frames / dt, "unit": "ETI frames/s", "x_realtime": frames / dt / REALTIME_FPS, "ms_per_decode": 1e3 * dt, "eti_frames_per_decode": frames,
           "stage_ms": {k: round(v / reps, 4) for k, v in stage.items() if not k.startswith("h2d")},
           "decoder_form": "one wave per code word (k_vitwave.hip) for the %d MSC code words and %d FIC blocks of this decode" % (12 * frames, 4 * (args.tfs - 1)),
           "sync_schedule": "K1 with the look-ahead pass (k_sync.hip: sync_ahead_kernel; default for <= 4 streams): %d of the chain's calls per decode took their estimators from "
                            "its table; with the plain chain this decode's sync stage is 0.58 ms (profiles/r05_batch_curve_plain_chain.json)" % round(stage.get("sync_spec_calls", 0.0) / reps),
           "batch_curve": "profiles/r06_batch_curve.json (B = 1 .. 256, tools/batch_curve.py; the plain chain beside the look-ahead schedule: r05_batch_curve_plain_chain.json)"}
    ntf = min(args.tfs, 40)
    hb = dab.HostBuffer(ntf * dab.TF_BYTES)
    assert dab.lib().dabhip_device_copy(hb.ptr, tensors[0].data_ptr(), ntf * dab.TF_BYTES, 0) == 0
    st = dab.Stream(1, device=dev.index or 0)
    lat, total = [], 0
    for k in range(ntf):
        t0 = time.perf_counter()
        n = st.feed_ptrs([hb.ptr + k * dab.TF_BYTES], [dab.TF_BYTES])
        if n:
            st.eti(0)
        lat.append(1e3 * (time.perf_counter() - t0))
        total += n
    st.close()
    hb.free()
    warm = lat[18:]
    out["live_session"] = {"segment": "1 TF = 393,216 B = 96 ms of signal, host -> device -> ETI frames back on the host", "segments": ntf, "eti_frames": total,
                           "ms_per_segment_median": float(np.median(warm)), "ms_per_segment_max": float(max(warm)), "x_realtime": 96.0 / float(np.median(warm))}
    return out


def rooflines(prof, args, world, frames_rank, ntf_rank, stage, fft, stream_ceiling):
    """roofline (K2, HBM), roofline_viterbi (VALU issue), roofline_ofdm_fused (VALU issue / LDS): live times of this run x per-unit figures
    read from the tracked profile; every input names the CSV cells it comes from."""
    out = {}
    decodes = prof.meta("full_decodes")
    same = prof.meta("streams_per_gpu") == args.streams and prof.meta("tf_per_stream") == args.tfs
    K2, VIT, FUSED = "ofdm_fft_kernel<false>", "viterbi_fused_kernel<1>", "ofdm_demap_kernel<false>"
    if fft:
        launches, tfs, ms = fft
        achieved = FFT_BYTES_PER_TF * tfs / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        k2_tfs = prof.meta("k2_roofline_tfs")                  # TFs ofdm_fft_kernel<false> transformed in the profiled run
        traffic_per_tf = (2.0 * prof.cell("fetch", K2, "FETCH_SIZE") + prof.cell("write", K2, "WRITE_SIZE")) * 1024.0 / k2_tfs
        out["roofline"] = {
            "bound": "hbm", "kernel": "ofdm_fft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_per_tf * tfs / max(launches, 1),
            "traffic_note": "HBM bytes per launch FROM THE TRACKED PROFILE, not from this run: (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 / TFs "
                            "(FETCH_SIZE doubled: gfx950 note in MI355X_MICROARCH.md) = %.0f B per TF = %.3f x algorithmic; cells %s, %s, meta k2_roofline_tfs"
                            % (traffic_per_tf, traffic_per_tf / FFT_BYTES_PER_TF, prof.ref("fetch", K2, "FETCH_SIZE"), prof.ref("write", K2, "WRITE_SIZE")),
            "measured_outside_the_timed_step": True,
            "how": "dabhip_engine_fft_roofline: K2 ALONE over the frame list of the timed step (same resident IQ, %d TF per launch), HIP events on the "
                   "engine's stream around every launch.  The timed step itself runs the fused transform + demap kernel (roofline_ofdm_fused), which "
                   "never writes the spectra; `frac` is this kernel's figure, not the pipeline's" % (tfs // max(launches, 1)),
            "stream_ceiling": dict(stream_ceiling or {}, unit="GB/s",
                                   note="bare grid-stride kernels of this library on this device, measured in this run, 4 GiB buffers (k_probe.hip): "
                                        "fill = write only, copy = 1 read : 1 written, k2_mix = 1 byte read per 4 written with K2's 16-byte nontemporal stores"),
            "achieved_vs_measured_copy_ceiling": (achieved / stream_ceiling["copy"]) if (stream_ceiling or {}).get("copy") else None,
            "achieved_vs_k2_mix_stream": (achieved / stream_ceiling["k2_mix"]) if (stream_ceiling or {}).get("k2_mix") else None,
            "launches": launches, "tf_per_launch": tfs / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
            "algorithmic_bytes_per_tf": FFT_BYTES_PER_TF}
    if stage.get("viterbi", 0) > 0 and not args.subchannels and not args.soft:
        # MSC Viterbi (hard decisions): bound by VALU issue.  Per wave (64 code words) and trellis step: insts = SQ_INSTS_VALU / wave-steps of the
        # profiled run (FIC and MSC launches of all its decodes); 64 of them are 2-cycle v_add_u32, the rest 4-cycle forms; the SIMDs' clock
        # under this kernel = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same pass.
        ws_prof = decodes * prof.meta("viterbi_wave_steps_per_decode") + prof.meta("viterbi_wave_steps_setup")
        insts = prof.cell("clk", VIT, "SQ_INSTS_VALU") / ws_prof
        clock = prof.cell("clk", VIT, "GRBM_GUI_ACTIVE") / XCDS / prof.cell("clk", VIT, "DURATION_NS")       # GHz
        cycles = CYC_SIMPLE * VIT_ADDS_PER_STEP + CYC_VOP3 * (insts - VIT_ADDS_PER_STEP)
        ws = wave_steps(frames_rank, 0)                                                   # the MSC launch of one step on this rank
        t = stage["viterbi"] * 1e-3
        rec_read = 2.0 * prof.cell("fetch", VIT, "FETCH_SIZE") * 1024.0 / decodes          # FETCH_SIZE doubled: gfx950 note in MI355X_MICROARCH.md
        rec_written = prof.cell("write", VIT, "WRITE_SIZE") * 1024.0 / decodes
        rec_bytes = rec_read + rec_written
        io_bytes = frames_rank * (1728 * 4 + 27264 // 8)                                  # grouped received bits in, decoded sub-channel bytes out
        out["roofline_viterbi"] = {
            "kernel": VIT, "bound": "valu issue", "achieved": cycles * ws / t / 1e9, "peak": SIMDS * clock, "unit": "G issue cycles/s",
            "frac": cycles * ws / t / 1e9 / (SIMDS * clock),
            "frac_if_every_instruction_took_2_cycles": CYC_SIMPLE * insts * ws / t / 1e9 / (SIMDS * clock),
            "frac_from_counters": {"value": counter_busy(prof, VIT), "from": prof.ref("f1", VIT, "SQ_ACTIVE_INST_VALU") + " x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) of the same pass",
                                   "note": "the profiled run's own busy fraction (SQ_ACTIVE_INST_VALU is quad-granular), beside the priced figure of THIS run"},
            "cost_model": "64 v_add_u32 per step at 2 cycles, every other instruction (v_pk_max_u16, v_perm_b32, LDS-table words, ...) at 4: tools/ubench/valu_rates.hip measures 2.4 .. 2.9 / 4.2 .. 4.5 "
                          "clocks; MI355X_MICROARCH.md prices every wave64 VALU instruction at 2 cycles, which gives the lower figure beside it",
            "inputs": {
                "valu_insts_per_wave_step": {"value": insts, "from": prof.ref("clk", VIT, "SQ_INSTS_VALU") + " / (meta full_decodes x viterbi_wave_steps_per_decode + viterbi_wave_steps_setup)"},
                "effective_clock_ghz": {"value": clock, "from": prof.ref("clk", VIT, "GRBM_GUI_ACTIVE") + " / 8 XCDs / DURATION_NS of the same pass (dense VALU: the chip clocks below its 2.4 GHz)"},
                "issue_cycles_per_wave_step": {"value": cycles, "from": "%d v_add_u32 x %.0f cycles + the other instructions x %.0f cycles" % (VIT_ADDS_PER_STEP, CYC_SIMPLE, CYC_VOP3)},
                "wave_steps_per_launch": ws, "simds": SIMDS, "avg_ms": stage["viterbi"], "profile_matches_this_workload": same},
            "survivor_traffic": {"bytes_per_step": rec_bytes, "written": rec_written, "read_back": rec_read, "x_stage_io": rec_bytes / io_bytes, "GBps": rec_bytes / t / 1e9, "frac_of_hbm_peak": rec_bytes / t / 1e9 / HBM_PEAK_GBS,
                                 "from": "%s, %s (x2, + WRITE_SIZE) / meta full_decodes" % (prof.ref("fetch", VIT, "FETCH_SIZE"), prof.ref("write", VIT, "WRITE_SIZE")),
                                 "note": "one lane per code word with a full traceback: 64 B of survivor records per 8 steps and code word written, the 16-byte part "
                                         "holding the path's state read back; the stage's own input + output is %.2f GB per step.  Measured with the records kept in L2 "
                                         "(build DABHIP_VIT_NOSTORE): the HBM traffic costs 0.3 of the 5.0 ms, the chain-back 0.15" % (io_bytes / 1e9)}}
    if stage.get("fft", 0) > 0 and not args.two_kernel_ofdm and not args.soft and not args.no_parity_guard:
        # the default OFDM stage: ofdm_demap_kernel (guarded build): not HBM-bound -- VALU issue with LDS round trips and barriers per symbol
        tr_prof = decodes * prof.meta("ofdm_transforms_per_decode") + prof.meta("ofdm_transforms_setup")
        insts = prof.cell("clk", FUSED, "SQ_INSTS_VALU") / (4.0 * tr_prof)                   # per wave (4 per transform)
        clock = prof.cell("clk", FUSED, "GRBM_GUI_ACTIVE") / XCDS / prof.cell("clk", FUSED, "DURATION_NS")
        transforms = 77 * ntf_rank                                                           # symbols 0..3 (FIC launch) + 3..75 (MSC launch) per TF
        t = stage["fft"] * 1e-3
        hbm = FUSED_BYTES_PER_TF * ntf_rank / t / 1e9
        tfs_prof = decodes * prof.meta("tf_per_decode") + prof.meta("tf_setup")
        mix = json.load(open(PROFILE_FUSED_MIX))
        # issue cycles of one 2048-point transform + demap per wave: the symbol loop's static mix on its steady-state path (weights in the file),
        # and what the measured count has beyond that (the guard's per-bin repeats, prologue) at the repeat blocks' mean cost
        cyc_transform = mix["issue_cycles_per_unit"] + max(0.0, insts - mix["valu_per_unit"]) * mix["remainder_cycles_per_valu"]
        out["roofline_ofdm_fused"] = {
            "kernel": FUSED + " (transform + DQPSK + demap + de-interleave scatter, parity guard's test inline)", "bound": "valu issue",
            "achieved": 4.0 * cyc_transform * transforms / t / 1e9, "peak": SIMDS * clock,
            "unit": "G issue cycles/s", "frac": 4.0 * cyc_transform * transforms / t / 1e9 / (SIMDS * clock),
            "frac_from_counters": {"value": counter_busy(prof, FUSED), "from": prof.ref("f1", FUSED, "SQ_ACTIVE_INST_VALU") + " x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) of the same pass"},
            "valu_issue": {"insts_per_wave_per_transform": insts, "from": prof.ref("clk", FUSED, "SQ_INSTS_VALU") + " / 4 waves / transforms of the profiled run",
                           "issue_cycles_per_wave_per_transform": cyc_transform,
                           "mix": {"file": os.path.relpath(PROFILE_FUSED_MIX, ROOT), "static_per_transform": mix["per_unit"], "static_valu": mix["valu_per_unit"],
                                   "static_cycles": mix["issue_cycles_per_unit"], "remainder_cycles_per_valu": mix["remainder_cycles_per_valu"], "cost_model": mix["cost_model"]},
                           "effective_clock_ghz": clock, "clock_from": prof.ref("clk", FUSED, "GRBM_GUI_ACTIVE") + " / 8 / DURATION_NS",
                           "note": "instruction mix of the symbol loop from the compiler's assembly (tools/isa_mix.py: packed fp32 5, VOP3 forms 4, plain 32-bit forms 2 cycles), "
                                   "scaled to the measured instruction count"},
            "lds": {"bank_conflict_pct": 100.0 * prof.cell("f2", FUSED, "SQ_LDS_BANK_CONFLICT") / prof.cell("f2", FUSED, "SQ_LDS_IDX_ACTIVE"),
                    "from": prof.ref("f2", FUSED, "SQ_LDS_BANK_CONFLICT") + " / SQ_LDS_IDX_ACTIVE",
                    "wait_inst_lds_pct_of_wave_cycles": 100.0 * prof.cell("f1", FUSED, "SQ_WAIT_INST_LDS") / prof.cell("clk", FUSED, "SQ_WAVE_CYCLES")},
            "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "algorithmic_bytes_per_tf": FUSED_BYTES_PER_TF,
                    "write_amplification": prof.cell("write", FUSED, "WRITE_SIZE") * 1024.0 / (28800.0 * tfs_prof),
                    "from": prof.ref("write", FUSED, "WRITE_SIZE") + " x 1024 / (28,800 B x TFs of the profiled run)",
                    "note": "NOT the bound of this kernel (it never writes the spectra): stated so that nobody reads it as one"},
            "transforms_per_step": transforms, "avg_ms": stage["fft"], "profile_matches_this_workload": same}
    # `roofline` is SURVEY 8(d)'s number -- K2 by itself -- and K2 is not a kernel of the timed step.  So that a reader of `roofline` alone (the driver's
    # record keeps that object) sees what the step's own dominant kernels achieve, their summaries ride inside it.
    if "roofline" in out:
        inside = {}
        for key, name in (("roofline_ofdm_fused", "ofdm_demap_kernel"), ("roofline_viterbi", "viterbi_fused_kernel")):
            if key in out:
                r = out[key]
                inside[name] = {"bound": r["bound"], "frac": r["frac"], "frac_from_counters": r["frac_from_counters"]["value"], "avg_ms_per_step": r["avg_ms"] if "avg_ms" in r else r["inputs"]["avg_ms"],
                                "details": key}
        if "roofline_ofdm_fused" in out:
            inside["ofdm_demap_kernel"]["hbm_frac"] = out["roofline_ofdm_fused"]["hbm"]["frac"]
        if "roofline_viterbi" in out:
            inside["viterbi_fused_kernel"]["survivor_traffic_frac_of_hbm_peak"] = out["roofline_viterbi"]["survivor_traffic"]["frac_of_hbm_peak"]
        out["roofline"]["in_timed_step"] = dict(inside, note="the kernels the timed step DOES run (K2's work is done inside ofdm_demap_kernel, which never writes the spectra): "
                                                             "VALU-issue-bound both; `frac` above is K2 alone against the HBM peak")
    return out


def run_rank(args, coord):
    rank, world = coord.rank, coord.world
    from dabtools_amd import shard
    mine = shard.shard_streams(world * args.streams, world, rank)          # global stream indices of this rank
    rank_info = {"rank": rank, "first_stream": mine[0], "last_stream": mine[-1], "streams": len(mine),
                 "first_seed": shard.stream_seed(2, mine[0]), "host_threads": os.environ.get("DABHIP_HOST_THREADS", "auto"),
                 "visible_devices": os.environ.get("ROCR_VISIBLE_DEVICES"), "cpu_budget": dict(zip(("affinity", "cfs_quota"), shard.cpu_budget()))}
    prof = None
    if rank == 0 and not args.dry_run and not args.profile_pass:
        prof = load_profile()                                              # fails loudly BEFORE any GPU time is spent

    ntf_rank = 0
    if args.dry_run:
        if os.environ.get("DABHIP_BENCH_DIE_RANK") == str(rank):           # test knob: this rank dies before the first barrier
            sys.stderr.write("rank %d: dying on request\n" % rank)
            os._exit(7)
        coord.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * args.steps * (1 + rank))                         # ranks finish at different times: MAX is what counts
        elapsed = time.perf_counter() - t0
        coord.barrier()
        frames = 4 * (args.tfs - 15) * len(mine)
        stage, fft, fused_off, extra = {}, None, None, {}
        # (no GPU: a stand-in identity, so that the distinct-devices check below runs in the CPU suite too)
        rank_info["device"] = {"index": 0 if os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1" else rank,
                               "pci_bus_id": os.environ.get("DABHIP_BENCH_DRY_BUS_ID") or "dry-run:%02d" % (0 if os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1" else rank),
                               "name": "none (dry run)"}                   # (DABHIP_BENCH_DRY_BUS_ID: test knob -- every rank claims this device)
    else:
        import torch
        import dabtools_amd as dab
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1":
            local_rank = 0                                                 # test knob: all ranks share GPU 0 (exercises the N-rank path on a 1-GPU box)
        if os.environ.get("DABHIP_BENCH_DEVICE", "").isdigit():
            local_rank = int(os.environ["DABHIP_BENCH_DEVICE"])            # launch_ranks made this rank's GPU the only visible one
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        try:                                                               # which physical GPU this rank decodes on (checked across ranks by rank 0)
            bus_id, dev_name = dab.device_identity(local_rank)
        except dab.DabhipError as e:                                       # (never seen; an identity that cannot be read must not take the measurement down)
            bus_id, dev_name = None, "unknown (%s)" % e
        rank_info["device"] = {"index": local_rank, "pci_bus_id": bus_id, "name": dev_name}
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
        setup_tf = 20
        setup_frames = eng.decode_device(ptrs[:2], [min(s, setup_tf * 393216) for s in sizes[:2]])   # loads the code objects (tiny, untimed, part of set-up)

        def barrier():
            torch.cuda.synchronize(dev)
            coord.barrier()
            torch.cuda.synchronize(dev)

        frames = 0
        call = eng.marshal(ptrs, sizes)                                    # the C argument arrays, built once (a C caller has them anyway)
        for _ in range(args.warmup):
            frames = eng.decode_marshalled(call)
        barrier()
        t0 = time.perf_counter()
        stage = {}
        for _ in range(args.steps):
            frames = eng.decode_marshalled(call)
            for k, v in eng.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v
        barrier()
        elapsed = time.perf_counter() - t0
        stage = {k: v / args.steps for k, v in stage.items() if not k.startswith("h2d")}   # (sync_fp64_calls: a count per step, not ms)
        ntf_rank = eng.guard_stats()[1] // 230400 if not (args.soft or args.no_parity_guard) else 0

        fft = None
        fused_off = None
        extra = {}
        if rank == 0:
            flagged, decisions = eng.guard_stats()
            level = 0 if (args.no_parity_guard or args.soft) else eng.parity_guard_level()
            extra["parity_guard"] = {"on": level > 0, "level": level, "level_is": {0: "off", 1: "measured band", 2: "proven band (rigorous forward-error bound: DESIGN.md section 3)"}[level],
                                     "constants": dict(zip(("bin_error_over_l2", "product_rounding_over_l1l1"), dab.guard_constants(level))) if level else None,
                                     "decisions_per_step": decisions, "redecided_in_fp64_per_step": flagged,
                                     "note": "hard decisions whose fp32 margin lies inside the error band are re-decided in fp64 from the int8 samples "
                                             "(k_parity.hip); both levels audited on the shipping kernel: profiles/r06_decision_audit.json; price by level: profiles/r06_guard_levels.json"}
            # K2 = ofdm_fft_kernel by itself, on the same resident IQ and the frame list of the step just timed
            fft = eng.fft_roofline(max(3, min(args.steps, 10)))
            try:
                extra["stream_ceiling"] = dab.stream_ceiling(local_rank, 4 << 30, 3)
            except dab.DabhipError as e:
                extra["stream_ceiling"] = {"error": str(e)}
            # what the profile passes of tools/refresh_profiles.sh need to know about this run (see rooflines())
            k2_launches, k2_tfs, _ = fft
            reps = max(3, min(args.steps, 10)) + 1                                         # fft_roofline: one untimed pass first
            tf_step = k2_tfs // max(reps - 1, 1)
            extra["profile_meta"] = {"full_decodes": args.warmup + args.steps, "streams_per_gpu": args.streams, "tf_per_stream": args.tfs,
                                     "viterbi_wave_steps_per_decode": wave_steps(frames, tf_step), "viterbi_wave_steps_setup": wave_steps(setup_frames, 2 * (setup_tf - 1)),
                                     "ofdm_transforms_per_decode": 77 * tf_step, "ofdm_transforms_setup": 77 * 2 * (setup_tf - 1),
                                     "tf_per_decode": tf_step, "tf_setup": 2 * (setup_tf - 1), "k2_roofline_tfs": tf_step * reps}
            ntf_rank = tf_step
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
                if args.snr < 100.0:                                           # noisy input: what the OTHER guard level costs (on clean input nothing is listed at either)
                    other = 1 if level == 2 else 2
                    eng.set_parity_guard(other)
                    eng.decode_device(ptrs, sizes)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        f4 = eng.decode_device(ptrs, sizes)
                    torch.cuda.synchronize(dev)
                    e4 = time.perf_counter() - t1
                    extra["parity_guard_other_level_variant"] = {"level": other, "value": f4 * args.steps / e4, "unit": "ETI frames/s", "ms_per_step": 1e3 * e4 / args.steps,
                                                                 "redecided_in_fp64_per_step": eng.guard_stats()[0],
                                                                 "stage_ms_per_step": {k: v for k, v in eng.stage_ms().items() if k in ("fft", "demap")}}
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
            if not args.soft and not args.subchannels and not args.no_variants:
                try:
                    extra["single_ensemble"] = single_ensemble(dab, torch, dev, eng, tensors, args)
                except Exception as e:                                        # a side measurement: never takes `value` down with it
                    extra["single_ensemble"] = {"error": "%s: %s" % (type(e).__name__, e)}
            extra["data"] = (("synthetic (%d distinct ensembles per GPU tiled to %d streams, host modulator)" % (ndistinct, args.streams)) if args.host_synth
                             else ("synthetic (%d distinct ensembles per GPU, device-side modulator, %.1f s)" % (args.streams, t_gen)))
            if args.snr < 100.0:
                eng.decode_device(ptrs, sizes)
                extra["payload"] = payload_stats(dab, eng, rank * args.streams, min(16, args.streams), args.tfs)
            if not args.no_h2d and world == 1 and not args.soft and not args.subchannels and not args.host_synth:
                eng.close()                                                   # its buffers (survivor records ...) make room for the session's windows
                eng = None
                try:
                    extra["steady_state_session"] = steady_state(dab, torch, dev, tensors, args)
                except Exception as e:
                    extra["steady_state_session"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    extra["h2d_inclusive"] = h2d_inclusive(dab, local_rank, tensors, sizes, frames, args)
                except Exception as e:                                        # a side measurement (6.4 GB of page-locked host memory): never takes `value` down with it
                    extra["h2d_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    extra["h2d_inclusive"]["cli"] = cli_leg(min(args.streams, 128), args.tfs)
                except Exception as e:
                    extra["h2d_inclusive"]["cli"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not args.no_cpu_baseline and world == 1:
                if args.snr < 100.0:                                          # configs[4]: CPU BER on the 16 streams the GPU payload was checked on
                    extra["cpu_baseline"] = cpu_baseline(tensors, args.tfs, min(16, args.streams), ber_first_stream=rank * args.streams, snr=args.snr)
                else:
                    extra["cpu_baseline"] = cpu_baseline(tensors, args.tfs, args.cpu_sample)

    rank_info["host_ms_per_step"] = {k: round(stage[k], 4) for k in ("control", "host_worklist", "host_setup", "host_frames", "wall") if k in stage}
    rows = coord.gather({"info": rank_info, "elapsed": elapsed, "frames": frames})
    if rank == 0:
        # N ranks must have sat on N distinct GPUs, or the aggregate is not an N-GPU figure: shown by the PCI bus ids, refused otherwise -- except in
        # the declared one-GPU rehearsal (DABHIP_BENCH_ONE_DEVICE=1), which the line then says of itself
        bus_ids = [r["info"].get("device", {}).get("pci_bus_id") for r in rows]
        distinct = len(set(bus_ids))
        rehearsal = os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1"
        if None in bus_ids:                                                 # identities unreadable: nothing can be shown either way -- say so in the line
            distinct = None
        elif distinct != world and not rehearsal:
            sys.stderr.write("bench.py: %d ranks decoded on %d distinct devices (%s): not a %d-GPU measurement -- no line printed "
                             "(DABHIP_BENCH_ONE_DEVICE=1 declares a one-GPU rehearsal)\n" % (world, distinct, ", ".join(map(str, bus_ids)), world))
            coord.close()
            sys.exit(3)
        elapsed_max = max(r["elapsed"] for r in rows)
        frames_step = sum(r["frames"] for r in rows)
        value = frames_step * args.steps / elapsed_max
        if args.soft:
            ofdm_stage = "K2 + K2b (two kernels), soft values" if args.two_kernel_ofdm else "fused transform + demap, 4-bit soft values (k_fused.hip, DABHIP_FUSED_SOFT build)"
        else:
            ofdm_stage = "K2 + K2b (two kernels)" if args.two_kernel_ofdm else "fused transform + demap (default)"
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
                       "eti_frames_per_step": frames_step, "ofdm_stage": ofdm_stage,
                       "sharding": "independent ensembles, %d per GPU, stream s on rank s // %d, no collective" % (args.streams, args.streams)},
            "ranks": [dict(r["info"], elapsed_s=r["elapsed"], eti_frames_per_step=r["frames"]) for r in rows],
            "devices": {"distinct_pci_bus_ids": distinct, "pci_bus_ids": bus_ids, "names": sorted(set(r["info"].get("device", {}).get("name") for r in rows)),
                        "rehearsal_on_one_device": rehearsal and world > 1},
        }
        if rehearsal and world > 1:
            out["devices"]["note"] = "DABHIP_BENCH_ONE_DEVICE=1: all %d ranks share one GPU -- a rehearsal of the %d-rank path, NOT a %d-GPU figure" % (world, world, world)
        if args.dry_run:
            out["dry_run"] = True
        if stage:
            out["stage_ms_per_step"] = stage
        if prof is not None:
            out.update(rooflines(prof, args, world, frames, ntf_rank, stage, fft, extra.get("stream_ceiling")))
        elif fft:
            launches, tfs, ms = fft                       # --profile-pass: the K2 figure that needs no profile input
            achieved = FFT_BYTES_PER_TF * tfs / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out["roofline"] = {"bound": "hbm", "kernel": "ofdm_fft_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                               "traffic": None, "launches": launches, "tf_per_launch": tfs / max(launches, 1), "avg_launch_ms": ms / max(launches, 1),
                               "note": "--profile-pass: no profile-derived inputs"}
        if fused_off:
            out["two_kernel_ofdm_variant"] = fused_off
        if args.subchannels:
            out["config"]["subchannel_filter"] = args.subchannels
        if args.snr < 100.0:
            out["config"]["snr_db"] = args.snr
            out["config"]["decisions"] = "soft (4-bit)" if args.soft else "hard"
        for k in ("parity_guard", "parity_guard_off_variant", "parity_guard_other_level_variant", "single_ensemble", "steady_state_session", "payload", "h2d_inclusive", "cpu_baseline", "profile_meta"):
            if k in extra:
                out[k] = extra[k]
        print(json.dumps(out))
        sys.stdout.flush()
    coord.close()


def run_in_process(args):
    """`--in-process`: the N GPUs of the node driven from THIS process through dabhip_multi (one engine + one host thread per device, streams
    dealt in contiguous slices, no exchange) instead of one rank process per GPU.  Same workload, same timing rule; one JSON line."""
    import torch
    import dabtools_amd as dab
    from dabtools_amd import payload
    n = args.gpus
    one_device = os.environ.get("DABHIP_BENCH_ONE_DEVICE") == "1"
    devices = [0 if one_device else i for i in range(n)]
    tensors = []
    for sl, d in enumerate(devices):
        torch.cuda.set_device(d)
        cfgs = [payload.bench_cfg(dab, sl * args.streams + i, args.snr) for i in range(args.streams)]
        ts = [torch.empty(dab.synth_bytes(c, args.tfs), dtype=torch.uint8, device=torch.device("cuda", d)) for c in cfgs]
        dab.synth_generate_device(cfgs, args.tfs, [t.data_ptr() for t in ts], d)
        tensors += ts
    for d in set(devices):
        torch.cuda.synchronize(d)
    multi = dab.Multi(devices)
    if args.soft:
        multi.set_soft(True)
    ptrs, sizes = [t.data_ptr() for t in tensors], [t.numel() for t in tensors]
    frames = 0
    for _ in range(args.warmup):
        frames = multi.decode_device(ptrs, sizes)
    for d in set(devices):
        torch.cuda.synchronize(d)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frames = multi.decode_device(ptrs, sizes)
    for d in set(devices):
        torch.cuda.synchronize(d)
    elapsed = time.perf_counter() - t0
    value = frames * args.steps / elapsed
    per_slice = []
    for i in range(n):
        st = multi.engine(i).stage_ms()
        per_slice.append({"slice": i, "device": devices[i], "wall_ms": round(multi.wall_ms(i), 3), "control_ms": round(st["control"], 3),
                          "host_worklist_ms": round(st["host_worklist"], 3), "eti_frames": sum(multi.eti_count(b) for b in range(i * args.streams, (i + 1) * args.streams))})
    ids = [dab.device_identity(d)[0] for d in devices]
    if len(set(ids)) != n and not one_device:
        sys.exit("bench.py: %d slices on %d distinct devices: not a %d-GPU measurement" % (n, len(set(ids)), n))
    print(json.dumps({
        "metric": "ETI frames/s (24 ms each), Mode-I batch, synthetic IQ resident in HBM", "value": value, "unit": "ETI frames/s", "x_realtime": value / REALTIME_FPS,
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 sync / f32 OFDM / u16 ACS / u8 ETI", "data": "synthetic (%d distinct ensembles per slice, device-side modulator)" % args.streams,
        "config": {"workload": workload_text(args), "streams_per_gpu": args.streams, "tf_per_stream": args.tfs, "eti_frames_per_step": frames,
                   "launch": "in-process: dabhip_multi over devices %s (one engine + host thread per entry)" % devices,
                   "sharding": "independent ensembles, %d per slice, stream s on slice s // %d, no collective" % (args.streams, args.streams)},
        "slices": per_slice,
        "devices": {"distinct_pci_bus_ids": len(set(ids)), "pci_bus_ids": ids, "rehearsal_on_one_device": one_device and n > 1}}))


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
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-fed measurement (h2d_inclusive: 6.4 GB of page-locked host memory at the default size)")
    ap.add_argument("--h2d-segment-tfs", type=int, default=8, help="segment length of the host-fed streaming session, in transmission frames")
    ap.add_argument("--profile-pass", action="store_true", help="run under tools/refresh_profiles.sh: no profile-derived objects (they are being produced), emit profile_meta")
    ap.add_argument("--in-process", action="store_true", help="drive the --gpus N devices from this one process through dabhip_multi instead of N rank processes")
    args = ap.parse_args()

    if args.in_process:
        run_in_process(args)
        return
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world_env == 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))          # before any torch / HIP call in this process
    if os.environ.get("DABHIP_BENCH_PIPES") == "1" and world_env > 1:
        coord = Pipes(int(os.environ["RANK"]), world_env)
    elif world_env > 1:
        from dabtools_amd import shard
        os.environ.setdefault("DABHIP_HOST_THREADS", str(shard.host_threads_per_rank(world_env)))
        coord = Gloo()
    else:
        coord = Single()
    run_rank(args, coord)


if __name__ == "__main__":
    main()
