#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (GPU box): batches of synthetic captures with random seeds, CIF counters, start offsets,
amplitudes, noise levels and small carrier offsets, decoded by the batch engine in parity mode (hard decisions, guard on) and, per
stream, by the CPU oracle (oracle/or_replay, a process pool over the host cores): ETI bytes and per-call traces must be equal.

  python tools/stress_parity.py [--rounds 6] [--streams 48] [--tfs 24] [--workers 32]
Prints one JSON line (totals plus one record per capture: its parameters, calls, frames, resyncs, equal or not); exit code 1 on any
difference.  tests/test_gpu_parity_r3.py runs a one-minute cut of it under -m gpu.  Checker use of oracle/ only (like tests/)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def oracle_job(args):
    import oracle_lib as ol
    path, = args
    iq = np.load(path)
    eti, trace = ol.or_replay(iq)
    return eti, [(t.ok, t.read_frame, t.coarse_timeshift, t.fine_timeshift, t.coarse_freq_shift, t.fifo_count) for t in trace]


def reference_job(args):
    """The checker of --reference: dab2eti's own loop over the reference's REAL front end (oracle/_ref/libdabref_frontend.so: input_sdr.c, sdr_sync.c,
    sdr_fifo.c unmodified over hipFFTW) and REAL back end (oracle/_ref/libdabref.so).  read_frame is not visible from outside sdr_demod: -1."""
    import oracle_lib as ol
    path, = args
    iq = np.load(path)
    eti, calls, _ = ol.ref_frontend_replay(iq)
    return eti, [(c[0], -1, c[1], c[2], c[3], c[4]) for c in calls]


UEP_SIZE_CU = None
EEP_SIZE_MUL = [12, 8, 6, 4, 27, 21, 18, 15]        # CUs per n for protection levels 1-A .. 4-A, 1-B .. 4-B (dab_tables.c eeptable / fic.c:84-94)


def random_layout(dab, cfg, rng):
    """A random multiplex in place of the preset's: 1 .. 16 sub-channels, UEP rows and EEP levels / sizes drawn at random, random SubChIds, gaps between
    the sub-channels, inside 864 CUs and one ETI frame."""
    global UEP_SIZE_CU
    if UEP_SIZE_CU is None:
        t = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))["ueptable"]
        UEP_SIZE_CU = [(int(r[0]), int(r[1])) for r in t]           # (bitrate, size) of the 64 rows
    want = int(rng.integers(1, 17))
    ids = rng.permutation(64)[:want]
    cu, kbps, k = int(rng.integers(0, 40)), 0, 0
    cfg.nsub = 0
    for sid in sorted(int(x) for x in ids):
        if rng.integers(0, 2):
            idx = int(rng.integers(0, 64))
            rate, size = UEP_SIZE_CU[idx]
            slform, lev = 0, 0
        else:
            lev = int(rng.integers(0, 8))
            n = int(rng.integers(1, 13 if lev < 4 else 5))
            size, rate, slform, idx = n * EEP_SIZE_MUL[lev], n * (8 if lev < 4 else 32), 1, 0
        if cu + size > 864 or 3 * (kbps + rate) + 8 + 4 * (k + 1) + 4 + 96 + 8 > 6000:
            continue
        sc = cfg.sub[k]
        sc.id, sc.start_cu, sc.slform, sc.uep_index, sc.eep_protlev, sc.size_cu = sid, cu, slform, idx, lev, size
        cu += size + int(rng.integers(0, 6))
        kbps += rate
        k += 1
    if k == 0:                                                       # nothing fitted: one small sub-channel
        sc = cfg.sub[0]
        sc.id, sc.start_cu, sc.slform, sc.uep_index, sc.eep_protlev, sc.size_cu = 7, 0, 1, 0, 2, 6
        k = 1
    cfg.nsub = k
    return cfg


def run(rounds=6, streams=48, tfs=24, workers=None, seed=20261002, log=None, reference=False, harsh=False, layouts=False):
    """The sweep itself -> result dict (one record per capture under "cases")."""
    import dabtools_amd as dab
    workers = workers or min(32, os.cpu_count() or 1)
    job = reference_job if reference else oracle_job
    rng = np.random.default_rng(seed)
    tmp = "/tmp/stress_parity_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    eng = dab.Engine(0)
    total_frames = total_calls = 0
    bad, cases = [], []
    t0 = time.time()
    with mp.get_context("spawn").Pool(workers) as pool:
        for r in range(rounds):
            cfgs, paths, iqs, ragged = [], [], [], []
            for i in range(streams):
                if harsh:       # where the receiver is fragile: noise down to 5 dB (lock comes and goes), up to 1.4 carriers off tune (forced re-synchronisation), weak and clipped signals
                    snr = float(rng.choice([1000.0, 12.0, 9.0, 8.0, 7.0, 6.0, 5.0]))
                    cfg = dab.synth_preset(int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)), cif_count0=int(rng.integers(0, 5000)),
                                           skip_samples=int(rng.choice([0, int(rng.integers(1, 196608))])), snr_db=snr,
                                           amplitude=float(rng.choice([2.5, 1.0, 0.5, 0.2, 0.08])), cfo_hz=float(rng.choice([0.0, rng.uniform(-400, 400), rng.uniform(-1400, 1400)])))
                else:
                    snr = float(rng.choice([1000.0, 1000.0, 20.0, 14.0, 11.0, 9.0]))
                    cfg = dab.synth_preset(int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)), cif_count0=int(rng.integers(0, 5000)),
                                           skip_samples=int(rng.choice([0, 0, int(rng.integers(1, 196608))])), snr_db=snr,
                                           amplitude=float(rng.choice([1.0, 0.8, 0.5, 0.35])), cfo_hz=float(rng.choice([0.0, 0.0, rng.uniform(-400, 400)])))
                if layouts:
                    random_layout(dab, cfg, rng)
                ntf = int(tfs + rng.integers(0, 5))
                iq = dab.synth_generate(cfg, ntf)
                cut = 0
                if rng.integers(0, 4) == 0:                                   # ragged: not a whole number of 262144-byte calls
                    cut = int(rng.integers(1, 262144))
                    iq = iq[: iq.size - cut]
                p = os.path.join(tmp, "s%d.npy" % i)
                np.save(p, iq)
                cfgs.append(cfg); paths.append(p); iqs.append(iq); ragged.append(cut)
            pending = pool.map_async(job, [(p,) for p in paths], chunksize=1)
            eng.decode(iqs)
            got = [(eng.eti(b), eng.trace(b, iqs[b].size // 262144)[0]) for b in range(len(iqs))]
            want = pending.get()
            for b in range(len(iqs)):
                eti, trace = got[b]
                weti, wtrace = want[b]
                gtrace = [tuple(int(x) for x in t) for t in trace]
                if reference:
                    gtrace = [(t[0], -1) + t[2:] for t in gtrace]
                total_frames += len(weti)
                total_calls += len(wtrace)
                equal = eti.shape == weti.shape and np.array_equal(eti, weti) and gtrace == wtrace
                cases.append({"round": r, "stream": b, "preset": 0 if cfgs[b].nsub == 12 else 1, "subchannels": int(cfgs[b].nsub), "seed": int(cfgs[b].seed), "cif_count0": int(cfgs[b].cif_count0),
                              "skip_samples": int(cfgs[b].skip_samples), "snr_db": float(cfgs[b].snr_db), "amplitude": float(cfgs[b].amplitude),
                              "cfo_hz": round(float(cfgs[b].cfo_hz), 2), "bytes_cut": ragged[b], "calls": len(wtrace), "eti_frames": int(len(weti)),
                              "resyncs": int(sum(1 for t in wtrace[2:] if t[2] != 0)), "equal": bool(equal)})
                if not equal:
                    bad.append({"round": r, "stream": b, "seed": int(cfgs[b].seed), "snr_db": float(cfgs[b].snr_db), "frames": [int(len(eti)), int(len(weti))],
                                "trace_equal": gtrace == wtrace})
            if log:
                print("round %d: %d streams, %d ETI frames so far, %d differences" % (r, len(iqs), total_frames, len(bad)), file=log, flush=True)
    eng.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return {"layouts": "random multiplexes (1 .. 16 sub-channels, any UEP row / EEP level and size)" if layouts else "the two presets",
            "mix": "harsh (5 dB ... clean, up to 1.4 carriers off tune, amplitudes 0.08 ... 2.5)" if harsh else "default",
            "checker": "the reference itself: real front end over hipFFTW + real back end (oracle/_ref)" if reference else "oracle/or_replay",
            "rounds": rounds, "streams_per_round": streams, "eti_frames_compared": total_frames, "calls_compared": total_calls,
            "differences": bad, "seconds": round(time.time() - t0, 1), "seed": seed, "cases": cases}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--streams", type=int, default=48)
    ap.add_argument("--tfs", type=int, default=24)
    ap.add_argument("--workers", type=int, default=min(32, os.cpu_count() or 1))
    ap.add_argument("--seed", type=int, default=20261002)
    ap.add_argument("--reference", action="store_true", help="check against the reference's real front end + back end (needs oracle/_ref/libdabref_frontend.so; "
                                                             "every worker opens the GPU for hipFFTW: keep --workers small)")
    ap.add_argument("--harsh", action="store_true", help="the mix where the receiver is fragile: 5 dB ... clean, up to 1.4 carriers off tune, weak and clipped signals")
    ap.add_argument("--layouts", action="store_true", help="every capture its own random multiplex instead of one of the two presets")
    args = ap.parse_args()
    res = run(args.rounds, args.streams, args.tfs, args.workers, args.seed, log=sys.stderr, reference=args.reference, harsh=args.harsh, layouts=args.layouts)
    print(json.dumps(res))
    sys.exit(1 if res["differences"] else 0)


if __name__ == "__main__":
    main()
