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
    _uep_table()
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


def _uep_table():
    global UEP_SIZE_CU
    if UEP_SIZE_CU is None:
        t = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))["ueptable"]
        UEP_SIZE_CU = [(int(r[0]), int(r[1])) for r in t]           # (bitrate, size) of the 64 rows
    return UEP_SIZE_CU


def _rate_size(slform, idx, lev, size):
    if not slform:
        return _uep_table()[idx]
    n = size // EEP_SIZE_MUL[lev]
    return n * (8 if lev < 4 else 32), size


def mutate_multiplex(subs, rng, merged_rate):
    """One assemblable reconfiguration of the multiplex `subs` (tuples as SynthCfg.set_reconf takes them): a sub-channel added, moved, given another
    protection (UEP row / EEP level) or size, or dropped from the FIC (the reference keeps it: misc.c:14-21) -- one to three of those at once.
    merged_rate: {SubChId: largest bit rate signalled so far}; the frame the reference assembles carries EVERY id it ever heard, so their sum is
    what has to fit one ETI frame (misc.c:233,246-296)."""
    subs = [tuple(int(x) for x in t) for t in subs]

    def size_of(t):
        return _rate_size(t[2], t[3], t[4], t[5])[1]

    def fits(cand, skip=None):
        """the sub-channels of `cand` do not overlap and stay inside 864 CUs, and everything ever signalled still fits one ETI frame"""
        spans = sorted((t[1], t[1] + size_of(t)) for t in cand)
        if any(a < 0 or b > 864 for a, b in spans) or any(spans[i][1] > spans[i + 1][0] for i in range(len(spans) - 1)):
            return False
        rates = dict(merged_rate)
        for t in cand:
            rates[t[0]] = max(rates.get(t[0], 0), _rate_size(t[2], t[3], t[4], t[5])[0])
        return len(rates) <= 40 and 3 * sum(rates.values()) + 8 + 4 * len(rates) + 4 + 96 + 8 <= 6000

    def random_sub(sid, start):
        if rng.integers(0, 2):
            return (sid, start, 0, int(rng.integers(0, 64)), 0, 0)
        lev = int(rng.integers(0, 8))
        n = int(rng.integers(1, 13 if lev < 4 else 5))
        return (sid, start, 1, 0, lev, n * EEP_SIZE_MUL[lev])

    changed = []
    for _ in range(int(rng.integers(1, 4))):
        for attempt in range(40):
            op = str(rng.choice(["add", "move", "protect", "protect", "drop"]))
            cand = list(subs)
            if op == "add":
                free = sorted(set(range(64)) - {t[0] for t in cand})
                cand.append(random_sub(int(rng.choice(free)), int(rng.integers(0, 860))))
            elif op == "drop":
                if len(cand) < 2:
                    continue
                cand.pop(int(rng.integers(0, len(cand))))
            else:
                k = int(rng.integers(0, len(cand)))
                t = cand[k]
                cand[k] = (t[0], int(rng.integers(0, 860)), t[2], t[3], t[4], t[5]) if op == "move" else random_sub(t[0], t[1])
            if cand != subs and fits(cand):
                subs = cand
                changed.append(op)
                break
    for t in subs:
        merged_rate[t[0]] = max(merged_rate.get(t[0], 0), _rate_size(t[2], t[3], t[4], t[5])[0])
    return sorted(subs, key=lambda t: t[0]), changed


def random_reconfigurations(cfg, rng, ntf):
    """One or two reconfigurations inside the part of the capture that is emitted (lock after ~11 TF, 16 CIFs of ring): -> what was done, for the record"""
    merged = {}
    subs = cfg.multiplex()
    for t in subs:
        merged[t[0]] = _rate_size(t[2], t[3], t[4], t[5])[0]
    done = []
    at = int(rng.integers(58, max(59, 4 * ntf - 12)))
    for k in range(int(rng.integers(1, 3))):
        subs, ops = mutate_multiplex(subs, rng, merged)
        if not ops:
            break
        lead = int(rng.integers(0, 11))
        cfg.set_reconf(k, at, subs, fic_lead=lead)
        done.append({"at_cif": at, "fic_lead": lead, "ops": ops, "subchannels": len(subs)})
        at += int(rng.integers(1, 30))
    return done


def random_channel(cfg, rng):
    """One to three impairments of dabhip_channel_cfg switched on -> what was set, for the record"""
    kinds = list(rng.choice(["sro", "sro", "echo", "echo", "fade", "iq"], size=int(rng.choice([1, 1, 2, 3])), replace=False))
    ch, rec = cfg.channel, {}
    if "sro" in kinds:
        ch.sro_ppm = float(rng.choice([20.0, -20.0, 50.0, -50.0, 100.0, -100.0, rng.uniform(-110, 110), rng.uniform(-110, 110)]))
        rec["sro_ppm"] = round(ch.sro_ppm, 2)
    if "echo" in kinds:
        for e in range(int(rng.choice([1, 1, 2]))):
            ch.echo_delay[e] = int(rng.choice([50, 400, 600, int(rng.integers(1, 900))]))
            # (beyond the 504-sample prefix an echo is interference: mostly weak ones there; one in six echoes is stronger than the direct path)
            ch.echo_gain[e] = float(rng.uniform(0.1, 0.6) if ch.echo_delay[e] > 480 and rng.integers(0, 4) else rng.choice([rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.9, 1.4)]))
            ch.echo_phase[e] = float(rng.uniform(0, 1))
            ch.echo_doppler_hz[e] = float(rng.choice([0.0, 0.0, rng.uniform(-20, 20)]))
        rec["echo"] = [(int(ch.echo_delay[e]), round(ch.echo_gain[e], 3), round(ch.echo_phase[e], 3), round(ch.echo_doppler_hz[e], 2)) for e in range(2) if ch.echo_delay[e]]
    if "fade" in kinds:
        ch.fade_depth, ch.fade_hz = float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.3, 9.0))
        rec["fade"] = (round(ch.fade_depth, 3), round(ch.fade_hz, 2))
    if "iq" in kinds:
        ch.iq_gain_db, ch.iq_phase_deg = float(rng.uniform(-2.5, 2.5)), float(rng.uniform(-12, 12))
        rec["iq"] = (round(ch.iq_gain_db, 2), round(ch.iq_phase_deg, 2))
    return rec


def run(rounds=6, streams=48, tfs=24, workers=None, seed=20261002, log=None, reference=False, harsh=False, layouts=False, channel=False, reconf=False, checkpoint=None, audit_tfs=2):
    """The sweep itself -> result dict (one record per capture under "cases").  checkpoint: a path the result so far is written to after every round
    (a run that is cut off -- the reference as checker takes minutes per round -- then leaves the rounds it finished, "rounds_finished" says how many)."""
    import dabtools_amd as dab
    workers = workers or min(32, os.cpu_count() or 1)
    job = reference_job if reference else oracle_job
    rng = np.random.default_rng(seed)
    tmp = "/tmp/stress_parity_%d" % os.getpid()
    os.makedirs(tmp, exist_ok=True)
    eng = dab.Engine(0)
    audit_eng = dab.Engine(0) if audit_tfs > 0 else None
    audit = {"tfs_per_capture": audit_tfs, "decisions": 0, "disagree_guard_off": 0, "disagree_outside_band": 0, "listed_by_the_kernel": 0, "worst_bin_err_over_sqrt_energy": 0.0}
    total_frames = total_calls = 0
    bad, cases = [], []
    t0 = time.time()

    def summary(done):
        return {"layouts": "random multiplexes (1 .. 16 sub-channels, any UEP row / EEP level and size)" if layouts else "the two presets",
                "reconfiguration": "one or two assemblable changes of the multiplex mid-stream (add / move / re-protect / resize / drop from the FIC), FIC leading by 0 .. 10 CIFs" if reconf else "none",
                "channel": "one to three of: sample-rate offset (+-20, +-50, +-100, uniform +-110 ppm), one or two echoes (50 / 400 / 600 / random < 900 samples, gain 0.1 .. 1.4, Doppler), "
                           "slow fading (depth 0.3 .. 0.9, 0.3 .. 9 Hz), I/Q imbalance (+-2.5 dB, +-12 deg)" if channel else "ideal",
                "mix": "harsh (5 dB ... clean, up to 1.4 carriers off tune, amplitudes 0.08 ... 2.5)" if harsh else "default",
                "checker": "the reference itself: real front end over hipFFTW + real back end (oracle/_ref)" if reference else "oracle/or_replay",
                "rounds": rounds, "rounds_finished": done, "streams_per_round": streams, "eti_frames_compared": total_frames, "calls_compared": total_calls,
                "differences": bad, "decision_audit_of_the_fused_ofdm_kernel": audit, "seconds": round(time.time() - t0, 1), "seed": seed, "cases": cases}

    with mp.get_context("spawn").Pool(workers) as pool:
        for r in range(rounds):
            cfgs, paths, iqs, ragged, extras = [], [], [], [], []
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
                if layouts or (reconf and rng.integers(0, 2)):
                    random_layout(dab, cfg, rng)
                ntf = int(tfs + rng.integers(0, 5))
                extra = {}
                if reconf:
                    extra["reconf"] = random_reconfigurations(cfg, rng, ntf)
                if channel:
                    extra["channel"] = random_channel(cfg, rng)
                iq = dab.synth_generate(cfg, ntf)
                cut = 0
                if rng.integers(0, 4) == 0:                                   # ragged: not a whole number of 262144-byte calls
                    cut = int(rng.integers(1, 262144))
                    iq = iq[: iq.size - cut]
                p = os.path.join(tmp, "s%d.npy" % i)
                np.save(p, iq)
                cfgs.append(cfg); paths.append(p); iqs.append(iq); ragged.append(cut); extras.append(extra)
            pending = pool.map_async(job, [(p,) for p in paths], chunksize=1)
            eng.decode(iqs)
            got = [(eng.eti(b), eng.trace(b, iqs[b].size // 262144)[0]) for b in range(len(iqs))]
            # the parity guard's ground, on these very captures: the one-kernel OFDM stage the decode just ran against fp64 transforms of the
            # same samples, guard off -- how many raw fp32 decisions differ, and how many of those lie OUTSIDE the band the guard re-decides (must be 0).
            # Any 393216 bytes serve a numerical audit: `audit_tfs` consecutive frame-sized pieces from the middle of every capture.
            if audit_tfs > 0:
                pieces = [iq[(iq.size // 2) // 393216 * 393216:][: audit_tfs * 393216] for iq in iqs]
                pieces = [x[: x.size // 393216 * 393216] for x in pieces if x.size >= 393216]
                if pieces:
                    a = audit_eng.decision_audit(frames=np.concatenate(pieces), guard=False, fused=True)
                    audit["decisions"] += int(a["decisions"]); audit["disagree_guard_off"] += int(a["disagree"]); audit["disagree_outside_band"] += int(a["disagree_outside_guard"])
                    audit["listed_by_the_kernel"] += int(a["listed"])
                    audit["worst_bin_err_over_sqrt_energy"] = max(audit["worst_bin_err_over_sqrt_energy"], float(a["max_bin_err"]))
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
                              "resyncs": int(sum(1 for t in wtrace[2:] if t[2] != 0)), "short_reads": int(sum(1 for t in wtrace[3:] if t[2] + t[3] < 0)),
                              "layouts_in_eti": len({(int(f[5]), int(f[6]) & 7, bytes(f[7:8 + 4 * (int(f[5]) & 0x7f)])) for f in weti}), "equal": bool(equal), **extras[b]})
                if not equal:
                    bad.append({"round": r, "stream": b, "seed": int(cfgs[b].seed), "snr_db": float(cfgs[b].snr_db), "frames": [int(len(eti)), int(len(weti))],
                                "trace_equal": gtrace == wtrace})
            if log:
                print("round %d: %d streams, %d ETI frames so far, %d differences" % (r, len(iqs), total_frames, len(bad)), file=log, flush=True)
            if checkpoint:
                with open(checkpoint + ".tmp", "w") as f:
                    f.write(json.dumps(summary(r + 1)) + "\n")
                os.replace(checkpoint + ".tmp", checkpoint)
    eng.close()
    if audit_eng:
        audit_eng.close()
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
    return summary(rounds)


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
    ap.add_argument("--channel", action="store_true", help="every capture through a random impaired channel (sample-rate offset, echoes, fading, I/Q imbalance)")
    ap.add_argument("--reconf", action="store_true", help="every capture reconfigures its multiplex once or twice mid-stream")
    ap.add_argument("--checkpoint", type=str, default=None, help="write the result so far to this file after every round")
    ap.add_argument("--audit-tfs", type=int, default=2, help="frame-sized pieces of every capture put through the decision audit of the fused OFDM kernel (0: none)")
    args = ap.parse_args()
    res = run(args.rounds, args.streams, args.tfs, args.workers, args.seed, log=sys.stderr, reference=args.reference, harsh=args.harsh, layouts=args.layouts, channel=args.channel,
              reconf=args.reconf, checkpoint=args.checkpoint, audit_tfs=args.audit_tfs)
    print(json.dumps(res))
    sys.exit(1 if res["differences"] or res["decision_audit_of_the_fused_ofdm_kernel"]["disagree_outside_band"] else 0)


if __name__ == "__main__":
    main()
