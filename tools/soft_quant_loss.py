#!/usr/bin/env python3
"""tools/soft_quant_loss.py -- what the 4-bit soft values cost (SURVEY.md 8(f) rank 2 names 8 bits).

CPU only, the oracle's restatement of the soft rule (oracle/or_soft.c) in its three quantisations -- 4-bit (what the product
computes), 8-bit (same scale, steps of 1/16) and unquantised -- plus the reference's hard decisions, on the same noisy captures of
the benchmark ensemble: ETI frames out, error-free frames and payload BER against what the modulator sent.  The decoder is the same
in all four; only the values differ.  One JSON document (profiles/r04_soft_quantisation.json).
"""
import argparse
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


GAIN = 0.0


def one(job):
    snr, g, ntf = job
    import ctypes
    import dabtools_amd as dab
    import oracle_lib as ol
    from dabtools_amd import payload
    if GAIN > 0:
        ol.oracle().or_soft_set_gain(ctypes.c_double(GAIN))
    cfg = payload.bench_cfg(dab, g, snr)
    iq = dab.synth_generate(cfg, ntf)
    out = {}
    for name, mode in (("hard", 0), ("soft_4bit", ol.SOFT_Q4), ("soft_8bit", ol.SOFT_Q8), ("soft_float", ol.SOFT_FLOAT)):
        frames = ol.or_replay(iq)[0] if mode == 0 else ol.or_replay_soft(iq, mode)[0]
        chk = payload.PayloadCheck()
        chk.add_stream(dab, cfg, ntf, frames)
        out[name] = (chk.frames, chk.good, chk.bit_err, chk.bits, chk.expected)
    return snr, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--snrs", type=str, default="5,6,7")
    ap.add_argument("--streams", type=int, default=8)
    ap.add_argument("--tfs", type=int, default=24)
    ap.add_argument("--workers", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--gain", type=float, default=0.0, help="experiment: mean |value| of a clean symbol (the product's rule: 7.0)")
    args = ap.parse_args()
    global GAIN
    GAIN = args.gain
    snrs = [float(x) for x in args.snrs.split(",")]
    jobs = [(s, g, args.tfs) for s in snrs for g in range(args.streams)]
    acc = {s: {} for s in snrs}
    with ProcessPoolExecutor(max_workers=args.workers) as ex:
        for snr, out in ex.map(one, jobs):
            for name, v in out.items():
                a = acc[snr].setdefault(name, [0, 0, 0, 0, 0])
                for i in range(5):
                    a[i] += v[i]
    rows = []
    for s in snrs:
        row = {"snr_db": s}
        for name, (frames, good, err, bits, expected) in acc[s].items():
            row[name] = {"eti_frames_out": frames, "frames_expected_if_locked": expected, "error_free_frames": good,
                         "payload_ber_over_fic_matched_frames": (err / bits) if bits else None, "payload_bits_compared": bits}
        if row["soft_float"]["payload_ber_over_fic_matched_frames"]:
            f = row["soft_float"]["payload_ber_over_fic_matched_frames"]
            row["ber_ratio_4bit_over_float"] = row["soft_4bit"]["payload_ber_over_fic_matched_frames"] / f
            row["ber_ratio_8bit_over_float"] = row["soft_8bit"]["payload_ber_over_fic_matched_frames"] / f
        rows.append(row)
    print(json.dumps({"what": "payload BER of the oracle's soft rule in three quantisations and of the reference's hard decisions, same captures, same decoder",
                      "workload": "benchmark ensemble (12 sub-channels, 1136 kbit/s), %d captures x %d TF per SNR, AWGN over the 2.048 MHz band" % (args.streams, args.tfs),
                      "scale": "v = %.2f / 0.9428 x Re|Im(cur conj(prev)) / (s(l) s(l-1)): mean |v| = %.2f on a clean symbol; 4-bit clamps at +-7, 8-bit at +-7.94" % (args.gain or 7.0, args.gain or 7.0),
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
