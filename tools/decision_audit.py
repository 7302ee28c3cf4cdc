#!/usr/bin/env python3
"""Measures the tolerance of the fp32 OFDM stage against fp64 (north_star: "pre-decision soft metrics allowed a stated float
tolerance"): hard-decision disagreement rate vs fp64 transforms of the same int8 samples, with the parity guard off (raw rate, and
whether any disagreement falls outside the guard band) and on (must be zero), the worst error of any bin and of any differential
product relative to the guard's units, and the margin the guard's constants keep over them.

  --fused (round 5, the default since): the kernel the DEFAULT decode runs -- ofdm_demap_kernel's guarded build, audited through its
          fourth build (the same source lines plus stores of its bins and products: k_fused.hip, DABHIP_FUSED_AUDIT) -- and, for
          every case, whether the shipping build left exactly the same bits and listed the same number of decisions on the same frames.
  --two-kernel: K2 + K2b (rounds 2-4: profiles/r02_decision_audit.json).
  --channels N: besides the noise / amplitude grid, N transmission frames through each of the impaired channels of round 5 (sample-rate
          offset, echoes inside and beyond the prefix, fading, I/Q imbalance; host generator, unaligned frames: any bytes serve a numerical audit).
Round 6: every case is audited at BOTH guard levels (dabhip.h: 1 = measured band, 2 = proven band -- the rigorous forward-error bound of
tools/fft_error_bound.py); the worst measured errors must lie below the proven bound, and no disagreement may fall outside either band.
Needs a GPU.  Prints one JSON object; profiles/r06_decision_audit.json holds the run DESIGN.md section 3 quotes.  The JSON carries the sha-256 of the
kernel sources it was measured on (tests/test_bench_launch.py holds it against the tree)."""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

AUDITED_SOURCES = ("k_fused.hip", "fft_core.hpp", "device_types.hpp", "k_parity.hip")

CHANNELS = [
    ("sro +100 ppm", dict(sro_ppm=100.0), 12.0),
    ("sro -60 ppm", dict(sro_ppm=-60.0), 1000.0),
    ("echo 50 / 0.7", dict(echo_delay=[50, 0], echo_gain=[0.7, 0], echo_phase=[0.3, 0]), 9.0),
    ("echo 400 / 0.9 + 600 / 0.4, Doppler", dict(echo_delay=[400, 600], echo_gain=[0.9, 0.4], echo_phase=[0.61, 0.2], echo_doppler_hz=[4.0, -9.0]), 14.0),
    ("fading 0.9 at 3 Hz", dict(fade_depth=0.9, fade_hz=3.0), 8.0),
    ("iq imbalance 2 dB / 10 deg", dict(iq_gain_db=2.0, iq_phase_deg=10.0), 6.0),
    ("all of them", dict(sro_ppm=45.0, echo_delay=[120, 700], echo_gain=[0.6, 0.3], echo_phase=[0.2, 0.8], fade_depth=0.6, fade_hz=1.1, iq_gain_db=-1.0, iq_phase_deg=-5.0), 10.0),
]


def source_sha():
    src = b"".join(open(os.path.join(ROOT, "dabtools_amd", "csrc", n), "rb").read() for n in AUDITED_SOURCES)
    return hashlib.sha256(src).hexdigest()


def _channel_capture(args):
    idx, seed, ntf = args
    import dabtools_amd as dab
    name, fields, snr = CHANNELS[idx]
    cfg = dab.synth_preset(0, seed=seed, cif_count0=(37 * seed) % 5000, snr_db=snr, amplitude=(1.0, 0.5)[seed & 1])
    for k, v in fields.items():
        if isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                getattr(cfg.channel, k)[i] = x
        else:
            setattr(cfg.channel, k, v)
    iq = dab.synth_generate(cfg, ntf)
    return iq[: (iq.size // 393216) * 393216]


def audit(eng, fused, **kw):
    rec = {}
    for level in (1, 2):
        eng.set_parity_guard(level)                       # the rule the audit counts with, and the band the guarded pass re-decides
        off = eng.decision_audit(guard=False, fused=fused, **kw)
        on = eng.decision_audit(guard=True, fused=fused, **kw)
        part = {"guard_off": {"disagree": off["disagree"], "rate": off["disagree"] / off["decisions"],
                              "disagree_outside_guard_band": off["disagree_outside_guard"], "flagged_by_rule": off["flagged_by_rule"],
                              "flag_rate": off["flagged_by_rule"] / off["decisions"], "listed_by_the_kernel": off["listed"]},
                "guard_on": {"disagree": on["disagree"], "listed": on["listed"]}}
        if fused:
            part["shipping_kernel_same_bits"] = bool(off["shipping_kernel_same_bits"] == 1.0)
            part["shipping_kernel_same_list_count"] = bool(off["shipping_kernel_same_list_count"] == 1.0 and on["shipping_kernel_same_list_count"] == 1.0)
        if level == 1:                                    # (the errors themselves do not depend on the level)
            rec.update(part)
            rec.update({"decisions": off["decisions"], "max_bin_err_over_sqrt_energy": off["max_bin_err"], "max_product_err_over_unit": off["max_dec_err"],
                        "max_residual_product_err": off["max_prod_err"]})
        else:
            rec["proven_level"] = part
    eng.set_parity_guard(True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tfs", type=int, default=4400, help="transmission frames per case (4400 x 230,400 = 1.01e9 decisions)")
    ap.add_argument("--snrs", type=str, default="5,6,8,10,1000")
    ap.add_argument("--amplitudes", type=str, default="1.0,0.35")
    ap.add_argument("--two-kernel", action="store_true", help="audit K2 + K2b instead of the fused kernel")
    ap.add_argument("--channels", type=int, default=0, help="transmission frames per impaired-channel case (0: none)")
    ap.add_argument("--workers", type=int, default=8)
    args = ap.parse_args()
    fused = not args.two_kernel
    import torch
    import dabtools_amd as dab
    eng = dab.Engine(0)
    per_stream = 40
    cases = []
    for amp in [float(x) for x in args.amplitudes.split(",")]:
        for snr in [float(x) for x in args.snrs.split(",")]:
            nstreams = (args.tfs + per_stream - 1) // per_stream
            cfgs = [dab.synth_preset(0, seed=9000 + 131 * i + int(snr), cif_count0=(61 * i) % 5000, snr_db=snr, amplitude=amp) for i in range(nstreams)]
            nbytes = dab.synth_bytes(cfgs[0], per_stream)
            big = torch.empty(nstreams * nbytes, dtype=torch.uint8, device="cuda")       # aligned captures back to back = contiguous frames
            dab.synth_generate_device(cfgs, per_stream, [big.data_ptr() + i * nbytes for i in range(nstreams)])
            torch.cuda.synchronize()
            n = nstreams * per_stream
            t0 = time.time()
            rec = dict({"channel": "ideal", "snr_db": snr, "amplitude": amp, "tfs": n}, **audit(eng, fused, device_ptr=big.data_ptr(), nframes=n))
            rec["seconds"] = round(time.time() - t0, 2)
            cases.append(rec)
            print(json.dumps(rec), file=sys.stderr)
            del big
    if args.channels > 0:
        per = 50
        with mp.get_context("spawn").Pool(args.workers) as pool:
            for idx, (name, fields, snr) in enumerate(CHANNELS):
                caps = pool.map(_channel_capture, [(idx, 7000 + 100 * idx + i, per) for i in range((args.channels + per - 1) // per)])
                frames = np.concatenate(caps)
                t0 = time.time()
                rec = dict({"channel": name, "snr_db": snr, "tfs": frames.size // 393216}, **audit(eng, fused, frames=frames))
                rec["seconds"] = round(time.time() - t0, 2)
                cases.append(rec)
                print(json.dumps(rec), file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fft_error_bound
    bound = fft_error_bound.constants()
    measured_c, proven_c = dab.guard_constants(1), dab.guard_constants(2)
    worst_bin = max(c["max_bin_err_over_sqrt_energy"] for c in cases)
    worst_dec = max(c["max_product_err_over_unit"] for c in cases)
    worst_res = max(c["max_residual_product_err"] for c in cases)
    out = {"what": ("the default decode's one-kernel OFDM stage (ofdm_demap_kernel, guarded build, through its audit build)" if fused else "fp32 OFDM stage (K2 + K2b)")
                   + " vs fp64 transforms of the same samples, hard decisions",
           "kernel": "ofdm_demap_kernel<false> (k_fused.hip)" if fused else "ofdm_fft_kernel + demap_kernel (k_fft.hip)",
           "source_sha256": source_sha(), "sources": list(AUDITED_SOURCES),
           "default_guard_level": dab.guard_default_level(),
           "guard_constants": {"measured": {"kGuardC": measured_c[0], "kGuardProd": measured_c[1]}, "proven": {"kGuardCProven": proven_c[0], "kGuardProdProven": proven_c[1]}},
           # the rigorous forward-error bound of this transform and product (tools/fft_error_bound.py): the proven level's constants are >= it,
           # the measured level's are below it and stand on the worst errors measured here
           "proven_bound": {"bin_err_over_l2": bound["bin_bound"], "product_rounding_over_l2l2": bound["prod_bound"], "from": "tools/fft_error_bound.py"},
           "worst_max_bin_err_over_sqrt_energy": worst_bin, "worst_max_product_err_over_unit": worst_dec, "worst_max_residual_product_err": worst_res,
           "worst_measured_below_proven_bound": bool(worst_bin <= bound["bin_bound"] and worst_res <= bound["prod_bound"]),
           "proven_constants_cover_the_bound": bool(proven_c[0] >= bound["bin_bound"] and proven_c[1] >= bound["prod_bound"]),
           "margin_kGuardC_over_worst": measured_c[0] / max(worst_bin, worst_dec, 1e-30), "margin_kGuardProd_over_worst": measured_c[1] / max(worst_res, 1e-30),
           "margin_proven_bound_over_worst_bin": bound["bin_bound"] / max(worst_bin, 1e-30), "margin_proven_bound_over_worst_product": bound["prod_bound"] / max(worst_res, 1e-30),
           "total_decisions": sum(c["decisions"] for c in cases),
           "total_disagree_guard_off": sum(c["guard_off"]["disagree"] for c in cases),
           "total_disagree_outside_band": sum(c["guard_off"]["disagree_outside_guard_band"] for c in cases),
           "total_disagree_guard_on": sum(c["guard_on"]["disagree"] for c in cases),
           "proven_level": {"total_disagree_outside_band": sum(c["proven_level"]["guard_off"]["disagree_outside_guard_band"] for c in cases),
                            "total_disagree_guard_on": sum(c["proven_level"]["guard_on"]["disagree"] for c in cases),
                            "total_flagged_by_rule": sum(c["proven_level"]["guard_off"]["flagged_by_rule"] for c in cases)},
           "total_flagged_by_rule": sum(c["guard_off"]["flagged_by_rule"] for c in cases),
           "cases": cases}
    if fused:
        out["shipping_kernel_same_bits_in_every_case"] = all(c["shipping_kernel_same_bits"] and c["proven_level"]["shipping_kernel_same_bits"] for c in cases)
        out["shipping_kernel_same_list_count_in_every_case"] = all(c["shipping_kernel_same_list_count"] and c["proven_level"]["shipping_kernel_same_list_count"] for c in cases)
    print(json.dumps(out))
    ok = (out["total_disagree_outside_band"] == 0 and out["total_disagree_guard_on"] == 0 and out["proven_level"]["total_disagree_outside_band"] == 0 and
          out["proven_level"]["total_disagree_guard_on"] == 0 and out["worst_measured_below_proven_bound"] and out["proven_constants_cover_the_bound"] and
          out.get("shipping_kernel_same_bits_in_every_case", True))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
