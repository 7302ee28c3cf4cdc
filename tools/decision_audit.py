#!/usr/bin/env python3
"""Measures the tolerance of the fp32 OFDM stage against fp64 (north_star: "pre-decision soft metrics allowed a stated float
tolerance"): hard-decision disagreement rate of K2 + K2b vs fp64 transforms of the same int8 samples, over >= 10^9 decisions per
SNR, with the parity guard off (raw rate, and whether any disagreement falls outside the guard band) and on (must be zero).
Needs a GPU.  Prints one JSON object; profiles/r02_decision_audit.json holds the run DESIGN.md section 3 quotes."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tfs", type=int, default=4400, help="transmission frames per case (4400 x 230,400 = 1.01e9 decisions)")
    ap.add_argument("--snrs", type=str, default="5,6,8,10,1000")
    ap.add_argument("--amplitudes", type=str, default="1.0,0.35")
    args = ap.parse_args()
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
            off = eng.decision_audit(device_ptr=big.data_ptr(), nframes=n, guard=False)
            on = eng.decision_audit(device_ptr=big.data_ptr(), nframes=n, guard=True)
            rec = {"snr_db": snr, "amplitude": amp, "tfs": n, "decisions": off["decisions"],
                   "guard_off": {"disagree": off["disagree"], "rate": off["disagree"] / off["decisions"],
                                 "disagree_outside_guard_band": off["disagree_outside_guard"], "flagged_by_rule": off["flagged_by_rule"],
                                 "flag_rate": off["flagged_by_rule"] / off["decisions"]},
                   "guard_on": {"disagree": on["disagree"], "listed": on["listed"]},
                   "max_bin_err_over_sqrt_energy": off["max_bin_err"], "max_product_err_over_unit": off["max_dec_err"],
                   "max_residual_product_err": off["max_prod_err"], "seconds": time.time() - t0}
            cases.append(rec)
            print(json.dumps(rec), file=sys.stderr)
            del big
    worst_bin = max(c["max_bin_err_over_sqrt_energy"] for c in cases)
    worst_dec = max(c["max_product_err_over_unit"] for c in cases)
    out = {"what": "fp32 OFDM stage (K2 + K2b) vs fp64 transforms of the same samples, hard decisions",
           "guard_constants": {"kGuardC": 5.0e-6, "kGuardProd": 5.0e-7},
           "worst_max_bin_err_over_sqrt_energy": worst_bin, "worst_max_product_err_over_unit": worst_dec,
           "margin_kGuardC_over_worst": 5.0e-6 / max(worst_bin, worst_dec, 1e-30),
           "total_decisions": sum(c["decisions"] for c in cases),
           "total_disagree_guard_off": sum(c["guard_off"]["disagree"] for c in cases),
           "total_disagree_outside_band": sum(c["guard_off"]["disagree_outside_guard_band"] for c in cases),
           "total_disagree_guard_on": sum(c["guard_on"]["disagree"] for c in cases),
           "cases": cases}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
