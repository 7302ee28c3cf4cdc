#!/usr/bin/env python3
"""Micro-benchmark of the OFDM FFT stage alone (K2): achieved algorithmic HBM GB/s."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import dabtools_amd as dab

ap = argparse.ArgumentParser()
ap.add_argument("--tfs", type=int, default=1024)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--offset", type=int, default=0, help="byte offset of the first frame (alignment experiment)")
args = ap.parse_args()
iq = torch.randint(1, 255, (args.tfs * dab.TF_BYTES + 64,), dtype=torch.uint8, device="cuda")
eng = dab.Engine(0)
for _ in range(2):
    _, ms = eng.stage_ofdm_fft(None, reps=args.reps, device_ptr=iq.data_ptr() + args.offset, nframes=args.tfs, want_output=False)
    gbs = 1556480 * args.tfs / (ms * 1e-3) / 1e9
    print("ofdm_fft: %d TF per launch, %.3f ms per launch, %.0f GB/s algorithmic = %.1f %% of 8 TB/s" % (args.tfs, ms, gbs, gbs / 80))
