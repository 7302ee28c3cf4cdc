#!/usr/bin/env python3
"""One guard level (LEVEL = 0 / 1 / 2), one SNR (SNR, default 5 dB), REPS full-size decodes: to be run under `rocprofv3 --kernel-trace --stats` -- the
per-kernel times of the OFDM stage (ofdm_demap_kernel, exact_decide_kernel) at that level."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dabtools_amd as dab  # noqa: E402

level, snr, reps = int(os.environ.get("LEVEL", 2)), float(os.environ.get("SNR", 5.0)), int(os.environ.get("REPS", 5))
nstreams, ntf = int(os.environ.get("STREAMS", 256)), 64
cfgs = [dab.synth_preset(0, seed=2000 + i, cif_count0=(97 * i) % 5000, snr_db=snr) for i in range(nstreams)]
bufs = [torch.empty(dab.synth_bytes(c, ntf), dtype=torch.uint8, device="cuda") for c in cfgs]
dab.synth_generate_device(cfgs, ntf, [b.data_ptr() for b in bufs])
torch.cuda.synchronize()
ptrs, sizes = [b.data_ptr() for b in bufs], [b.numel() for b in bufs]
eng = dab.Engine(0)
eng.set_parity_guard(level)
eng.decode_device(ptrs, sizes)
t0 = time.perf_counter()
for _ in range(reps):
    n = eng.decode_device(ptrs, sizes)
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"level": level, "snr": snr, "ms": 1e3 * dt, "frames": n, "redecided": eng.guard_stats()[0], "stage_ms": eng.stage_ms()}))
