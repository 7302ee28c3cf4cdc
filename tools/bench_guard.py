#!/usr/bin/env python3
"""Cost of the parity guard: the benchmark batch decoded with the guard on and off (stage times, frames/s), clean and at 5 dB."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dabtools_amd as dab  # noqa: E402

nstreams, ntf = int(os.environ.get("STREAMS", 256)), 64
out = []
for snr in (1000.0, 5.0):
    cfgs = [dab.synth_preset(0, seed=2000 + i, cif_count0=(97 * i) % 5000, snr_db=snr) for i in range(nstreams)]
    bufs = [torch.empty(dab.synth_bytes(c, ntf), dtype=torch.uint8, device="cuda") for c in cfgs]
    dab.synth_generate_device(cfgs, ntf, [b.data_ptr() for b in bufs])
    torch.cuda.synchronize()
    ptrs, sizes = [b.data_ptr() for b in bufs], [b.numel() for b in bufs]
    eng = dab.Engine(0)
    for fused in (True, False):
        for guard in (False, True, False, True):
            eng.set_fused(fused)
            eng.set_parity_guard(guard)
            eng.decode_device(ptrs, sizes)
            t0 = time.perf_counter()
            acc = {}
            for _ in range(5):
                n = eng.decode_device(ptrs, sizes)
                for k, v in eng.stage_ms().items():
                    acc[k] = acc.get(k, 0) + v / 5
            dt = (time.perf_counter() - t0) / 5
            rec = {"snr": snr, "fused": fused, "guard": guard, "ms": 1e3 * dt, "frames": n, "flagged": eng.guard_stats()[0],
                   **{k: round(acc[k], 3) for k in ("sync", "fft", "demap", "fic", "viterbi")}}
            out.append(rec)
            print(json.dumps(rec))
    eng.close()
    del bufs
