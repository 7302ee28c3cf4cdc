#!/usr/bin/env python3
"""Where a call of the K1 chain spends its time: phase stamps of stream 0's workgroup (measurement build
variants/libdabhip_synctimes.so = tools/build_variant_sync.sh synctimes -DDABHIP_SYNC_TIMES=1), on the benchmark workload
(256 streams x 64 TF, so that every CU carries a chain workgroup as in the timed step).
  DABHIP_LIB=variants/libdabhip_synctimes.so python tools/sync_times.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dabtools_amd as dab
from dabtools_amd import payload

NAMES = ["fifo bookkeeping + barrier", "null-symbol test (266 loads, block sum) + barrier", "samples -> LDS + barrier", "DFT 2048 (4 passes)",
         "x conj(PRS), decimate + barrier", "3 x IDFT 512 (3 passes)", "radix-3 combine + magnitudes", "block arg-max", "descriptor + barrier"]


def main():
    streams, tfs = 256, 64
    cfgs = [payload.bench_cfg(dab, g) for g in range(streams)]
    nbytes = dab.synth_bytes(cfgs[0], tfs)
    bufs = [dab.DeviceBuffer(nbytes) for _ in range(streams)]
    dab.synth_generate_device(cfgs, tfs, [b.ptr for b in bufs], 0)
    eng = dab.Engine(0)
    ptrs, sizes = [b.ptr for b in bufs], [nbytes] * streams
    for _ in range(3):
        eng.decode_device(ptrs, sizes)
    L = dab.lib()
    raw = np.zeros(96 * 16, np.uint64)
    assert L.dabhip_debug_sync_times(raw.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
    t = raw.reshape(96, 16)[:, :10].astype(np.float64) * 10.0      # wall_clock64: 100 MHz -> ns
    calls = range(16, 60)
    # stamps 1 .. 8 bracket the phases inside a call; the call's period comes from stamp 0 of successive calls (stamps 0 / 9 come out on another time base
    # than 1 .. 8 in this build -- not understood --, so the wave-0 bookkeeping at both ends of a call is reported as the remainder)
    phases = np.array([[t[c, i + 1] - t[c, i] for i in range(1, 8)] for c in calls]) / 1000.0     # us
    per_call = np.array([t[c + 1, 0] - t[c, 0] for c in calls]) / 1000.0
    inside = {NAMES[i]: round(float(phases[:, i - 1].mean()), 3) for i in range(1, 8)}
    out = {"what": "K1 chain, stream 0, calls 16..59 of the benchmark workload: mean us per phase", "us_per_call": round(float(per_call.mean()), 3),
           "phases": inside,
           "bookkeeping_remainder_us": round(float(per_call.mean()) - sum(inside.values()), 3),
           "remainder_is": "FIFO bookkeeping by the first wave + barrier, descriptor write + barrier, loop overhead",
           "stage_ms": {k: round(v, 3) for k, v in eng.stage_ms().items() if k in ("sync",)}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
