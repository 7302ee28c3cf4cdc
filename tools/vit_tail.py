#!/usr/bin/env python3
"""Where the MSC Viterbi launch's time goes between its waves: per-wave start / forward-pass-done / end stamps from the
measurement build variants/libdabhip_times.so (tools/build_variant_decode.sh times -DDABHIP_VIT_TIMES=1), on the benchmark workload.

  DABHIP_LIB=variants/libdabhip_times.so python tools/vit_tail.py [--streams 256] [--tfs 64]
Prints one JSON line: resident waves over time, the share of the launch with fewer than 4 / 3 / 2 waves per SIMD resident, per length class
the time per trellis step early and late in the launch."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dabtools_amd as dab
from dabtools_amd import payload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--tfs", type=int, default=64)
    a = ap.parse_args()
    L = dab.lib()
    if not hasattr(L, "dabhip_debug_vit_times"):
        sys.exit("vit_tail.py needs the measurement build: DABHIP_LIB=variants/libdabhip_times.so")
    cfgs = [payload.bench_cfg(dab, g) for g in range(a.streams)]
    nbytes = dab.synth_bytes(cfgs[0], a.tfs)
    bufs = [dab.DeviceBuffer(nbytes) for _ in range(a.streams)]
    dab.synth_generate_device(cfgs, a.tfs, [b.ptr for b in bufs], 0)
    eng = dab.Engine(0)
    for _ in range(3):
        frames = eng.decode_device([b.ptr for b in bufs], [nbytes] * a.streams)
    vit_ms = eng.stage_ms()["viterbi"]
    ngroups = 12 * (-(-frames // 64))
    raw = np.zeros((ngroups, 4), dtype=np.uint64)
    assert L.dabhip_debug_vit_times(raw.ctypes.data_as(C.POINTER(C.c_ulonglong)), ngroups) == 0
    tick_us = 0.01                                           # wall_clock64: 100 MHz
    t0 = int(raw[:, 0].min())
    start = (raw[:, 0].astype(np.int64) - t0) * tick_us
    fwd = (raw[:, 1].astype(np.int64) - t0) * tick_us
    end = (raw[:, 2].astype(np.int64) - t0) * tick_us
    nsteps = (raw[:, 3] >> np.uint64(8)).astype(np.int64)
    xcc = (raw[:, 3] & np.uint64(15)).astype(np.int64)
    total = float(end.max())
    grid = np.arange(0.0, total, 10.0)
    resident = np.array([np.count_nonzero((start <= t) & (end > t)) for t in grid])
    in_fwd = np.array([np.count_nonzero((start <= t) & (fwd > t)) for t in grid])
    slots = 4096
    out = {"workload": "%d streams x %d TF, %d wave-groups (waves) of the MSC launch" % (a.streams, a.tfs, ngroups), "viterbi_stage_ms": vit_ms,
           "launch_us_first_start_to_last_end": total, "wave_slots": slots,
           "mean_resident_waves": float(resident.mean()), "mean_waves_in_forward_pass": float(in_fwd.mean()),
           "share_of_launch_time_with_resident_waves_below": {"4_per_simd(4096)": float(np.mean(resident < 4096)), "3_per_simd(3072)": float(np.mean(resident < 3072)),
                                                              "2_per_simd(2048)": float(np.mean(resident < 2048)), "1_per_simd(1024)": float(np.mean(resident < 1024))},
           "idle_wave_slot_time_pct": 100.0 * (1.0 - resident.mean() / slots),
           "resident_waves_every_250us": [int(x) for x in resident[::25]],
           "waves_per_xcc": [int(np.count_nonzero(xcc == x)) for x in range(8)], "classes": []}
    for n in sorted(set(nsteps.tolist()), reverse=True):
        m = nsteps == n
        dur = fwd[m] - start[m]
        order = np.argsort(start[m])
        k = max(1, len(order) // 4)
        out["classes"].append({"nsteps": int(n), "waves": int(m.sum()), "first_start_us": float(start[m].min()), "last_end_us": float(end[m].max()),
                               "forward_ns_per_step_mean": float(1e3 * dur.mean() / n), "forward_ns_per_step_first_quarter": float(1e3 * dur[order[:k]].mean() / n),
                               "forward_ns_per_step_last_quarter": float(1e3 * dur[order[-k:]].mean() / n),
                               "chain_back_us_mean": float((end[m] - fwd[m]).mean())})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
