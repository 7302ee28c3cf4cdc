#!/usr/bin/env python3
"""tools/soak_batch.py -- the batch entries called again and again (a service decoding captures as they come in): does the process grow? (GPU box)

--reps decodes (default 3,000) of --streams device-resident captures of --tf transmission frames each, alternately through dabhip_engine_decode and through
dabhip_multi_decode over two slices of the one GPU, the frames read back three ways in turn (eti_read per stream, one eti_fetch + wait, drain callback);
heap in use (mallinfo2) and the device's free memory sampled every 100 decodes: the second half's peak no higher than the first half's.  The ETI bytes of every decode must equal the
first decode's (same input).  One JSON object; exit code 1 when a check fails."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from soak import device_free_bytes, heap_in_use_kb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3000)
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--tf", type=int, default=20)
    ap.add_argument("--only", default="", help="engine | multi: one entry only; + :read | :fetch | :drain: one way of reading only (diagnosis)")
    a = ap.parse_args()
    import dabtools_amd as dab
    hip = C.CDLL("libamdhip64.so")
    cfgs = [dab.synth_preset(b % 2, seed=69000 + b, snr_db=[1000.0, 15.0][b % 2]) for b in range(a.streams)]
    nbytes = dab.synth_bytes(cfgs[0], a.tf)
    bufs = [dab.DeviceBuffer(nbytes) for _ in cfgs]
    dab.synth_generate_device(cfgs, a.tf, [b.ptr for b in bufs])
    ptrs, sizes = [b.ptr for b in bufs], [nbytes] * a.streams
    eng = dab.Engine(0)
    multi = dab.Multi([0, 0])
    pinned = dab.HostBuffer(a.streams * 4 * a.tf * dab.ETI_BYTES)
    want, wrong, samples = None, 0, []
    t0 = time.time()
    entry, _, way = a.only.partition(":")
    for k in range(a.reps):
        use_engine = entry == "engine" or (entry == "" and k % 2 == 0)
        if use_engine:
            n = eng.decode_device(ptrs, sizes)
            if way == "read" or (way == "" and k % 4 == 0):
                got = np.concatenate([eng.eti(b) for b in range(a.streams)])
            elif way == "drain":
                acc = []
                sink = C.CFUNCTYPE(None, dab.u8p, C.c_int, C.c_void_p)(lambda p, b, _u: acc.append(C.string_at(p, dab.ETI_BYTES)))
                assert dab.lib().dabhip_engine_eti_drain(eng._h, C.cast(sink, C.c_void_p), None) == n
                got = np.frombuffer(b"".join(acc), np.uint8)
            else:
                m = eng.eti_fetch(pinned.ptr, a.streams * 4 * a.tf)
                eng.eti_fetch_wait()
                got = pinned.array[: m * dab.ETI_BYTES].reshape(m, dab.ETI_BYTES).copy()
        else:
            n = multi.decode_device(ptrs, sizes)
            if way == "read" or (way == "" and k % 4 == 1):
                got = np.concatenate([multi.eti(b) for b in range(a.streams)])
            elif way == "drain0":
                sink = C.CFUNCTYPE(None, dab.u8p, C.c_int, C.c_void_p)(lambda p, b, _u: None)
                assert dab.lib().dabhip_multi_eti_drain(multi._h, C.cast(sink, C.c_void_p), None) == n
                got = np.zeros(1, np.uint8)
            else:
                got = np.frombuffer(b"".join(f for _, f in sorted(multi.drain(), key=lambda x: x[0])), np.uint8).reshape(-1, dab.ETI_BYTES)
        h = hashlib.sha256(got.tobytes()).hexdigest()
        if want is None:
            want, frames = h, int(n)
        wrong += h != want or int(n) != frames
        if k % 100 == 99:
            samples.append((heap_in_use_kb(), device_free_bytes(hip)))
    seconds = time.time() - t0
    eng.close()
    multi.close()
    half = samples[len(samples) // 2:]
    # the heap breathes by ~12 MB with a period of ~1,200 decodes (garbage of the interpreter and of the runtime, let go of in bulk): peaks are compared
    peak_growth = max(s[0] for s in half) - max(s[0] for s in samples[: len(samples) // 2])
    out = {"what": "%d decodes of %d streams x %d TF, dabhip_engine_decode and dabhip_multi_decode (two slices of one GPU) in turn, frames read by eti_read / eti_fetch / drain"
                   % (a.reps, a.streams, a.tf), "only": a.only, "seconds": round(seconds, 1), "eti_frames_per_decode": frames, "decodes_with_other_bytes": int(wrong),
           "heap_in_use_kb_by_sample": [s[0] for s in samples], "heap_peak_second_half_minus_first_half_kb": peak_growth,
           "device_free_change_over_the_second_half": half[0][1] - half[-1][1]}
    out["ok"] = bool(wrong == 0 and frames == a.streams * 4 * (a.tf - 15) and peak_growth < 512 and out["device_free_change_over_the_second_half"] < (16 << 20))
    print(json.dumps(out))
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
