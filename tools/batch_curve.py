#!/usr/bin/env python3
"""tools/batch_curve.py -- the small-batch regime (BASELINE configs[1] = ONE ensemble on one MI355X).

For B in {1, 2, 4, ..., 256} streams x --tfs TF of the benchmark ensemble (resident in HBM, device modulator): ETI frames/s,
ms per decode and the per-stage times of the batch engine; then a ONE-stream streaming session fed one transmission frame
(393,216 B = 96 ms of signal) at a time, which is how a live receiver (dab2eti.c:60-115: one sdr_demod call per 262,144-byte
buffer) would drive the library: latency per segment.  One JSON document on stdout (profiles/r04_batch_curve.json).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REALTIME_FPS = 1000.0 / 24.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--max-batch", type=int, default=256)
    ap.add_argument("--batches", type=str, default="")
    ap.add_argument("--session-tfs", type=int, default=48)
    ap.add_argument("--soft", action="store_true")
    args = ap.parse_args()
    import torch
    import dabtools_amd as dab
    from dabtools_amd import payload

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B_all = args.max_batch
    cfgs = [payload.bench_cfg(dab, i) for i in range(B_all)]
    tensors = [torch.empty(dab.synth_bytes(c, args.tfs), dtype=torch.uint8, device=dev) for c in cfgs]
    dab.synth_generate_device(cfgs, args.tfs, [t.data_ptr() for t in tensors], 0)
    torch.cuda.synchronize()
    ptrs, sizes = [t.data_ptr() for t in tensors], [t.numel() for t in tensors]
    eng = dab.Engine(0)
    if args.soft:
        eng.set_soft(True)
    batches = [int(x) for x in args.batches.split(",")] if args.batches else [b for b in (1, 2, 4, 8, 16, 32, 64, 128, 256) if b <= B_all]
    rows = []
    for B in batches:
        call = eng.marshal(ptrs[:B], sizes[:B])
        for _ in range(3):
            frames = eng.decode_marshalled(call)
        torch.cuda.synchronize()
        stage = {}
        t0 = time.perf_counter()
        for _ in range(args.steps):
            frames = eng.decode_marshalled(call)
            for k, v in eng.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        rows.append({"streams": B, "tf_per_stream": args.tfs, "eti_frames_per_decode": frames, "ms_per_decode": 1e3 * dt,
                     "eti_frames_per_s": frames / dt, "x_realtime_aggregate": frames / dt / REALTIME_FPS,
                     "x_realtime_per_stream": frames / dt / REALTIME_FPS / B,
                     "stage_ms": {k: round(v / args.steps, 4) for k, v in stage.items() if not k.startswith("h2d")}})
        print("B=%4d  %8.3f ms  %10.0f frames/s  %8.0f x real time" % (B, 1e3 * dt, frames / dt, frames / dt / REALTIME_FPS), file=sys.stderr)
    eng.close()

    # one live ensemble: a session fed TF by TF from page-locked host memory (upload included), frames read back to the host
    import numpy as np
    nb = args.session_tfs * dab.TF_BYTES
    hb = dab.HostBuffer(nb)
    assert dab.lib().dabhip_device_copy(hb.ptr, tensors[0].data_ptr(), nb, 0) == 0
    sess = []
    for seg_tfs in (1, 4):
        st = dab.Stream(1, device=0, soft=args.soft)
        lat = []
        total = 0
        for k in range(0, args.session_tfs, seg_tfs):
            t0 = time.perf_counter()
            n = st.feed_ptrs([hb.ptr + k * dab.TF_BYTES], [seg_tfs * dab.TF_BYTES])
            if n:
                st.eti(0)                                     # the frames on the host, as the CLI writes them
            lat.append(1e3 * (time.perf_counter() - t0))
            total += n
        st.close()
        warm = lat[max(3, 16 // seg_tfs + 1):]
        sess.append({"segment_tfs": seg_tfs, "segments": len(lat), "eti_frames": total,
                     "ms_per_segment_median": float(np.median(warm)), "ms_per_segment_p95": float(np.percentile(warm, 95)), "ms_per_segment_max": float(max(warm)),
                     "signal_ms_per_segment": 96.0 * seg_tfs, "x_realtime": 96.0 * seg_tfs / float(np.median(warm)),
                     "note": "host -> device upload, decode, ETI frames back on the host, per segment; after lock-in"})
    hb.free()
    print(json.dumps({"what": "small-batch curve of the batch engine (IQ resident) + one-stream sessions fed segment by segment (host-fed)",
                      "soft": args.soft, "steps": args.steps, "curve": rows, "single_stream_session": sess}, indent=1))


if __name__ == "__main__":
    main()
