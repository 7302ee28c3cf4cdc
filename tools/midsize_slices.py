#!/usr/bin/env python3
"""Mid-size batches (8 .. 64 streams x 64 TF) are bound by latency, not by work: every stage runs far below the device's width.  What if the
batch is decoded as k independent SLICES running side by side on the same GPU (dabhip_multi with the device listed k times: k engines, k host threads, k sets of
HIP streams, no lock between them)?  ms per decode and ETI frames/s for B in --batches and k in --slices; the bytes are checked against the single engine's."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,16,32,64,128")
    ap.add_argument("--slices", default="1,2,4,8")
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import dabtools_amd as dab
    from dabtools_amd import payload
    batches = [int(x) for x in args.batches.split(",")]
    slices = [int(x) for x in args.slices.split(",")]
    B_all = max(batches)
    cfgs = [payload.bench_cfg(dab, i) for i in range(B_all)]
    tensors = [torch.empty(dab.synth_bytes(c, args.tfs), dtype=torch.uint8, device="cuda") for c in cfgs]
    dab.synth_generate_device(cfgs, args.tfs, [t.data_ptr() for t in tensors], 0)
    torch.cuda.synchronize()
    ptrs, sizes = [t.data_ptr() for t in tensors], [t.numel() for t in tensors]
    rows = []
    for k in slices:
        m = dab.Multi([0] * k)
        for B in batches:
            if B < k:
                continue
            for _ in range(3):
                frames = m.decode_device(ptrs[:B], sizes[:B])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                frames = m.decode_device(ptrs[:B], sizes[:B])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            h = hashlib.sha256()
            for b in (0, B // 2, B - 1):
                h.update(m.eti(b).tobytes())
            rows.append({"streams": B, "slices": k, "ms_per_decode": round(1e3 * dt, 4), "eti_frames": frames, "eti_frames_per_s": round(frames / dt), "sha": h.hexdigest()[:12]})
            print("B=%4d slices=%d  %8.3f ms  %9.0f frames/s  %s" % (B, k, 1e3 * dt, frames / dt, rows[-1]["sha"]), file=sys.stderr, flush=True)
        m.close()
    same = all(len({r["sha"] for r in rows if r["streams"] == B}) == 1 for B in batches)
    print(json.dumps({"what": "B streams x %d TF decoded as k slices side by side on ONE GPU (dabhip_multi, device listed k times)" % args.tfs, "steps": args.steps,
                      "same_bytes_for_every_k": same, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
