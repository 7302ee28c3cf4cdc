#!/usr/bin/env python3
"""tools/session_steady.py -- the steady state of a device-fed session by itself (bench.py: steady_state_session), for profiling:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/session_steady.py
then tools/timeline.py OUT prints the kernel timeline of the last but one segment."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--segments", type=int, default=6)
    ap.add_argument("--prefetch", action="store_true", help="hand segment k + 1 over (dabhip_stream_prefetch) before segment k is decoded")
    args = ap.parse_args()
    import torch
    import dabtools_amd as dab
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    tensors, _ = bench.make_streams(torch, dev, args.streams, args.tfs, args.streams, 0)
    ptrs, sizes = [t.data_ptr() for t in tensors], [t.numel() for t in tensors]
    st = dab.Stream(args.streams, device=0)
    first = st.feed_ptrs(ptrs, sizes, on_device=True)
    n = st.feed_ptrs(ptrs, sizes, on_device=True)
    torch.cuda.synchronize(dev)
    lat, stage = [], {}
    if args.prefetch:
        st.prefetch_ptrs(ptrs, sizes, on_device=True)
    for k in range(args.segments):
        t0 = time.perf_counter()
        if args.prefetch:
            # (a second hand-over of the same addresses: the session checks addresses and sizes only)
            st.prefetch_ptrs(ptrs, sizes, on_device=True)
        n = st.feed_ptrs(ptrs, sizes, on_device=True)
        lat.append(1e3 * (time.perf_counter() - t0))
        for kk, v in st.stage_ms().items():
            stage[kk] = stage.get(kk, 0.0) + v / args.segments
    print(json.dumps({"first": first, "frames_per_segment": n, "ms_per_segment": lat, "stage_ms": {k: round(v, 3) for k, v in stage.items()}}))
    st.close()


if __name__ == "__main__":
    main()
