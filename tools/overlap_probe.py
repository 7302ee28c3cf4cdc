#!/usr/bin/env python3
"""How much of the step is lost to running its stages one after the other?  K engines on ONE GPU, each decoding its own share
of the streams from its own host thread and HIP stream, so that the stages of different shares overlap on the device
(the sync search with the decoder of another share, ...).  Prints frames/s for K = 1, 2, 3, 4 at the same total batch.

  python tools/overlap_probe.py [--streams 256] [--tfs 64] [--steps 8]
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dabtools_amd as dab  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--ks", type=int, nargs="+", default=[1, 2, 3, 4])
    args = ap.parse_args()
    cfgs = [dab.synth_preset(0, seed=2000 + i, cif_count0=(611 * i) % 5000) for i in range(args.streams)]
    bufs = [dab.DeviceBuffer(dab.synth_bytes(c, args.tfs)) for c in cfgs]
    dab.synth_generate_device(cfgs, args.tfs, [b.ptr for b in bufs])
    ptrs, sizes = [b.ptr for b in bufs], [b.nbytes for b in bufs]
    out = {}
    for k in args.ks:
        engines = [dab.Engine(0) for _ in range(k)]
        share = [(i * args.streams // k, (i + 1) * args.streams // k) for i in range(k)]
        frames = [0] * k
        barrier = threading.Barrier(k + 1)

        def work(i):
            lo, hi = share[i]
            for _ in range(2):
                engines[i].decode_device(ptrs[lo:hi], sizes[lo:hi])
            barrier.wait()
            n = 0
            for _ in range(args.steps):
                n += engines[i].decode_device(ptrs[lo:hi], sizes[lo:hi])
            frames[i] = n
            barrier.wait()

        threads = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        for t in threads:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        dt = time.perf_counter() - t0
        for t in threads:
            t.join()
        out[k] = {"frames_per_s": sum(frames) / dt, "ms_per_step": 1e3 * dt / args.steps, "frames": sum(frames)}
        print(k, out[k], flush=True)
        del engines
    print(json.dumps(out))


if __name__ == "__main__":
    main()
