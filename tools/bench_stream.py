#!/usr/bin/env python3
"""Streaming-session throughput (dabhip_stream_*): B captures fed from HOST memory in segments, PCIe included.
   python tools/bench_stream.py [--streams B] [--tfs T] [--segment-calls N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import dabtools_amd as dab


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--tfs", type=int, default=256)
    ap.add_argument("--segment-calls", type=int, default=64)
    ap.add_argument("--distinct", type=int, default=4)
    a = ap.parse_args()
    caps = [dab.synth_generate(dab.synth_preset(0, seed=100 + i, cif_count0=97 * i), a.tfs) for i in range(min(a.distinct, a.streams))]
    caps = [caps[i % len(caps)] for i in range(a.streams)]
    seg = a.segment_calls * 262144
    for rep in range(2):                      # first pass warms allocations up
        st = dab.Stream(a.streams)
        frames, t0 = 0, time.perf_counter()
        for pos in range(0, caps[0].size, seg):
            frames += st.feed([c[pos:pos + seg] for c in caps])
            for b in range(min(a.streams, 2)):
                st.eti(b)                     # read back like a consumer would (two streams as a sample)
        dt = time.perf_counter() - t0
        st.close()
    print("streams=%d tfs=%d segment=%d calls (%.1f MiB/stream): %d ETI frames in %.3f s = %.0f frames/s = %.0fx real time, %.2f GB/s of IQ from host"
          % (a.streams, a.tfs, a.segment_calls, seg / 2**20, frames, dt, frames / dt, frames / dt / (1000 / 24), a.streams * caps[0].size / dt / 1e9))


if __name__ == "__main__":
    main()
