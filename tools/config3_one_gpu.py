#!/usr/bin/env python3
"""BASELINE configs[3] EXECUTED on a one-GPU box: 2048 synthetic Mode-I streams x 64 TF sharded 256 per slice over 8 slices of one
dabhip_multi object -- with all eight slices mapped onto GPU 0 (the boxes of this pool have one GPU; a device may be listed more than
once).  Everything configs[3] does runs for real -- eight engines with their host threads, control planes, work lists and HIP streams
side by side in one process, 51.5 GB of IQ resident, 401,408 ETI frames out -- except that the eight slices share one GPU's time
instead of owning a GPU each.  So: correctness and the host side's behaviour at eight slices, NOT a throughput figure for 8 GPUs.

  python tools/config3_one_gpu.py [--slices 8] [--streams-per-slice 256] [--tfs 64] [--steps 3] [--oracle-streams 8]
Prints one JSON line (kept as profiles/r03_config3_on_one_gpu.json)."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dabtools_amd as dab
from dabtools_amd import payload


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--streams-per-slice", type=int, default=256)
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--oracle-streams", type=int, default=8)
    a = ap.parse_args()
    import eti_check
    import oracle_lib as ol
    n = a.slices * a.streams_per_slice
    cfgs = [payload.bench_cfg(dab, g) for g in range(n)]
    nbytes = dab.synth_bytes(cfgs[0], a.tfs)
    t0 = time.perf_counter()
    bufs = [dab.DeviceBuffer(nbytes) for _ in range(n)]
    for s in range(0, n, 256):                               # the device modulator, 256 ensembles per call
        dab.synth_generate_device(cfgs[s:s + 256], a.tfs, [b.ptr for b in bufs[s:s + 256]], 0)
    t_gen = time.perf_counter() - t0
    multi = dab.Multi([0] * a.slices)
    ptrs, sizes = [b.ptr for b in bufs], [nbytes] * n
    walls, slice_walls = [], []
    total = 0
    for _ in range(a.steps):
        t0 = time.perf_counter()
        total = multi.decode_device(ptrs, sizes)
        walls.append(1e3 * (time.perf_counter() - t0))
        slice_walls.append([multi.wall_ms(i) for i in range(a.slices)])
    want_per_stream = 4 * (a.tfs - 15)
    counts = [multi.eti_count(b) for b in range(n)]
    # every stream: frame count; a digest per stream (all ensembles differ, so all digests must); a sample against the oracle and the
    # frame checker; the sharding rule
    digests = set()
    for b in range(0, n, 7):
        digests.add(hashlib.sha256(multi.eti(b).tobytes()).digest())
    sample = sorted(set(int(x) for x in np.linspace(0, n - 1, a.oracle_streams)))
    oracle_equal = 0
    for b in sample:
        want, _ = ol.or_replay(bufs[b].download())
        got = multi.eti(b)
        oracle_equal += int(got.shape == want.shape and np.array_equal(got, want))
        assert eti_check.check_sequence(got) == want_per_stream
    per = -(-n // a.slices)
    sharding_ok = all(multi.slice_of(b)[0] == b // per for b in range(0, n, 37))
    eng0 = multi.engine(0).stage_ms()
    host = []
    for i in range(a.slices):
        st = multi.engine(i).stage_ms()
        host.append({"slice": i, "control_ms": round(st["control"], 3), "host_worklist_ms": round(st["host_worklist"], 3), "wall_ms": round(st["wall"], 3),
                     "sync_ms": round(st["sync"], 3), "fft_ms": round(st["fft"], 3), "viterbi_ms": round(st["viterbi"], 3)})
    best = min(walls)
    out = {"what": "BASELINE configs[3] (batch=%d sharded %d per slice over %d slices, %d TF per stream) executed on ONE MI355X: all slices of one dabhip_multi "
                   "object mapped onto GPU 0, decoding concurrently (time-sliced)" % (n, a.streams_per_slice, a.slices, a.tfs),
           "not": "a throughput figure for 8 GPUs: the slices share one GPU",
           "streams": n, "iq_resident_GB": n * nbytes / 1e9, "modulate_s": round(t_gen, 2), "host_cores": os.cpu_count(),
           "eti_frames": total, "eti_frames_expected": n * want_per_stream, "all_streams_full_count": all(c == want_per_stream for c in counts),
           "distinct_digests_of_sampled_streams": len(digests), "streams_digested": len(range(0, n, 7)),
           "oracle_byte_equal": "%d of %d sampled streams" % (oracle_equal, len(sample)), "sharding_rule_ok": sharding_ok,
           "wall_ms_per_decode": [round(w, 2) for w in walls], "value_one_gpu_time_sliced": total / (best * 1e-3), "unit": "ETI frames/s (ONE GPU)",
           "slice_wall_ms_last": [round(w, 2) for w in slice_walls[-1]], "per_slice_last": host,
           "single_engine_reference": "one engine alone at 256 streams: about 11.4 ms per decode, control 0.52-0.55 ms, work lists 0.4-0.76 ms (profiles/r03_bench.json)"}
    print(json.dumps(out))
    ok = out["all_streams_full_count"] and oracle_equal == len(sample) and total == n * want_per_stream and sharding_ok
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
