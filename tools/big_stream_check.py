#!/usr/bin/env python3
"""tools/big_stream_check.py -- maximum sizes: ONE stream longer than 4 GiB (GPU box).

A capture of --tfs transmission frames (default 11,200 = 4.40 GB of cu8, 17.9 minutes of signal; byte offsets beyond 2^32, 44,740 ETI frames,
the CIF counter wrapping at 5000 eight times) is modulated on the device and decoded
  (a) in one dabhip_engine_decode call (device pointer),
  (b) by a session in segments of odd sizes (device pointers),
and, with --oracle-tfs N > 0, its first N frames also by the CPU oracle (oracle/or_replay, a single core: ~200 ETI frames/s).
All three must give the same bytes; the frame count must be dab2eti's 4 (T - 15).  Prints one JSON line; exit code 1 on any difference.
Checker use of oracle/ only (like tests/)."""
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


def run(tfs=11200, oracle_tfs=400):
    class args:
        pass
    args.tfs, args.oracle_tfs = tfs, oracle_tfs
    import dabtools_amd as dab
    cfg = dab.synth_preset(0, seed=31337, cif_count0=4321, snr_db=14.0)
    nbytes = dab.synth_bytes(cfg, args.tfs)
    buf = dab.DeviceBuffer(nbytes)
    t0 = time.time()
    dab.synth_generate_device([cfg], args.tfs, [buf.ptr])
    t_gen = time.time() - t0
    out = {"what": "one stream of %d TF = %d bytes (%.2f GiB) of cu8, device-resident" % (args.tfs, nbytes, nbytes / 2.0 ** 30), "modulate_s": round(t_gen, 2)}
    ok = True
    eng = dab.Engine(0)
    t0 = time.time()
    n = eng.decode_device([buf.ptr], [nbytes])
    out["one_shot"] = {"eti_frames": int(n), "expected": 4 * (args.tfs - 15), "seconds": round(time.time() - t0, 3), "stream_status": eng.stream_status(0)}
    ok = ok and n == 4 * (args.tfs - 15) and eng.stream_status(0) == 0
    whole = eng.eti(0)
    eng.close()
    out["one_shot"]["sha256"] = hashlib.sha256(whole.tobytes()).hexdigest()
    # FCT (byte 4) must count 0..249 over and over, the FIG 0/0 CIF counter inside the FIBs wraps at 5000: both are inside the bytes compared below
    fct = whole[:, 4].astype(np.int32)
    out["one_shot"]["fct_steps_ok"] = bool(np.all((np.diff(fct) % 250) == 1))
    ok = ok and out["one_shot"]["fct_steps_ok"]
    # (b) session, odd segment sizes, some beyond 2^31 bytes into the stream, one segment itself > 2^31 bytes when the capture allows
    st = dab.Stream(1, device=0)
    cuts = [0]
    rng = np.random.default_rng(7)
    big = min(nbytes // 2, (1 << 31) + 12345)
    while cuts[-1] < nbytes:
        step = big if len(cuts) == 3 else int(rng.integers(30_000_001, 400_000_003))
        cuts.append(min(nbytes, cuts[-1] + step))
    got = []
    t0 = time.time()
    for a, z in zip(cuts, cuts[1:]):
        st.feed_ptrs([buf.ptr + a], [z - a], on_device=True)
        got.append(st.eti(0))
    sess = np.concatenate(got)
    out["session"] = {"segments": len(cuts) - 1, "largest_segment_bytes": int(max(z - a for a, z in zip(cuts, cuts[1:]))), "eti_frames": int(sess.shape[0]),
                      "seconds": round(time.time() - t0, 3), "equal_to_one_shot": bool(sess.shape == whole.shape and np.array_equal(sess, whole)), "stream_status": st.status(0)}
    ok = ok and out["session"]["equal_to_one_shot"]
    st.close()
    # (c) one segment just below 4 GiB: the copy kernel's 64 KB pieces were counted in 32 bits, and for sizes within 4 MiB of 2^32 its loop offset
    # wrapped below the size again -- a kernel that never ended.  The first feed is such a segment, the second the rest.
    if nbytes > (1 << 32):
        st = dab.Stream(1, device=0)
        first = (1 << 32) - 70000
        t0 = time.time()
        got = []
        for a, z in ((0, first), (first, nbytes)):
            st.feed_ptrs([buf.ptr + a], [z - a], on_device=True)
            got.append(st.eti(0))
        sess = np.concatenate(got)
        out["session_segment_just_below_4gib"] = {"segment_bytes": first, "seconds": round(time.time() - t0, 3),
                                                  "equal_to_one_shot": bool(sess.shape == whole.shape and np.array_equal(sess, whole))}
        ok = ok and out["session_segment_just_below_4gib"]["equal_to_one_shot"]
        st.close()
    if args.oracle_tfs > 0:
        import oracle_lib as ol
        m = min(args.oracle_tfs, args.tfs)
        host = np.empty(m * dab.TF_BYTES, np.uint8)
        assert dab.lib().dabhip_device_copy(host.ctypes.data, buf.ptr, host.size, 0) == 0
        t0 = time.time()
        want, _ = ol.or_replay(host, cap_frames=4 * m)
        out["oracle"] = {"tfs": m, "eti_frames": int(want.shape[0]), "seconds": round(time.time() - t0, 1),
                         "equal_to_the_first_frames_of_one_shot": bool(np.array_equal(want, whole[: want.shape[0]]))}
        ok = ok and out["oracle"]["equal_to_the_first_frames_of_one_shot"] and want.shape[0] == 4 * (m - 15)
        # ... and the END of the capture: the oracle replays the last m TFs from scratch; once it has locked, its frames are the one-shot decode's last ones
        tail = np.empty(m * dab.TF_BYTES, np.uint8)
        assert dab.lib().dabhip_device_copy(tail.ctypes.data, buf.ptr + (args.tfs - m) * dab.TF_BYTES, tail.size, 0) == 0
        want2, _ = ol.or_replay(tail, cap_frames=4 * m)
        k = want2.shape[0]
        # the first 16 CIFs after a fresh lock mix in zero history where the long decode has real history: compare behind them
        eq = bool(k > 64 and np.array_equal(want2[64:], whole[whole.shape[0] - k + 64:]))
        out["oracle_tail"] = {"tfs": m, "eti_frames": int(k), "equal_to_the_last_frames_of_one_shot_after_the_first_64": eq, "first_byte_offset": int((args.tfs - m) * dab.TF_BYTES)}
        ok = ok and eq
    buf.free()
    out["ok"] = bool(ok)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tfs", type=int, default=11200)
    ap.add_argument("--oracle-tfs", type=int, default=400)
    a = ap.parse_args()
    out = run(a.tfs, a.oracle_tfs)
    print(json.dumps(out))
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
