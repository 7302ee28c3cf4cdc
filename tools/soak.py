#!/usr/bin/env python3
"""tools/soak.py -- a session left running (GPU box): does anything grow, drift or wrap?

B streams (default 4: clean, 16 dB, 20 dB - 60 Hz, 30 dB + 25 Hz; both synthetic multiplexes), each a capture of --loop-tf transmission frames
(default 125 = 500 CIFs, so that the ETI frame counter FCT, mod 250, is continuous where the capture starts over) played round and round, are fed to ONE
session in live-sized segments (--calls 262,144-byte calls per feed, default 2 = what `dab2eti-hip -` picks for a pipe; every 7th feed an odd size
instead) until every stream is --total-tf frames
long (default 11,200 = 4.4 GB: byte offsets past 2^32).  That is ~8,400 feeds of 4 streams.  Checked:
  * every ETI frame against the frame 1000 frames (two rounds) earlier, once both are past the first round: the same signal bytes -> the same frame,
    whatever the receiver's carried state (1000 because the header's frame phase FP counts mod 8 and FCT mod 250; the FIG 0/0 CIF counter's upper part
    jumps where the capture starts over, and it is part of the bytes, so it jumps alike every round).  Required of the streams that are on tune; the
    two that are off tune with the AFC off (dab2eti would re-tune the dongle) decode with every symbol's phase 12 .. 27 degrees towards a decision
    boundary, their decoded bits hang on which samples the time synchronisation put into each window, and that differs from round to round: their
    count is reported, and what holds them is the oracle (next item);
  * the first --oracle-tf frames of every stream (default 700: five and a half rounds, non-periodic frames included) against the CPU oracle's bytes;
  * frames per stream = 4 (T - 15), stream status 0, FCT stepping by one throughout;
  * the process's heap in use (mallinfo2), resident set and the device's free memory, sampled every 256 feeds: flat once the first rounds are through
    (page-locked staging and the windows are sized by then, and this script has stopped keeping frames for the oracle).  Round 6 found 2 .. 3.6 KB
    per feed here: the HIP runtime's records of copies on streams nobody synchronised (tools/hip_retained_commands.py; engine.hpp: blocking_copy).
One JSON object; exit code 1 when a check fails.  Checker use of oracle/ only (like tests/)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rss_kb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1])
    return -1


class _MallInfo(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks", "uordblks", "fordblks", "keepcost")]


_libc = C.CDLL("libc.so.6")
_libc.mallinfo2.restype = _MallInfo


def heap_in_use_kb():
    """bytes the process holds from malloc (glibc): what a leak -- or a runtime that never lets go of its records -- shows up in first"""
    return int(_libc.mallinfo2().uordblks) // 1024


def device_free_bytes(hip):
    free, total = C.c_size_t(0), C.c_size_t(0)
    return int(free.value) if hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0 else -1


def run(streams=4, loop_tf=125, total_tf=11200, calls=2, oracle_tf=700, devices=None):
    import dabtools_amd as dab
    hip = C.CDLL("libamdhip64.so")
    snr = [1000.0, 16.0, 20.0, 30.0]      # (noise levels at which no decoded bit hangs on a marginal decision: see the docstring's second check)
    cfo = [0.0, 0.0, -60.0, 25.0]           # (without the AFC the receiver follows a few tens of Hz only: dab2eti re-tunes the dongle for more)
    caps = [dab.synth_generate(dab.synth_preset(b % 2, seed=66000 + b, cif_count0=0, snr_db=snr[b % 4], cfo_hz=cfo[b % 4]), loop_tf) for b in range(streams)]
    loop_bytes = loop_tf * dab.TF_BYTES
    assert all(c.size == loop_bytes for c in caps)
    total_bytes = total_tf * dab.TF_BYTES
    st = dab.MultiStream(streams, devices) if devices else dab.Stream(streams)
    period = int(np.lcm(4 * loop_tf, 1000))                  # ETI frames after which the same bytes are due again (FCT mod 250, FP mod 8)
    ref = [np.zeros((period, dab.ETI_BYTES), np.uint8) for _ in range(streams)]     # the newest `period` frames of a stream, by frame index mod period
    count = [0] * streams
    first = [[] for _ in range(streams)]                     # the first frames, for the oracle
    differ = [0] * streams
    differ_in_window = [0] * streams                         # ... of them, among the frames the oracle checks
    compared = [0] * streams
    fct_bad = [0] * streams
    last_fct = [None] * streams
    rng = np.random.default_rng(20261003)
    fed, feeds, samples = 0, 0, []
    t0 = time.time()
    while fed < total_bytes:
        n = calls * dab.CHUNK_BYTES
        if feeds % 7 == 6:
            n = int(rng.integers(1, 6 * dab.CHUNK_BYTES)) | 1                    # odd sizes, from one byte to six calls
        n = min(n, total_bytes - fed)
        a = fed % loop_bytes
        segs = []
        for c in caps:
            piece = c[a:a + n]
            left = n - piece.size
            while left > 0:                                   # across the place where the capture starts over
                more = c[:left]
                piece = np.concatenate([piece, more])
                left -= more.size
            segs.append(piece)
        st.feed(segs)
        fed += n
        feeds += 1
        for b in range(streams):
            k = st.eti_count(b)
            if k == 0:
                continue
            fr = st.eti(b)
            fct = fr[:, 4].astype(np.int32)
            if last_fct[b] is not None:
                fct_bad[b] += int(np.count_nonzero((np.diff(np.concatenate([[last_fct[b]], fct])) % 250) != 1))
            else:
                fct_bad[b] += int(np.count_nonzero((np.diff(fct) % 250) != 1))
            last_fct[b] = int(fct[-1])
            for i in range(k):
                g = count[b] + i
                if g < 4 * oracle_tf:
                    first[b].append(fr[i].copy())
                slot = g % period
                if g >= period + 4 * loop_tf:                 # both frames past the first round (lock-in, 16 CIFs of empty history)
                    compared[b] += 1
                    if not np.array_equal(ref[b][slot], fr[i]):
                        differ[b] += 1
                        differ_in_window[b] += g < 4 * (oracle_tf - 15)
                ref[b][slot] = fr[i]
            count[b] += k
        if feeds % 256 == 0:
            samples.append((fed // dab.TF_BYTES, rss_kb(), device_free_bytes(hip), heap_in_use_kb()))
    seconds = time.time() - t0
    status = [st.status(b) for b in range(streams)]
    st.close()
    out = {"what": "%d looped captures of %d TF fed to one session in %d-call segments (every 7th an odd size) up to %d TF = %.2f GiB per stream"
                   % (streams, loop_tf, calls, total_tf, total_bytes / 2.0 ** 30),
           "devices": devices or [0], "feeds": feeds, "seconds": round(seconds, 1), "x_realtime_per_stream": round(total_tf * 0.096 / seconds, 1),
           "eti_frames": count, "expected_per_stream": 4 * (total_tf - 15), "stream_status": status, "fct_steps_wrong": fct_bad,
           "frames_compared_with_%d_frames_earlier" % period: compared, "frames_that_differ": differ}
    on_tune = [b for b in range(streams) if cfo[b % 4] == 0.0]
    out["periodicity_required_of_streams"] = on_tune
    ok = (all(c == 4 * (total_tf - 15) for c in count) and not any(status) and not any(fct_bad) and not any(differ[b] for b in on_tune) and
          all(c > 0 for c in compared))
    # memory: after the first round's samples, resident set and device memory must not move by more than noise
    # (from the point where this script stops keeping frames for the oracle: those are its own 6 KB per frame)
    settled = [s for s in samples if s[0] >= max(2 * loop_tf, oracle_tf + 32)]
    if len(settled) >= 4:
        rss = [s[1] for s in settled]
        free = [s[2] for s in settled]
        heap = [s[3] for s in settled]
        out["memory"] = {"samples": len(settled), "heap_in_use_kb_by_sample": heap, "rss_kb_by_sample": rss, "rss_kb_first": rss[0], "rss_kb_last": rss[-1], "rss_kb_max": max(rss),
                         "device_free_first": free[0], "device_free_last": free[-1], "device_free_min": min(free)}
        # heap in use: 1 MB over thousands of feeds (256 bytes per feed) is noise; a runtime that keeps a record per copy shows 10 .. 20 MB here
        # (tools/hip_retained_commands.py).  Resident set: fragmentation of thousands of numpy temporaries allowed for.
        out["memory"]["flat"] = bool(heap[-1] - heap[0] < 1024 and rss[-1] - rss[0] < 16 * 1024 and free[0] - free[-1] < (64 << 20))
        ok = ok and out["memory"]["flat"]
    if oracle_tf > 0:
        import oracle_lib as ol
        m = min(oracle_tf, total_tf)
        eq, secs = [], 0.0
        for b in range(streams):
            reps = -(-m // loop_tf)
            iq = np.concatenate([caps[b]] * reps)[: m * dab.TF_BYTES]
            t1 = time.time()
            want, _ = ol.or_replay(iq, cap_frames=4 * m)
            secs += time.time() - t1
            got = np.array(first[b][: want.shape[0]])
            eq.append(bool(want.shape[0] == 4 * (m - 15) and got.shape == want.shape and np.array_equal(got, want)))
        out["oracle"] = {"tfs": m, "seconds": round(secs, 1), "first_frames_equal": eq, "non_periodic_frames_among_them": [int(x) for x in differ_in_window]}
        ok = ok and all(eq)
    out["ok"] = bool(ok)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--loop-tf", type=int, default=125)
    ap.add_argument("--total-tf", type=int, default=11200)
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--oracle-tf", type=int, default=700)
    ap.add_argument("--devices", default="", help="comma-separated: a session over several devices (dabhip_multi_stream); a device may be listed twice")
    a = ap.parse_args()
    out = run(a.streams, a.loop_tf, a.total_tf, a.calls, a.oracle_tf, [int(x) for x in a.devices.split(",")] if a.devices else None)
    print(json.dumps(out))
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
