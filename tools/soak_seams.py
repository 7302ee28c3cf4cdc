#!/usr/bin/env python3
"""tools/soak_seams.py -- the reference's own call pattern, for a long time: sdr_demod + dab_process_frame (S2 + S3) per 262,144-byte buffer (GPU box).

A --loop-tf capture (default 125 TF) goes round and round through dabhip_sdr_demod (input_sdr.h:44) call by call; every transmission frame it returns is handed
to dabhip_dab_process_frame (dab.h:92), whose ETI frames come back through the callback -- what `dab2eti` does per buffer (dab2eti.c:40-69) with
integration/input_sdr_hip.c and integration/dab_hip.c in place.  --calls buffers in all (default 9,000 = 6,000 TF = 9.6 minutes of signal).  Checked:
ETI frames = 4 (T - 15) for the T transmission frames fed, FCT stepping by one, every frame equal to the frame 1000 earlier (tools/soak.py has the why) and the
first --oracle-tf TF equal to the CPU oracle's bytes; heap in use (mallinfo2) sampled every 250 calls: flat over the second half (each call makes three to
five blocking copies; round 6 found the HIP runtime keeping up to 970 bytes per copy, tools/hip_retained_commands.py).
One JSON object; exit code 1 when a check fails.  Checker use of oracle/ only (like tests/)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from soak import heap_in_use_kb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loop-tf", type=int, default=125)
    ap.add_argument("--calls", type=int, default=9000)
    ap.add_argument("--oracle-tf", type=int, default=300)
    a = ap.parse_args()
    import dabtools_amd as dab
    cap = dab.synth_generate(dab.synth_preset(0, seed=68001, snr_db=18.0), a.loop_tf)
    call = dab.CHUNK_BYTES
    sdr, back = dab.Sdr(), dab.Dab()
    pos, tfs, checked, differ, fct_bad = 0, 0, 0, 0, 0
    period = 1000
    ring = np.zeros((period, dab.ETI_BYTES), np.uint8)
    first, count, last_fct, samples = [], 0, None, []
    t0 = time.time()
    for k in range(a.calls):
        piece = cap[pos:pos + call]
        if piece.size < call:
            piece = np.concatenate([piece, cap[:call - piece.size]])
        pos = (pos + call) % cap.size
        if sdr.demod(piece) == 1:
            tfs += 1
            back.fic[:] = sdr.fic
            back.msc[:] = sdr.msc
            back.process_frame()
            for fr in back.frames:
                if last_fct is not None and (int(fr[4]) - last_fct) % 250 != 1:
                    fct_bad += 1
                last_fct = int(fr[4])
                if count < 4 * a.oracle_tf:
                    first.append(fr)
                slot = count % period
                if count >= period + 4 * a.loop_tf:
                    checked += 1
                    differ += not np.array_equal(ring[slot], fr)
                ring[slot] = fr
                count += 1
            back.frames.clear()
        if k % 250 == 249:
            samples.append(heap_in_use_kb())
    seconds = time.time() - t0
    status = back.status
    sdr.close()
    back.close()
    half = samples[len(samples) // 2:]
    expected = 4 * (a.calls * call // dab.TF_BYTES - 15)         # dab2eti's count for a capture of that many whole transmission frames
    out = {"what": "dabhip_sdr_demod + dabhip_dab_process_frame, %d buffers of 262,144 bytes (a %d-TF capture round and round, 18 dB)" % (a.calls, a.loop_tf),
           "seconds": round(seconds, 1), "x_realtime": round(a.calls * 0.064 / seconds, 2), "transmission_frames": tfs, "eti_frames": count, "expected": expected,
           "back_end_status": status, "fct_steps_wrong": fct_bad, "frames_compared_with_1000_frames_earlier": checked, "frames_that_differ": int(differ),
           "heap_in_use_kb_by_sample": samples, "heap_growth_kb_over_the_second_half": half[-1] - half[0]}
    ok = count == expected and status == 0 and fct_bad == 0 and differ == 0 and checked > 0 and half[-1] - half[0] < 512
    if a.oracle_tf > 0:
        import oracle_lib as ol
        m = a.oracle_tf
        iq = np.concatenate([cap] * (-(-m // a.loop_tf)))[: m * dab.TF_BYTES]
        want, _ = ol.or_replay(iq, cap_frames=4 * m)
        got = np.array(first[: want.shape[0]])
        out["oracle"] = {"tfs": m, "eti_frames": int(want.shape[0]), "equal": bool(want.shape[0] == 4 * (m - 15) and got.shape == want.shape and np.array_equal(got, want))}
        ok = ok and out["oracle"]["equal"]
    out["ok"] = bool(ok)
    print(json.dumps(out))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
