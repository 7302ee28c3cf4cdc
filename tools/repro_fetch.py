#!/usr/bin/env python3
"""bench.py's host-fed session loop (prefetch k + 1 / feed k / wait k - 1 / fetch k) at a small size on noisy captures: the loop that reported
"two fetches are outstanding" in the 5 and 7 dB bench lines of round 5."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import dabtools_amd as dab  # noqa: E402

B, ntf, seg_tfs = int(os.environ.get("STREAMS", 8)), 64, 8
snr = float(os.environ.get("SNR", 5.0))
caps = [dab.synth_generate(dab.synth_preset(0, seed=2000 + i, cif_count0=97 * i, snr_db=snr), ntf) for i in range(B)]
nbytes = caps[0].size
pinned = [dab.HostBuffer(nbytes) for _ in range(B)]
for hb, c in zip(pinned, caps):
    hb.array[:] = c
eti_host = [dab.HostBuffer(B * 4 * ntf * 6144) for _ in range(2)]
seg = seg_tfs * dab.TF_BYTES
cuts = list(range(0, nbytes, seg)) + [nbytes]
segs = [([hb.ptr + a for hb in pinned], [z - a] * B) for a, z in zip(cuts, cuts[1:])]
st = dab.Stream(B)
try:
    st.prefetch_ptrs(*segs[0])
    for k in range(len(segs)):
        if k + 1 < len(segs):
            st.prefetch_ptrs(*segs[k + 1])
        n = st.feed_ptrs(*segs[k])
        if k >= 1:
            st.eti_fetch_wait()
        got = st.eti_fetch(eti_host[k & 1].ptr, n)
        print("segment", k, "frames", n, "fetched", got, flush=True)
    st.eti_fetch_wait()
    print("ok")
except Exception:
    traceback.print_exc()
    sys.exit(1)
