#!/usr/bin/env python3
"""`rtl_sdr ... - | dab2eti-hip -` as a live source would drive it (dab2eti.c:117-135): one capture written to the CLI's stdin in 262,144-byte calls at the real-time
rate (RATE x real time, default 1), stdout polled after every call.  Reports, per ETI burst, how long after the write of the call that completed it the frames
were on stdout (the CLI picks segments of 2 calls by itself because stdin is a pipe).  One JSON object (profiles/r06_live_latency.json)."""
import json
import os
import select
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import dabtools_amd as dab  # noqa: E402

EXE = os.path.join(ROOT, "dabtools_amd", "dab2eti-hip")
rate_x = float(os.environ.get("RATE", 1.0))
ntf = int(os.environ.get("TFS", 40))
iq = dab.synth_generate(dab.synth_preset(1, seed=6301, cif_count0=120), ntf)
call = dab.CHUNK_BYTES
ncalls = iq.size // call
st = dab.Stream(1)
per_call = [st.feed([iq[k * call:(k + 1) * call]]) for k in range(ncalls)]       # frames each call lets a session emit (offline, call by call)
st.close()
p = subprocess.Popen([EXE, "-"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
os.set_blocking(p.stdout.fileno(), False)
time.sleep(1.5)                                          # HIP initialisation and buffer page-locking: not part of the latency
rate = rate_x * 4096000.0
t0 = time.monotonic()
written, arrivals, got = {}, [], 0
for k in range(ncalls):
    due = t0 + (k + 1) * call / rate
    while time.monotonic() < due:
        r, _, _ = select.select([p.stdout], [], [], 0.0005)
        if r:
            chunk = p.stdout.read()
            if chunk:
                got += len(chunk)
                arrivals.append((time.monotonic(), got // 6144))
    p.stdin.write(iq[k * call:(k + 1) * call].tobytes())
    p.stdin.flush()
    written[k] = time.monotonic()
p.stdin.close()
os.set_blocking(p.stdout.fileno(), True)
rest = p.stdout.read()
got += len(rest)
arrivals.append((time.monotonic(), got // 6144))
p.wait()
# frame f (1-based count) is complete when the call that lets the session emit it has been written
cum, need = 0, []
for k, n in enumerate(per_call):
    for _ in range(n):
        cum += 1
        need.append(k)
lat = []
for t, count in arrivals:
    while len(lat) < count:
        f = len(lat)
        lat.append(t - written[min(need[f], ncalls - 1)])
lat_ms = [1e3 * x for x in lat]
print(json.dumps({"what": "dab2eti-hip - fed through a pipe at %.1f x real time, %d TF, 262,144-byte writes; latency = frame on stdout - write of the call that completes it" % (rate_x, ntf),
                  "segment_calls": "default for a pipe (2 = 128 ms of signal)", "eti_frames": got // 6144, "expected": sum(per_call),
                  "first_frame_ms": lat_ms[0] if lat_ms else None, "median_ms": float(np.median(lat_ms)) if lat_ms else None,
                  "p95_ms": float(np.percentile(lat_ms, 95)) if lat_ms else None, "max_ms": max(lat_ms) if lat_ms else None,
                  "note": "a frame waits for the rest of its 2-call segment (up to one call = 64 ms of signal at real time) and for the poll (0.5 ms)"}))
