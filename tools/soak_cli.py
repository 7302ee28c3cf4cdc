#!/usr/bin/env python3
"""tools/soak_cli.py -- `dab2eti-hip -` (or `--stream` over several inputs) left running: does the PROCESS grow? (GPU box)

One capture of --loop-tf transmission frames (default 125: FCT continuous where it starts over) is written to the CLI's stdin round and round in
262,144-byte writes, --total-tf frames in all (default 6,000 = 9,000 calls = 4,500 of the 2-call segments the CLI picks for a pipe: prefetch on the
upload stream, feed, fetch on the download stream, the writer thread waiting for it -- the paths a live receiver runs for weeks); stdout is read and
counted by a thread.  Every 0.2 s the child's /proc/PID/status is sampled (RssAnon = heap + anonymous maps; page-locked buffers are RssShmem / RssFile).
Checked: frame count = 4 (T - 15) - what the CLI still holds back at EOF (none: stdin closes, everything is flushed), and RssAnon flat over the
second half of the run (less than 1 MB; round 6 found 2 KB per segment here, tools/hip_retained_commands.py).
With --inputs N > 1 the CLI gets N named pipes and `--stream` (one session of N streams); --devices as `dab2eti-hip --devices`.
One JSON object; exit code 1 when a check fails."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dabtools_amd as dab  # noqa: E402

EXE = os.path.join(ROOT, "dabtools_amd", "dab2eti-hip")


def status_kb(pid):
    out = {}
    try:
        with open("/proc/%d/status" % pid) as f:
            for line in f:
                k = line.split(":")[0]
                if k in ("VmRSS", "RssAnon", "RssFile", "RssShmem"):
                    out[k] = int(line.split()[1])
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loop-tf", type=int, default=125)
    ap.add_argument("--total-tf", type=int, default=6000)
    ap.add_argument("--inputs", type=int, default=1)
    ap.add_argument("--devices", default="")
    a = ap.parse_args()
    caps = [dab.synth_generate(dab.synth_preset(b % 2, seed=67000 + b, snr_db=[1000.0, 18.0][b % 2]), a.loop_tf) for b in range(a.inputs)]
    call = dab.CHUNK_BYTES
    ncalls = a.total_tf * dab.TF_BYTES // call
    tmp = tempfile.mkdtemp(prefix="soakcli")
    if a.inputs == 1:
        cmd = [EXE, "--quiet", "-"]
    else:
        fifos = [os.path.join(tmp, "in%d.iq" % b) for b in range(a.inputs)]
        for f in fifos:
            os.mkfifo(f)
        cmd = [EXE, "--quiet", "--stream"] + (["--devices", a.devices] if a.devices else []) + fifos
    p = subprocess.Popen(cmd, stdin=subprocess.PIPE if a.inputs == 1 else None, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    got = [0]

    def drain():
        while True:
            chunk = p.stdout.read(1 << 20)
            if not chunk:
                return
            got[0] += len(chunk)
    err = []
    rd = threading.Thread(target=drain)
    rd.start()
    er = threading.Thread(target=lambda: err.append(p.stderr.read()))
    er.start()
    written = [0] * a.inputs

    def feed(b):
        sink = p.stdin if a.inputs == 1 else open(fifos[b], "wb", buffering=0)
        c = caps[b]
        pos = 0
        for k in range(ncalls):                              # 125 TF = 187.5 calls: walk the capture by bytes, not by calls
            piece = bytes(c[pos:pos + call])
            if len(piece) < call:
                piece += bytes(c[:call - len(piece)])
            try:
                sink.write(piece)
            except BrokenPipeError:
                break
            pos = (pos + call) % c.size
            written[b] = k + 1
        sink.close()
    samples = []
    t0 = time.time()
    feeders = [threading.Thread(target=feed, args=(b,)) for b in range(a.inputs)]
    for t in feeders:
        t.start()
    while any(t.is_alive() for t in feeders):
        time.sleep(0.2)
        s = status_kb(p.pid)
        s["calls"] = min(written)
        samples.append(s)
    for t in feeders:
        t.join()
    rd.join()
    er.join()
    rc = p.wait()
    seconds = time.time() - t0
    frames = got[0] // dab.ETI_BYTES
    # (samples taken once everything has been written show the process coming down, not running: left out)
    half = [s for s in samples if ncalls // 2 <= s["calls"] < ncalls and "RssAnon" in s]
    growth = half[-1]["RssAnon"] - half[0]["RssAnon"] if len(half) >= 2 else None
    out = {"what": "%s fed %d x %d calls of 262,144 bytes (a %d-TF capture round and round) through %s" %
                   (" ".join(os.path.basename(x) if x.startswith("/") else x for x in cmd[:4]), a.inputs, ncalls, a.loop_tf, "stdin" if a.inputs == 1 else "named pipes"),
           "rc": rc, "seconds": round(seconds, 1), "x_realtime_per_input": round(a.total_tf * 0.096 / seconds, 1), "eti_frames": frames,
           "expected": a.inputs * 4 * (ncalls * call // dab.TF_BYTES - 15),
           "rss_anon_kb_by_sample": [s.get("RssAnon") for s in samples], "rss_anon_growth_kb_over_the_second_half": growth,
           "stderr_tail": (err[0] or b"").decode("ascii", "replace")[-400:]}
    # frames: whole TFs only reach the decoder; the capture's tail (less than a TF) may hold back up to 4 frames per input
    ok = rc == 0 and abs(frames - out["expected"]) <= 4 * a.inputs and growth is not None and growth < 1024
    out["ok"] = bool(ok)
    print(json.dumps(out))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
