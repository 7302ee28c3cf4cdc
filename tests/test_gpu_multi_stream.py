"""GPU tests of sessions over several devices (dabhip_multi_stream_*, `dab2eti-hip --stream --devices ...`) and of the CLI's operator
feedback / live-input defaults (item 7).  The reference's unit is one session on one device (dab2eti.c:60-130,237).  The pool's boxes have ONE GPU: the
slices are all mapped onto device 0 (a device may be listed more than once; every entry is its own session, host thread and HIP streams), which proves
the dealing rule, the slices' concurrency, the carried state and byte-equality -- not a scaling figure."""
import os
import select
import subprocess
import time

import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(dab.LIB_PATH), "dab2eti-hip")


def _captures(n, ntf=20):
    caps = []
    for i in range(n):
        cfg = dab.synth_preset(i % 2, seed=6200 + i, cif_count0=(433 * i) % 5000, skip_samples=(0, 50021, 0, 777)[i % 4],
                               snr_db=(1000.0, 11.0, 1000.0)[i % 3], cfo_hz=(0.0, 130.0, -410.0)[i % 3], amplitude=0.8)
        caps.append(dab.synth_generate(cfg, ntf + i % 3))
    return caps


def _segments(caps, cuts):
    """per segment: the streams' parts (a stream shorter than a cut contributes an empty part)"""
    return [[c[a:z] for c in caps] for a, z in zip(cuts, cuts[1:])]


@pytest.mark.parametrize("afc", [False, True])
def test_four_slices_on_one_gpu_equal_single_session_and_oracle(afc):
    """10 streams dealt to 4 slices (3, 3, 2, 2) on GPU 0, fed in segments of odd sizes (not multiples of a call, one of them a single byte): per
    segment and per stream the frames equal those of ONE dabhip_stream session over all streams fed the same segments; concatenated they equal the
    one-shot decode and (without AFC) the oracle; status, need_from and the operator messages agree; with the software AFC on, its state is carried
    across the segments in every slice."""
    caps = _captures(10)
    cuts = [0, 1000001, 1000002, 2621440, 4400000, 4400000 + 262144 * 3 + 17, 7000000, 10 ** 9]
    segs = _segments(caps, cuts)
    one = dab.Stream(len(caps), afc=afc)
    many = dab.MultiStream(len(caps), [0, 0, 0, 0], afc=afc)
    assert [many.slice_of(b)[0] for b in range(10)] == [0, 0, 0, 1, 1, 1, 2, 2, 3, 3]
    assert many.slice_of(7) == (2, 0, 6, 2)
    got = [[] for _ in caps]
    logs_one, logs_many = [""] * len(caps), [""] * len(caps)
    total = 0
    for parts in segs:
        n1 = one.feed(parts)
        n2 = many.feed(parts)
        assert n1 == n2
        total += n2
        for b in range(len(caps)):
            a, m = one.eti(b), many.eti(b)
            assert a.shape == m.shape and np.array_equal(a, m), b
            got[b].append(m)
            assert one.status(b) == many.status(b) == 0
            assert one.need_from(b) == many.need_from(b)
            logs_one[b] += one.log(b)
            logs_many[b] += many.log(b)
    assert total >= 40
    eng = dab.Engine(0)
    eng.set_afc(afc)
    assert eng.decode(caps) == total
    for b, iq in enumerate(caps):
        assert np.array_equal(np.concatenate(got[b]), eng.eti(b)), b
        assert logs_many[b] == logs_one[b] == eng.log(b), b
        if len(got[b]) and sum(len(g) for g in got[b]):   # a stream that emitted frames locked and showed its ensemble once
            assert logs_many[b].startswith("Locked\nENSEMBLE_INFO: EId=0x") and logs_many[b].count("ENSEMBLE_INFO") == 1, b
        if not afc and b in (0, 4, 9):
            assert np.array_equal(eng.eti(b), ol.or_replay(iq)[0]), b
    one.close()
    many.close()
    eng.close()


def test_multi_session_prefetch_fetch_pipeline_and_resident_feed():
    """The host-fed pipeline `dab2eti-hip --stream --devices` runs -- prefetch(k + 1) / feed(k) / eti_fetch into page-locked memory, the wait made while
    the next segment is already being fed -- and the in-place form (feed_resident over device buffers): the fetched bytes are the frames in global stream
    order, all segments together those of the one-shot decode.  More devices than streams: the extra slices stay empty."""
    caps = _captures(7, ntf=19)
    eng = dab.Engine(0)
    eng.decode(caps)
    want = [eng.eti(b) for b in range(len(caps))]
    eng.close()
    cuts = [0, 2800000, 2800000 + 262144 * 4, 5500001, 10 ** 9]
    segs = []
    for parts in _segments(caps, cuts):
        hbs = [dab.HostBuffer(max(p.size, 16)) for p in parts]
        for hb, p in zip(hbs, parts):
            hb.array[: p.size] = p
        segs.append(([hb.ptr for hb in hbs], [p.size for p in parts], hbs))
    many = dab.MultiStream(len(caps), [0, 0, 0])
    out = [dab.HostBuffer(64 * len(caps) * dab.ETI_BYTES) for _ in range(2)]
    got = [[] for _ in caps]
    pending = []                                         # (output buffer index, per-stream frame counts) of the fetches not yet waited for: up to TWO, as in the CLI
    many.prefetch_ptrs(*segs[0][:2])

    def collect(item):
        o, counts = item
        many.eti_fetch_wait()
        at = 0
        for b, n in enumerate(counts):
            got[b].append(out[o].array[at * dab.ETI_BYTES:(at + n) * dab.ETI_BYTES].reshape(n, dab.ETI_BYTES).copy())
            at += n

    for k in range(len(segs)):
        if k + 1 < len(segs):
            many.prefetch_ptrs(*segs[k + 1][:2])
        n = many.feed_ptrs(*segs[k][:2])
        if len(pending) == 2:
            collect(pending.pop(0))                      # (segment k - 2's frames: its output buffer is the one segment k's go to; segment k - 1's fetch stays outstanding)
        counts = [many.eti_count(b) for b in range(len(caps))]
        assert sum(counts) == n
        assert many.eti_fetch(out[k & 1].ptr, 64 * len(caps)) == n
        pending.append((k & 1, counts))
    while pending:
        collect(pending.pop(0))                          # oldest first: a slice that had no frames in a fetch is not waited on for it
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    many.close()
    # feed_resident: the captures live in device memory and grow; 9 slices for 7 streams
    many = dab.MultiStream(len(caps), [0] * 9)
    assert [many.slice_of(b)[0] for b in range(7)] == list(range(7))
    bufs = [dab.DeviceBuffer(c.size) for c in caps]
    for buf, c in zip(bufs, caps):
        buf.upload(c)
    got = [[] for _ in caps]
    for avail in (1500000, 1500000 + 262144, 4000001, 10 ** 9):
        many.feed_resident([b.ptr for b in bufs], [min(avail, c.size) for c in caps])
        for b in range(len(caps)):
            got[b].append(many.eti(b))
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    many.close()
    for buf in bufs:
        buf.free()
    for _, _, hbs in segs:
        for hb in hbs:
            hb.free()
    for o in out:
        o.free()
    with pytest.raises(dab.DabhipError):
        dab.MultiStream(3, [0, 99])
    with pytest.raises(dab.DabhipError):
        dab.MultiStream(0, [0])


def test_cli_stream_on_several_devices_equals_single_device_and_batch(tmp_path):
    """`dab2eti-hip --stream --devices 0,0,0 --segment-calls 5 f0 .. f6` (three sessions on GPU 0) writes the frames of `--stream` on one device segment
    by segment, and -- re-ordered from per-segment to per-file -- the bytes of the batch mode."""
    names, caps = [], _captures(7, ntf=19)
    for i, iq in enumerate(caps):
        p = tmp_path / ("cap%d.cu8" % i)
        iq.tofile(p)
        names.append(str(p))
    one = subprocess.run([EXE, "--stream", "--segment-calls", "5"] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    three = subprocess.run([EXE, "--stream", "--segment-calls", "5", "--devices", "0,0,0", "--stats"] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert len(one.stdout) >= 32 * dab.ETI_BYTES and one.stdout == three.stdout      # (the captures with a carrier offset do not decode without --afc)
    err = three.stderr.decode()
    assert "(device 0)" in err and '"devices": 3' in err
    # operator messages, per input (prefixed: several inputs); --quiet removes them
    emitted = [name for name in names if ("%s: 0 ETI frames" % name) not in err]
    assert len(emitted) >= 2
    for name in emitted:
        assert "%s: Locked\n" % name in err and "%s: ENSEMBLE_INFO: EId=0x" % name in err
    quiet = subprocess.run([EXE, "--stream", "--quiet", "--devices", "0,0"] + names[:2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert b"Locked" not in quiet.stderr and b"ETI frames" in quiet.stderr
    batch = subprocess.run([EXE] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert ("%s: Locked\n" % emitted[0]).encode() in batch.stderr
    a = np.frombuffer(three.stdout, np.uint8).reshape(-1, dab.ETI_BYTES)
    b = np.frombuffer(batch.stdout, np.uint8).reshape(-1, dab.ETI_BYTES)
    assert a.shape == b.shape
    assert sorted(f.tobytes() for f in a) == sorted(f.tobytes() for f in b)


def test_cli_live_stdin_first_frame_latency_and_reference_stderr(tmp_path):
    """A live source on stdin (`rtl_sdr ... - | dab2eti-hip -`, dab2eti.c:117-130): samples arrive at the real-time rate (we feed 2 x real time to keep the
    test short), the CLI picks short segments by itself because stdin is a pipe, and the first ETI frame is on stdout less than 0.3 s after the last byte of
    call that lets the first frame be assembled was written (lock after 10 TFs + a ring of 16 CIFs).  stderr carries the reference's operator text (dab.c:51,57,
    78-82) for a capture with a lock loss -- the text the batch engine reports for the same capture, which tests/test_host.py holds against the REAL
    reference's stderr -- and nothing else but the frame count."""
    cfg = dab.synth_preset(1, seed=6301, cif_count0=120)
    iq = dab.synth_generate(cfg, 38).copy()
    # destroy the FIC symbols of the 20th frame: the lock is lost and found again ten frames later (dab.c:48-61)
    tf = 20
    a = tf * dab.TF_BYTES + 2 * (2656 + 2552)
    iq[a:a + 3 * 2 * 2552] = 127
    want_eng = dab.Engine(0)
    want_eng.decode([iq])
    want = want_eng.eti(0).tobytes()
    want_log = want_eng.log(0)
    want_eng.close()
    assert want_log.count("Locked\n") == 2 and "Lock lost, resetting ringbuffer\n" in want_log and want_log.count("ENSEMBLE_INFO") == 1
    p = subprocess.Popen([EXE, "-"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    os.set_blocking(p.stdout.fileno(), False)
    call = dab.CHUNK_BYTES
    rate = 2 * 4096000.0                                   # bytes per second: twice real time
    out, first_frame_at, written_at = b"", None, {}
    ncalls = iq.size // call
    time.sleep(1.0)                                        # HIP initialisation and buffer page-locking: not part of the latency under test
    t0 = time.monotonic()
    for k in range(ncalls):
        due = t0 + (k + 1) * call / rate
        while time.monotonic() < due:
            time.sleep(0.001)
        p.stdin.write(iq[k * call:(k + 1) * call].tobytes())
        p.stdin.flush()
        written_at[k] = time.monotonic()
        r, _, _ = select.select([p.stdout], [], [], 0)
        if r:
            chunk = p.stdout.read()
            if chunk:
                if first_frame_at is None:
                    first_frame_at = (time.monotonic(), k)
                out += chunk
    p.stdin.close()
    os.set_blocking(p.stdout.fileno(), True)
    out += p.stdout.read()
    err = p.stderr.read().decode()
    assert p.wait() == 0, err
    assert out == want
    # the first call whose arrival lets a session emit frames (found offline, call by call), and when the live run's first frames appeared
    st = dab.Stream(1)
    k_emit = next(k for k in range(ncalls) if st.feed([iq[k * call:(k + 1) * call]]) > 0)
    st.close()
    assert first_frame_at is not None
    t_first, k_first = first_frame_at
    assert k_first >= k_emit
    assert k_first - k_emit <= 2, (k_first, k_emit)         # segments of two calls: the frames leave with the segment that holds call k_emit
    assert t_first - written_at[k_emit] < 0.3, (t_first - written_at[k_emit], k_first, k_emit)
    lines = [ln for ln in err.splitlines() if not ln.endswith("ETI frames")]
    assert "\n".join(lines) + "\n" == want_log, err
