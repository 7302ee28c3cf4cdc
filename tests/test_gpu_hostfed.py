"""GPU tests of the host-fed paths: the reference's input arrives in host buffers (dab2eti.c:117-130,238), so the batch entry
and the streaming sessions take host memory -- page-locked (plain asynchronous DMA) or pageable (through the engine's staging
ring) -- and must produce the ETI bytes of the device-resident decode."""
import os

import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _caps():
    caps = []
    for i in range(7):
        cfg = dab.synth_preset(i % 2, seed=4100 + i, cif_count0=333 * i, skip_samples=(0, 9001, 150000)[i % 3], snr_db=(1000.0, 11.0)[i % 2])
        iq = dab.synth_generate(cfg, 19 + i % 2)
        if i in (2, 5):
            iq = iq[: iq.size - (1237 + 16 * i)]           # ragged: neither whole calls nor a multiple of 16 bytes
        caps.append(iq)
    caps.append(np.zeros(0, np.uint8))                     # an empty stream in the batch
    return caps


def test_host_fed_decode_pinned_pageable_and_mixed_equal_resident_decode():
    caps = _caps()
    eng = dab.Engine(0)
    total = eng.decode(caps)                               # pageable numpy arrays: staging ring
    st = eng.stage_ms()
    assert abs(st["h2d_mbytes"] - sum(c.size for c in caps) * 1e-6) < 1e-2 and st["h2d_pinned_mbytes"] == 0 and st["h2d"] > 0
    want = [eng.eti(b) for b in range(len(caps))]
    assert total == sum(len(w) for w in want) > 7 * 8
    for b in (0, 2, 5):
        assert np.array_equal(want[b], ol.or_replay(caps[b])[0]), b
    # the same bytes resident on the device
    bufs = [dab.DeviceBuffer(max(c.size, 16)) for c in caps]
    for buf, c in zip(bufs, caps):
        if c.size:
            buf.upload(c)
    assert eng.decode_device([b.ptr for b in bufs], [c.size for c in caps]) == total
    assert eng.stage_ms()["h2d_mbytes"] == 0
    for b, w in enumerate(want):
        assert np.array_equal(eng.eti(b), w), b
    # page-locked host memory (all streams), then a mix of page-locked and pageable streams in one batch
    pinned = [dab.HostBuffer(max(c.size, 16)) for c in caps]
    for hb, c in zip(pinned, caps):
        hb.array[: c.size] = c
    assert eng.decode_host_ptrs([hb.ptr for hb in pinned], [c.size for c in caps]) == total
    st = eng.stage_ms()
    assert abs(st["h2d_pinned_mbytes"] - st["h2d_mbytes"]) < 1e-3 and st["h2d_mbytes"] > 50
    for b, w in enumerate(want):
        assert np.array_equal(eng.eti(b), w), b
    mixed = [pinned[b].ptr if b % 2 else caps[b].ctypes.data for b in range(len(caps))]
    assert eng.decode_host_ptrs(mixed, [c.size for c in caps]) == total
    st = eng.stage_ms()
    assert 0 < st["h2d_pinned_mbytes"] < st["h2d_mbytes"]
    for b, w in enumerate(want):
        assert np.array_equal(eng.eti(b), w), b
    # device memory handed in through the HOST form of the call (a caller's mistake the library absorbs: the copy engine knows the pointer)
    devs = [b.ptr if c.size else pinned[i].ptr for i, (b, c) in enumerate(zip(bufs, caps))]
    assert eng.decode_host_ptrs(devs, [c.size for c in caps]) == total
    for b, w in enumerate(want):
        assert np.array_equal(eng.eti(b), w), b
    for x in bufs + pinned:
        x.free()
    eng.close()


def test_stream_session_with_prefetched_segments_equals_one_shot_decode():
    """dabhip_stream_prefetch: segment k + 1 uploads on its own stream while segment k decodes (three device windows per
    stream).  Segments of unrelated sizes in page-locked memory; order prefetch(0) feed(0) / prefetch(k+1) feed(k), with one
    plain feed in between; the concatenated frames equal the one-shot decode; misuse is refused."""
    caps = _caps()[:6]
    eng = dab.Engine(0)
    eng.decode(caps)
    want = [eng.eti(b) for b in range(len(caps))]
    eng.close()
    cuts = [0, 3000000, 3000000 + 262144 * 5, 5500000, 5500001, 6900000, 10 ** 9]
    segs = []                                              # per segment: (HostBuffers, sizes)
    for a, z in zip(cuts, cuts[1:]):
        parts = [c[a:z] for c in caps]
        hbs = [dab.HostBuffer(max(p.size, 16)) for p in parts]
        for hb, p in zip(hbs, parts):
            hb.array[: p.size] = p
        segs.append(([hb.ptr for hb in hbs], [p.size for p in parts], hbs))
    st = dab.Stream(len(caps))
    got = [[] for _ in caps]

    def collect():
        for b in range(len(caps)):
            got[b].append(st.eti(b))

    st.prefetch_ptrs(*segs[0][:2])
    for k in range(len(segs)):
        if k == 3:                                         # a segment that was not prefetched: plain feed (nothing may be waiting)
            st.feed_ptrs(*segs[k][:2])
            collect()
            if k + 1 < len(segs):
                st.prefetch_ptrs(*segs[k + 1][:2])
            continue
        if k + 1 < len(segs) and k + 1 != 3:
            st.prefetch_ptrs(*segs[k + 1][:2])             # uploads while segment k decodes
        st.feed_ptrs(*segs[k][:2])
        collect()
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    st.close()
    # the same through the copy engine instead of the gather kernel (a fresh process: the choice is read once)
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import dabtools_amd as dab\n"
            "cfg = dab.synth_preset(1, seed=4100); iq = dab.synth_generate(cfg, 19); e = dab.Engine(0); e.decode([iq]); want = e.eti(0); e.close()\n"
            "hb = dab.HostBuffer(iq.size); hb.array[:] = iq; st = dab.Stream(1); got = []\n"
            "cuts = [0, 2500000, 5000001, iq.size]\n"
            "segs = [([hb.ptr + a], [z - a]) for a, z in zip(cuts, cuts[1:])]\n"
            "st.prefetch_ptrs(*segs[0])\n"
            "for k in range(3):\n"
            "    if k < 2: st.prefetch_ptrs(*segs[k + 1])\n"
            "    st.feed_ptrs(*segs[k]); got.append(st.eti(0))\n"
            "assert np.array_equal(np.concatenate(got), want) and len(want) > 8; print('copy-engine prefetch ok')\n") % os.path.dirname(os.path.dirname(os.path.abspath(dab.__file__)))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DABHIP_PREFETCH_KERNEL="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "copy-engine prefetch ok" in r.stdout, r.stderr[-2000:]
    # ... and with a window reserve too small for the history K1 still needs (64 KB): every feed moves its segment into a larger window
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DABHIP_WINDOW_RESERVE="65536"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and "copy-engine prefetch ok" in r.stdout, r.stderr[-2000:]
    # segments that are already on the device (on_device = 1) prefetch as device-to-device copies on the upload stream
    st = dab.Stream(len(caps))
    dsegs = []
    for ptrs_k, sizes_k, hbs in segs:
        bufs = [dab.DeviceBuffer(max(n, 16)) for n in sizes_k]
        for buf, hb, n in zip(bufs, hbs, sizes_k):
            if n:
                buf.upload(hb.array[:n])
        dsegs.append(([b.ptr for b in bufs], sizes_k, bufs))
    got = [[] for _ in caps]
    st.prefetch_ptrs(dsegs[0][0], dsegs[0][1], on_device=True)
    for k in range(len(dsegs)):
        if k + 1 < len(dsegs):
            st.prefetch_ptrs(dsegs[k + 1][0], dsegs[k + 1][1], on_device=True)
        st.feed_ptrs(dsegs[k][0], dsegs[k][1], on_device=True)
        for b in range(len(caps)):
            got[b].append(st.eti(b))
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    st.close()
    # ... and fed directly (one copy kernel per feed for the segments, one for the history in front of them), from addresses of every alignment
    st = dab.Stream(len(caps))
    got = [[] for _ in caps]
    odd = []
    for k, (ptrs_k, sizes_k, hbs) in enumerate(segs):
        bufs = [dab.DeviceBuffer(max(n, 16) + 32) for n in sizes_k]
        shift = [(3 * b + 5 * k + 1) % 17 for b in range(len(bufs))]
        for buf, hb, n, sh in zip(bufs, hbs, sizes_k, shift):
            if n:
                tmp = np.zeros(n + 32, np.uint8)
                tmp[sh:sh + n] = hb.array[:n]
                buf.upload(tmp)
        odd.append(bufs)
        st.feed_ptrs([b.ptr + sh for b, sh in zip(bufs, shift)], sizes_k, on_device=True)
        for b in range(len(caps)):
            got[b].append(st.eti(b))
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    st.close()
    for bufs in odd:
        for buf in bufs:
            buf.free()
    for _, _, bufs in dsegs:
        for buf in bufs:
            buf.free()
    # misuse: feeding something else than the segment handed over first; three segments waiting
    st = dab.Stream(len(caps))
    st.prefetch_ptrs(*segs[0][:2])
    with pytest.raises(dab.DabhipError, match="prefetched first"):
        st.feed_ptrs(*segs[1][:2])
    st.prefetch_ptrs(*segs[1][:2])
    with pytest.raises(dab.DabhipError, match="already waiting"):
        st.prefetch_ptrs(*segs[2][:2])
    st.close()
    for _, _, hbs in segs:
        for hb in hbs:
            hb.free()


def test_eti_fetch_is_the_drain_order_and_overlaps_the_next_segment():
    """dabhip_engine_eti_fetch / dabhip_stream_eti_fetch (the output leg of dab2eti.c:132-135 without a stall): one asynchronous download of ALL
    frames, stream by stream in emission order = the bytes eti_read returns per stream, also when the next segment is fed before the wait."""
    cfgs = [dab.synth_preset(1, seed=610 + i, cif_count0=33 * i, skip_samples=(0, 91000, 0)[i]) for i in range(3)]
    streams = [dab.synth_generate(c, 24) for c in cfgs]
    eng = dab.Engine(0)
    total = eng.decode(streams)
    hb = dab.HostBuffer(total * 6144)
    assert eng.eti_fetch(hb.ptr, total) == total
    eng.eti_fetch_wait()
    got = np.array(hb.array[:total * 6144]).reshape(total, 6144)
    want = np.concatenate([eng.eti(b) for b in range(3)])
    assert np.array_equal(got, want)
    # two fetches outstanding (fetch k + 1 issued before fetch k has been waited for: the CLI's decode thread and writer thread), each wait takes the
    # oldest; a third without a wait is refused, and the ETI buffer growing between the two (a larger decode) does not disturb the first
    hb2 = dab.HostBuffer(4 * total * 6144)
    assert eng.eti_fetch(hb.ptr, total) == total
    big = [dab.synth_generate(c, 30) for c in cfgs] + streams
    total2 = eng.decode(big)
    assert total2 > total and eng.eti_fetch(hb2.ptr, total2) == total2
    with pytest.raises(dab.DabhipError):
        eng.eti_fetch(hb.ptr, 1)
    eng.eti_fetch_wait()
    assert np.array_equal(np.array(hb.array[:total * 6144]).reshape(total, 6144), want)
    eng.eti_fetch_wait()
    assert np.array_equal(np.array(hb2.array[:total2 * 6144]).reshape(total2, 6144), np.concatenate([eng.eti(b) for b in range(6)]))
    eng.eti_fetch_wait()                              # nothing outstanding: returns at once
    hb2.free()
    eng.close()
    # session: the download of segment k is waited for only AFTER segment k + 1 has been fed (it overlaps that segment's upload and decode;
    # only the K4 of segment k + 1, which rewrites the ETI buffer, is ordered behind it)
    st = dab.Stream(3, device=0)
    cuts = [0, 7 * dab.TF_BYTES, 15 * dab.TF_BYTES + 1234, 20 * dab.TF_BYTES, 24 * dab.TF_BYTES]
    bufs = [dab.HostBuffer(64 * 3 * 6144) for _ in range(2)]
    pending = None                                   # (buffer, frames, expected bytes) of the segment before
    nframes = 0
    for k, (a, z) in enumerate(zip(cuts, cuts[1:])):
        n = st.feed([s[a:min(z, s.size)] for s in streams])
        if pending is not None:
            st.eti_fetch_wait()
            assert np.array_equal(np.array(bufs[pending[0]].array[:pending[1] * 6144]).reshape(-1, 6144), pending[2])
        expect = np.concatenate([st.eti(b) for b in range(3)]) if n else np.zeros((0, 6144), np.uint8)
        assert expect.shape[0] == n
        assert st.eti_fetch(bufs[k & 1].ptr, n) == n
        pending = (k & 1, n, expect)
        nframes += n
    st.eti_fetch_wait()
    assert np.array_equal(np.array(bufs[pending[0]].array[:pending[1] * 6144]).reshape(-1, 6144), pending[2])
    assert nframes == total
    st.close()
    for b in bufs + [hb]:
        b.free()


def test_resident_session_reads_the_streams_in_place():
    """dabhip_stream_feed_resident: the captures live in device memory and grow; nothing is copied.  Frames of all feeds == one decode; what lies
    below dabhip_stream_need_from is really not read any more (it is overwritten between the feeds); a session cannot mix the two ways of feeding."""
    caps = _caps()
    eng = dab.Engine(0)
    eng.decode(caps)
    want = [eng.eti(b) for b in range(len(caps))]
    eng.close()
    bufs = [dab.DeviceBuffer(max(c.size, 16) + 64) for c in caps]
    shift = [(5 * b + 3) % 16 for b in range(len(caps))]                  # base addresses of every alignment
    for buf, c, sh in zip(bufs, caps, shift):
        if c.size:
            tmp = np.zeros(c.size + sh, np.uint8)
            tmp[sh:] = c
            buf.upload(tmp)
    base = [buf.ptr + sh for buf, sh in zip(bufs, shift)]
    st = dab.Stream(len(caps))
    rng = np.random.default_rng(99)
    avail = [0] * len(caps)
    got = [[] for _ in caps]
    prev_need = [0] * len(caps)
    feeds = 0
    while any(a < c.size for a, c in zip(avail, caps)):
        avail = [min(c.size, a + int(rng.integers(1, 3_000_000))) for a, c in zip(avail, caps)]
        st.feed_resident(base, avail)
        feeds += 1
        for b in range(len(caps)):
            got[b].append(st.eti(b))
            need = st.need_from(b)
            assert prev_need[b] <= need <= avail[b], (b, prev_need[b], need, avail[b])
            prev_need[b] = need
            # the caller may recycle what lies below need_from: scribble over it
            if need > 4096:
                junk = np.full(need - (need % 2), 0xA5, np.uint8)
                assert dab.lib().dabhip_device_copy(base[b], junk.ctypes.data, junk.size, 1) == 0
    assert feeds > 5
    for b, w in enumerate(want):
        assert np.array_equal(np.concatenate(got[b]), w), b
    with pytest.raises(dab.DabhipError, match="feed_resident"):
        st.feed([c[:0] for c in caps])
    st.close()
    st = dab.Stream(1)
    with pytest.raises(dab.DabhipError, match="not device memory"):
        st.feed_resident([caps[0].ctypes.data], [caps[0].size])           # a host address
    st.feed([caps[0][:500000]])
    with pytest.raises(dab.DabhipError, match="windows"):
        st.feed_resident([base[0]], [600000])
    st.close()
    for buf in bufs:
        buf.free()
