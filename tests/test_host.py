"""Host-side product code without a GPU: the C-ABI library loads and exports every symbol
include/dabhip.h declares, the host control plane matches the reference's golden vectors,
the synthetic modulator is deterministic, and the decode entry points fail loudly (no CPU
fallback exists)."""
import ctypes as C
import hashlib
import os
import re
import sys

import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "dabhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dabhip_[a-z0-9_]+)\s*\(", txt)) - {"dabhip_eti_callback", "dabhip_eti_sink"})


def test_library_exports_every_declared_symbol():
    L = C.CDLL(dab.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "libdabhip.so lacks %s" % n
    # the Python binding declares a signature for each of them too
    assert set(names) == set(dab.exported_symbols())


def test_no_cpu_fallback():
    if dab.lib().dabhip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(dab.DabhipError, match="no HIP device"):
        dab.Engine(0)
    with pytest.raises(dab.DabhipError):
        dab.viterbi(np.full(4 * 774, 128, np.uint8), 768)
    with pytest.raises(dab.DabhipError):
        dab.Sdr(0)
    with pytest.raises(dab.DabhipError):
        dab.Dab(0)
    with pytest.raises(dab.DabhipError, match="no HIP device"):
        dab.Multi([0, 0])
    with pytest.raises(dab.DabhipError):
        dab.Stream(2)


def test_product_does_not_link_the_oracle():
    out = os.popen("ldd %s" % dab.LIB_PATH).read()
    assert "oracle" not in out and "dabref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "dabtools_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f), errors="ignore").read()
                assert "liboracle" not in src and "oracle_lib" not in src and "or_replay" not in src, f


def test_host_fib_parse_and_header_match_reference_golden():
    kat = np.load(os.path.join(G, "backend_kat.npz"))
    hdr, sub = dab.host_parse_fibs(kat["fibdec_fibs"], kat["fibdec_ok"])
    assert list(hdr) == list(kat["fibdec_hdr"])
    for i in range(64):
        want = kat["fibdec_sub"][i]
        assert sub[i][0] == want[0]
        if want[0] >= 0:
            assert [sub[i][k] for k in (0, 1, 3, 4, 5, 6)] == [want[k] for k in (0, 1, 3, 4, 5, 6)], i
        assert sub[i][7] == want[7]
    for tag in ("a", "b"):
        hi, lo = kat["etihdr_%s_cif" % tag]
        rows = np.full((64, 8), -1, np.int32)
        for i in range(64):
            r = kat["etihdr_sub5"][i]
            rows[i] = [r[0], r[1], 0, r[2], 0, r[3], r[4], -1]
        got = dab.host_eti_header([0xC181, hi, lo], rows)
        assert np.array_equal(got, kat["etihdr_%s" % tag])


def test_host_control_plane_vs_reference_eti_sequence():
    """Lock FSM, ring and header sequence vs the ETI frames the real reference emitted for the
    golden back-end run (includes a lock loss).  FIBs come from the oracle's FIC decoder."""
    g = np.load(os.path.join(G, "backend_e2e.npz"))
    O = ol.oracle()
    fibs, oks = [], []
    for row in g["tf_bits"]:
        bits = np.ascontiguousarray(np.unpackbits(row)[:9216])
        f = np.zeros((12, 32), np.uint8)
        o = np.zeros(12, np.uint8)
        O.or_fic_decode(ol._ptr(bits), ol._ptr(f), ol._ptr(o))
        fibs.append(f.reshape(-1))
        oks.append(o)
    first, headers = dab.host_control_replay(np.array(fibs), np.array(oks))
    eti = g["eti"]
    assert len(headers) == len(eti)
    for i, h in enumerate(headers):
        assert np.array_equal(h, eti[i][: h.size]), i
        # the FIBs that follow the header are those of the oldest CIF in the ring
        tf, q = divmod(int(first[i]), 4)
        assert np.array_equal(eti[i][h.size:h.size + 96], np.array(fibs[tf]).reshape(12, 32)[3 * q:3 * q + 3].reshape(-1)), i


def test_operator_messages_match_the_reference_stderr():
    """What dab_process_frame prints for its operator -- 'Locked' (dab.c:51), 'Lock lost, resetting ringbuffer' (dab.c:57), the one-time ensemble dump
    (dab.c:78-82 -> misc.c:316-328) -- comes out of the host control plane as the same text in the same order: against the REAL reference's stderr on the
    golden back-end run (tests/golden/backend_e2e_stderr.txt, made by make_operator_log.py: lock, dump, a lock loss, a re-lock without a second dump)."""
    g = np.load(os.path.join(G, "backend_e2e.npz"))
    want = open(os.path.join(G, "backend_e2e_stderr.txt")).read()
    assert want.count("Locked\n") == 2 and "Lock lost, resetting ringbuffer\n" in want and want.count("ENSEMBLE_INFO") == 1
    O = ol.oracle()
    fibs, oks = [], []
    for row in g["tf_bits"]:
        bits = np.ascontiguousarray(np.unpackbits(row)[:9216])
        f = np.zeros((12, 32), np.uint8)
        o = np.zeros(12, np.uint8)
        O.or_fic_decode(ol._ptr(bits), ol._ptr(f), ol._ptr(o))
        fibs.append(f.reshape(-1))
        oks.append(o)
    dab.host_control_replay(np.array(fibs), np.array(oks))
    assert dab.host_control_replay_log() == want
    if os.path.isdir("/root/reference/src"):       # in the build container: the reference itself, live
        sys.path.insert(0, G)
        import make_operator_log
        live, _ = make_operator_log.reference_stderr(g["tf_bits"])
        assert live == want


def test_synth_is_deterministic_and_wellformed():
    g = np.load(os.path.join(G, "backend_e2e.npz"))
    preset, seed, cif0, ntf = [int(x) for x in g["synth"]]
    cfg = dab.synth_preset(preset, seed=seed, cif_count0=cif0, snr_db=float(g["snr_db"]))
    iq = dab.synth_generate(cfg, 4)
    assert iq.size == 4 * dab.TF_BYTES and iq.min() >= 1 and iq.max() <= 254
    assert np.array_equal(iq, dab.synth_generate(cfg, 4))
    # null symbol carries only noise, the signal rails ~28 LSB rms
    clean = dab.synth_generate(dab.synth_preset(preset, seed=seed), 1)
    assert (clean[: 2 * 2656] == 127).all()
    assert 24 < np.std(clean[2 * 2656:].astype(float)) < 32
    # every FIB the modulator emits carries a valid CRC and the configured ensemble id
    for c in (0, 1, 249, 250, 4999, 5000):
        f = dab.synth_fibs(dab.synth_preset(0, cif_count0=0), c).reshape(3, 32)
        for fib in f:
            assert ol.oracle().or_check_fib_crc(ol._ptr(np.ascontiguousarray(fib))) == 1
        assert f[0][2] == 0xC1 and f[0][3] == 0x81
        assert f[0][4] == (c % 5000) // 250 and f[0][5] == (c % 5000) % 250
    full = dab.synth_generate(cfg, ntf) if os.environ.get("DABHIP_SLOW") else None
    if full is not None:
        assert hashlib.sha256(full.tobytes()).digest() == g["iq_sha256"].tobytes()


def test_synth_rejects_bad_configs():
    cfg = dab.synth_preset(1)
    cfg.sub[1].start_cu = 10                      # overlaps sub-channel 0
    with pytest.raises(dab.DabhipError, match="overlap"):
        dab.synth_generate(cfg, 1)
    cfg = dab.synth_preset(1)
    cfg.sub[0].start_cu = 800                     # 96 CU do not fit
    with pytest.raises(dab.DabhipError):
        dab.synth_generate(cfg, 1)
    with pytest.raises(dab.DabhipError):
        dab.synth_preset(9)


def test_oracle_replay_roundtrip_on_cpu():
    """modulator -> oracle receiver returns the payload (keeps the CPU checker honest without a GPU)."""
    cfg = dab.synth_preset(1, seed=8, cif_count0=77, skip_samples=31337)
    iq = dab.synth_generate(cfg, 22)
    eti, trace = ol.or_replay(iq)
    assert len(eti) >= 4
    assert any(t.coarse_timeshift != 0 for t in trace)        # the unaligned start needed a coarse resync
    e = eti[0].astype(int)
    nst = e[5] & 0x7f
    pos = 12 + 4 * nst
    fibs = eti[0][pos:pos + 96]
    cif = next(c for c in range(80) if np.array_equal(dab.synth_fibs(cfg, c), fibs))
    pos += 96
    for k in range(nst):
        p = dab.synth_payload(cfg, cif, k)
        assert np.array_equal(eti[0][pos:pos + p.size], p)
        pos += p.size


def test_eti2mpa_extracts_subchannel_from_reference_eti(tmp_path):
    """The eti2mpa counterpart (eti2mpa.c:16-68) on the ETI frames the real reference produced (golden fixture)."""
    import subprocess
    import eti_check
    exe = os.path.join(os.path.dirname(dab.LIB_PATH), "eti2mpa")
    assert os.path.exists(exe), "eti2mpa not built"
    g = np.load(os.path.join(G, "backend_e2e.npz"))
    eti = g["eti"]
    for scid in (1, 5, 9):
        out = subprocess.run([exe, str(scid)], input=eti.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        want = b"".join(bytes(p["subch"][[s[0] for s in p["stc"]].index(scid)]) for p in map(eti_check.parse, eti))
        assert out == want
    r = subprocess.run([exe, "33"], input=eti.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 4 and r.stdout == b""
    # ... and next to the reference's own eti2mpa.c, compiled unmodified by oracle/Makefile (it ends with "Read error", exit 1, at
    # the end of its input: eti2mpa.c:33-36)
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "eti2mpa_ref")
    if not os.path.exists(ref_exe):
        pytest.skip("oracle/_ref/eti2mpa_ref not built (reference sources absent)")
    eti_file = tmp_path / "golden.eti"            # a regular file: the reference's single read() per frame (eti2mpa.c:32) comes back short on a pipe
    eti.tofile(eti_file)
    for scid in (1, 2, 5, 9):
        ours = subprocess.run([exe, str(scid)], input=eti.tobytes(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
        with open(eti_file, "rb") as f:
            ref = subprocess.run([ref_exe, str(scid)], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert ref.returncode == 1 and len(ref.stdout) > 0 and ref.stdout == ours, scid


def _fifo_view_matches_oracle(iq):
    """Replays the K1 FIFO / frame-buffer bookkeeping (fifo_view.hpp via dabhip_host_fifo_*) next to the oracle front end,
    whose buffer is filled byte by byte like sdr_fifo.c:43-61: after every call the view must describe exactly the
    oracle's 393216-byte frame buffer."""
    O = ol.oracle()
    S = O.or_sdr_new()
    fifo = dab.HostFifo()
    fic = np.zeros(dab.FIC_BITS, np.uint8)
    msc = np.zeros(dab.MSC_BITS, np.uint8)
    tr = ol.SdrTrace()
    coarse = fine = 0
    reads = short_after_skip = 0
    for off in range(0, iq.size - dab.CHUNK_BYTES + 1, dab.CHUNK_BYTES):
        O.or_sdr_demod(S, ol._ptr(iq[off:off + dab.CHUNK_BYTES]), dab.CHUNK_BYTES, ol._ptr(fic), ol._ptr(msc))
        O.or_sdr_get_trace(S, C.byref(tr))
        status, view, count = fifo.call(coarse, fine, iq)
        assert (status > 0) == bool(tr.read_frame) and count == tr.fifo_count, off
        if status:
            reads += 1
            want = np.ctypeslib.as_array(O.or_sdr_buffer(S), (dab.TF_BYTES,))
            got = dab.HostFifo.materialise(iq, view, fifo.tail)
            assert np.array_equal(got, want), "call at byte %d: view %r" % (off, view)
            short_after_skip += int(len(view) > 1 and coarse + fine > view[0][0])
        coarse, fine = tr.coarse_timeshift, tr.fine_timeshift
    O.or_sdr_free(S)
    fifo.close()
    return reads, short_after_skip


def test_fifo_view_bookkeeping_matches_oracle_buffer():
    # aligned, mid-frame start (coarse resync at the start), negative and positive fine corrections
    for seed, skip in ((3, 0), (4, 123457)):
        reads, _ = _fifo_view_matches_oracle(dab.synth_generate(dab.synth_preset(1, seed=seed, skip_samples=skip), 12))
        assert reads >= 10


def test_fifo_view_resync_while_locked_short_read_after_large_skip():
    """Samples vanish mid-stream after lock: the coarse correction (373,220 bytes) exceeds the bytes the FIFO can still
    deliver after skipping them, so sdr_read_fifo leaves SKIPPED bytes in buffer[len .. shift) (sdr_fifo.c:49-55)."""
    iq = dab.synth_generate(dab.synth_preset(1, seed=1), 40)
    cut = 2 * (21 * 196608 + 30000)
    iq = np.concatenate([iq[:cut], iq[cut + 20000:]])
    reads, short_after_skip = _fifo_view_matches_oracle(iq)
    assert reads >= 30 and short_after_skip >= 1


def test_fifo_view_receiver_clock_fast_every_read_is_short():
    """A receiver whose sample clock runs fast against the transmitter's (+80 ppm here) gets a negative time shift on EVERY call: every read is
    short and the end of sdr->buffer is never rewritten in full (sdr_fifo.c:56-59).  As views into the stream that history nested without bound
    and pinned the stream's first bytes for ever (rounds 1-4: a decode of such a stream failed with 'more than kMaxSeg nested short reads', a live
    session kept the whole stream); the last 1536 bytes now travel as bytes, and the views stay at ONE segment."""
    cfg = dab.synth_preset(1, seed=9)
    cfg.channel.sro_ppm = 80.0
    iq = dab.synth_generate(cfg, 60)
    O = ol.oracle()
    S = O.or_sdr_new()
    fifo = dab.HostFifo()
    fic = np.zeros(dab.FIC_BITS, np.uint8)
    msc = np.zeros(dab.MSC_BITS, np.uint8)
    tr = ol.SdrTrace()
    coarse = fine = shorts = 0
    for off in range(0, iq.size - dab.CHUNK_BYTES + 1, dab.CHUNK_BYTES):
        O.or_sdr_demod(S, ol._ptr(iq[off:off + dab.CHUNK_BYTES]), dab.CHUNK_BYTES, ol._ptr(fic), ol._ptr(msc))
        O.or_sdr_get_trace(S, C.byref(tr))
        status, view, count = fifo.call(coarse, fine, iq)
        if status:
            assert np.array_equal(dab.HostFifo.materialise(iq, view, fifo.tail), np.ctypeslib.as_array(O.or_sdr_buffer(S), (dab.TF_BYTES,))), off
            shorts += int(coarse + fine < 0)
            if off > 30 * dab.CHUNK_BYTES:                            # settled: nothing older than this call's own read is referenced
                assert len(view) == 1 and view[0][1] > off - 3 * dab.TF_BYTES, view
        coarse, fine = tr.coarse_timeshift, tr.fine_timeshift
    O.or_sdr_free(S)
    fifo.close()
    assert shorts >= 50


def test_fifo_shifted_read_against_reference_fifo():
    """The same bookkeeping against the REAL sdr_fifo.c (oracle/_ref): shift sequences including positive shifts
    larger than what remains queued afterwards."""
    R = ol.ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(11)
    stream = rng.integers(1, 255, 150 * dab.CHUNK_BYTES, dtype=np.uint8)
    F = R.refh_fifo_new(C.c_uint32(196608 * 2 * 4))                 # input_sdr.c:184
    fifo = dab.HostFifo()
    buf = np.zeros(dab.TF_BYTES, np.uint8)
    shifts = [0, 0, -300, 40, 380000, -1500, 250000, 16, 389000, 389000, -2, 0, 120000, 300000, -766, 389430]
    k = nshort = 0
    for c in range(70):
        R.refh_fifo_write(F, ol._ptr(stream[c * dab.CHUNK_BYTES:(c + 1) * dab.CHUNK_BYTES]), dab.CHUNK_BYTES)
        shift = shifts[k % len(shifts)]
        if R.refh_fifo_count(F) >= 196608 * 3:                      # input_sdr.c:41-47
            R.refh_fifo_read(F, dab.TF_BYTES, shift, ol._ptr(buf))
            status, view, count = fifo.call(shift, 0, stream)
            assert status > 0 and count == R.refh_fifo_count(F)
            assert np.array_equal(dab.HostFifo.materialise(stream, view, fifo.tail), buf), (c, shift, view)
            nshort += int(shift > view[0][0])
            k += 1
        else:
            assert fifo.call(shift, 0, stream)[0] == 0
    assert k > 20 and nshort >= 2
    fifo.close()
    # the same with calls of every length (input_buffer_len is whatever the callback left, input_sdr.c:36-38): long runs of ever larger negative
    # shifts (each read shorter than the one before: the old views nested once per read), dry reads in between
    F = R.refh_fifo_new(C.c_uint32(196608 * 2 * 4))
    fifo = dab.HostFifo()
    buf[:] = 0
    shifts = [0] + [-2 * i for i in range(1, 40)] + [300000, -1536, -1534, 16, -4, -800, 389000, -6, -10, -20]
    fed = k = 0
    while fed + dab.CHUNK_BYTES <= stream.size and k < 3 * len(shifts):
        n = int(rng.choice([dab.CHUNK_BYTES, dab.CHUNK_BYTES, 131072, 65536, 2 * int(rng.integers(0, 131073))]))
        R.refh_fifo_write(F, ol._ptr(stream[fed:fed + max(n, 1)]), n)
        fed += n
        shift = shifts[k % len(shifts)]
        if R.refh_fifo_count(F) >= 196608 * 3:
            R.refh_fifo_read(F, dab.TF_BYTES, shift, ol._ptr(buf))
            status, view, count = fifo.call(shift, 0, stream, n)
            assert status > 0 and count == R.refh_fifo_count(F), (k, shift)
            assert np.array_equal(dab.HostFifo.materialise(stream, view, fifo.tail), buf), (k, shift, view)
            assert len(view) <= 3, view
            k += 1
        else:
            assert fifo.call(shift, 0, stream, n)[0] == 0
    assert k >= 60


def test_product_tables_match_reference_arrays():
    """dab_tables.hpp (generated from the ETSI rules; what the kernels and the control plane use) against the reference's
    literal arrays held in tests/golden/tables.npz (ueptable dab_tables.c:16-81, pvec :102-127, rev_freq_deint_tab :164,
    sdr_prstab.c)."""
    t = np.load(os.path.join(G, "tables.npz"))
    uep = dab.host_table(0)
    want = t["ueptable"].astype(np.int64)
    assert np.array_equal(uep[:, :7], want[:, :7])                       # bitrate, size, protection level, L1..L4
    assert np.array_equal(uep[:, 7:] - 1, want[:, 7:])                   # the reference stores PI - 1 (-1 = unused segment)
    assert np.array_equal(dab.host_table(1), t["pvec"])
    assert np.array_equal(dab.host_table(2), t["rev_freq_deint_tab"])
    assert np.array_equal(dab.host_table(3), t["prs_quarter_turns"])


def test_integration_shims_are_built_against_the_reference_callers():
    """integration/viterbi_hip.c (S1) and integration/dab_hip.c (S3) compile and link against the reference's own sources /
    headers (oracle/Makefile) and define exactly the symbols the reference's callers bind: viterbi.h:6-8, viterbi_spiral.h:22,
    dab.h:91-92.  (Running them needs a GPU: tests/test_gpu_parity_r2.py.)"""
    import subprocess
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("needs /root/reference at build time")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = {"libdabref_hipS1.so": ["viterbi", "init_viterbi", "create_viterbi", "fic_decode", "dab_process_frame"],
            "libdabref_hipS3.so": ["init_dab_state", "dab_process_frame"]}
    for so, syms in want.items():
        path = os.path.join(ROOT, "oracle", "_ref", so)
        assert os.path.exists(path), so
        defined = {l.split()[-1] for l in os.popen("nm -D --defined-only %s" % path).read().splitlines() if l.strip()}
        undefined = {l.split()[-1] for l in os.popen("nm -D --undefined-only %s" % path).read().splitlines() if l.strip()}
        for sym in syms:
            assert sym in defined, (so, sym)
        assert any(u.startswith("dabhip_") for u in undefined), so          # ... and they are bound to libdabhip, not to CPU code
    # every dabhip_* entry point the shims call is declared in the public header
    declared = set(_declared_functions())
    for f in ("viterbi_hip.c", "dab_hip.c", "input_sdr_hip.c"):
        src = open(os.path.join(ROOT, "integration", f)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for name in set(re.findall(r"\b(dabhip_[a-z0-9_]+)\s*\(", src)):
            assert name in declared, (f, name)


def test_control_plane_silences_a_multiplex_the_reference_cannot_assemble():
    """Fault isolation on the host (no GPU): CRC-valid FIBs that signal a sub-channel past the CIF / a frame past 6144 bytes / a non-standard EEP
    option -- where misc.c:233,246-296, depuncture.c:84-132 and fic.c:84 run off their arrays -- make the control plane stop emitting frames for
    that ensemble (ring and counter move on) instead of handing the decoders an impossible plan."""
    uep = lambda sid, cu, idx: [(sid << 2) | (cu >> 8), cu & 0xff, idx & 0x3f]
    eep = lambda sid, cu, lev, size: [(sid << 2) | (cu >> 8), cu & 0xff, 0x80 | ((lev >> 2) << 4) | ((lev & 3) << 2) | (size >> 8), size & 0xff]
    fig01 = lambda e: bytes([len(e) + 1, 0x01]) + bytes(e)
    cases = {"clean": None, "outside": fig01(uep(40, 1000, 63)), "overflow": fig01(sum((uep(50 + k, 0, 63) for k in range(9)), [])),
             "option": fig01(eep(41, 400, 9, 24)), "size0": fig01(eep(42, 300, 5, 13)), "overlap_but_fine": fig01(uep(33, 0, 35))}
    counts = {}
    for name, patch in cases.items():
        cfg = dab.synth_preset(1, seed=5, cif_count0=100)
        if patch:
            cfg.set_fib_patch(patch, from_cif=4 * 16)
        fibs = np.array([[dab.synth_fibs(cfg, 4 * t + q) for q in range(4)] for t in range(24)], dtype=np.uint8).reshape(24, 384)
        first, hdrs = dab.host_control_replay(fibs, np.ones((24, 12), np.uint8))
        counts[name] = len(first)
    assert counts["clean"] == 4 * (24 - 13) and counts["overlap_but_fine"] == counts["clean"]
    for name in ("outside", "overflow", "option", "size0"):
        assert counts[name] == 4 * (16 - 13), (name, counts)      # the frames before the poisoned FIBs arrived, none after


def test_host_pools_are_sized_from_the_cpus_the_container_grants():
    """The library's budget (placement.hpp: affinity mask capped by the cgroup's CFS quota) equals the launcher's pure-Python one (shard.cpu_budget),
    and neither is the machine's thread count when the container grants less."""
    from dabtools_amd import shard
    n, affinity, quota = dab.host_cpu_budget()
    pa, pq = shard.cpu_budget()
    assert (affinity, quota) == (pa, pq or 0) and n == shard.usable_cpus() >= 1
    assert n <= affinity <= (os.cpu_count() or affinity)
    if quota:
        assert n <= quota


def test_host_placement_plan_masks_are_numa_local_and_disjoint():
    """VERDICT r3 item 7 (no GPU, no sysfs needed): 8 slices on a two-socket node -- GPUs 0..3 on node 0, 4..7 on node 1, the kernel's usual
    interleaved SMT numbering -- get disjoint, equally sized CPU chunks of their own node; a slice on an unknown node stays unbound."""
    lists = ["0-63,128-191", "64-127,192-255"]
    n, cpu_slice = dab.host_placement_plan([0, 0, 0, 0, 1, 1, 1, 1], lists, 256)
    assert n == 8
    node_of = lambda c: 0 if (c % 128) < 64 else 1
    per = {}
    for c, sl in enumerate(cpu_slice):
        assert sl >= 0, "every CPU of a node with slices is handed out"
        assert node_of(c) == (0 if sl < 4 else 1), "CPU %d (node %d) given to slice %d" % (c, node_of(c), sl)
        per.setdefault(sl, []).append(c)
    assert sorted(per) == list(range(8)) and all(len(v) == 32 for v in per.values())          # disjoint by construction of the map, balanced
    # uneven: three slices on one node, one unknown, a node without slices keeps its CPUs
    n, cpu_slice = dab.host_placement_plan([1, -1, 1, 1], ["0-3", "4-13"], 16)
    assert n == 3 and cpu_slice[:4] == [-1] * 4 and cpu_slice[14:] == [-1, -1]
    assert [cpu_slice[4:14].count(s) for s in (0, 2, 3)] == [3, 3, 4] and 1 not in cpu_slice
    # more slices than CPUs on a node: nobody goes unbound
    n, _ = dab.host_placement_plan([0, 0, 0], ["5-6"], 8)
    assert n == 3


def test_look_ahead_prediction_of_the_read_pointer_equals_the_fifo_call_by_call():
    """K1's look-ahead pass (k_sync.hip: sync_ahead_kernel) predicts where a stream's reads will begin by advancing the FIFO's counters over n calls in
    closed form (fifo_view.hpp: fifo_skip_unshifted -- whole periods of three calls in one step).  Held here against n single calls of the FIFO itself
    (dabhip_host_fifo_call, which the suite holds against the reference's sdr_fifo.c), from every phase of the period and after shifted reads, for
    n from 0 to a stream of hours."""
    import dabtools_amd as dab

    def walk(prefix):                                  # a FIFO taken through `prefix` = [(coarse, fine), ...] calls -> (fifo, fed, consumed)
        f = dab.HostFifo()
        fed = consumed = 0
        for cts, fts in prefix:
            status, view, count = f.call(cts, fts)
            fed += 262144
            consumed = fed - count
        f.call  # (the last call's shifts stay pending in the state: the prefixes below end with (0, 0))
        return f, fed, consumed

    prefixes = [[(0, 0)] * k for k in range(4, 10)]                                              # every phase of the three-call period
    prefixes += [[(0, 0)] * 4 + [(293220, 0)] + [(0, -2)] + [(0, 0)] * k for k in range(1, 5)]    # behind a coarse correction and a negative fine shift
    prefixes += [[(0, 0)] * 5 + [(0, 1236)] + [(0, -234)] + [(0, 0)] * k for k in range(2, 6)]    # behind a skipping and a short read
    checked = 0
    for prefix in prefixes:
        for n in (0, 1, 2, 3, 7, 8, 9, 10, 11, 12, 13, 59, 60, 61, 1000, 16785, 100003):
            a, fed, consumed = walk(prefix)
            want_fed, want_consumed = fed, consumed
            b, _, _ = walk(prefix)
            for _ in range(min(n, 70)):                # call by call (the closed form is linear in whole periods beyond that: checked by arithmetic below)
                status, view, count = b.call(0, 0)
                want_fed += 262144
                want_consumed = want_fed - count
            got_fed, got_consumed = a.skip_unshifted(min(n, 70))
            assert (got_fed, got_consumed) == (want_fed, want_consumed), (prefix, n)
            if n > 70:                                 # n calls = 70 + whole periods + a remainder: the same counters as stepping the remainder after the periods
                c, _, _ = walk(prefix)
                big_fed, big_consumed = c.skip_unshifted(n)
                periods = (n - 70) // 3
                d, _, _ = walk(prefix)
                d.skip_unshifted(70)
                step_fed, step_consumed = d.skip_unshifted((n - 70) % 3)
                assert (big_fed, big_consumed) == (step_fed + periods * 3 * 262144, step_consumed + periods * 2 * 393216), (prefix, n)
            checked += 1
            for h in (a, b):
                h.close()
    assert checked == len(prefixes) * 17
    # refused while a shift is pending or before the first frame has been dropped
    f = dab.HostFifo()
    f.call(0, 0)
    with pytest.raises(dab.DabhipError):
        f.skip_unshifted(5)
    f.close()
    g, _, _ = walk([(0, 0)] * 5 + [(0, 16)])
    with pytest.raises(dab.DabhipError):
        g.skip_unshifted(5)
    g.close()


def test_documents_name_profile_files_that_exist_and_hold_no_placeholders():
    """README / DESIGN / COVERAGE / INTEGRATION / profiles/README quote measurements by file: every profiles/rNN_* they name is in the tree, and no
    unfilled placeholder is left in them."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for doc in ("README.md", "DESIGN.md", "COVERAGE.md", "INTEGRATION.md", os.path.join("profiles", "README.md")):
        text = open(os.path.join(root, doc)).read()
        names = set(re.findall(r"profiles/(r0\d_[A-Za-z0-9_.\-]+\.(?:json|txt|csv))", text))
        if doc.endswith(os.path.join("profiles", "README.md")):
            names |= set(re.findall(r"`(r0\d_[A-Za-z0-9_.\-]+\.(?:json|txt|csv))`", text))
        missing += [(doc, n) for n in sorted(names) if not os.path.exists(os.path.join(root, "profiles", n))]
        assert not re.search(r"\bR\d_[A-Z]+\b|TODO|TBD|XXX", text), doc
    assert not missing, missing


def test_two_lane_decoder_index_algebra_model():
    """tools/models/twolane_model.py: the rotating lane-bit schedule of vit_two_lanes.hpp (register maps, per-lane metric tables, re-pairing, record byte
    positions) equals a plain 64-state add-compare-select step by step; the header's constexpr helpers are this model's functions."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("twolane_model", os.path.join(ROOT, "tools", "models", "twolane_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.check_layouts()
    old = sys.argv
    sys.argv = ["twolane_model.py", "192"]
    try:
        m.main()
    finally:
        sys.argv = old
    # the header states the same schedule
    text = open(os.path.join(ROOT, "dabtools_amd", "csrc", "vit_two_lanes.hpp")).read()
    assert "return (3 + t) % 6" in text and "side * 8 + remove_bit(r, L < tau ? L : L - 1)" in text
    # ... and 2^NL lanes without per-lane tables (vit_four_lanes.hpp): tools/models/multilane_model.py
    spec = importlib.util.spec_from_file_location("multilane_model", os.path.join(ROOT, "tools", "models", "multilane_model.py"))
    mm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mm)
    for starts in ((3,), (3, 5)):
        assert mm.run(starts, 192, 5)
    text = open(os.path.join(ROOT, "dabtools_amd", "csrc", "vit_four_lanes.hpp")).read()
    assert "return ((i == 0 ? 3 : 5) + t) % 6" in text
