"""K1's look-ahead schedule (round 5; k_sync.hip: sync_ahead_kernel, include/dabhip.h: dabhip_engine_set_sync_speculation).

The chain of sdr_demod calls (input_sdr.c:36-84) is sequential because call n's time shifts position the read of call n + 1.  What a call computes from
its frame, though, depends only on WHERE in the stream its read began, so a pass over all remaining calls x all start positions near the predicted one
can compute it ahead of the chain, which then only looks up.  The schedule must give the chain's results call for call -- status, both time shifts,
coarse frequency shift, FIFO count, fine frequency shift, and every ETI byte -- on any input: where the prediction holds (a locked receiver on a steady
signal: the reference's limit cycle of +20, -8, -8, -2 bytes), where it holds for a while (a drop-out, a late start), and where it never does (a
drifting sample clock walks every read out of the window; noise).  Mode 0 (the plain chain) is what rounds 1-4 shipped and what the sweeps against the
reference and the oracle have held; here mode 1 is held against it and against the oracle directly.
"""
import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _captures():
    rng = np.random.default_rng(20261003)
    clean = dab.synth_generate(dab.synth_preset(1, seed=901, cif_count0=1200), 44)
    caps = {
        "clean, aligned": clean,
        "clean, starts mid-frame": dab.synth_generate(dab.synth_preset(0, seed=902, cif_count0=77, skip_samples=123457), 40),
        "7 dB": dab.synth_generate(dab.synth_preset(1, seed=903, cif_count0=3, snr_db=7.0), 36),
        "5 dB, weak": dab.synth_generate(dab.synth_preset(1, seed=904, cif_count0=9, snr_db=5.0, amplitude=0.3), 30),
        "ragged end": clean[: 30 * dab.TF_BYTES + 77777],
        "drop-out and return": np.concatenate([clean[: 20 * dab.TF_BYTES + 3331], np.full(3 * dab.TF_BYTES + 17, 128, np.uint8), clean[20 * dab.TF_BYTES + 3331:]]),
        "noise, then signal": np.concatenate([rng.integers(96, 160, 2 * dab.TF_BYTES + 1001, dtype=np.uint8), clean[: 30 * dab.TF_BYTES]]),
        "short (12 calls)": clean[: 12 * 262144],
    }
    for name, ppm, seed in (("sample clock +60 ppm", 60.0, 905), ("sample clock -100 ppm", -100.0, 906), ("sample clock +8 ppm", 8.0, 907)):
        cfg = dab.synth_preset(1, seed=seed, cif_count0=500)
        cfg.channel.sro_ppm = ppm
        caps[name] = dab.synth_generate(cfg, 34)
    cfg = dab.synth_preset(1, seed=908, cif_count0=40, snr_db=14.0)
    cfg.channel.echo_delay[0], cfg.channel.echo_gain[0], cfg.channel.echo_phase[0], cfg.channel.echo_doppler_hz[0] = 600, 1.2, 0.1, 3.0
    caps["echo beyond the prefix, stronger than the direct path"] = dab.synth_generate(cfg, 30)
    return caps


def _decode(eng, caps, mode):
    eng.set_sync_speculation(mode)
    total = eng.decode(list(caps.values()))
    out = []
    for b, iq in enumerate(caps.values()):
        n = iq.size // 262144                              # the stream's own calls (a batch's trace is as long as its longest stream's)
        ints, ffs = eng.trace(b, n)
        out.append((ints[:n].copy(), ffs[:n].copy(), eng.eti(b).copy()))
    return total, out, eng.stage_ms()


def _same(a, b, names, what):
    for name, (ia, fa, ea), (ib, fb, eb) in zip(names, a, b):
        assert ia.shape == ib.shape, (what, name)
        bad = np.nonzero((ia != ib).any(axis=1))[0]
        assert bad.size == 0, "%s, %r: per-call trace differs first at call %d: %s vs %s" % (what, name, bad[0], ia[bad[0]], ib[bad[0]])
        assert np.array_equal(fa, fb, equal_nan=True), "%s, %r: fine frequency shift differs" % (what, name)
        assert ea.shape == eb.shape and np.array_equal(ea, eb), "%s, %r: ETI differs" % (what, name)


def test_look_ahead_schedule_gives_the_chains_results_call_for_call():
    caps = _captures()
    names = list(caps)
    eng = dab.Engine(0)
    total0, plain, st0 = _decode(eng, caps, 0)
    total1, ahead, st1 = _decode(eng, caps, 1)
    assert st0["sync_spec_calls"] == 0 and st1["sync_spec_calls"] > 0
    assert total0 == total1 and total0 > 500
    _same(plain, ahead, names, "look-ahead against the plain chain")
    # one engine per capture: a batch of one is what the default mode (-1) applies the schedule to
    hits = {}
    for b, name in enumerate(names):
        one = dab.Engine(0)
        n = one.decode([caps[name]])
        ncalls = caps[name].size // 262144
        ints, ffs = one.trace(0, ncalls)
        _same([plain[b]], [(ints[:ncalls], ffs[:ncalls], one.eti(0))], [name], "default mode, one capture")
        hits[name] = (one.stage_ms()["sync_spec_calls"], int((ints[:, 0] == 1).sum()))
        one.close()
    # where the prediction holds the table serves (nearly) every demodulated call behind the lock-in; a drifting sample clock leaves the window
    assert hits["clean, aligned"][0] >= hits["clean, aligned"][1] - 3, hits
    assert hits["7 dB"][0] >= hits["7 dB"][1] - 6, hits
    assert hits["short (12 calls)"][0] == 0, hits                                   # fewer than 16 calls: the plain chain
    assert hits["sample clock -100 ppm"][0] < hits["sample clock -100 ppm"][1] // 2, hits
    # ... and against the CPU oracle directly (status, time shifts, coarse frequency shift, FIFO count; ETI)
    for b, name in enumerate(names):
        if name not in ("clean, aligned", "drop-out and return", "sample clock +60 ppm", "7 dB"):
            continue
        eti, trace = ol.or_replay(caps[name])
        rows = [(t.ok, t.coarse_timeshift, t.fine_timeshift, t.coarse_freq_shift, t.fifo_count) for t in trace]
        got = [(int(r[0]), int(r[2]), int(r[3]), int(r[4]), int(r[5])) for r in ahead[b][0]]
        assert got == rows, "%r: trace differs from the oracle's" % name
        assert np.array_equal(ahead[b][2], eti), "%r: ETI differs from the oracle's" % name
    eng.close()


def test_look_ahead_schedule_in_a_session_cut_into_odd_segments():
    """Further segments of a session start from a locked state (no lock-in chain before the pass) and may hold any number of calls."""
    caps = _captures()
    pick = ["clean, aligned", "drop-out and return", "sample clock +8 ppm", "7 dB"]
    iqs = [caps[k] for k in pick]
    eng = dab.Engine(0)
    eng.set_sync_speculation(0)
    eng.decode(iqs)
    want = [eng.eti(b).copy() for b in range(len(iqs))]
    eng.close()
    for mode, cuts in ((1, (5 * 262144 + 1234, 23 * 262144, 24 * 262144 + 2, 41 * 262144 + 99999)), (-1, (19 * 262144 + 7,))):
        ses = dab.Stream(len(iqs))
        ses.set_sync_speculation(mode)
        got = [[] for _ in iqs]
        edges = (0,) + cuts + (1 << 40,)
        used = False
        for a, b in zip(edges[:-1], edges[1:]):
            ses.feed([iq[a:b] for iq in iqs])
            used = used or ses.stage_ms()["sync_spec_calls"] > 0
            for s in range(len(iqs)):
                got[s].append(ses.eti(s))
        assert used, "mode %d: the schedule never ran" % mode
        for s, name in enumerate(pick):
            eti = np.concatenate(got[s]) if got[s] else np.zeros((0, 6144), np.uint8)
            assert eti.shape == want[s].shape and np.array_equal(eti, want[s]), "mode %d, %r: session ETI differs from the one-shot decode's" % (mode, name)
        ses.close()
