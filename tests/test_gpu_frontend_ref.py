"""The front end against the REFERENCE'S OWN front end (round 4).

Until now the front-end rows of SURVEY.md 8(a) -- sdr_demod, the four estimators, the FIFO, the demapper (input_sdr.c:27-165, sdr_sync.c:34-302,
sdr_fifo.c:26-61) -- were green only against the restatement oracle/or_frontend.c: the reference's files include <fftw3.h>, libfftw3 is not in the
image.  The image's ROCm installation, however, ships AMD's own implementation of the FFTW3 API (hipfft/hipfftw.h, libhipfftw.so, double precision, a
front for rocFFT); oracle/Makefile compiles the reference's front-end sources UNMODIFIED against it (oracle/_ref/libdabref_frontend.so,
oracle/ref_frontend_harness.c).  That is the reference's own object code for everything but the DFT behind fftw_execute -- which is a third party's
library there as here, and whose results any correct fp64 DFT matches to ~1e-13, far inside what the sign tests and arg-maxima resolve.

  * or_frontend.c (the CPU oracle the whole test-suite leans on) == the real front end, call by call: sdr_demod's return, coarse / fine time shift,
    coarse frequency shift, FIFO count, the fine frequency estimate, and all 230,400 demapped bits of every frame -- aligned, mid-frame start, noisy,
    off-tune (forced re-synchronisation) captures;
  * the GPU front end == the real front end on the same captures (per-call trace, bits of every frame);
  * real front end + real back end (oracle/_ref/libdabref.so) = the reference end to end: its ETI bytes == the batch engine's.

It needs the GPU (hipFFTW executes its plans there), so it lives in the -m gpu suite; it is skipped where oracle/_ref was built without the library.
"""
import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

pytestmark = pytest.mark.gpu

CASES = [  # (preset, seed, skip_samples, snr_db, cfo_hz, TFs)
    (1, 901, 0, 1000.0, 0.0, 19),
    (1, 902, 61000, 1000.0, 0.0, 20),
    (0, 903, 0, 9.0, 0.0, 18),
    (1, 904, 150001, 12.0, 180.0, 19),
    (1, 905, 0, 1000.0, 2300.0, 8),          # more than one carrier off tune: coarse frequency != 0, forced re-synchronisation, never a frame
    (1, 906, 0, 20.0, -1000.0, 8),           # exactly one carrier off: accepted (|k| <= 1, input_sdr.c:105-109)
]


def _captures():
    out = []
    for preset, seed, skip, snr, cfo, ntf in CASES:
        cfg = dab.synth_preset(preset, seed=seed, cif_count0=31 * seed % 5000, skip_samples=skip, snr_db=snr, cfo_hz=cfo)
        out.append(dab.synth_generate(cfg, ntf))
    return out


@pytest.fixture(scope="module")
def real():
    if ol.ref_frontend() is None or ol.ref() is None:
        pytest.skip("oracle/_ref/libdabref_frontend.so not built (no hipFFTW in this image?)")
    caps = _captures()
    return caps, [ol.ref_frontend_replay(iq) for iq in caps]


def test_oracle_front_end_equals_the_real_front_end(real):
    caps, ref = real
    O = ol.oracle()
    nframes = 0
    for iq, (_, calls, frames) in zip(caps, ref):
        S = O.or_sdr_new()
        fic, msc = np.zeros(9216, np.uint8), np.zeros(221184, np.uint8)
        tr = ol.SdrTrace()
        k = f = 0
        for off in range(0, iq.size - 262144 + 1, 262144):
            ok = O.or_sdr_demod(S, ol._ptr(iq[off:off + 262144]), 262144, ol._ptr(fic), ol._ptr(msc))
            O.or_sdr_get_trace(S, tr)
            want = calls[k]
            assert (ok, tr.coarse_timeshift, tr.fine_timeshift, tr.coarse_freq_shift, tr.fifo_count) == want[:5], (k, want)
            assert abs(tr.fine_freq_shift - want[5]) < 1e-6, (k, tr.fine_freq_shift, want[5])
            if ok:
                assert np.array_equal(fic, frames[f][0]) and np.array_equal(msc, frames[f][1]), "frame %d: demapped bits differ from the real front end" % f
                f += 1
            k += 1
        assert f == len(frames)
        nframes += f
        O.or_sdr_free(S)
    assert nframes >= 50


def test_gpu_front_end_equals_the_real_front_end(real):
    caps, ref = real
    eng = dab.Engine(0)
    eng.decode(caps)
    for b, (iq, (_, calls, frames)) in enumerate(zip(caps, ref)):
        ints, ffs = eng.trace(b, len(calls))
        assert len(ints) == len(calls)
        for k, want in enumerate(calls):
            assert (ints[k][0], ints[k][2], ints[k][3], ints[k][4], ints[k][5]) == want[:5], (b, k, list(ints[k]), want)
            assert abs(ffs[k] - want[5]) < 1e-6, (b, k)
        for t, (fic, msc) in enumerate(frames):
            gf, gm = eng.demapped_tf(b, t)
            assert np.array_equal(gf.astype(np.uint8), fic) and np.array_equal(gm.astype(np.uint8), msc), "stream %d frame %d: bits differ from the real front end" % (b, t)
    eng.close()


def test_batch_engine_equals_the_reference_end_to_end(real):
    """Real sdr_demod + real dab_process_frame = dab2eti's demod thread (dab2eti.c:60-75) on a file: the ETI bytes the reference itself emits."""
    caps, ref = real
    eng = dab.Engine(0)
    total = eng.decode(caps)
    produced = 0
    for b, (eti, _, _) in enumerate(ref):
        got = eng.eti(b)
        assert got.shape == eti.shape and np.array_equal(got, eti), "stream %d: ETI differs from the reference (real front end + real back end)" % b
        produced += eti.shape[0]
    assert produced == total and produced >= 40
    # the same through a session in odd-sized segments, and with the two-kernel OFDM stage
    eng.set_fused(False)
    assert eng.decode(caps) == total
    for b, (eti, _, _) in enumerate(ref):
        assert np.array_equal(eng.eti(b), eti)
    eng.close()


def test_seam_s2_binding_under_the_reference_harness(real):
    """integration/input_sdr_hip.c -- the binding a maintainer would link instead of input_sdr.o sdr_sync.o sdr_fifo.o -- compiled against the
    reference's own headers and driven by the same harness as the reference's real front end: call by call the same return values, shifts, FIFO
    counts, estimates and bits, and with the real back end behind it the same ETI bytes."""
    if ol.ref_frontend("hipS2") is None:
        pytest.skip("oracle/_ref/libdabref_hipS2.so not built")
    caps, ref = real
    for b in (0, 1, 3, 4):
        eti, calls, frames = ol.ref_frontend_replay(caps[b], which="hipS2")
        want_eti, want_calls, want_frames = ref[b]
        assert len(calls) == len(want_calls)
        for k, (g, w) in enumerate(zip(calls, want_calls)):
            # (the binding does not expose fifo.count -- struct sdr_state_t's FIFO is not used behind it --: everything dab2eti.c reads is compared)
            assert g[:4] == w[:4] and abs(g[5] - w[5]) < 1e-6, (b, k, g, w)
        assert len(frames) == len(want_frames)
        for (gf, gm), (wf, wm) in zip(frames, want_frames):
            assert np.array_equal(gf, wf) and np.array_equal(gm, wm)
        assert np.array_equal(eti, want_eti)


def test_randomised_captures_against_the_reference_end_to_end():
    """A fresh draw every run (the seed is printed): ensembles, CIF counters, start offsets, amplitudes, noise down to 7 dB, carrier offsets up to
    +-1.3 carriers, ragged lengths -- the batch engine's per-call traces and ETI bytes against dab2eti's own loop over the reference's real front end and
    real back end."""
    import time
    if ol.ref_frontend() is None or ol.ref() is None:
        pytest.skip("oracle/_ref/libdabref_frontend.so not built")
    from conftest import fresh_seed
    seed = fresh_seed("test_randomised_captures_against_the_reference_end_to_end") % 1000003     # printed and kept in gpurun_out/test_seeds.txt
    rng = np.random.default_rng(seed)
    caps = []
    for i in range(10):
        cfg = dab.synth_preset(int(rng.integers(0, 2)), seed=int(rng.integers(1, 1 << 30)), cif_count0=int(rng.integers(0, 5000)),
                               skip_samples=int(rng.integers(0, 196608)) if rng.integers(0, 2) else 0,
                               snr_db=float(rng.choice([1000.0, 20.0, 12.0, 9.0, 7.0])), amplitude=float(rng.choice([1.0, 0.6, 0.35])),
                               cfo_hz=float(rng.choice([0.0, 0.0, 120.0, -400.0, 900.0, -1300.0])))
        iq = dab.synth_generate(cfg, int(rng.integers(17, 23)))
        caps.append(iq[: iq.size - int(rng.integers(0, 300000))])
    eng = dab.Engine(0)
    total = eng.decode(caps)
    frames = 0
    for b, iq in enumerate(caps):
        eti, calls, _ = ol.ref_frontend_replay(iq)
        ints, ffs = eng.trace(b, len(calls))
        for k, want in enumerate(calls):
            assert (ints[k][0], ints[k][2], ints[k][3], ints[k][4], ints[k][5]) == want[:5], (seed, b, k)
            assert abs(ffs[k] - want[5]) < 1e-6, (seed, b, k)
        got = eng.eti(b)
        assert got.shape == eti.shape and np.array_equal(got, eti), (seed, b)
        frames += eti.shape[0]
    assert frames == total
    eng.close()


def _degenerate_captures():
    """What a receiver meets besides a clean ensemble: silence, a dead ADC, noise only, clipping, a signal that drops out and comes back, a DC offset,
    a spectrum-inverted (I/Q swapped) signal the receiver can never lock to -- 0/0 and atan2(0, 0) territory in the estimators (sdr_sync.c:34-302)."""
    rng = np.random.default_rng(4242)
    n = 12 * dab.TF_BYTES
    good = dab.synth_generate(dab.synth_preset(1, seed=777, cif_count0=40), 22)
    long = dab.synth_generate(dab.synth_preset(1, seed=781, cif_count0=4990), 46)            # (the CIF counter wraps at 5000 inside it)
    caps = {
        "mid_scale_silence": np.full(n, 128, np.uint8),                   # the converted samples are all +0.5 ... or 0: no energy contrast at all
        "all_zero_bytes": np.zeros(n, np.uint8),
        "all_ones_bytes": np.full(n, 255, np.uint8),
        "uniform_noise": rng.integers(0, 256, n, dtype=np.uint8),
        "clipped": dab.synth_generate(dab.synth_preset(1, seed=778, amplitude=6.0), 18),        # the modulator saturates at 0 / 255
        "clipped_mildly": dab.synth_generate(dab.synth_preset(1, seed=780, amplitude=1.7), 18),
        "very_weak": dab.synth_generate(dab.synth_preset(1, seed=779, amplitude=0.02), 18),     # a few LSBs of signal
        "drop_out_and_return": np.concatenate([long[: 24 * dab.TF_BYTES + 3331], np.full(3 * dab.TF_BYTES + 17, 128, np.uint8), long[24 * dab.TF_BYTES + 3331:]]),
        "drop_out_before_lock": np.concatenate([good[: 9 * dab.TF_BYTES + 3331], np.full(3 * dab.TF_BYTES + 17, 128, np.uint8), good[9 * dab.TF_BYTES + 3331:]]),
        "noise_then_signal": np.concatenate([rng.integers(96, 160, 2 * dab.TF_BYTES + 1001, dtype=np.uint8), good]),
        "dc_offset": np.clip(good.astype(np.int32) + 23, 0, 255).astype(np.uint8),
        "iq_swapped": good.reshape(-1, 2)[:, ::-1].reshape(-1).copy(),
    }
    return caps


def test_degenerate_inputs_against_the_reference_end_to_end():
    """Silence, noise, clipping, drop-outs: the engine's per-call trace (return value, shifts, FIFO count, fine-frequency estimate) and ETI bytes
    against the reference's real front end + real back end; the CPU oracle is held to the same answers, so that it can be trusted on such inputs too."""
    if ol.ref_frontend() is None or ol.ref() is None:
        pytest.skip("oracle/_ref/libdabref_frontend.so not built")
    caps = _degenerate_captures()
    names = sorted(caps)
    eng = dab.Engine(0)
    eng.decode([caps[k] for k in names])
    frames_of = {}
    for b, name in enumerate(names):
        eti, calls, _ = ol.ref_frontend_replay(caps[name])
        o_eti, o_trace = ol.or_replay(caps[name])
        ints, ffs = eng.trace(b, len(calls))
        assert len(ints) == len(calls) == len(o_trace), name
        for k, want in enumerate(calls):
            got = (ints[k][0], ints[k][2], ints[k][3], ints[k][4], ints[k][5])
            assert got == want[:5], (name, k, got, want)
            ot = o_trace[k]
            assert (ot.ok, ot.coarse_timeshift, ot.fine_timeshift, ot.coarse_freq_shift, ot.fifo_count) == want[:5], ("oracle", name, k)
            if np.isnan(want[5]):
                assert np.isnan(ffs[k]), (name, k)
            else:
                assert abs(ffs[k] - want[5]) < 1e-6, (name, k, ffs[k], want[5])
        got_eti = eng.eti(b)
        assert got_eti.shape == eti.shape and np.array_equal(got_eti, eti), name
        assert o_eti.shape == eti.shape and np.array_equal(o_eti, eti), ("oracle", name)
        frames_of[name] = eti.shape[0]
    print("ETI frames per capture:", frames_of)
    assert sum(v > 0 for v in frames_of.values()) >= 2, frames_of          # some of them do produce frames: the test is not vacuous
    eng.close()
