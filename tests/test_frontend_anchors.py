"""Independent anchors for the front end that run WITHOUT a GPU.  The reference's input_sdr.c / sdr_sync.c need an FFTW3 library; libfftw3 is absent,
and the build of them over AMD's hipFFTW (tests/test_gpu_frontend_ref.py, round 4: the real pin) executes on the GPU.  So the CPU suite keeps these.
Nothing below shares code with oracle/or_frontend.c or with the kernels:

  * numpy.fft (pocketfft, fp64) against the oracle's own mixed-radix DFT, both signs, the three sizes used;
  * a NumPy restatement of dab_coarse_time_sync / dab_fine_time_sync / dab_coarse_freq_sync_2 / dab_fine_freq_corr and of
    the OFDM + DQPSK + demap loop, written from sdr_sync.c / input_sdr.c by reading, with the PRS and the frequency
    de-interleaver taken from the reference's literal arrays (tests/golden/tables.npz), against or_* on seeded frames.
These narrow what a consistent misreading could hide; the run of the real front end is tests/test_gpu_frontend_ref.py."""
import ctypes as C
import os

import numpy as np

import dabtools_amd as dab
import oracle_lib as ol

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_T = np.load(os.path.join(G, "tables.npz"))
PRS = np.exp(1j * np.pi / 2 * _T["prs_quarter_turns"].astype(np.float64))       # prs_static, sdr_prstab.c
REV = _T["rev_freq_deint_tab"].astype(np.int64)                                 # dab_tables.c:164


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def test_oracle_dft_against_numpy_fft():
    O = ol.oracle()
    rng = np.random.default_rng(1)
    for n in (2048, 1536, 128):
        for trial in range(3):
            x = rng.integers(-128, 128, n) + 1j * rng.integers(-128, 128, n) if trial else rng.standard_normal(n) + 1j * rng.standard_normal(n)
            xin = np.ascontiguousarray(np.stack([x.real, x.imag], axis=1).astype(np.float64))
            out = np.zeros((n, 2))
            for sign, want in ((-1, np.fft.fft(x)), (+1, np.fft.ifft(x) * n)):      # FFTW_FORWARD = -1, FFTW_BACKWARD = +1, unnormalised
                O.or_dft(n, _dp(xin), _dp(out), sign)
                got = out[:, 0] + 1j * out[:, 1]
                assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max(), (n, sign)


# ---- NumPy restatement, written from the reference text ------------------------------------------------------
def np_coarse_time_sync(real, force):                      # sdr_sync.c:34-68
    a = np.abs(real.astype(np.int32))
    if float(a[0:2656:10].sum()) < 5000 and not force:
        return 0
    sub = a[::10]                                          # real[j + k] with j, k multiples of 10
    nwin = (196608 - 2656) // 10
    c = np.concatenate([[0], np.cumsum(sub)])
    filt = (c[266:266 + nwin] - c[:nwin]).astype(np.float32)      # 266 taps: k = 0, 10, .. 2650
    return int(np.argmin(filt)) * 10 * 2                   # first minimum


def np_fine_time_sync(frame):                              # sdr_sync.c:71-202
    spec = np.fft.fft(frame[2656 + 504:2656 + 504 + 2048])
    idx = np.concatenate([np.arange(768) + 1280, np.arange(768, 1536) - 765])
    conv = spec[idx] * np.conj(PRS)
    mag = np.abs(np.fft.ifft(conv) * 1536).astype(np.float32)
    pos = int(np.argmax(mag))                              # first maximum
    return pos * 2 + 16 if pos < 768 else (pos - 1536) * 2


def np_coarse_freq_sync(sym0):                             # sdr_sync.c:205-258 on the fftshifted symbol 0
    best, best_k = np.float32(-99999), 0
    for k in range(-14, 15):
        conv = np.conj(PRS[14:14 + 128]) * sym0[14 + k + 256:14 + k + 256 + 128]
        m = np.abs(np.fft.ifft(conv) * 128).astype(np.float32).max()
        if m > best:
            best, best_k = m, k
    return best_k


def np_fine_freq_corr(frame):                              # sdr_sync.c:259-302
    left, right = frame[2656 + 2048:2656 + 2048 + 504], frame[2656:2656 + 504]
    return float(np.angle(left * np.conj(right)).sum() / 504 / (2 * np.pi) * 1000)


def np_demap(frame):                                       # input_sdr.c:115-162
    syms = np.stack([np.fft.fftshift(np.fft.fft(frame[2656 + 2552 * i + 504:2656 + 2552 * i + 504 + 2048])) for i in range(76)])
    carriers = np.array([i for i in range(2048) if 255 < i < 1793 and i != 1024])
    bits = np.zeros((75, 3072), np.uint8)
    for j in range(1, 76):
        cur, prev = syms[j][carriers], syms[j - 1][carriers]
        d = cur * np.conj(prev) / np.abs(prev) ** 2
        bits[j - 1, REV] = (d.real <= 0)
        bits[j - 1, 1536 + REV] = (-d.imag > 0)           # the stored imaginary part has the opposite sign (input_sdr.c:139-143)
    return bits[:3].reshape(-1), bits[3:].reshape(-1)


def _frames(n, seed):
    """TF-sized windows at random positions of noisy, frequency-offset synthetic captures (the PRS lands anywhere), plus noise"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        cfg = dab.synth_preset(1, seed=int(rng.integers(1, 1 << 30)), snr_db=float(rng.choice([1000.0, 20.0, 10.0, 6.0, 3.0])),
                               cfo_hz=float(rng.choice([0.0, 0.0, 120.0, -400.0, 1000.0, -3000.0, 9000.0])), amplitude=float(rng.choice([1.0, 0.6])))
        iq = dab.synth_generate(cfg, 4)
        for _ in range(8):
            if rng.random() < 0.5:
                off = int(rng.integers(0, 2 * 196608)) * 2                      # anywhere
            else:
                off = 393216 + int(rng.integers(-800, 800)) * 2                 # near alignment: fine-time range
            out.append(iq[off:off + 393216])
    out.append(rng.integers(0, 256, 393216, dtype=np.uint8))
    return out[:n]


def test_sync_estimators_against_numpy_restatement():
    O = ol.oracle()
    O.or_fine_time_sync.restype = C.c_int32
    O.or_coarse_freq_sync.restype = C.c_int32
    nz_coarse = nz_k = 0
    fines = set()
    for f, buf in enumerate(_frames(100, 7)):
        x = ((buf.astype(np.int32) - 127 + 128) & 255) - 128                    # int8 wrap, input_sdr.c:61-62
        real = np.ascontiguousarray(x[0::2].astype(np.int8))
        frame = (x[0::2] + 1j * x[1::2]).astype(np.complex128)
        fr = np.ascontiguousarray(np.stack([frame.real, frame.imag], axis=1))
        for force in (0, 1):
            want = np_coarse_time_sync(real, force)
            assert O.or_coarse_time_sync(real.ctypes.data_as(C.POINTER(C.c_int8)), force) == want, (f, force)
            nz_coarse += int(want != 0)
        fts = np_fine_time_sync(frame)
        assert O.or_fine_time_sync(_dp(fr)) == fts, f
        fines.add(fts)
        start = 2656 + 505 + fts
        if 0 <= start and start + 2048 <= 196608:
            sym0 = np.fft.fftshift(np.fft.fft(frame[start:start + 2048]))
            s0 = np.ascontiguousarray(np.stack([sym0.real, sym0.imag], axis=1))
            k = np_coarse_freq_sync(sym0)
            assert O.or_coarse_freq_sync(_dp(s0)) == k, f
            nz_k += int(k != 0)
        assert abs(O.or_fine_freq_corr(_dp(fr)) - np_fine_freq_corr(frame)) < 1e-9, f
    assert nz_coarse > 20 and nz_k > 10 and len(fines) > 20                     # the branches were exercised


def test_whole_front_end_bits_against_numpy_restatement():
    """sdr_demod's 230,400 hard bits per TF: oracle vs numpy.fft + the reference's literal de-interleaver table."""
    O = ol.oracle()
    iq = dab.synth_generate(dab.synth_preset(0, seed=77, snr_db=9.0, skip_samples=4321), 8)
    S = O.or_sdr_new()
    fic = np.zeros(dab.FIC_BITS, np.uint8)
    msc = np.zeros(dab.MSC_BITS, np.uint8)
    n = 0
    for off in range(0, iq.size - dab.CHUNK_BYTES + 1, dab.CHUNK_BYTES):
        if O.or_sdr_demod(S, ol._ptr(iq[off:off + dab.CHUNK_BYTES]), dab.CHUNK_BYTES, ol._ptr(fic), ol._ptr(msc)):
            buf = np.ctypeslib.as_array(O.or_sdr_buffer(S), (dab.TF_BYTES,)).astype(np.int32)
            x = ((buf - 127 + 128) & 255) - 128
            wf, wm = np_demap((x[0::2] + 1j * x[1::2]).astype(np.complex128))
            assert np.array_equal(fic, wf) and np.array_equal(msc, wm), off
            n += 1
    O.or_sdr_free(S)
    assert n >= 3
