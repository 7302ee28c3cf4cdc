"""Oracle restatement vs the real reference objects (oracle/_ref/libdabref.so) on fresh random
inputs.  Skipped when oracle/_ref is not built (it cannot be rebuilt without /root/reference)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from oracle_lib import _ptr


@pytest.fixture(scope="module")
def libs():
    R = ol.ref()
    if R is None:
        pytest.skip("oracle/_ref not built")
    R.refh_new()          # init_dab_state -> init_viterbi() (fills the reference's metric table)
    return ol.oracle(), R


def test_viterbi_random_with_errors_and_erasures(libs):
    O, R = libs
    rng = np.random.default_rng(11)
    failed_decodes = 0
    for trial in range(60):
        nbits = int(rng.choice([192, 768, 1536, 3072]))
        d = rng.integers(0, 256, nbits // 8, dtype=np.uint8)
        s = 127 + 2 * ol.or_encode(d).astype(np.int32)
        s = np.where(rng.random(s.size) < rng.choice([0, 0.02, 0.08, 0.15]), 256 - s, s).astype(np.uint8)
        s[rng.random(s.size) < rng.choice([0, 0.1, 0.3, 0.5])] = 128
        a = ol.or_viterbi(s, nbits)
        b = np.zeros_like(a)
        R.refh_viterbi(None, _ptr(s), _ptr(b), nbits)
        assert np.array_equal(a, b), trial
        failed_decodes += int((a != d).any())
    assert failed_decodes > 5        # the regime where decoded != sent is covered


def test_depuncture_all_profiles(libs):
    O, R = libs
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 2, 60000, dtype=np.uint8)
    for idx in range(64):
        a, b = np.zeros(40000, np.uint8), np.zeros(40000, np.uint8)
        la = O.or_msc_depuncture(_ptr(a), _ptr(bits), C.byref(ol.SubCh(id=1, slform=0, uep_index=idx)))
        lb = R.refh_uep_depuncture(_ptr(b), _ptr(bits), idx)
        assert la == lb and np.array_equal(a, b), idx
    sizemul = [12, 8, 6, 4, 27, 21, 18, 15]
    for pl in range(8):
        for n in (1, 2, 3, 8, 16):
            size, br = sizemul[pl] * n, n * (32 if pl >= 4 else 8)
            if size > 864:
                continue
            a, b = np.zeros(80000, np.uint8), np.zeros(80000, np.uint8)
            la = O.or_msc_depuncture(_ptr(a), _ptr(bits), C.byref(ol.SubCh(id=1, slform=1, protlev=pl, size=size, bitrate=br)))
            lb = R.refh_eep_depuncture(_ptr(b), _ptr(bits), pl, size, br)
            assert la == lb and np.array_equal(a, b), (pl, n)


def test_fifo_shifted_reads(libs):
    """or_sdr's FIFO restatement is exercised end-to-end elsewhere; here the reference's own
    sdr_read_fifo (sdr_fifo.c:43-61) pins the shift semantics the closed form relies on."""
    O, R = libs
    rng = np.random.default_rng(13)
    f = R.refh_fifo_new(4 * 393216)
    data = rng.integers(0, 256, 3 * 262144, dtype=np.uint8)
    R.refh_fifo_write(f, _ptr(data), data.size)
    buf = np.full(393216, 0xAA, np.uint8)
    R.refh_fifo_read(f, 393216, 100, _ptr(buf))                   # positive shift: skip 100, read a full frame
    assert np.array_equal(buf, data[100:100 + 393216])
    assert R.refh_fifo_count(f) == data.size - 100 - 393216
    R.refh_fifo_write(f, _ptr(data), 262144)
    before = buf.copy()
    pos = 100 + 393216
    R.refh_fifo_read(f, 393216, -40, _ptr(buf))                   # negative shift: short read, stale tail kept
    rest = np.concatenate([data[pos:], data[:262144]])
    assert np.array_equal(buf[:393216 - 40], rest[:393216 - 40])
    assert np.array_equal(buf[393216 - 40:], before[393216 - 40:])


def test_backend_process_frame_random_bits(libs):
    """dab_process_frame on structured FIC + random MSC bits: oracle == reference, frame by frame."""
    O, R = libs
    import dabtools_amd as dab
    cfg = dab.synth_preset(0, seed=5, cif_count0=4980)
    rng = np.random.default_rng(14)
    frames = []
    CB = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.c_void_p)
    cb = CB(lambda p, u: frames.append(np.ctypeslib.as_array(p, (6144,)).copy()))
    d = O.or_dab_new(C.cast(cb, C.c_void_p), None)
    H = R.refh_new()
    dep = np.zeros(3096, np.uint8)
    O.or_fic_depuncture(_ptr(dep), _ptr(np.zeros(2304, np.uint8)))
    keep = dep != 128
    for t in range(17):
        fic = np.zeros(9216, np.uint8)
        for q in range(4):
            f = dab.synth_fibs(cfg, 4 * t + q).copy()
            O.or_descramble(_ptr(f), 96)
            fic[2304 * q:2304 * (q + 1)] = ol.or_encode(f)[keep]
        msc = rng.integers(0, 2, 221184, dtype=np.uint8)
        for lib, h, ficp, mscp in ((O, d, O.or_dab_tf_fic, O.or_dab_tf_msc), (R, H, R.refh_tf_fic, R.refh_tf_msc)):
            C.memmove(ficp(h), _ptr(fic), fic.size)
            C.memmove(mscp(h), _ptr(msc), msc.size)
        O.or_dab_process_frame(d)
        R.refh_process(H)
    n = R.refh_neti(H)
    want = np.ctypeslib.as_array(R.refh_eti(H), (n, 6144))
    assert n == 16 and len(frames) == n and np.array_equal(np.array(frames), want)
    O.or_dab_free(d)
