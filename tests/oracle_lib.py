"""ctypes loaders for the CPU oracle (oracle/liboracle.so) and, when present, the real
reference back end (oracle/_ref/libdabref.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

u8p = C.POINTER(C.c_uint8)


def _ptr(a, t=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(t))


class SubCh(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("id", "slform", "uep_index", "start_cu", "size", "bitrate", "protlev", "ascty")]


class EnsInfo(C.Structure):
    _fields_ = [("eid", C.c_uint16), ("cif_hi", C.c_uint8), ("cif_lo", C.c_uint8), ("sub", SubCh * 64)]


class SdrTrace(C.Structure):
    _fields_ = [("ok", C.c_int32), ("read_frame", C.c_int32), ("coarse_timeshift", C.c_int32),
                ("fine_timeshift", C.c_int32), ("coarse_freq_shift", C.c_int32), ("fifo_count", C.c_int32),
                ("fine_freq_shift", C.c_double)]


_oracle = None
_ref = {}


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("or_tables.c", "or_backend.c", "or_frontend.c", "or_soft.c", "dab_oracle.h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def oracle():
    global _oracle
    if _oracle is None:
        L = C.CDLL(build_oracle())
        L.or_uep_table.restype = C.c_void_p
        L.or_puncture_masks.restype = C.POINTER(C.c_uint32)
        L.or_rev_freq_deint_tab.restype = C.POINTER(C.c_uint16)
        L.or_prs_phase.restype = C.POINTER(C.c_int8)
        L.or_msc_depuncture.restype = C.c_int
        L.or_msc_depuncture.argtypes = [u8p, u8p, C.POINTER(SubCh)]
        L.or_crc16_ccitt.restype = C.c_uint16
        L.or_crc16_ccitt.argtypes = [u8p, C.c_int, C.c_uint16]
        L.or_dab_new.restype = C.c_void_p
        L.or_dab_new.argtypes = [C.c_void_p, C.c_void_p]
        for f in ("or_dab_tf_fic", "or_dab_tf_msc"):
            getattr(L, f).restype = u8p
            getattr(L, f).argtypes = [C.c_void_p]
        L.or_dab_process_frame.argtypes = [C.c_void_p]
        L.or_dab_free.argtypes = [C.c_void_p]
        L.or_dab_locked.argtypes = [C.c_void_p]
        L.or_sdr_new.restype = C.c_void_p
        L.or_sdr_free.argtypes = [C.c_void_p]
        L.or_sdr_demod.argtypes = [C.c_void_p, u8p, C.c_int, u8p, u8p]
        L.or_sdr_get_trace.argtypes = [C.c_void_p, C.POINTER(SdrTrace)]
        L.or_sdr_symbols.restype = C.POINTER(C.c_double)
        L.or_sdr_symbols.argtypes = [C.c_void_p]
        L.or_sdr_buffer.restype = u8p
        L.or_sdr_buffer.argtypes = [C.c_void_p]
        L.or_replay.restype = C.c_int
        L.or_replay.argtypes = [u8p, C.c_size_t, u8p, C.c_int, C.POINTER(SdrTrace), C.c_int, C.POINTER(C.c_int)]
        L.or_fine_freq_corr.restype = C.c_double
        L.or_coarse_time_sync.restype = C.c_uint32
        # soft-decision extension (or_soft.c)
        f32p = C.POINTER(C.c_float)
        L.or_soft_quantise.restype = C.c_double
        L.or_soft_quantise.argtypes = [C.c_double, C.c_int]
        L.or_viterbi_soft.argtypes = [f32p, u8p, C.c_int, C.c_int]
        L.or_soft_demap.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
        L.or_fic_decode_soft.restype = C.c_int
        L.or_fic_decode_soft.argtypes = [f32p, C.c_int, u8p, u8p]
        L.or_dab_set_soft.argtypes = [C.c_void_p, C.c_int]
        for f in ("or_dab_tf_sfic", "or_dab_tf_smsc"):
            getattr(L, f).restype = f32p
            getattr(L, f).argtypes = [C.c_void_p]
        L.or_dab_last_fibs.restype = u8p
        L.or_dab_last_fibs.argtypes = [C.c_void_p, u8p]
        L.or_replay_afc.restype = C.c_int
        L.or_replay_afc.argtypes = [u8p, C.c_size_t, u8p, C.c_int, C.POINTER(SdrTrace), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]
        L.or_replay_soft.restype = C.c_int
        L.or_replay_soft.argtypes = [u8p, C.c_size_t, C.c_int, u8p, C.c_int, f32p, C.c_int, C.POINTER(C.c_int)]
        _oracle = L
    return _oracle


def ref(sse=False):
    """The real reference back end, or None when oracle/_ref was not built."""
    key = "sse" if sse else "scalar"
    if key not in _ref:
        so = os.path.join(ORACLE_DIR, "_ref", "libdabref_sse.so" if sse else "libdabref.so")
        if not os.path.exists(so) and os.path.isdir("/root/reference/src"):
            subprocess.call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        if not os.path.exists(so):
            _ref[key] = None
        else:
            L = C.CDLL(so)
            L.refh_new.restype = C.c_void_p
            for f in ("refh_tf_fic", "refh_tf_msc", "refh_eti", "refh_fibs", "refh_fib_ok"):
                getattr(L, f).restype = u8p
            for f in ("refh_tf_fic", "refh_tf_msc", "refh_eti", "refh_process", "refh_neti", "refh_locked", "refh_tfidx"):
                getattr(L, f).argtypes = [C.c_void_p]
            L.refh_fibs.argtypes = [C.c_void_p, C.c_int]
            L.refh_fib_ok.argtypes = [C.c_void_p, C.c_int]
            L.refh_viterbi.argtypes = [C.c_void_p, u8p, u8p, C.c_int]
            L.refh_rev_freq_deint_tab.restype = C.POINTER(C.c_uint16)
            L.refh_pvec.restype = C.POINTER(C.c_char)
            L.refh_fifo_new.restype = C.c_void_p
            L.refh_fifo_write.argtypes = [C.c_void_p, u8p, C.c_int]
            L.refh_fifo_read.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, u8p]
            L.refh_fifo_count.argtypes = [C.c_void_p]
            L.refh_fifo_count.restype = C.c_uint32
            _ref[key] = L
    return _ref[key]


_ref_frontend = {}


def ref_frontend(which="frontend"):
    """which = "frontend": the REAL reference front end (input_sdr.c, sdr_sync.c, sdr_fifo.c, unmodified) with its FFTW3 calls served by the image's
    hipFFTW (oracle/_ref/libdabref_frontend.so; see oracle/ref_frontend_harness.c); which = "hipS2": the same harness over integration/input_sdr_hip.c
    (seam S2 over libdabhip).  None when it was not built.  Both need a GPU at run time."""
    if which not in _ref_frontend:
        so = os.path.join(ORACLE_DIR, "_ref", "libdabref_%s.so" % which)
        if not os.path.exists(so):
            _ref_frontend[which] = None
        else:
            L = C.CDLL(so)
            L.reff_new.restype = C.c_void_p
            L.reff_free.argtypes = [C.c_void_p]
            L.reff_demod.restype = C.c_int
            L.reff_demod.argtypes = [C.c_void_p, u8p, C.c_int, u8p, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
            L.reff_symbols.restype = C.POINTER(C.c_double)
            L.reff_symbols.argtypes = [C.c_void_p]
            L.reff_buffer.restype = u8p
            L.reff_buffer.argtypes = [C.c_void_p]
            _ref_frontend[which] = L
    return _ref_frontend[which]


def ref_frontend_replay(iq, with_backend=True, which="frontend"):
    """dab2eti's loop (dab2eti.c:60-75, no tuner) over the REAL reference objects: sdr_demod of libdabref_frontend.so per 262,144-byte buffer and,
    for every frame it returns, dab_process_frame of libdabref.so.  -> (ETI frames, per-call (ok, cts, fts, cfs, fifo_count, ffs), per-frame bits)"""
    F = ref_frontend(which)
    R = ref() if with_backend else None
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    h = F.reff_new()
    H = R.refh_new() if R is not None else None
    fic, msc = np.zeros(9216, np.uint8), np.zeros(221184, np.uint8)
    ints, ffs = (C.c_int32 * 6)(), C.c_double(0)
    calls, frames = [], []
    for off in range(0, iq.size - 262144 + 1, 262144):
        ok = F.reff_demod(h, _ptr(iq[off:off + 262144]), 262144, _ptr(fic), _ptr(msc), ints, C.byref(ffs))
        calls.append((ints[0], ints[2], ints[3], ints[4], ints[5], ffs.value))
        if ok:
            frames.append((fic.copy(), msc.copy()))
            if H is not None:
                C.memmove(R.refh_tf_fic(H), _ptr(fic), fic.size)
                C.memmove(R.refh_tf_msc(H), _ptr(msc), msc.size)
                R.refh_process(H)
    eti = None
    if H is not None:
        n = R.refh_neti(H)
        eti = np.ctypeslib.as_array(R.refh_eti(H), (n, 6144)).copy() if n else np.zeros((0, 6144), np.uint8)
    F.reff_free(h)
    return eti, calls, frames


# ---- thin numpy wrappers around the oracle -------------------------------------------
def or_viterbi(symbols, nbits):
    sym = np.ascontiguousarray(symbols, dtype=np.uint8)
    assert sym.size >= 4 * (nbits + 6)
    out = np.zeros((nbits + 7) // 8, dtype=np.uint8)
    oracle().or_viterbi(_ptr(sym), _ptr(out), C.c_int(nbits))
    return out


def or_encode(data):
    d = np.ascontiguousarray(data, dtype=np.uint8)
    out = np.zeros(4 * (8 * d.size + 6), dtype=np.uint8)
    oracle().or_encode(_ptr(out), _ptr(d), C.c_uint(d.size))
    return out


def or_replay(iq, cap_frames=4096, trace_cap=4096):
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    eti = np.zeros((cap_frames, 6144), dtype=np.uint8)
    tr = (SdrTrace * trace_cap)()
    nt = C.c_int(0)
    n = oracle().or_replay(_ptr(iq), C.c_size_t(iq.size), _ptr(eti), cap_frames, tr, trace_cap, C.byref(nt))
    assert n <= cap_frames
    return eti[:n], [tr[i] for i in range(min(nt.value, trace_cap))]


SOFT_Q4, SOFT_Q8, SOFT_FLOAT = 4, 8, 32


def or_viterbi_soft(values, nbits, mode=SOFT_Q4):
    """values: 4 (nbits + 6) soft values (> 0: bit 0; 0: punctured) -> decoded bytes (or_soft.c)."""
    v = np.ascontiguousarray(values, dtype=np.float32)
    assert v.size >= 4 * (nbits + 6)
    out = np.zeros((nbits + 7) // 8, dtype=np.uint8)
    oracle().or_viterbi_soft(_ptr(v, C.c_float), _ptr(out), nbits, mode)
    return out


def or_replay_soft(iq, mode=SOFT_Q4, cap_frames=4096, values_tf=0):
    """(ETI frames, values [TF][9216 + 221184] of the first values_tf demodulated TFs, number of demodulated TFs)."""
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    eti = np.zeros((cap_frames, 6144), dtype=np.uint8)
    vals = np.zeros((max(values_tf, 1), 9216 + 221184), dtype=np.float32)
    ntf = C.c_int(0)
    n = oracle().or_replay_soft(_ptr(iq), C.c_size_t(iq.size), mode, _ptr(eti), cap_frames, _ptr(vals, C.c_float) if values_tf else None,
                                values_tf, C.byref(ntf))
    assert n <= cap_frames
    return eti[:n], vals[:min(values_tf, ntf.value)], ntf.value


class SoftDab:
    """or_dab in soft mode: feed values per TF (fic 9216, msc 221184), collect ETI frames and the FIBs of every TF."""

    def __init__(self, mode=SOFT_Q4):
        self.frames, self.fibs = [], []
        self._cbt = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.c_void_p)
        self._cb = self._cbt(lambda p, u: self.frames.append(np.ctypeslib.as_array(p, (6144,)).copy()))
        self.d = oracle().or_dab_new(C.cast(self._cb, C.c_void_p), None)
        oracle().or_dab_set_soft(self.d, mode)

    def process(self, fic, msc):
        O = oracle()
        f = np.ascontiguousarray(fic, dtype=np.float32)
        m = np.ascontiguousarray(msc, dtype=np.float32)
        assert f.size == 9216 and m.size == 221184
        # the TF that process_frame will decode is tfs[tfidx] BEFORE the call; its FIBs are read from there afterwards
        C.memmove(O.or_dab_tf_sfic(self.d), _ptr(f, C.c_float), 4 * f.size)
        C.memmove(O.or_dab_tf_smsc(self.d), _ptr(m, C.c_float), 4 * m.size)
        O.or_dab_process_frame(self.d)

    def close(self):
        if self.d:
            oracle().or_dab_free(self.d)
            self.d = None


def or_replay_afc(iq, cap_frames=4096, trace_cap=4096):
    """or_replay with the tuner feedback of dab2eti.c:76-103 steering an NCO: (ETI frames, per-call traces, per-call NCO frequency)."""
    iq = np.ascontiguousarray(iq, dtype=np.uint8)
    eti = np.zeros((cap_frames, 6144), dtype=np.uint8)
    tr = (SdrTrace * trace_cap)()
    nco = np.zeros(trace_cap, dtype=np.int32)
    nt = C.c_int(0)
    n = oracle().or_replay_afc(_ptr(iq), C.c_size_t(iq.size), _ptr(eti), cap_frames, tr, nco.ctypes.data_as(C.POINTER(C.c_int32)), trace_cap, C.byref(nt))
    assert n <= cap_frames
    k = min(nt.value, trace_cap)
    return eti[:n], [tr[i] for i in range(k)], nco[:k]
