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
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("or_tables.c", "or_backend.c", "or_frontend.c", "dab_oracle.h")]
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
