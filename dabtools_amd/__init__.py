"""dabtools_amd — Python host side over libdabhip.so (the C ABI of include/dabhip.h).

The product is the shared library; this module is plumbing: it loads the library, declares
the C signatures and offers thin numpy-facing wrappers whose names follow the reference
(`sdr_demod`, `dab_process_frame`, `viterbi`, ...).  There is NO CPU fallback: if the
library is missing, or no MI355X is visible when a decode entry point is used, the call
raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DABHIP_LIB") or os.path.join(_HERE, "libdabhip.so")   # override: experiments with another build

STREAM_MUX_OVERFLOW, STREAM_SUBCH_OUTSIDE_CIF, STREAM_EEP_OPTION, STREAM_SUBCH_SIZE = 1, 2, 4, 8
TF_BYTES = 393216
CHUNK_BYTES = 262144
TAIL_BYTES = 1536      # the last bytes of sdr->buffer, kept as bytes beside the views (csrc/device_types.hpp: kTailBytes)
FIC_BITS = 9216
MSC_BITS = 221184
ETI_BYTES = 6144

u8p = C.POINTER(C.c_uint8)


class DabhipError(RuntimeError):
    pass


class SubChCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("id", "start_cu", "slform", "uep_index", "eep_protlev", "size_cu")]


class ReconfCfg(C.Structure):
    _fields_ = [("at_cif", C.c_int32), ("fic_lead", C.c_int32), ("nsub", C.c_int32), ("pad", C.c_int32), ("sub", SubChCfg * 64)]


class ChannelCfg(C.Structure):
    _fields_ = [("sro_ppm", C.c_double), ("echo_delay", C.c_int32 * 2), ("echo_gain", C.c_double * 2), ("echo_phase", C.c_double * 2),
                ("echo_doppler_hz", C.c_double * 2), ("fade_depth", C.c_double), ("fade_hz", C.c_double),
                ("iq_gain_db", C.c_double), ("iq_phase_deg", C.c_double)]


class SynthCfg(C.Structure):
    _fields_ = [("eid", C.c_uint32), ("nsub", C.c_int32), ("sub", SubChCfg * 64), ("seed", C.c_uint64),
                ("cif_count0", C.c_int32), ("skip_samples", C.c_int32), ("amplitude", C.c_double),
                ("snr_db", C.c_double), ("cfo_hz", C.c_double),
                ("fib_patch_len", C.c_int32), ("fib_patch_from_cif", C.c_int32), ("fib_patch", C.c_uint8 * 32),
                ("reconf", ReconfCfg * 2), ("channel", ChannelCfg)]

    def set_reconf(self, k, at_cif, subs, fic_lead=0):
        """Reconfiguration k (0 / 1): from logical CIF at_cif on the multiplex is `subs`, a list of (id, start_cu, slform, uep_index, eep_protlev,
        size_cu); the FIC announces it fic_lead CIFs earlier."""
        r = self.reconf[k]
        r.at_cif, r.fic_lead, r.nsub = at_cif, fic_lead, len(subs)
        for i, t in enumerate(subs):
            r.sub[i].id, r.sub[i].start_cu, r.sub[i].slform, r.sub[i].uep_index, r.sub[i].eep_protlev, r.sub[i].size_cu = t

    def multiplex(self):
        """The initial multiplex as the tuples set_reconf takes."""
        return [(s.id, s.start_cu, s.slform, s.uep_index, s.eep_protlev, s.size_cu) for s in self.sub[: self.nsub]]

    def set_fib_patch(self, data, from_cif=0):
        """Third FIB of every CIF from `from_cif` on = these bytes (<= 30) under a valid CRC: hostile / non-standard FIGs for tests."""
        data = bytes(data)[:30]
        self.fib_patch_len, self.fib_patch_from_cif = len(data), from_cif
        for i, b in enumerate(data):
            self.fib_patch[i] = b


ETI_CALLBACK = C.CFUNCTYPE(None, u8p)

_SIGNATURES = {
    "dabhip_last_error": (C.c_char_p, []),
    "dabhip_device_count": (C.c_int, []),
    "dabhip_device_identity": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int]),
    "dabhip_create_viterbi": (C.c_void_p, [C.c_int]),
    "dabhip_init_viterbi": (C.c_int, []),
    "dabhip_viterbi": (None, [C.c_void_p, u8p, u8p, C.c_int]),
    "dabhip_viterbi_batch": (C.c_int, [C.c_void_p, u8p, u8p, C.c_int, C.c_int]),
    "dabhip_sdr_init": (C.c_void_p, [C.c_int]),
    "dabhip_sdr_free": (None, [C.c_void_p]),
    "dabhip_sdr_demod": (C.c_int, [C.c_void_p, u8p, C.c_int, u8p, u8p]),
    "dabhip_sdr_coarse_timeshift": (C.c_int32, [C.c_void_p]),
    "dabhip_sdr_fine_timeshift": (C.c_int32, [C.c_void_p]),
    "dabhip_sdr_coarse_freq_shift": (C.c_int32, [C.c_void_p]),
    "dabhip_sdr_fine_freq_shift": (C.c_double, [C.c_void_p]),
    "dabhip_dab_init": (C.c_void_p, [C.c_int, ETI_CALLBACK]),
    "dabhip_dab_free": (None, [C.c_void_p]),
    "dabhip_dab_tf_fic": (u8p, [C.c_void_p]),
    "dabhip_dab_tf_msc": (u8p, [C.c_void_p]),
    "dabhip_dab_process_frame": (C.c_int, [C.c_void_p]),
    "dabhip_dab_locked": (C.c_int, [C.c_void_p]),
    "dabhip_dab_last_fibs": (C.c_int, [C.c_void_p, u8p, u8p]),
    "dabhip_engine_create": (C.c_void_p, [C.c_int]),
    "dabhip_engine_create_ex": (C.c_void_p, [C.c_int, C.c_int]),
    "dabhip_engine_destroy": (None, [C.c_void_p]),
    "dabhip_multi_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int]),
    "dabhip_multi_destroy": (None, [C.c_void_p]),
    "dabhip_multi_slices": (C.c_int, [C.c_void_p]),
    "dabhip_multi_slice_of": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "dabhip_multi_decode": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    "dabhip_multi_eti_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_multi_eti_read": (C.c_int64, [C.c_void_p, C.c_int, u8p, C.c_int64]),
    "dabhip_multi_eti_drain": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dabhip_multi_trace": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int]),
    "dabhip_multi_engine": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dabhip_multi_wall_ms": (C.c_float, [C.c_void_p, C.c_int]),
    "dabhip_multi_set_afc": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_set_soft": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_set_parity_guard": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_set_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_set_subchannels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_engine_decode": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_int]),
    "dabhip_engine_eti_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_engine_eti_read": (C.c_int64, [C.c_void_p, C.c_int, u8p, C.c_int64]),
    "dabhip_engine_eti_drain": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dabhip_engine_eti_device_ptr": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_int64)]),
    "dabhip_engine_set_afc": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_set_soft": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_trace": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int]),
    "dabhip_engine_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    "dabhip_engine_fft_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "dabhip_engine_fft_roofline": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "dabhip_stage_ofdm_fft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "dabhip_stage_demap": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int, u8p, u8p]),
    "dabhip_stage_fic_decode": (C.c_int, [C.c_void_p, u8p, C.c_int, u8p, u8p]),
    "dabhip_host_parse_fibs": (C.c_int, [u8p, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "dabhip_host_eti_header": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), u8p, C.c_int]),
    "dabhip_host_control_replay": (C.c_int, [u8p, u8p, C.c_int, C.POINTER(C.c_int32), u8p, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_host_control_replay_log": (C.c_int64, [C.c_char_p, C.c_int64]),
    "dabhip_engine_stream_log": (C.c_int64, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]),
    "dabhip_stream_log": (C.c_int64, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]),
    "dabhip_multi_stream_log": (C.c_int64, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]),
    "dabhip_multi_stream_log_of": (C.c_int64, [C.c_void_p, C.c_int, C.c_char_p, C.c_int64]),
    "dabhip_dab_take_log": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "dabhip_stream_create_on_cpus": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_multi_stream_create": (C.c_void_p, [C.POINTER(C.c_int), C.c_int, C.c_int]),
    "dabhip_multi_stream_destroy": (None, [C.c_void_p]),
    "dabhip_multi_stream_slices": (C.c_int, [C.c_void_p]),
    "dabhip_multi_stream_streams": (C.c_int, [C.c_void_p]),
    "dabhip_multi_stream_slice_of": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dabhip_multi_stream_session": (C.c_void_p, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_prefetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "dabhip_multi_stream_feed": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "dabhip_multi_stream_feed_resident": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "dabhip_multi_stream_need_from": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_eti_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_status_of": (C.c_uint32, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_eti_read": (C.c_int64, [C.c_void_p, C.c_int, u8p, C.c_int64]),
    "dabhip_multi_stream_eti_drain": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dabhip_multi_stream_eti_fetch": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "dabhip_multi_stream_eti_fetch_wait": (C.c_int, [C.c_void_p]),
    "dabhip_multi_stream_set_afc": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_set_soft": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_set_parity_guard": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_set_sync_speculation": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_set_subchannels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_host_table": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_host_fifo_new": (C.c_void_p, []),
    "dabhip_host_fifo_free": (None, [C.c_void_p]),
    "dabhip_host_fifo_call": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, u8p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int32), u8p]),
    "dabhip_host_fifo_skip_unshifted": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dabhip_synth_preset": (C.c_int, [C.c_int, C.POINTER(SynthCfg)]),
    "dabhip_synth_bytes": (C.c_size_t, [C.POINTER(SynthCfg), C.c_int]),
    "dabhip_synth_generate": (C.c_int64, [C.POINTER(SynthCfg), C.c_int, u8p, C.c_size_t]),
    "dabhip_synth_payload": (C.c_int, [C.POINTER(SynthCfg), C.c_int, C.c_int, u8p, C.c_int]),
    "dabhip_synth_fibs": (C.c_int, [C.POINTER(SynthCfg), C.c_int, u8p]),
    "dabhip_engine_set_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_set_sync_speculation": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_stream_set_sync_speculation": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_set_parity_guard": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_parity_guard_level": (C.c_int, [C.c_void_p]),
    "dabhip_parity_guard_default_level": (C.c_int, []),
    "dabhip_parity_guard_constants": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dabhip_parity_guard_bin_scale": (C.c_double, [C.c_int]),
    "dabhip_engine_guard_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "dabhip_engine_guard_overflows": (C.c_int, [C.c_void_p]),
    "dabhip_engine_set_guard_list_cap": (C.c_int, [C.c_void_p, C.c_uint32]),
    "dabhip_stream_set_parity_guard": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_stage_decision_audit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "dabhip_stage_decision_audit_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "dabhip_engine_set_subchannels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_stream_set_subchannels": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_stream_create": (C.c_void_p, [C.c_int, C.c_int]),
    "dabhip_stream_destroy": (None, [C.c_void_p]),
    "dabhip_stream_feed": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "dabhip_stream_prefetch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]),
    "dabhip_stream_eti_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_stream_eti_read": (C.c_int64, [C.c_void_p, C.c_int, u8p, C.c_int64]),
    "dabhip_stream_eti_drain": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dabhip_stream_set_afc": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_stream_set_soft": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_stream_ceiling": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "dabhip_device_alloc": (C.c_void_p, [C.c_size_t, C.c_int]),
    "dabhip_device_free": (None, [C.c_void_p]),
    "dabhip_device_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "dabhip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "dabhip_host_free": (None, [C.c_void_p]),
    "dabhip_synth_generate_device": (C.c_int, [C.POINTER(SynthCfg), C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int]),
    "dabhip_dab_set_soft": (C.c_int, [C.c_void_p, C.c_int]),
    "dabhip_engine_trace_nco": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_multi_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dabhip_multi_slice_cpus": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "dabhip_engine_create_on_cpus": (C.c_void_p, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_engine_host_cpus": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int)]),
    "dabhip_host_placement_plan": (C.c_int, [C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "dabhip_host_cpu_budget": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dabhip_engine_stream_status": (C.c_uint32, [C.c_void_p, C.c_int]),
    "dabhip_multi_stream_status": (C.c_uint32, [C.c_void_p, C.c_int]),
    "dabhip_stream_status": (C.c_uint32, [C.c_void_p, C.c_int]),
    "dabhip_stream_feed_resident": (C.c_int64, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "dabhip_stream_need_from": (C.c_int64, [C.c_void_p, C.c_int]),
    "dabhip_stream_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int]),
    "dabhip_dab_status": (C.c_uint32, [C.c_void_p]),
    "dabhip_engine_eti_fetch": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "dabhip_engine_eti_fetch_wait": (C.c_int, [C.c_void_p]),
    "dabhip_stream_eti_fetch": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64]),
    "dabhip_stream_eti_fetch_wait": (C.c_int, [C.c_void_p]),
    "dabhip_engine_demapped_tf": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int8), C.POINTER(C.c_int8)]),
}

_lib = None


def build_library(force=False):
    """Compile dabtools_amd/csrc into dabtools_amd/libdabhip.so (hipcc, gfx950)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", csrc, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """The loaded libdabhip.so.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DabhipError("libdabhip.so is missing (%s): run __graft_entry__.build() / make -C dabtools_amd/csrc" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(L, name)   # AttributeError here = header/library mismatch
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def last_error():
    return (lib().dabhip_last_error() or b"").decode()


def _p(a):
    return a.ctypes.data_as(u8p)


def _need(cond, what):
    if not cond:
        raise DabhipError("%s: %s" % (what, last_error()))


def device_identity(device):
    """(PCI bus id, name) of a device index (dabhip_device_identity)."""
    bus, name = C.create_string_buffer(64), C.create_string_buffer(256)
    _need(lib().dabhip_device_identity(int(device), bus, len(bus), name, len(name)) == 0, "device_identity")
    return bus.value.decode(), name.value.decode()


def _guard_level(level):
    """False / 0 -> off, True -> the library's default level (-1), 1 / 2 -> that level (dabhip.h: DABHIP_GUARD_*)."""
    if level is True:
        return -1
    if level is False or level is None:
        return 0
    return int(level)


def guard_constants(level):
    """(bound on a bin's error relative to |x|_2, bound on the product's rounding relative to |cur|_1 |prev|_1) of guard level 1 or 2."""
    a, b = C.c_double(0), C.c_double(0)
    _need(lib().dabhip_parity_guard_constants(int(level), C.byref(a), C.byref(b)) == 0, "parity_guard_constants")
    return a.value, b.value


def guard_bin_scale(raw_bin):
    """Fraction of the proven level's bin constant that bounds raw bin k's error (dabhip_parity_guard_bin_scale)."""
    return lib().dabhip_parity_guard_bin_scale(int(raw_bin))


def guard_default_level():
    return lib().dabhip_parity_guard_default_level()


# ---- synthetic modulator --------------------------------------------------------------------
def synth_preset(preset=0, seed=1, cif_count0=0, skip_samples=0, snr_db=1000.0, amplitude=1.0, cfo_hz=0.0):
    cfg = SynthCfg()
    _need(lib().dabhip_synth_preset(preset, C.byref(cfg)) == 0, "synth_preset")
    cfg.seed = seed
    cfg.cif_count0 = cif_count0
    cfg.skip_samples = skip_samples
    cfg.snr_db = snr_db
    cfg.amplitude = amplitude
    cfg.cfo_hz = cfo_hz
    return cfg


def synth_generate(cfg, ntf):
    n = lib().dabhip_synth_bytes(C.byref(cfg), ntf)
    iq = np.empty(n, dtype=np.uint8)
    got = lib().dabhip_synth_generate(C.byref(cfg), ntf, _p(iq), n)
    if cfg.channel.sro_ppm != 0.0:             # the resampler decides the count; n is the capacity
        _need(0 < got <= n, "synth_generate")
        return iq[:got].copy()
    _need(got == n, "synth_generate")
    return iq


def synth_bytes(cfg, ntf):
    return lib().dabhip_synth_bytes(C.byref(cfg), ntf)


def synth_generate_device(cfgs, ntf, ptrs, device=0):
    """Modulate the ensembles cfgs on the GPU into the device buffers at ptrs (ints; synth_bytes(cfg, ntf) bytes each)."""
    arr = (SynthCfg * len(cfgs))(*cfgs)
    p = (C.c_void_p * len(ptrs))(*ptrs)
    _need(lib().dabhip_synth_generate_device(arr, len(cfgs), ntf, p, device) == 0, "synth_generate_device")


def synth_payload(cfg, cif_index, slot):
    buf = np.zeros(1152 * 2, dtype=np.uint8)
    n = lib().dabhip_synth_payload(C.byref(cfg), cif_index, slot, _p(buf), buf.size)
    _need(n >= 0, "synth_payload")
    return buf[:n].copy()


def synth_fibs(cfg, cif_index):
    buf = np.zeros(96, dtype=np.uint8)
    _need(lib().dabhip_synth_fibs(C.byref(cfg), cif_index, _p(buf)) == 96, "synth_fibs")
    return buf


def stream_ceiling(device=0, nbytes=4 << 30, reps=3):
    """GB/s a bare streaming kernel reaches on this device: {"fill", "copy", "k2_mix"} (k_probe.hip)."""
    g = (C.c_double * 3)()
    _need(lib().dabhip_stream_ceiling(device, nbytes, reps, g) == 0, "stream_ceiling")
    return {"fill": g[0], "copy": g[1], "k2_mix": g[2]}


class HostBuffer:
    """nbytes of page-locked host memory (dabhip_host_alloc): .array is a numpy view, .ptr its address.  Uploads from it are
    plain asynchronous DMA at the PCIe rate; pageable memory goes through the engine's staging ring instead."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().dabhip_host_alloc(max(self.nbytes, 1))
        _need(self.ptr, "host_alloc")
        self.array = np.ctypeslib.as_array(C.cast(self.ptr, u8p), (self.nbytes,))

    def free(self):
        if self.ptr:
            self.array = None
            lib().dabhip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """nbytes of device memory (dabhip_device_alloc): .ptr for the entry points that take device pointers."""

    def __init__(self, nbytes, device=0):
        self.nbytes = int(nbytes)
        self.ptr = lib().dabhip_device_alloc(self.nbytes, device)
        _need(self.ptr, "device_alloc")

    def upload(self, array):
        a = np.ascontiguousarray(array, dtype=np.uint8)
        _need(a.size <= self.nbytes and lib().dabhip_device_copy(self.ptr, a.ctypes.data, a.size, 1) == 0, "device_copy")

    def download(self):
        out = np.empty(self.nbytes, dtype=np.uint8)
        _need(lib().dabhip_device_copy(out.ctypes.data, self.ptr, self.nbytes, 0) == 0, "device_copy")
        return out

    def free(self):
        if self.ptr:
            lib().dabhip_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- host-side control plane (no GPU needed) ----------------------------------------------------
def host_parse_fibs(fibs, crc_ok):
    fibs = np.ascontiguousarray(fibs, dtype=np.uint8)
    crc_ok = np.ascontiguousarray(crc_ok, dtype=np.uint8)
    hdr = np.zeros(3, dtype=np.int32)
    sub = np.zeros((64, 8), dtype=np.int32)
    _need(lib().dabhip_host_parse_fibs(_p(fibs), _p(crc_ok), hdr.ctypes.data_as(C.POINTER(C.c_int32)),
                                       sub.ctypes.data_as(C.POINTER(C.c_int32))) == 0, "host_parse_fibs")
    return hdr, sub


def host_eti_header(hdr, sub):
    hdr = np.ascontiguousarray(hdr, dtype=np.int32)
    sub = np.ascontiguousarray(sub, dtype=np.int32)
    out = np.zeros(272, dtype=np.uint8)
    n = lib().dabhip_host_eti_header(hdr.ctypes.data_as(C.POINTER(C.c_int32)), sub.ctypes.data_as(C.POINTER(C.c_int32)), _p(out), out.size)
    _need(n > 0, "host_eti_header")
    return out[:n].copy()


def host_control_replay(fibs, crc_ok):
    fibs = np.ascontiguousarray(fibs, dtype=np.uint8).reshape(-1, 384)
    crc_ok = np.ascontiguousarray(crc_ok, dtype=np.uint8).reshape(-1, 12)
    ntf = fibs.shape[0]
    cap = 4 * ntf + 4
    first = np.zeros(cap, dtype=np.int32)
    hlen = np.zeros(cap, dtype=np.int32)
    hdrs = np.zeros((cap, 272), dtype=np.uint8)
    n = lib().dabhip_host_control_replay(_p(fibs), _p(crc_ok), ntf, first.ctypes.data_as(C.POINTER(C.c_int32)), _p(hdrs),
                                         hlen.ctypes.data_as(C.POINTER(C.c_int32)), cap)
    _need(n >= 0, "host_control_replay")
    return first[:n], [hdrs[i, :hlen[i]].copy() for i in range(n)]


def host_control_replay_log():
    """The operator messages (dab.c:51,57,78-82) of this thread's last host_control_replay, as text."""
    buf = C.create_string_buffer(1 << 16)
    _need(lib().dabhip_host_control_replay_log(buf, len(buf)) >= 0, "host_control_replay_log")
    return buf.value.decode("ascii")


def host_placement_plan(slice_node, node_cpulist, ncpu):
    """dabhip_host_placement_plan: cpu_slice[c] = slice CPU c is given to (-1: none), for slices on NUMA nodes slice_node[i]."""
    nodes = (C.c_int32 * len(slice_node))(*slice_node)
    lists = (C.c_char_p * len(node_cpulist))(*[s.encode() for s in node_cpulist])
    out = (C.c_int32 * ncpu)()
    n = lib().dabhip_host_placement_plan(nodes, len(slice_node), lists, len(node_cpulist), out, ncpu)
    _need(n >= 0, "host_placement_plan")
    return n, list(out)


def host_cpu_budget():
    """(usable CPUs, CPUs in the affinity mask, CFS quota in CPUs or 0) as the library sizes its host pools (csrc/placement.hpp)."""
    a, q = C.c_int(0), C.c_int(0)
    n = lib().dabhip_host_cpu_budget(C.byref(a), C.byref(q))
    return n, a.value, q.value


def host_table(which):
    """The product's constant tables (dabhip_host_table): 0 UEP profiles (64 x 11), 1 puncturing vectors (24 x 32),
    2 frequency de-interleaver (1536), 3 phase reference symbol quarter turns (1536)."""
    out = np.zeros(4096, dtype=np.int32)
    n = lib().dabhip_host_table(which, out.ctypes.data_as(C.POINTER(C.c_int32)), out.size)
    _need(n > 0, "host_table")
    shape = {0: (64, 11), 1: (24, 32)}.get(which, (n,))
    return out[:n].reshape(shape)


class HostFifo:
    """The FIFO / frame-buffer bookkeeping of the sync-scan kernel, on the host (dabhip_host_fifo_*)."""

    def __init__(self):
        self._h = lib().dabhip_host_fifo_new()
        _need(self._h, "host_fifo_new")

    def call(self, coarse_timeshift, fine_timeshift, stream=None, chunk_bytes=CHUNK_BYTES):
        """One sdr_demod call appending chunk_bytes -> (status 0/1/2, [(seg_end, seg_src), ...], fifo_count).  stream (uint8 array holding at least
        the bytes fed so far): the last 1536 bytes of sdr->buffer are tracked as bytes and left in self.tail."""
        nseg, cnt = C.c_int32(0), C.c_int32(0)
        ends = np.zeros(12, dtype=np.int32)
        srcs = np.zeros(12, dtype=np.int64)
        self.tail = np.zeros(TAIL_BYTES, dtype=np.uint8) if stream is not None else None
        r = lib().dabhip_host_fifo_call(self._h, coarse_timeshift, fine_timeshift, chunk_bytes, _p(stream) if stream is not None else None, C.byref(nseg),
                                        ends.ctypes.data_as(C.POINTER(C.c_int32)), srcs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(cnt),
                                        _p(self.tail) if stream is not None else None)
        _need(r >= 0, "host_fifo_call")
        return r, [(int(ends[i]), int(srcs[i])) for i in range(nseg.value)], cnt.value

    def skip_unshifted(self, ncalls):
        """ncalls further 262144-byte calls without any time shift, counters only (dabhip_host_fifo_skip_unshifted) -> (fed, consumed)"""
        fed, consumed = C.c_int64(0), C.c_int64(0)
        _need(lib().dabhip_host_fifo_skip_unshifted(self._h, ncalls, C.byref(fed), C.byref(consumed)) == 0, "host_fifo_skip_unshifted")
        return fed.value, consumed.value

    @staticmethod
    def materialise(stream, view, tail=None):
        """The 393216 bytes of sdr->buffer a view (and the tail bytes kept beside it) describe."""
        buf = np.zeros(TF_BYTES, dtype=np.uint8)
        lo = 0
        for end, src in view:
            if src >= 0:
                buf[lo:end] = stream[src + lo:src + end]
            lo = end
        if tail is not None:
            buf[TF_BYTES - TAIL_BYTES:] = tail
        return buf

    def close(self):
        if self._h:
            lib().dabhip_host_fifo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- S1 -----------------------------------------------------------------------------------------
def viterbi(symbols, framebits, n=1):
    """Batch form of the reference's viterbi(p, symbols, data, framebits) (viterbi.h:8)."""
    sym = np.ascontiguousarray(symbols, dtype=np.uint8)
    assert sym.size == n * 4 * (framebits + 6)
    out = np.zeros((n, framebits // 8), dtype=np.uint8)
    r = lib().dabhip_viterbi_batch(None, _p(sym), _p(out), framebits, n)
    _need(r == n, "viterbi")
    return out


# ---- S2 -----------------------------------------------------------------------------------------
class Sdr:
    """sdr_init + sdr_demod (input_sdr.h:43-44) for one stream."""

    def __init__(self, device=0):
        self._h = lib().dabhip_sdr_init(device)
        _need(self._h, "sdr_init")
        self.fic = np.zeros(FIC_BITS, dtype=np.uint8)
        self.msc = np.zeros(MSC_BITS, dtype=np.uint8)

    def demod(self, chunk):
        chunk = np.ascontiguousarray(chunk, dtype=np.uint8)
        r = lib().dabhip_sdr_demod(self._h, _p(chunk), chunk.size, _p(self.fic), _p(self.msc))
        _need(r >= 0, "sdr_demod")
        return r

    @property
    def state(self):
        L = lib()
        return (L.dabhip_sdr_coarse_timeshift(self._h), L.dabhip_sdr_fine_timeshift(self._h),
                L.dabhip_sdr_coarse_freq_shift(self._h), L.dabhip_sdr_fine_freq_shift(self._h))

    def close(self):
        if self._h:
            lib().dabhip_sdr_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- S3 -----------------------------------------------------------------------------------------
class Dab:
    """init_dab_state + dab_process_frame (dab.h:91-92) for one stream."""

    def __init__(self, device=0, soft=False):
        self.frames = []
        self._cb = ETI_CALLBACK(lambda p: self.frames.append(np.frombuffer(C.string_at(p, ETI_BYTES), np.uint8)))
        self._h = lib().dabhip_dab_init(device, self._cb)
        _need(self._h, "dab_init")
        self.fic = np.ctypeslib.as_array(lib().dabhip_dab_tf_fic(self._h), (FIC_BITS,))
        self.msc = np.ctypeslib.as_array(lib().dabhip_dab_tf_msc(self._h), (MSC_BITS,))
        if soft:       # extension: the hand-off carries signed 4-bit values (int8 views of the same arrays)
            _need(lib().dabhip_dab_set_soft(self._h, 1) == 0, "dab_set_soft")
            self.fic, self.msc = self.fic.view(np.int8), self.msc.view(np.int8)

    def process_frame(self):
        r = lib().dabhip_dab_process_frame(self._h)
        _need(r >= 0, "dab_process_frame")
        return r

    @property
    def locked(self):
        return bool(lib().dabhip_dab_locked(self._h))

    def take_log(self):
        """What the reference's dab_process_frame would have printed on stderr since the last call (dabhip_dab_take_log)."""
        buf = C.create_string_buffer(1 << 16)
        _need(lib().dabhip_dab_take_log(self._h, buf, len(buf)) >= 0, "dab_take_log")
        return buf.value.decode("ascii")

    @property
    def status(self):
        return int(lib().dabhip_dab_status(self._h))

    def last_fibs(self):
        fibs = np.zeros((12, 32), dtype=np.uint8)
        ok = np.zeros(12, dtype=np.uint8)
        lib().dabhip_dab_last_fibs(self._h, _p(fibs), _p(ok))
        return fibs, ok

    def close(self):
        if self._h:
            lib().dabhip_dab_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- batch engine -------------------------------------------------------------------------------
class Engine:
    def __init__(self, device=0, host_threads=0, _borrowed=None):
        self._owned = _borrowed is None
        self._h = _borrowed or lib().dabhip_engine_create_ex(device, host_threads)
        _need(self._h, "engine_create")
        self.nstreams = 0

    def set_afc(self, enable):
        """Software AFC (NCO per stream steered by the reference's tuner rule); off = parity mode."""
        _need(lib().dabhip_engine_set_afc(self._h, 1 if enable else 0) == 0, "set_afc")

    def set_soft(self, enable):
        """Soft-decision decoding (4-bit soft values into the Viterbi metrics); off = parity mode."""
        _need(lib().dabhip_engine_set_soft(self._h, 1 if enable else 0) == 0, "set_soft")

    def set_subchannels(self, ids):
        """Decode and carry only these SubChIds (None / empty = all, the reference's frames)."""
        ids = list(ids or [])
        arr = (C.c_int32 * max(len(ids), 1))(*ids)
        _need(lib().dabhip_engine_set_subchannels(self._h, arr, len(ids)) == 0, "set_subchannels")

    def set_fused(self, enable):
        """True (default): one kernel for OFDM transform + demap (spectra never written); False: K2 + K2b.  Identical output."""
        _need(lib().dabhip_engine_set_fused(self._h, 1 if enable else 0) == 0, "set_fused")

    def set_sync_speculation(self, mode):
        """K1's chain: 0 = call after call, 1 = with the look-ahead pass, -1 (default) = the pass for small batches.  Identical results."""
        _need(lib().dabhip_engine_set_sync_speculation(self._h, int(mode)) == 0, "set_sync_speculation")

    def set_parity_guard(self, level=True):
        """Decisions inside the fp32 error band are re-decided in fp64 -> bits of exact arithmetic.  level: False / 0 = off, 1 = the measured
        band, 2 = the proven band, True = the library's default level (dabhip.h: DABHIP_GUARD_*)."""
        _need(lib().dabhip_engine_set_parity_guard(self._h, _guard_level(level)) == 0, "set_parity_guard")

    def parity_guard_level(self):
        return lib().dabhip_engine_parity_guard_level(self._h)

    def guard_stats(self):
        """(decisions re-decided by the parity guard, hard decisions taken) of the last decode."""
        a, b = C.c_int64(0), C.c_int64(0)
        _need(lib().dabhip_engine_guard_stats(self._h, C.byref(a), C.byref(b)) == 0, "guard_stats")
        return a.value, b.value

    def guard_overflows(self):
        """Launches of the last decode whose guard list overflowed (their frames were decided again in full, in fp64)."""
        return lib().dabhip_engine_guard_overflows(self._h)

    def set_guard_list_cap(self, cap):
        """Test knob: capacity of the guard's list per launch (0 = automatic)."""
        _need(lib().dabhip_engine_set_guard_list_cap(self._h, cap) == 0, "set_guard_list_cap")

    def decision_audit(self, frames=None, device_ptr=None, nframes=None, guard=False, fused=False):
        """fp32 OFDM stage vs fp64 on contiguous cu8 frames -> dict (see dabhip_stage_decision_audit); fused: the default decode's one-kernel stage
        (dabhip_stage_decision_audit_fused) instead of K2 + K2b."""
        if device_ptr is None:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            nframes = frames.size // TF_BYTES
            src, on_dev = frames.ctypes.data, 0
        else:
            src, on_dev = device_ptr, 1
        keys = ("decisions", "disagree", "disagree_outside_guard", "flagged_by_rule", "max_bin_err", "max_dec_err", "max_prod_err", "listed")
        if fused:
            out = (C.c_double * 10)()
            _need(lib().dabhip_stage_decision_audit_fused(self._h, src, nframes, on_dev, 1 if guard else 0, out) == nframes, "stage_decision_audit_fused")
            d = dict(zip(keys, list(out)[:8]))
            d["shipping_kernel_same_bits"], d["shipping_kernel_same_list_count"] = out[8], out[9]
            return d
        out = (C.c_double * 8)()
        _need(lib().dabhip_stage_decision_audit(self._h, src, nframes, on_dev, 1 if guard else 0, out) == nframes, "stage_decision_audit")
        return dict(zip(keys, list(out)))

    def decode(self, streams):
        """streams: list of numpy uint8 arrays (host) -> total ETI frames."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        sizes = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
        n = lib().dabhip_engine_decode(self._h, ptrs, sizes, len(arrs), 0)
        _need(n >= 0, "engine_decode")
        self.nstreams = len(arrs)
        return n

    def decode_host_ptrs(self, ptrs, sizes):
        """ptrs: HOST addresses (ints; page-locked memory from HostBuffer uploads at the PCIe rate), sizes: byte counts."""
        p = (C.c_void_p * len(ptrs))(*ptrs)
        s = (C.c_size_t * len(sizes))(*sizes)
        n = lib().dabhip_engine_decode(self._h, p, s, len(ptrs), 0)
        _need(n >= 0, "engine_decode")
        self.nstreams = len(ptrs)
        return n

    def decode_device(self, ptrs, sizes):
        """ptrs: device addresses (ints, e.g. torch tensor.data_ptr()), sizes: byte counts."""
        return self.decode_marshalled(self.marshal(ptrs, sizes), on_device=True)

    @staticmethod
    def marshal(ptrs, sizes):
        """The C argument arrays of a decode, built once for callers that decode the same buffers again and again."""
        return (C.c_void_p * len(ptrs))(*ptrs), (C.c_size_t * len(sizes))(*sizes), len(ptrs)

    def decode_marshalled(self, args, on_device=True):
        p, s, n_streams = args
        n = lib().dabhip_engine_decode(self._h, p, s, n_streams, 1 if on_device else 0)
        _need(n >= 0, "engine_decode")
        self.nstreams = n_streams
        return n

    def eti(self, stream):
        n = lib().dabhip_engine_eti_count(self._h, stream)
        _need(n >= 0, "eti_count")
        out = np.zeros((n, ETI_BYTES), dtype=np.uint8)
        if n:
            _need(lib().dabhip_engine_eti_read(self._h, stream, _p(out), n) == n, "eti_read")
        return out

    def eti_count(self, stream):
        return lib().dabhip_engine_eti_count(self._h, stream)

    def stream_status(self, stream):
        """dabhip_engine_stream_status: 0 = fine, else STREAM_* fault bits (the stream emitted no frames while its multiplex was un-assemblable)."""
        return int(lib().dabhip_engine_stream_status(self._h, stream))

    def log(self, stream):
        """The reference's operator messages for one stream of the last decode, cleared by the call (dabhip_engine_stream_log): 'Locked' (dab.c:51),
        'Lock lost, resetting ringbuffer' (dab.c:57), the one-time ensemble dump (dab.c:78-82, misc.c:316-328)."""
        buf = C.create_string_buffer(1 << 16)
        _need(lib().dabhip_engine_stream_log(self._h, stream, buf, len(buf)) >= 0, "engine_stream_log")
        return buf.value.decode("ascii")

    def eti_fetch(self, dst_ptr, cap_frames):
        """dabhip_engine_eti_fetch: all frames of the last decode (stream-major) on their way to (page-locked) host memory."""
        n = lib().dabhip_engine_eti_fetch(self._h, dst_ptr, cap_frames)
        _need(n >= 0, "eti_fetch")
        return n

    def eti_fetch_wait(self):
        _need(lib().dabhip_engine_eti_fetch_wait(self._h) == 0, "eti_fetch_wait")

    def demapped_tf(self, stream, tf):
        """(fic[9216], msc[221184]) int8: what the OFDM stage of the last decode left for that TF (0/1, or -7..7 with soft decisions)."""
        fic, msc = np.zeros(FIC_BITS, dtype=np.int8), np.zeros(MSC_BITS, dtype=np.int8)
        _need(lib().dabhip_engine_demapped_tf(self._h, stream, tf, fic.ctypes.data_as(C.POINTER(C.c_int8)), msc.ctypes.data_as(C.POINTER(C.c_int8))) == 0, "demapped_tf")
        return fic, msc

    def eti_device_ptr(self):
        n = C.c_int64(0)
        p = lib().dabhip_engine_eti_device_ptr(self._h, C.byref(n))
        return p, n.value

    def trace(self, stream, ncalls):
        ints = np.zeros((ncalls, 6), dtype=np.int32)
        ffs = np.zeros(ncalls, dtype=np.float64)
        n = lib().dabhip_engine_trace(self._h, stream, ints.ctypes.data_as(C.POINTER(C.c_int32)),
                                      ffs.ctypes.data_as(C.POINTER(C.c_double)), ncalls)
        _need(n >= 0, "engine_trace")
        return ints[:n], ffs[:n]

    def trace_nco(self, stream, ncalls):
        """per call: the re-tuning in Hz the software AFC applied to that call's samples (dabhip_engine_trace_nco)"""
        out = np.zeros(ncalls, dtype=np.int32)
        n = lib().dabhip_engine_trace_nco(self._h, stream, out.ctypes.data_as(C.POINTER(C.c_int32)), ncalls)
        _need(n >= 0, "trace_nco")
        return out[:n]

    def stage_ms(self):
        names = (C.c_char_p * 24)()
        ms = (C.c_float * 24)()
        n = lib().dabhip_engine_stage_ms(self._h, names, ms, 24)
        return {names[i].decode(): ms[i] for i in range(n)}

    def fft_stats(self):
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_double(0)
        lib().dabhip_engine_fft_stats(self._h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def fft_roofline(self, reps=3):
        """K2 alone over the frames of the last decode -> (launches, TFs, total kernel ms)."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_double(0)
        _need(lib().dabhip_engine_fft_roofline(self._h, reps, C.byref(a), C.byref(b), C.byref(c)) == 0, "fft_roofline")
        return a.value, b.value, c.value

    def stage_ofdm_fft(self, frames, reps=1, device_ptr=None, nframes=None, want_output=True):
        if device_ptr is None:
            frames = np.ascontiguousarray(frames, dtype=np.uint8)
            nframes = frames.size // TF_BYTES
            src, on_dev = frames.ctypes.data, 0
        else:
            src, on_dev = device_ptr, 1
        out = np.zeros((nframes, 76, 2048, 2), dtype=np.float32) if want_output else None
        ms = C.c_float(0)
        r = lib().dabhip_stage_ofdm_fft(self._h, src, nframes, out.ctypes.data_as(C.POINTER(C.c_float)) if want_output else None,
                                        on_dev, reps, C.byref(ms))
        _need(r == nframes, "stage_ofdm_fft")
        return out, ms.value

    def stage_demap(self, spectra):
        spectra = np.ascontiguousarray(spectra, dtype=np.float32)
        n = spectra.shape[0]
        fic = np.zeros((n, FIC_BITS), dtype=np.uint8)
        msc = np.zeros((n, MSC_BITS), dtype=np.uint8)
        _need(lib().dabhip_stage_demap(self._h, spectra.ctypes.data_as(C.POINTER(C.c_float)), n, _p(fic), _p(msc)) == n, "stage_demap")
        return fic, msc

    def stage_fic_decode(self, fic):
        fic = np.ascontiguousarray(fic, dtype=np.uint8).reshape(-1, FIC_BITS)
        n = fic.shape[0]
        fibs = np.zeros((n, 12, 32), dtype=np.uint8)
        ok = np.zeros((n, 12), dtype=np.uint8)
        _need(lib().dabhip_stage_fic_decode(self._h, _p(fic), n, _p(fibs), _p(ok)) == n, "stage_fic_decode")
        return fibs, ok

    def close(self):
        if self._h and self._owned:
            lib().dabhip_engine_destroy(self._h)
        self._h = None

    def __del__(self):
        try:                       # at interpreter shutdown the module globals may already be gone
            self.close()
        except Exception:
            pass


class Multi:
    """The batch engine over several devices of one node (dabhip_multi_*): streams dealt to the devices in contiguous slices
    (2048 on 8 = stream s on device s // 256), all slices decoding at once, no exchange between them.  A device may be listed
    several times (every entry is its own slice)."""

    def __init__(self, devices):
        devs = list(devices)
        self._h = lib().dabhip_multi_create((C.c_int * len(devs))(*devs), len(devs))
        _need(self._h, "multi_create")
        self.devices = devs
        self.nstreams = 0

    def _set(self, fn, enable):
        _need(fn(self._h, 1 if enable else 0) == 0, "multi_set")

    def set_afc(self, enable):
        self._set(lib().dabhip_multi_set_afc, enable)

    def set_soft(self, enable):
        self._set(lib().dabhip_multi_set_soft, enable)

    def set_parity_guard(self, level=True):
        _need(lib().dabhip_multi_set_parity_guard(self._h, _guard_level(level)) == 0, "multi_set_parity_guard")

    def set_fused(self, enable):
        self._set(lib().dabhip_multi_set_fused, enable)

    def set_subchannels(self, ids):
        ids = list(ids or [])
        _need(lib().dabhip_multi_set_subchannels(self._h, (C.c_int32 * max(len(ids), 1))(*ids), len(ids)) == 0, "multi_set_subchannels")

    def decode(self, streams):
        """streams: list of numpy uint8 arrays (host) -> total ETI frames."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in streams]
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        sizes = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
        n = lib().dabhip_multi_decode(self._h, ptrs, sizes, len(arrs), 0)
        _need(n >= 0, "multi_decode")
        self.nstreams = len(arrs)
        return n

    def decode_device(self, ptrs, sizes):
        """ptrs[b]: device address on the device of stream b's slice (slice_of)."""
        p = (C.c_void_p * len(ptrs))(*ptrs)
        s = (C.c_size_t * len(sizes))(*sizes)
        n = lib().dabhip_multi_decode(self._h, p, s, len(ptrs), 1)
        _need(n >= 0, "multi_decode")
        self.nstreams = len(ptrs)
        return n

    def plan(self, nstreams, stream):
        """(slice, device) that will decode `stream` of a batch of `nstreams` (dabhip_multi_plan: the dealing rule, before any decode)."""
        sl, dev = C.c_int(-1), C.c_int(-1)
        _need(lib().dabhip_multi_plan(self._h, nstreams, stream, C.byref(sl), C.byref(dev)) == 0, "multi_plan")
        return sl.value, dev.value

    def slice_cpus(self, slice_index):
        """(cpus, numa_node) the host threads of a slice are bound to ([] = unbound)."""
        buf, node = (C.c_int32 * 4096)(), C.c_int(-1)
        n = lib().dabhip_multi_slice_cpus(self._h, slice_index, buf, 4096, C.byref(node))
        _need(n >= 0, "multi_slice_cpus")
        return list(buf[:n]), node.value

    def slice_of(self, stream):
        """(slice, device) of a stream in a decode of the last decode's size."""
        dev = C.c_int(0)
        sl = lib().dabhip_multi_slice_of(self._h, stream, C.byref(dev))
        _need(sl >= 0, "multi_slice_of")
        return sl, dev.value

    def eti_count(self, stream):
        return lib().dabhip_multi_eti_count(self._h, stream)

    def log(self, stream):
        buf = C.create_string_buffer(1 << 16)
        _need(lib().dabhip_multi_stream_log(self._h, stream, buf, len(buf)) >= 0, "multi_stream_log")
        return buf.value.decode("ascii")

    def eti(self, stream):
        n = lib().dabhip_multi_eti_count(self._h, stream)
        _need(n >= 0, "multi_eti_count")
        out = np.zeros((n, ETI_BYTES), dtype=np.uint8)
        if n:
            _need(lib().dabhip_multi_eti_read(self._h, stream, _p(out), n) == n, "multi_eti_read")
        return out

    def drain(self):
        """All frames through dabhip_multi_eti_drain -> [(stream, frame bytes)] in emission order."""
        got = []
        sink = C.CFUNCTYPE(None, u8p, C.c_int, C.c_void_p)(lambda p, b, _u: got.append((b, C.string_at(p, ETI_BYTES))))      # (not numpy's as_array: it holds on to ~100 bytes per distinct address)
        n = lib().dabhip_multi_eti_drain(self._h, C.cast(sink, C.c_void_p), None)
        _need(n == len(got), "multi_eti_drain")
        return got

    def trace(self, stream, ncalls):
        ints = np.zeros((ncalls, 6), dtype=np.int32)
        ffs = np.zeros(ncalls, dtype=np.float64)
        n = lib().dabhip_multi_trace(self._h, stream, ints.ctypes.data_as(C.POINTER(C.c_int32)), ffs.ctypes.data_as(C.POINTER(C.c_double)), ncalls)
        _need(n >= 0, "multi_trace")
        return ints[:n], ffs[:n]

    def engine(self, slice_index):
        """The slice's engine (borrowed: stage_ms(), guard_stats(), ...)."""
        h = lib().dabhip_multi_engine(self._h, slice_index)
        _need(h, "multi_engine")
        return Engine(_borrowed=h)

    def wall_ms(self, slice_index=-1):
        return lib().dabhip_multi_wall_ms(self._h, slice_index)

    def close(self):
        if self._h:
            lib().dabhip_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """Streaming session (dabhip_stream_*): B unbounded captures decoded segment by segment; the concatenated ETI
    frames equal one Engine.decode of the whole captures."""

    _PREFIX = "dabhip_stream_"
    _RENAMED = {}

    def _f(self, name):
        return getattr(lib(), self._PREFIX + self._RENAMED.get(name, name))

    def __init__(self, nstreams, device=0, afc=False, soft=False, subchannels=None, guard=None):
        self._h = lib().dabhip_stream_create(device, nstreams)
        _need(self._h, "stream_create")
        self.nstreams = nstreams
        self._configure(afc, soft, subchannels, guard)

    def _configure(self, afc, soft, subchannels, guard):
        if subchannels:
            ids = list(subchannels)
            _need(self._f("set_subchannels")(self._h, (C.c_int32 * len(ids))(*ids), len(ids)) == 0, "stream_set_subchannels")
        if afc:
            _need(self._f("set_afc")(self._h, 1) == 0, "stream_set_afc")
        if soft:
            _need(self._f("set_soft")(self._h, 1) == 0, "stream_set_soft")
        if guard is not None:
            self.set_parity_guard(guard)

    def set_parity_guard(self, level=True):
        """see Engine.set_parity_guard"""
        _need(self._f("set_parity_guard")(self._h, _guard_level(level)) == 0, "stream_set_parity_guard")

    def log(self, stream):
        """The reference's operator messages for one stream since the last call (dabhip_stream_log): 'Locked', 'Lock lost, resetting ringbuffer', ensemble dump."""
        buf = C.create_string_buffer(1 << 16)
        n = self._f("log")(self._h, stream, buf, len(buf))
        _need(n >= 0, "stream_log")
        return buf.value.decode("ascii")

    def feed(self, segments):
        """segments: one numpy uint8 array (possibly empty) per stream -> ETI frames produced by this segment."""
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) for s in segments]
        _need(len(arrs) == self.nstreams, "stream_feed: one segment per stream")
        ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
        sizes = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
        n = self._f("feed")(self._h, ptrs, sizes, 0)
        _need(n >= 0, "stream_feed")
        return n

    def feed_ptrs(self, ptrs, sizes, on_device=False):
        """feed() over raw addresses (page-locked host memory from host_alloc(), or device memory)."""
        p = (C.c_void_p * len(ptrs))(*ptrs)
        s = (C.c_size_t * len(sizes))(*sizes)
        n = self._f("feed")(self._h, p, s, 1 if on_device else 0)
        _need(n >= 0, "stream_feed")
        return n

    def feed_resident(self, base_ptrs, avail):
        """dabhip_stream_feed_resident: the streams live in device memory (base_ptrs[b] + x = byte x of stream b) and are read in place;
        avail[b] = bytes there now."""
        p = (C.c_void_p * len(base_ptrs))(*base_ptrs)
        a = (C.c_size_t * len(avail))(*avail)
        n = self._f("feed_resident")(self._h, p, a)
        _need(n >= 0, "stream_feed_resident")
        return n

    def need_from(self, stream):
        return int(self._f("need_from")(self._h, stream))

    def prefetch_ptrs(self, ptrs, sizes, on_device=False):
        """dabhip_stream_prefetch: start uploading the segment a later feed_ptrs() with the same arguments will consume."""
        p = (C.c_void_p * len(ptrs))(*ptrs)
        s = (C.c_size_t * len(sizes))(*sizes)
        _need(self._f("prefetch")(self._h, p, s, 1 if on_device else 0) == 0, "stream_prefetch")

    def status(self, stream):
        return int(self._f("status")(self._h, stream))

    def set_sync_speculation(self, mode):
        """see Engine.set_sync_speculation"""
        _need(self._f("set_sync_speculation")(self._h, int(mode)) == 0, "stream_set_sync_speculation")

    def stage_ms(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().dabhip_stream_stage_ms(self._h, names, ms, 16)
        return {names[i].decode(): ms[i] for i in range(n)}

    def eti_fetch(self, dst_ptr, cap_frames):
        """dabhip_stream_eti_fetch: all frames of the segment fed last on their way to (page-locked) host memory; returns their number."""
        n = self._f("eti_fetch")(self._h, dst_ptr, cap_frames)
        _need(n >= 0, "stream_eti_fetch")
        return n

    def eti_fetch_wait(self):
        _need(self._f("eti_fetch_wait")(self._h) == 0, "stream_eti_fetch_wait")

    def eti_count(self, stream):
        return int(self._f("eti_count")(self._h, stream))

    def eti(self, stream):
        n = self._f("eti_count")(self._h, stream)
        _need(n >= 0, "stream_eti_count")
        out = np.zeros((n, ETI_BYTES), dtype=np.uint8)
        if n:
            _need(self._f("eti_read")(self._h, stream, _p(out), n) == n, "stream_eti_read")
        return out

    def close(self):
        if self._h:
            self._f("destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiStream(Stream):
    """Sessions over several devices (dabhip_multi_stream_*): nstreams unbounded captures dealt ONCE to the listed devices in contiguous slices (the rule
    of Multi.plan), every slice a complete Stream session on its device; every call is made on all slices at once.  Frames per segment and in total are
    those of one Stream over all captures.  A device may be listed several times (every entry is its own slice)."""
    _PREFIX = "dabhip_multi_stream_"
    _RENAMED = {"status": "status_of", "log": "log_of"}

    def __init__(self, nstreams, devices, afc=False, soft=False, subchannels=None, guard=None):
        devs = list(devices)
        self._h = lib().dabhip_multi_stream_create((C.c_int * len(devs))(*devs), len(devs), nstreams)
        _need(self._h, "multi_stream_create")
        self.nstreams = nstreams
        self.devices = devs
        self._configure(afc, soft, subchannels, guard)

    def slice_of(self, stream):
        """(slice, device, first stream of the slice, streams in the slice)"""
        dev, first, count = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        sl = lib().dabhip_multi_stream_slice_of(self._h, stream, C.byref(dev), C.byref(first), C.byref(count))
        _need(sl >= 0, "multi_stream_slice_of")
        return sl, dev.value, first.value, count.value

    def stage_ms(self, slice_index=0):
        h = lib().dabhip_multi_stream_session(self._h, slice_index)
        _need(h, "multi_stream_session")
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().dabhip_stream_stage_ms(h, names, ms, 16)
        return {names[i].decode(): ms[i] for i in range(n)}
