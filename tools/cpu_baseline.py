#!/usr/bin/env python3
"""CPU baselines for bench.py, run as a child process that never touches the GPU.

Checker code used as the *reported baseline* only (kind "port" = oracle/ restatement, "reference" = oracle/_ref, the
reference's own objects).  Input: cu8 captures of the benchmark workload written by bench.py (--iq files).

  1. front end + back end, scalar viterbi.c semantics: oracle/or_replay on the captures, ONE core   -> frames/s/core
  2. the REAL reference back end (dab_process_frame: fic_decode + create_eti with scalar viterbi.c, and the
     ENABLE_SPIRAL_VITERBI SSE2 build) on the demapped frames of those captures:
       one core, and one process per core on k cores at once (the reference's globals are not thread-safe)
  3. micro-timings: viterbi() Mbit/s on 4608-bit code words; the oracle's fp64 DFT in microseconds per 2048 points
     (libfftw3 is absent, so the reference front end cannot be built or timed)
Prints one JSON object.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402

_TFS = None      # demapped frames, inherited by the forked workers


def _silence_stderr():
    devnull, saved = os.open(os.devnull, os.O_WRONLY), os.dup(2)
    os.dup2(devnull, 2)                 # the reference prints its ensemble table to stderr
    return devnull, saved


def _restore_stderr(devnull, saved):
    os.dup2(saved, 2)
    os.close(devnull)
    os.close(saved)


def _backend_pass(sse, reps):
    """dab_process_frame of the real reference objects over the shared demapped frames -> (ETI frames, seconds)"""
    R = oracle_lib.ref(sse=sse)
    n, dt = 0, 0.0
    tok = _silence_stderr()
    try:
        for _ in range(reps):
            for tfs in _TFS:
                H = R.refh_new()
                t0 = time.perf_counter()
                for f, m in tfs:
                    C.memmove(R.refh_tf_fic(H), oracle_lib._ptr(f), f.size)
                    C.memmove(R.refh_tf_msc(H), oracle_lib._ptr(m), m.size)
                    R.refh_process(H)
                dt += time.perf_counter() - t0
                n += R.refh_neti(H)
    finally:
        _restore_stderr(*tok)
    return n, dt


def _backend_payload(sse, args):
    """ETI frames of the real reference back end over every capture's demapped frames -> payload statistics vs the modulator."""
    sys.path.insert(0, ROOT)
    import dabtools_amd as dab          # host-only entry points (dabhip_synth_*): no GPU call
    from dabtools_amd import payload
    R = oracle_lib.ref(sse=sse)
    chk = payload.PayloadCheck()
    tok = _silence_stderr()
    try:
        for i, tfs in enumerate(_TFS):
            H = R.refh_new()
            for f, m in tfs:
                C.memmove(R.refh_tf_fic(H), oracle_lib._ptr(f), f.size)
                C.memmove(R.refh_tf_msc(H), oracle_lib._ptr(m), m.size)
                R.refh_process(H)
            n = R.refh_neti(H)
            eti = np.ctypeslib.as_array(R.refh_eti(H), (max(n, 1), 6144))[:n].copy()
            chk.add_stream(dab, payload.bench_cfg(dab, args.ber_first_stream + i, args.snr), args.tfs, eti)
    finally:
        _restore_stderr(*tok)
    return chk.result()


def _noop(_):
    time.sleep(0.01)


_CAPS = None     # the captures, inherited by the forked workers of the whole-path run


def _port_worker(args):
    """or_replay (the whole path, CPU restatement) of the shared captures in one process of k running at once"""
    which, barrier_at = args
    while time.time() < barrier_at:
        time.sleep(0.001)
    t0 = time.time()
    n = 0
    for iq in _CAPS[which % len(_CAPS):][:1] + _CAPS[:1]:
        n += len(oracle_lib.or_replay(iq)[0])
    return n, time.time() - t0, t0, time.time()


def _worker(args):
    sse, reps, barrier_at = args
    while time.time() < barrier_at:      # all workers start together: they compete for the cores like real processes would
        time.sleep(0.001)
    t0 = time.time()
    n, dt = _backend_pass(sse, reps)
    return n, dt, t0, time.time()


def main():
    global _TFS
    ap = argparse.ArgumentParser()
    ap.add_argument("--iq", nargs="+", required=True)
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--cores", type=int, default=0, help="processes for the per-core runs (0: every CPU this process may run on)")
    ap.add_argument("--backend-streams", type=int, default=2)
    ap.add_argument("--ber-first-stream", type=int, default=-1, help="global index of the first capture: also report the reference back ends' payload BER "
                    "(hard decisions of the oracle front end on these very captures) against what bench.py's modulator sent")
    ap.add_argument("--snr", type=float, default=1000.0)
    args = ap.parse_args()
    ncores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = ncores
    # BASELINE.md section 3 asks for "x cores": every CPU this process is ALLOWED to use -- its affinity mask, capped by the container's CFS quota
    # (cgroup cpu.max: on the GPU boxes of this pool 16 CPUs' worth of time on a 256-thread host; 256 processes would only take turns on them)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, -(-int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = max(1, -(-q // per))
        except (OSError, ValueError):
            pass
    k = args.cores or min(usable, quota or usable)
    caps = [np.fromfile(p, dtype=np.uint8)[: args.tfs * 393216] for p in args.iq]

    # 1. whole path, CPU restatement, one core
    neti, t0 = 0, time.perf_counter()
    for iq in caps:
        neti += len(oracle_lib.or_replay(iq)[0])
    dt = time.perf_counter() - t0
    out = {
        "value": neti / dt, "unit": "ETI frames/s", "cores": 1, "kind": "port",
        "sample": "oracle/or_replay (whole path IQ -> ETI, scalar viterbi.c semantics, own fp64 DFT: libfftw3 absent) on %d streams x %d TF "
                  "of the same workload: %d ETI frames in %.2f s on 1 of %d host cores" % (len(caps), args.tfs, neti, dt, ncores),
        "host_cores": ncores, "usable_cpus": usable, "cfs_quota_cpus": quota,
    }

    # the same on every core at once: one process per core, two captures each (the oracle is re-entrant, but the reference it stands for is not:
    # dab2eti is one stream per process)
    global _CAPS
    _CAPS = caps
    if k > 1:
        ctx = mp.get_context("fork")
        with ctx.Pool(k) as pool:
            pool.map(_noop, range(4 * k))
            start = time.time() + 0.5
            res = pool.map(_port_worker, [(i, start) for i in range(k)], chunksize=1)
        rate = sum(r[0] / r[1] for r in res)
        overlap = min(r[3] for r in res) - max(r[2] for r in res)
        out["all_cores"] = {"value": rate, "unit": "ETI frames/s", "cores": k, "per_core": rate / k, "kind": "port",
                            "sample": "%d processes at once (one per CPU this container may use: affinity %d, CFS quota %s, host %d), each oracle/or_replay of two %d-TF captures of the workload: sum of the per-process "
                                      "rates; all ran concurrently for %.2f s of the slowest one's %.2f s" % (k, usable, quota, ncores, args.tfs, max(overlap, 0.0), max(r[1] for r in res))}

    # demapped frames of a few captures (untimed; oracle front end), shared with the forked workers
    O = oracle_lib.oracle()
    fic = np.zeros(9216, np.uint8)
    msc = np.zeros(221184, np.uint8)
    _TFS = []
    nback = len(caps) if args.ber_first_stream >= 0 else args.backend_streams
    for iq in caps[:nback]:
        S, tfs = O.or_sdr_new(), []
        for off in range(0, iq.size - 262144 + 1, 262144):
            if O.or_sdr_demod(S, oracle_lib._ptr(iq[off:off + 262144]), 262144, oracle_lib._ptr(fic), oracle_lib._ptr(msc)):
                tfs.append((fic.copy(), msc.copy()))
        O.or_sdr_free(S)
        _TFS.append(tfs)
    ntfs = sum(len(t) for t in _TFS)

    # 3a. DFT micro-timing (the oracle's mixed-radix fp64 DFT; FFTW would be several times faster)
    x = np.random.default_rng(0).standard_normal((2048, 2))
    y = np.zeros((2048, 2))
    O.or_dft(2048, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), -1)
    reps = 2000
    t0 = time.perf_counter()
    for _ in range(reps):
        O.or_dft(2048, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double)), -1)
    out["oracle_dft2048_us"] = 1e6 * (time.perf_counter() - t0) / reps

    for key, sse in (("reference_backend_scalar", False), ("reference_backend_sse", True)):
        if oracle_lib.ref(sse=sse) is None:
            continue
        if args.ber_first_stream >= 0:
            # BASELINE configs[4]: the CPU back end's own decoding quality on the same noisy input (its hard decisions; the SSE decoder's
            # tie rule and metrics differ from the scalar one, viterbi_spiral_sse16.c:130-133, depuncture.c:36-43)
            rec_ber = _backend_payload(sse, args)
        reps = 4 if sse else 1
        n, dt = _backend_pass(sse, reps)
        rec = {"value": n / dt, "unit": "ETI frames/s", "cores": 1, "kind": "reference",
               "sample": "real reference dab_process_frame (%s) on %d demapped TF x %d: %d ETI frames in %.2f s; back end only (its front end "
                         "needs libfftw3)" % ("viterbi_spiral SSE2" if sse else "scalar viterbi.c", ntfs, reps, n, dt)}
        # one process per core on k cores at once
        if k > 1:
            ctx = mp.get_context("fork")
            preps = reps * (3 if sse else 2)                     # ~2 s per process: start-up skew stays small against it
            with ctx.Pool(k) as pool:
                pool.map(_noop, range(4 * k))                    # all workers forked and idle before the start time is set
                start = time.time() + 0.5
                res = pool.map(_worker, [(sse, preps, start)] * k, chunksize=1)
            frames = sum(r[0] for r in res)
            overlap = min(r[3] for r in res) - max(r[2] for r in res)      # time during which ALL k processes were running
            rate = sum(r[0] / r[1] for r in res)                            # each process's own frames / own decode seconds
            rec["all_cores"] = {"value": rate, "unit": "ETI frames/s", "cores": k, "per_core": rate / k,
                                "sample": "%d processes at once (one per core, %d of %d host cores), each the same pass x %d: %d ETI frames; "
                                          "sum of the per-process rates; all %d ran concurrently for %.2f s of the slowest one's %.2f s"
                                          % (k, k, ncores, preps // reps, frames, k, max(overlap, 0.0), max(r[3] - r[2] for r in res))}
        # 3b. the decoder alone: data Mbit/s on 4608-bit code words (192 kbit/s sub-channel)
        R = oracle_lib.ref(sse=sse)
        nbits, vreps = 4608, (40 if sse else 8)
        rng = np.random.default_rng(1)
        sym = np.where(rng.integers(0, 2, 4 * (nbits + 6)) > 0, 255 if sse else 129, 0 if sse else 127).astype(np.uint8)
        data = np.zeros(nbits // 8 + 8, np.uint8)
        H = R.refh_new()
        t0 = time.perf_counter()
        for _ in range(vreps):
            R.refh_viterbi(H, oracle_lib._ptr(sym), oracle_lib._ptr(data), nbits)
        rec["viterbi_mbit_s"] = vreps * nbits / (time.perf_counter() - t0) / 1e6
        if args.ber_first_stream >= 0:
            rec["payload"] = rec_ber
        out[key] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
