#!/usr/bin/env python3
"""tools/hip_retained_commands.py -- which copy patterns let the HIP runtime's books grow without bound? (GPU box)

Heap in use of this process (glibc mallinfo2) before and after N repetitions of a copy pattern, straight on libamdhip64 (no libdabhip).  Measured on
ROCm 7.2 / MI355X (profiles/r06_hip_retained_commands.txt): a blocking hipMemcpy into PAGEABLE host memory keeps 970 bytes per call, an asynchronous copy
+ hipEventRecord + hipEventSynchronize on a stream of its own 2.0 KB per round -- for as long as nobody synchronises THAT STREAM (hipStreamQuery does
not count, nor does a hipDeviceSynchronize afterwards for the null stream's copies); with a hipStreamSynchronize every 32 rounds, or one per copy, nothing
grows.  What libdabhip does about it: engine.hpp (blocking_copy, kReapEvery).  Usage: hip_retained_commands.py [N=20000]"""
import ctypes as C, json, sys
class MI(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks", "uordblks", "fordblks", "keepcost")]
libc = C.CDLL("libc.so.6"); libc.mallinfo2.restype = MI
hip = C.CDLL("libamdhip64.so")
def used(): return libc.mallinfo2().uordblks // 1024
H2D, D2H, D2D = 1, 2, 3
n = 64 * 6144
d = C.c_void_p(); d2 = C.c_void_p(); pin = C.c_void_p()
assert hip.hipMalloc(C.byref(d), n) == 0 and hip.hipMalloc(C.byref(d2), n) == 0 and hip.hipHostMalloc(C.byref(pin), n, 0) == 0
page = (C.c_uint8 * n)()
st = C.c_void_p(); ev = C.c_void_p()
assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0 and hip.hipEventCreateWithFlags(C.byref(ev), 2) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
def run(name, body, every=None, reap=None):
    hip.hipDeviceSynchronize()
    a = used()
    for i in range(N):
        body()
        if every and i % every == every - 1: reap()
    b = used()
    hip.hipDeviceSynchronize()
    c = used()
    print(json.dumps({"pattern": name, "reps": N, "heap_kb_before": a, "after": b, "after_device_sync": c, "bytes_per_rep": round((b - a) * 1024 / N, 1)}), flush=True)
def async_ev(dst):
    hip.hipMemcpyAsync(dst, d, n, D2H, st); hip.hipEventRecord(ev, st); hip.hipEventSynchronize(ev)
run("hipMemcpy D2H pageable (null stream)", lambda: hip.hipMemcpy(page, d, n, D2H))
run("hipMemcpy D2H page-locked (null stream)", lambda: hip.hipMemcpy(pin, d, n, D2H))
run("hipMemcpy H2D pageable (null stream)", lambda: hip.hipMemcpy(d, page, n, H2D))
run("hipMemcpy D2D (null stream)", lambda: hip.hipMemcpy(d2, d, n, D2D))
run("hipMemcpy D2H pageable + hipStreamSynchronize(0) every 32", lambda: hip.hipMemcpy(page, d, n, D2H), 32, lambda: hip.hipStreamSynchronize(None))
run("hipMemcpy D2H pageable + hipDeviceSynchronize every 32", lambda: hip.hipMemcpy(page, d, n, D2H), 32, lambda: hip.hipDeviceSynchronize())
run("async D2H page-locked on a non-blocking stream + event record + event synchronize", lambda: async_ev(pin))
run("... + hipStreamSynchronize(stream) every 32", lambda: async_ev(pin), 32, lambda: hip.hipStreamSynchronize(st))
run("... + hipStreamQuery(stream) every time", lambda: async_ev(pin), 1, lambda: hip.hipStreamQuery(st))
run("async D2H pageable on the stream + hipStreamSynchronize(stream) each", lambda: (hip.hipMemcpyAsync(page, d, n, D2H, st), hip.hipStreamSynchronize(st)))
run("async D2H page-locked on the stream + hipStreamSynchronize(stream) each", lambda: (hip.hipMemcpyAsync(pin, d, n, D2H, st), hip.hipStreamSynchronize(st)))
