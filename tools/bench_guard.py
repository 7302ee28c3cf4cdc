#!/usr/bin/env python3
"""Cost of the parity guard by level (0 = off, 1 = measured band, 2 = proven band; dabhip.h: DABHIP_GUARD_*): the benchmark batch decoded at each level,
clean and at 9 / 7 / 5 dB -- stage times, frames/s, decisions re-decided in fp64, and whether the ETI bytes of levels 1 and 2 are the same (they must be:
both are the bits of exact arithmetic wherever the narrower band suffices).  One JSON record per line, a summary object last."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import dabtools_amd as dab  # noqa: E402

nstreams, ntf = int(os.environ.get("STREAMS", 256)), int(os.environ.get("TFS", 64))
reps = int(os.environ.get("REPS", 5))
snrs = [float(s) for s in os.environ.get("SNRS", "1000,9,7,5").split(",")]
fused_modes = [True] + ([False] if os.environ.get("TWO_KERNEL", "0") == "1" else [])
out, summary = [], {}
for snr in snrs:
    cfgs = [dab.synth_preset(0, seed=2000 + i, cif_count0=(97 * i) % 5000, snr_db=snr) for i in range(nstreams)]
    bufs = [torch.empty(dab.synth_bytes(c, ntf), dtype=torch.uint8, device="cuda") for c in cfgs]
    dab.synth_generate_device(cfgs, ntf, [b.data_ptr() for b in bufs])
    torch.cuda.synchronize()
    ptrs, sizes = [b.data_ptr() for b in bufs], [b.numel() for b in bufs]
    eng = dab.Engine(0)
    for fused in fused_modes:
        digests = {}
        for level in (0, 1, 2, 0, 1, 2):
            eng.set_fused(fused)
            eng.set_parity_guard(level)
            eng.decode_device(ptrs, sizes)
            acc = {}
            t0 = time.perf_counter()
            for _ in range(reps):
                n = eng.decode_device(ptrs, sizes)
                for k, v in eng.stage_ms().items():
                    acc[k] = acc.get(k, 0) + v / reps
            dt = (time.perf_counter() - t0) / reps
            h = hashlib.sha256()
            for b in range(min(nstreams, 32)):
                h.update(eng.eti(b).tobytes())
            digests[level] = h.hexdigest()
            rec = {"snr": snr, "fused": fused, "guard_level": level, "ms": round(1e3 * dt, 4), "frames": n, "frames_per_s": round(n / dt),
                   "redecided": eng.guard_stats()[0], "decisions": eng.guard_stats()[1], "overflows": eng.guard_overflows(),
                   "eti_sha256_32_streams": digests[level][:16], **{k: round(acc[k], 3) for k in ("sync", "fft", "demap", "fic", "viterbi", "wall")}}
            out.append(rec)
            print(json.dumps(rec), flush=True)
        key = "%s_%s" % ("clean" if snr > 100 else "%gdB" % snr, "fused" if fused else "two_kernel")
        best = {lv: min(r["ms"] for r in out if r["snr"] == snr and r["fused"] == fused and r["guard_level"] == lv) for lv in (0, 1, 2)}
        flag = {lv: max(r["redecided"] for r in out if r["snr"] == snr and r["fused"] == fused and r["guard_level"] == lv) for lv in (0, 1, 2)}
        summary[key] = {"ms_best_of_two": best, "redecided_per_step": flag, "proven_over_measured": round(best[2] / best[1], 4),
                        "proven_over_off": round(best[2] / best[0], 4), "eti_level1_equals_level2": digests[1] == digests[2],
                        "eti_level0_equals_level2": digests[0] == digests[2]}
    eng.close()
    del bufs
print(json.dumps({"summary": summary, "streams": nstreams, "tf": ntf, "constants": {"measured": dab.guard_constants(1), "proven": dab.guard_constants(2)},
                  "default_level": dab.guard_default_level()}))
