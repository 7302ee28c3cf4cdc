#!/usr/bin/env python3
"""The 1 -> 8-GPU curve in one table (north_star: "reported at 1/2/4/8 GPUs"; BASELINE configs[3] = 2048 streams as 256 per GPU x 8).

Runs `bench.py --gpus N` for every N of --gpus (default 1,2,4,8) as a FRESH child process each -- bench.py itself starts one rank process per GPU before
anything touches a GPU, ensembles are independent, no collective -- and prints per N: ETI frames/s, x real time, ms per step, efficiency against N x the
N = 1 value of this very sweep, the PCI bus ids the ranks decoded on, and next to the N = 1 value the last driver-measured one (BENCH_rNN.json).

What makes the table self-checking:
  * bench.py refuses to print a line whose N ranks did not sit on N distinct devices (ranks[].device.pci_bus_id) unless DABHIP_BENCH_ONE_DEVICE=1 declares
    a one-GPU rehearsal; this tool repeats the check on the lines it collects and labels every row `measured` or `REHEARSAL (one GPU)`;
  * a sweep with any rehearsal row says so in its summary and reports NO scaling efficiency for it as a result (sharing one GPU is not scaling).
Options: --one-device (sets DABHIP_BENCH_ONE_DEVICE=1: the rehearsal this pool's one-GPU boxes allow), --dry-run (no GPU: bench.py's dry run; CPU suite),
--steps / --warmup / --streams / --tfs are passed through.  Output: one JSON object (last line of stdout) + the table on stderr."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_driver_value():
    best = None
    for path in glob.glob(os.path.join(ROOT, "BENCH_r*.json")):
        m = re.search(r"BENCH_r(\d+)\.json$", path)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        parsed = d.get("parsed") or {}
        if m and isinstance(parsed.get("value"), (int, float)) and parsed.get("n_gpus") == 1:
            if best is None or int(m.group(1)) > best[0]:
                best = (int(m.group(1)), parsed["value"], os.path.basename(path))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--tfs", type=int, default=64)
    ap.add_argument("--one-device", action="store_true")
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--timeout", type=int, default=1800)
    args = ap.parse_args()
    counts = [int(x) for x in args.gpus.split(",")]
    env = dict(os.environ)
    if args.one_device:
        env["DABHIP_BENCH_ONE_DEVICE"] = "1"
    rows, errors = [], []
    for n in counts:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--streams", str(args.streams), "--tfs", str(args.tfs), "--no-cpu-baseline", "--no-h2d", "--no-variants"]
        if args.dry_run:
            cmd.append("--dry-run")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=args.timeout)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            errors.append({"n_gpus": n, "returncode": r.returncode, "stderr_tail": r.stderr[-600:]})
            continue
        d = json.loads(line)
        dev = d.get("devices", {})
        ids = dev.get("pci_bus_ids") or [rk.get("device", {}).get("pci_bus_id") for rk in d.get("ranks", [])]
        distinct = len(set(ids))
        rows.append({"n_gpus": n, "value": d["value"], "x_realtime": d["x_realtime"], "ms_per_step": d["ms_per_step"], "eti_frames_per_step": d["config"]["eti_frames_per_step"],
                     "ranks": len(d.get("ranks", [])), "distinct_devices": distinct, "pci_bus_ids": ids,
                     "kind": "measured" if distinct == n else "REHEARSAL (one GPU)" if distinct == 1 else "INVALID (%d ranks on %d devices)" % (n, distinct)})
    base = next((r for r in rows if r["n_gpus"] == 1), None)
    for r in rows:
        # efficiency is a statement about N devices: none for a row whose ranks shared a device
        r["efficiency_vs_n1"] = (r["value"] / (r["n_gpus"] * base["value"])) if base and r["kind"] == "measured" else None
        r["ratio_to_n1_value"] = r["value"] / base["value"] if base else None
    drv = last_driver_value()
    out = {"what": "bench.py --gpus N as fresh children, N in %s" % counts, "rows": rows, "errors": errors, "dry_run": args.dry_run,
           "any_rehearsal": any(r["kind"] != "measured" for r in rows),
           "n1_value": base["value"] if base else None,
           "last_driver_bench": {"round": drv[0], "value": drv[1], "file": drv[2], "n1_over_driver": (base["value"] / drv[1]) if base and not args.dry_run else None} if drv else None}
    w = sys.stderr.write
    w("%5s %14s %11s %9s %10s %9s  %s\n" % ("N", "ETI frames/s", "x realtime", "ms/step", "x N=1", "eff.", "devices"))
    for r in rows:
        w("%5d %14.0f %11.0f %9.3f %10.3f %9s  %d distinct: %s\n" % (r["n_gpus"], r["value"], r["x_realtime"], r["ms_per_step"], r["ratio_to_n1_value"] or 0,
                                                                     "%.3f" % r["efficiency_vs_n1"] if r["efficiency_vs_n1"] is not None else "-", r["distinct_devices"], r["kind"]))
    if out["any_rehearsal"]:
        w("NOTE: rows marked REHEARSAL shared ONE GPU between their ranks: they exercise the N-rank path, they are not an N-GPU figure; no efficiency is reported for them.\n")
    if drv and base and not args.dry_run:
        w("N = 1 here: %.0f; last driver-measured (%s): %.0f; ratio %.3f\n" % (base["value"], drv[2], drv[1], base["value"] / drv[1]))
    for e in errors:
        w("N = %d FAILED (rc %d): %s\n" % (e["n_gpus"], e["returncode"], e["stderr_tail"].strip().splitlines()[-1] if e["stderr_tail"].strip() else ""))
    print(json.dumps(out))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
