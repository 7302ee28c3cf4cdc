#!/bin/bash
# host-side phase times (DABHIP_TRACE_HOST=1) over many steps: median / p90 / max of every mark, the slowest steps in full, and the box's load
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/hosttrace
uptime
DABHIP_TRACE_HOST=1 python3 bench.py --no-cpu-baseline --no-h2d --no-variants --steps ${STEPS:-60} --warmup 1 > gpurun_out/hosttrace/bench.json 2> gpurun_out/hosttrace/trace.txt
python3 - <<'PY'
import re, json, statistics as st
steps, cur = [], {}
for l in open("gpurun_out/hosttrace/trace.txt"):
    m = re.match(r"\[host\] (.*?)\s+([0-9.]+) ms", l)
    if not m:
        continue
    k, v = m.group(1).strip(), float(m.group(2))
    cur[k] = v
    if k == "return":
        steps.append(cur)
        cur = {}
steps = steps[3:]
keys = ["scan done", "ofdm queued", "fibs on host", "control plane done", "work lists built", "all queued", "stream drained", "return"]
print("steps", len(steps))
for k in keys:
    v = sorted(s.get(k, 0) for s in steps)
    print("%-20s med %.3f  p90 %.3f  max %.3f" % (k, st.median(v), v[int(0.9 * len(v))], v[-1]))
for s in sorted(steps, key=lambda s: -s["return"])[:4]:
    print({k: s.get(k) for k in keys})
d = json.loads(open("gpurun_out/hosttrace/bench.json").read().strip().splitlines()[-1])
print(round(d["value"]), round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["stage_ms_per_step"].items()})
PY
uptime
