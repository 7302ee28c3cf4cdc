#!/bin/bash
set -u
# one engine with 512 and 1024 streams x 64 TF (two and four times the benchmark batch): does everything still index correctly, and at what rate
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/bigbatch
for n in 512 1024; do
  timeout 900 python bench.py --streams $n --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-h2d > gpurun_out/bigbatch/b$n.json 2> gpurun_out/bigbatch/b$n.err; echo "streams $n rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/bigbatch/b$n.json").read().strip().splitlines()[-1])
    print(round(d["value"]), round(d["ms_per_step"], 2), d["config"].get("eti_frames_per_step"), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items() if k in ("sync", "fft", "viterbi", "gather", "eti")})
except Exception as e:
    print("no result:", e); print(open("gpurun_out/bigbatch/b$n.err").read()[-1500:])
PY
done
