#!/bin/bash
set -u
# GPU test-suite + two short bench lines (no CPU baseline, no host-fed leg); optional: the cross-lane microbenchmarks.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/quick; mkdir -p $O
timeout 1500 python -m pytest tests -q -x -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 5 $O/gpu_tests.log
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-h2d ${BENCH_EXTRA:-} > $O/bench_$i.json 2> $O/bench_$i.err || echo "bench rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(round(d["value"]), round(d["ms_per_step"], 3), {k: round(s[k], 3) for k in ("sync", "fft", "fic", "viterbi", "eti", "control", "host_worklist")}, "plain", d.get("parity_guard_off_variant", {}).get("stage_ms_per_step"))
PY
done
