#!/bin/bash
set -u
# round 5, first pass: the new channel / reconfiguration / call-length tests first (all of them, no -x), then the whole GPU suite, then two bench lines
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_channel.py -q -m gpu > $O/channel.log 2>&1; echo "channel tests rc=$?"; tail -n 40 $O/channel.log
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_channel.py > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 15 $O/gpu_tests.log
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-h2d > $O/bench_$i.json 2> $O/bench_$i.err || echo "bench rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(round(d["value"]), round(d["ms_per_step"], 3), {k: round(s[k], 3) for k in ("sync", "fft", "fic", "viterbi", "eti", "control", "host_worklist")})
PY
done
