#!/bin/bash
set -u
# round 5: scan results and FIBs of small decodes downloaded by one kernel each: the small-batch curve, the timeline of a one-ensemble decode, the whole suite
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5p; mkdir -p $O
timeout 600 python tools/batch_curve.py --steps 20 > $O/curve.json 2> $O/curve.err; echo "curve rc=$?"; tail -n 10 $O/curve.err
bash tools/gpu/timeline_b1.sh 2>&1 | sed -n '/== default/,/== 0/p' | cut -c1-120 | head -40
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 5 $O/gpu_tests.log | cut -c1-300
timeout 600 python bench.py --steps 20 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
h = d["h2d_inclusive"]
print(round(d["value"]), round(d["ms_per_step"], 3), "h2d", h.get("error"), h.get("value"), "single", d["single_ensemble"]["ms_per_decode"], d["single_ensemble"]["live_session"]["ms_per_segment_median"])
PY
