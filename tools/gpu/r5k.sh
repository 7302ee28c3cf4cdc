#!/bin/bash
set -u
# round 5: after the tail-byte update of the look-up chain was deferred and bench.py's host-fed session leg kept to two outstanding fetches:
# the look-ahead tests, the whole suite, the default bench line (the tracked one), the small-batch curve
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ahead.py -q -m gpu > $O/ahead.log 2>&1; echo "ahead tests rc=$?"; tail -n 5 $O/ahead.log | cut -c1-400
timeout 600 python tools/batch_curve.py --batches 1,2,4 --steps 20 > $O/curve.json 2> $O/curve.err; echo "curve rc=$?"; tail -n 4 $O/curve.err
timeout 900 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
h = d["h2d_inclusive"]
print(round(d["value"]), round(d["ms_per_step"], 3), "h2d", h.get("error"), h.get("value"), (h.get("one_shot") or {}).get("value"), (h.get("cli") or {}).get("value"))
print("single", d["single_ensemble"]["ms_per_decode"], d["single_ensemble"]["stage_ms"]["sync"], d["single_ensemble"]["live_session"]["ms_per_segment_median"])
PY
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 $O/gpu_tests.log | cut -c1-400
