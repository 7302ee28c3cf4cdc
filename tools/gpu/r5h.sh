#!/bin/bash
set -u
# round 5: K1's look-ahead schedule again -- its tests, the small-batch curve (plain chain / default), the whole suite
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ahead.py -q -m gpu > $O/ahead.log 2>&1; echo "ahead tests rc=$?"; tail -n 30 $O/ahead.log | cut -c1-600
for mode in 0 default; do
  if [ $mode = default ]; then unset DABHIP_K1_SPEC; else export DABHIP_K1_SPEC=$mode; fi
  timeout 600 python tools/batch_curve.py --batches ${BATCHES:-1,2,4,8} --steps 20 --session-tfs 24 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/curve_$mode.json").read())
for r in d["curve"]:
    s = r["stage_ms"]
    print("$mode", "B", r["streams"], "ms", round(r["ms_per_decode"], 3), "sync", round(s["sync"], 3), "fft", round(s["fft"], 3), "fic", round(s["fic"], 3), "vit", round(s["viterbi"], 3), "spec", s.get("sync_spec_calls"))
print("$mode", "session", [(x["segment_tfs"], round(x["ms_per_segment_median"], 3)) for x in d.get("single_stream_session")])
PY
done
unset DABHIP_K1_SPEC
[ "${SUITE:-1}" = 1 ] && { timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 12 $O/gpu_tests.log | cut -c1-600; }
