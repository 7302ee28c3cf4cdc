#!/bin/bash
set -u
# round 5, seventh pass: K1's look-ahead schedule -- its own tests (all of them), the small-batch curve with the schedule off / forced on / default, then the
# whole suite under the default
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ahead.py -q -m gpu > $O/ahead.log 2>&1; echo "ahead tests rc=$?"; tail -n 40 $O/ahead.log | cut -c1-400
for mode in 0 1 default; do
  if [ $mode = default ]; then unset DABHIP_K1_SPEC; else export DABHIP_K1_SPEC=$mode; fi
  timeout 600 python tools/batch_curve.py --batches 1,2,4,8,16,32,64 --steps 20 --session-tfs 24 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/curve_$mode.json").read().strip().splitlines()[-1])
for r in d["curve"]:
    s = r["stage_ms"]
    print("$mode", "B", r["streams"], "ms", round(r["ms_per_decode"], 3), "sync", round(s["sync"], 3), "fft", round(s["fft"], 3), "fic", round(s["fic"], 3), "vit", round(s["viterbi"], 3), "spec", s.get("sync_spec_calls"))
print("$mode", "session", json.dumps(d.get("single_stream_session"))[:600])
PY
done
unset DABHIP_K1_SPEC
timeout 1800 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 12 $O/gpu_tests.log | cut -c1-400
timeout 1200 python tools/decision_audit.py --channels 300 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
python - <<PY
import json
d = json.loads(open("$O/decision_audit.json").read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k != "cases"})
PY
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-h2d > $O/bench_$i.json 2> $O/bench_$i.err || echo "bench rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
s = d["stage_ms_per_step"]
print(round(d["value"]), round(d["ms_per_step"], 3), {k: round(s[k], 3) for k in ("sync", "fft", "fic", "viterbi", "eti", "control", "host_worklist")}, "single", d["single_ensemble"]["ms_per_decode"], d["single_ensemble"]["stage_ms"])
PY
done
