#!/bin/bash
set -u
# The whole GPU suite, then the decision audit at both guard levels (tools/decision_audit.py --channels 300 -> gpurun_out/audit/decision_audit.json:
# what profiles/rNN_decision_audit.json is made of; tests/test_bench_launch.py holds its source hash against the tree).
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/audit; mkdir -p $O
if [ "${SKIP_SUITE:-0}" != "1" ]; then
  timeout 3000 python -m pytest tests -q -m gpu ${PYTEST_EXTRA:-} > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 15 $O/gpu_tests.log
fi
timeout 1500 python tools/decision_audit.py --channels 300 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/audit/decision_audit.json"))
print({k: v for k, v in d.items() if k != "cases"})
PY
