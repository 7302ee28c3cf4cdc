#!/bin/bash
set -u
# round 5, sixth pass: the decision audit of the fused kernel at full size (>= 1e10 decisions + the impaired channels), then the two new sweeps against THE
# REFERENCE ITSELF (real front end + back end), 512 captures per class in runs of 256 (a run that is cut off loses only itself)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
timeout 2400 python tools/decision_audit.py --channels 600 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
python - <<PY
import json
d = json.loads(open("$O/decision_audit.json").read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if k != "cases"})
PY
NAME=r05_channel_ref_a ROUNDS=4 STREAMS=64 TFS=24 SEED=5303 LIMIT=2400 STRESS_ARGS="--channel --reference --workers 14" bash tools/gpu/stress.sh
NAME=r05_reconf_ref_a ROUNDS=4 STREAMS=64 TFS=24 SEED=5404 LIMIT=2400 STRESS_ARGS="--reconf --reference --workers 14" bash tools/gpu/stress.sh
NAME=r05_channel_ref_b ROUNDS=4 STREAMS=64 TFS=24 SEED=5505 LIMIT=2400 STRESS_ARGS="--channel --reference --workers 14" bash tools/gpu/stress.sh
NAME=r05_reconf_ref_b ROUNDS=4 STREAMS=64 TFS=24 SEED=5606 LIMIT=2400 STRESS_ARGS="--reconf --reference --workers 14" bash tools/gpu/stress.sh
