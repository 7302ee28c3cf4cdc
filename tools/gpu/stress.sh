#!/bin/bash
set -u
# the randomised end-to-end parity sweep against the oracle (tools/stress_parity.py) at the size kept under profiles/
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/stress
PYTHONPATH=tools timeout 3000 python tools/stress_parity.py --rounds ${ROUNDS:-16} --streams 64 --tfs 28 --seed ${SEED:-40404} ${STRESS_ARGS:-} > gpurun_out/stress/stress_parity.json 2> gpurun_out/stress/err.txt; echo "rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/stress/stress_parity.json").read().strip().splitlines()[-1])
print({k: (v if not isinstance(v, list) else len(v)) for k, v in d.items()})
PY
tail -3 gpurun_out/stress/err.txt
