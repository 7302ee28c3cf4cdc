#!/bin/bash
set -u
# the randomised end-to-end parity sweep (tools/stress_parity.py) at the size kept under profiles/: NAME names the output, STRESS_ARGS the mix and the checker
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
N=${NAME:-stress_parity}
mkdir -p gpurun_out/stress
PYTHONPATH=tools timeout ${LIMIT:-3000} python tools/stress_parity.py --rounds ${ROUNDS:-16} --streams ${STREAMS:-64} --tfs ${TFS:-28} --seed ${SEED:-40404} --checkpoint gpurun_out/stress/$N.partial.json ${STRESS_ARGS:-} > gpurun_out/stress/$N.json 2> gpurun_out/stress/$N.err; echo "rc=$?"
# a run that was cut off leaves the rounds it finished ("rounds_finished" in the record)
[ -s gpurun_out/stress/$N.json ] || cp gpurun_out/stress/$N.partial.json gpurun_out/stress/$N.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/stress/$N.json").read().strip().splitlines()[-1])
print({k: (v if not isinstance(v, list) else len(v)) for k, v in d.items()})
c = d["cases"]
print("captures", len(c), "with short reads on most calls", sum(1 for x in c if x["short_reads"] > 0.8 * x["calls"]), "with >= 2 multiplexes in their ETI", sum(1 for x in c if x["layouts_in_eti"] >= 2),
      "without any frame", sum(1 for x in c if x["eti_frames"] == 0))
PY
tail -3 gpurun_out/stress/$N.err
