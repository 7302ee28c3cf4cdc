#!/bin/bash
set -u
# round 5: the harsh mix (5 dB ... clean, off tune, weak / clipped) and the impaired-channel mix against the oracle once more, now with the per-capture decision audit
# of the fused OFDM kernel in the record (disagree_outside_band: raw fp32 decisions that differ from fp64 OUTSIDE the band the parity guard re-decides -- must be 0)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
NAME=r05_harsh_oracle_audit ROUNDS=4 STREAMS=64 TFS=24 SEED=6111 LIMIT=900 STRESS_ARGS="--harsh --audit-tfs 4" bash tools/gpu/stress.sh
NAME=r05_channel_oracle_audit ROUNDS=4 STREAMS=64 TFS=24 SEED=6212 LIMIT=900 STRESS_ARGS="--channel --audit-tfs 4" bash tools/gpu/stress.sh
python - <<'PY'
import json
for n in ("r05_harsh_oracle_audit", "r05_channel_oracle_audit"):
    d = json.loads(open("gpurun_out/stress/%s.json" % n).read().strip().splitlines()[-1])
    print(n, len(d["cases"]), d["eti_frames_compared"], d["differences"], d["decision_audit_of_the_fused_ofdm_kernel"])
PY
