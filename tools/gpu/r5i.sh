#!/bin/bash
set -u
# round 5: the rest of the two new sweeps against THE REFERENCE ITSELF (the record is checkpointed every round: a cut-off run keeps what it finished)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
NAME=r05_reconf_ref_c ROUNDS=${RECONF_ROUNDS:-3} STREAMS=64 TFS=24 SEED=5707 LIMIT=2700 STRESS_ARGS="--reconf --reference --workers 14" bash tools/gpu/stress.sh
NAME=r05_channel_ref_c ROUNDS=${CHANNEL_ROUNDS:-3} STREAMS=64 TFS=24 SEED=5808 LIMIT=900 STRESS_ARGS="--channel --reference --workers 14" bash tools/gpu/stress.sh
