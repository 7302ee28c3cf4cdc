#!/bin/bash
set -u
# round 5: the two new sweeps against THE REFERENCE ITSELF brought to 512 captures per class (checkpointed every round)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
NAME=r05_channel_ref_d ROUNDS=2 STREAMS=64 TFS=24 SEED=5909 LIMIT=1000 STRESS_ARGS="--channel --reference --workers 14" bash tools/gpu/stress.sh
NAME=r05_reconf_ref_d ROUNDS=2 STREAMS=64 TFS=24 SEED=6010 LIMIT=1800 STRESS_ARGS="--reconf --reference --workers 14" bash tools/gpu/stress.sh
