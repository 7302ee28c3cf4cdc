#!/bin/bash
set -u
# round 5: a short cut of the two new sweeps against the oracle and against the reference (timing the latter), then K1 A/B
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
NAME=channel_oracle_cut ROUNDS=2 STREAMS=48 TFS=24 SEED=51 STRESS_ARGS="--channel" bash tools/gpu/stress.sh
NAME=reconf_oracle_cut ROUNDS=2 STREAMS=48 TFS=24 SEED=52 STRESS_ARGS="--reconf" bash tools/gpu/stress.sh
NAME=channel_ref_cut ROUNDS=1 STREAMS=48 TFS=24 SEED=53 STRESS_ARGS="--channel --reference --workers 12" bash tools/gpu/stress.sh
for rep in 1 2; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base notail; done
