#!/bin/bash
set -u
# Round 6: the randomised end-to-end sweeps under the NEW default (proven guard level, rewritten re-decision): harsh mix and impaired channels against the
# REFERENCE itself, reconfigurations against the oracle.  256 captures each (ROUNDS=4 x 64); records under gpurun_out/stress/.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
export ROUNDS=${ROUNDS:-4} LIMIT=${LIMIT:-1000}
NAME=r06_harsh_vs_reference STRESS_ARGS="--reference --harsh" SEED=60601 bash tools/gpu/stress.sh
NAME=r06_channel_vs_reference STRESS_ARGS="--reference --channel" SEED=60602 bash tools/gpu/stress.sh
NAME=r06_reconf_vs_oracle STRESS_ARGS="--reconf" SEED=60603 bash tools/gpu/stress.sh
