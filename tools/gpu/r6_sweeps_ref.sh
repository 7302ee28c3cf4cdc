#!/bin/bash
set -u
# Round 6: more captures against the REFERENCE itself (real front end over hipFFTW + real back end) under the new default: mid-stream reconfigurations, and a
# second pair of rounds of the harsh mix and the impaired channels with other seeds.  128 captures per run (the reference's front end is slow on the box's CPUs).
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
export LIMIT=${LIMIT:-1150} ROUNDS=2
NAME=r06_reconf_vs_reference STRESS_ARGS="--reference --reconf" SEED=60621 bash tools/gpu/stress.sh
NAME=r06_harsh_vs_reference_b STRESS_ARGS="--reference --harsh" SEED=60622 bash tools/gpu/stress.sh
NAME=r06_channel_vs_reference_b STRESS_ARGS="--reference --channel" SEED=60623 bash tools/gpu/stress.sh
