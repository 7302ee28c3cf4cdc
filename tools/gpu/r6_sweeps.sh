#!/bin/bash
set -u
# Round 6: the randomised end-to-end sweeps under the NEW default (proven guard level listing per bin, rewritten re-decision): harsh mix and impaired channels
# against the REFERENCE itself (its front end over hipFFTW is slow on the box's CPUs: 128 captures each in the time allowed) and, eight rounds each, harsh mix,
# impaired channels and mid-stream reconfigurations against the oracle (512 captures each).  Records under gpurun_out/stress/.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
export LIMIT=${LIMIT:-1100}
ROUNDS=8 NAME=r06_harsh_vs_oracle STRESS_ARGS="--harsh" SEED=60611 bash tools/gpu/stress.sh
ROUNDS=8 NAME=r06_channel_vs_oracle STRESS_ARGS="--channel" SEED=60612 bash tools/gpu/stress.sh
ROUNDS=8 NAME=r06_reconf_vs_oracle STRESS_ARGS="--reconf" SEED=60613 bash tools/gpu/stress.sh
ROUNDS=2 NAME=r06_harsh_vs_reference STRESS_ARGS="--reference --harsh" SEED=60601 bash tools/gpu/stress.sh
ROUNDS=2 NAME=r06_channel_vs_reference STRESS_ARGS="--reference --channel" SEED=60602 bash tools/gpu/stress.sh
timeout 300 python tools/live_latency.py > gpurun_out/stress/live_latency.json 2> gpurun_out/stress/live_latency.err; echo "latency rc=$?"; cat gpurun_out/stress/live_latency.json
