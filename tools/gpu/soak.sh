#!/bin/bash
set -u
# A session left running (tools/soak.py): 4 looped captures in live-sized feeds past 2^32 bytes per stream, frames against the oracle and against the round
# before, resident set and device memory sampled throughout; then the same, shorter, through a session over two slices of the one GPU (dabhip_multi_stream).
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/soak
timeout ${LIMIT:-1500} python3 tools/soak.py --total-tf ${TOTAL_TF:-11200} > gpurun_out/soak/r06_soak_session.json 2> gpurun_out/soak/r06_soak_session.err; echo "session rc=$?"
cat gpurun_out/soak/r06_soak_session.json; tail -5 gpurun_out/soak/r06_soak_session.err
timeout ${LIMIT:-1500} python3 tools/soak.py --total-tf ${TOTAL_TF2:-2500} --devices 0,0 > gpurun_out/soak/r06_soak_two_slices.json 2> gpurun_out/soak/r06_soak_two_slices.err; echo "two slices rc=$?"
cat gpurun_out/soak/r06_soak_two_slices.json; tail -5 gpurun_out/soak/r06_soak_two_slices.err
