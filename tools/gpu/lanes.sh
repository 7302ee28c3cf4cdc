#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/quick
variants/lane_swap_check | tee gpurun_out/quick/lane_swap.txt
variants/valu_rates lanes | tee gpurun_out/quick/lanes.txt
