#!/bin/bash
set -u
# K1 work: the whole GPU suite (every trace test), the chain's phase times (measurement build), A/B bench lines against variants/libdabhip_prev.so
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/k1
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_synctimes.so python tools/sync_times.py | tee gpurun_out/k1/sync_times.json
BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev base prev | tee gpurun_out/k1/lines.txt
