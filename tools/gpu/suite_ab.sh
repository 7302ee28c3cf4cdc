#!/bin/bash
set -u
# the whole GPU suite, then A/B bench lines on ONE box: the in-tree library against variants/libdabhip_prev.so (three rounds)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/suite_ab
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
for rep in 1 2 3; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev; done | tee gpurun_out/suite_ab/lines.txt
