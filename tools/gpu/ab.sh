#!/bin/bash
set -u
# A/B on ONE box: the in-tree library against variants/libdabhip_$1.so, alternating, bench lines only
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab
for rep in 1 2 3; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base ${1:-prev}; done | tee gpurun_out/ab/lines.txt
