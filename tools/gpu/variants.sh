#!/bin/bash
set -u
# bench lines of the library variants named in $VARIANTS (tools/build_variant.sh), no CPU baseline / host-fed leg
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/variants
BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh ${VARIANTS:-base} | tee gpurun_out/variants/lines.txt
