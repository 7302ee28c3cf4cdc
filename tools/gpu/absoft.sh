#!/bin/bash
set -u
# soft-decision tests, then the 5 dB soft workload A/B on one box: in-tree library against variants/libdabhip_prev.so
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/absoft
timeout 900 python -m pytest tests -q -x -m gpu -k "soft or config4" 2>&1 | tail -3
for rep in 1 2; do BENCH_EXTRA="--no-h2d --snr 5 --soft" bash tools/bench_variants.sh base prev; done | tee gpurun_out/absoft/lines.txt
