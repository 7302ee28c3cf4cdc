#!/bin/bash
set -u
# the OFDM / Viterbi / ETI parity tests, then an A/B on ONE box: the in-tree library against variants/libdabhip_prev.so
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev; done | tee gpurun_out/ab/lines.txt
