#!/bin/bash
set -u
# parity tests of the decoders, then A/B on one box for the default (hard, clean) and the 5 dB soft workload: in-tree library against variants/libdabhip_prev.so
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/abboth
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev; done | tee gpurun_out/abboth/hard.txt
for rep in 1 2; do BENCH_EXTRA="--no-h2d --snr 5 --soft" bash tools/bench_variants.sh base prev; done | tee gpurun_out/abboth/soft.txt
