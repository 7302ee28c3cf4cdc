#!/bin/bash
set -u
# the guard's parity tests, then the noisy hard-decision workloads (5 and 7 dB) A/B on one box: in-tree library against variants/libdabhip_prev.so
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/abnoisy
timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py -q -x -m gpu -k "guard or zero or sweep or noisy or config" 2>&1 | tail -3
for snr in 5 7; do echo "snr $snr"; BENCH_EXTRA="--no-h2d --snr $snr" bash tools/bench_variants.sh base prev; done | tee gpurun_out/abnoisy/lines.txt
