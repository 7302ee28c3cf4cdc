#!/bin/bash
set -u
# round 5, fifth pass (the fourth pass's results were lost with its container): fused audit + new seam / fetch / near-4-GiB tests, packed 16-bit issue
# rates, scatter probe A/B, the whole suite, the host-side stall histogram over 1000 steps, then the two new sweeps against the oracle
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
variants/valu_rates p16 | tee $O/valu_rates_p16.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 12 $O/gpu_tests.log
grep -E "fused audit" $O/gpu_tests.log | cut -c1-600 | tail -5
for rep in 1 2 3; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base noscatter; done | tee $O/scatter_probe.txt
STEPS=1000 bash tools/gpu/hoststats.sh 2>&1 | tee $O/hoststats_1000.txt
cp gpurun_out/test_seeds.txt $O/ 2>/dev/null
NAME=r05_channel_oracle ROUNDS=8 STREAMS=64 TFS=24 SEED=5101 STRESS_ARGS="--channel" bash tools/gpu/stress.sh
NAME=r05_reconf_oracle ROUNDS=8 STREAMS=64 TFS=24 SEED=5202 STRESS_ARGS="--reconf" bash tools/gpu/stress.sh
