#!/bin/bash
set -u
# round 5, fourth pass: the fused audit + new seam / fetch / near-4-GiB tests, the packed 16-bit issue rates, what the decision scatter costs the fused
# kernel at most (A/B against variants/libdabhip_noscatter.so: wrong output, timing only), the whole suite, host-side stall histogram over 1000 steps
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_channel.py tests/test_gpu_hostfed.py tests/test_gpu_big.py -q -m gpu -k "audit or two_states or eti_fetch or four_gib or call_length" -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"; grep -E "fused audit|passed|failed|^E " $O/new_tests.log | cut -c1-600 | tail -20
variants/valu_rates p16 | tee $O/valu_rates_p16.txt
for rep in 1 2 3; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base noscatter; done | tee $O/scatter_probe.txt
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 $O/gpu_tests.log
STEPS=1000 bash tools/gpu/hoststats.sh 2>&1 | tee $O/hoststats_1000.txt
cp gpurun_out/test_seeds.txt $O/ 2>/dev/null
