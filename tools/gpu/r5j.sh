#!/bin/bash
set -u
# round 5, final pass: the whole GPU suite, the profile refresh (bench + rocprofv3 passes), the 8-rank rehearsal at full size with the quota-aware pools,
# configs[3] on one GPU, the small-batch curve, CLI throughput
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/full; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 $O/gpu_tests.log | cut -c1-400
timeout 2400 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"
grep -v "rocprofv3\|amdgpu.ids\|^W2026\|^E2026" gpurun_out/profiles_new/bench.err | tail -n 10
timeout 900 bash tools/gpu/eight_full.sh 2>&1 | tail -n 5; echo "eight_full rc=$?"
timeout 900 python tools/config3_one_gpu.py > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
for mode in 0 default; do
  if [ $mode = default ]; then unset DABHIP_K1_SPEC; else export DABHIP_K1_SPEC=$mode; fi
  timeout 600 python tools/batch_curve.py --steps 20 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"; tail -n 12 $O/curve_$mode.err
done
unset DABHIP_K1_SPEC
