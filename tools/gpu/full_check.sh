#!/bin/bash
set -u
# One GPU-box pass over everything the round is judged on: the GPU test-suite, the profile refresh (bench + rocprofv3 passes), the
# 8-rank rehearsal, configs[3] on one GPU.  Results under gpurun_out/full/.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/full; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 $O/gpu_tests.log
timeout 3000 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"
timeout 900 bash tools/eight_ranks_one_gpu.sh; echo "rehearsal rc=$?"
timeout 1500 python tools/config3_one_gpu.py > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_times.so timeout 600 python tools/vit_tail.py > $O/vit_tail.json 2> $O/vit_tail.err; echo "tail rc=$?"
grep -v "rocprofv3\|amdgpu.ids\|^W2026\|^E2026" gpurun_out/profiles_new/bench.err | tail -n 10
