#!/bin/bash
set -u
# after a change of the OFDM stage / the guard: the GPU suite, the decision audit, the profile refresh (bench lines + rocprofv3 passes), the guard's price by level
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 4 $O/gpu_tests.log
timeout 1200 python tools/decision_audit.py --channels 300 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
timeout 3000 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"
timeout 900 python tools/bench_guard.py > $O/guard_levels.jsonl 2> $O/guard_levels.err; echo "guard levels rc=$?"; tail -n 1 $O/guard_levels.jsonl | cut -c1-900
grep -v "rocprofv3\|amdgpu.ids\|^W2026\|^E2026" gpurun_out/profiles_new/bench.err | tail -n 6
