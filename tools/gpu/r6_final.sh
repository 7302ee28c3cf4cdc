#!/bin/bash
set -u
# Round 6, the evidence pass on ONE box: GPU suite, decision audit (both guard levels), profile refresh (bench lines + rocprofv3 passes), the guard's price by
# level, the 8-rank rehearsal + scale sweep (one GPU: labelled as such), configs[3] on one GPU, the small-batch curve.  Results under gpurun_out/.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
if [ "${SKIP_SUITE:-0}" != "1" ]; then timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 4 $O/gpu_tests.log; fi
timeout 1200 python tools/decision_audit.py --channels 300 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
timeout 3000 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"
timeout 900 python tools/bench_guard.py > $O/guard_levels.jsonl 2> $O/guard_levels.err; echo "guard levels rc=$?"; tail -n 1 $O/guard_levels.jsonl | cut -c1-600
timeout 2400 bash tools/gpu/eight_full.sh > $O/eight_full.log 2>&1; echo "eight rc=$?"; tail -n 8 $O/eight_full.log
timeout 1500 python tools/config3_one_gpu.py > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"
timeout 900 python tools/batch_curve.py > $O/batch_curve.json 2> $O/batch_curve.err; echo "curve rc=$?"
grep -v "rocprofv3\|amdgpu.ids\|^W2026\|^E2026" gpurun_out/profiles_new/bench.err | tail -n 10
