#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 4 $O/gpu_tests.log
timeout 1200 python tools/decision_audit.py --channels 300 > $O/decision_audit.json 2> $O/decision_audit.err; echo "audit rc=$?"
bash tools/gpu/k1_ahead_sweep.sh
mkdir -p gpurun_out/probes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/acs_split.hip -o /tmp/acs_split && timeout 300 /tmp/acs_split > gpurun_out/probes/acs_split.txt 2>&1; echo "acs rc=$?"; cat gpurun_out/probes/acs_split.txt
