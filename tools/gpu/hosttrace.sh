#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/hosttrace
DABHIP_TRACE_HOST=1 python3 bench.py --no-cpu-baseline --no-h2d --no-variants --steps 4 --warmup 1 > gpurun_out/hosttrace/bench.json 2> gpurun_out/hosttrace/trace.txt
grep -v "amdgpu.ids" gpurun_out/hosttrace/trace.txt | tail -45
