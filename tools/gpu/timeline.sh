#!/bin/bash
set -u
# kernel timeline of one default step (rocprofv3 --kernel-trace), printed
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-h2d --no-variants --steps 3 --warmup 1 > "$GRAFT_REPO_ROOT/$O/bench.json" 2> "$GRAFT_REPO_ROOT/$O/bench.err"
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
python3 tools/timeline.py $O/trace > $O/step_timeline.txt
cut -c1-110 $O/step_timeline.txt | head -60
