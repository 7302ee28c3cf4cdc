#!/bin/bash
# kernel timeline of one default step (rocprofv3 --kernel-trace), printed
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/timeline; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline --no-h2d --no-variants --steps 3 --warmup 1 > "$GRAFT_REPO_ROOT/$O/bench.json" 2> "$GRAFT_REPO_ROOT/$O/bench.err"
cd "$GRAFT_REPO_ROOT"
python3 tools/timeline.py $O/trace > $O/step_timeline.txt
cut -c1-110 $O/step_timeline.txt | head -60
