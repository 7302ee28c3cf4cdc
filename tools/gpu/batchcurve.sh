#!/bin/bash
set -u
# the small-batch curve (tools/batch_curve.py): B = 1 .. 256 streams x 64 TF + one-stream sessions fed TF by TF; host-side phase trace at B = 1
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/batchcurve; mkdir -p $O
timeout 900 python tools/batch_curve.py ${CURVE_EXTRA:-} > $O/curve.json 2> $O/curve.err; echo "curve rc=$?"; cat $O/curve.err | tail -n 30
DABHIP_TRACE_HOST=1 timeout 300 python tools/batch_curve.py --max-batch 1 --steps 2 --session-tfs 20 > /dev/null 2> $O/trace_b1.err; tail -n 60 $O/trace_b1.err
