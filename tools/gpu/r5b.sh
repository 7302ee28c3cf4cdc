#!/bin/bash
set -u
# round 5: the channel tests again, then what the tail bytes cost the K1 chain: in-tree library against variants/libdabhip_notail.so on one box
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_channel.py -q -m gpu > $O/channel.log 2>&1; echo "channel tests rc=$?"; tail -n 30 $O/channel.log
for rep in 1 2 3; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base notail; done | tee $O/lines.txt
