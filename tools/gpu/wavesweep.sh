#!/bin/bash
set -u
# where the two decoder forms cross: the batch curve at B = 4 .. 24 with the wave form forced on / off for the MSC and for the FIC
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wavesweep; mkdir -p $O
for cfg in "1000000 1000000" "0 0" "1000000 0" "0 1000000"; do
  set -- $cfg
  echo "== MSC wave max $1, FIC wave max $2"
  DABHIP_VIT_WAVE_MAX=$1 DABHIP_FIC_WAVE_MAX=$2 timeout 600 python tools/batch_curve.py --max-batch 24 --batches 4,6,8,10,12,16,24 --session-tfs 20 > $O/curve_$1_$2.json 2> $O/err_$1_$2.txt
  grep "^B=" $O/err_$1_$2.txt
done
