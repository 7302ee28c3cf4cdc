#!/bin/bash
set -u
# per-kernel times of mid-size decodes (rocprofv3 --kernel-trace --stats over tools/batch_curve.py --batches 16 and 32) with the default rule (four / two lanes
# per code word) and with the lane form forced -> gpurun_out/lanesk/kernel_stats_{default,lane}_B{16,32}.csv
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/lanesk; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for B in 16 32; do
  for mode in default lane; do
    if [ $mode = lane ]; then export DABHIP_VIT_TWO_LANES=0 DABHIP_VIT_FOUR_LANES=0; else unset DABHIP_VIT_TWO_LANES DABHIP_VIT_FOUR_LANES; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $GRAFT_REPO_ROOT/tools/batch_curve.py --batches $B --steps 20 > $O/${mode}_B$B.json 2> $O/${mode}_B$B.err
    cp "$(find $O/t -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_${mode}_B$B.csv; rm -rf $O/t
    echo "B=$B $mode:"; grep -E "viterbi|regroup|ofdm_demap|sync_chain" $O/kernel_stats_${mode}_B$B.csv | cut -c1-220
  done
done
