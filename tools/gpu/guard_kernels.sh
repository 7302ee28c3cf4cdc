#!/bin/bash
set -u
# per-kernel times of the OFDM stage at guard levels 1 and 2, 5 dB (rocprofv3 --kernel-trace --stats over tools/guard_kernels.py) -> gpurun_out/guardk/
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/guardk; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in ${LEVELS:-1 2}; do
  export LEVEL=$L
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/l$L -- python3 $GRAFT_REPO_ROOT/tools/guard_kernels.py > $O/l$L.json 2> $O/l$L.err
  cp "$(find $O/l$L -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_level$L.csv; rm -rf $O/l$L
  echo "level $L:"; cat $O/l$L.json | cut -c1-300; grep -E "ofdm_demap|exact_decide|viterbi_fused|regroup" $O/kernel_stats_level$L.csv | cut -c1-200
done
