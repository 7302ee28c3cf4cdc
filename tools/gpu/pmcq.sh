#!/bin/bash
set -u
# LDS counters of the fused OFDM kernel (one PMC pass) + the test-suite's OFDM parity tests + two bench lines
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcq; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$R/$O/pmc" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --no-h2d --steps 2 --warmup 1 > /dev/null 2>&1 )
python3 tools/sq_pmc_summary.py guard="$O/pmc" | grep -E "ofdm_demap" | cut -c1-120
rm -rf $O/pmc
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r2.py -q -x -m gpu 2>&1 | tail -3
BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev base prev | tee $O/lines.txt
