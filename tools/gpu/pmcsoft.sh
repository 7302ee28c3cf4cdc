#!/bin/bash
set -u
# SQ counters of the soft-decision Viterbi kernel (two PMC passes)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcsoft; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --no-h2d --soft --snr 5 --steps 2 --warmup 1 > /dev/null 2>&1 )
python3 tools/sq_pmc_summary.py soft="$O/pmc" | grep -E "viterbi_fused|ofdm_demap|regroup" | cut -c1-120
rm -rf $O/pmc
