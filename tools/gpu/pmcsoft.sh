#!/bin/bash
set -u
# SQ / memory counters of the soft-decision Viterbi kernel next to the hard one (three PMC passes each)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmcsoft; rm -rf $O; mkdir -p $O
R=$GRAFT_REPO_ROOT
pass() { ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc $2 --output-format csv -d "$R/$O/$1" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --no-h2d $3 --steps 2 --warmup 1 > /dev/null 2>&1 ); }
pass s1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "--soft --snr 5"
pass s2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" "--soft --snr 5"
pass h1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" ""
pass h2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" ""
python3 tools/sq_pmc_summary.py s1="$O/s1" s2="$O/s2" h1="$O/h1" h2="$O/h2" | grep -E "viterbi_fused" | cut -c1-110
rm -rf $O/s1 $O/s2 $O/h1 $O/h2
