set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_k1" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
python3 "$R/tools/sq_pmc_summary.py" k1="$R/gpurun_out/pmc_k1" | grep -E "sync_"
