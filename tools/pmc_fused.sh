set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_g" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_p" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --no-parity-guard --steps 2 --warmup 1 > /dev/null 2>&1
python3 "$R/tools/sq_pmc_summary.py" guard="$R/gpurun_out/pmc_g" plain="$R/gpurun_out/pmc_p" | grep -E "ofdm_demap|viterbi_fused"
