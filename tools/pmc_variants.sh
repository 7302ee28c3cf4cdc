# usage: bash tools/pmc_variants.sh name1 name2 ...   (variants/libdabhip_NAME.so; "base" = the in-tree library)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
C="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS"
for V in "$@"; do
  if [ "$V" = base ]; then unset DABHIP_LIB; else export DABHIP_LIB="$R/variants/libdabhip_"$V.so; fi
  python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 10 ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['value']), round(d['ms_per_step'],3), 'fft', round(d['stage_ms_per_step']['fft'],3), 'k2', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  rocprofv3 --pmc $C --output-format csv -d "$R/gpurun_out/pmc_"$V -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 2 --warmup 1 ${BENCH_EXTRA:-} > /dev/null 2>&1
  python3 "$R/tools/sq_pmc_summary.py" $V="$R/gpurun_out/pmc_"$V | grep -E "${KERNELS:-ofdm_demap}" | grep -E "${COUNTERS:-INSTS_VALU|INSTS_LDS|INSTS_SALU}"
done
