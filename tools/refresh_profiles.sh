#!/bin/bash
# Produces the round's profile artefacts under gpurun_out/profiles_new/ (copy the ones to keep into profiles/ as rNN_<name>).
# Every rocprofv3 command has the program itself after "--" (python3 bench.py ...), and counters are collected in their own runs.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/profiles_new"
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
# PMC passes (one decode = the set-up decode is tiny; warmup 1 + steps 2 = 3 full decodes + the K2 roofline launches)
P="--no-cpu-baseline --no-variants --no-h2d --profile-pass --steps 2 --warmup 1"
CLK="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES"
C1="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU"
# dynamic instruction classes (beside the static mix of tools/fused_isa_mix.sh)
C3="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64"
C2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --kernel-trace --pmc $CLK --output-format csv -d "$O/pmc_clk" -- python3 "$B" $P > "$O/pmc_clk_bench.json" 2>> "$O/bench.err"
rocprofv3 --pmc $C1 --output-format csv -d "$O/pmc_f1" -- python3 "$B" $P > /dev/null 2>> "$O/bench.err"
rocprofv3 --pmc $C2 --output-format csv -d "$O/pmc_f2" -- python3 "$B" $P > /dev/null 2>> "$O/bench.err"
rocprofv3 --pmc $C3 --output-format csv -d "$O/pmc_mix" -- python3 "$B" $P > /dev/null 2>> "$O/bench.err" || echo "mix pass failed (counter set not accepted): left out"
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for CN in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CN --output-format csv -d "$O/pmc_$CN" -- python3 "$B" $P > /dev/null 2>> "$O/bench.err"
done
META=$(python3 - "$O/pmc_clk_bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d["profile_meta"]
print(" ".join("--meta %s=%s" % (k, v) for k, v in sorted(p.items())))
PY
)
( echo "# rocprofv3 [--kernel-trace] --pmc <counters> --output-format csv -- python3 bench.py $P   (passes: clk (+ kernel trace), f1, f2, fetch, write)"
  echo "# counter sums over all dispatches of a kernel in the run; FETCH_SIZE / WRITE_SIZE in KiB; meta rows describe the profiled run; tools/sq_pmc_summary.py"
  MIX=""; [ -n "$(find "$O/pmc_mix" -name '*counter_collection.csv' 2>/dev/null | head -n 1)" ] && MIX="mix=$O/pmc_mix"
  python3 "$R/tools/sq_pmc_summary.py" $META clk="$O/pmc_clk" f1="$O/pmc_f1" f2="$O/pmc_f2" fetch="$O/pmc_FETCH_SIZE" write="$O/pmc_WRITE_SIZE" $MIX ) > "$O/pmc_summary.csv"
rm -rf "$O"/pmc_clk "$O"/pmc_f1 "$O"/pmc_f2 "$O"/pmc_FETCH_SIZE "$O"/pmc_WRITE_SIZE "$O"/pmc_mix
# the bench runs below read the profile just made (bench.py: PROFILE_PMC); keep what the tree held, restore it afterwards
ROUND=$(python3 -c "import re,sys; print(re.search(r'profiles\", \"(r\d+)_pmc_summary', open(sys.argv[1]).read()).group(1))" "$B")
mkdir -p "$R/profiles"
[ -f "$R/profiles/${ROUND}_pmc_summary.csv" ] && cp "$R/profiles/${ROUND}_pmc_summary.csv" "$O/pmc_summary_previous.csv"
cp "$O/pmc_summary.csv" "$R/profiles/${ROUND}_pmc_summary.csv"
python3 "$B" --steps 20 > "$O/bench.json" 2>> "$O/bench.err"
python3 "$B" --steps 10 --snr 5 --soft > "$O/bench_soft5db.json" 2>> "$O/bench.err"
python3 "$B" --steps 10 --snr 5 --no-cpu-baseline > "$O/bench_hard5db.json" 2>> "$O/bench.err"
python3 "$B" --steps 10 --snr 7 --soft --no-cpu-baseline > "$O/bench_soft7db.json" 2>> "$O/bench.err"
python3 "$B" --steps 10 --snr 7 --no-cpu-baseline > "$O/bench_hard7db.json" 2>> "$O/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$B" --no-cpu-baseline --no-h2d --steps 10 > "$O/bench_under_rocprof.json" 2>> "$O/bench.err"
cp "$(find "$O/trace" -name '*kernel_stats.csv' | head -1)" "$O/bench_kernel_stats.csv"
cp "$(find "$O/trace" -name '*domain_stats.csv' | head -1)" "$O/bench_domain_stats.csv" 2>/dev/null || true
rm -rf "$O/trace"
rocprofv3 --kernel-trace --output-format csv -d "$O/trace2" -- python3 "$B" --no-cpu-baseline --no-variants --no-h2d --steps 5 > /dev/null 2>> "$O/bench.err"
python3 "$R/tools/timeline.py" "$O/trace2" > "$O/step_timeline.txt"      # the default step (no variants after it)
rm -rf "$O/trace2"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w "$R/tools/ubench/valu_rates.hip" -o /tmp/valu_rates && /tmp/valu_rates > "$O/valu_rates.txt" 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w "$R/tools/ubench/hbm_rates.hip" -o /tmp/hbm_rates && /tmp/hbm_rates > "$O/hbm_rates.txt" 2>&1
ls -la "$O"
