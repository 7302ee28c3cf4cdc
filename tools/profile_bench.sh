# kernel-trace statistics of the default bench step: gpurun_out/prof_stats/*kernel_stats.csv
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 10 ${BENCH_EXTRA:-} > "$R/gpurun_out/prof_stats_bench.json" 2>/dev/null
F=$(find "$R/gpurun_out/prof_stats" -name "*kernel_stats.csv" | head -1)
cp $F "$R/gpurun_out/kernel_stats.csv"
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kernel_stats.csv")))
for r in rows[:18]:
    n=r["Name"].replace("dabhip::(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-36s calls %5s  avg %10.1f us  total %9.2f ms  %5s%%" % (n[:36], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
