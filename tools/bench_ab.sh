# usage: VAR=NAME bash tools/bench_ab.sh : alternates runs with $VAR unset / set to 1 (same box, interleaved)
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2 3; do for V in off on; do
  if [ $V = on ]; then export $VAR=1; else unset $VAR; fi
  python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 10 ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$VAR=$V', round(d['value']), round(d['ms_per_step'],3), 'sync', round(s['sync'],3), 'fft', round(s['fft'],3), 'vit', round(s['viterbi'],3), 'fic', round(s['fic'],3), 'gather', round(s['gather'],3), 'eti', round(s['eti'],3))"
done; done
