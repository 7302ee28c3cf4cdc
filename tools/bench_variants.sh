# usage: bash tools/bench_variants.sh name1 name2 ...  (variants/libdabhip_NAME.so; "base" = the in-tree library): bench lines only
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for V in "$@"; do
  if [ "$V" = base ]; then unset DABHIP_LIB; else export DABHIP_LIB="$R/variants/libdabhip_"$V.so; fi
  for i in 1 2; do python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 10 ${BENCH_EXTRA:-} 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('$V', round(d['value']), round(d['ms_per_step'],3), 'sync', round(s['sync'],3), 'fft', round(s['fft'],3), 'vit', round(s['viterbi'],3), 'fic', round(s['fic'],3), 'eti', round(s['eti'],3))"; done
done
