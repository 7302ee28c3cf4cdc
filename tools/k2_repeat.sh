set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for V in "$@"; do
  if [ "$V" = base ]; then unset DABHIP_LIB; else export DABHIP_LIB="$R/variants/libdabhip_"$V.so; fi
  for i in 1 2 3; do python3 "$R/bench.py" --no-cpu-baseline --no-variants --steps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['value']), round(d['ms_per_step'],3), 'fft', round(d['stage_ms_per_step']['fft'],3), 'k2', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"; done
done
