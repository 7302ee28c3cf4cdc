#!/bin/bash
# A/B on one box: the in-tree library against variants/libdabhip_cbprio{1,3}.so (k_decode.hip with -DDABHIP_CB_PRIO=N: s_setprio N before the chain-back), three rounds
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do for lib in "" variants/libdabhip_cbprio1.so variants/libdabhip_cbprio3.so; do
  DABHIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-intree}', round(d['value']), round(d['ms_per_step'],2), round(d['stage_ms_per_step']['viterbi'],3))"
done; done
