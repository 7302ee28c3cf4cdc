#!/bin/bash
# A/B at 256 and 512 streams: in-tree library against variants/libdabhip_k1lb4.so (tools/build_variant_k1lb4.sh), then the trace / end-to-end parity tests on the variant
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
for n in 256 512; do for lib in "" variants/libdabhip_k1lb4.so; do
  DABHIP_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} python bench.py --streams $n --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-h2d 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', '${lib:-intree}', round(d['value']), round(d['ms_per_step'],2), round(d['stage_ms_per_step']['sync'],3))"
done; done
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_k1lb4.so python -m pytest tests/test_gpu_parity.py -q -x -k "e2e or trace" 2>&1 | tail -2
