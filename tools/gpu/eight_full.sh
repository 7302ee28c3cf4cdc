#!/bin/bash
# The 8-rank rehearsal at FULL size on one GPU: 8 bench.py rank processes x 256 streams x 64 TF sharing GPU 0 (profiles/r03_eight_ranks_full_size_one_gpu.json),
# twice, then one rank alone with the same flags.
set -euo pipefail
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/eight_full; mkdir -p $O
F="--streams 256 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-h2d"
for i in 1 2; do DABHIP_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 8 $F > $O/eight_$i.json 2> $O/eight_$i.err || echo "run $i rc=$?"; done
timeout 600 python bench.py --gpus 1 $F > $O/one.json 2> $O/one.err || echo "one rc=$?"
python - <<'PY'
import json
for n in ("eight_1", "eight_2", "one"):
    d = json.loads(open("gpurun_out/eight_full/%s.json" % n).read().strip().splitlines()[-1])
    print(n, round(d["value"]), round(d["ms_per_step"], 2), [(r["host_ms_per_step"]["control"], r["host_ms_per_step"]["host_worklist"]) for r in d["ranks"]])
PY
# the 1 / 2 / 4 / 8 table on this one GPU: every multi-rank row is labelled a rehearsal by the tool itself (one distinct PCI bus id), no efficiency reported
DABHIP_BENCH_ONE_DEVICE=1 timeout 1500 python tools/scale_sweep.py --one-device --gpus 1,2,4,8 --steps 5 > $O/scale_sweep.json 2> $O/scale_sweep.txt || echo "sweep rc=$?"
cat $O/scale_sweep.txt
