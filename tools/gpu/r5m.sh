#!/bin/bash
set -u
# round 5: after the look-ahead pass's prediction moved into fifo_view.hpp: its tests + the trace tests, then ONE stream of 4.4 GB (16,785 calls:
# five look-ahead passes of 4096 calls) with the plain chain and with the schedule, oracle-checked on its first and last 400 TF
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_frontend_ref.py tests/test_gpu_big.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 5 $O/tests.log | cut -c1-400
for mode in 0 default; do
  if [ $mode = default ]; then unset DABHIP_K1_SPEC; else export DABHIP_K1_SPEC=$mode; fi
  timeout 900 python tools/big_stream_check.py > $O/big_$mode.json 2> $O/big_$mode.err; echo "big $mode rc=$?"
  python - <<PY
import json
d = json.loads(open("$O/big_$mode.json").read().strip().splitlines()[-1])
print("$mode", {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("seconds", "eti_frames", "equal", "equal_to_one_shot", "first_equal", "last_equal")}) for k, v in d.items()})
PY
done
