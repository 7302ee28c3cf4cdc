#!/bin/bash
set -u
# the FIC decode through the four-lane decoder: parity tests with every FIC decode forced through it (no wave form: DABHIP_FIC_WAVE_MAX=0, DABHIP_FIC_FOUR_LANES=1),
# then the mid-size curve with it off and on (the default rule: up to 800 tiles of 64 blocks)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fic_lanes; mkdir -p $O
DABHIP_FIC_WAVE_MAX=0 DABHIP_FIC_FOUR_LANES=1 timeout 900 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py tests/test_gpu_two_lanes.py tests/test_gpu_channel.py -q -x -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log | cut -c1-300
for mode in 0 800; do
  DABHIP_FIC_FOUR_LANES=$mode timeout 600 python tools/batch_curve.py --batches 16,32,64,128 --steps 20 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"
  python - <<PY
import json
d = json.load(open("$O/curve_$mode.json"))
for r in d["curve"]:
    print("fic_four_lanes=$mode", r["streams"], round(r["ms_per_decode"], 3), round(r["eti_frames_per_s"]), {k: round(v, 3) for k, v in r.get("stage_ms", {}).items() if k in ("viterbi", "sync", "fft", "fic", "control")})
PY
done
