#!/bin/bash
set -u
# The two-lanes-per-code-word decoder (vit_two_lanes.hpp).  STAGE=tests: the decoder / end-to-end parity tests with every lane-form decode forced through it
# (DABHIP_VIT_WAVE_MAX=0: no wave form; DABHIP_VIT_TWO_LANES=1: always two lanes); STAGE=suite: the whole GPU suite that way; STAGE=curve: the small-batch curve
# with the lane form and with two lanes, B = 4 .. 128.  (DABHIP_VIT_FOUR_LANES=0 throughout: four lanes would take the small batches first.)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/two_lanes; mkdir -p $O
for stage in ${STAGE:-tests curve}; do
  case $stage in
    tests) DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_TWO_LANES=1 DABHIP_VIT_FOUR_LANES=0 timeout 1200 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity.py -q -x -m gpu ${K:+-k "$K"} > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 15 $O/tests.log | cut -c1-300;;
    suite) DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_TWO_LANES=1 DABHIP_VIT_FOUR_LANES=0 timeout 2400 python -m pytest tests -q -m gpu -k "not big" > $O/suite.log 2>&1; echo "suite rc=$?"; tail -n 6 $O/suite.log | cut -c1-300;;
    curve) for mode in 0 1; do
             DABHIP_VIT_WAVE_MAX=${WAVE_MAX:-12288} DABHIP_VIT_TWO_LANES=$mode DABHIP_VIT_FOUR_LANES=0 timeout 600 python tools/batch_curve.py --batches ${BATCHES:-4,8,16,32,64,128} --steps 20 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"
             python - <<PY
import json
d = json.load(open("$O/curve_$mode.json"))
for r in d["curve"]:
    print("two_lanes=$mode", r["streams"], round(r["ms_per_decode"], 3), round(r["eti_frames_per_s"]), {k: round(v, 3) for k, v in r.get("stage_ms", {}).items() if k in ("viterbi", "sync", "fft")})
PY
           done;;
  esac
done
