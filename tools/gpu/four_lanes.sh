#!/bin/bash
set -u
# The decoder with 2^NL lanes per code word and no per-lane tables (vit_four_lanes.hpp).  STAGE=tests: the decoder / end-to-end parity tests with every
# lane-form decode forced through the four-lane form, then through the table-free two-lane form; STAGE=suite: the whole GPU suite through four lanes;
# STAGE=curve: the small-batch curve with the lane form, vit_two_lanes.hpp, the table-free two lanes and four lanes for every batch above the wave form's range.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/four_lanes; mkdir -p $O
T="tests/test_gpu_parity_r2.py tests/test_gpu_parity.py tests/test_gpu_two_lanes.py"
for stage in ${STAGE:-tests curve}; do
  case $stage in
    tests) DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_FOUR_LANES=1 timeout 1200 python -m pytest $T -q -x -m gpu > $O/tests4.log 2>&1; echo "four lanes: tests rc=$?"; tail -n 4 $O/tests4.log | cut -c1-300
           DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_TWO_LANES=1 DABHIP_VIT_FOUR_LANES=0 DABHIP_VIT_LANES_PLAIN=1 timeout 1200 python -m pytest $T -q -x -m gpu > $O/tests2.log 2>&1; echo "two lanes, no tables: tests rc=$?"; tail -n 4 $O/tests2.log | cut -c1-300;;
    suite) DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_FOUR_LANES=1 timeout 2400 python -m pytest tests -q -m gpu -k "not big" > $O/suite.log 2>&1; echo "suite rc=$?"; tail -n 6 $O/suite.log | cut -c1-300;;
    curve) for mode in lane two two_plain four; do
             case $mode in lane) E="DABHIP_VIT_TWO_LANES=0 DABHIP_VIT_FOUR_LANES=0";; two) E="DABHIP_VIT_TWO_LANES=1 DABHIP_VIT_FOUR_LANES=0";; two_plain) E="DABHIP_VIT_TWO_LANES=1 DABHIP_VIT_FOUR_LANES=0 DABHIP_VIT_LANES_PLAIN=1";; four) E="DABHIP_VIT_FOUR_LANES=1";; esac
             env $E timeout 600 python tools/batch_curve.py --batches ${BATCHES:-8,16,32,64,128} --steps 20 > $O/curve_$mode.json 2> $O/curve_$mode.err; echo "curve $mode rc=$?"
             python - <<PY
import json
d = json.load(open("$O/curve_$mode.json"))
for r in d["curve"]:
    print("$mode", r["streams"], round(r["ms_per_decode"], 3), round(r["eti_frames_per_s"]), {k: round(v, 3) for k, v in r.get("stage_ms", {}).items() if k in ("viterbi", "sync", "fft")})
PY
           done;;
  esac
done
