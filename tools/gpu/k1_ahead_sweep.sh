#!/bin/bash
set -u
# K1's look-ahead pass for mid-size batches (second half): the small-batch curve at B = 8, 16, 32, 64 with the plain chain and with the pass
# forced on at windows of 33 / 17 / 9 start positions (DABHIP_K1_HYP); sync stage, calls served from the table, decode time.  -> gpurun_out/k1ahead/
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/k1ahead; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python tools/batch_curve.py --batches 8,16,32,64 --max-batch 64 --steps 10 --session-tfs 48 > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for c in d["curve"]:
    s = c["stage_ms"]
    print(sys.argv[2], "B=%d" % c["streams"], "decode %.3f ms" % c["ms_per_decode"], "sync %.3f" % s["sync"], "table hits %d" % s.get("sync_spec_calls", 0), "%.0f frames/s" % c["eti_frames_per_s"])
PY
}
run chain DABHIP_K1_SPEC=0
run ahead33 DABHIP_K1_SPEC=1 DABHIP_K1_HYP=33
run ahead17 DABHIP_K1_SPEC=1 DABHIP_K1_HYP=17
run ahead9 DABHIP_K1_SPEC=1 DABHIP_K1_HYP=9
