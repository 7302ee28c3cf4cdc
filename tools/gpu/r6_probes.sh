#!/bin/bash
set -u
# Round 6 probes: (1) matrix pipe beside the packed-fp32 VALU stream (tools/ubench/mfma_coissue.hip); (2) the CLI contract host -> host at
# a size where its steady state is measurable, one session and two sessions on GPU 0 (tools/cli_throughput.py); (3) bench.py's host-fed session loop at full size, 5 dB.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/probes; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_coissue.hip -o /tmp/mfma_coissue && timeout 300 /tmp/mfma_coissue > $O/mfma_coissue.txt 2>&1; echo "mfma rc=$?"; cat $O/mfma_coissue.txt
D=/dev/shm/dabhip_cli_$$; mkdir -p $D
timeout 1500 python tools/cli_throughput.py --streams ${CLI_STREAMS:-256} --tfs ${CLI_TFS:-128} --dir $D > $O/cli_throughput.json 2> $O/cli_throughput.err; echo "cli rc=$?"; rm -rf $D
python - <<'PY'
import json
d = json.load(open("gpurun_out/probes/cli_throughput.json"))
for k in ("stream_pipeline", "stream_pipeline_two_sessions_on_one_gpu", "one_batch"):
    print(k, round(d[k]["eti_frames_per_s"]), d[k]["inside_the_process"])
PY
STREAMS=256 SNR=5 timeout 600 python tools/repro_fetch.py 2>&1 | tail -4
