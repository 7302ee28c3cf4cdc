GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
set -u
mkdir -p gpurun_out/sweep
timeout 2400 python tools/stress_parity.py --rounds 16 --streams 64 --tfs 28 --seed 30303 > gpurun_out/sweep/stress_parity.json 2> gpurun_out/sweep/err.txt; echo rc=$?
tail -n 3 gpurun_out/sweep/err.txt
python3 -c "
import json; d=json.load(open('gpurun_out/sweep/stress_parity.json')); print({k:d[k] for k in ('eti_frames_compared','calls_compared','differences','seconds')}, len(d['cases']))"
