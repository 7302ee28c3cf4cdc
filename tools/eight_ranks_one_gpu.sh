#!/bin/bash
# Rehearsal of BASELINE configs[3]'s host side on a ONE-GPU box: eight bench.py ranks (one process each, 32 streams each so that
# they fit one GPU's HBM and time slices) share GPU 0, next to one rank alone at the same 32 streams.  What it measures is the
# per-rank HOST work (control plane, work lists, wall) with eight ranks' thread pools on the host at once -- not throughput.
set -euo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O="$R/gpurun_out/rehearsal"
mkdir -p "$O"
cd "$R"
python3 bench.py --gpus 1 --streams 32 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > "$O/one_rank.json" 2> "$O/one_rank.err"
DABHIP_BENCH_ONE_DEVICE=1 python3 bench.py --gpus 8 --streams 32 --steps 10 --warmup 2 --no-cpu-baseline --no-variants > "$O/eight_ranks.json" 2> "$O/eight_ranks.err"
python3 - "$O" <<'PY'
import json, sys, os
o = sys.argv[1]
one = json.loads(open(os.path.join(o, "one_rank.json")).read().strip().splitlines()[-1])
eight = json.loads(open(os.path.join(o, "eight_ranks.json")).read().strip().splitlines()[-1])
def host(r): return r.get("host_ms_per_step", {})
out = {"what": "8 bench.py ranks x 32 streams sharing GPU 0 (DABHIP_BENCH_ONE_DEVICE=1) vs 1 rank x 32 streams alone; host-side ms per step per rank",
       "host_cores": os.cpu_count(),
       "one_rank": {"ms_per_step": one["ms_per_step"], "value": one["value"], "host_threads": one["ranks"][0]["host_threads"], "host_ms_per_step": host(one["ranks"][0])},
       "eight_ranks": {"ms_per_step": eight["ms_per_step"], "value": eight["value"],
                       "ranks": [{"rank": r["rank"], "host_threads": r["host_threads"], "elapsed_s": r["elapsed_s"], "host_ms_per_step": host(r)} for r in eight["ranks"]]}}
h1 = host(one["ranks"][0])
for k in ("control", "host_worklist"):
    if k in h1 and h1[k] > 0:
        out.setdefault("growth", {})[k] = max(host(r).get(k, 0.0) for r in eight["ranks"]) / h1[k]
json.dump(out, open(os.path.join(o, "eight_ranks_one_gpu.json"), "w"), indent=1)
print(json.dumps(out.get("growth")))
PY
