cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/soft5
timeout 900 python3 bench.py --steps 10 --snr 5 --soft > gpurun_out/soft5/bench_soft5db.json 2> gpurun_out/soft5/err.txt; echo rc=$?
tail -n 5 gpurun_out/soft5/err.txt
python3 -c "
import json; d=json.loads(open('gpurun_out/soft5/bench_soft5db.json').read().strip().splitlines()[-1]); print(d['value'], d['payload']); c=d['cpu_baseline']; print(c.get('error')); print(c['value'], c['reference_backend_scalar'].get('payload'), c['reference_backend_sse'].get('payload'), c['reference_backend_sse']['value'])"
timeout 600 python -m pytest tests/test_gpu_parity_r3.py -q -m gpu -k "sync_verification" 2>&1 | tail -n 2
