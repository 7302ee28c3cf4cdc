cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_r2.py tests/test_gpu_parity_r3.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/tests.log
for i in 1 2; do
for V in 1 0; do
DABHIP_VERIFY_FP32=$V python3 bench.py --no-cpu-baseline --no-variants --no-h2d --steps 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('VERIFY_FP32=$V', round(d['value']), round(d['ms_per_step'],3), 'sync', round(s['sync'],3), 'fp64 calls', s.get('sync_fp64_calls'), 'fft', round(s['fft'],3), 'vit', round(s['viterbi'],3))"
done; done
python3 bench.py --no-cpu-baseline --no-variants --no-h2d --steps 5 --snr 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms_per_step']; print('5 dB', round(d['value']), 'sync', round(s['sync'],3), 'fp64 calls', s.get('sync_fp64_calls'))"
