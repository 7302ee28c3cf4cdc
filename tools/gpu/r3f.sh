cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity_r3.py tests/test_gpu_hostfed.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/tests.log
for M in "DABHIP_PREFETCH_KERNEL=0 DABHIP_UP_STREAMS=1" "DABHIP_PREFETCH_KERNEL=0 DABHIP_UP_STREAMS=2" "DABHIP_PREFETCH_KERNEL=0 DABHIP_UP_STREAMS=4" "DABHIP_PREFETCH_KERNEL=1" "DABHIP_PREFETCH_KERNEL=16" "DABHIP_PREFETCH_KERNEL=256"; do
  env $M timeout 600 python tools/bench_hostfed.py --skip-pageable --reps 2 2>>$O/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); s=d['session_prefetch']; print('$M', 'oneshot %.0f (%.1f GB/s)' % (d['one_shot_pinned']['value'], d['one_shot_pinned']['h2d_GBps']), 'session %.0f steady %.0f fps %.2f GB/s %.2f ms/seg' % (s['value'], s['steady_state']['value'], s['steady_state']['GBps'], s['steady_state']['ms_per_segment']))"
done
timeout 1500 python tools/config3_one_gpu.py > $O/config3.json 2> $O/config3.err; echo "config3 rc=$?"; tail -n 3 $O/config3.err; python3 -c "
import json; d=json.load(open('$O/config3.json')); print({k:d[k] for k in ('eti_frames','all_streams_full_count','oracle_byte_equal','wall_ms_per_decode','value_one_gpu_time_sliced')}); print(d['per_slice_last'])"
