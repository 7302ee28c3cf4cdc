cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hostfed.py -x -q -m gpu > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -n 3 $O/new_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-variants > $O/bench_lane_$i.json 2>> $O/bench.err; echo "lane rc=$?"
DABHIP_HOST_FRESH_THREAD=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-variants > $O/bench_fresh_$i.json 2>> $O/bench.err; echo "fresh rc=$?"
done
DABHIP_TRACE_HOST=1 timeout 300 python bench.py --steps 3 --no-cpu-baseline --no-variants > /dev/null 2> $O/trace_host.txt
timeout 900 python tools/bench_hostfed.py > $O/hostfed.json 2> $O/hostfed.err; echo "hostfed rc=$?"; tail -n 3 $O/hostfed.err
timeout 600 python tools/bench_hostfed.py --segment-tfs 16 --skip-pageable > $O/hostfed16.json 2>> $O/hostfed.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); s=d['stage_ms_per_step']
    print(f, round(d['value']), round(d['ms_per_step'],3), 'control %.3f worklist %.3f' % (s['control'], s['host_worklist']))
PY
