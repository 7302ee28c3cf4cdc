cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ranks; mkdir -p $O
export DABHIP_BENCH_ONE_DEVICE=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --streams 64 --no-cpu-baseline > $O/torchrun2.json 2> $O/torchrun2.err; echo "torchrun rc=$?"
timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --streams 64 --no-cpu-baseline > $O/self2.json 2> $O/self2.err; echo "self-launch rc=$?"
timeout 600 python bench.py --gpus 4 --in-process --steps 3 --warmup 1 --streams 64 > $O/inproc4.json 2> $O/inproc4.err; echo "in-process rc=$?"
python3 - <<'PY'
import json
for f in ('torchrun2','self2','inproc4'):
    try:
        d=json.loads([l for l in open('gpurun_out/ranks/%s.json'%f).read().splitlines() if l.startswith('{')][-1])
        print(f, d['n_gpus'], round(d['value']), round(d['ms_per_step'],2), d['config']['eti_frames_per_step'], [r.get('eti_frames_per_step', r.get('eti_frames')) for r in d.get('ranks', d.get('slices'))], 'roofline' in d)
    except Exception as e:
        print(f, 'ERR', e); print(open('gpurun_out/ranks/%s.err'%f).read()[-800:])
PY
