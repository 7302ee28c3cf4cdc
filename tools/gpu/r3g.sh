cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_times.so timeout 600 python tools/vit_tail.py > $O/vit_tail.json 2> $O/vit_tail.err; echo "tail rc=$?"; tail -n 3 $O/vit_tail.err
python3 -c "
import json; d=json.load(open('$O/vit_tail.json')); print({k:v for k,v in d.items() if k not in ('classes','resident_waves_every_250us')}); print(d['resident_waves_every_250us']); [print(c) for c in d['classes']]"
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 5 $O/gpu_tests.log
