#!/bin/bash
set -u
# round 5: the fine time search in single precision first (fine_time_sync32): A/B on one box against variants/libdabhip_prev.so (HEAD before it) at 256
# streams and on the small-batch curve, the trace tests, then the whole suite in the default mode and the sync tests with every call handed on (DABHIP_FINE_FP32=2)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5n; mkdir -p $O
for rep in 1 2; do BENCH_EXTRA="--no-h2d" bash tools/bench_variants.sh base prev; done | tee $O/ab_lines.txt
for lib in base prev; do
  if [ $lib = base ]; then unset DABHIP_LIB; else export DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_$lib.so; fi
  timeout 300 python tools/batch_curve.py --batches 1,4,16,64 --steps 20 --session-tfs 0 2>&1 >/dev/null | grep "B=" | sed "s/^/$lib /"
done | tee $O/curve_lines.txt
unset DABHIP_LIB
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 6 $O/gpu_tests.log | cut -c1-400
DABHIP_FINE_FP32=2 timeout 900 python -m pytest tests/test_gpu_ahead.py tests/test_gpu_frontend_ref.py tests/test_gpu_channel.py -q -m gpu > $O/distrust.log 2>&1; echo "distrust tests rc=$?"; tail -n 4 $O/distrust.log | cut -c1-400
python - <<'PY'
import numpy as np, dabtools_amd as dab
caps = [dab.synth_generate(dab.synth_preset(1, seed=31 + i, cif_count0=10 * i, snr_db=s), 30) for i, s in enumerate((1000.0, 9.0, 6.0, 5.0))]
eng = dab.Engine(0)
eng.set_sync_speculation(0)
eng.decode(caps)
st = eng.stage_ms()
print("fine time searches decided in double on 4 captures (clean, 9, 6, 5 dB) x 45 calls:", st["sync_fine_fp64_calls"], "coarse frequency:", st["sync_fp64_calls"])
PY
