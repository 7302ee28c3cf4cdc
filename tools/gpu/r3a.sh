cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_hostfed.py -x -q -m gpu > gpurun_out/r3a/new_tests.log 2>&1; echo "new tests rc=$?" 
timeout 1500 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_multi.py --deselect tests/test_gpu_hostfed.py > gpurun_out/r3a/all_tests.log 2>&1; echo "old tests rc=$?"
timeout 600 python bench.py --steps 10 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
timeout 900 bash tools/eight_ranks_one_gpu.sh; echo "rehearsal rc=$?"
tail -5 gpurun_out/r3a/new_tests.log gpurun_out/r3a/all_tests.log
