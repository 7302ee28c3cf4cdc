cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 5 $O/gpu_tests.log
timeout 2400 bash tools/refresh_profiles.sh > $O/refresh.log 2>&1; echo "refresh rc=$?"; tail -n 25 $O/refresh.log
tail -n 20 gpurun_out/profiles_new/bench.err
