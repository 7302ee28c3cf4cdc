#!/bin/bash
set -u
# Round 6: sessions over several devices + the CLI's operator feedback / live defaults (tests/test_gpu_multi_stream.py), the host-fed session tests beside them
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sessions; mkdir -p $O
timeout 1500 python -m pytest -q -x tests/test_gpu_multi_stream.py tests/test_gpu_hostfed.py tests/test_gpu_multi.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 25 $O/tests.log
