#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
python tools/audit_one.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest -q -x tests/test_gpu_multi_stream.py 2>&1 | tail -15
