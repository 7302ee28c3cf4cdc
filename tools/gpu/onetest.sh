#!/bin/bash
set -u
# one test selection ($K) of the GPU suite
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -q -x -m gpu -k "${K:-zero}" 2>&1 | tail -15
