#!/bin/bash
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/synctimes
DABHIP_LIB=$GRAFT_REPO_ROOT/variants/libdabhip_synctimes.so python tools/sync_times.py | tee gpurun_out/synctimes/sync_times.json
