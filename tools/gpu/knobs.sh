#!/bin/bash
# the GPU suite under the product's environment knobs: every decode in the wave-per-code-word form, every decode in the lane form with the
# fp64-only K1 verification, host placement off.  (End of round 4: 77 + 80 + 80 passed.)
set -u
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
DABHIP_VIT_WAVE_MAX=100000000 python -m pytest tests -m gpu -q -x -k "not big and not config4 and not full_size and not config2" 2>&1 | tail -3
DABHIP_VIT_WAVE_MAX=0 DABHIP_VIT_TWO_LANES=0 DABHIP_VIT_FOUR_LANES=0 DABHIP_FIC_FOUR_LANES=0 DABHIP_VERIFY_FP32=0 python -m pytest tests -m gpu -q -x -k "not big" 2>&1 | tail -3
DABHIP_NUMA=0 DABHIP_PREFETCH_KERNEL=0 python -m pytest tests -m gpu -q -x -k "not big" 2>&1 | tail -3
