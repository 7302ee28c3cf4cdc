#!/bin/bash
set -u
# The parity guard by level: the guard's parity tests (both levels), then tools/bench_guard.py (levels 0 / 1 / 2, clean .. 5 dB)
# -> gpurun_out/guard/levels.jsonl.  TESTS: pytest node selection (default: the guard / audit tests); SNRS / REPS / STREAMS go to bench_guard.py.
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/guard; mkdir -p $O
TESTS=${TESTS:-tests/test_gpu_parity_r2.py::test_parity_guard_makes_fp32_decisions_exact tests/test_gpu_parity_r2.py::test_parity_guard_end_to_end_and_off_switch tests/test_gpu_channel.py::test_decision_audit_of_the_fused_kernel tests/test_gpu_parity_r3.py::test_parity_guard_list_overflow_degrades_to_a_full_fp64_decision tests/test_gpu_parity_r3.py::test_exact_zero_products_are_decided_as_the_reference_decides_them}
timeout 1200 python -m pytest -q -x $TESTS > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 12 $O/tests.log
timeout 1500 python tools/bench_guard.py > $O/levels.jsonl 2> $O/levels.err; echo "bench_guard rc=$?"; tail -n 3 $O/levels.err
tail -n 1 $O/levels.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); [print(k, json.dumps(v)) for k, v in d['summary'].items()]"
