#!/bin/bash
set -u
# Things left running.  (1) tools/soak.py: 4 looped captures fed to a session in live-sized feeds past 2^32 bytes per stream, frames against the oracle and
# against the frames 1000 earlier, heap in use / resident set / device memory sampled throughout; the same, shorter, through a session over two slices of
# the one GPU (dabhip_multi_stream).  (2) tools/soak_cli.py: the CLI under a pipe for 4,500 segments, its resident set sampled from outside -- with the
# runtime's records let go of (the default) and, for the "before" column, not (DABHIP_NO_REAP=1); then two inputs over named pipes on two slices.
# (3) tools/hip_retained_commands.py: what the HIP runtime keeps per copy pattern (the reason for (2)'s difference).
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/soak; mkdir -p $O
run() { name=$1; shift; timeout ${LIMIT:-900} "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; cut -c1-1800 $O/$name.json; tail -3 $O/$name.err; }
run r06_soak_session python3 tools/soak.py --total-tf ${TOTAL_TF:-11200}
run r06_soak_two_slices python3 tools/soak.py --total-tf ${TOTAL_TF2:-2500} --devices 0,0
run r06_soak_cli python3 tools/soak_cli.py
DABHIP_NO_REAP=1 run r06_soak_cli_without_reaping python3 tools/soak_cli.py      # (rc=1 expected: this is the growth)
run r06_soak_cli_two_inputs_two_slices python3 tools/soak_cli.py --inputs 2 --devices 0,0 --total-tf 3000
python3 tools/hip_retained_commands.py > $O/r06_hip_retained_commands.txt 2>&1; cat $O/r06_hip_retained_commands.txt | cut -c1-300
# (4) tools/soak_seams.py: the reference's per-buffer call pattern through the S2 + S3 seams, 9,000 buffers; and its "before" column
run r06_soak_seams python3 tools/soak_seams.py
DABHIP_NO_REAP=1 run r06_soak_seams_without_reaping python3 tools/soak_seams.py --oracle-tf 0
# (5) tools/soak_batch.py: the batch entries (engine and multi over two slices) 4,800 times, frames read back three ways
run r06_soak_batch python3 tools/soak_batch.py --reps 4800
