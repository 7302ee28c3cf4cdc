#!/bin/bash
set -u
# kernel timeline of ONE-ensemble decodes (64 TF, IQ resident) with K1's look-ahead schedule (default) and with the plain chain (DABHIP_K1_SPEC=0)
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}; export GRAFT_REPO_ROOT; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/timeline_b1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in default 0; do
  if [ $mode = default ]; then unset DABHIP_K1_SPEC; else export DABHIP_K1_SPEC=$mode; fi
  rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$O/trace_$mode" -- python3 "$GRAFT_REPO_ROOT/tools/batch_curve.py" --batches 1 --steps 4 --session-tfs 0 > /dev/null 2> "$GRAFT_REPO_ROOT/$O/err_$mode.txt"
  python3 - "$GRAFT_REPO_ROOT/$O/trace_$mode" <<'PY' > "$GRAFT_REPO_ROOT/$O/timeline_$mode.txt"
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'scan_setup_kernel' in r['Kernel_Name']]
start, stop = idx[-2], idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:stop]:
    n = r['Kernel_Name'].replace('dabhip::(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:34]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-34s start %8.1f us  dur %8.1f us  gap %7.1f  grid %s" % (n, s / 1e3, (e - s) / 1e3, (int(r['Start_Timestamp']) - prev_end) / 1e3, r.get('Grid_Size_X', '') + "x" + r.get('Grid_Size_Y', '') + "x" + r.get('Grid_Size_Z', '')))
    prev_end = max(prev_end, int(r['End_Timestamp']))
PY
  echo "== $mode"; cut -c1-120 "$GRAFT_REPO_ROOT/$O/timeline_$mode.txt" | head -50
done
