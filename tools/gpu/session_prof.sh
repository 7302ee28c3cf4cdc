#!/bin/bash
# steady state of a device-fed session: plain, with prefetch, and the kernel + copy timeline of one segment
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/session
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/session_steady.py > $O/plain.json 2> $O/err.txt
python3 $R/tools/session_steady.py --prefetch > $O/prefetch.json 2>> $O/err.txt
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace -- python3 $R/tools/session_steady.py --segments 3 > /dev/null 2>> $O/err.txt
python3 $R/tools/timeline.py $O/trace > $O/timeline.txt 2>> $O/err.txt
python3 - $O/trace <<'PY' > $O/copies.txt
import csv, glob, sys, os
fs = glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    print(len(rows), "copies; columns", list(rows[0].keys()) if rows else None)
    tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
    print("sum of durations ms", tot / 1e6)
PY
rm -rf $O/trace
cat $O/plain.json $O/prefetch.json; tail -5 $O/err.txt; cat $O/copies.txt
