#!/usr/bin/env python3
"""Kernel timeline of the last full decode in a rocprofv3 --kernel-trace csv: start, duration, gap to the previous kernel's end."""
import csv, glob, os, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'sync_scan_kernel' in r['Kernel_Name']]
start, stop = idx[-3], idx[-2]
t0 = int(rows[start]['Start_Timestamp'])
prev_end = t0
for r in rows[start:stop + 1]:
    n = r['Kernel_Name'].replace('dabhip::(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:30]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-30s start %8.1f us  dur %8.1f us  gap %7.1f  q=%s" % (n, s / 1e3, (e - s) / 1e3, (int(r['Start_Timestamp']) - prev_end) / 1e3, r.get('Queue_Id', '')))
    prev_end = max(prev_end, int(r['End_Timestamp']))
