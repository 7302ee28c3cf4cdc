#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel (the profiles/rNN_pmc_summary.csv format that bench.py reads).

usage: sq_pmc_summary.py [--meta key=value ...] pass_name=dir [pass_name=dir ...] > summary.csv

One row per (pass, kernel, counter): the counter summed over all dispatches of the kernel in that pass.  A pass collected together
with --kernel-trace also gets a row DURATION_NS (sum of End - Start over the same dispatches), so that a rate or the effective clock
(GRBM_GUI_ACTIVE / XCDs / DURATION_NS) can be formed from two cells of one pass.  --meta rows ("meta,<key>,0,VALUE,<value>") record
what the profiled command was (decodes in the run, wave-steps per decode ...)."""
import csv
import glob
import sys
from collections import defaultdict


def short(name):
    return name.replace("void ", "").replace("dabhip::(anonymous namespace)::", "").split("(")[0]


def main(argv):
    print("pass,kernel,dispatches,counter,sum")
    it = iter(argv)
    for arg in it:
        if arg == "--meta":
            k, v = next(it).split("=", 1)
            print("meta,%s,0,VALUE,%s" % (k, v))
            continue
        name, d = arg.split("=")
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        trace = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
        dur = {}
        if trace:
            for r in csv.DictReader(open(trace[0])):
                dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        acc, ndisp = defaultdict(float), defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in ndisp[k]:
                ndisp[k].add(r["Dispatch_Id"])
                if dur:
                    acc[(k, "DURATION_NS")] += dur.get(r["Dispatch_Id"], 0)
        for (k, c), v in sorted(acc.items()):
            print("%s,%s,%d,%s,%.1f" % (name, k, len(ndisp[k]), c, v))


if __name__ == "__main__":
    main(sys.argv[1:])
