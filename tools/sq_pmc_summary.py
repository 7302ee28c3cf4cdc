#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel (the profiles/rNN_sq_pmc_summary.csv format).
usage: sq_pmc_summary.py pass_name=dir [pass_name=dir ...] > summary.csv"""
import csv
import glob
import sys
from collections import defaultdict

print("pass,kernel,dispatches,counter,sum")
for arg in sys.argv[1:]:
    name, d = arg.split("=")
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, ndisp = defaultdict(float), defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("dabhip::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    for (k, c), v in sorted(acc.items()):
        print("%s,%s,%d,%s,%.1f" % (name, k, len(ndisp[k]), c, v))
