#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/pmc_passes.sh) per kernel: sum of every counter over all dispatches, plus
dispatch count and total duration from the same pass.  usage: tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
dur = collections.defaultdict(float)
for f in sorted(glob.glob(d + "/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if filt and filt not in r["Kernel_Name"]:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in disp[k]:
            disp[k].add(key)
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print("   %-44s %.6g" % (c, v))
