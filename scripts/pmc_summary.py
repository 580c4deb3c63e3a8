#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel name: sum of each counter, dispatch count."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
files = glob.glob(root + "/**/*counter_collection.csv", recursive=True)
agg = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ie::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)          # kernel names contain commas (template arguments): quote them
w.writerow(["kernel", "dispatches"] + names)
for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
    w.writerow([k, len(disp[k])] + [f"{agg[k].get(c, 0):.0f}" for c in names])
