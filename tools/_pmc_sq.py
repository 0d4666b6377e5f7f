#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch. Usage: _pmc_sq.py DIR [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "wgrad3"
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub not in k:
            continue
        k = k[:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    n = len(cnt[k])
    print(k, "dispatches", n)
    for c, x in sorted(v.items()):
        print("   %-28s %.4g" % (c, x / n))
