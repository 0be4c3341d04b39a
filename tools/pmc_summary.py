#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean of each counter per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for name, v in sorted(c.items()):
        print("   %-28s n=%-4d mean=%.6g" % (name, len(v), sum(v) / len(v)))
