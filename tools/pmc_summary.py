#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per launch per kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "trace" in k or "emit" in k:
                short = k.split("(")[0].split("::")[-1]
                agg[(short, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            print(f"{d:28s} {k:28s} {c:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
