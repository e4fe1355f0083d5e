#!/usr/bin/env python3
"""Mean per-launch counters from a rocprofv3 --pmc CSV, grouped by kernel name."""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            m = re.search(r"(trace_\w+<[^>]*>|emit_kernel)", r["Kernel_Name"])
            if m:
                agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            print(d.split("/")[-1], k, {c: float("%.4g" % (sum(v) / len(v))) for c, v in cs.items()})
