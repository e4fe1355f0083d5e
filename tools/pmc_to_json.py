#!/usr/bin/env python3
"""rocprofv3 --pmc output directories -> one JSON: mean per launch of every counter, per kernel.
usage: python tools/pmc_to_json.py OUT.json DIR [DIR ...]   (the profiles/rNN/vK_pmc_summary.json files)"""
import collections
import csv
import glob
import json
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(trace_queue_kernel|trace_kernel|fold_kernel|emit_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
           for k, cs in sorted(agg.items()) if k.startswith(("trace", "fold", "emit"))}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
