#!/usr/bin/env python3
"""Collapse rocprofv3 counter_collection CSVs under a directory into per-kernel averages."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "nidreg" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
