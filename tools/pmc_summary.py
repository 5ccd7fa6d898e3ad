#!/usr/bin/env python3
"""Collapse rocprofv3 counter_collection CSVs under a directory into per-kernel averages.  Where the CSV carries the dispatch's
timestamps, the mean duration of the kernel IN THAT PASS is printed per counter too (`dur_us`): a counter such as GRBM_GUI_ACTIVE
(cycles, summed over the XCDs) is only meaningful against the duration of the pass that collected it (profiled passes run slower
and at another clock than un-profiled ones)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "nidreg" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
            try:
                dur[short][row["Counter_Name"]].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3)
            except (KeyError, ValueError, TypeError):
                pass
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        t = dur[k].get(c)
        extra = f"  dur_us={sum(t) / len(t):.2f}" if t else ""
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}{extra}")
