#!/usr/bin/env python3
"""One line per nidreg kernel of a rocprofv3 `--kernel-trace --stats` CSV: short name, calls, average / min / max microseconds.
Usage: kernel_stats_brief.py <kernel_stats.csv>"""
import csv
import re
import sys

for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"]
    if "nidreg" not in name and "k_" not in name:
        continue
    short = re.sub(r"\(.*", "", name).replace("void ", "").replace("nidreg::", "").replace("(anonymous namespace)::", "")
    print(f"{short:70s} calls {int(row['Calls']):6d}  avg {float(row['AverageNs']) / 1e3:8.2f} us  min {float(row['MinNs']) / 1e3:8.2f}  max {float(row['MaxNs']) / 1e3:8.2f}")
