#!/usr/bin/env python3
"""Kernel durations and the GAPS between consecutive nidreg kernels from a rocprofv3 --kernel-trace CSV
(…_kernel_trace.csv): where the wall clock of an evaluation goes beyond the kernels themselves.
Usage: trace_gaps.py <dir containing *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

import numpy as np

paths = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        name = r.get("Kernel_Name", "")
        if "nidreg" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("nidreg::", "")
        short = short.split("<")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
dur = {}
gap = {}
for i, (s, e, n) in enumerate(rows):
    dur.setdefault(n, []).append((e - s) / 1e3)
    if i + 1 < len(rows):
        s2, e2, n2 = rows[i + 1]
        gap.setdefault(f"{n} -> {n2}", []).append((s2 - e) / 1e3)
print("kernel durations (us): median / mean / n")
for k, v in sorted(dur.items()):
    v = np.array(v[len(v) // 4:])  # drop the warm-up quarter
    print(f"  {k:24s} {np.median(v):8.2f} {v.mean():8.2f} {len(v):5d}")
print("gaps end->start (us): median / p10 / p90 / n")
for k, v in sorted(gap.items()):
    v = np.array(v[len(v) // 4:])
    if len(v) < 5:
        continue
    print(f"  {k:44s} {np.median(v):8.2f} {np.percentile(v, 10):8.2f} {np.percentile(v, 90):8.2f} {len(v):5d}")
# one evaluation = hist .. grad: span from hist start to the next hist start
starts = [s for s, e, n in rows if n == "k_spline_hist"]
if len(starts) > 8:
    per = np.diff(np.array(starts[len(starts) // 4:])) / 1e3
    print(f"hist start -> next hist start: median {np.median(per):.2f} us, p10 {np.percentile(per, 10):.2f}, p90 {np.percentile(per, 90):.2f}")
