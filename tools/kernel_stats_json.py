#!/usr/bin/env python3
"""profiles/<tag>_kernel_stats.json from a rocprofv3 --kernel-trace --stats run: average duration per nidreg kernel,
stamped with the kernel-source hash and the workload like *_traffic.json, so that bench.py can quote rocprof kernel
durations (no HIP-event markers inside them) only when they were measured on the kernel build it runs.
Usage: kernel_stats_json.py <kernel_stats.csv> <out.json> points width height bins precision [note] [camera]"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from direct_visual_lidar_calibration_amd import _lib  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
points, width, height, bins = (int(v) for v in sys.argv[3:7])
precision = sys.argv[7]
note = sys.argv[8] if len(sys.argv) > 8 else ""
camera = sys.argv[9] if len(sys.argv) > 9 else "pinhole_1080p"  # a key of synth.CONFIG_CAMERAS
kernels = {}
for r in csv.DictReader(open(src)):
    m = re.search(r"nidreg::(k_\w+)", r["Name"])
    if not m:
        continue
    k = kernels.setdefault(m.group(1), {"calls": 0, "total_ns": 0.0, "min_ns": None, "max_ns": None, "instantiations": []})
    calls = int(r["Calls"])
    k["calls"] += calls
    k["total_ns"] += float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * calls
    k["min_ns"] = float(r["MinNs"]) if k["min_ns"] is None else min(k["min_ns"], float(r["MinNs"]))
    k["max_ns"] = float(r["MaxNs"]) if k["max_ns"] is None else max(k["max_ns"], float(r["MaxNs"]))
    k["instantiations"].append({"name": r["Name"][:160], "calls": calls, "avg_ns": float(r["AverageNs"])})
for k in kernels.values():
    k["avg_ns"] = k["total_ns"] / max(k["calls"], 1)
json.dump({"source": src, "note": note, "kernel_build": _lib.stamp_or_refuse(), "workload": dict(points=points, width=width, height=height, bins=bins, precision=precision, camera=camera), "kernels": kernels},
          open(dst, "w"), indent=1)
print(json.dumps({k: round(v["avg_ns"] / 1e3, 2) for k, v in kernels.items()}))
