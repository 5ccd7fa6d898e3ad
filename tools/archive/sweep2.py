#!/usr/bin/env python3
"""Sweep (flags, gw, target_blocks, precision) on a cached scene, torch-free.  Usage: sweep2.py scene.npz tag [spec ...]
spec = prec:bins:gw:tb:flags[:copies]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
tag = sys.argv[2]
specs = sys.argv[3:]
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
rng = np.random.default_rng(1)
poses = [se3.plus(z["T_true"], rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, 0.0087, 0.0087, 0.0087])) for _ in range(10)]
rows = []
ref_cost = None
for spec in specs:
    f = spec.split(":")
    prec, bins, gw, tb, flags = f[:5]
    copies = int(f[5]) if len(f) > 5 else 0
    c = nid.NIDCost(proj, img64, pts, ints, int(bins), precision=prec, columns_per_group=int(gw), target_blocks=int(tb), flags=int(flags), lds_copies=copies)
    c.set_timing(True)
    acc = {}
    for k, x in enumerate(poses):
        ok, cost, g = c(x)
        if k >= 2:
            for key, v in c.timing_ms().items():
                acc.setdefault(key, []).append(v)
    row = dict(spec=spec, chunks=c.info()["num_chunks"], cost=cost)
    row.update({k: round(float(np.mean(v)), 4) for k, v in acc.items()})
    rows.append(row)
    print(row, flush=True)
    c.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"sweep2_{tag}.json"), "w") as f:
    json.dump(rows, f, indent=1)
