#!/usr/bin/env python3
"""Construction of a cost object from a device-resident cloud (nidreg_create_from_cloud: view cull + keys + radix sort + gather + chunk
tables), K times on a cached scene: wall time per construction; meant to be wrapped by `rocprofv3 --kernel-trace --stats` for the
per-kernel split (the rocPRIM sort passes included).  Usage: build_profile.py scene.npz [repeats] [bins]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
cloud = nid.Cloud(pts, ints)
T = se3.to_matrix(z["T_true"])
ts, tn = [], []
for k in range(reps):
    t0 = time.perf_counter()
    c = nid.NIDCost.from_cloud(proj, img64, cloud, bins, cull=(T, 0.0, True))
    ts.append(time.perf_counter() - t0)
    n = c.info()["num_points"]
    c.close()
    t0 = time.perf_counter()
    c = nid.NIDCost.from_cloud(proj, img64, cloud, bins)
    tn.append(time.perf_counter() - t0)
    c.close()
print(json.dumps({"points": int(pts.shape[0]), "kept": int(n), "bins": bins, "create_culled_ms": [round(1e3 * t, 2) for t in ts], "create_plain_ms": [round(1e3 * t, 2) for t in tn],
                  "culled_median_ms": round(1e3 * float(np.median(ts[1:])), 2), "plain_median_ms": round(1e3 * float(np.median(tn[1:])), 2)}))
