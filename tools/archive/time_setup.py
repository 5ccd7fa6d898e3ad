#!/usr/bin/env python3
"""Handle-creation time: host path (CPU view culling by the oracle + host bucketing + H2D) vs the
device-resident path (cloud uploaded once; cull + bucket + sort + gather on the GPU).
Usage: time_setup.py scene.npz [bins]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
model, intr, dist = str(z["model"]), list(z["intrinsics"]), list(z["distortion"])
proj = nid.create_camera(model, intr, dist)
img8 = z["image_u8"]
img64 = img8.astype(np.float64) * (1.0 / 255.0)
W, H = int(z["width"]), int(z["height"])
T = se3.to_matrix(z["T_init"])
out = {"points": int(pts.shape[0])}
t0 = time.perf_counter()
min_z = float(np.cos(nid.estimate_camera_fov(proj, (W, H))))
out["estimate_fov_s"] = round(time.perf_counter() - t0, 4)

t0 = time.perf_counter()
cloud = nid.Cloud(pts, ints)
out["cloud_upload_s"] = round(time.perf_counter() - t0, 4)
ts = []
for k in range(4):
    t0 = time.perf_counter()
    c = nid.NIDCost.from_cloud(proj, img64, cloud, bins, cull=(T, min_z, True))
    ts.append(time.perf_counter() - t0)
    kept = c.num_points
    ok, cd, gd = c(z["T_init"])
    c.close()
out["device_cull_build_s"] = round(min(ts), 4)
out["kept"] = int(kept)

vc = nid.ViewCulling(proj, (W, H), min_z=min_z)
t0 = time.perf_counter()
idx = vc.cull(pts, T)
out["gpu_cull_host_io_s"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter()
ph, ih = np.ascontiguousarray(pts[idx]), np.ascontiguousarray(ints[idx])
out["host_sample_s"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter()
c = nid.NIDCost(proj, img64, ph, ih, bins)
out["host_build_s"] = round(time.perf_counter() - t0, 4)
ok, ch, gh = c(z["T_init"])
c.close()
out["same_cost"] = bool(cd == ch)
if pts.shape[0] <= 2_000_000:
    import oracle_lib

    t0 = time.perf_counter()
    ridx = oracle_lib.view_culling(model, intr, dist, W, H, pts, T, True)
    out["cpu_cull_s"] = round(time.perf_counter() - t0, 4)
    out["cull_identical"] = bool(np.array_equal(ridx, idx))
print(json.dumps(out))
