#!/usr/bin/env python3
"""What constructing the cost object costs for small clouds (the reference builds a new NIDCost per pair in every outer
iteration, visual_camera_calibration.cpp:199-208): milliseconds per construction + destruction from host arrays and from a
device-resident cloud (with the view cull), beside the time of one evaluation.  Usage: setup_cost.py [bins] [points,...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3, synth  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sizes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [30_000, 100_000, 1_000_000]
for n in sizes:
    s = synth.make_scene("pinhole_vga", num_points=n, seed=3, device="cuda:0")
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    x = s.T_camera_lidar_init
    T = se3.to_matrix(x)
    row = {"points": n, "bins": bins}
    for label in ("host_arrays", "device_cloud_cull"):
        cloud = nid.Cloud(s.points, s.intensities, device=0) if label == "device_cloud_cull" else None
        ts, te = [], []
        for it in range(30):
            t0 = time.perf_counter()
            if cloud is None:
                c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins)
            else:
                c = nid.NIDCost.from_cloud(proj, s.image_f64, cloud, bins, cull=(T, 0.1, False))
            t1 = time.perf_counter()
            c(x)
            t2 = time.perf_counter()
            c.close()
            t3 = time.perf_counter()
            ts.append((t1 - t0) + (t3 - t2))
            te.append(t2 - t1)
        row[label] = {"create_destroy_ms": round(1e3 * float(np.median(ts[5:])), 3), "first_eval_ms": round(1e3 * float(np.median(te[5:])), 3)}
        if cloud is not None:
            cloud.close()
    print(json.dumps(row), flush=True)
