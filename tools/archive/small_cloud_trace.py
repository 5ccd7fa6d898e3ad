#!/usr/bin/env python3
"""Driver for `rocprofv3 --kernel-trace --stats`: K synchronous cost+Jacobian evaluations of a small cloud (default configs[0]:
100k points, VGA pinhole, 16 bins), so that the kernels' own durations can be set against the wall time per evaluation it
prints.  Usage: small_cloud_trace.py [points] [bins] [evals]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 16
evals = int(sys.argv[3]) if len(sys.argv) > 3 else 400
s = synth.make_scene("pinhole_vga", num_points=n, seed=20250523 + 7, device="cpu")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
rng = np.random.default_rng(3)
poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins)
c.eval_batch(poses[:5])
ts = []
for _ in range(max(1, evals // len(poses))):
    t0 = time.perf_counter()
    c.eval_batch(poses)
    ts.append((time.perf_counter() - t0) / len(poses))
t0 = time.perf_counter()
for _ in range(max(1, evals // len(poses))):
    c.eval_batch(poses, want_grad=False)
cost_only = (time.perf_counter() - t0) / (max(1, evals // len(poses)) * len(poses))
print(json.dumps({"points": n, "bins": bins, "chunks": c.info()["num_chunks"], "us_per_eval": round(1e6 * float(np.median(ts)), 2), "us_per_eval_cost_only": round(1e6 * cost_only, 2)}))
c.close()
