#!/usr/bin/env python3
"""A/B of the chunk rule on a VIEW-CULLED cloud (the clouds `calibrate` really evaluates: visual_camera_calibration.cpp:201-206
culls before every inner solve).  The rank equalisation makes the columns of the whole cloud equally full; the culled
subset's columns are not, and the chunk rule of rounds 1-3 could give its tables a few workgroups more than one round holds
(DESIGN.md section 4).  Builds the cost object with nidreg_create_from_cloud (device-resident cull + build) under
NIDREG_CHUNKS_NO_FIT=1 (old rule) and without (split_groups' growth step), prints table sizes and microseconds per evaluation.
Usage: culled_cloud_ab.py [points] [camera]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
camera = sys.argv[2] if len(sys.argv) > 2 else "pinhole_1080p"
s = synth.make_scene(camera, num_points=n, seed=20250525, device="cuda:0")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
# a cloud that covers more than the view, so that culling removes a non-uniform part of every column: the scene's points
# plus a copy pushed sideways out of the image
pts = np.concatenate([s.points, s.points + np.array([6.0, 0.0, 0.0, 0.0])])
ints = np.concatenate([s.intensities, s.intensities[::-1]])
cloud = nid.Cloud(pts, ints)
T = se3.to_matrix(s.T_camera_lidar_init)
rng = np.random.default_rng(5)
poses = [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)]
out = {}
for label, env in (("old_rule", "1"), ("fit_one_round", None)):
    if env:
        os.environ["NIDREG_CHUNKS_NO_FIT"] = env
    else:
        os.environ.pop("NIDREG_CHUNKS_NO_FIT", None)
    cost = nid.NIDCost.from_cloud(proj, s.image_f64, cloud, 256, cull=(T, 0.0, True))
    info = cost.info()
    for x in poses[:5]:
        cost(x)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        for x in poses:
            cost(x)
        ts.append((time.perf_counter() - t0) / len(poses))
    out[label] = {"kept_points": int(info.get("num_points", -1)), "num_chunks": int(info.get("num_chunks", -1)), "us_per_eval": round(1e6 * float(np.median(ts)), 2)}
    cost.close()
print(json.dumps(out))
