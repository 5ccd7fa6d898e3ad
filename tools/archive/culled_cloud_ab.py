#!/usr/bin/env python3
"""A/B of the chunk rule on a VIEW-CULLED cloud (the clouds `calibrate` really evaluates: visual_camera_calibration.cpp:201-206
culls before every inner solve).  The rank equalisation makes the columns of the whole cloud equally full; the culled
subset's columns are not.  Builds the cost object with nidreg_create_from_cloud (device-resident cull + build) with chunk
tables of one column group per chunk (NIDREG_MAX_SEGS=1: the constraint of rounds 1-3) and with chunks that may run across
groups (round 4, csrc/nidreg_plan.hip split_groups), prints table sizes, microseconds per evaluation, per-kernel event times and
nanoseconds per kept point.
`skew`: the pushed-out copy carries darker surfaces (intensity^2) and the intensities are rank-equalised over the WHOLE cloud,
as preprocess.cpp:464-473 does for a whole map: the culled subset's histogram columns are then far from equally full.
Usage: culled_cloud_ab.py [points] [camera] [shift_m] [skew]"""
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
shift = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
skew = len(sys.argv) > 4 and sys.argv[4] == "skew"
s = synth.make_scene(camera, num_points=n, seed=20250525, device="cuda:0")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
# a cloud that covers more than the view, so that culling removes a non-uniform part of every column: the scene's points
# plus a copy pushed sideways out of the image (float32-representable like PLY data, so the records stay 16 bytes)
moved = (s.points + np.array([shift, 0.0, 0.0, 0.0])).astype(np.float32).astype(np.float64)
pts = np.concatenate([s.points, moved])
ints = np.concatenate([s.intensities, s.intensities[::-1]])
if skew:
    raw = np.concatenate([s.intensities, s.intensities[::-1] ** 2])
    order = np.argsort(raw, kind="stable")
    ranks = np.empty(raw.shape[0], dtype=np.int64)
    ranks[order] = np.arange(raw.shape[0])
    ints = np.floor(256.0 * ranks / raw.shape[0]) / 256.0
cloud = nid.Cloud(pts, ints)
T = se3.to_matrix(s.T_camera_lidar_init)
rng = np.random.default_rng(5)
poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
out = {"camera": camera, "input_points": int(pts.shape[0]), "skew": bool(skew)}
ref = None
for label, env in (("one_group_per_chunk", "1"), ("chunks_across_groups", None)):
    if env:
        os.environ["NIDREG_MAX_SEGS"] = env
    else:
        os.environ.pop("NIDREG_MAX_SEGS", None)
    cost = nid.NIDCost.from_cloud(proj, s.image_f64, cloud, 256, cull=(T, 0.0, True))
    info = cost.info()
    cost.eval_batch(poses[:5])
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        ok, cs, gs = cost.eval_batch(poses)
        ts.append((time.perf_counter() - t0) / len(poses))
    cost.set_timing(True)
    acc = {}
    for x in poses[:12]:
        cost(x)
        for key, v in cost.timing_ms().items():
            acc.setdefault(key, []).append(v)
    cost.set_timing(False)
    kept = int(info.get("num_points", -1))
    us = 1e6 * float(np.median(ts))
    out[label] = {"kept_points": kept, "record_bytes": int(info["record_bytes"]), "num_chunks": int(info.get("num_chunks", -1)), "segmented": [int(info["segmented"]), int(info["segmented_hist"])],
                  "us_per_eval": round(us, 2), "ns_per_point": round(1e3 * us / max(kept, 1), 3), "kernel_us": {k: round(1e3 * float(np.mean(v[2:])), 2) for k, v in acc.items()}}
    if ref is None:
        ref = (cs.copy(), gs.copy())
    else:
        out["cost_bits_equal"] = bool(np.array_equal(ref[0], cs))
        out["grad_max_rel_diff"] = float(np.max(np.abs(ref[1] - gs) / (np.abs(ref[1]) + 1e-300)))
    cost.close()
print(json.dumps(out))
