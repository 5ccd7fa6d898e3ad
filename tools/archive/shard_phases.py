#!/usr/bin/env python3
"""Where a sharded evaluation's time goes on ONE GPU (one-GPU protocol measurement): event times of the leader shard's phases
(histogram, k_entropy_repl = push / wait / entropy, gradient) for a 4096-point cloud, plain handle against n co-located
shards driven by their worker threads.  Usage: shard_phases.py [bins] [shards,shards,...]"""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ["NIDREG_SHARD_COLOCATED_WORKERS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shard_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3]
s = synth.make_scene("pinhole_1080p", num_points=4096, seed=5, device="cuda:0")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
rng = np.random.default_rng(3)
poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
out = {}
for n in shard_list:
    c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins, devices=None if n <= 1 else [0] * n)
    c.eval_batch(poses[:5])
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        c.eval_batch(poses)
        ts.append((time.perf_counter() - t0) / len(poses))
    row = {"us_per_eval": round(1e6 * float(np.median(ts)), 2)}
    for want_grad in (True, False):
        c.set_timing(True)
        acc = {}
        for x in poses:
            c(x, want_grad=want_grad)
            for k, v in c.timing_ms().items():
                acc.setdefault(k, []).append(v)
        c.set_timing(False)
        row["events_cost_grad" if want_grad else "events_cost_only"] = {k: round(1e3 * float(np.median(v[3:])), 2) for k, v in acc.items()}
    out[f"shards_{n}"] = row
    c.close()
print(json.dumps(out))
