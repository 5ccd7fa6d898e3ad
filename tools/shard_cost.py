#!/usr/bin/env python3
"""Protocol cost of the in-library sharded evaluation on ONE GPU (the same device listed n times, every shard driven by its
own host thread on its own hardware queue like shards on different devices): with a cloud so small that the kernels' work
is negligible, the time per evaluation beyond the plain handle's is the protocol -- k_entropy_repl's push of the owned column
blocks into every replica, its one flag wait, the host-thread hand-off.
Also the 10M-point case for the record (there the shards' kernels share the one GPU, so nothing is gained -- it only shows
the protocol at full size).
Usage: shard_cost.py [bins] [--tiny-only]"""
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # co-located shards driven by worker threads must not share an in-order hardware queue
os.environ["NIDREG_SHARD_COLOCATED_WORKERS"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

bins = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tiny_only = "--tiny-only" in sys.argv  # bench.py's shard_proxy leg: the protocol cost alone, 2 and 3 shards
out = {}
for label, n_points in ((("tiny", 4096),) if tiny_only else (("tiny", 4096), ("10M", 10_000_000))):
    s = synth.make_scene("pinhole_1080p", num_points=n_points, seed=5, device="cuda:0")
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    rng = np.random.default_rng(3)
    poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
    row = {}
    for n in ((1, 2, 3) if tiny_only else (1, 2, 3, 4, 8)):
        c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins, devices=None if n <= 1 else [0] * n)
        c.eval_batch(poses[:5])
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            ok, costs, grads = c.eval_batch(poses)
            ts.append((time.perf_counter() - t0) / len(poses))
        ts2 = []
        for _ in range(15):
            t0 = time.perf_counter()
            c.eval_batch(poses, want_grad=False)
            ts2.append((time.perf_counter() - t0) / len(poses))
        row["plain_three_kernels" if n == 1 else f"shards_{n}"] = {"us_per_eval_cost_grad": round(1e6 * float(np.median(ts)), 2), "us_per_eval_cost_only": round(1e6 * float(np.median(ts2)), 2), "cost0": float(costs[0])}
        c.close()
    out[label] = row
print(json.dumps(out))
