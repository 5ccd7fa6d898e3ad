#!/usr/bin/env python3
"""torch-free driver: load a cached scene (.npz), run K cost+Jacobian evaluations through the C ABI and
print per-kernel HIP-event times.  Meant to be wrapped by rocprofv3 (--kernel-trace / --pmc).
Usage: run_scene.py scene.npz [steps] [precision] [bins] [gw] [target_blocks]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
prec = sys.argv[3] if len(sys.argv) > 3 else "fp64"
bins = int(sys.argv[4]) if len(sys.argv) > 4 else 256
gw = int(sys.argv[5]) if len(sys.argv) > 5 else 0
tb = int(sys.argv[6]) if len(sys.argv) > 6 else 0
flags = int(sys.argv[7]) if len(sys.argv) > 7 else 0
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
cost = nid.NIDCost(proj, img64, pts, ints, bins, precision=prec, columns_per_group=gw, target_blocks=tb, flags=flags)
import time  # noqa: E402

poses = [se3.plus(z["T_true"], np.random.default_rng(7).uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, 0.0087, 0.0087, 0.0087])) for _ in range(8)]
for x in poses[:3]:
    cost(x)
t0 = time.perf_counter()
NW = 8 if os.environ.get("RUN_SCENE_QUICK") else 200  # RUN_SCENE_QUICK=1: counter passes (every kernel serialised) need launches, not statistics
for k in range(NW):
    cost(poses[k & 7])
wall_ms = (time.perf_counter() - t0) * 1e3 / NW
# the same through nidreg_eval_batch (no Python between the evaluations: what bench.py times), median of 5 blocks
block = np.ascontiguousarray([poses[k & 7] for k in range(50)])
wb = []
for _ in range(1 if os.environ.get("RUN_SCENE_QUICK") else 5):
    t0 = time.perf_counter()
    cost.eval_batch(block[:8] if os.environ.get("RUN_SCENE_QUICK") else block)
    wb.append((time.perf_counter() - t0) * 1e3 / (8 if os.environ.get("RUN_SCENE_QUICK") else len(block)))
wall_batch_ms = float(np.median(wb))
# the whole evaluation between two HIP events
cost.set_timing(2)
whole = []
for k in range(max(steps, 10)):
    cost(poses[k & 7])
    if k >= 2:
        whole.append(cost.timing_ms()["total"])
whole_ms = float(np.mean(whole))
cost.set_timing(True)
rng = np.random.default_rng(1)
acc = {}
for k in range(steps):
    d = rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, np.radians(0.5), np.radians(0.5), np.radians(0.5)])
    ok, c, g = cost(se3.plus(z["T_true"], d))
    if k >= 2:
        for key, v in cost.timing_ms().items():
            acc.setdefault(key, []).append(v)
print(json.dumps({"prec": prec, "bins": bins, "info": cost.info(), "kernel_ms": {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "wall_ms": round(wall_ms, 4), "wall_batch_ms": round(wall_batch_ms, 4), "whole_eval_event_ms": round(whole_ms, 4), "last_cost": c, "last_grad": [float(v) for v in g], "evals_per_s": round(1e3 / float(np.mean(acc["total"])), 1) if "total" in acc else None}))
