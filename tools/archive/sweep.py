#!/usr/bin/env python3
"""Kernel-time sweep over tiling / precision knobs on one resident 10M-point workload (GPU box).
Writes gpurun_out/sweep_<tag>.json.  Usage: python tools/sweep.py [tag] [points]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "default"
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
camera = sys.argv[3] if len(sys.argv) > 3 else "pinhole_1080p"
scene = synth.make_scene(camera, num_points=npts, seed=20250525, device="cuda:0")
proj = nid.create_camera(scene.model, scene.intrinsics, scene.distortion)
rng = np.random.default_rng(1)
poses = [synth.random_pose_near(scene.T_camera_lidar_true, rng) for _ in range(12)]
rows = []
configs = []
for prec in ("fp64",):
    for bins in (256, 16):
        for gw in (1, 4, 16, 32, 64):
            if gw > bins:
                continue
            for tb in (1024, 2048, 4096):
                configs.append((prec, bins, gw, tb))
if os.environ.get("SWEEP_SMALL"):
    configs = [c for c in configs if c[3] == 2048]
for prec, bins, gw, tb in configs:
    try:
        c = nid.NIDCost(proj, scene.image_f64, scene.points, scene.intensities, bins, precision=prec, columns_per_group=gw, target_blocks=tb)
    except Exception as e:  # noqa: BLE001
        rows.append(dict(prec=prec, bins=bins, gw=gw, tb=tb, error=str(e)))
        continue
    c.set_timing(True)
    acc = {}
    t_wall = []
    for k, x in enumerate(poses):
        t0 = time.perf_counter()
        ok, cost, g = c(x)
        t_wall.append(time.perf_counter() - t0)
        if k >= 2:
            for key, v in c.timing_ms().items():
                acc.setdefault(key, []).append(v)
    row = dict(prec=prec, bins=bins, gw=gw, tb=tb, chunks=c.info()["num_chunks"], wall_ms=round(1e3 * float(np.median(t_wall[2:])), 4))
    row.update({k: round(float(np.mean(v)), 4) for k, v in acc.items()})
    rows.append(row)
    print(row, flush=True)
    c.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"sweep_{tag}.json"), "w") as f:
    json.dump(rows, f, indent=1)
