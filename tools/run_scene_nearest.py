#!/usr/bin/env python3
"""torch-free driver of the NEAREST twin (CostCalculatorNID::calculate, cost_calculator_nid.cpp:21-67): load a cached scene
(.npz), run K evaluations through the C ABI, print the wall time per evaluation.  Meant to be wrapped by rocprofv3
(--kernel-trace / --pmc), like run_scene.py for the SPLINE path.
Usage: run_scene_nearest.py scene.npz [steps] [bins]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 256
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
calc = nid.CostCalculatorNID(proj, z["image_u8"], pts, ints, nid.NIDCostParams(bins))
rng = np.random.default_rng(7)
mats = [se3.to_matrix(se3.plus(z["T_true"], rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, 0.0087, 0.0087, 0.0087]))) for _ in range(8)]
for m in mats[:3]:
    c = calc.calculate(m)
t0 = time.perf_counter()
for k in range(steps):
    c = calc.calculate(mats[k & 7])
wall_ms = (time.perf_counter() - t0) * 1e3 / steps
print(json.dumps({"mode": "nearest", "bins": bins, "steps": steps, "wall_ms": round(wall_ms, 4), "evals_per_s": round(1e3 / wall_ms, 1), "last_cost": c, "info": calc.info()}))
