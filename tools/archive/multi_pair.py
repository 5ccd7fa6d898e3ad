#!/usr/bin/env python3
"""MultiNIDCost on ONE GPU: k pairs of N/k points each (a multi-bag dataset) evaluated through nidreg_eval_multi -- every
pair's histogram pass is queued before the rest, so one pair's entropy kernel and launch turnaround hide behind another
pair's streaming kernels -- against one pair of N points.  Usage: multi_pair.py [total_points]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import _lib, nid, synth  # noqa: E402

total = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
out = {}
lib = _lib.load()
tb_list = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for k, tb in [(k, tb) for k in (1, 2, 4, 8) for tb in tb_list]:
    scenes = [synth.make_scene("pinhole_1080p", num_points=total // k, seed=100 + i, device="cuda:0") for i in range(k)]
    proj = nid.create_camera(scenes[0].model, scenes[0].intrinsics, scenes[0].distortion)
    costs = [nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 256, target_blocks=tb) for s in scenes]
    rng = np.random.default_rng(3)
    poses = [synth.random_pose_near(scenes[0].T_camera_lidar_true, rng) for _ in range(40)]
    arr = (ctypes.c_void_p * k)(*[c.h for c in costs])
    cost = ctypes.c_double(0.0)
    grad = np.empty(7)
    gp = grad.ctypes.data_as(_lib.c_double_p)
    xs = [np.ascontiguousarray(p) for p in poses]
    xps = [x.ctypes.data_as(_lib.c_double_p) for x in xs]
    for x in xps[:5]:
        lib.nidreg_eval_multi(arr, k, None, x, ctypes.byref(cost), gp)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        for x in xps:
            lib.nidreg_eval_multi(arr, k, None, x, ctypes.byref(cost), gp)
        ts.append((time.perf_counter() - t0) / len(xps))
    out[f"pairs_{k}_tb{tb}"] = {"points_per_pair": total // k, "us_per_multi_eval": round(1e6 * float(np.median(ts)), 2), "point_evals_per_s": round(total / float(np.median(ts)) / 1e9, 3)}
    for c in costs:
        c.close()
print(json.dumps(out))
