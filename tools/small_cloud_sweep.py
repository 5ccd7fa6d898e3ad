#!/usr/bin/env python3
"""Small clouds (configs[0] is 100k points, VGA pinhole, 16 bins): microseconds per synchronous cost+Jacobian evaluation
against the number of chunks -- the rule that sizes chunk tables (nidreg_plan.hip round_chunks) is set from this table.
Usage: small_cloud_sweep.py [bins,bins,...] [points,points,...] [target_blocks,...] [nearest]   (target_blocks 0 = the library's rule;
"nearest": the NEAREST twin, CostCalculatorNID::calculate, cost only)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, se3, synth  # noqa: E402


def ints(i, default):
    return [int(v) for v in sys.argv[i].split(",")] if len(sys.argv) > i else default


bins_list = ints(1, [16])
points_list = ints(2, [100_000])
tb_list = ints(3, [0, 32, 64, 128, 256, 512, 1024, 2048])
nearest = len(sys.argv) > 4 and sys.argv[4] == "nearest"
rng = np.random.default_rng(3)
for n in points_list:
    s = synth.make_scene("pinhole_vga", num_points=n, seed=20250523 + 7, device="cuda:0")
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
    for bins in bins_list:
        row = {"points": n, "bins": bins, "us_per_eval": {}, "chunks": {}}
        for tb in tb_list:
            print("points", n, "bins", bins, "target_blocks", tb, file=sys.stderr, flush=True)
            if nearest:
                c = nid.CostCalculatorNID(proj, s.image_u8, s.points, s.intensities, nid.NIDCostParams(bins), target_blocks=tb)
                mats = np.ascontiguousarray([se3.to_matrix(x) for x in poses])
                for m in mats[:5]:
                    c.calculate(m)
                ts = []
                for _ in range(9):
                    t0 = time.perf_counter()
                    for m in mats:
                        c.calculate(m)
                    ts.append((time.perf_counter() - t0) / len(mats))
                row["us_per_eval"][str(tb)] = round(1e6 * float(np.median(ts)), 2)
                row["chunks"][str(tb)] = c.info()["num_chunks"]
                c.close()
                continue
            c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins, target_blocks=tb)
            c.eval_batch(poses[:5])
            ts = []
            for _ in range(9):
                t0 = time.perf_counter()
                c.eval_batch(poses)
                ts.append((time.perf_counter() - t0) / len(poses))
            row["us_per_eval"][str(tb)] = round(1e6 * float(np.median(ts)), 2)
            row["chunks"][str(tb)] = c.info()["num_chunks"]
            row.setdefault("fused", {})[str(tb)] = [c.info()["fused"], c.info()["fused_full_stash"]]  # one launch per evaluation (nid_fused.hpp), stash format
            if tb == 0:
                row["segmented"] = [c.info()["segmented"], c.info()["segmented_hist"]]
                c.set_timing(True)
                acc = {}
                for x in poses[:12]:
                    c(x)
                    for k, v in c.timing_ms().items():
                        acc.setdefault(k, []).append(v)
                row["kernel_us_rule"] = {k: round(1e3 * float(np.mean(v[2:])), 2) for k, v in acc.items()}
            c.close()
        print(json.dumps(row), flush=True)
