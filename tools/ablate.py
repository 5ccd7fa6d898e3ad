#!/usr/bin/env python3
"""Phase ablation of the two streaming kernels (needs `make -C csrc libnidreg_ablate.so`).
For each mask the kernel is run with that phase compiled out; the time that disappears is the
phase's marginal cost.  Usage: ablate.py scene.npz [precision] [bins] [gw] [tb]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NIDREG_LIB"] = os.path.join(ROOT, "direct_visual_lidar_calibration_amd", "csrc", "libnidreg_ablate.so")
from direct_visual_lidar_calibration_amd import nid, se3  # noqa: E402

z = np.load(sys.argv[1])
prec = sys.argv[2] if len(sys.argv) > 2 else "fp64"
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gw = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tb = int(sys.argv[5]) if len(sys.argv) > 5 else 0
flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
cost = nid.NIDCost(proj, img64, pts, ints, bins, precision=prec, columns_per_group=gw, target_blocks=tb, flags=flags)
cost.set_timing(True)
rng = np.random.default_rng(1)
poses = [se3.plus(z["T_true"], rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, 0.0087, 0.0087, 0.0087])) for _ in range(8)]
names_h = {0: "full", 1: "-atomics", 2: "-imgloads", 3: "-atomics-imgloads", 4: "-projection", 5: "-atomics-projection", 6: "-imgloads-projection", 7: "-atomics-imgloads-projection", 8: "-flush",
           9: "-atomics-flush", 15: "only load+transform+spline"}
names_g = {0: "full", 1: "-Greads", 2: "-imgloads", 3: "-Greads-imgloads", 4: "-projection", 5: "-Greads-projection", 6: "-imgloads-projection", 7: "only load+transform+spline+accum"}
out = {"prec": prec, "bins": bins, "info": cost.info(), "hist": {}, "grad": {}}
for m, name in names_h.items():
    os.environ["NIDREG_ABLATE_HIST"] = str(m)
    os.environ["NIDREG_ABLATE_GRAD"] = "0"
    ts = []
    for x in poses:
        cost(x)
        ts.append(cost.timing_ms()["hist"])
    out["hist"][name] = round(float(np.mean(ts[2:])), 4)
    print("hist", m, name, out["hist"][name], flush=True)
os.environ["NIDREG_ABLATE_HIST"] = "0"
for m, name in names_g.items():
    os.environ["NIDREG_ABLATE_GRAD"] = str(m)
    ts = []
    for x in poses:
        cost(x)
        ts.append(cost.timing_ms()["grad"])
    out["grad"][name] = round(float(np.mean(ts[2:])), 4)
    print("grad", m, name, out["grad"][name], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", f"ablate_{prec}_{bins}.json"), "w") as f:
    json.dump(out, f, indent=1)
