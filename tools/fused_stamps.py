#!/usr/bin/env python3
"""Per-phase timeline of the one-launch evaluation (k_fused): needs a library built with -DNID_FUSED_STAMP
(tools/build_variants.sh stamp="-DNID_FUSED_STAMP"; NIDREG_LIB=variants/libnidreg_stamp.so).  Every workgroup stamps the
100 MHz wall clock at its phase boundaries; prints, per boundary, when the first / mean / last workgroup passes it
(microseconds after the first workgroup started) and the mean time spent in each phase.
Usage: fused_stamps.py scene.npz [evals]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import _lib, nid, se3  # noqa: E402

z = np.load(sys.argv[1])
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
cost = nid.NIDCost(proj, img64, pts, ints, int(sys.argv[3]) if len(sys.argv) > 3 else 256)
info = cost.info()
lib = _lib.load()
lib.nidreg_debug_fused_stamps.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
rng = np.random.default_rng(7)
names = ["start", "hist done", "barrier 1 passed", "entropy share done", "barrier 2 passed", "scalars + G tile done", "gradient loop done", "partials stored"]
rows = []
for k in range(evals):
    d = rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, np.radians(0.5), np.radians(0.5), np.radians(0.5)])
    cost(se3.plus(z["T_true"], d))
    buf = (ctypes.c_uint64 * (8 * 2048))()
    rc = lib.nidreg_debug_fused_stamps(buf, 8 * 2048)
    assert rc == 0, rc
    st = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 8).astype(np.int64)
    st = st[st[:, 0] > 0]
    st = st[st[:, 7] >= st[:, 0]]
    if k >= 2:
        rows.append((st - st[:, 0].min()) * 0.01)  # us
nwg = rows[0].shape[0]
print(json.dumps({"workgroups_stamped": int(nwg), "info": info}))
print(f"{'boundary':<26}{'first':>9}{'mean':>9}{'last':>9}   mean time in the phase before it [us]")
for j, nm in enumerate(names):
    first = np.mean([r[:, j].min() for r in rows])
    mean = np.mean([r[:, j].mean() for r in rows])
    last = np.mean([r[:, j].max() for r in rows])
    dur = np.mean([(r[:, j] - r[:, j - 1]).mean() for r in rows]) if j else 0.0
    print(f"{nm:<26}{first:9.2f}{mean:9.2f}{last:9.2f}   {dur:9.2f}")
# where the spread comes from: mean duration of the two streaming phases by dispatch slot (first / second workgroup of a
# CU: the grid is dispatched in index order, 256 CUs), by XCD (index mod 8) and by column pair
for name, j in (("hist phase", 1), ("gradient loop", 6)):
    dur = np.mean([r[:, j] - r[:, j - 1] for r in rows], axis=0)
    n = dur.shape[0]
    half = n // 2
    print(f"{name}: mean {dur.mean():.2f} us, std {dur.std():.2f}; first half of the grid {dur[:half].mean():.2f}, second half {dur[half:].mean():.2f}; by index mod 8: " +
          " ".join(f"{dur[k::8].mean():.2f}" for k in range(8)))
    # the two workgroups of a column are neighbours in the chunk table: how much of the spread is between columns?
    pair = dur[: n - n % 2].reshape(-1, 2)
    print(f"    std of column means {pair.mean(axis=1).std():.2f}, mean |difference| within a column {np.abs(pair[:, 0] - pair[:, 1]).mean():.2f}; run-to-run std of one workgroup's duration "
          f"{np.std([r[:, j] - r[:, j - 1] for r in rows], axis=0).mean():.2f}")
cost.close()
