#!/usr/bin/env python3
"""Where the fused single-launch evaluation's time goes (csrc/nid_fused.hpp): wall-clock stamps (100 MHz) taken by every workgroup at its
stage boundaries in an INSTRUMENTED build (tools/build_variants.sh stamp="-DNID_STAMP", loaded through NIDREG_LIB):
0 entry, 1 points done (pass A), 2 flush issued, 3 barrier passed, 4 G tile built (phase P), 5 taps done (pass B), 6 ticket drawn, 7 (last
workgroup) results written.  Usage: NIDREG_LIB=variants/libnidreg_stamp.so fused_stage_times.py [points] [bins]"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import _lib, nid, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s = synth.make_scene("pinhole_vga", num_points=n, seed=20250523 + 7, device="cpu")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
rng = np.random.default_rng(3)
poses = np.ascontiguousarray([synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(40)])
c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins)
lib = _lib.load()
c.eval_batch(poses[:5])
info = c.info()
assert info["fused"] == 1, info
nwg = info["num_chunks"]
rows = []
for x in poses[5:25]:
    c(x)
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert lib.nidreg_debug_fused_stage_stamps(buf, 8 * nwg) == 0
    st = np.array(buf, dtype=np.int64).reshape(nwg, 8)
    t0 = st[:, 0].min()
    k = int(np.argmax(st[:, 7]))  # the finalising workgroup of THIS launch holds the newest slot-7 stamp
    row = {"entry_spread_us": 0.01 * float(st[:, 0].max() - t0)}
    names = ["A_points", "flush", "barrier", "P_entropy_gtile", "B_taps", "reduce_ticket"]
    for i, nm in enumerate(names):
        d = 0.01 * (st[:, i + 1] - st[:, i])
        row[nm + "_mean"] = float(d.mean())
        row[nm + "_max"] = float(d.max())
    row["last_arrival_at_barrier_us"] = 0.01 * float(st[:, 2].max() - t0)
    row["first_release_us"] = 0.01 * float(st[:, 3].min() - t0)
    row["last_release_us"] = 0.01 * float(st[:, 3].max() - t0)
    row["last_ticket_us"] = 0.01 * float(st[:, 6].max() - t0)
    row["final_us"] = 0.01 * float(st[k, 7] - st[k, 6])
    row["kernel_span_us"] = 0.01 * float(st[k, 7] - t0)
    rows.append(row)
t = []
for _ in range(9):
    t0 = time.perf_counter()
    c.eval_batch(poses)
    t.append((time.perf_counter() - t0) / len(poses))
out = {"points": n, "bins": bins, "workgroups": nwg, "full_stash": info["fused_full_stash"], "wall_us_per_eval_instrumented_build": round(1e6 * float(np.median(t)), 2)}
out.update({k_: round(float(np.median([r[k_] for r in rows])), 2) for k_ in rows[0]})
print(json.dumps(out))
c.close()
