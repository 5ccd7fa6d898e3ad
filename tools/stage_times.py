#!/usr/bin/env python3
"""Where a small evaluation's kernel time goes: wall-clock stamps (100 MHz) taken by every workgroup at its stage boundaries in
an INSTRUMENTED build of the library (make EXTRA=-DNID_STAMP in a copy of csrc/, loaded through NIDREG_LIB).
histogram kernel: 0 entry, 1 tile zeroed, 2 point loop done, 3 flushed;  gradient kernel: 0 entry, 1 entropy tail / scalars,
2 G tile built, 3 point loop done, 4 partial stored, 5 ticket taken, 6 (last workgroup) results written.
Usage: NIDREG_LIB=<instrumented .so> stage_times.py [points] [bins]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import _lib, nid, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 16
s = synth.make_scene("pinhole_vga", num_points=n, seed=20250523 + 7, device="cpu")
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
rng = np.random.default_rng(3)
poses = [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(12)]
c = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins)
lib = _lib.load()
nwg = c.info()["num_chunks"]


def stamps():
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert lib.nidreg_debug_stage_stamps(buf, 8 * nwg) == 0
    return np.array(buf, dtype=np.int64).reshape(nwg, 8)


def summarise(st, nstage, last_stage=None):
    t0 = st[:, 0].min()
    out = {"first_wg_start_to_last_wg_start_us": round(0.01 * float(st[:, 0].max() - t0), 2)}
    for i in range(1, nstage):
        d = 0.01 * (st[:, i] - st[:, i - 1])
        out[f"stage_{i - 1}_to_{i}_us"] = {"mean": round(float(d.mean()), 2), "max": round(float(d.max()), 2)}
    out["kernel_span_us"] = round(0.01 * float(st[:, nstage - 1].max() - t0), 2)
    if last_stage is not None:
        k = int(np.argmax(st[:, last_stage]))
        out["last_wg_final_us"] = round(0.01 * float(st[k, last_stage] - st[k, last_stage - 1]), 2)
        out["kernel_span_us"] = round(0.01 * float(st[k, last_stage] - t0), 2)
    return out


for x in poses[:4]:
    c(x)
hist_rows, grad_rows = [], []
for x in poses[4:]:
    c(x, want_grad=False)
    hist_rows.append(summarise(stamps(), 4))
    st0 = stamps()
    c(x)
    st = stamps()
    st[:, 6] = np.where(st[:, 6] > st[:, 5], st[:, 6], 0)  # slot 6 is written by the last workgroup only (others keep an older value)
    grad_rows.append(summarise(st, 6, last_stage=6))


def med(rows):
    out = {}
    for k in rows[0]:
        if isinstance(rows[0][k], dict):
            out[k] = {kk: float(np.median([r[k][kk] for r in rows])) for kk in rows[0][k]}
        else:
            out[k] = float(np.median([r[k] for r in rows]))
    return out


print(json.dumps({"points": n, "bins": bins, "workgroups": nwg, "hist": med(hist_rows), "grad": med(grad_rows)}, indent=1))
c.close()
