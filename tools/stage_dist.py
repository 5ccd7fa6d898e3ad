#!/usr/bin/env python3
"""Per-workgroup timing DISTRIBUTION of the two streaming kernels on a cached scene (tools/make_scene_cache.py), from the wall-clock
stamps (100 MHz) of an INSTRUMENTED library (tools/build_variants.sh stamp="-DNID_STAMP", loaded through NIDREG_LIB): when each
workgroup starts, how long its prologue / point loop / epilogue take, when it ends -- i.e. how much of a kernel's duration is launch
ramp, imbalance between workgroups and the finalising workgroup rather than the point loop (tools/stage_times.py prints the means of
a small evaluation; this one prints percentiles over the 1024 workgroups of a large one).
histogram kernel stamps: 0 entry, 1 tile zeroed, 2 point loop done, 3 flushed;  gradient kernel: 0 entry, 1 entropy tail / scalars,
2 G tile built, 3 point loop done, 4 partial stored, 5 ticket taken, 6 (last workgroup) results written.
Usage: NIDREG_LIB=<instrumented .so> stage_dist.py scene.npz [bins]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import _lib, nid, se3  # noqa: E402

z = np.load(sys.argv[1])
bins = int(sys.argv[2]) if len(sys.argv) > 2 else 256
pts = z["points"].astype(np.float64)
ints = z["intensities"].astype(np.float64)
proj = nid.create_camera(str(z["model"]), list(z["intrinsics"]), list(z["distortion"]))
img64 = z["image_u8"].astype(np.float64) * (1.0 / 255.0)
c = nid.NIDCost(proj, img64, pts, ints, bins)
lib = _lib.load()
nwg = c.info()["num_chunks"]
rng = np.random.default_rng(7)
poses = [se3.plus(z["T_true"], rng.uniform(-1, 1, 6) * np.array([0.05, 0.05, 0.05, 0.0087, 0.0087, 0.0087])) for _ in range(12)]


def stamps():
    buf = (ctypes.c_ulonglong * (8 * nwg))()
    assert lib.nidreg_debug_stage_stamps(buf, 8 * nwg) == 0
    return np.array(buf, dtype=np.int64).reshape(nwg, 8)


def pct(v):
    v = np.asarray(v, dtype=np.float64) * 0.01  # -> us
    return {k: round(float(np.percentile(v, q)), 2) for k, q in (("min", 0), ("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))}


def dist(st, loop_from, loop_to, end_stage, final_stage=None):
    t0 = st[:, 0].min()
    out = {"start_offset_us": pct(st[:, 0] - t0), "prologue_us": pct(st[:, loop_from] - st[:, 0]), "loop_us": pct(st[:, loop_to] - st[:, loop_from]),
           "epilogue_us": pct(st[:, end_stage] - st[:, loop_to]), "loop_end_offset_us": pct(st[:, loop_to] - t0), "end_offset_us": pct(st[:, end_stage] - t0),
           "alive_fraction_of_span": round(float((st[:, end_stage] - st[:, 0]).mean() / (st[:, end_stage].max() - t0)), 3)}
    # workgroup i runs on XCD i mod 8 (observed dispatch order): loop time per XCD
    out["loop_us_mean_by_xcd"] = [round(float(0.01 * (st[x::8, loop_to] - st[x::8, loop_from]).mean()), 2) for x in range(8)]
    out["span_us"] = round(0.01 * float(st[:, end_stage].max() - t0), 2)
    if final_stage is not None:
        k = int(np.argmax(st[:, final_stage]))
        out["final_workgroup_us"] = round(0.01 * float(st[k, final_stage] - st[k, final_stage - 1]), 2)
        out["span_us"] = round(0.01 * float(st[k, final_stage] - t0), 2)
    return out


for x in poses[:4]:
    c(x)
rows_h, rows_g, raw = [], [], []
for x in poses[4:]:
    c(x, want_grad=False)
    st = stamps()
    st = st[st[:, 0] >= st[:, 0].max() - 30000]  # the histogram kernel may run fewer (wider) workgroups than the gradient kernel: rows stamped by THIS launch
    nh = len(st)
    rows_h.append(dist(st, 1, 2, 3))
    c(x)
    st = stamps()
    st[:, 6] = np.where(st[:, 6] > st[:, 5], st[:, 6], 0)
    rows_g.append(dist(st, 2, 3, 5, final_stage=6))
    raw.append(st.copy())
if os.environ.get("STAGE_RAW"):  # the raw gradient-kernel stamps of every timed evaluation ([evaluation][workgroup][stage], 10 ns ticks)
    np.save(os.environ["STAGE_RAW"], np.array(raw))


def med(rows):
    out = {}
    for k in rows[0]:
        if isinstance(rows[0][k], dict):
            out[k] = {kk: float(np.median([r[k][kk] for r in rows])) for kk in rows[0][k]}
        elif isinstance(rows[0][k], list):
            out[k] = [float(np.median([r[k][i] for r in rows])) for i in range(len(rows[0][k]))]
        else:
            out[k] = float(np.median([r[k] for r in rows]))
    return out


print(json.dumps({"scene": os.path.basename(sys.argv[1]), "bins": bins, "workgroups": nwg, "hist_workgroups": nh, "hist": med(rows_h), "grad": med(rows_g)}, indent=1))
c.close()
