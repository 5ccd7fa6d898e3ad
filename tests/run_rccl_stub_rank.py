"""One rank of the two-process test of the in-library sharded evaluation over tests/cxx/rccl_stub.cpp (NIDREG_RCCL_LIB points at it).
usage: run_rccl_stub_rank.py <rank> <world> <dir> <case>   -- writes <dir>/rank<r>.json"""
import json
import os
import sys
import time

import numpy as np

T0 = time.time()


def stamp(what):
    sys.stderr.write(f"[rank {sys.argv[1]}] {time.time() - T0:7.2f} s  {what}\n")
    sys.stderr.flush()


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from direct_visual_lidar_calibration_amd import _lib, nid, parallel, se3, synth  # noqa: E402

stamp("imports done")
rank, world, d, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
s = synth.make_scene("pinhole_vga", num_points=60_000, seed=91)
proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
n = s.points.shape[0]
lo, hi = parallel.shard_slice(n, rank, world)
bins = 64
kw = dict(scale_points=n)
if case == "mismatch_unit":  # the ranks name different totals (2^30 points against the shard's own count): different fixed-point units
    kw = dict(scale_points=1 << 30) if rank == 0 else {}
if case == "mismatch_bins" and rank == 1:
    bins = 32
idfile = os.path.join(d, "uid.bin")
if rank == 0:
    uid = nid.NIDCost.rccl_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        time.sleep(0.01)
        assert time.time() - t0 < 60
    uid = open(idfile, "rb").read()
stamp("scene + id exchanged")
out = {"rank": rank, "case": case}
rng = np.random.default_rng(4)
poses = [s.T_camera_lidar_init] + [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(3)]
if case == "nearest":
    import oracle_lib

    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    h = nid.CostCalculatorNID(proj, s.image_u8, s.points[lo:hi], s.intensities[lo:hi], nid.NIDCostParams(bins), max_fov=max_fov, **kw)
    h.comm_init(world, rank, uid)
    out["costs"] = [h.calculate(se3.to_matrix(x)) for x in poses]
    out["hist_sum"] = int(h.histogram_fixed()[0].sum())
else:
    h = nid.NIDCost(proj, s.image_f64, s.points[lo:hi], s.intensities[lo:hi], bins, **kw)
    stamp("handle created")
    try:
        h.comm_init(world, rank, uid)
        out["attached"] = True
    except RuntimeError as exc:
        out["attached"] = False
        out["error"] = str(exc)
    if out["attached"]:
        res = [h(x) for x in poses]
        out["ok"] = [bool(r[0]) for r in res]
        out["costs"] = [r[1] for r in res]
        out["grads"] = [list(map(float, r[2])) for r in res]
        out["cost_only"] = [h(x, want_grad=False)[1] for x in poses]
stamp("evaluated")
h.close()
stamp("closed")
with open(os.path.join(d, f"rank{rank}.json"), "w") as f:
    json.dump(out, f)
