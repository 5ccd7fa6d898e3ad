"""Helper of test_sharded_concurrent.py (its own process: GPU_MAX_HW_QUEUES is read when HIP initialises): k pairs, each
sharded over `shards` co-located slices of GPU 0, evaluated (a) one after the other and (b) by k threads at once -- the
reference's OpenMP loop over the pairs of a multi-bag dataset (visual_camera_calibration.cpp:161) under NIDREG_DEVICES --
against plain single-GPU handles of the same pairs.  Prints one JSON line."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
shards = int(sys.argv[2]) if len(sys.argv) > 2 else 2
bins = int(sys.argv[3]) if len(sys.argv) > 3 else 256
scenes = [synth.make_scene("pinhole_vga", num_points=40000 + 1000 * i, seed=700 + i, device="cuda:0") for i in range(k)]
proj = nid.create_camera(scenes[0].model, scenes[0].intrinsics, scenes[0].distortion)
plain = [nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins) for s in scenes]
workers = len(sys.argv) > 4 and sys.argv[4] == "workers"
if workers:  # every shard driven by its own host thread, as shards on devices of their own are (needs a hardware queue per stream):
    os.environ["NIDREG_SHARD_COLOCATED_WORKERS"] = "1"  # the launch pattern real multi-GPU sets run
os.environ["NIDREG_DEVICES"] = ",".join(["0"] * shards)  # the unchanged caller: nothing about GPUs in the constructor
sharded = [nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins) for s in scenes]
del os.environ["NIDREG_DEVICES"]
assert all(c.num_shards() == shards for c in sharded) and all(c.num_shards() == 1 for c in plain)
rng = np.random.default_rng(9)
poses = [synth.random_pose_near(scenes[0].T_camera_lidar_true, rng) for _ in range(10)]
ref = [[c(x) for c in plain] for x in poses]
serial = [[c(x) for c in sharded] for x in poses]
cost_only_ok = all(c(x, want_grad=False)[:2] == r[:2] for x, row in zip(poses[:4], ref) for c, r in zip(sharded, row))  # (the tail then ends the evaluation)
threaded = [[None] * k for _ in poses]
bar = threading.Barrier(k)
errors = []


def work(i):
    try:
        for j, x in enumerate(poses):
            bar.wait()
            threaded[j][i] = sharded[i](x)
    except Exception as exc:  # a timeout inside the exchange surfaces as a RuntimeError
        errors.append(repr(exc))
        bar.abort()


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0


def same(a, b):
    return a[0] == b[0] and a[1] == b[1] and np.allclose(a[2], b[2], rtol=1e-12, atol=1e-15)


ok_serial = all(same(a, b) for ra, rb in zip(ref, serial) for a, b in zip(ra, rb))
ok_threads = not errors and all(same(a, b) for ra, rb in zip(ref, threaded) for a, b in zip(ra, rb))
hist_ok = all(np.array_equal(a.histogram_fixed()[0], b.histogram_fixed()[0]) and a.histogram_fixed()[1] == b.histogram_fixed()[1] for a, b in zip(plain, sharded))
print(json.dumps({"pairs": k, "shards": shards, "bins": bins, "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"), "serial_ok": bool(ok_serial), "threads_ok": bool(ok_threads),
                  "hist_ok": bool(hist_ok), "cost_only_ok": bool(cost_only_ok), "workers": bool(workers), "errors": errors[:3], "us_per_round_of_k": round(1e6 * dt / len(poses), 1)}))
for c in plain + sharded:
    c.close()
