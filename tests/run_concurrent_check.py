"""Helper of test_concurrent_callers.py (its own process):
k pairs evaluated one by one, then by k threads that call their own NIDCost at the same pose behind a barrier -- the
reference's OpenMP loop over pairs (visual_camera_calibration.cpp:161).  Prints one JSON line."""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import nid, synth  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_points = int(sys.argv[2]) if len(sys.argv) > 2 else 120000
scenes = [synth.make_scene("pinhole_vga", num_points=n_points, seed=500 + i, device="cuda:0") for i in range(k)]
proj = nid.create_camera(scenes[0].model, scenes[0].intrinsics, scenes[0].distortion)
costs = [nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 64) for s in scenes]
rng = np.random.default_rng(9)
poses = [synth.random_pose_near(scenes[0].T_camera_lidar_true, rng) for _ in range(12)]
single = [[c(x) for c in costs] for x in poses]  # (ok, cost, grad) per pose and pair, one caller at a time
threaded = [[None] * k for _ in poses]
bar = threading.Barrier(k)


def work(i):
    for j, x in enumerate(poses):
        bar.wait()
        threaded[j][i] = costs[i](x)


t0 = time.perf_counter()
th = [threading.Thread(target=work, args=(i,)) for i in range(k)]
for t in th:
    t.start()
for t in th:
    t.join()
dt = time.perf_counter() - t0
# the cost comes from the integer histogram and integer entropy sums: identical bit for bit however the kernels of the
# callers interleave; the gradient's workgroup partials follow the chunk table: equal to rounding
same = all(a[0] == b[0] and a[1] == b[1] and np.allclose(a[2], b[2], rtol=1e-12, atol=1e-15) for ra, rb in zip(single, threaded) for a, b in zip(ra, rb))
maxdiff = max(abs(a[1] - b[1]) for ra, rb in zip(single, threaded) for a, b in zip(ra, rb))
# mixed use afterwards: different poses per thread
mixed = [None] * k


def work_mixed(i):
    bar.wait()
    mixed[i] = costs[i](poses[i % len(poses)])


th = [threading.Thread(target=work_mixed, args=(i,)) for i in range(k)]
for t in th:
    t.start()
for t in th:
    t.join()
mixed_ok = all(mixed[i][1] == single[i % len(poses)][i][1] for i in range(k))
print(json.dumps({"pairs": k, "cost_identical_grad_equal": bool(same), "max_cost_diff": float(maxdiff), "mixed_poses_ok": bool(mixed_ok),
                  "us_per_multi_eval": round(1e6 * dt / len(poses), 1)}))
for c in costs:
    c.close()
