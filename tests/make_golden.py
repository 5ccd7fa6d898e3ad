"""Generates tests/golden/nid_golden.json from the C++ oracle AFTER it has been cross-checked against
the independent Python oracle (tests/test_oracle.py).  These vectors pin the oracle against regressions;
they are oracle outputs.  Outputs of the reference's own sources (compiled against stand-in third-party
headers) are in tests/golden/reference_cases.npz, written by tests/make_reference_golden.py."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
import pyoracle  # noqa: E402
from direct_visual_lidar_calibration_amd import se3, synth  # noqa: E402
from test_gpu_parity import CAMERAS  # noqa: E402

cases = []
for camera, bins, n, seed in [("plumb_bob", 16, 4000, 101), ("plumb_bob", 256, 4000, 102), ("fisheye", 64, 3000, 103), ("omnidir", 16, 3000, 104),
                              ("equirectangular", 256, 3000, 105), ("atan", 16, 2000, 106), ("rational_polynomial", 32, 2000, 107)]:
    s = synth.make_scene(CAMERAS[camera], num_points=n, seed=seed)
    x = s.T_camera_lidar_init
    r = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x, want_hist=True)
    p = pyoracle.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x)
    assert abs(r["cost"] - p["cost"]) < 1e-12 and np.allclose(r["grad"], p["grad"], rtol=1e-9, atol=1e-12)
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    c, h = oracle_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, s.intensities, bins, max_fov, se3.to_matrix(x), want_hist=True)
    flat = r["hist"].reshape(-1)
    probe = np.argsort(-flat)[:8]
    cases.append(dict(camera=camera, bins=bins, num_points=n, seed=seed, se3=list(map(float, x)), cost=r["cost"], grad=list(map(float, r["grad"])),
                      hist_sum=float(flat.sum()), hist_probe_idx=list(map(int, probe)), hist_probe_val=list(map(float, flat[probe])), max_fov=max_fov,
                      nearest_cost=c, nearest_inliers=int(h.sum())))
os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
with open(os.path.join(HERE, "golden", "nid_golden.json"), "w") as f:
    json.dump(dict(generator="tests/make_golden.py", note="oracle outputs (cross-checked vs pyoracle), not reference outputs", cases=cases), f, indent=1)
print("wrote", len(cases), "cases")
