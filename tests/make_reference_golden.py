"""Generates tests/golden/reference_cases.npz: small self-contained cases (camera, 8-bit image, float32 cloud,
pose) together with the outputs of the REFERENCE'S OWN SOURCES on them -- oracle/_ref/libref.so, i.e.
include/vlcal/costs/nid_cost.hpp, include/camera/*.hpp, src/vlcal/calib/{cost_calculator_nid,view_culling}.cpp,
src/vlcal/preprocess/generate_lidar_image.cpp compiled unmodified from /root/reference against the stand-in
third-party headers of oracle/shim/ (`make -C oracle ref`).  Run in the build container (the reference tree
is not available on the GPU boxes); the committed fixture is what travels.

    python tests/make_reference_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import ref_lib  # noqa: E402
from direct_visual_lidar_calibration_amd import se3, synth  # noqa: E402

CASES = [  # (model, intrinsics, distortion, W, H, bins, points, seed)
    ("plumb_bob", [66.0, 64.0, 48.0, 36.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 96, 72, 16, 1500, 201),
    ("plumb_bob", [66.0, 64.0, 48.0, 36.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 96, 72, 256, 1500, 202),
    ("fisheye", [42.0, 42.0, 48.0, 36.0], [-0.01, 0.002, -1e-4, 1e-5], 96, 72, 64, 1500, 203),
    ("omnidir", [33.0, 33.0, 48.0, 48.0, 1.0], [-0.02, 0.003, 1e-4, -2e-4], 96, 96, 16, 1500, 204),
    ("equirectangular", [128.0, 64.0], [], 128, 64, 256, 1500, 205),
    ("atan", [66.0, 64.0, 48.0, 36.0], [0.6], 96, 72, 16, 1200, 206),
    ("rational_polynomial", [66.0, 64.0, 48.0, 36.0], [0.05, -0.02, 1e-4, -2e-4, 0.01, 0.03, -0.01, 0.002], 96, 72, 100, 1200, 207),
    # more bins than the kernels hold per axis (256): what the reference returns at --nid_bins 512 on its own kind of data (8-bit
    # image, 256-level intensities) -- 256 occupied rows and columns of a 512 x 512 table, i.e. the 256-bin NID again (ref_cost_at_256)
    ("plumb_bob", [66.0, 64.0, 48.0, 36.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 96, 72, 512, 1500, 208),
]

out = {"num_cases": np.array(len(CASES))}
for k, (model, intr, dist, W, H, bins, n, seed) in enumerate(CASES):
    s = synth.make_scene((model, intr, dist, W, H), num_points=n, seed=seed)
    x = np.asarray(s.T_camera_lidar_init, dtype=np.float64)
    T = se3.to_matrix(x)
    Tinv = np.linalg.inv(T)
    # occluded copies, points behind the camera and exact duplicates, so that culling / z-buffering have work to do
    pc = s.points[:300, :3] @ T[:3, :3].T + T[:3, 3]
    far = pc * (1.0 + 1.0 / np.linalg.norm(pc, axis=1, keepdims=True))
    extra = np.concatenate([far, -pc]) @ Tinv[:3, :3].T + Tinv[:3, 3]
    xyz = np.concatenate([s.points[:, :3], extra, s.points[5:60, :3]]).astype(np.float32)  # float32 like a PLY-loaded cloud
    rng = np.random.default_rng(seed)
    inten = np.concatenate([s.intensities, np.floor(rng.random(extra.shape[0]) * 256) / 256, s.intensities[5:60]])
    pts = np.ones((xyz.shape[0], 4))
    pts[:, :3] = xyz
    img64 = s.image_u8.astype(np.float64) * (1.0 / 255.0)
    n_cost = s.points.shape[0]  # the cost functors get the un-augmented cloud (NIDCost has no FoV / z test)
    r = ref_lib.nid_cost(model, intr, dist, img64, pts[:n_cost], inten[:n_cost], bins, x)
    rd = ref_lib.nid_cost(model, intr, dist, img64, pts[:n_cost], inten[:n_cost], bins, x, want_grad=False)
    assert r["ok"] and rd["ok"]
    fov = ref_lib.estimate_camera_fov(model, intr, dist, W, H)
    uv, jac = ref_lib.project(model, intr, dist, pts[:64, :3] @ T[:3, :3].T + T[:3, 3], jacobian=True)
    lidar_img, lidar_idx = ref_lib.generate_lidar_image(model, intr, dist, W, H, pts, inten, T)
    ncol = 400  # every 5th point + the occluded / behind-camera tail would do; per-point independent, so a prefix + tail
    csel = np.concatenate([np.arange(0, ncol), np.arange(pts.shape[0] - 100, pts.shape[0])])
    icol = (np.floor(rng.random((csel.shape[0], 4)) * 64) / 64).astype(np.float32)
    colors, min_nz = ref_lib.points_color_update(model, intr, dist, s.image_u8, pts[csel], icol, T, 0.7)
    # the reference's whole calibrate() (outer loop + Nelder-Mead inner solve) on this bag, from its own source
    T_nm, n_callbacks = ref_lib.calibrate_nelder_mead(model, intr, dist, [(s.image_u8, pts, inten)], T, bins=min(bins, 64))
    p = f"c{k}_"
    out.update({
        p + "ref_nm_T_camera_lidar": T_nm, p + "ref_nm_callbacks": np.array(n_callbacks), p + "nm_bins": np.array(min(bins, 64)),
        p + "color_points": csel, p + "intensity_colors": icol, p + "ref_colors": colors, p + "ref_min_nz": np.array(min_nz),
        p + "model": np.array(model), p + "intrinsics": np.array(intr), p + "distortion": np.array(dist, dtype=np.float64), p + "size": np.array([W, H]), p + "bins": np.array(bins),
        p + "image_u8": s.image_u8, p + "xyz": xyz, p + "intensities": inten, p + "num_cost_points": np.array(n_cost), p + "se3": x,
        p + "ref_cost": np.array(r["cost"]), p + "ref_grad": r["grad"], p + "ref_cost_double": np.array(rd["cost"]), p + "ref_fov": np.array(fov),
        p + "ref_nearest_cost": np.array(ref_lib.cost_calculator_nid(model, intr, dist, s.image_u8, pts, inten, bins, T)),
        p + "ref_cull_depth": ref_lib.view_culling(model, intr, dist, W, H, pts, T, True), p + "ref_cull_nodepth": ref_lib.view_culling(model, intr, dist, W, H, pts, T, False),
        p + "ref_uv": uv, p + "ref_jac": jac, p + "ref_lidar_index": lidar_idx, p + "ref_lidar_intensity": lidar_img,
    })
    if bins > 256:
        out[p + "ref_cost_at_256"] = np.array(ref_lib.nid_cost(model, intr, dist, img64, pts[:n_cost], inten[:n_cost], 256, x)["cost"])
        out[p + "ref_nearest_cost_at_256"] = np.array(ref_lib.cost_calculator_nid(model, intr, dist, s.image_u8, pts, inten, 256, T))
    print(model, bins, "cost", r["cost"], "culled", out[p + "ref_cull_depth"].shape[0], "of", pts.shape[0], "lidar pixels", int((lidar_idx >= 0).sum()))
os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
path = os.path.join(HERE, "golden", "reference_cases.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
