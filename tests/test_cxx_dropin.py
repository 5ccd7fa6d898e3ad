"""The C++ drop-in classes (include/vlcal_amd/*.hpp: vlcal::NIDCost, vlcal::CostCalculatorNID,
camera::create_camera) compile against the C ABI (CPU test) and, on a GPU, reproduce the oracle
through a MultiNIDCost-style functor instantiated with double and Jet<double,7> (gpu test)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "direct_visual_lidar_calibration_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cxx", "test_dropin.bin")


def build_exe():
    import __graft_entry__

    if not os.path.exists(os.path.join(CSRC, "libnidreg.so")):
        __graft_entry__.build()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", "test_dropin.cpp"), "-L", CSRC, "-lnidreg",
           f"-Wl,-rpath,{CSRC}", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


def test_dropin_headers_compile_and_link():
    assert os.path.exists(build_exe())


@pytest.mark.gpu
def test_dropin_matches_oracle(tmp_path):
    import oracle_lib
    from direct_visual_lidar_calibration_amd import se3, synth
    from test_gpu_parity import CAMERAS

    exe = build_exe()
    for name, bins in (("plumb_bob", 16), ("fisheye", 256)):
        s = synth.make_scene(CAMERAS[name], num_points=20000, seed=31)
        x = s.T_camera_lidar_init
        T = se3.to_matrix(x)
        max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
        intr = np.zeros(5)
        intr[: len(s.intrinsics)] = s.intrinsics
        dist = np.zeros(8)
        dist[: len(s.distortion)] = s.distortion
        path = tmp_path / f"{name}.bin"
        with open(path, "wb") as f:
            f.write(s.model.encode().ljust(64, b"\0"))
            f.write(struct.pack("<6i", s.width, s.height, s.points.shape[0], bins, len(s.intrinsics), len(s.distortion)))
            f.write(intr.tobytes() + dist.tobytes() + np.asarray(x, dtype=np.float64).tobytes() + struct.pack("<d", max_fov) + T.astype(np.float64).tobytes())
            f.write(np.ascontiguousarray(s.image_u8).tobytes())
            f.write(np.ascontiguousarray(s.points, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(s.intensities, dtype=np.float64).tobytes())
        out = subprocess.check_output([exe, str(path)]).decode().split()
        vals = np.array([float(v) for v in out])
        ref = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x)
        parity.check_cost(vals[0], ref["cost"])
        parity.check_cost(vals[8], ref["cost"])
        parity.check_grad(vals[1:8], ref["grad"])
        cn, _ = oracle_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, s.intensities, bins, max_fov, T)
        assert abs(vals[9] - cn) <= 1e-12
        # PointsColorUpdater / generate_lidar_image drop-in headers: checksums against the oracle
        ones = np.ones((s.points.shape[0], 4), dtype=np.float32)
        col, _ = oracle_lib.points_color_update(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, ones, T, 0.7)
        assert abs(vals[10] - float(col.astype(np.float64).sum())) <= 1e-6 * max(1.0, abs(vals[10]))
        assert int(vals[11]) == int((col[:, 3] > 0).sum())
        li, lidx = oracle_lib.generate_lidar_image(s.model, s.intrinsics, s.distortion, s.width, s.height, s.points, s.intensities, T)
        assert int(vals[12]) == int((lidx >= 0).sum()) and int(vals[12]) > 0
        assert vals[13] == float(lidx[lidx >= 0].astype(np.float64).sum())
        assert abs(vals[14] - float(li.sum())) <= 1e-9 * max(1.0, abs(vals[14]))
        kept = oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, s.points, T, True)
        assert int(vals[15]) == kept.shape[0] and vals[16] == float(kept.astype(np.float64).sum())
