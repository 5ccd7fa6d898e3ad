"""The hand-written CPU restatement (oracle/nid_oracle.cpp, the checker of every GPU parity test) against the
REFERENCE'S OWN SOURCE FILES, compiled unmodified where they lie under /root/reference into
oracle/_ref/libref.so (`make -C oracle ref`; third-party headers replaced by the stand-ins in oracle/shim/).

Both sides share the dual-number arithmetic (oracle/jet.hpp) and evaluate sums in index order, so on the
same inputs they must agree to the last bit wherever the restatement follows the reference's expression
tree -- which is what these tests assert.  Skipped where the reference tree is not mounted (GPU boxes)."""
import numpy as np
import pytest

import oracle_lib
import ref_lib
from direct_visual_lidar_calibration_amd import se3, synth

pytestmark = pytest.mark.skipif(not ref_lib.available(), reason="reference tree (/root/reference) not present and oracle/_ref/libref.so not built")

CAMERAS = {
    "plumb_bob": ("plumb_bob", [210.0, 205.0, 160.0, 120.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 320, 240),
    "fisheye": ("fisheye", [140.0, 140.0, 160.0, 120.0], [-0.01, 0.002, -1e-4, 1e-5], 320, 240),
    "omnidir": ("omnidir", [110.0, 110.0, 160.0, 160.0, 1.0], [-0.02, 0.003, 1e-4, -2e-4], 320, 320),
    "equirectangular": ("equirectangular", [384.0, 256.0], [], 384, 256),
    "atan": ("atan", [210.0, 205.0, 160.0, 120.0], [0.6], 320, 240),
    "rational_polynomial": ("rational_polynomial", [210.0, 205.0, 160.0, 120.0], [0.05, -0.02, 1e-4, -2e-4, 0.01, 0.03, -0.01, 0.002], 320, 240),
}
_scenes = {}


def scene(name, n=12000):
    if (name, n) not in _scenes:
        _scenes[(name, n)] = synth.make_scene(CAMERAS[name], num_points=n, seed=13)
    return _scenes[(name, n)]


def same(a, b):
    """bit-for-bit equality, NaNs in the same places"""
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("model", list(CAMERAS))
def test_camera_models_value_and_jet_jacobian_bit_identical(model):
    m, intr, dist, W, H = CAMERAS[model]
    rng = np.random.default_rng(3)
    p = rng.normal(size=(5000, 3)) * np.array([4.0, 3.0, 6.0]) + np.array([0.0, 0.0, 5.0])
    p[:50, 2] *= -1.0           # behind the camera: no z test in the models (fisheye takes abs(z))
    p[50] = [0.0, 0.0, 3.0]     # optical axis (fisheye: 0/0)
    p[51] = [0.01, 0.0, 0.01]   # |p|^2 < 1e-3 (equirectangular centre rule)
    p[52] = [0.0, 0.0, 0.0]
    p[53] = [0.0, 2.0, 0.0]     # equirectangular pole, atan2(0, 0)
    uv_ref = ref_lib.project(m, intr, dist, p)
    uv_orc = oracle_lib.project(m, intr, dist, p)
    assert same(uv_ref, uv_orc)
    uvj_ref, jac_ref = ref_lib.project(m, intr, dist, p, jacobian=True)
    uvj_orc, jac_orc = oracle_lib.project_jacobian(m, intr, dist, p)
    assert same(uvj_ref, uvj_orc) and same(jac_ref, np.asarray(jac_orc).reshape(-1, 2, 3))
    assert np.isfinite(uv_ref).mean() > 0.9


def test_create_camera_error_behaviour_and_distortion_padding():
    p = np.array([[0.3, -0.2, 2.0]])
    assert ref_lib.project("pinhole", [1, 2, 3, 4], [], p) is None and oracle_lib.project("pinhole", [1, 2, 3, 4], [], p) is None
    assert ref_lib.project("plumb_bob", [1, 2, 3], [], p) is None and oracle_lib.project("plumb_bob", [1, 2, 3], [], p) is None
    intr = [210.0, 205.0, 160.0, 120.0]
    # missing distortion coefficients are zero-padded, surplus ones dropped (create_camera.cpp:24-27)
    for dist in ([], [-0.04], [-0.04, 0.08, 1e-4, -3e-4, -0.04, 9.0, 9.0]):
        assert same(ref_lib.project("plumb_bob", intr, dist, p), oracle_lib.project("plumb_bob", intr, dist, p))
    assert same(ref_lib.project("equidistant", intr, [0.01], p), oracle_lib.project("fisheye", intr, [0.01, 0, 0, 0], p))


@pytest.mark.parametrize("model,bins", [("plumb_bob", 16), ("plumb_bob", 256), ("fisheye", 16), ("omnidir", 7), ("equirectangular", 64), ("atan", 16), ("rational_polynomial", 100)])
def test_nid_cost_functor_bit_identical(model, bins):
    s = scene(model)
    for x in (s.T_camera_lidar_init, s.T_camera_lidar_true):
        ref = ref_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x)
        orc = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x)
        assert ref["ok"] and orc["ok"]
        assert ref["cost"] == orc["cost"]
        assert same(ref["grad"], orc["grad"])
        ref_d = ref_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x, want_grad=False)
        orc_d = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x, want_grad=False)
        assert ref_d["ok"] and ref_d["cost"] == orc_d["cost"]
    # un-normalised quaternion (the functor differentiates the un-normalised formula)
    xq = np.array(s.T_camera_lidar_init, dtype=np.float64)
    xq[:4] *= 1.0003
    ref = ref_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, xq)
    orc = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, xq)
    assert ref["cost"] == orc["cost"] and same(ref["grad"], orc["grad"])


def test_nid_cost_returns_false_without_inliers_and_handles_borders():
    s = scene("plumb_bob")
    away = se3.from_matrix(np.diag([-1.0, 1.0, -1.0, 1.0]) @ se3.to_matrix(s.T_camera_lidar_init))  # look the other way
    ref = ref_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points[:500], s.intensities[:500], 16, away)
    orc = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points[:500], s.intensities[:500], 16, away)
    assert ref["ok"] == orc["ok"]
    # knots on the image border (clamped taps) and just outside
    T = se3.to_matrix(s.T_camera_lidar_init)
    Tinv = np.linalg.inv(T)
    rays = []
    for u, v in [(0.2, 0.3), (0.9, 100.5), (s.width - 0.5, 50.2), (100.3, s.height - 0.2), (-0.3, 10.0), (s.width + 0.2, 10.0), (1.5, 1.5), (s.width - 1.5, s.height - 1.5)]:
        d = np.array([(u - s.intrinsics[2]) / s.intrinsics[0], (v - s.intrinsics[3]) / s.intrinsics[1], 1.0]) * 4.0
        rays.append(np.append(Tinv[:3, :3] @ d + Tinv[:3, 3], 1.0))
    pts = np.concatenate([np.array(rays), s.points[:300]])
    inten = np.concatenate([np.linspace(0.0, 0.999, len(rays)), s.intensities[:300]])
    ref = ref_lib.nid_cost(s.model, s.intrinsics, [0.0] * 5, s.image_f64, pts, inten, 16, s.T_camera_lidar_init)
    orc = oracle_lib.nid_cost(s.model, s.intrinsics, [0.0] * 5, s.image_f64, pts, inten, 16, s.T_camera_lidar_init)
    assert ref["ok"] and ref["cost"] == orc["cost"] and same(ref["grad"], orc["grad"])


@pytest.mark.parametrize("model", ["plumb_bob", "fisheye", "omnidir", "equirectangular"])
def test_fov_nearest_cost_culling_and_lidar_image_identical(model):
    s = scene(model)
    fov_ref = ref_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    fov_orc = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    assert abs(fov_ref - fov_orc) <= 1e-12  # AngleAxis products written out differently in the two restatements of to_dir
    T = se3.to_matrix(s.T_camera_lidar_init)
    Tinv = np.linalg.inv(T)
    pc = s.points[:3000, :3] @ T[:3, :3].T + T[:3, 3]
    far = pc * (1.0 + 1.0 / np.linalg.norm(pc, axis=1, keepdims=True))
    near = pc * (1.0 + 0.05 / np.linalg.norm(pc, axis=1, keepdims=True))
    extra = np.concatenate([far, near, -pc]) @ Tinv[:3, :3].T + Tinv[:3, 3]
    pts = np.ascontiguousarray(np.concatenate([s.points, np.concatenate([extra, np.ones((extra.shape[0], 1))], -1), s.points[10:200]]))
    inten = np.concatenate([s.intensities, np.random.default_rng(1).random(extra.shape[0]), s.intensities[10:200]])
    # CostCalculatorNID::calculate (the reference estimates max_fov in its constructor; hand the oracle the same value)
    for bins in (16, 256):
        c_ref = ref_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, pts, inten, bins, T)
        c_orc, _ = oracle_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, pts, inten, bins, fov_ref, T)
        assert c_ref == c_orc
    if fov_ref == fov_orc:  # the oracle's culling / rendering entry points estimate the FoV themselves
        for depth in (True, False):
            assert np.array_equal(ref_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, T, depth),
                                  oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, T, depth))
        img_ref, idx_ref = ref_lib.generate_lidar_image(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, inten, T)
        img_orc, idx_orc = oracle_lib.generate_lidar_image(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, inten, T)
        assert np.array_equal(idx_ref, idx_orc) and same(img_ref, img_orc)
        assert (idx_ref >= 0).mean() > 0.05
    else:
        pytest.fail(f"estimate_camera_fov differs in the last bits: {fov_ref!r} vs {fov_orc!r}")


@pytest.mark.parametrize("model", ["plumb_bob", "fisheye", "omnidir", "equirectangular"])
def test_points_color_updater_identical(model):
    """src/vlcal/common/points_color_updater.cpp (constructor's min_nz, update's blend) vs the restatement; the
    Vector4f * double products follow Eigen's scalar promotion (the double is converted to float first)."""
    s = scene(model)
    T = se3.to_matrix(s.T_camera_lidar_init)
    rng = np.random.default_rng(8)
    pts = np.concatenate([s.points, np.concatenate([-s.points[:500, :3], np.ones((500, 1))], -1)])  # some behind the camera
    ic = rng.random((pts.shape[0], 4)).astype(np.float32)
    for w in (0.7, 0.0, 1.0, 1.0 / 3.0):
        c_ref, nz_ref = ref_lib.points_color_update(s.model, s.intrinsics, s.distortion, s.image_u8, pts, ic, T, w)
        c_orc, nz_orc = oracle_lib.points_color_update(s.model, s.intrinsics, s.distortion, s.image_u8, pts, ic, T, w)
        assert nz_ref == nz_orc
        assert same(c_ref, c_orc)
    assert (c_ref[:, 3] > 0).mean() > 0.1 and np.all(c_ref[9] == c_orc[9])


def _bags(n_bags, n_points, seed, init_delta):
    return [synth.make_scene(CAMERAS["plumb_bob"], num_points=n_points, seed=seed + k, init_delta=init_delta) for k in range(n_bags)]


def test_multi_nid_cost_functor_and_trust_gate_identical():
    """The reference's MultiNIDCost (visual_camera_calibration.cpp:141-178: trust gate 0.2 m / 2 deg on
    init^-1 * T, per-pair NIDCost, plain sum, false if any pair failed) reached through its own
    estimate_pose_bfgs wiring (view culling at the initial pose, convertTo(CV_64FC1, 1/255), NIDCost per pair,
    AutoDiffFirstOrderFunction<MultiNIDCost, 7>); ceres::Solve is the evaluate-and-record stand-in."""
    bags = _bags(2, 6000, 50, (0.02, 0.4))
    s = bags[0]
    T = se3.to_matrix(s.T_camera_lidar_init)
    pairs = [(b.image_u8, b.points, b.intensities) for b in bags]
    x0 = se3.from_matrix(T)
    rng = np.random.default_rng(0)
    probes = [se3.plus(x0, rng.uniform(-1, 1, 6) * np.array([0.01, 0.01, 0.01, 0.004, 0.004, 0.004])) for _ in range(4)]
    probes.append(se3.plus(x0, np.array([0.19, 0, 0, 0, 0, 0])))        # just inside the 0.2 m gate
    probes.append(se3.plus(x0, np.array([0.25, 0, 0, 0, 0, 0])))        # outside
    probes.append(se3.plus(x0, np.array([0, 0, 0, 0, 0.034, 0])))       # 1.95 deg: inside the 2 deg gate
    probes.append(se3.plus(x0, np.array([0, 0, 0, 0.05, 0, 0])))        # 2.9 deg: outside
    q = np.array(probes[0])
    q[:4] *= 1.0002                                                       # un-normalised quaternion
    probes.append(q)
    for bins, depth in ((16, True), (64, False)):
        r = ref_lib.multi_nid_cost_probes(s.model, s.intrinsics, s.distortion, pairs, T, probes, bins=bins, disable_culling=not depth)
        assert np.allclose(r["start"], x0, rtol=0, atol=1e-15)  # Sophus::SE3d(matrix).data() vs se3.from_matrix
        culled = []
        for b in bags:
            idx = oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, b.points, T, depth)
            culled.append((b.image_u8.astype(np.float64) * (1.0 / 255.0), b.points[idx], b.intensities[idx]))
        oks = []
        for k, x in enumerate([r["start"]] + probes):
            ok, c, g = oracle_lib.multi_nid_cost(s.model, s.intrinsics, s.distortion, culled, bins, r["start"], x)
            okd, cd, _ = oracle_lib.multi_nid_cost(s.model, s.intrinsics, s.distortion, culled, bins, r["start"], x, want_grad=False)
            assert bool(r["ok_grad"][k]) == ok and bool(r["ok_value"][k]) == okd
            oks.append(ok)
            if ok:
                assert r["cost_grad"][k] == c and same(r["grad"][k], g)
                assert r["cost_value"][k] == cd
        assert oks == [True] * 6 + [False, True, False, True]


@pytest.mark.parametrize("n_bags,init_delta,disable_culling", [(1, (0.02, 0.4), False), (2, (0.03, 1.5), False), (2, (0.03, 1.5), True)])
def test_outer_loop_with_nelder_mead_identical(n_bags, init_delta, disable_culling):
    """vlcal::VisualCameraCalibration::calibrate with NID_NELDER_MEAD compiled from the reference
    (visual_camera_calibration.cpp:35-68 outer loop, :70-139 inner solve: culling per outer iteration,
    CostCalculatorNID per pair, init * Pose3::Expmap(x), dfo::NelderMead<6>) against this repository's host
    driver (calibration.VisualCameraCalibration + dfo.NelderMead + se3.pose3_expmap) running on the oracle."""
    from direct_visual_lidar_calibration_amd import calibration
    from test_calibration import OracleNearest, oracle_cull

    bags = _bags(n_bags, 5000, 60, init_delta)
    s = bags[0]
    T = se3.to_matrix(s.T_camera_lidar_init)
    pairs = [(b.image_u8, b.points, b.intensities) for b in bags]
    T_ref, n_cb = ref_lib.calibrate_nelder_mead(s.model, s.intrinsics, s.distortion, pairs, T, bins=16, disable_culling=disable_culling)
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    calls = [0]
    p = calibration.VisualCameraCalibrationParams(nid_bins=16, registration_type="nid_nelder_mead", disable_z_buffer_culling=disable_culling)
    cull = oracle_cull(s, depth=not disable_culling)
    cal = calibration.VisualCameraCalibration(pairs, p, nearest_cost_factory=lambda i, pt, it, b: OracleNearest(s, i, pt, it, b, max_fov), cull=cull,
                                              callback=lambda x: calls.__setitem__(0, calls[0] + 1))
    x = cal.calibrate(se3.from_matrix(T))
    dt, dr = se3.delta_trans_rot(se3.from_matrix(T_ref), x)
    assert dt <= 1e-9 and dr <= 1e-9, (dt, dr, cal.log)
    assert calls[0] == n_cb
    if init_delta[1] > 1.0:
        assert len(cal.log) >= 2  # the outer loop did go round again


def test_nelder_mead_trajectory_identical():
    def rosen2(x):
        return (1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2

    def quad6(x):
        return float(np.sum((x - np.arange(6) * 0.01) ** 2 * (1 + np.arange(6))) + 0.1 * np.sin(5 * x[0]))

    for f, x0, kw in ((rosen2, [-1.2, 1.0], {}), (rosen2, [0.0, 0.0], dict(init_step=0.05, conv_thresh=1e-10, max_iterations=300)),
                      (quad6, [0.1] * 6, dict(init_step=1e-3, conv_thresh=1e-8, max_iterations=256))):
        calls_ref, calls_orc = [], []
        r = ref_lib.nelder_mead(lambda x: (calls_ref.append(x.copy()), f(x))[1], x0, **kw)
        o = oracle_lib.nelder_mead(lambda x: (calls_orc.append(x.copy()), f(x))[1], x0, **kw)
        assert r["iterations"] == o["num_iterations"] and r["converged"] == o["converged"]
        assert same(r["x"], o["x"]) and r["y"] == o["y"]
        assert len(calls_ref) == len(calls_orc) and all(same(a, b) for a, b in zip(calls_ref, calls_orc))  # every probe, in order
