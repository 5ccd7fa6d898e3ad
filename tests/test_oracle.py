"""CPU tests (no GPU): pin the C++ oracle, since the reference has no tests / goldens for this path.

(i) C++ oracle == independent numpy/torch oracle (value, histograms, autograd gradient);
(ii) Jet gradient == central finite differences on the 6-D tangent;
(iii) reverse-mode factorisation used by the GPU path == Jet gradient;
(iv) committed golden vectors (tests/golden/, produced by tests/make_golden.py) still reproduce;
(v) CostCalculatorNID restatement, camera factory error behaviour, trust gate, Nelder-Mead.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib
import pyoracle
from direct_visual_lidar_calibration_amd import se3, synth
from test_gpu_parity import CAMERAS

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def scene_for(name, n=6000, seed=11):
    key = (name, n, seed)
    if key not in _cache:
        _cache[key] = synth.make_scene(CAMERAS[name], num_points=n, seed=seed)
    return _cache[key]


@pytest.mark.parametrize("model", list(CAMERAS))
@pytest.mark.parametrize("bins", [16, 256])
def test_cpp_oracle_matches_python_oracle(model, bins):
    s = scene_for(model)
    x = s.T_camera_lidar_init
    a = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x, want_hist=True)
    b = pyoracle.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x)
    assert a["ok"] and b["ok"]
    assert abs(a["cost"] - b["cost"]) < 1e-12
    assert np.allclose(a["grad"], b["grad"], rtol=1e-9, atol=1e-12)
    assert np.abs(a["hist"] - b["hist"]).max() < 1e-10
    assert np.abs(a["hist_image"] - b["hist_image"]).max() < 1e-9
    assert np.array_equal(a["hist_points"], b["hist_points"])
    assert a["outliers"] == s.points.shape[0] - b["num_inliers"]
    # cost-only instantiation agrees with the Jet one
    c = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, bins, x, want_grad=False)
    assert abs(c["cost"] - a["cost"]) < 1e-14


@pytest.mark.parametrize("model", list(CAMERAS))
def test_jet_gradient_matches_finite_differences(model):
    s = scene_for(model)
    x = s.T_camera_lidar_init
    a = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, 16, x)
    gt = se3.plus_jacobian(x).T @ a["grad"]
    h = 1e-6
    for k in range(6):
        e = np.zeros(6)
        e[k] = h
        cp = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, 16, se3.plus(x, e), want_grad=False)["cost"]
        cm = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, 16, se3.plus(x, -e), want_grad=False)["cost"]
        fd = (cp - cm) / (2 * h)
        # inlier flips at the image border make NID piecewise smooth; allow a loose band
        assert abs(fd - gt[k]) <= 2e-3 * max(1.0, abs(gt[k])), (k, fd, gt[k])


@pytest.mark.parametrize("model", ["plumb_bob", "fisheye", "equirectangular"])
def test_reverse_mode_factorisation(model):
    s = scene_for(model)
    x = s.T_camera_lidar_init
    a = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, 64, x)
    g, G = pyoracle.reverse_mode_gradient(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, 64, x)
    assert np.allclose(g, a["grad"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("model", list(CAMERAS))
def test_projection_and_jacobian(model):
    import torch

    from direct_visual_lidar_calibration_amd import camera_models

    s = scene_for(model, n=500)
    T = se3.to_matrix(s.T_camera_lidar_true)
    pc = s.points[:, :3] @ T[:3, :3].T + T[:3, 3]
    uv, jac = oracle_lib.project_jacobian(s.model, s.intrinsics, s.distortion, pc)
    p = torch.tensor(pc, requires_grad=True)
    tuv = camera_models.project(s.model, s.intrinsics, s.distortion, p)
    assert np.allclose(uv, tuv.detach().numpy(), rtol=1e-12, atol=1e-9)
    ju = torch.autograd.grad(tuv[:, 0].sum(), p, retain_graph=True)[0].numpy()
    jv = torch.autograd.grad(tuv[:, 1].sum(), p)[0].numpy()
    assert np.allclose(jac[:, 0, :], ju, rtol=1e-8, atol=1e-8)
    assert np.allclose(jac[:, 1, :], jv, rtol=1e-8, atol=1e-8)


@pytest.mark.parametrize("model", list(CAMERAS))
def test_cost_calculator_nid_matches_python(model):
    s = scene_for(model)
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    assert 0.1 < max_fov <= np.pi
    T = se3.to_matrix(s.T_camera_lidar_init)
    for bins in (16, 256):
        c, h = oracle_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, s.intensities, bins, max_fov, T, want_hist=True)
        pc, ph = pyoracle.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, s.intensities, bins, max_fov, T)
        assert np.array_equal(h, ph)
        assert abs(c - pc) < 1e-12


def test_create_camera_rules():
    p = np.array([[0.1, -0.2, 2.0]])
    assert oracle_lib.project("nope", [1, 2, 3, 4], [], p) is None
    assert oracle_lib.project("plumb_bob", [1, 2, 3], [], p) is None
    a = oracle_lib.project("plumb_bob", [100, 100, 50, 50], [0.1], p)  # zero padded
    b = oracle_lib.project("plumb_bob", [100, 100, 50, 50], [0.1, 0, 0, 0, 0, 9, 9], p)  # truncated
    assert np.array_equal(a, b)
    assert np.array_equal(oracle_lib.project("fisheye", [100, 100, 50, 50], [], p), oracle_lib.project("equidistant", [100, 100, 50, 50], [], p))


def test_trust_gate():
    init = synth.true_T_camera_lidar()
    assert oracle_lib.trust_gate(init, init)
    assert oracle_lib.trust_gate(init, se3.plus(init, np.array([0.19, 0, 0, 0, 0, 0])))
    assert not oracle_lib.trust_gate(init, se3.plus(init, np.array([0.21, 0, 0, 0, 0, 0])))
    assert oracle_lib.trust_gate(init, se3.plus(init, np.array([0, 0, 0, np.radians(1.9), 0, 0])))
    assert not oracle_lib.trust_gate(init, se3.plus(init, np.array([0, 0, 0, 0, np.radians(2.1), 0])))


def test_nelder_mead_cpp_matches_host_python():
    from direct_visual_lidar_calibration_amd.dfo import NelderMead, NelderMeadParams

    def f(x):
        return (x[0] - 0.3) ** 2 + 3 * (x[1] + 0.2) ** 2 + 0.5 * (x[2] - 0.1) ** 4 + 0.1 * x[0] * x[1]

    a = oracle_lib.nelder_mead(f, np.zeros(3), init_step=1e-2, conv_thresh=1e-12, max_iterations=300)
    b = NelderMead(NelderMeadParams(init_step=1e-2, convergence_var_thresh=1e-12, max_iterations=300)).optimize(f, np.zeros(3))
    assert a["num_iterations"] == b.num_iterations
    assert np.allclose(a["x"], b.x, atol=1e-12) and abs(a["y"] - b.y) < 1e-14


def test_view_culling_properties():
    s = scene_for("plumb_bob", n=8000)
    T = se3.to_matrix(s.T_camera_lidar_true)
    # add occluded points: the same directions 1 m farther, plus points behind the camera
    Tinv = np.linalg.inv(T)
    pc = s.points[:2000, :3] @ T[:3, :3].T + T[:3, 3]
    far = pc * (1.0 + 1.0 / np.linalg.norm(pc, axis=1, keepdims=True))
    behind = pc * -1.0
    extra = np.concatenate([far, behind]) @ Tinv[:3, :3].T + Tinv[:3, 3]
    pts = np.concatenate([s.points, np.concatenate([extra, np.ones((4000, 1))], -1)])
    idx = oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, T, True)
    idx_nod = oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, T, False)
    n = s.points.shape[0]
    assert np.all(np.diff(idx) > 0)
    assert not np.any(idx >= n + 2000)  # points behind the camera never survive
    assert np.sum(idx_nod >= n) >= 1900  # without the depth buffer the far copies stay
    assert np.sum((idx >= n) & (idx < n + 2000)) < 200  # with it, almost all are removed
    assert set(idx).issubset(set(idx_nod))


def test_golden_vectors():
    path = os.path.join(GOLDEN, "nid_golden.json")
    with open(path) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        s = synth.make_scene(CAMERAS[case["camera"]], num_points=case["num_points"], seed=case["seed"])
        x = np.array(case["se3"])
        r = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_f64, s.points, s.intensities, case["bins"], x, want_hist=True)
        assert abs(r["cost"] - case["cost"]) < 1e-12
        assert np.allclose(r["grad"], case["grad"], rtol=1e-10, atol=1e-12)
        assert abs(r["hist"].sum() - case["hist_sum"]) < 1e-6
        assert np.allclose(r["hist"].reshape(-1)[case["hist_probe_idx"]], case["hist_probe_val"], atol=1e-10)
        max_fov = case["max_fov"]
        c, h = oracle_lib.cost_calculator_nid(s.model, s.intrinsics, s.distortion, s.image_u8, s.points, s.intensities, case["bins"], max_fov, se3.to_matrix(x), want_hist=True)
        assert abs(c - case["nearest_cost"]) < 1e-12
        assert int(h.sum()) == case["nearest_inliers"]
