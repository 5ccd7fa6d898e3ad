"""The one-launch evaluation (k_fused, csrc/nid_fused.hpp: histogram -> grid barrier -> distributed entropy -> grid
barrier -> gradient in ONE kernel; opt-in: NIDREG_FUSED=1 at handle creation) against the default three-kernel path and the
oracle.  The two routes must give the same fixed-point histogram and the same cost bit for bit
(the entropies are integer sums, csrc/nid_kernels.hpp ent_fixed) and gradients equal up to the order of the workgroup
partials."""
import numpy as np
import pytest

import oracle_lib
import parity
from direct_visual_lidar_calibration_amd import nid, synth
from test_gpu_parity import CAMERAS, oracle_nid, scene_for

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, proj, s, bins, **kw):
    monkeypatch.setenv("NIDREG_FUSED", "0")
    three = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins, **kw)
    monkeypatch.setenv("NIDREG_FUSED", "1")
    fused = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, bins, **kw)
    monkeypatch.delenv("NIDREG_FUSED")
    return three, fused


@pytest.mark.parametrize("model", list(CAMERAS))
@pytest.mark.parametrize("bins", [16, 100, 256])
def test_fused_matches_three_kernel_path_and_oracle(monkeypatch, model, bins):
    s = scene_for(model)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    three, fused = _pair(monkeypatch, proj, s, bins)
    # eligible whenever the histogram pass's chunk table is one co-resident round of k_fused as well (nidreg.hip checks the
    # kernel's own occupancy); the pinhole family always is
    assert three.info()["fused"] == 0 and (fused.info()["fused"] == 1 or model in ("fisheye", "equirectangular"))
    rng = np.random.default_rng(3)
    poses = [s.T_camera_lidar_init, s.T_camera_lidar_true] + [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(4)]
    for k, x in enumerate(poses):  # back to back: both histogram buffers, the barrier counter's running base
        ok3, c3, g3 = three(x)
        okf, cf, gf = fused(x)
        assert ok3 and okf and cf == c3
        assert np.allclose(gf, g3, rtol=1e-12, atol=1e-15)
        assert np.array_equal(fused.histogram_fixed()[0], three.histogram_fixed()[0]) and fused.histogram_fixed()[1] == three.histogram_fixed()[1]
        j3, hi3, hp3 = three.histograms()
        jf, hif, hpf = fused.histograms()
        assert np.array_equal(hif, hi3) and np.array_equal(hpf, hp3)
        okc, cc, gc = fused(x, want_grad=False)  # cost only: one barrier, last workgroup finalises
        assert okc and cc == c3 and gc is None
        if k < 2:
            ref = oracle_nid(s, bins, x)
            parity.check_cost(cf, ref["cost"])
            parity.check_grad(gf, ref["grad"])
    three.close()
    fused.close()


@pytest.mark.parametrize("n,target_blocks", [(700, 0), (30000, 37), (30000, 300), (200000, 0)])
def test_fused_with_few_and_many_workgroups(monkeypatch, n, target_blocks):
    """Row segments of the entropy phase: fewer workgroups than bins (whole rows each), more (row segments), and workgroups
    without entropy work."""
    s = scene_for("plumb_bob", n=n)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    for bins in (7, 64, 256):
        three, fused = _pair(monkeypatch, proj, s, bins, target_blocks=target_blocks)
        for x in (s.T_camera_lidar_init, s.T_camera_lidar_true):
            ok3, c3, g3 = three(x)
            okf, cf, gf = fused(x)
            assert ok3 and okf and cf == c3 and np.allclose(gf, g3, rtol=1e-12, atol=1e-15)
            assert fused(x, want_grad=False)[1] == c3
        ref = oracle_nid(s, bins, s.T_camera_lidar_true)
        parity.check_cost(cf, ref["cost"])
        parity.check_grad(gf, ref["grad"])
        three.close()
        fused.close()


def test_fused_rejects_like_the_reference_when_nothing_projects(monkeypatch):
    """sum == 0 -> NaN -> `return false` (nid_cost.hpp:98-102) through the fused route as well."""
    s = scene_for("plumb_bob", n=5000)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    three, fused = _pair(monkeypatch, proj, s, 16)
    x = np.array(s.T_camera_lidar_true, dtype=np.float64).copy()
    x[4:7] += 1.0e4  # the whole cloud leaves the image
    for h in (three, fused):
        ok, c, g = h(x)
        assert not ok and not np.isfinite(c)
        ok, c, g = h(x, want_grad=False)
        assert not ok
        ok, c, g = h(s.T_camera_lidar_true)  # and recovers
        assert ok and np.isfinite(c)
    three.close()
    fused.close()


def test_fused_barrier_timeout_falls_back_to_three_kernels(monkeypatch):
    """A grid barrier that cannot complete in time (here: an absurdly short timeout) abandons the launch; the kernel ends by
    itself, the evaluation is repeated with the three-kernel path and the handle stays there -- same results."""
    s = scene_for("plumb_bob", n=200000)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    monkeypatch.setenv("NIDREG_FUSED", "0")
    three = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 256)
    monkeypatch.setenv("NIDREG_FUSED", "1")
    monkeypatch.setenv("NIDREG_FUSED_TIMEOUT_US", "1")
    short = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 256)
    monkeypatch.delenv("NIDREG_FUSED")
    monkeypatch.delenv("NIDREG_FUSED_TIMEOUT_US")
    rng = np.random.default_rng(4)
    for x in [s.T_camera_lidar_init] + [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(3)]:
        ok3, c3, g3 = three(x)
        oks, cs, gs = short(x)
        assert ok3 and oks and cs == c3 and np.allclose(gs, g3, rtol=1e-12, atol=1e-15)
        assert short(x, want_grad=False)[1] == c3
    three.close()
    short.close()


def test_fused_headline_shape_matches_three_kernel_path(monkeypatch):
    """BASELINE configs[1] at 2M points (WIDE histogram tile, 512-thread workgroups, 32-copy G tile)."""
    s = synth.make_scene("pinhole_1080p", num_points=2_000_000, seed=20250523 + 2)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    three, fused = _pair(monkeypatch, proj, s, 256)
    assert fused.info()["lds_copies"] == 32
    rng = np.random.default_rng(1)
    for x in [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(3)]:
        ok3, c3, g3 = three(x)
        okf, cf, gf = fused(x)
        assert ok3 and okf and cf == c3 and np.allclose(gf, g3, rtol=1e-12, atol=1e-15)
        assert np.array_equal(fused.histogram_fixed()[0], three.histogram_fixed()[0])
    ref = oracle_nid(s, 256, x, threads=oracle_lib.num_threads())
    parity.check_cost(cf, ref["cost"])
    parity.check_grad(gf, ref["grad"])
    three.close()
    fused.close()
