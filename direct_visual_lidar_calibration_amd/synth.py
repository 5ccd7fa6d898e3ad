"""Seeded synthetic LiDAR-camera pairs in the value distributions the reference's preprocessing
produces: a textured box room seen by a camera and a co-located LiDAR.

* image: 8-bit gray, histogram-equalised like ``cv::equalizeHist`` (preprocess.cpp:419);
* points: float32-representable xyz (PLY round trip, preprocess.cpp:163-169) stored as the
  reference's ``Frame`` does -- (x, y, z, 1) doubles (frame.hpp:66) -- in sweep/azimuth/beam order;
* intensities: rank-equalised to {0, 1/256, ..., 255/256} (preprocess.cpp:464-473);
* ``T_camera_lidar_true`` known, ``T_camera_lidar_init`` = truth * exp(small delta).

This is data generation for tests and bench.py, not part of the registration path.  Runs on any
torch device (the bench builds its 10M-point cloud on the GPU in a second or two).
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import camera_models, se3

ROOM_LO = (-8.0, -9.0, -1.5)
ROOM_HI = (14.0, 7.0, 4.0)

# camera z forward = LiDAR x forward; camera sits at t_lidar_camera in the LiDAR frame
R_CAMERA_LIDAR = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])
T_LIDAR_CAMERA_POS = np.array([0.03, 0.002, 0.11])

CONFIG_CAMERAS = {
    # name: (model, intrinsics, distortion, W, H)
    "pinhole_vga": ("plumb_bob", [400.0, 400.0, 320.0, 240.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 640, 480),
    "pinhole_1080p": ("plumb_bob", [1100.0, 1100.0, 960.0, 540.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 1920, 1080),
    "pinhole_4k": ("plumb_bob", [2200.0, 2200.0, 1920.0, 1080.0], [-0.04, 0.08, 1e-4, -3e-4, -0.04], 3840, 2160),
    "equirect_2k": ("equirectangular", [2048.0, 2048.0], [], 2048, 2048),
    "omnidir_2k": ("omnidir", [600.0, 600.0, 1024.0, 1024.0, 1.0], [-0.02, 0.003, 1e-4, -2e-4], 2048, 2048),
    "fisheye_1080p": ("fisheye", [800.0, 800.0, 960.0, 540.0], [-0.01, 0.002, -1e-4, 1e-5], 1920, 1080),
    # the two models no BASELINE config names, for their kernel-stats lines
    "atan_1080p": ("atan", [1100.0, 1100.0, 960.0, 540.0], [0.6], 1920, 1080),
    "rational_1080p": ("rational_polynomial", [1100.0, 1100.0, 960.0, 540.0], [0.05, -0.02, 1e-4, -2e-4, 0.01, 0.03, -0.01, 0.002], 1920, 1080),
}


@dataclass
class Scene:
    model: str
    intrinsics: list
    distortion: list
    width: int
    height: int
    image_u8: np.ndarray  # (H, W) uint8
    points: np.ndarray  # (N, 4) float64, xyz1
    intensities: np.ndarray  # (N,) float64
    T_camera_lidar_true: np.ndarray  # 7-vector [qx qy qz qw tx ty tz]
    T_camera_lidar_init: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def image_f64(self):
        """``image.convertTo(CV_64FC1, 1/255.0)`` (visual_camera_calibration.cpp:204)."""
        return self.image_u8.astype(np.float64) * (1.0 / 255.0)


def true_T_camera_lidar():
    T_lidar_camera = np.eye(4)
    T_lidar_camera[:3, :3] = R_CAMERA_LIDAR.T
    T_lidar_camera[:3, 3] = T_LIDAR_CAMERA_POS
    return se3.from_matrix(np.linalg.inv(T_lidar_camera))


def texture(X):
    """Procedural surface albedo in [0,1] at world points X (..., 3)."""
    x, y, z = X[..., 0], X[..., 1], X[..., 2]
    t = 0.5 + 0.22 * torch.sin(2.1 * x + 1.3 * y) * torch.cos(1.7 * z + 0.5 * x)
    t = t + 0.13 * torch.sin(5.3 * y - 3.1 * z + 0.7) + 0.07 * torch.sin(17.0 * x + 11.0 * y + 13.0 * z)
    t = t + 0.05 * torch.sin(41.0 * x - 29.0 * y + 37.0 * z + 1.1)
    checker = (torch.floor(x / 0.6) + torch.floor(y / 0.6) + torch.floor(z / 0.6)) % 2
    t = t + 0.10 * (checker - 0.5)
    return t.clamp(0.0, 1.0)


def ray_room(origin, dirs):
    """First hit of rays origin + s * dirs (origin strictly inside the box) with the room walls."""
    lo = torch.tensor(ROOM_LO, dtype=dirs.dtype, device=dirs.device)
    hi = torch.tensor(ROOM_HI, dtype=dirs.dtype, device=dirs.device)
    o = torch.as_tensor(origin, dtype=dirs.dtype, device=dirs.device)
    target = torch.where(dirs > 0, hi, lo)
    safe = torch.where(dirs.abs() < 1e-12, torch.full_like(dirs, 1e-12), dirs)
    s = ((target - o) / safe).abs()
    s = torch.where(dirs.abs() < 1e-12, torch.full_like(s, float("inf")), s)
    smin = s.min(-1, keepdim=True).values
    return o + smin * dirs


def equalize_hist_u8(img):
    """``cv::equalizeHist`` on a uint8 torch tensor."""
    hist = torch.bincount(img.flatten().to(torch.int64), minlength=256).to(torch.float64)
    nz = torch.nonzero(hist).flatten()
    i0 = int(nz[0])
    total = float(hist.sum())
    if total == float(hist[i0]):
        return torch.full_like(img, i0)
    scale = 255.0 / (total - float(hist[i0]))
    csum = torch.cumsum(hist, 0) - hist[: i0 + 1].sum()
    lut = torch.round(csum * scale).clamp(0, 255)
    lut[: i0 + 1] = 0
    return lut.to(torch.uint8)[img.to(torch.int64)]


def _unproject_chunked(model, intr, dist, uv, chunk=2_000_000):
    outs = []
    for s in range(0, uv.shape[0], chunk):
        outs.append(camera_models.unproject(model, intr, dist, uv[s : s + chunk]))
    return torch.cat(outs, 0)


def make_scene(camera="pinhole_vga", num_points=100_000, seed=20250523, device="cpu", init_delta=(0.03, 0.5), range_noise=0.02, model_override=None):
    """Build one seeded LiDAR-camera pair.  ``camera`` is a key of CONFIG_CAMERAS or a tuple
    (model, intrinsics, distortion, W, H).  ``init_delta`` = (max |dt| per axis [m], max rot per
    axis [deg]) of the initial-guess perturbation."""
    model, intr, dist, W, H = CONFIG_CAMERAS[camera] if isinstance(camera, str) else camera
    if model_override is not None:
        model = model_override
    dev = torch.device(device)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    f64 = torch.float64

    T_true = true_T_camera_lidar()
    T_cl = torch.tensor(se3.to_matrix(T_true), dtype=f64, device=dev)
    R_cl, t_cl = T_cl[:3, :3], T_cl[:3, 3]
    cam_center = -(R_cl.T @ t_cl)  # camera position in the LiDAR frame

    # ---- image: one ray per pixel centre
    vs, us = torch.meshgrid(torch.arange(H, dtype=f64, device=dev), torch.arange(W, dtype=f64, device=dev), indexing="ij")
    uv = torch.stack([us.flatten(), vs.flatten()], -1)
    bear = _unproject_chunked(model, intr, dist, uv)
    hit = ray_room(cam_center, bear @ R_cl)  # R_cl^T applied to row vectors
    img = torch.round(texture(hit) * 255.0).clamp(0, 255).to(torch.uint8).reshape(H, W)
    img = equalize_hist_u8(img)

    # ---- cloud: sweep -> azimuth -> beam order, directions spread over the image
    N = int(num_points)
    n_beams = 128
    n_az = max(8, int(round(n_beams * W / H)))
    per_sweep = n_beams * n_az
    k = torch.arange(N, dtype=torch.int64)
    sweep = k // per_sweep
    rem = k % per_sweep
    az = (rem // n_beams).to(f64)
    beam = (rem % n_beams).to(f64)
    n_sweeps = int(sweep.max()) + 1 if N > 0 else 1
    jit = torch.rand((n_sweeps, 2), generator=gen, dtype=f64)
    noise = torch.rand((N, 2), generator=gen, dtype=f64)
    margin = 2.5
    pu = margin + (az + 0.8 * jit[sweep, 0] + 0.2 * noise[:, 0]) / n_az * (W - 2 * margin)
    pv = margin + (beam + 0.8 * jit[sweep, 1] + 0.2 * noise[:, 1]) / n_beams * (H - 2 * margin)
    puv = torch.stack([pu, pv], -1).to(dev)
    pbear = _unproject_chunked(model, intr, dist, puv)
    X = ray_room(cam_center, pbear @ R_cl)
    albedo = texture(X)
    rng = torch.sqrt((X * X).sum(-1, keepdim=True))
    rn = (torch.rand((N, 1), generator=gen, dtype=f64) * 2.0 - 1.0).to(dev) * range_noise
    X = X * (1.0 + rn / rng)
    X = X.to(torch.float32).to(f64)  # PLY float32 round trip
    inten = albedo + 0.02 * torch.randn((N,), generator=gen, dtype=f64).to(dev)

    # rank equalisation (preprocess.cpp:464-473): value_i = floor(256 * rank / N) / 256
    order = torch.argsort(inten, stable=True)
    ranks = torch.empty(N, dtype=torch.int64, device=dev)
    ranks[order] = torch.arange(N, dtype=torch.int64, device=dev)
    inten = torch.floor(256.0 * ranks.to(f64) / max(N, 1)) / 256.0

    pts = torch.cat([X, torch.ones((N, 1), dtype=f64, device=dev)], -1)

    # ---- initial guess: truth * exp(delta)
    d = (torch.rand(6, generator=gen, dtype=f64) * 2.0 - 1.0).numpy()
    delta = np.concatenate([d[:3] * init_delta[0], d[3:] * math.radians(init_delta[1])])
    T_init = se3.plus(T_true, delta)

    return Scene(
        model=model,
        intrinsics=list(intr),
        distortion=list(dist),
        width=W,
        height=H,
        image_u8=img.cpu().numpy(),
        points=np.ascontiguousarray(pts.cpu().numpy()),
        intensities=np.ascontiguousarray(inten.cpu().numpy()),
        T_camera_lidar_true=T_true,
        T_camera_lidar_init=T_init,
        meta={"camera": camera if isinstance(camera, str) else "custom", "seed": int(seed), "num_points": N},
    )


def random_pose_near(T, rng, dt=0.05, drot_deg=0.5):
    """T * exp(delta) with delta ~ U(+-dt m, +-drot deg): distinct evaluation poses for benches."""
    d = rng.uniform(-1.0, 1.0, 6)
    delta = np.concatenate([d[:3] * dt, d[3:] * math.radians(drot_deg)])
    return se3.plus(T, delta)
