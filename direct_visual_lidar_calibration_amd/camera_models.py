"""Host-side float64 torch statement of the six projection models.

Used for (a) synthetic-scene generation (``synth.py``: render an image / sample a cloud through a
camera) and (b) as the camera half of the *second* oracle in ``tests/pyoracle.py`` (torch autograd
supplies the Jacobian there).  It is NOT the product path: the product projects on the GPU in
``csrc/nidreg_kernels.hip``; nothing here is called by ``NIDCost`` / ``CostCalculatorNID``.

Formulas follow the reference functors (include/camera/pinhole.hpp:11-52, fisheye.hpp:12-37,
omnidir.hpp:12-42, equirectangular.hpp:12-29, atan.hpp:12-40, rational_polynomial.hpp:9-59) and
the factory's parameter-count rules (src/camera/create_camera.cpp:17-51).
"""
import math

import torch

# model name -> (canonical id, #intrinsics, #distortion)   (create_camera.cpp:34-51, traits)
MODEL_TABLE = {
    "plumb_bob": (0, 4, 5),
    "fisheye": (1, 4, 4),
    "equidistant": (1, 4, 4),
    "omnidir": (2, 5, 4),
    "equirectangular": (3, 2, 0),
    "atan": (4, 4, 1),
    "rational_polynomial": (5, 4, 8),
}


def canonical_params(model, intrinsics, distortion):
    """create_camera.cpp:17-32: intrinsic count must match (else None); distortion is
    zero-padded / truncated to the model's count."""
    if model not in MODEL_TABLE:
        return None
    _, n_i, n_d = MODEL_TABLE[model]
    intrinsics = [float(x) for x in intrinsics]
    if len(intrinsics) != n_i:
        return None
    d = [0.0] * n_d
    for i in range(min(len(distortion), n_d)):
        d[i] = float(distortion[i])
    return intrinsics, d


def _normalized(p):
    n2 = (p * p).sum(-1, keepdim=True)
    n = torch.sqrt(torch.where(n2 > 0, n2, torch.ones_like(n2)))
    return torch.where(n2 > 0, p / n, p)


def project(model, intrinsics, distortion, p):
    """p: (..., 3) float64 tensor in the camera frame -> (..., 2) pixel coordinates."""
    params = canonical_params(model, intrinsics, distortion)
    if params is None:
        raise ValueError(f"bad camera model / intrinsic count: {model} {len(intrinsics)}")
    intr, dist = params
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    mid = MODEL_TABLE[model][0]

    if mid == 0:  # plumb_bob, storage k1 k2 p1 p2 k3
        k1, k2, p1, p2, k3 = dist
        px, py = x / z, y / z
        x2, y2 = px * px, py * py
        r2 = x2 + y2
        r4 = r2 * r2
        r6 = r2 * r4
        rc = 1.0 + k1 * r2 + k2 * r4 + k3 * r6
        t1 = 2.0 * px * py
        t2 = r2 + 2.0 * x2
        t3 = r2 + 2.0 * y2
        dx = rc * px + p1 * t1 + p2 * t2
        dy = rc * py + p1 * t3 + p2 * t1
        return torch.stack([intr[0] * dx + intr[2], intr[1] * dy + intr[3]], -1)

    if mid == 1:  # fisheye / equidistant
        k1, k2, k3, k4 = dist
        r = torch.sqrt(x * x + y * y)
        theta = torch.atan2(r, torch.abs(z))
        t2 = theta * theta
        t4 = t2 * t2
        t6 = t4 * t2
        t8 = t4 * t4
        theta_d = theta * (1.0 + k1 * t2 + k2 * t4 + k3 * t6 + k4 * t8)
        s = theta_d / r
        return torch.stack([intr[0] * (s * x) + intr[2], intr[1] * (s * y) + intr[3]], -1)

    if mid == 2:  # omnidir (unified + plumb_bob), intr fx fy cx cy xi, dist k1 k2 p1 p2
        k1, k2, p1, p2 = dist
        xi = intr[4]
        s = _normalized(p)
        den = s[..., 2] + xi
        ux, uy = s[..., 0] / den, s[..., 1] / den
        r2 = ux * ux + uy * uy
        r4 = r2 * r2
        dr = 1.0 + k1 * r2 + k2 * r4
        x2, y2, xy = ux * ux, uy * uy, ux * uy
        nx = ux * dr + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2)
        ny = uy * dr + p1 * (r2 + 2.0 * y2) + 2.0 * p2 * xy
        return torch.stack([intr[0] * nx + intr[2], intr[1] * ny + intr[3]], -1)

    if mid == 3:  # equirectangular, intr = [W, H]
        n2 = (p * p).sum(-1)
        b = _normalized(p)
        lat = -torch.asin(b[..., 1].clamp(-1.0, 1.0))
        lon = torch.atan2(b[..., 0], b[..., 2])
        u = intr[0] * (0.5 + lon / (2.0 * math.pi))
        v = intr[1] * (0.5 - lat / math.pi)
        small = n2 < 1e-3
        u = torch.where(small, torch.full_like(u, intr[0] / 2), u)
        v = torch.where(small, torch.full_like(v, intr[1] / 2), v)
        return torch.stack([u, v], -1)

    if mid == 4:  # atan (FOV model)
        d0 = dist[0]
        px, py = x / z, y / z
        r = torch.sqrt(px * px + py * py)
        if d0 < 1e-7:
            dx, dy = px, py
        else:
            d1 = 1.0 / d0
            d2 = 2.0 * math.tan(d0 / 2.0)
            safe_r = torch.where(r < 1e-3, torch.ones_like(r), r)
            factor = d1 * torch.atan(safe_r * d2) / safe_r
            factor = torch.where(r < 1e-3, torch.ones_like(r), factor)
            dx, dy = factor * px, factor * py
        return torch.stack([intr[0] * dx + intr[2], intr[1] * dy + intr[3]], -1)

    # rational polynomial, storage k1 k2 p1 p2 k3 k4 k5 k6
    k1, k2, p1, p2, k3, k4, k5, k6 = dist
    px, py = x / z, y / z
    x2, y2 = px * px, py * py
    r2 = x2 + y2
    r4 = r2 * r2
    r6 = r2 * r4
    num = 1.0 + k1 * r2 + k2 * r4 + k3 * r6
    den = 1.0 + k4 * r2 + k5 * r4 + k6 * r6
    rc = torch.where(den > 1e-8, num / torch.where(den > 1e-8, den, torch.ones_like(den)), num)
    t1 = 2.0 * px * py
    t2 = r2 + 2.0 * x2
    t3 = r2 + 2.0 * y2
    dx = rc * px + p1 * t1 + p2 * t2
    dy = rc * py + p1 * t3 + p2 * t1
    return torch.stack([intr[0] * dx + intr[2], intr[1] * dy + intr[3]], -1)


def _dir_from_m(m):
    """Equidistant chart: m = theta * (cos phi, sin phi) -> unit bearing.  Smooth for theta < pi."""
    theta = torch.sqrt((m * m).sum(-1))
    small = theta < 1e-12
    safe = torch.where(small, torch.ones_like(theta), theta)
    sinc = torch.where(small, torch.ones_like(theta), torch.sin(safe) / safe)
    return torch.stack([sinc * m[..., 0], sinc * m[..., 1], torch.cos(theta)], -1)


def unproject(model, intrinsics, distortion, uv, iters=25):
    """Numerical inverse of ``project``: (..., 2) pixels -> (..., 3) unit bearings in the camera
    frame.  Damped Gauss-Newton on the equidistant chart with a finite-difference 2x2 Jacobian;
    closed form for the equirectangular model.  Used only to synthesise test scenes."""
    params = canonical_params(model, intrinsics, distortion)
    intr, _ = params
    mid = MODEL_TABLE[model][0]
    uv = uv.to(torch.float64)
    if mid == 3:
        lon = (uv[..., 0] / intr[0] - 0.5) * 2.0 * math.pi
        lat = (0.5 - uv[..., 1] / intr[1]) * math.pi
        # lat = -asin(b_y) ; lon = atan2(b_x, b_z)
        by = -torch.sin(lat)
        c = torch.cos(lat)
        return torch.stack([c * torch.sin(lon), by, c * torch.cos(lon)], -1)

    fx, fy, cx, cy = intr[0], intr[1], intr[2], intr[3]
    nx = (uv[..., 0] - cx) / fx
    ny = (uv[..., 1] - cy) / fy
    rn = torch.sqrt(nx * nx + ny * ny)
    if mid in (0, 4, 5):
        theta0 = torch.atan(rn)
    elif mid == 1:
        theta0 = rn.clamp(max=3.0)
    else:
        xi = intr[4]
        # unified model: r = sin(theta) / (cos(theta) + xi); start from the xi-scaled pinhole guess
        theta0 = torch.atan(rn * (1.0 + xi)).clamp(max=3.0)
    safe_rn = torch.where(rn < 1e-12, torch.ones_like(rn), rn)
    m = torch.stack([theta0 * nx / safe_rn, theta0 * ny / safe_rn], -1)

    h = 1e-6
    for _ in range(iters):
        f0 = project(model, intrinsics, distortion, _dir_from_m(m)) - uv
        ex = torch.zeros_like(m)
        ex[..., 0] = h
        ey = torch.zeros_like(m)
        ey[..., 1] = h
        fxp = project(model, intrinsics, distortion, _dir_from_m(m + ex)) - uv
        fyp = project(model, intrinsics, distortion, _dir_from_m(m + ey)) - uv
        j00 = (fxp[..., 0] - f0[..., 0]) / h
        j10 = (fxp[..., 1] - f0[..., 1]) / h
        j01 = (fyp[..., 0] - f0[..., 0]) / h
        j11 = (fyp[..., 1] - f0[..., 1]) / h
        det = j00 * j11 - j01 * j10
        det = torch.where(det.abs() < 1e-12, torch.full_like(det, 1e-12), det)
        dx = (j11 * f0[..., 0] - j01 * f0[..., 1]) / det
        dy = (-j10 * f0[..., 0] + j00 * f0[..., 1]) / det
        step = torch.stack([dx, dy], -1)
        # damp: never move more than 0.3 rad per iteration
        sn = torch.sqrt((step * step).sum(-1, keepdim=True))
        step = step * torch.clamp(0.3 / torch.where(sn > 0, sn, torch.ones_like(sn)), max=1.0)
        step = torch.nan_to_num(step, nan=0.0, posinf=0.0, neginf=0.0)
        m = m - step
    return _dir_from_m(m)
