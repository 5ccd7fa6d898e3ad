"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (direct_visual_lidar_calibration_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "liboracle.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)

_lib = None


def build():
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("nid_oracle.cpp", "cameras.hpp", "jet.hpp", "Makefile")]
    if os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs if os.path.exists(s)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_view_culling.restype = ctypes.c_int64
        _lib.oracle_equalize_intensities.restype = None
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def _cam_args(model, intrinsics, distortion):
    intr = np.ascontiguousarray(intrinsics, dtype=np.float64)
    dist = np.ascontiguousarray(distortion if len(distortion) else [0.0], dtype=np.float64)
    n_dist = len(distortion)
    return model.encode(), _dp(intr), ctypes.c_int(len(intr)), _dp(dist), ctypes.c_int(n_dist), (intr, dist)


def project(model, intrinsics, distortion, p3):
    p3 = np.ascontiguousarray(p3, dtype=np.float64).reshape(-1, 3)
    uv = np.empty((p3.shape[0], 2))
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_project(m, ip, ni, dp, nd, _dp(p3), ctypes.c_int64(p3.shape[0]), _dp(uv))
    if rc != 0:
        return None
    return uv


def project_jacobian(model, intrinsics, distortion, p3):
    p3 = np.ascontiguousarray(p3, dtype=np.float64).reshape(-1, 3)
    uv = np.empty((p3.shape[0], 2))
    jac = np.empty((p3.shape[0], 2, 3))
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_project_jacobian(m, ip, ni, dp, nd, _dp(p3), ctypes.c_int64(p3.shape[0]), _dp(uv), _dp(jac))
    if rc != 0:
        return None
    return uv, jac


def nid_cost(model, intrinsics, distortion, image_f64, points, intensities, bins, se3, want_grad=True, want_hist=False, want_hist_grad=False, threads=1):
    """NIDCost::operator() restatement.  Returns dict(ok, cost, grad, hist, hist_image, hist_points,
    hist_grad, outliers); hist arrays are RAW (un-normalised), hist is [bin_image][bin_points]."""
    image_f64 = np.ascontiguousarray(image_f64, dtype=np.float64)
    points = np.ascontiguousarray(points, dtype=np.float64)
    intensities = np.ascontiguousarray(intensities, dtype=np.float64)
    se3 = np.ascontiguousarray(se3, dtype=np.float64)
    rows, cols = image_f64.shape
    n = points.shape[0]
    assert points.shape == (n, 4) and intensities.shape == (n,)
    cost = ctypes.c_double(float("nan"))
    grad = np.full(7, np.nan) if want_grad else None
    hist = np.zeros((bins, bins)) if want_hist else None
    hist_image = np.zeros(bins) if want_hist else None
    hist_points = np.zeros(bins) if want_hist else None
    hist_grad = np.zeros((7, bins, bins)) if want_hist_grad else None
    outliers = ctypes.c_int64(0)
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_nid_cost(
        m, ip, ni, dp, nd, _dp(image_f64), ctypes.c_int(rows), ctypes.c_int(cols), _dp(points), _dp(intensities), ctypes.c_int64(n), ctypes.c_int(bins),
        _dp(se3), ctypes.c_int(threads), ctypes.byref(cost), _dp(grad), _dp(hist), _dp(hist_image), _dp(hist_points), _dp(hist_grad), ctypes.byref(outliers),
    )
    if rc < 0:
        raise ValueError("oracle: bad camera model")
    return dict(ok=bool(rc), cost=cost.value, grad=grad, hist=hist, hist_image=hist_image, hist_points=hist_points, hist_grad=hist_grad, outliers=outliers.value)


def multi_nid_cost(model, intrinsics, distortion, pairs, bins, init_se3, se3, want_grad=True):
    """MultiNIDCost::operator() restatement over pairs = [(image_f64, points, intensities), ...]."""
    n_pairs = len(pairs)
    imgs = [np.ascontiguousarray(p[0], dtype=np.float64) for p in pairs]
    pts = [np.ascontiguousarray(p[1], dtype=np.float64) for p in pairs]
    ints = [np.ascontiguousarray(p[2], dtype=np.float64) for p in pairs]
    rows, cols = imgs[0].shape
    PP = c_double_p * n_pairs
    img_tab = PP(*[_dp(a) for a in imgs])
    pts_tab = PP(*[_dp(a) for a in pts])
    int_tab = PP(*[_dp(a) for a in ints])
    nums = np.array([a.shape[0] for a in pts], dtype=np.int64)
    init_se3 = np.ascontiguousarray(init_se3, dtype=np.float64)
    se3 = np.ascontiguousarray(se3, dtype=np.float64)
    cost = ctypes.c_double(float("nan"))
    grad = np.full(7, np.nan) if want_grad else None
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_multi_nid_cost(
        m, ip, ni, dp, nd, ctypes.c_int(n_pairs), img_tab, ctypes.c_int(rows), ctypes.c_int(cols), pts_tab, int_tab, nums.ctypes.data_as(c_int64_p), ctypes.c_int(bins),
        _dp(init_se3), _dp(se3), ctypes.byref(cost), _dp(grad),
    )
    if rc < 0:
        raise ValueError("oracle: bad camera model")
    return bool(rc), cost.value, grad


def trust_gate(init_se3, se3):
    a = np.ascontiguousarray(init_se3, dtype=np.float64)
    b = np.ascontiguousarray(se3, dtype=np.float64)
    return bool(lib().oracle_trust_gate(_dp(a), _dp(b)))


def cost_calculator_nid(model, intrinsics, distortion, image_u8, points, intensities, bins, max_fov, T, want_hist=False):
    image_u8 = np.ascontiguousarray(image_u8, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.float64)
    intensities = np.ascontiguousarray(intensities, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    rows, cols = image_u8.shape
    cost = ctypes.c_double(float("nan"))
    hist = np.zeros((bins, bins), dtype=np.int64) if want_hist else None
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_cost_calculator_nid(
        m, ip, ni, dp, nd, image_u8.ctypes.data_as(c_uint8_p), ctypes.c_int(rows), ctypes.c_int(cols), _dp(points), _dp(intensities), ctypes.c_int64(points.shape[0]),
        ctypes.c_int(bins), ctypes.c_double(max_fov), _dp(T), ctypes.byref(cost), None if hist is None else hist.ctypes.data_as(c_int64_p),
    )
    if rc < 0:
        raise ValueError("oracle: bad camera model")
    return cost.value, hist


def estimate_camera_fov(model, intrinsics, distortion, width, height):
    out = ctypes.c_double(0.0)
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_estimate_camera_fov(m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), ctypes.byref(out))
    if rc < 0:
        raise ValueError("oracle: bad camera model")
    return out.value


def view_culling(model, intrinsics, distortion, width, height, points, T, enable_depth_buffer_culling=True):
    points = np.ascontiguousarray(points, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    idx = np.empty(points.shape[0], dtype=np.int32)
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    n = lib().oracle_view_culling(
        m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(1 if enable_depth_buffer_culling else 0), _dp(points), ctypes.c_int64(points.shape[0]),
        _dp(T), idx.ctypes.data_as(c_int_p),
    )
    if n < 0:
        raise ValueError("oracle: bad camera model")
    return idx[:n].copy()


def points_color_update(model, intrinsics, distortion, image_u8, points, intensity_colors, T, blend_weight):
    """PointsColorUpdater::update restatement.  Returns (colors n x 4 float32, min_nz)."""
    image_u8 = np.ascontiguousarray(image_u8, dtype=np.uint8)
    points = np.ascontiguousarray(points, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    ic = np.ascontiguousarray(intensity_colors, dtype=np.float32).reshape(-1, 4)
    out = np.empty((points.shape[0], 4), dtype=np.float32)
    min_nz = ctypes.c_double(0.0)
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_points_color_update(
        m, ip, ni, dp, nd, image_u8.ctypes.data_as(c_uint8_p), ctypes.c_int(image_u8.shape[0]), ctypes.c_int(image_u8.shape[1]), _dp(points), ctypes.c_int64(points.shape[0]),
        ic.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _dp(T), ctypes.c_double(blend_weight), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), ctypes.byref(min_nz),
    )
    if rc != 0:
        raise ValueError("oracle: bad camera model")
    return out, min_nz.value


def generate_lidar_image(model, intrinsics, distortion, width, height, points, intensities, T):
    """generate_lidar_image restatement.  Returns (intensity_image H x W float64, index_image H x W int32)."""
    points = np.ascontiguousarray(points, dtype=np.float64)
    intensities = np.ascontiguousarray(intensities, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    iimg = np.empty((height, width), dtype=np.float64)
    idx = np.empty((height, width), dtype=np.int32)
    m, ip, ni, dp, nd, keep = _cam_args(model, intrinsics, distortion)
    rc = lib().oracle_generate_lidar_image(
        m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), _dp(points), _dp(intensities), ctypes.c_int64(points.shape[0]), _dp(T), _dp(iimg), idx.ctypes.data_as(c_int_p),
    )
    if rc != 0:
        raise ValueError("oracle: bad camera model")
    return iimg, idx


def equalize_intensities(intensities):
    """preprocess.cpp:464-473 rank equalisation (stable order among ties)."""
    a = np.array(intensities, dtype=np.float64, copy=True)
    lib().oracle_equalize_intensities(_dp(a), ctypes.c_int64(a.shape[0]))
    return a


_NM_FN = ctypes.CFUNCTYPE(ctypes.c_double, c_double_p, ctypes.c_void_p)


def nelder_mead(f, x0, init_step=0.1, conv_thresh=1e-5, max_iterations=1024):
    """dfo::NelderMead<N>::optimize restatement driven by a Python callable."""
    n = len(x0)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)

    def cb(xp, _user):
        return float(f(np.array([xp[i] for i in range(n)])))

    fn = _NM_FN(cb)
    x_out = np.empty(n)
    y_out = ctypes.c_double(0.0)
    iters = ctypes.c_int(0)
    conv = lib().oracle_nelder_mead(
        ctypes.c_int(n), ctypes.c_double(init_step), ctypes.c_double(conv_thresh), ctypes.c_int(max_iterations), fn, None, _dp(x0), _dp(x_out), ctypes.byref(y_out),
        ctypes.byref(iters),
    )
    return dict(x=x_out, y=y_out.value, num_iterations=iters.value, converged=bool(conv))


def num_threads():
    return int(lib().oracle_num_threads())
