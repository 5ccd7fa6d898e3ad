"""ctypes binding of oracle/_ref/libref.so -- TEST INFRASTRUCTURE ONLY.

libref.so is the reference's own hot-path source files (src/camera/create_camera.cpp,
src/vlcal/calib/{cost_calculator_nid,view_culling}.cpp, src/vlcal/preprocess/generate_lidar_image.cpp,
include/vlcal/costs/nid_cost.hpp, include/camera/*.hpp, include/dfo/nelder_mead.hpp) compiled unmodified,
where they lie under /root/reference, against the stand-in third-party headers in oracle/shim/
(`make -C oracle ref`).  It exists only where the reference tree is mounted; `available()` says so."""
import ctypes
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "_ref", "libref.so")
REFERENCE = "/root/reference"

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
_lib = None


def available():
    """True when libref.so exists or can be built (the reference tree is present)."""
    return os.path.exists(_LIB_PATH) or os.path.isdir(os.path.join(REFERENCE, "include", "vlcal"))


def build():
    if os.path.isdir(os.path.join(REFERENCE, "include", "vlcal")):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.ref_view_culling.restype = ctypes.c_int64
    return _lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _cam(model, intrinsics, distortion):
    intr = np.ascontiguousarray(intrinsics, dtype=np.float64)
    dist = np.ascontiguousarray(distortion if len(distortion) else [0.0], dtype=np.float64)
    return model.encode(), _dp(intr), ctypes.c_int(len(intr)), _dp(dist), ctypes.c_int(len(distortion)), (intr, dist)


def project(model, intrinsics, distortion, p3, jacobian=False):
    p3 = np.ascontiguousarray(p3, dtype=np.float64).reshape(-1, 3)
    uv = np.empty((p3.shape[0], 2))
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    if jacobian:
        jac = np.empty((p3.shape[0], 2, 3))
        rc = lib().ref_project_jacobian(m, ip, ni, dp, nd, _dp(p3), ctypes.c_int64(p3.shape[0]), _dp(uv), _dp(jac))
        return None if rc != 0 else (uv, jac)
    rc = lib().ref_project(m, ip, ni, dp, nd, _dp(p3), ctypes.c_int64(p3.shape[0]), _dp(uv))
    return None if rc != 0 else uv


def nid_cost(model, intrinsics, distortion, image_f64, points, intensities, bins, se3, want_grad=True):
    img = np.ascontiguousarray(image_f64, dtype=np.float64)
    pts = np.ascontiguousarray(points, dtype=np.float64)
    inten = np.ascontiguousarray(intensities, dtype=np.float64)
    x = np.ascontiguousarray(se3, dtype=np.float64)
    cost = ctypes.c_double(0.0)
    grad = np.zeros(7)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_nid_cost(m, ip, ni, dp, nd, _dp(img), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]), _dp(pts), _dp(inten), ctypes.c_int64(pts.shape[0]), ctypes.c_int(bins),
                            _dp(x), ctypes.c_int(1 if want_grad else 0), ctypes.byref(cost), _dp(grad))
    if rc < 0:
        raise ValueError("reference build: bad camera model")
    return dict(ok=rc == 1, cost=cost.value, grad=grad if want_grad else None)


def estimate_camera_fov(model, intrinsics, distortion, width, height):
    out = ctypes.c_double(0.0)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_estimate_camera_fov(m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), ctypes.byref(out))
    if rc != 0:
        raise ValueError("reference build: bad camera model")
    return out.value


def cost_calculator_nid(model, intrinsics, distortion, image_u8, points, intensities, bins, T):
    img = np.ascontiguousarray(image_u8, dtype=np.uint8)
    pts = np.ascontiguousarray(points, dtype=np.float64)
    inten = np.ascontiguousarray(intensities, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    cost = ctypes.c_double(0.0)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_cost_calculator_nid(m, ip, ni, dp, nd, img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]), _dp(pts), _dp(inten),
                                       ctypes.c_int64(pts.shape[0]), ctypes.c_int(bins), _dp(T), ctypes.byref(cost))
    if rc != 0:
        raise ValueError("reference build: bad camera model")
    return cost.value


def view_culling(model, intrinsics, distortion, width, height, points, T, enable_depth_buffer_culling=True):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    idx = np.empty(pts.shape[0], dtype=np.int32)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    n = lib().ref_view_culling(m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), ctypes.c_int(1 if enable_depth_buffer_culling else 0), _dp(pts),
                               ctypes.c_int64(pts.shape[0]), _dp(T), idx.ctypes.data_as(c_int_p))
    if n < 0:
        raise ValueError("reference build: bad camera model")
    return idx[:n].copy()


def generate_lidar_image(model, intrinsics, distortion, width, height, points, intensities, T):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    inten = np.ascontiguousarray(intensities, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    iimg = np.empty((height, width), dtype=np.float64)
    idx = np.empty((height, width), dtype=np.int32)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_generate_lidar_image(m, ip, ni, dp, nd, ctypes.c_int(width), ctypes.c_int(height), _dp(pts), _dp(inten), ctypes.c_int64(pts.shape[0]), _dp(T), _dp(iimg),
                                        idx.ctypes.data_as(c_int_p))
    if rc != 0:
        raise ValueError("reference build: bad camera model")
    return iimg, idx


def points_color_update(model, intrinsics, distortion, image_u8, points, intensity_colors, T, blend_weight):
    """vlcal::PointsColorUpdater (constructor + update) from src/vlcal/common/points_color_updater.cpp; the viewer
    side (glk / guik) is a stand-in that hands the colours back.  Returns (colors n x 4 float32, min_nz)."""
    img = np.ascontiguousarray(image_u8, dtype=np.uint8)
    pts = np.ascontiguousarray(points, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.float64).reshape(4, 4)
    ic = np.ascontiguousarray(intensity_colors, dtype=np.float32).reshape(-1, 4)
    out = np.empty((pts.shape[0], 4), dtype=np.float32)
    min_nz = ctypes.c_double(0.0)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib().ref_points_color_update(m, ip, ni, dp, nd, img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]), _dp(pts),
                                       ctypes.c_int64(pts.shape[0]), ic.ctypes.data_as(fp), _dp(T), ctypes.c_double(blend_weight), out.ctypes.data_as(fp), ctypes.byref(min_nz))
    if rc != 0:
        raise ValueError(f"reference build: points_color_update failed ({rc})")
    return out, min_nz.value


def _dataset_tables(pairs):
    imgs = [np.ascontiguousarray(p[0], dtype=np.uint8) for p in pairs]
    pts = [np.ascontiguousarray(p[1], dtype=np.float64) for p in pairs]
    ints = [np.ascontiguousarray(p[2], dtype=np.float64) for p in pairs]
    n = len(pairs)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    img_tab = (u8p * n)(*[a.ctypes.data_as(u8p) for a in imgs])
    pts_tab = (c_double_p * n)(*[_dp(a) for a in pts])
    int_tab = (c_double_p * n)(*[_dp(a) for a in ints])
    nums = np.array([a.shape[0] for a in pts], dtype=np.int64)
    return imgs, pts, ints, img_tab, pts_tab, int_tab, nums


def calibrate_nelder_mead(model, intrinsics, distortion, pairs, T_init, bins=16, max_outer_iterations=10, max_inner_iterations=256, init_step=1e-3, convergence=1e-8,
                          disable_culling=False):
    """vlcal::VisualCameraCalibration::calibrate (NID_NELDER_MEAD) from src/vlcal/calib/visual_camera_calibration.cpp.
    pairs = [(image_u8, points (n,4), intensities), ...].  Returns (T_camera_lidar 4x4, number of callback calls)."""
    imgs, pts, ints, img_tab, pts_tab, int_tab, nums = _dataset_tables(pairs)
    T_init = np.ascontiguousarray(T_init, dtype=np.float64).reshape(4, 4)
    T_out = np.empty((4, 4))
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_calibrate_nelder_mead(m, ip, ni, dp, nd, ctypes.c_int(len(pairs)), img_tab, ctypes.c_int(imgs[0].shape[0]), ctypes.c_int(imgs[0].shape[1]), pts_tab, int_tab,
                                         nums.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int(bins), ctypes.c_int(max_outer_iterations), ctypes.c_int(max_inner_iterations),
                                         ctypes.c_double(init_step), ctypes.c_double(convergence), ctypes.c_int(1 if disable_culling else 0), _dp(T_init), _dp(T_out))
    if rc < 0:
        raise ValueError("reference build: bad camera model")
    return T_out, rc


def multi_nid_cost_probes(model, intrinsics, distortion, pairs, T_init, probes, bins=16, disable_culling=False):
    """The reference's MultiNIDCost (private to visual_camera_calibration.cpp) evaluated through its own
    estimate_pose_bfgs wiring at the start and at `probes` ((k, 7) parameter vectors).  Returns a dict of arrays
    over the 1 + k points: ok_value, ok_grad, cost_value, cost_grad, grad (.., 7) and the 7 start parameters."""
    imgs, pts, ints, img_tab, pts_tab, int_tab, nums = _dataset_tables(pairs)
    T_init = np.ascontiguousarray(T_init, dtype=np.float64).reshape(4, 4)
    probes = np.ascontiguousarray(probes, dtype=np.float64).reshape(-1, 7)
    k = probes.shape[0] + 1
    okv, okg = np.zeros(k, dtype=np.int32), np.zeros(k, dtype=np.int32)
    cv, cg, grads, start = np.empty(k), np.empty(k), np.empty((k, 7)), np.empty(7)
    m, ip, ni, dp, nd, keep = _cam(model, intrinsics, distortion)
    rc = lib().ref_multi_nid_cost_probes(m, ip, ni, dp, nd, ctypes.c_int(len(pairs)), img_tab, ctypes.c_int(imgs[0].shape[0]), ctypes.c_int(imgs[0].shape[1]), pts_tab, int_tab,
                                         nums.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), ctypes.c_int(bins), ctypes.c_int(1 if disable_culling else 0), _dp(T_init), _dp(probes),
                                         ctypes.c_int(probes.shape[0]), okv.ctypes.data_as(c_int_p), okg.ctypes.data_as(c_int_p), _dp(cv), _dp(cg), _dp(grads), _dp(start))
    if rc != 0:
        raise ValueError(f"reference build: multi_nid_cost_probes failed ({rc})")
    return dict(ok_value=okv.astype(bool), ok_grad=okg.astype(bool), cost_value=cv, cost_grad=cg, grad=grads, start=start)


_NM_FN = ctypes.CFUNCTYPE(ctypes.c_double, c_double_p, ctypes.c_void_p)


def nelder_mead(f, x0, init_step=0.1, conv_thresh=1e-5, max_iterations=1024):
    """dfo::NelderMead<N>::optimize (N = len(x0) in {2, 6}) driven by a Python callable."""
    n = len(x0)
    x0 = np.ascontiguousarray(x0, dtype=np.float64)

    def cb(xp, _user):
        return float(f(np.array([xp[i] for i in range(n)])))

    fn = _NM_FN(cb)
    x = np.empty(n)
    y = ctypes.c_double(0.0)
    iters = ctypes.c_int(0)
    conv = lib().ref_nelder_mead(ctypes.c_int(n), ctypes.c_double(init_step), ctypes.c_double(conv_thresh), ctypes.c_int(max_iterations), fn, None, _dp(x0), _dp(x), ctypes.byref(y),
                                 ctypes.byref(iters))
    if conv < 0:
        raise ValueError("reference build: NelderMead<N> is instantiated for N = 2 and 6 only")
    return dict(x=x, y=y.value, iterations=iters.value, converged=bool(conv))
