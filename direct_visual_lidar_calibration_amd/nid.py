"""Host-side mirror of the reference's cost-function interface for the NID hot path.

Same names, argument meaning and error behaviour as the reference classes, so parity tests read
like tests of the reference would:

=====================================  ==========================================================
here                                   reference
=====================================  ==========================================================
``create_camera(model, intr, dist)``   ``camera::create_camera`` (src/camera/create_camera.cpp:34-51):
                                       returns ``None`` on unknown model / intrinsic-count mismatch
``GenericCamera.project(p)``           ``GenericCameraBase::project`` (generic_camera_base.hpp:29)
``NIDCost(proj, img64, pts, I, bins)`` ``vlcal::NIDCost`` ctor (include/vlcal/costs/nid_cost.hpp:23)
``NIDCost.__call__(params)``           ``NIDCost::operator()<T>`` (:36-107): ``(ok, cost, grad7)``
``MultiNIDCost(init).add(c)``          ``MultiNIDCost`` (visual_camera_calibration.cpp:141-178)
``CostCalculatorNID(proj, img8, ...)`` ``vlcal::CostCalculatorNID`` (cost_calculator_nid.cpp:13-17)
``CostCalculatorNID.calculate(T)``     ``CostCalculatorNID::calculate`` (:21-67)
``estimate_camera_fov(proj, size)``    ``vlcal::estimate_camera_fov`` (estimate_fov.cpp:36-51)
=====================================  ==========================================================

All arithmetic of the cost functions runs in the HIP library behind ``include/nidreg.h``; this file
only marshals numpy arrays across the C ABI.  There is no CPU implementation to fall back to.
"""
import ctypes
import math

import numpy as np

from . import _lib
from .dfo import NelderMead, NelderMeadParams

_MODEL_COUNTS = {0: (4, 5), 1: (4, 4), 2: (5, 4), 3: (2, 0), 4: (4, 1), 5: (4, 8)}


def _dp(a):
    return None if a is None else a.ctypes.data_as(_lib.c_double_p)


class GenericCamera:
    """What ``camera::create_camera`` hands back: an opaque projection object.  Unlike the
    reference's ``GenericCameraBase`` it also exposes the model id and parameters, which the GPU
    cost objects need (SURVEY.md hard part 5)."""

    def __init__(self, model, model_id, intrinsics, distortion):
        self.model = model
        self.model_id = model_id
        self.intrinsics = list(intrinsics)
        self.distortion = list(distortion)
        self._intr5 = np.zeros(5)
        self._intr5[: len(intrinsics)] = intrinsics
        self._dist8 = np.zeros(8)
        self._dist8[: len(distortion)] = distortion

    def project(self, point_3d, device=0, precision="fp64", jacobian=False):
        """``GenericCameraBase::project`` evaluated by the device projection code (any (n,3) batch)."""
        lib = _lib.load()
        p = np.ascontiguousarray(np.asarray(point_3d, dtype=np.float64).reshape(-1, 3))
        n = p.shape[0]
        uv = np.empty((n, 2))
        jac = np.empty((n, 2, 3)) if jacobian else None
        rc = lib.nidreg_project_model(self.model_id, _dp(self._intr5), _dp(self._dist8), device, _prec(precision), _dp(p), n, _dp(uv), _dp(jac))
        _lib.check(rc, "nidreg_project_model")
        if np.ndim(point_3d) == 1:
            return (uv[0], jac[0]) if jacobian else uv[0]
        return (uv, jac) if jacobian else uv

    __call__ = project


def create_camera(camera_model, intrinsics, distortion_coeffs):
    """``camera::create_camera``: ``None`` when the model is unknown or the number of intrinsics is
    wrong (create_camera.cpp:19-22,49-50); distortion zero-padded / truncated (:24-27)."""
    lib = _lib.load()
    ni, nd = ctypes.c_int(0), ctypes.c_int(0)
    mid = lib.nidreg_model_from_name(str(camera_model).encode(), ctypes.byref(ni), ctypes.byref(nd))
    if mid < 0:
        return None
    if len(intrinsics) != ni.value:
        return None
    dist = [0.0] * nd.value
    for i in range(min(len(distortion_coeffs), nd.value)):
        dist[i] = float(distortion_coeffs[i])
    return GenericCamera(camera_model, mid, [float(x) for x in intrinsics], dist)


def _prec(precision):
    if precision in ("fp64", "f64", _lib.PREC_FP64):
        return _lib.PREC_FP64
    if isinstance(precision, int):
        return precision  # (passed through: nidreg_create refuses what it does not know, with its own message)
    raise ValueError(f"unknown precision {precision!r} (the core computes in double; the float-geometry mode of rounds 1-4 bought 8 % and was removed)")


class Cloud:
    """One LiDAR cloud resident on the GPU (``Frame::points`` + ``Frame::intensities`` uploaded once per
    pair).  Handles built from it (``NIDCost.from_cloud`` / ``CostCalculatorNID.from_cloud``) are culled,
    bucketed and sorted on the device -- no host round trip per outer iteration."""

    def __init__(self, points, intensities, device=0):
        lib = _lib.load()
        self._lib = lib
        points = np.ascontiguousarray(points, dtype=np.float64)
        intensities = np.ascontiguousarray(intensities, dtype=np.float64)
        if points.ndim != 2 or points.shape[1] != 4 or intensities.shape != (points.shape[0],):
            raise ValueError("points must be (N, 4) float64 (x y z 1) and intensities (N,)")
        c = ctypes.c_void_p()
        _lib.check(lib.nidreg_cloud_create(int(device), _dp(points), points.strides[0] if points.shape[0] else 32, _dp(intensities), points.shape[0], ctypes.byref(c)),
                   "nidreg_cloud_create")
        self.c = c
        self.device = int(device)
        self.num_points = points.shape[0]

    def close(self):
        if getattr(self, "c", None):
            self._lib.nidreg_cloud_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Handle:
    """RAII owner of one ``nidreg_handle`` (device residency of one LiDAR-camera pair)."""

    def __init__(self, proj, image, points, intensities, bins, mode, precision, device, max_fov=0.0, columns_per_group=0, target_blocks=0, scale_points=0, flags=0, lds_copies=0, ext_stream=None, ext_hist=None, ext_out=None, cloud=None, cull=None, devices=None):
        if proj is None:
            raise ValueError("camera is None (create_camera failed)")
        lib = _lib.load()
        self._lib = lib
        if cloud is not None:
            points = np.zeros((0, 4))
            intensities = np.zeros(0)
            device = cloud.device
        points = np.ascontiguousarray(points, dtype=np.float64)
        intensities = np.ascontiguousarray(intensities, dtype=np.float64)
        if points.ndim != 2 or points.shape[1] != 4:
            raise ValueError("points must be (N, 4) float64 (x y z 1), like Frame::points")
        if intensities.shape != (points.shape[0],):
            raise ValueError("intensities must be (N,)")
        if mode == _lib.MODE_SPLINE:
            image = np.ascontiguousarray(image, dtype=np.float64)
            image_dtype = _lib.IMAGE_F64
        else:
            image = np.ascontiguousarray(image, dtype=np.uint8)
            image_dtype = _lib.IMAGE_U8
        d = _lib.NidregDesc()
        d.struct_size = ctypes.sizeof(_lib.NidregDesc)
        d.device_id = int(device)
        d.model_id = proj.model_id
        d.mode = mode
        d.precision = _prec(precision)
        d.bins = int(bins)
        for i in range(5):
            d.intrinsics[i] = proj._intr5[i]
        for i in range(8):
            d.distortion[i] = proj._dist8[i]
        d.height, d.width = image.shape
        d.image_dtype = image_dtype
        d.image = image.ctypes.data
        d.image_row_stride = image.strides[0]
        d.num_points = points.shape[0]
        d.points = points.ctypes.data
        d.point_stride = points.strides[0] if points.shape[0] else 32
        d.intensities = intensities.ctypes.data
        d.max_fov = float(max_fov)
        d.columns_per_group = int(columns_per_group)
        d.target_blocks = int(target_blocks)
        d.scale_points = int(scale_points)
        d.flags = int(flags)
        d.lds_copies = int(lds_copies)
        d.ext_stream = ext_stream
        d.ext_hist = ext_hist
        d.ext_out = ext_out
        if devices is not None:
            # one pair sharded over several GPUs inside the library (single process, direct GPU-to-GPU histogram exchange)
            devices = [int(v) for v in devices]
            if len(devices) > 16:
                raise ValueError("at most 16 devices")
            d.num_devices = len(devices)
            for i, v in enumerate(devices):
                d.device_ids[i] = v
            if devices:
                d.device_id = devices[0]
        h = ctypes.c_void_p()
        if cloud is None:
            _lib.check(lib.nidreg_create(ctypes.byref(d), ctypes.byref(h)), "nidreg_create")
        else:
            # cull = None | (T_camera_lidar 4x4, min_z, enable_depth_buffer_culling)
            T = None if cull is None else np.ascontiguousarray(np.asarray(cull[0], dtype=np.float64).reshape(4, 4))
            min_z = 0.0 if cull is None else float(cull[1])
            depth = 0 if cull is None else (1 if cull[2] else 0)
            _lib.check(lib.nidreg_create_from_cloud(ctypes.byref(d), cloud.c, _dp(T), min_z, depth, ctypes.byref(h)), "nidreg_create_from_cloud")
        self.h = h
        self.bins = int(bins)
        self.num_points = self.info()["num_points"] if cloud is not None else points.shape[0]

    def close(self):
        if getattr(self, "h", None):
            self._lib.nidreg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_shards(self):
        return int(self._lib.nidreg_num_shards(self.h))

    # ---- one process per GPU, RCCL inside the library (include/nidreg.h): after comm_init / attach_rccl every evaluation of
    # this handle is a collective over the communicator's ranks
    @staticmethod
    def rccl_unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().nidreg_rccl_unique_id(buf), "nidreg_rccl_unique_id")
        return buf.raw

    def comm_init(self, world_size, rank, unique_id):
        _lib.check(self._lib.nidreg_shard_comm_init(self.h, int(world_size), int(rank), bytes(unique_id)), "nidreg_shard_comm_init")

    def attach_rccl(self, comm_ptr):
        _lib.check(self._lib.nidreg_shard_attach_rccl(self.h, ctypes.c_void_p(comm_ptr) if comm_ptr else None), "nidreg_shard_attach_rccl")

    def shard_devices(self):
        ids = (ctypes.c_int * 16)()
        n = self._lib.nidreg_shard_devices(self.h, ids, 16)
        return [int(ids[i]) for i in range(min(n, 16))]

    # ---- diagnostics shared by both cost classes
    def histograms(self):
        """Raw (un-normalised) histograms of the last evaluation: joint [bin_image][bin_points],
        hist_image, hist_points."""
        B = self.bins
        joint, hi, hp = np.empty((B, B)), np.empty(B), np.empty(B)
        _lib.check(self._lib.nidreg_get_hist(self.h, _dp(joint), _dp(hi), _dp(hp)), "nidreg_get_hist")
        return joint, hi, hp

    def histogram_fixed(self):
        B = self.bins
        joint = np.empty((B, B), dtype=np.int64)
        inl = ctypes.c_int64(0)
        frac = ctypes.c_int(0)
        rc = self._lib.nidreg_get_hist_fixed(self.h, joint.ctypes.data_as(_lib.c_int64_p), ctypes.byref(inl), ctypes.byref(frac))
        _lib.check(rc, "nidreg_get_hist_fixed")
        return joint, inl.value, frac.value

    def set_timing(self, enable=True):
        """True / 1: HIP events around every kernel (the evaluation then runs as three kernels); 2: events around the
        whole evaluation, whichever route runs it (timing_ms()["total"]); False / 0: off."""
        mode = 2 if enable == 2 and enable is not True else (1 if enable else 0)
        _lib.check(self._lib.nidreg_set_timing(self.h, mode), "nidreg_set_timing")

    def timing_ms(self):
        ms = (ctypes.c_float * 6)()
        _lib.check(self._lib.nidreg_get_timing(self.h, ms), "nidreg_get_timing")
        return dict(total=ms[0], memset=ms[1], hist=ms[2], entropy=ms[3], grad=ms[4], grad_final=ms[5])

    def info(self):
        v = (ctypes.c_int64 * 8)()
        _lib.check(self._lib.nidreg_get_info(self.h, v), "nidreg_get_info")
        keys = ["record_bytes", "num_chunks", "columns_per_group", "frac_bits", "lds_bytes", "image_pitch", "num_points", "float32_records"]
        out = dict(zip(keys, [int(x) for x in v]))
        out["lds_copies"] = (out["float32_records"] >> 8) & 0xFF
        out["fused"] = (out["float32_records"] >> 5) & 1          # cost+Jacobian evaluations that have the device to themselves run as ONE launch (csrc/nid_fused.hpp)
        out["fused_chunks"] = (out["float32_records"] >> 16) & 0xFFF
        out["fused_full_stash"] = (out["float32_records"] >> 28) & 1
        out["segmented"] = (out["float32_records"] >> 1) & 1       # gradient-pass table: chunks run across column groups (looped kernels)
        out["segmented_hist"] = (out["float32_records"] >> 2) & 1  # the WIDE histogram kernel's table
        out["nearest_fast"] = (out["float32_records"] >> 3) & 1    # NEAREST: the fast decision tier is in use
        out["grad_sums_table"] = (out["float32_records"] >> 4) & 1  # B <= 32: cost+Jacobian evaluations launch no entropy kernel
        out["float32_records"] &= 1
        return out


class NIDCost(_Handle):
    """``vlcal::NIDCost`` (nid_cost.hpp:21-116) on the GPU.

    ``NIDCost(proj, normalized_image, points, intensities, bins=16)`` -- ``normalized_image`` is the
    CV_64FC1 image in [0,1] (``convertTo(CV_64FC1, 1/255)``), ``points`` the (N,4) xyz1 doubles of
    ``Frame::points``, ``intensities`` the (N,) doubles of ``Frame::intensities``.
    """

    def __init__(self, proj, normalized_image, points, intensities, bins=16, device=0, precision="fp64", **tuning):
        super().__init__(proj, normalized_image, points, intensities, bins, _lib.MODE_SPLINE, precision, device, **tuning)

    @classmethod
    def from_cloud(cls, proj, normalized_image, cloud, bins=16, cull=None, precision="fp64", **tuning):
        """``cull -> new NIDCost`` (visual_camera_calibration.cpp:201-206) fused on the device: ``cull`` is
        ``None`` or ``(T_camera_lidar 4x4, min_z, enable_depth_buffer_culling)``."""
        self = cls.__new__(cls)
        _Handle.__init__(self, proj, normalized_image, None, None, bins, _lib.MODE_SPLINE, precision, cloud.device, cloud=cloud, cull=cull, **tuning)
        return self

    def __call__(self, T_camera_lidar_params, want_grad=True):
        """``operator()(params, residual)``: params = [qx qy qz qw tx ty tz].  Returns
        ``(ok, cost, grad7)``; ``ok`` False = the functor's ``return false`` (non-finite NID).
        ``want_grad=False`` is the T=double instantiation (cost only, ``grad7`` None)."""
        # staging buffers and their pointers are created once: the marshalling of a call is then one 7-double copy
        # (ctypes pointer construction per call cost more than 10 us of a 180 us evaluation)
        st = self.__dict__.get("_stage")
        if st is None:
            x, g, c = np.empty(7), np.empty(7), ctypes.c_double(float("nan"))
            st = self._stage = (x, g, c, _dp(x), _dp(g), ctypes.byref(c))
        x, g, c, xp, gp, cp = st
        x[:] = T_camera_lidar_params
        c.value = float("nan")
        rc = self._lib.nidreg_eval(self.h, xp, cp, gp if want_grad else None)
        if rc < 0:
            _lib.check(rc, "nidreg_eval")
        return rc == _lib.NIDREG_OK, c.value, (g.copy() if want_grad else None)

    def eval_batch(self, poses, want_grad=True, pipelined=False):
        """``n`` evaluations back to back inside the library (an optimiser's inner loop without the per-call Python /
        ctypes overhead): synchronous -- each completes, host sync included, before the next starts -- or, for INDEPENDENT
        poses, ``pipelined`` through the submit / wait pair (no host round trip between them).  Returns
        ``(all_ok, costs[n], grads[n,7] | None)``."""
        x = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
        costs = np.empty(x.shape[0])
        grads = np.empty((x.shape[0], 7)) if want_grad else None
        fn = self._lib.nidreg_eval_pipelined if pipelined else self._lib.nidreg_eval_batch
        rc = _lib.check(fn(self.h, _dp(x), x.shape[0], _dp(costs), _dp(grads)), "nidreg_eval_pipelined" if pipelined else "nidreg_eval_batch")
        return rc == _lib.NIDREG_OK, costs, grads

    def submit(self, T_camera_lidar_params, want_grad=True):
        """``nidreg_submit``: queue one evaluation, return its ticket (at most 8 in flight per handle)."""
        x = np.ascontiguousarray(T_camera_lidar_params, dtype=np.float64).reshape(7)
        t = ctypes.c_int64(0)
        _lib.check(self._lib.nidreg_submit(self.h, _dp(x), 1 if want_grad else 0, ctypes.byref(t)), "nidreg_submit")
        return (t.value, want_grad)

    def wait(self, ticket):
        """``nidreg_wait``: ``(ok, cost, grad | None)`` of the evaluation behind ``ticket`` (from :meth:`submit`)."""
        t, want_grad = ticket
        c = ctypes.c_double(float("nan"))
        g = np.empty(7) if want_grad else None
        rc = _lib.check(self._lib.nidreg_wait(self.h, t, ctypes.byref(c), _dp(g)), "nidreg_wait")
        return rc == _lib.NIDREG_OK, c.value, g

    # split-phase API for point-sharded multi-GPU evaluation (see parallel.ShardedNIDCost)
    def shard_hist(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        _lib.check(self._lib.nidreg_shard_hist(self.h, _dp(x)), "nidreg_shard_hist")

    def shard_entropy(self):
        _lib.check(self._lib.nidreg_shard_entropy(self.h), "nidreg_shard_entropy")

    def shard_grad(self):
        _lib.check(self._lib.nidreg_shard_grad(self.h), "nidreg_shard_grad")

    def shard_finish(self, want_grad=True):
        cost = ctypes.c_double(float("nan"))
        grad = np.empty(7) if want_grad else None
        rc = _lib.check(self._lib.nidreg_shard_finish(self.h, ctypes.byref(cost), _dp(grad)), "nidreg_shard_finish")
        return rc == _lib.NIDREG_OK, cost.value, grad


class MultiNIDCost:
    """``MultiNIDCost`` (visual_camera_calibration.cpp:141-178): trust gate (0.2 m / 2 deg from
    ``init_T_camera_lidar``), all pairs evaluated concurrently, residual = plain sum, ``False`` if
    the gate rejects or any pair failed."""

    def __init__(self, init_T_camera_lidar):
        self.init = None if init_T_camera_lidar is None else np.ascontiguousarray(init_T_camera_lidar, dtype=np.float64)
        self.costs = []

    def add(self, cost):
        self.costs.append(cost)

    def __call__(self, params, want_grad=True):
        lib = _lib.load()
        x = np.ascontiguousarray(params, dtype=np.float64)
        n = len(self.costs)
        arr = (ctypes.c_void_p * n)(*[c.h for c in self.costs])
        cost = ctypes.c_double(float("nan"))
        grad = np.empty(7) if want_grad else None
        rc = _lib.check(lib.nidreg_eval_multi(arr, n, _dp(self.init), _dp(x), ctypes.byref(cost), _dp(grad)), "nidreg_eval_multi")
        return rc == _lib.NIDREG_OK, cost.value, grad


def estimate_direction(proj, pt_2d, device=-1):
    """estimate_fov.cpp:17-34: invert the projection at one pixel with NelderMead<2> defaults.  Host work in the reference
    and here (device = -1 = NIDREG_DEVICE_HOST: the device's scalar projection code compiled for the host; ~80 probes of
    one point each -- on the GPU they were 8 of the 13 ms a whole configs[0] calibration took)."""

    def to_dir(x):
        # AngleAxis(x0, X) * AngleAxis(x1, Y) * UnitZ through quaternions, as Eigen evaluates it
        aw, ax = math.cos(0.5 * x[0]), math.sin(0.5 * x[0])
        bw, by = math.cos(0.5 * x[1]), math.sin(0.5 * x[1])
        qw, qx, qy, qz = aw * bw, ax * bw, aw * by, ax * by
        ux, uy, uz = 2.0 * qy, -2.0 * qx, 0.0  # 2 (vec x ez)
        return np.array([qw * ux + (qy * uz - qz * uy), qw * uy + (qz * ux - qx * uz), (1.0 + qw * uz) + (qx * uy - qy * ux)])

    def f(x):
        uv = proj.project(to_dir(x), device=device)
        err = float((pt_2d[0] - uv[0]) ** 2 + (pt_2d[1] - uv[1]) ** 2)
        return err if math.isfinite(err) else float(np.finfo(np.float64).max)

    result = NelderMead(NelderMeadParams()).optimize(f, np.zeros(2))
    return to_dir(result.x)


def estimate_camera_fov(proj, image_size, device=None):
    """estimate_fov.cpp:36-51: max view angle over (0,0), (W/2,0), (0,H/2) (integer division).  Host work in the reference
    and here: nidreg_estimate_camera_fov runs the three NelderMead<2> inversions natively on the device's scalar projection
    code compiled for the host (`device` is accepted for compatibility and ignored).  ``estimate_direction`` above is the
    same procedure spelled out in Python (tests compare the two)."""
    out = ctypes.c_double(0.0)
    rc = _lib.load().nidreg_estimate_camera_fov(proj.model_id, _dp(proj._intr5), _dp(proj._dist8), int(image_size[0]), int(image_size[1]), ctypes.byref(out))
    _lib.check(rc, "nidreg_estimate_camera_fov")
    return out.value


def estimate_camera_fov_py(proj, image_size):
    """The same through ``estimate_direction`` (Python NelderMead, one host projection per probe)."""
    w, h = int(image_size[0]), int(image_size[1])
    corners = [(0.0, 0.0), (float(w // 2), 0.0), (0.0, float(h // 2))]
    max_fov = 0.0
    for c in corners:
        d = estimate_direction(proj, c, device=-1)
        n = np.linalg.norm(d)
        fov = math.acos((d / n)[2] if n > 0 else d[2])
        if fov > max_fov:
            max_fov = fov
    return max_fov


class ViewCullingParams:
    """view_culling.hpp:8-16"""

    def __init__(self, enable_depth_buffer_culling=True):
        self.enable_depth_buffer_culling = enable_depth_buffer_culling


class ViewCulling:
    """``vlcal::ViewCulling`` (view_culling.hpp:18-39) on the GPU: ``cull(points, T_camera_lidar)`` returns
    the indices of the points that survive the FoV gate, the in-image test and the depth buffer --
    the same list, in the same order, as the reference's sequential loop."""

    def __init__(self, proj, image_size, params=None, device=0, min_z=None):
        self.proj = proj
        self.image_size = (int(image_size[0]), int(image_size[1]))
        self.params = params or ViewCullingParams()
        self.device = device
        # view_culling.cpp:17: min_z(cos(estimate_camera_fov(proj, image_size)))
        self.min_z = math.cos(estimate_camera_fov(proj, image_size, device=device)) if min_z is None else float(min_z)

    def cull(self, points, T_camera_lidar):
        lib = _lib.load()
        pts = np.ascontiguousarray(points, dtype=np.float64)
        T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
        idx = np.empty(pts.shape[0], dtype=np.int32)
        n = lib.nidreg_view_culling(self.proj.model_id, _dp(self.proj._intr5), _dp(self.proj._dist8), self.device, self.image_size[0], self.image_size[1], self.min_z,
                                    1 if self.params.enable_depth_buffer_culling else 0, _dp(pts), pts.strides[0] if pts.shape[0] else 32, pts.shape[0], _dp(T),
                                    idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        _lib.check(n, "nidreg_view_culling")
        return idx[:n].copy()


class NIDCostParams:
    """cost_calculator_nid.cpp:7-9"""

    def __init__(self, bins=16):
        self.bins = bins


class CostCalculatorNID(_Handle):
    """``vlcal::CostCalculatorNID`` on the GPU.  ``image`` is the 8-bit gray image
    (``VisualLiDARData::image``); ``max_fov`` defaults to ``estimate_camera_fov(proj, size)`` like
    the reference constructor (cost_calculator_nid.cpp:13-17)."""

    def __init__(self, proj, image, points, intensities, params=None, max_fov=None, device=0, precision="fp64", **tuning):
        params = params or NIDCostParams()
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if max_fov is None:
            max_fov = estimate_camera_fov(proj, (image.shape[1], image.shape[0]), device=device)
        self.max_fov = float(max_fov)
        super().__init__(proj, image, points, intensities, params.bins, _lib.MODE_NEAREST, precision, device, max_fov=self.max_fov, **tuning)

    @classmethod
    def from_cloud(cls, proj, image, cloud, params=None, max_fov=None, cull=None, precision="fp64", **tuning):
        """``cull -> new CostCalculatorNID`` (visual_camera_calibration.cpp:76-85) fused on the device."""
        params = params or NIDCostParams()
        image = np.ascontiguousarray(image, dtype=np.uint8)
        if max_fov is None:
            max_fov = estimate_camera_fov(proj, (image.shape[1], image.shape[0]), device=cloud.device)
        self = cls.__new__(cls)
        self.max_fov = float(max_fov)
        _Handle.__init__(self, proj, image, None, None, params.bins, _lib.MODE_NEAREST, precision, cloud.device, max_fov=self.max_fov, cloud=cloud, cull=cull, **tuning)
        return self

    def calculate(self, T_camera_lidar):
        """``calculate(const Eigen::Isometry3d&)``: 4x4 matrix -> NID (no finite check, like the
        reference)."""
        T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
        cost = ctypes.c_double(float("nan"))
        _lib.check(self._lib.nidreg_eval_iso(self.h, _dp(T), ctypes.byref(cost)), "nidreg_eval_iso")
        return cost.value

    def submit(self, T_camera_lidar):
        """``nidreg_submit_iso``: queue one ``calculate``, return its ticket (at most 8 in flight per handle)."""
        T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
        t = ctypes.c_int64(0)
        _lib.check(self._lib.nidreg_submit_iso(self.h, _dp(T), ctypes.byref(t)), "nidreg_submit_iso")
        return t.value

    def wait(self, ticket):
        cost = ctypes.c_double(float("nan"))
        _lib.check(self._lib.nidreg_wait(self.h, ticket, ctypes.byref(cost), None), "nidreg_wait")
        return cost.value


def sum_costs(costs, T_camera_lidar):
    """The Nelder-Mead objective's ``sum_i costs[i]->calculate(T)`` (visual_camera_calibration.cpp:
    106-110), all pairs in flight at once."""
    lib = _lib.load()
    T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
    n = len(costs)
    arr = (ctypes.c_void_p * n)(*[c.h for c in costs])
    cost = ctypes.c_double(float("nan"))
    _lib.check(lib.nidreg_eval_iso_multi(arr, n, _dp(T), ctypes.byref(cost)), "nidreg_eval_iso_multi")
    return cost.value
