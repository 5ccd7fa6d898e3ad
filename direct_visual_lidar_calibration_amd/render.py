"""Host-side mirror of the two point-to-image components next to the NID path.

==========================================  =========================================================
here                                        reference
==========================================  =========================================================
``PointsColorUpdater(proj, image, points)`` ``vlcal::PointsColorUpdater`` (points_color_updater.cpp:26-35)
``PointsColorUpdater.update(T, w)``         ``PointsColorUpdater::update`` (:37-61), returns the colours
                                            instead of pushing them to the viewer
``generate_lidar_image(proj, size, T, ..)`` ``vlcal::generate_lidar_image`` (generate_lidar_image.cpp:7-41)
``equalize_intensities(intensities)``       the rank equalisation loop (preprocess.cpp:464-473)
``colormap_turbo(x)``                       ``glk::colormapf(glk::COLORMAP::TURBO, x)`` (Iridescence, not in
                                            the reference tree: polynomial fit of the TURBO map, display only)
==========================================  =========================================================

All per-point work runs in the HIP library behind ``include/nidreg.h``; no CPU implementation exists here.
"""
import ctypes
import math

import numpy as np

from . import _lib
from .nid import estimate_camera_fov


def _dp(a):
    return None if a is None else a.ctypes.data_as(_lib.c_double_p)


def colormap_turbo(x):
    """RGBA float32 colours for values in [0, 1] (clamped).  The reference takes these from Iridescence's
    256-entry TURBO table, which is not part of the reference tree; this is the published degree-5
    polynomial fit of the same map (display only: nothing on the cost path depends on it)."""
    x = np.clip(np.asarray(x, dtype=np.float64), 0.0, 1.0)
    r = 0.13572138 + x * (4.61539260 + x * (-42.66032258 + x * (132.13108234 + x * (-152.94239396 + x * 59.28637943))))
    g = 0.09140261 + x * (2.19418839 + x * (4.84296658 + x * (-14.18503333 + x * (4.27729857 + x * 2.82956604))))
    b = 0.10667330 + x * (12.64194608 + x * (-60.58204836 + x * (110.36276771 + x * (-89.90310912 + x * 27.34824973))))
    rgba = np.stack([r, g, b, np.ones_like(x)], axis=-1)
    return np.clip(rgba, 0.0, 1.0).astype(np.float32)


class PointsColorUpdater:
    """``vlcal::PointsColorUpdater``: colours a cloud with the camera image under a candidate extrinsic.
    The constructor uploads image, points and intensity colours once; ``update`` is one kernel."""

    def __init__(self, proj, image, points, intensities=None, intensity_colors=None, device=0):
        lib = _lib.load()
        self.proj = proj
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if img.ndim != 2:
            raise ValueError("PointsColorUpdater: 8-bit single-channel image expected")
        self.image = img
        pts = np.ascontiguousarray(points, dtype=np.float64)
        if pts.ndim != 2 or pts.shape[1] != 4:
            raise ValueError("PointsColorUpdater: points must be (n, 4) homogeneous doubles")
        self.num_points = pts.shape[0]
        # points_color_updater.cpp:12 / :28: min_nz = cos(estimate_camera_fov + 0.5 deg)
        self.min_nz = math.cos(estimate_camera_fov(proj, (img.shape[1], img.shape[0]), device=device) + 0.5 * math.pi / 180.0)
        if intensity_colors is None and intensities is not None:
            intensity_colors = colormap_turbo(intensities)  # :33-35
        ic = None if intensity_colors is None else np.ascontiguousarray(intensity_colors, dtype=np.float32).reshape(-1, 4)
        if ic is not None and ic.shape[0] != self.num_points:
            raise ValueError("PointsColorUpdater: one RGBA colour per point expected")
        self.intensity_colors = ic
        h = ctypes.c_void_p()
        rc = lib.nidreg_colorizer_create(device, proj.model_id, _dp(proj._intr5), _dp(proj._dist8), img.shape[1], img.shape[0], img.ctypes.data_as(ctypes.c_void_p), img.strides[0],
                                         self.num_points, _dp(pts), pts.strides[0] if self.num_points else 32,
                                         None if ic is None else ic.ctypes.data_as(_lib.c_float_p), self.min_nz, ctypes.byref(h))
        _lib.check(rc, "nidreg_colorizer_create")
        self._h = h

    def update(self, T_camera_lidar, blend_weight):
        """Returns the (n, 4) float32 RGBA colours the reference hands to ``cloud_buffer->add_color``."""
        T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
        out = np.empty((self.num_points, 4), dtype=np.float32)
        rc = _lib.load().nidreg_colorizer_update(self._h, _dp(T), float(blend_weight), out.ctypes.data_as(_lib.c_float_p))
        _lib.check(rc, "nidreg_colorizer_update")
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().nidreg_colorizer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_lidar_image(proj, image_size, T_camera_lidar, points, intensities, device=0, min_z=None):
    """``vlcal::generate_lidar_image``: returns ``(intensity_image float64 HxW, index_image int32 HxW)``."""
    lib = _lib.load()
    w, h = int(image_size[0]), int(image_size[1])
    if min_z is None:
        min_z = math.cos(estimate_camera_fov(proj, (w, h), device=device))  # generate_lidar_image.cpp:10-11
    pts = np.ascontiguousarray(points, dtype=np.float64)
    inten = np.ascontiguousarray(intensities, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 4 or inten.shape[0] != pts.shape[0]:
        raise ValueError("generate_lidar_image: points (n, 4) and intensities (n,) expected")
    T = np.ascontiguousarray(np.asarray(T_camera_lidar, dtype=np.float64).reshape(4, 4))
    iimg = np.empty((h, w), dtype=np.float64)
    idx = np.empty((h, w), dtype=np.int32)
    rc = lib.nidreg_generate_lidar_image(proj.model_id, _dp(proj._intr5), _dp(proj._dist8), device, w, h, float(min_z), _dp(pts), pts.strides[0] if pts.shape[0] else 32, _dp(inten),
                                         pts.shape[0], _dp(T), _dp(iimg), idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    _lib.check(rc, "nidreg_generate_lidar_image")
    return iimg, idx


def equalize_intensities(intensities, device=0):
    """preprocess.cpp:464-473: rank-equalised copy of the intensities (256 levels in [0, 1))."""
    a = np.array(intensities, dtype=np.float64, copy=True)
    rc = _lib.load().nidreg_equalize_intensities(device, _dp(a), a.shape[0])
    _lib.check(rc, "nidreg_equalize_intensities")
    return a
