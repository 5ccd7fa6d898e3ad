"""ctypes loader for csrc/libnidreg.so (the C ABI of include/nidreg.h).

There is no Python or CPU fallback: if the HIP library is missing or no GPU is usable, loading /
handle creation raises.  Build it with ``__graft_entry__.build()`` (``make -C csrc``).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, "csrc")
# NIDREG_LIB lets development tools (tools/ablate.py) load an instrumented build of the same ABI
LIB_PATH = os.environ.get("NIDREG_LIB", os.path.join(CSRC_DIR, "libnidreg.so"))

NIDREG_OK = 0
NIDREG_FALSE = 1
NIDREG_OUT_DOUBLES = 16

MODE_SPLINE, MODE_NEAREST = 0, 1
PREC_FP64 = 0  # (1 = NIDREG_PREC_FP32: removed in round 5, refused by nidreg_create)
IMAGE_F64, IMAGE_U8 = 0, 1
FLAG_INPUT_ORDER = 1
FLAG_EXT_STREAM = 2
FLAG_NEAREST_EXACT = 4

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_float_p = ctypes.POINTER(ctypes.c_float)


class NidregDesc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("device_id", ctypes.c_int32),
        ("model_id", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("bins", ctypes.c_int32),
        ("intrinsics", ctypes.c_double * 5),
        ("distortion", ctypes.c_double * 8),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("image_dtype", ctypes.c_int32),
        ("flags", ctypes.c_int32),
        ("image", ctypes.c_void_p),
        ("image_row_stride", ctypes.c_int64),
        ("num_points", ctypes.c_int64),
        ("points", ctypes.c_void_p),
        ("point_stride", ctypes.c_int64),
        ("intensities", ctypes.c_void_p),
        ("max_fov", ctypes.c_double),
        ("columns_per_group", ctypes.c_int32),
        ("target_blocks", ctypes.c_int32),
        ("lds_copies", ctypes.c_int32),
        ("reserved1", ctypes.c_int32),
        ("scale_points", ctypes.c_int64),
        ("ext_stream", ctypes.c_void_p),
        ("ext_hist", ctypes.c_void_p),
        ("ext_out", ctypes.c_void_p),
        ("num_devices", ctypes.c_int32),
        ("device_ids", ctypes.c_int32 * 16),
    ]


EXPORTS = [
    "nidreg_model_from_name", "nidreg_device_count", "nidreg_create", "nidreg_destroy", "nidreg_cloud_create", "nidreg_cloud_destroy", "nidreg_create_from_cloud", "nidreg_eval", "nidreg_eval_iso", "nidreg_eval_multi",
    "nidreg_eval_iso_multi", "nidreg_get_hist", "nidreg_get_hist_fixed", "nidreg_project", "nidreg_project_model", "nidreg_view_culling", "nidreg_hist_words", "nidreg_shard_hist",
    "nidreg_shard_entropy", "nidreg_shard_grad", "nidreg_shard_finish", "nidreg_set_timing", "nidreg_get_timing", "nidreg_get_info", "nidreg_last_error",
    "nidreg_version", "nidreg_colorizer_create", "nidreg_colorizer_update", "nidreg_colorizer_device_colors", "nidreg_colorizer_destroy", "nidreg_generate_lidar_image",
    "nidreg_equalize_intensities", "nidreg_num_shards", "nidreg_shard_devices", "nidreg_trim", "nidreg_eval_batch", "nidreg_submit", "nidreg_submit_iso", "nidreg_wait", "nidreg_eval_pipelined",
    "nidreg_estimate_camera_fov", "nidreg_rccl_unique_id", "nidreg_shard_comm_init", "nidreg_shard_attach_rccl", "nidreg_kernel_build",
]

_lib = None


def load():
    """Load libnidreg.so.  Raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: the NID core has no CPU fallback; run __graft_entry__.build() (make -C {CSRC_DIR})")
    lib = ctypes.CDLL(LIB_PATH)
    lib.nidreg_last_error.restype = ctypes.c_char_p
    lib.nidreg_version.restype = ctypes.c_char_p
    lib.nidreg_hist_words.restype = ctypes.c_int64
    lib.nidreg_hist_words.argtypes = [ctypes.c_int]
    lib.nidreg_destroy.restype = None
    lib.nidreg_destroy.argtypes = [ctypes.c_void_p]
    lib.nidreg_create.argtypes = [ctypes.POINTER(NidregDesc), ctypes.POINTER(ctypes.c_void_p)]
    lib.nidreg_cloud_create.argtypes = [ctypes.c_int, c_double_p, ctypes.c_int64, c_double_p, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
    lib.nidreg_cloud_destroy.restype = None
    lib.nidreg_cloud_destroy.argtypes = [ctypes.c_void_p]
    lib.nidreg_create_from_cloud.argtypes = [ctypes.POINTER(NidregDesc), ctypes.c_void_p, c_double_p, ctypes.c_double, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.nidreg_eval.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]
    lib.nidreg_eval_iso.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
    lib.nidreg_eval_batch.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p, c_double_p]
    lib.nidreg_eval_pipelined.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p, c_double_p]
    lib.nidreg_submit.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    lib.nidreg_submit_iso.argtypes = [ctypes.c_void_p, c_double_p, ctypes.POINTER(ctypes.c_int64)]
    lib.nidreg_wait.argtypes = [ctypes.c_void_p, ctypes.c_int64, c_double_p, c_double_p]
    lib.nidreg_eval_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.nidreg_eval_iso_multi.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_double_p, c_double_p]
    lib.nidreg_get_hist.argtypes = [ctypes.c_void_p, c_double_p, c_double_p, c_double_p]
    lib.nidreg_get_hist_fixed.argtypes = [ctypes.c_void_p, c_int64_p, c_int64_p, ctypes.POINTER(ctypes.c_int)]
    lib.nidreg_project.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_int64, c_double_p, c_double_p]
    lib.nidreg_project_model.argtypes = [ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p, ctypes.c_int64, c_double_p, c_double_p]
    lib.nidreg_view_culling.restype = ctypes.c_int64
    lib.nidreg_view_culling.argtypes = [ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, c_double_p, ctypes.c_int64,
                                        ctypes.c_int64, c_double_p, ctypes.POINTER(ctypes.c_int32)]
    lib.nidreg_shard_hist.argtypes = [ctypes.c_void_p, c_double_p]
    lib.nidreg_shard_entropy.argtypes = [ctypes.c_void_p]
    lib.nidreg_shard_grad.argtypes = [ctypes.c_void_p]
    lib.nidreg_shard_finish.argtypes = [ctypes.c_void_p, c_double_p, c_double_p]
    lib.nidreg_set_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.nidreg_get_timing.argtypes = [ctypes.c_void_p, c_float_p]
    lib.nidreg_get_info.argtypes = [ctypes.c_void_p, c_int64_p]
    lib.nidreg_model_from_name.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.nidreg_colorizer_create.argtypes = [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, c_double_p,
                                            ctypes.c_int64, c_float_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]
    lib.nidreg_colorizer_update.argtypes = [ctypes.c_void_p, c_double_p, ctypes.c_double, c_float_p]
    lib.nidreg_colorizer_device_colors.restype = ctypes.c_void_p
    lib.nidreg_colorizer_device_colors.argtypes = [ctypes.c_void_p]
    lib.nidreg_colorizer_destroy.restype = None
    lib.nidreg_colorizer_destroy.argtypes = [ctypes.c_void_p]
    lib.nidreg_generate_lidar_image.argtypes = [ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_double_p, ctypes.c_int64, c_double_p,
                                                ctypes.c_int64, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int32)]
    lib.nidreg_equalize_intensities.argtypes = [ctypes.c_int, c_double_p, ctypes.c_int64]
    lib.nidreg_num_shards.argtypes = [ctypes.c_void_p]
    lib.nidreg_shard_devices.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.nidreg_trim.restype = None
    lib.nidreg_trim.argtypes = []
    lib.nidreg_estimate_camera_fov.argtypes = [ctypes.c_int, c_double_p, c_double_p, ctypes.c_int, ctypes.c_int, c_double_p]
    lib.nidreg_rccl_unique_id.argtypes = [ctypes.c_char_p]
    lib.nidreg_shard_comm_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    lib.nidreg_shard_attach_rccl.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    _lib = lib
    return lib


def kernel_source_hash():
    """sha256 (16 hex digits) over the sources of the evaluation kernels (device code, launch wrappers, build flags --
    not the host side of the ABI): stamps PMC summaries so that bench.py never reports counter traffic measured on a
    different kernel build."""
    import hashlib

    h = hashlib.sha256()
    names = ["nid_device.hpp", "nid_atan_table.hpp", "nid_log_table.hpp", "nid_multi.hpp", "nid_kernels.hpp", "nid_fused.hpp", "nid_launch_impl.hpp", "nid_kernels_f64.hip", "nid_kernels_f64_exact.hip", "nid_kernels_fused.hip", "Makefile"]
    for path in [os.path.join(CSRC_DIR, n) for n in names]:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def library_kernel_build():
    """The kernel-source hash libnidreg.so was BUILT from (nidreg_kernel_build): equals kernel_source_hash() unless the library is
    stale against the sources next to it."""
    lib = load()
    lib.nidreg_kernel_build.restype = ctypes.c_char_p
    return lib.nidreg_kernel_build().decode()


def stamp_or_refuse():
    """For the measurement tools: the hash to stamp a summary with -- refuses (SystemExit) when the loaded library was not built from the
    sources on disk, because the summary would then be attributed to kernels that did not produce it."""
    built, tree = library_kernel_build(), kernel_source_hash()
    if built != tree:
        raise SystemExit(f"refusing to stamp: libnidreg.so was built from kernel sources {built}, the tree holds {tree} (rebuild, then measure again)")
    return built


def last_error():
    return load().nidreg_last_error().decode()


def check(rc, what):
    if rc < 0:
        raise RuntimeError(f"{what} failed ({rc}): {last_error()}")
    return rc
