"""CPU tests of the C-ABI boundary: the HIP library builds for gfx950, loads, and exports exactly the
entry points include/nidreg.h declares.  No compute calls (there is no GPU here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "nidreg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nidreg_[a-z0-9_]+)\s*\(", src)))


def test_build_entry_point():
    import __graft_entry__

    __graft_entry__.build()


def test_library_exports_every_declared_symbol():
    from direct_visual_lidar_calibration_amd import _lib

    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libnidreg.so does not export {s}"
    assert sorted(_lib.EXPORTS) == syms
    assert b"gfx950" in lib.nidreg_version()


def test_desc_struct_layout_matches_header():
    """sizeof(nidreg_desc) seen by the C compiler == the ctypes mirror."""
    from direct_visual_lidar_calibration_amd import _lib

    code = '#include <stdio.h>\n#include "nidreg.h"\nint main(){printf("%zu %zu %zu\\n", sizeof(nidreg_desc), __builtin_offsetof(nidreg_desc, image), __builtin_offsetof(nidreg_desc, ext_out));return 0;}\n'
    exe = "/tmp/_nidreg_sizeof"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=code.encode(), check=True)
    out = subprocess.check_output([exe]).decode().split()
    assert int(out[0]) == ctypes.sizeof(_lib.NidregDesc)
    assert int(out[1]) == _lib.NidregDesc.image.offset
    assert int(out[2]) == _lib.NidregDesc.ext_out.offset


def test_model_table_and_no_device_error():
    from direct_visual_lidar_calibration_amd import _lib, nid

    lib = _lib.load()
    ni, nd = ctypes.c_int(), ctypes.c_int()
    table = {"plumb_bob": (0, 4, 5), "fisheye": (1, 4, 4), "equidistant": (1, 4, 4), "omnidir": (2, 5, 4), "equirectangular": (3, 2, 0), "atan": (4, 4, 1),
             "rational_polynomial": (5, 4, 8)}
    for name, (mid, a, b) in table.items():
        assert lib.nidreg_model_from_name(name.encode(), ctypes.byref(ni), ctypes.byref(nd)) == mid
        assert (ni.value, nd.value) == (a, b)
    assert lib.nidreg_model_from_name(b"pinhole", None, None) == -1
    assert nid.create_camera("pinhole", [1, 2, 3, 4], []) is None
    # bins: 2..4096 like the reference's `int bins` (src/calibrate.cpp:175 --nid_bins); beyond 256 only while <= 256 bins per
    # axis are OCCUPIED (8-bit images, 256-level intensities: the reference's own data path) -- an input that occupies more is
    # refused with a message that says so, before any device is touched; never truncated
    import numpy as np

    cam0 = nid.create_camera("plumb_bob", [100, 100, 50, 50], [])
    for bad in (4097, 100000, 1, 0, -3):
        with pytest.raises(RuntimeError, match=r"bins must be in \[2, 4096\]"):
            nid.NIDCost(cam0, np.zeros((100, 100)), np.zeros((4, 4)), np.zeros(4), bad)
        with pytest.raises(RuntimeError, match=r"bins must be in \[2, 4096\]"):
            nid.CostCalculatorNID(cam0, np.zeros((100, 100), dtype=np.uint8), np.zeros((4, 4)), np.zeros(4), nid.NIDCostParams(bad), max_fov=1.0)
    ramp = np.linspace(0.0, 1.0, 100 * 100).reshape(100, 100)  # 10^4 distinct grey values: 512 occupied image bins at bins = 512
    with pytest.raises(RuntimeError, match=r"512 occupied image bins .* refused, not truncated"):
        nid.NIDCost(cam0, ramp, np.zeros((4, 4)), np.zeros(4), 512)
    with pytest.raises(RuntimeError, match=r"300 occupied intensity bins"):
        nid.NIDCost(cam0, np.zeros((100, 100)), np.zeros((300, 4)), (np.arange(300) + 0.5) / 300.0, 300)
    if lib.nidreg_device_count() == 0:
        # the product path fails loudly without a GPU: no CPU fallback
        import numpy as np

        cam = nid.create_camera("plumb_bob", [100, 100, 50, 50], [])
        with pytest.raises(RuntimeError, match="no HIP device"):
            nid.NIDCost(cam, np.zeros((100, 100)), np.zeros((4, 4)), np.zeros(4), 16)


def test_library_carries_the_hash_of_the_kernel_sources_it_was_built_from(monkeypatch):
    """nidreg_kernel_build() = sha256 over the kernel sources at build time; the measurement tools stamp rocprofv3 summaries with it
    and REFUSE when the library is stale against the sources on disk (a summary re-stamped by hand)."""
    from direct_visual_lidar_calibration_amd import _lib

    built = _lib.library_kernel_build()
    assert len(built) == 16 and built == _lib.kernel_source_hash() == _lib.stamp_or_refuse()
    monkeypatch.setattr(_lib, "kernel_source_hash", lambda: "0123456789abcdef")
    with pytest.raises(SystemExit, match="refusing to stamp"):
        _lib.stamp_or_refuse()
