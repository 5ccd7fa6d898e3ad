"""Register / scratch budget of the hot kernels, read from the gfx950 assembly hipcc emits for the double-precision
translation unit (no GPU needed: hipcc cross-compiles).  The two spline passes of the headline configuration run 16 waves
per CU -- two 8-wave histogram workgroups, four 4-wave gradient workgroups -- which needs <= 128 VGPRs per lane and no
scratch; the chunk tables are sized from that occupancy (nidreg_plan.hip), so a change that pushes a kernel over the limit
silently costs a quarter of the latency hiding ("forcing 4 waves/EU ... +55 %")."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "direct_visual_lidar_calibration_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def kernel_metadata(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "nid_kernels_f64.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", os.path.join(CSRC, "nid_kernels_f64.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    meta = {}
    # the amdhsa.kernels YAML block: one entry per kernel with .name, .vgpr_count, .sgpr_spill_count, .vgpr_spill_count, .private_segment_fixed_size
    for block in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", block).group(1))  # noqa: E731
        meta[name.group(1)] = {"vgpr": get("vgpr_count"), "vgpr_spill": get("vgpr_spill_count"), "scratch": get("private_segment_fixed_size"), "lds_static": get("group_segment_fixed_size")}
    assert meta, "no kernel metadata found in the assembly"
    return text, meta


def _find(meta, fragment):
    hits = [k for k in meta if fragment in k]
    assert hits, fragment
    return {k: meta[k] for k in hits}


def test_headline_spline_kernels_fit_four_waves_per_simd(kernel_metadata):
    _, meta = kernel_metadata
    # k_spline_hist<plumb_bob, Rec32, double, WIDE, single / multi pair>, k_spline_grad<plumb_bob, Rec32, double, GW1, ...>
    for fragment in ("k_spline_histILi0ENS_5Rec32EdLb1E", "k_spline_gradILi0ENS_5Rec32EdLb1E"):
        for name, m in _find(meta, fragment).items():
            assert m["vgpr"] <= 128, (name, m)
            assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)


def test_no_spline_kernel_uses_scratch(kernel_metadata):
    _, meta = kernel_metadata
    for name, m in {**_find(meta, "k_spline_hist"), **_find(meta, "k_spline_grad")}.items():
        assert m["vgpr"] <= 168, (name, m)  # at least three waves per SIMD for every camera model
        looped = re.search(r"Lb1EEEv", name) is not None  # template <MODEL, Rec, real, WIDE | GW1, MULTI, SEG>: SEG = the segment loop
        if "k_spline_gradILi4E" in name:
            # the `atan` model (generic Dual3 forward mode) is held at three waves per SIMD by its launch bounds and may park a
            # few registers in scratch outside the point loop
            assert m["vgpr_spill"] <= 8 and m["scratch"] <= 64, (name, m)
        elif looped:
            # the looped (SEG) instantiations on float records are compiled for four waves per SIMD and may park a few
            # registers in scratch OUTSIDE the point loop (checked below: round 5's omnidir multi-pair kernel parks three, around
            # the segment loop's head, tail and the reduction); the double-record ones are left alone
            assert m["vgpr_spill"] <= 4 and m["scratch"] <= 16, (name, m)
        else:
            # the straight-line kernels (every table whose chunks lie inside one column group: the headline) are those of round 3
            assert m["vgpr_spill"] == 0 and m["scratch"] == 0, (name, m)


def test_looped_kernels_keep_scratch_out_of_the_point_loop(kernel_metadata):
    """Round 4: chunks may run across column groups; the SEG instantiations wrap the point loop in a segment loop.  Whatever
    they spill must stay outside the innermost loops (the 16 taps per point)."""
    text, meta = kernel_metadata
    for name, m in {**_find(meta, "k_spline_hist"), **_find(meta, "k_spline_grad")}.items():
        if not m["scratch"] or "ILi4E" in name:
            continue
        start = text.index("\n" + name + ":")
        body = text[start:text.index(".Lfunc_end", start)]
        # basic blocks that hold tap code: many LDS operations
        for block in re.split(r"\n\.LBB\d+_\d+:", body):
            lds = len(re.findall(r"^\s+ds_(add_u64|read_b64|add_rtn_u64)", block, flags=re.M))
            if lds >= 16:
                assert "scratch_" not in block, name


def test_every_model_but_atan_fits_four_waves_per_simd_in_the_gradient_pass(kernel_metadata):
    """Round 3: fast_atan2 from a table took the fisheye / equirectangular gradient kernels from 154-162 to <= 128 VGPRs."""
    _, meta = kernel_metadata
    for name, m in _find(meta, "k_spline_grad").items():
        if "k_spline_gradILi4E" not in name and "Rec32EdLb1E" in name:
            assert m["vgpr"] <= 128, (name, m)


def test_wave_sums_use_dpp_not_the_lds_crossbar(kernel_metadata):
    text, _ = kernel_metadata
    start = text.index("k_spline_gradILi0ENS_5Rec32EdLb1ELb0ELb0E")
    body = text[start:text.index(".Lfunc_end", start)]
    assert "row_bcast:31" in body and "ds_bpermute" not in body


def test_wide_histogram_kernel_keeps_the_tile_at_lds_address_zero(kernel_metadata):
    # the WIDE kernel builds a tap's LDS address with one v_perm_b32 on the assumption that its dynamic tile starts at 0
    _, meta = kernel_metadata
    for name, m in _find(meta, "k_spline_histILi0ENS_5Rec32EdLb1E").items():
        assert m["lds_static"] == 0, (name, m)


def test_spline_kernels_use_global_not_flat_memory_instructions(kernel_metadata):
    """Round 3: the multi-pair kernels read their pointers from a device table; a pointer loaded from memory is a generic
    pointer, and every record load, image gather and histogram flush through it became a FLAT instruction (64-bit lane
    addresses, counted on lgkmcnt as well as vmcnt: the waits for the LDS atomics also waited for the record prefetch).
    as_global() (nid_multi.hpp) restores the single-pair kernels' forms.  The one FLAT load the gradient kernels may keep is
    phi(q_r), read from LDS or global memory through one pointer in the prologue."""
    text, meta = kernel_metadata
    checked = 0
    for name in {**_find(meta, "k_spline_hist"), **_find(meta, "k_spline_grad")}:
        start = text.index("\n" + name + ":")
        body = text[start:text.index(".Lfunc_end", start)]
        flat = re.findall(r"^\s+(flat_\w+)", body, flags=re.M)
        # allowed: the system-scope stores of the host mirror (peer and host pointers come from tables too,
        # a handful per workgroup) and the phi(q_r) load; not allowed: record loads (dwordx3 / dwordx4) and the flush's atomics
        hot = [f for f in flat if f in ("flat_load_dwordx3", "flat_load_dwordx4") or f.startswith("flat_atomic_add_x2")]
        assert not hot, (name, hot)
        if re.search(r"Lb1ELb[01]EEEv", name) and "k_spline_hist" in name:  # the multi-pair histogram kernels (<.., MULTI = true, SEG>): nothing FLAT at all
            assert not flat, (name, flat)
        assert re.search(r"^\s+global_load_dwordx4\s+v\[\d+:\d+\], v\d+, s\[\d+:\d+\]", body, flags=re.M), name  # SGPR base + 32-bit lane offset
        checked += 1
    assert checked >= 168  # 6 models x 2 record types x (WIDE / generic) x (single / multi) x (straight / looped) histogram kernels + the gradient kernels (no looped generic one)
