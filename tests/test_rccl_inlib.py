"""RCCL inside the library (include/nidreg.h: nidreg_rccl_unique_id / nidreg_shard_comm_init / nidreg_shard_attach_rccl): the
C test program tests/cxx/test_rccl_world1.cpp builds against the C ABI + librccl (CPU test) and, on a GPU, runs the chain
histogram -> ncclAllReduce(int64) -> entropy -> gradient -> ncclAllReduce(f64 x 7) with one-rank communicators: the
all-reduce of one rank is the identity, so the results must be the plain handle's -- and the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

import parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "direct_visual_lidar_calibration_amd", "csrc")
EXE = os.path.join(ROOT, "tests", "cxx", "test_rccl_world1.bin")


def build_exe():
    import __graft_entry__

    if not os.path.exists(os.path.join(CSRC, "libnidreg.so")):
        __graft_entry__.build()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cxx", "test_rccl_world1.cpp"),
           "-L", CSRC, "-lnidreg", "-L", "/opt/rocm/lib", "-lrccl", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)
    return EXE


STUB = os.path.join(ROOT, "tests", "cxx", "librccl_stub.so")


def build_stub():
    """tests/cxx/rccl_stub.cpp: the six RCCL entry points libnidreg.so resolves, with a shared-memory reduction between PROCESSES --
    lets two ranks share the one GPU of a test box (RCCL refuses two ranks on one device)."""
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "cxx", "rccl_stub.cpp"),
           "-L", "/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread", "-Wl,-rpath,/opt/rocm/lib", "-o", STUB]
    subprocess.check_call(cmd)
    return STUB


def test_rccl_program_compiles_and_links():
    assert os.path.exists(build_exe())
    assert os.path.exists(build_stub())
    import ctypes

    lib = ctypes.CDLL(STUB)
    for sym in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclAllReduce", "ncclGetErrorString"):
        assert hasattr(lib, sym)  # exactly what csrc/nidreg_rccl.hip RcclApi resolves


def test_library_does_not_link_rccl():
    """librccl.so is opened at run time by the first call that needs it: a caller that never attaches a communicator never
    loads it."""
    out = subprocess.check_output(["ldd", os.path.join(CSRC, "libnidreg.so")]).decode()
    assert "rccl" not in out


@pytest.mark.gpu
def test_world1_communicators_return_the_plain_handles_bits(tmp_path):
    import oracle_lib
    from direct_visual_lidar_calibration_amd import se3, synth
    from test_gpu_parity import CAMERAS

    exe = EXE if os.path.exists(EXE) else build_exe()
    for name, bins in (("plumb_bob", 256), ("equirectangular", 16)):
        s = synth.make_scene(CAMERAS[name], num_points=40000, seed=77)
        x = s.T_camera_lidar_init
        intr = np.zeros(5)
        intr[: len(s.intrinsics)] = s.intrinsics
        dist = np.zeros(8)
        dist[: len(s.distortion)] = s.distortion
        path = tmp_path / f"{name}.bin"
        with open(path, "wb") as f:
            f.write(s.model.encode().ljust(64, b"\0"))
            f.write(struct.pack("<6i", s.width, s.height, s.points.shape[0], bins, len(s.intrinsics), len(s.distortion)))
            f.write(intr.tobytes() + dist.tobytes() + np.asarray(x, dtype=np.float64).tobytes() + struct.pack("<d", 1.0) + se3.to_matrix(x).astype(np.float64).tobytes())
            f.write(np.ascontiguousarray(s.image_u8).tobytes())
            f.write(np.ascontiguousarray(s.points, dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(s.intensities, dtype=np.float64).tobytes())
        res = subprocess.run([exe, str(path)], capture_output=True, timeout=300)
        assert res.returncode == 0, (res.returncode, res.stderr.decode()[-2000:])
        v = np.array([float(t) for t in res.stdout.decode().strip().splitlines()[-1].split()]).reshape(3, 8)  # (RCCL prints a version banner first)
        assert v[1, 0] == v[0, 0] and v[2, 0] == v[0, 0]  # the cost: same integers, same bits
        assert np.array_equal(v[1, 1:], v[0, 1:]) and np.array_equal(v[2, 1:], v[0, 1:])  # one rank: the partial IS the gradient
        ref = oracle_lib.nid_cost(s.model, s.intrinsics, s.distortion, s.image_u8.astype(np.float64) / 255.0, s.points, s.intensities, bins, x)
        parity.check_cost(v[1, 0], ref["cost"], what=f"in-library RCCL route, {name}")
        parity.check_grad(v[1, 1:], ref["grad"], what=f"in-library RCCL route, {name}")


@pytest.mark.gpu
def test_inlib_sharded_cost_python_world1():
    """parallel.InLibShardedNIDCost without a process group: a one-rank communicator created by the library."""
    from direct_visual_lidar_calibration_amd import nid, parallel, synth

    s = synth.make_scene("pinhole_vga", num_points=30000, seed=3)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    plain = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 64)
    sh = parallel.InLibShardedNIDCost(proj, s.image_f64, s.points, s.intensities, 64, device=0)
    for x in (s.T_camera_lidar_init, s.T_camera_lidar_true):
        ok, c, g = plain(x)
        ok2, c2, g2 = sh(x)
        assert ok and ok2 and c2 == c and np.array_equal(g2, g)
        ok3, c3, _ = sh(x, want_grad=False)
        assert ok3 and c3 == c
    sh.close()
    plain.close()
    # the derivative-free twin through the same route: histogram -> all-reduce -> entropy tail
    import oracle_lib
    from direct_visual_lidar_calibration_amd import se3

    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    T = se3.to_matrix(s.T_camera_lidar_init)
    a = nid.CostCalculatorNID(proj, s.image_u8, s.points, s.intensities, nid.NIDCostParams(64), max_fov=max_fov)
    b = nid.CostCalculatorNID(proj, s.image_u8, s.points, s.intensities, nid.NIDCostParams(64), max_fov=max_fov)
    b.comm_init(1, 0, nid.NIDCost.rccl_unique_id())
    assert b.calculate(T) == a.calculate(T)
    assert np.array_equal(b.histogram_fixed()[0], a.histogram_fixed()[0])
    a.close()
    b.close()


def _run_two_ranks(tmp_path, case):
    import json
    import sys

    stub = STUB if os.path.exists(STUB) else build_stub()
    tmp_path = tmp_path / case  # (rank 0 leaves the communicator's id in a file: one directory per run)
    os.makedirs(tmp_path)
    env = dict(os.environ, NIDREG_RCCL_LIB=stub, OMP_NUM_THREADS="8", MKL_NUM_THREADS="8")  # (two torch processes each spinning on every core of the box took 40-200 s to build a 60k-point scene)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "run_rccl_stub_rank.py"), str(r), "2", str(tmp_path), case], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se.decode()[-3000:]
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]


@pytest.mark.gpu
def test_two_ranks_all_reduce_through_the_library_chain(tmp_path):
    """nidreg_shard_comm_init -> nidreg_eval with world = 2 and a reduction that is NOT the identity: two processes, each with half
    of the cloud (index slices, desc.scale_points = the pair's total), share the GPU; librccl.so is replaced by a shared-memory stub
    (NIDREG_RCCL_LIB) since RCCL refuses two ranks on one device.  Every rank must return the unsharded handle's cost bit for bit
    (integer histogram, exact all-reduce) and its gradient up to the order of two partial sums; cost-only evaluations and the
    derivative-free twin go through the same chain."""
    from direct_visual_lidar_calibration_amd import nid, se3, synth

    ranks = _run_two_ranks(tmp_path, "spline")
    s = synth.make_scene("pinhole_vga", num_points=60_000, seed=91)
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    rng = np.random.default_rng(4)
    poses = [s.T_camera_lidar_init] + [synth.random_pose_near(s.T_camera_lidar_true, rng) for _ in range(3)]
    plain = nid.NIDCost(proj, s.image_f64, s.points, s.intensities, 64)
    ref = [plain(x) for x in poses]
    plain.close()
    for r in ranks:
        assert r["attached"] and all(r["ok"])
        assert r["costs"] == [v[1] for v in ref] and r["cost_only"] == [v[1] for v in ref]
        assert np.allclose(np.array(r["grads"]), np.array([v[2] for v in ref]), rtol=1e-12, atol=1e-15)
    assert ranks[0]["grads"] == ranks[1]["grads"]  # the same reduced 7 doubles on every rank
    import oracle_lib

    near = _run_two_ranks(tmp_path, "nearest")
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    a = nid.CostCalculatorNID(proj, s.image_u8, s.points, s.intensities, nid.NIDCostParams(64), max_fov=max_fov)
    want = [a.calculate(se3.to_matrix(x)) for x in poses]
    total = int(a.histogram_fixed()[0].sum())
    a.close()
    for r in near:
        assert r["costs"] == want and r["hist_sum"] == total


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mismatch_unit", "mismatch_bins"])
def test_ranks_that_disagree_on_the_table_are_refused_on_every_rank(tmp_path, case):
    """a rank created without desc.scale_points (another fixed-point unit) or with other bins would add incompatible
    integers into a cost that is wrong yet identical everywhere.  nidreg_shard_comm_init max-/min-reduces the table parameters
    once and refuses on every rank alike."""
    ranks = _run_two_ranks(tmp_path, case)
    for r in ranks:
        assert not r["attached"] and "disagree" in r["error"], r
