"""TEST / BENCH INFRASTRUCTURE: the whole `calibrate` outer loop (visual_camera_calibration.cpp:35-68 -> :190-238 / :70-139) run
twice on identical in-memory pairs -- once on the GPU engine (product code: calibrate.calibrate_pairs), once on the CPU oracle
(the reference's cost functors restated, optionally with the OpenMP split over points) under the SAME host driver -- and the
distance between the two final extrinsics (BASELINE.json: <= 1e-3 m / 1e-3 rad).

Used by tests/test_time_to_solution.py (-m gpu) and by bench.py's `time_to_solution` key (a CPU-baseline leg: the only place the
bench touches the oracle besides `cpu_baseline`).  BFGS: Ceres is absent, so both sides run calibration.bfgs_minimize; the
Nelder-Mead route can be checked against the reference's own calibrate() (oracle/_ref/libref.so) where that travelled along."""
import time

import numpy as np

import oracle_lib
from direct_visual_lidar_calibration_amd import calibrate, calibration, nid, se3, synth

# BASELINE.json configs[0] / configs[1] as synth scenes
CONFIGS = {
    "configs0": dict(camera="pinhole_vga", points=100_000, bins=16, seed=20250523 + 1),
    "configs1": dict(camera="pinhole_1080p", points=10_000_000, bins=256, seed=20250523 + 2),
}


class CpuBudgetExceeded(RuntimeError):
    pass


def _charge(counter, key):
    counter[key] += 1
    if counter.get("deadline") and time.perf_counter() > counter["deadline"]:
        raise CpuBudgetExceeded(f"the CPU side used up its wall-clock budget after {counter['nid'] + counter['nearest']} evaluations")


class _CountedOracleNID:
    def __init__(self, s, img64, pts, ints, bins, threads, counter):
        self.a = (s.model, s.intrinsics, s.distortion, img64, pts, ints, bins)
        self.threads, self.counter = threads, counter

    def __call__(self, x, want_grad=True):
        _charge(self.counter, "nid")
        r = oracle_lib.nid_cost(*self.a, x, want_grad=want_grad, threads=self.threads)
        return r["ok"], r["cost"], r["grad"]


class _CountedOracleNearest:
    def __init__(self, s, img8, pts, ints, bins, max_fov, counter):
        self.a = (s.model, s.intrinsics, s.distortion, img8, pts, ints, bins, max_fov)
        self.counter = counter

    def calculate(self, T):
        _charge(self.counter, "nearest")
        return oracle_lib.cost_calculator_nid(*self.a, T)[0]


def make_params(reg, bins, **kw):
    return calibration.VisualCameraCalibrationParams(nid_bins=bins, registration_type=reg, **kw)


def cpu_calibrate(scenes, init_x, params, threads=1, budget_s=None):
    """The same host driver on the CPU oracle: culling (view_culling.cpp) + cost objects per outer iteration, serial over
    points like the reference (threads = 1) or with the oracle's OpenMP split over points."""
    s = scenes[0]
    max_fov = oracle_lib.estimate_camera_fov(s.model, s.intrinsics, s.distortion, s.width, s.height)
    counter = {"nid": 0, "nearest": 0, "deadline": (time.perf_counter() + budget_s) if budget_s else None}
    cal = calibration.VisualCameraCalibration(
        [(sc.image_u8, sc.points, sc.intensities) for sc in scenes], params,
        nid_cost_factory=lambda i, pt, it, b: _CountedOracleNID(s, i, pt, it, b, threads, counter),
        nearest_cost_factory=lambda i, pt, it, b: _CountedOracleNearest(s, i, pt, it, b, max_fov, counter),
        cull=lambda pts, ints, T: oracle_lib.view_culling(s.model, s.intrinsics, s.distortion, s.width, s.height, pts, T, not params.disable_z_buffer_culling))
    t0 = time.perf_counter()
    x = cal.calibrate(init_x)
    return x, time.perf_counter() - t0, counter, cal.log


def gpu_calibrate(scenes, init_x, params, repeats=2):
    """Product path, `repeats` times (the first run of a process pays code-object loading and arena growth)."""
    s = scenes[0]
    proj = nid.create_camera(s.model, s.intrinsics, s.distortion)
    pairs = [(sc.image_u8, sc.points, sc.intensities) for sc in scenes]
    walls, stats, x = [], {}, None
    for _ in range(repeats):
        stats = {}
        t0 = time.perf_counter()
        x, cal = calibrate.calibrate_pairs(proj, pairs, init_x, params, stats=stats)
        walls.append(time.perf_counter() - t0)
    return x, walls, stats, cal.log


def compare(config, reg, bags=1, points=None, threads=1, repeats=2, device="cpu", cpu_budget_s=None, **param_kw):
    """One time-to-solution record: {config, gpu_wall_s, evals, setup_s, cpu_wall_s, dT, ...}."""
    c = CONFIGS[config]
    n = int(points or c["points"])
    scenes = [synth.make_scene(c["camera"], num_points=n, seed=c["seed"] + 100 * k, device=device, init_delta=(0.02, 0.4)) for k in range(bags)]
    init_x = scenes[0].T_camera_lidar_init
    params = make_params(reg, c["bins"], **param_kw)
    x_gpu, walls, stats, gpu_log = gpu_calibrate(scenes, init_x, params, repeats=repeats)
    s = scenes[0]
    head = {
        "config": f"BASELINE {config}: {bags} pair(s) x {n} pts, {s.width}x{s.height} {s.model}, {c['bins']} bins, {reg}",
        "registration_type": reg, "bags": bags, "points_per_bag": n, "bins": c["bins"],
        "bounds": {k: v for k, v in param_kw.items()} or None,
        "gpu_wall_s": [round(v, 4) for v in walls],
        "evals": stats.get("evaluations"), "outer_iterations": stats.get("outer_iterations"),
        "setup_s": round(stats.get("upload_s", 0.0) + stats.get("build_s", 0.0), 4),
        "setup_split_s": {"upload": round(stats.get("upload_s", 0.0), 4), "cull_and_build": round(stats.get("build_s", 0.0), 4)},
    }
    try:
        x_cpu, cpu_wall, counter, cpu_log = cpu_calibrate(scenes, init_x, make_params(reg, c["bins"], **param_kw), threads=threads, budget_s=cpu_budget_s)
    except CpuBudgetExceeded as exc:
        head.update(cpu_wall_s=None, cpu_threads=threads, dT=None, note=str(exc))
        return head
    dt, dr = se3.delta_trans_rot(x_cpu, x_gpu)
    dt0, dr0 = se3.delta_trans_rot(scenes[0].T_camera_lidar_true, init_x)
    dt1, dr1 = se3.delta_trans_rot(scenes[0].T_camera_lidar_true, x_gpu)
    head.update({
        "cpu_wall_s": round(cpu_wall, 2), "cpu_threads": threads, "cpu_evals": counter["nid"] + counter["nearest"], "cpu_outer_iterations": len(cpu_log),
        "cpu_kind": "the same host driver on the oracle (BFGS: Ceres absent, calibration.bfgs_minimize on both sides)",
        "speedup": round(cpu_wall / float(np.median(walls[1:] or walls)), 1),  # the median of the repeats after the first (which pays code-object loading)
        "dT": [dt, dr], "dT_unit": "m, rad (GPU vs CPU final T_camera_lidar)",
        "error_vs_truth_before": [dt0, dr0], "error_vs_truth_after": [dt1, dr1],
    })
    return head
