"""``calibrate`` on the MI355X engine: reads a preprocessed directory, refines ``T_lidar_camera`` with the
NID cost on the GPU and writes ``results.T_lidar_camera`` back into ``calib.json``.

Mirrors the reference executable (src/calibrate.cpp:27-200) without its viewer:

    python -m direct_visual_lidar_calibration_amd.calibrate <data_path> [--nid_bins 16]
        [--registration_type nid_bfgs|nid_nelder_mead] [--disable_culling] [--first_n_bags N]
        [--nelder_mead_init_step 1e-3] [--nelder_mead_convergence_criteria 1e-8] [--auto_quit] [--background]

Same option names, defaults and ``calib.json`` keys.  The per-outer-iteration work (view culling ->
new cost object, visual_camera_calibration.cpp:76-85 / :201-206) and every cost evaluation run on the
GPU from device-resident clouds; pairs are spread over the visible GPUs (pair k -> device k mod P), the
plain sum of ``MultiNIDCost`` happening on the host.  The optimisers (BFGS / Nelder-Mead) stay on the
host like the reference's Ceres / dfo.
"""
import argparse
import math
import sys
import time

import numpy as np

from . import _lib, calibration, dataset, nid, se3


def build_parser():
    p = argparse.ArgumentParser(prog="calibrate", description="calibrate")
    p.add_argument("data_path", help="directory that contains preprocessed data")
    p.add_argument("--first_n_bags", type=int, default=None, help="use only the first N bags (just for evaluation)")
    p.add_argument("--disable_culling", action="store_true", help="disable depth buffer-based hidden points removal")
    p.add_argument("--nid_bins", type=int, default=16, help="Number of histogram bins for NID")
    p.add_argument("--registration_type", default="nid_bfgs", help="nid_bfgs or nid_nelder_mead")
    p.add_argument("--nelder_mead_init_step", type=float, default=1e-3, help="Nelder-mead initial step size")
    p.add_argument("--nelder_mead_convergence_criteria", type=float, default=1e-8, help="Nelder-mead convergence criteria")
    p.add_argument("--auto_quit", action="store_true", help="accepted for compatibility (there is no viewer to keep open)")
    p.add_argument("--background", action="store_true", help="accepted for compatibility (there is no viewer to hide)")
    p.add_argument("--precision", default="fp64", choices=["fp64"], help="geometry precision of the cost kernels (extension; the reference is fp64)")
    p.add_argument("--dry_run", action="store_true", help="load and validate the dataset, do not optimise or write (extension)")
    return p


class _Multi:
    """MultiNIDCost over handles that may live on several GPUs (nid.MultiNIDCost -> nidreg_eval_multi)."""

    def __init__(self, init_x, costs):
        self.m = nid.MultiNIDCost(init_x)
        for c in costs:
            self.m.add(c)

    def __call__(self, x, want_grad=True):
        return self.m(x, want_grad)


def calibrate_pairs(proj, pairs, init_x, params, precision="fp64", stats=None):
    """The in-memory body of the command: ``pairs`` = [(image_u8 (H, W), points (N, 4), intensities (N,)), ...] as
    ``VisualLiDARData`` holds them, ``init_x`` the Sophus-order ``T_camera_lidar`` 7-vector.  One upload per pair; every outer
    iteration culls + rebuilds its cost object on the device (visual_camera_calibration.cpp:76-85 / :201-206); every cost
    evaluation runs on the GPU.  Returns ``(x, VisualCameraCalibration)``; ``stats`` (optional dict) receives the wall-clock
    split: ``upload_s`` (clouds to HBM), ``build_s`` (culling + record build of every outer iteration), ``evaluations``."""
    ndev = _lib.load().nidreg_device_count()
    if ndev <= 0:
        raise SystemExit("error: no MI355X / HIP device visible (the NID core has no CPU fallback)")
    t0 = time.perf_counter()
    clouds = [nid.Cloud(p[1], p[2], device=k % ndev) for k, p in enumerate(pairs)]
    images_f64 = [p[0].astype(np.float64) * (1.0 / 255.0) for p in pairs]  # convertTo(CV_64FC1, 1/255), :204
    upload_s = time.perf_counter() - t0
    size = (pairs[0][0].shape[1], pairs[0][0].shape[0])
    max_fov = nid.estimate_camera_fov(proj, size)
    min_z = math.cos(max_fov)
    depth = not params.disable_z_buffer_culling
    build_s = [0.0]

    def fused_nid(k, T, bins):
        t1 = time.perf_counter()
        c = nid.NIDCost.from_cloud(proj, images_f64[k], clouds[k], bins, cull=(T, min_z, depth), precision=precision)
        build_s[0] += time.perf_counter() - t1
        return c

    def fused_nearest(k, T, bins):
        t1 = time.perf_counter()
        c = nid.CostCalculatorNID.from_cloud(proj, pairs[k][0], clouds[k], nid.NIDCostParams(bins), max_fov=max_fov, cull=(T, min_z, depth), precision=precision)
        build_s[0] += time.perf_counter() - t1
        return c

    cal = calibration.VisualCameraCalibration(
        [(p[0], None, None) for p in pairs], params, fused_nid_factory=fused_nid, fused_nearest_factory=fused_nearest, multi_factory=lambda init, costs: _Multi(init, costs))
    x = cal.calibrate(init_x)
    for c in clouds:
        c.close()
    if stats is not None:
        stats.update(devices=min(ndev, len(pairs)), upload_s=upload_s, build_s=build_s[0], evaluations=int(sum(e.get("evaluations", 0) for e in cal.log)),
                     outer_iterations=len(cal.log))
    return x, cal


def run(args, log=print):
    config, bags = dataset.load_dataset(args.data_path, args.first_n_bags)
    if args.first_n_bags is not None:
        log(f"use only the first {args.first_n_bags} bags")
    model, intrinsics, distortion = dataset.camera_from_calib(config)
    proj = nid.create_camera(model, intrinsics, distortion)
    if proj is None:
        raise SystemExit(f"error: unknown camera model / wrong number of intrinsics: {model}")
    init_values, key = dataset.init_T_lidar_camera(config)
    if init_values is None:
        raise SystemExit("error: initial guess of T_lidar_camera must be computed before calibration!!")
    log("use manually estimated initial guess" if key == "init_T_lidar_camera" else "use automatically estimated initial guess")
    init_x = dataset.tum_to_T_camera_lidar(init_values)

    if args.registration_type not in ("nid_bfgs", "nid_nelder_mead"):
        log(f"warning: unknown registration type {args.registration_type}")  # the reference keeps its default, NID_BFGS
        args.registration_type = "nid_bfgs"
    params = calibration.VisualCameraCalibrationParams(
        disable_z_buffer_culling=args.disable_culling, nid_bins=args.nid_bins, registration_type=args.registration_type, nelder_mead_init_step=args.nelder_mead_init_step,
        nelder_mead_convergence_criteria=args.nelder_mead_convergence_criteria)
    for b in bags:
        log(f"loaded {b.bag_name}: image {b.image.shape[1]}x{b.image.shape[0]}, {b.points.shape[0]} points")
    if args.dry_run:
        return config, init_x, None

    stats = {}
    t0 = time.time()
    x, cal = calibrate_pairs(proj, [(b.image, b.points, b.intensities) for b in bags], init_x, params, precision=args.precision, stats=stats)
    elapsed = time.time() - t0
    ndev = stats["devices"]
    for entry in cal.log:
        log(f"outer {entry['outer']}: {entry['inner']} cost {entry.get('initial_cost', float('nan')):.6f} -> {entry['final_cost']:.6f}, "
            f"delta {entry['delta_t']:.5f} m / {entry['delta_r'] * 180.0 / math.pi:.4f} deg, {entry.get('evaluations', 0)} evaluations")

    values = dataset.T_camera_lidar_to_tum(x)
    config.setdefault("results", {})["T_lidar_camera"] = values
    dataset.write_calib(args.data_path, config)
    log("--- T_lidar_camera ---")
    log(str(np.linalg.inv(se3.to_matrix(x))))
    log(f"saved to {args.data_path}/calib.json  ({elapsed:.2f} s on {ndev} GPU(s))")
    return config, init_x, x


def main(argv=None):
    args = build_parser().parse_args(argv)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
