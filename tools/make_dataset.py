#!/usr/bin/env python3
"""Write a synthetic preprocessed directory (calib.json, <bag>.png, <bag>.ply) in the format `preprocess` produces, so
that `python -m direct_visual_lidar_calibration_amd.calibrate <dir>` can be tried without a ROS bag.
Usage: make_dataset.py out_dir [camera=pinhole_vga] [points=100000] [bags=1] [seed=1]
The initial guess stored under results.init_T_lidar_camera is the true extrinsic perturbed by <= 3 cm / 0.5 deg."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import dataset, synth  # noqa: E402

out = sys.argv[1]
camera = sys.argv[2] if len(sys.argv) > 2 else "pinhole_vga"
points = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
bags = int(sys.argv[4]) if len(sys.argv) > 4 else 1
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 1
scenes = [synth.make_scene(camera, num_points=points, seed=seed + k) for k in range(bags)]
s0 = scenes[0]
dataset.write_preprocessed(out, (s0.model, s0.intrinsics, s0.distortion), [(f"bag{k:02d}", s.image_u8, s.points, s.intensities) for k, s in enumerate(scenes)],
                           init_T_lidar_camera_tum=dataset.T_camera_lidar_to_tum(s0.T_camera_lidar_init), meta={"image_topic": "/synthetic/image", "points_topic": "/synthetic/points"})
print("wrote", out, "true T_lidar_camera (tx ty tz qx qy qz qw):", dataset.T_camera_lidar_to_tum(s0.T_camera_lidar_true))
