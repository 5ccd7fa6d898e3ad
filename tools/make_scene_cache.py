#!/usr/bin/env python3
"""Generate a synthetic scene (torch, on the GPU if present) and cache it as .npz so that profiled
runs need not import torch (rocprofv3 --pmc crashes under torch's datagen kernels on this pool).
Usage: make_scene_cache.py out.npz [camera] [points] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from direct_visual_lidar_calibration_amd import synth  # noqa: E402

out = sys.argv[1]
camera = sys.argv[2] if len(sys.argv) > 2 else "pinhole_1080p"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 20250525
dev = "cuda:0" if torch.cuda.is_available() else "cpu"
s = synth.make_scene(camera, num_points=n, seed=seed, device=dev)
np.savez(out, model=s.model, intrinsics=np.array(s.intrinsics), distortion=np.array(s.distortion), width=s.width, height=s.height, image_u8=s.image_u8,
         points=s.points.astype(np.float32), intensities=s.intensities.astype(np.float32), T_true=s.T_camera_lidar_true, T_init=s.T_camera_lidar_init)
print("wrote", out, s.points.shape)
