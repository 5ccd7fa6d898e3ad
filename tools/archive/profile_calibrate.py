#!/usr/bin/env python3
"""cProfile of `calibrate.run` on a synthetic preprocessed directory (configs[0]: 100k-point bags, VGA pinhole, 16 bins): where the
host side of a whole calibration spends its time once an evaluation costs 26 us.  Usage: profile_calibrate.py [bags] [registration_type]"""
import cProfile
import io
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_visual_lidar_calibration_amd import calibrate, dataset, synth  # noqa: E402

bags_n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reg = sys.argv[2] if len(sys.argv) > 2 else "nid_bfgs"
d = tempfile.mkdtemp()
scenes = [synth.make_scene("pinhole_vga", num_points=100_000, seed=61 + k, init_delta=(0.02, 0.4)) for k in range(bags_n)]
s0 = scenes[0]
dataset.write_preprocessed(d, (s0.model, s0.intrinsics, s0.distortion), [(f"bag{k}", s.image_u8, s.points, s.intensities) for k, s in enumerate(scenes)],
                           init_T_lidar_camera_tum=dataset.T_camera_lidar_to_tum(s0.T_camera_lidar_init), meta={"image_topic": "/image", "points_topic": "/points", "intensity_channel": "intensity"})
args = calibrate.build_parser().parse_args([d, "--registration_type", reg, "--auto_quit", "--background"])
for _ in range(2):
    t0 = time.perf_counter()
    calibrate.run(args, log=lambda *_: None)
    print("run: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
calibrate.run(args, log=lambda *_: None)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print(out.getvalue())
