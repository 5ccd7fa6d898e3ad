import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from direct_visual_lidar_calibration_amd import nid, synth
scenes = [synth.make_scene("pinhole_vga", num_points=n, seed=sd) for n, sd in ((20000, 7), (6000, 8))]
proj = nid.create_camera(scenes[0].model, scenes[0].intrinsics, scenes[0].distortion)
x = scenes[0].T_camera_lidar_init
for bins in (16, 256):
    handles = [nid.NIDCost(proj, sc.image_f64, sc.points, sc.intensities, bins) for sc in scenes]
    singles = [h(x) for h in handles]
    multi = nid.MultiNIDCost(None)
    for h in handles: multi.add(h)
    try:
        print(bins, multi(x), sum(s[1] for s in singles))
    except Exception as e:
        print(bins, "ERR", e)
    for h in handles: h.close()
